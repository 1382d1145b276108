#!/bin/bash
# rocprofv3 kernel trace of three Model_3D training steps at B = 256, grouped by (kernel, grid): $1 = tag, rest = env
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/r3_shape_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $R/tools/run_model3d_train.py 256 4 f16x3 > /dev/null 2>&1; echo prof rc=$?
cd $R
python tools/trace_by_shape.py $(find $O/prof -name "*kernel_trace.csv") 4 > $O/by_shape.txt
find $O/prof -name "*.csv" -delete
head -50 $O/by_shape.txt
