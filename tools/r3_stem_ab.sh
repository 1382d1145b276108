#!/bin/bash
# round 3: the stem on the planes GEMM (conv.stem_planes) against the direct fp32 kernel, same box; conv-path tests first
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_cycle.py -m gpu -q > gpurun_out/stem_tests.log 2>&1; tail -3 gpurun_out/stem_tests.log
for rep in 1 2; do
  for v in 0 1; do
    echo "POSELIFT_STEM_PLANES=$v"
    POSELIFT_STEM_PLANES=$v python tools/bench_model3d_train.py --B 256 --iters 5 --skip-eager 2>&1 | grep -E "planes GEMM|planes\)" | head -3
  done
done
O=$GRAFT_REPO_ROOT/gpurun_out/stemprof; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/run_model3d_train.py 256 4 f16x3 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT && python tools/trace_by_shape.py $(find $O/prof -name "*kernel_trace.csv") 4 > $O/by_shape.txt; find $O/prof -name "*.csv" -delete; grep -E "total|true, true, 2, true" $O/by_shape.txt
