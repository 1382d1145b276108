#!/bin/bash
# Round-2 conv-path measurement pass (from the repo root on the GPU box): Model_3D training step at B = 256 / 32 / 8 beside
# PyTorch-ROCm eager, the phase5 cycle step, and rocprofv3 kernel stats of three B = 256 steps per storage mode.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/conv_r02
rm -rf $O; mkdir -p $O
cd $R
python tools/bench_model3d_train.py --B 256 --iters 5 > $O/m3d_b256.txt 2>&1; echo b256 rc=$?
python tools/bench_model3d_train.py --B 32 --iters 10 --graph > $O/m3d_b32.txt 2>&1; echo b32 rc=$?
python tools/bench_model3d_train.py --B 8 --iters 20 --graph > $O/m3d_b8.txt 2>&1; echo b8 rc=$?
python tools/bench_cycle.py --B 128 > $O/cycle_b128.txt 2>&1; echo cycle rc=$?
cd /tmp && export TMPDIR=/tmp
for dt in f16x3 bf16p; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$dt -o t -- python3 $R/tools/run_model3d_train.py 256 3 $dt > /dev/null 2>&1; echo prof $dt rc=$?
  cp $(find $O/prof_$dt -name "*kernel_stats.csv") $O/m3d_b256_${dt}_kernel_stats.csv
  find $O/prof_$dt -name "*kernel_trace.csv" -delete
done
cd $R
grep -h "B=" $O/m3d_b256.txt $O/m3d_b32.txt $O/m3d_b8.txt $O/cycle_b128.txt
echo conv-profiles-done
