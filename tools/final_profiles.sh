#!/bin/bash
# Round-end measurement pass on the GPU box (run from the repo root): full GPU test suite, smoke(), the default
# bench line, then rocprofv3 kernel stats of the bench and of the Model_3D training / inference loops.
# Outputs under gpurun_out/final/; the summaries worth keeping are copied into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python3 $R/bench.py --steps 60 --warmup 10 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python3 $R/tools/run_model3d_train.py 32 3 > $O/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o i -- python3 $R/tools/run_model3d.py 64 5 > $O/infer.log 2>&1
cd $R
python tools/bench_model3d_train.py --B 32 --iters 3 > $O/train_b32.txt 2>/dev/null
python tools/bench_model3d_train.py --B 256 --iters 2 > $O/train_b256.txt 2>/dev/null
echo final-profiles-done
