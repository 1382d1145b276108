#!/bin/bash
# round 3: the persistent planes GEMM (one workgroup per CU walking a run of output tiles) against one workgroup per tile,
# same box, interleaved: conv-path tests first, then the Model_3D train step at B = 256 and B = 32 and the inference forward
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -x > $O/r3_persist_tests.log 2>&1; rc=$?; tail -5 $O/r3_persist_tests.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2; do
  for v in 1 0; do
    echo "== POSELIFT_PERSIST=$v (rep $rep)"
    POSELIFT_PERSIST=$v timeout -k 10 300 python tools/bench_model3d_train.py --B 256 --iters 5 --skip-eager 2>&1 | grep -v Warning | tail -3
  done
done
for v in 1 0; do
  echo "== POSELIFT_PERSIST=$v B=32"
  POSELIFT_PERSIST=$v timeout -k 10 300 python tools/bench_model3d_train.py --B 32 --iters 10 --skip-eager 2>&1 | grep -v Warning | tail -3
done
