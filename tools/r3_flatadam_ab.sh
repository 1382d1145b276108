#!/bin/bash
# round 3: arena.FlatAdam (gradients written into flat arenas, one Adam launch) against torch.optim.Adam on the same box
O=gpurun_out
python -m pytest tests/test_gpu_cycle.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
  for v in "" "--torch-adam"; do
    python tools/bench_model3d_train.py --B 256 --iters 5 --skip-eager $v 2>&1 | grep -E "optimizer|planes GEMM"
  done
done
for v in "" "--torch-adam"; do
  python tools/bench_model3d_train.py --B 8 --iters 20 --skip-eager --graph $v 2>&1 | grep -E "optimizer|planes"
done
