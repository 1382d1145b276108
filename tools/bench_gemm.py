#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM (pl_gemm_f32) at the lifter's shapes, HIP-event timed.
    python tools/bench_gemm.py [--iters 50]
Random data (zeros read high: MI355X_MICROARCH 'DVFS give-back')."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--arith", type=int, default=0, help="0 fp32, 1 bf16, 2 bf16x6, 5 bf16x6 planes, 6 bf16x6 fragment split")
    a = ap.parse_args()
    L, dev = pkg.lib(), "cuda:0"
    B, H = a.B, 1024
    shapes = [("fwd  NT z=aW^T", 0, B, H, H, 1), ("dX   NN da=dzW", 1, B, H, H, 1),
              ("dW   TN dW=dz^Ta splitK4", 2, H, H, B, 4), ("dW   TN no split", 2, H, H, B, 1)]
    s = torch.cuda.current_stream().cuda_stream
    for name, layout, M, N, K, sk in shapes:
        A = torch.randn((K, M) if layout == 2 else (M, K), device=dev)
        Bm = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        slabs = torch.empty(sk, M, N, device=dev) if sk > 1 else None

        def run():
            rc = L.pl_gemm_arith(layout, a.arith, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, None, sk,
                                 slabs.data_ptr() if sk > 1 else None, s)
            assert rc == 0, L.pl_last_error()
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"{name:28s} {M}x{N}x{K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s "
              f"({2.0 * M * N * K / us / 1e6 / 157.3 * 100:4.1f}% of fp32-matrix peak)")


if __name__ == "__main__":
    main()
