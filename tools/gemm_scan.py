#!/usr/bin/env python3
"""K scan of the 128x128-tile GEMM at M=4096, N=1024: separates the fixed cost of a launch from the
per-K-tile steady state, per arithmetic mode (0 fp32, 1 bf16, 5 bf16x6 planes, 6 bf16x6 fragment split).
    python tools/gemm_scan.py [arith ...]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
L = pkg.lib(); s = torch.cuda.current_stream().cuda_stream


def t(arith, layout, M, N, K, iters=40):
    A = torch.randn((K, M) if layout == 2 else (M, K), device="cuda")
    B = torch.randn((N, K) if layout == 0 else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    f = lambda: L.pl_gemm_arith(layout, arith, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, None, 1, None, s)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for arith in [int(a) for a in sys.argv[1:]] or [0, 1, 6, 5]:
    for layout in (0, 1):
        ts = {K: t(arith, layout, 4096, 1024, K) for K in (256, 512, 1024, 2048, 4096)}
        slope = (ts[4096] - ts[1024]) / ((4096 - 1024) / 32)
        print(f"arith {arith} layout {layout}: " + "  ".join(f"K={K}: {u:6.1f}" for K, u in ts.items()) +
              f"  | steady {slope * 1e3:6.0f} ns per 32-k tile, fixed {ts[1024] - 32 * slope:5.1f} us (incl. launch gap)")
