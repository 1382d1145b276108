#!/usr/bin/env python3
"""Run ONE GEMM shape many times (profiling target): python3 tools/gemm_one.py ARITH LAYOUT [ITERS]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
L = pkg.lib(); s = torch.cuda.current_stream().cuda_stream
arith, layout = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
M, N, K = (1024, 1024, 4096) if layout == 2 else (4096, 1024, 1024)
A = torch.randn((K, M) if layout == 2 else (M, K), device="cuda")
B = torch.randn((N, K) if layout == 0 else (K, N), device="cuda")
C = torch.empty(M, N, device="cuda")
for _ in range(iters):
    L.pl_gemm_arith(layout, arith, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, None, 1, None, s)
torch.cuda.synchronize()
print("done")
