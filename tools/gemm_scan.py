import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
L = pkg.lib(); s = torch.cuda.current_stream().cuda_stream
def t(layout, M, N, K, iters=30):
    A = torch.randn((K, M) if layout == 2 else (M, K), device="cuda")
    B = torch.randn((N, K) if layout == 0 else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    f = lambda: L.pl_gemm_f32(layout, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, None, 1, None, s)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for layout in (0, 1):
    for K in (128, 256, 512, 1024, 2048, 4096):
        us = t(layout, 4096, 1024, K)
        print(f"layout {layout} M=4096 N=1024 K={K:5d}: {us:8.1f} us  per-32k-tile {us / (K / 32):6.3f} us  {2*4096*1024*K/us/1e6:6.1f} TF")
for M in (2048, 4096, 8192, 16384):
    us = t(0, M, 1024, 1024)
    print(f"layout 0 M={M} N=1024 K=1024: {us:8.1f} us {2*M*1024*1024/us/1e6:6.1f} TF  blocks {M//128*8}")
