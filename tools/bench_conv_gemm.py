#!/usr/bin/env python3
"""The conv path's 1x1-convolution GEMM shapes through pl_gemm_planes_raw (NT, f16x3 planes, BatchNorm statistics in the
epilogue), timed back to back: what the persistent form (POSELIFT_PERSIST, round 3) does per shape, with the bytes each
launch has to move and its FLOPs.
    python tools/bench_conv_gemm.py [iters]          (run once per POSELIFT_PERSIST value for the A/B)"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
L = pkg.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
# (pixels M, Cout N, Cin K): ResNet-50 1x1 convolutions of a B = 256, 256 x 256 training step (forward shapes; the data
# gradients are the same shapes with N and K exchanged)
SHAPES = [(1 << 20, 256, 64), (1 << 20, 64, 256), (1 << 20, 64, 64), (1 << 18, 512, 128), (1 << 18, 128, 512), (1 << 18, 128, 256),
          (1 << 16, 1024, 256), (1 << 16, 256, 1024), (1 << 14, 2048, 512), (1 << 14, 512, 2048), (1 << 20, 1088, 256)]
print(f"POSELIFT_PERSIST={os.environ.get('POSELIFT_PERSIST', '(default 1)')}")
for M, N, K in SHAPES:
    A = (torch.randn(2, M, K, device=dev) * 0.5).half()
    W = (torch.randn(2, N, K, device=dev) * 0.5).half()
    C = torch.empty(M, N, device=dev)
    G = L.pl_gemm_stat_groups(M)
    stat = torch.empty(2 * G * N, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        rc = L.pl_gemm_planes_raw(0, 3, A.data_ptr(), M * K, K, W.data_ptr(), N * K, K, C.data_ptr(), M, N, K, None, 1.0, None,
                                  None, stat.data_ptr(), s)
        assert rc == 0, L.pl_last_error()
    for _ in range(3):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / iters * 1e6
    byt = M * K * 4 + N * K * 4 + M * N * 4
    print(f"M={M:8d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF  {byt / us / 1e6:6.2f} TB/s of the {byt / 1e6:7.1f} MB it moves")
    del A, W, C, stat
