#!/usr/bin/env python3
"""conv2d forward at the phase4 backbone's 3x3 / strided shapes: this library (NHWC implicit GEMM, bf16x6 =
fp32-grade) beside PyTorch-ROCm / MIOpen (channels_last fp32 and bf16) on the same GPU.
    python tools/bench_conv.py [--B 64]"""
import argparse, importlib, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")


def timeit(f, iters=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=64); a = ap.parse_args()
    B = a.B
    shapes = [("layer2 conv2 3x3 s1", 32, 128, 128, 3, 1, 1), ("layer3 conv2 3x3 s1", 16, 256, 256, 3, 1, 1),
              ("layer4 conv2 3x3 s1", 8, 512, 512, 3, 1, 1), ("layer3.0 conv2 3x3 s2", 32, 256, 256, 3, 2, 1),
              ("layer3.0 downsample 1x1 s2", 32, 512, 1024, 1, 2, 0), ("layer2 conv3 1x1", 32, 128, 512, 1, 1, 0)]
    for name, H, Cin, Cout, k, s, p in shapes:
        x = torch.randn(B, H, H, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
        wo = pkg.conv.to_ohwi(w)
        Ho = (H + 2 * p - k) // s + 1
        flops = 2.0 * B * Ho * Ho * Cout * Cin * k * k
        t_ours = timeit(lambda: pkg.conv.conv2d_nhwc(x, wo, s, p))
        xc = x.permute(0, 3, 1, 2)                      # NCHW view of NHWC memory = channels_last
        wc = w.contiguous(memory_format=torch.channels_last)
        t_f32 = timeit(lambda: F.conv2d(xc, wc, stride=s, padding=p))
        xb, wb = xc.bfloat16(), wc.bfloat16()
        t_bf = timeit(lambda: F.conv2d(xb, wb, stride=s, padding=p))
        print(f"{name:28s} B={B} {H}x{H} {Cin}->{Cout}: ours(bf16x6) {t_ours:8.1f} us = {flops / t_ours / 1e6:6.1f} TF | "
              f"MIOpen fp32 {t_f32:8.1f} us = {flops / t_f32 / 1e6:6.1f} TF | MIOpen bf16 {t_bf:8.1f} us = {flops / t_bf / 1e6:6.1f} TF")


if __name__ == "__main__":
    main()
