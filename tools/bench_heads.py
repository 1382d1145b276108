#!/usr/bin/env python3
"""Soft-argmax head at BASELINE config 4's shape: B frames x 17 joints x 64^3 logits.
HIP-event timed; reports achieved HBM GB/s (algorithmic bytes: 4 B/voxel forward, 8 B/voxel backward)
next to stock PyTorch-ROCm eager of the reference formulas."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")


def eager(out, J, D):
    B, C, H, W = out.shape
    hm = torch.softmax(out.reshape(B, J, -1), 2)
    hm = (hm / hm.sum(2, keepdim=True)).reshape(B, J, D, H, W)
    ar = torch.arange(64, dtype=out.dtype, device=out.device)
    cx = (hm.sum((2, 3)) * ar).sum(2, keepdim=True)
    cy = (hm.sum((2, 4)) * ar).sum(2, keepdim=True)
    cz = (hm.sum((3, 4)) * ar).sum(2, keepdim=True)
    return torch.cat(((cx / W - 0.5) * 2, (cy / H - 0.5) * 2, (cz / D - 0.5) * 2), 2).reshape(B, -1)


def timeit(f, n):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    J, D = 17, 64
    x = torch.randn(a.B, J * D, 64, 64, device="cuda") * 4
    nbytes = x.numel() * 4
    g = torch.randn(a.B, J * 3, device="cuda")
    xr = x.clone().requires_grad_(True)

    ms_f = timeit(lambda: pkg.soft_argmax_3d(x, J, D), a.iters)

    def fb():
        xr.grad = None
        pkg.soft_argmax_3d(xr, J, D).backward(g)
    ms_fb = timeit(fb, a.iters)
    print(f"B={a.B}: logits {nbytes / 1e9:.2f} GB")
    print(f"  HIP fused forward      {ms_f:8.3f} ms  {nbytes / ms_f / 1e6:8.0f} GB/s  ({nbytes / ms_f / 1e6 / 8000 * 100:.0f}% of 8 TB/s)")
    print(f"  HIP fused fwd+bwd      {ms_fb:8.3f} ms  {3 * nbytes / ms_fb / 1e6:8.0f} GB/s algorithmic (12 B/voxel)")
    if a.B <= 64:
        xe = x.clone().requires_grad_(True)
        ms_e = timeit(lambda: eager(x, J, D), max(2, a.iters // 3))

        def efb():
            xe.grad = None
            eager(xe, J, D).backward(g)
        ms_eb = timeit(efb, max(2, a.iters // 3))
        print(f"  PyTorch eager forward  {ms_e:8.3f} ms   eager fwd+bwd {ms_eb:8.3f} ms   "
              f"(speed-up {ms_e / ms_f:.1f}x / {ms_eb / ms_fb:.1f}x)")


if __name__ == "__main__":
    main()
