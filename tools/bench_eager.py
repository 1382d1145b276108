#!/usr/bin/env python3
"""PyTorch-ROCm eager of the same module (oracle/torch_twin.py on cuda:0) -- SURVEY 8d config 2's
comparison line.  fp32 and bf16-autocast train steps at batch 4096, HIP-event timed."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.torch_twin import TwinLifter, twin_train_step  # noqa: E402

pkg = importlib.import_module("3d_poseestimation_amd")


def run(autocast, B=4096, steps=50):
    torch.manual_seed(0)
    m = TwinLifter(34, 51).cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    x, y = pkg.synth.synthetic_batch(B, 1234, "cuda")

    def step():
        if autocast:
            opt.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = m(x).reshape(B, -1, 3)
            loss = torch.nn.functional.mse_loss(pred.float(), y)
            loss.backward()
            opt.step()
        else:
            twin_train_step(m, opt, x, y)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return B / dt, dt * 1e3


if __name__ == "__main__":
    for ac in (False, True):
        v, ms = run(ac)
        print(f"torch eager {'bf16 autocast' if ac else 'fp32'}: {v:12.1f} poses/s  {ms:.3f} ms/step")
