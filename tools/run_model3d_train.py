#!/usr/bin/env python3
"""Profiling target: N training steps of Model_3D on the HIP path.  python3 tools/run_model3d_train.py [B] [iters] [bf16x6|f16x3|bf16]"""
import importlib, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16x6"
m = pkg.Model_3D(compute_dtype=dtype).train()
m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
with torch.no_grad():
    m.final_layer.weight.mul_(1e-3)
m = m.to("cuda")
x = pkg.synth.seeded_frames(B, 5).to("cuda")
t = torch.randn(B, 51, device="cuda")
opt = pkg.FlatAdam(m, lr=1e-3)
for _ in range(iters):
    opt.zero_grad()
    pkg.mse_loss(m(x), t).backward()
    opt.step()
torch.cuda.synchronize()
print("done")
