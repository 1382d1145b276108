#!/usr/bin/env python3
"""Profiling target: N eval forwards of Model_3D on the HIP path.  python3 tools/run_model3d.py [B] [iters]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = pkg.Model_3D().eval()
m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
m = m.to("cuda")
x = pkg.synth.seeded_frames(B, 5).to("cuda")
for _ in range(iters):
    m(x)
torch.cuda.synchronize()
print("done")
