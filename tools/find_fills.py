"""Which Python lines issue aten::fill_ / zero_ / copy_ in one Model_3D training step (torch.profiler, with stacks)."""
import collections, importlib, os, sys, torch
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
m = pkg.Model_3D().train(); m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31)); m = m.to("cuda")
x = pkg.synth.seeded_frames(8, 5).to("cuda"); t = torch.randn(8, 51, device="cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def step():
    opt.zero_grad(); F.mse_loss(m(x), t).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
for op in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::contiguous", "aten::clone"):
    c = collections.Counter()
    for e in prof.events():
        if e.name == op:
            st = [s for s in (e.stack or []) if "3d_poseestimation_amd" in s or "tools/" in s or "optim" in s or "autograd" in s]
            c[st[0] if st else (e.stack[0] if e.stack else "?")] += 1
    print("==", op, sum(c.values()))
    for k, v in c.most_common(12): print(f"   {v:4d}  {k[:150]}")
