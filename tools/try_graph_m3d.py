"""Can a whole Model_3D training step (forward, loss, backward, Adam) be captured as one hipGraph and replayed?"""
import importlib, os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
def build():
    m = pkg.Model_3D(compute_dtype=dtype).train()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
    return m.to("cuda")
x = pkg.synth.seeded_frames(B, 5).to("cuda")
t = torch.randn(B, 51, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
# eager reference: 3 + 4 steps
m0 = build(); o0 = torch.optim.Adam(m0.parameters(), lr=1e-3, capturable=True)
def step(m, o, xx, tt):
    o.zero_grad(set_to_none=True)
    loss = F.mse_loss(m(xx), tt)
    loss.backward()
    o.step()
    return loss
for _ in range(7): l0 = step(m0, o0, x, t)
torch.cuda.synchronize()
# graphed: 3 eager warm-up steps on a side stream, then capture one step, replay 4 times
m1 = build(); o1 = torch.optim.Adam(m1.parameters(), lr=1e-3, capturable=True)
sx, st = x.clone(), t.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step(m1, o1, sx, st)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
o1.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    lg = step(m1, o1, sx, st)
torch.cuda.synchronize()
print("captured")
for _ in range(3): g.replay()          # capture does not execute: steps 4..6 of the graphed model... plus one more below
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("loss eager", float(l0), "graph", float(lg))
d = max(float((a - b).abs().max()) for a, b in zip(m0.parameters(), m1.parameters()))
print("max |param diff| eager vs graph after 7 steps:", d)
N = 30
t0 = time.perf_counter()
for _ in range(N): g.replay()
torch.cuda.synchronize()
print(f"replay {1e3*(time.perf_counter()-t0)/N:.2f} ms/step")
t0 = time.perf_counter()
for _ in range(N): step(m0, o0, x, t)
torch.cuda.synchronize()
print(f"eager-launch (this library) {1e3*(time.perf_counter()-t0)/N:.2f} ms/step")
