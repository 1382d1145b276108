"""Host-side profile of the Model_3D training step at a small batch (the reference trains phase4 at batch 8, train.py:187):
cProfile over 10 steps, top functions by own time and by cumulative time."""
import cProfile, importlib, os, pstats, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
m = pkg.Model_3D(compute_dtype=dtype).train()
m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
m = m.to("cuda")
x = pkg.synth.seeded_frames(B, 5).to("cuda")
t = torch.randn(B, 51, device="cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def step():
    opt.zero_grad()
    F.mse_loss(m(x), t).backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B} {dtype}: host {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
