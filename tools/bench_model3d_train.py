#!/usr/bin/env python3
"""phase4 train step (BASELINE configs[3] shape: 256x256 frames; phase4_joined/train.py:69-89: forward, MSE, backward,
Adam lr 1e-3) -- this library's conv path (fp32-grade and bf16 arithmetic) beside PyTorch-ROCm eager of the same stock
modules (fp32 and bf16 autocast).
    python tools/bench_model3d_train.py [--B 32] [--iters 5]"""
import argparse, copy, importlib, os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
from tools.bench_model3d import torch_forward  # noqa: E402


def timed(step, iters):
    for _ in range(2):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32); ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--skip-eager", action="store_true")
    ap.add_argument("--graph", action="store_true", help="also time train.GraphedModuleStep (the step as one hipGraph)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam for this library's model (round 2) instead of "
                                                              "arena.FlatAdam (one launch over the flat arenas; same-box A/B)")
    a = ap.parse_args()
    m = pkg.Model_3D().train()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-3)
    ours = copy.deepcopy(m).to("cuda")
    eager = copy.deepcopy(m).to("cuda")
    frames = pkg.synth.seeded_frames(a.B, 5).to("cuda")
    target = torch.randn(a.B, 51, device="cuda")
    opt_o = torch.optim.Adam(ours.parameters(), lr=1e-3) if a.torch_adam else pkg.FlatAdam(ours, lr=1e-3)
    print("optimizer of this library's model:", type(opt_o).__name__)
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-3)

    def step_ours():
        opt_o.zero_grad()
        (F.mse_loss if a.torch_adam else pkg.mse_loss)(ours(frames), target).backward()
        opt_o.step()

    xn = frames.permute(0, 3, 1, 2).contiguous()

    def step_eager():
        opt_e.zero_grad()
        F.mse_loss(torch_forward(eager, xn), target).backward()
        opt_e.step()

    def step_eager_bf16():
        opt_e.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = torch_forward(eager, xn)
        F.mse_loss(pred.float(), target).backward()
        opt_e.step()

    ours.compute_dtype = ours.preact.compute_dtype = "bf16x6"       # round 1's kernels: bf16x6 split inside every GEMM
    t = timed(step_ours, a.iters)
    print(f"B={a.B} this library (fp32-grade arithmetic)        : {t * 1e3:8.1f} ms/step = {a.B / t:7.1f} frames/s")
    ours.compute_dtype = ours.preact.compute_dtype = "f16x3"
    t = timed(step_ours, a.iters)
    print(f"B={a.B} this library (fp32-grade, planes GEMM convs)  : {t * 1e3:8.1f} ms/step = {a.B / t:7.1f} frames/s")
    ours.compute_dtype = ours.preact.compute_dtype = "bf16p"
    t = timed(step_ours, a.iters)
    print(f"B={a.B} this library (bf16 operand storage, planes)   : {t * 1e3:8.1f} ms/step = {a.B / t:7.1f} frames/s")
    if a.graph:                                                       # the same steps replayed from one hipGraph each
        for dt, label in (("f16x3", "fp32-grade, planes, hipGraph replay "), ("bf16p", "bf16 storage, planes, hipGraph replay")):
            gm = copy.deepcopy(m).to("cuda")
            gm.compute_dtype = gm.preact.compute_dtype = dt
            gopt = (torch.optim.Adam(gm.parameters(), lr=1e-3, capturable=True) if a.torch_adam
                    else pkg.FlatAdam(gm, lr=1e-3, capturable=True))
            gstep = pkg.GraphedModuleStep(gm, gopt, F.mse_loss if a.torch_adam else pkg.mse_loss, frames, target)
            t = timed(lambda: gstep(frames, target), a.iters)
            print(f"B={a.B} this library ({label}): {t * 1e3:8.1f} ms/step = {a.B / t:7.1f} frames/s")
            del gstep, gm
    if a.skip_eager:
        return
    ours.compute_dtype = ours.preact.compute_dtype = "bf16"
    t = timed(step_ours, a.iters)
    print(f"B={a.B} this library (bf16 arithmetic, fp32 storage) : {t * 1e3:8.1f} ms/step = {a.B / t:7.1f} frames/s")
    t = timed(step_eager, a.iters)
    print(f"B={a.B} PyTorch-ROCm eager fp32                      : {t * 1e3:8.1f} ms/step = {a.B / t:7.1f} frames/s")
    t = timed(step_eager_bf16, a.iters)
    print(f"B={a.B} PyTorch-ROCm eager bf16 autocast             : {t * 1e3:8.1f} ms/step = {a.B / t:7.1f} frames/s")


if __name__ == "__main__":
    main()
