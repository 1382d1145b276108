#!/usr/bin/env python3
"""phase4 Model_3D inference (BASELINE configs[3] shape: 256x256 frames, ResNet-50 + deconv head + soft-argmax),
eval-mode forward: this library (NHWC, bf16x6 = fp32-grade) beside PyTorch-ROCm eager of the SAME stock nn
modules (MIOpen; NCHW fp32, channels_last fp32, channels_last bf16 autocast), same GPU, same weights.
    python tools/bench_model3d.py [--B 64] [--iters 10]"""
import argparse, importlib, os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")


def torch_forward(m, x_nchw):
    r = m.preact
    x = F.max_pool2d(F.relu(r.bn1(r.conv1(x_nchw))), 3, 2, 1)
    for li in (1, 2, 3, 4):
        for blk in getattr(r, f"layer{li}"):
            idn = x if blk.downsample is None else blk.downsample(x)
            o = F.relu(blk.bn1(blk.conv1(x)))
            o = F.relu(blk.bn2(blk.conv2(o)))
            x = F.relu(blk.bn3(blk.conv3(o)) + idn)
    out = m.final_layer(m.deconv_layers(x))
    hm = torch.softmax(out.reshape(out.shape[0], 17, -1).float(), 2).reshape(out.shape[0], 17, 64, 64, 64)
    ar = torch.arange(64, device=out.device, dtype=torch.float32)
    cx = (hm.sum((2, 3)) * ar).sum(2, keepdim=True); cy = (hm.sum((2, 4)) * ar).sum(2, keepdim=True)
    cz = (hm.sum((3, 4)) * ar).sum(2, keepdim=True)
    return torch.cat(((cx / 64 - .5) * 2, (cy / 64 - .5) * 2, (cz / 64 - .5) * 2), 2).reshape(out.shape[0], 51)


def timed(f, iters):
    for _ in range(2):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64); ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    m = pkg.Model_3D().eval()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-4)
    m = m.to("cuda")
    frames = pkg.synth.seeded_frames(a.B, 5).to("cuda")
    with torch.no_grad():
        ours = m(frames)
        ref = torch_forward(m, frames.permute(0, 3, 1, 2).contiguous())
        print(f"B={a.B}: max |coords - torch| = {float((ours - ref).abs().max()):.2e}")
        t = timed(lambda: m(frames), a.iters)
        print(f"this library (f16x3: planes GEMM)      : {t * 1e3:8.2f} ms/batch = {a.B / t:8.1f} frames/s")
        for dt, label in (("bf16p", "bf16p: planes, bf16 storage"), ("bf16x6", "bf16x6: round-1 kernels   "),
                          ("bf16", "bf16: round-1, bf16 arith  ")):
            m.compute_dtype = m.preact.compute_dtype = dt
            m._cache = None; m.preact._cache = None
            fast = m(frames)
            t = timed(lambda: m(frames), a.iters)
            print(f"this library ({label}): {t * 1e3:8.2f} ms/batch = {a.B / t:8.1f} frames/s   "
                  f"(max |coords - f16x3| = {float((fast - ours).abs().max()):.2e})")
        m.compute_dtype = m.preact.compute_dtype = "bf16x6"
        xn = frames.permute(0, 3, 1, 2).contiguous()
        t = timed(lambda: torch_forward(m, xn), a.iters)
        print(f"PyTorch-ROCm eager fp32 NCHW           : {t * 1e3:8.2f} ms/batch = {a.B / t:8.1f} frames/s")
        mc = m.to(memory_format=torch.channels_last); xc = frames.permute(0, 3, 1, 2)     # NHWC memory
        t = timed(lambda: torch_forward(mc, xc), a.iters)
        print(f"PyTorch-ROCm eager fp32 channels_last  : {t * 1e3:8.2f} ms/batch = {a.B / t:8.1f} frames/s")
        with torch.autocast("cuda", dtype=torch.bfloat16):
            t = timed(lambda: torch_forward(mc, xc), a.iters)
        print(f"PyTorch-ROCm eager bf16 autocast       : {t * 1e3:8.2f} ms/batch = {a.B / t:8.1f} frames/s")


if __name__ == "__main__":
    main()
