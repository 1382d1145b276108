#!/usr/bin/env python3
"""Eval forward of Model_3D: planes path (f16x3) against round 1's kernels (bf16x6), stage by stage."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = pkg.Model_3D().eval()
m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
with torch.no_grad():
    m.final_layer.weight.mul_(1e-3)
m = m.to("cuda")
x = pkg.synth.seeded_frames(B, 5).to("cuda")
outs = {}
for dt in ("bf16x6", "f16x3"):
    m.compute_dtype = m.preact.compute_dtype = dt
    m.preact._cache = None; m._cache = None
    with torch.no_grad():
        feat = m.preact(x)
        logits = m.heatmap_logits_nhwc(x)
        coords = m(x)
    outs[dt] = (feat.double(), logits.double(), coords.double())
    print(dt, "feat max", float(feat.abs().max()), "logits max", float(logits.abs().max()), "finite", bool(torch.isfinite(logits).all()))
for i, name in enumerate(("features", "logits", "coords")):
    a, b = outs["bf16x6"][i], outs["f16x3"][i]
    print(f"{name}: max |diff| {float((a - b).abs().max()):.3e}  rel to max {float((a - b).abs().max() / a.abs().max()):.3e}")
# ---- head stage by stage from the SAME backbone output
conv = importlib.import_module("3d_poseestimation_amd.conv")
m.compute_dtype = m.preact.compute_dtype = "f16x3"
m.preact._cache = None; m._cache = None
f = m._folded()
with torch.no_grad():
    x0, x0p = m.preact._forward_eval_planes(x)
    out, outp = x0, x0p
    for i in (0, 3, 6):
        ref = conv.deconv4x4s2_nhwc(out, f[i], f[i + 1][0], f[i + 1][1], relu=1, arith="bf16x6")
        y, yp = conv.deconv_planes_eval(outp, f[f"{i}@p"], f[i].shape[1], f[i + 1][0], f[i + 1][1], relu=1, want_f32=True, want_planes=True)
        n = y.numel(); q = yp.reshape(-1).view(torch.float16)
        back = ((q[:n].float() + q[n:].float() / 2048) / conv.ACT_PLANE_SCALE).reshape(y.shape)
        print(f"deconv {i}: ref max {float(ref.abs().max()):.4g}  f32 vs ref {float((y - ref).abs().max()):.3e}  planes vs f32 {float((back - y).abs().max()):.3e} finite {bool(torch.isfinite(back).all())}")
        out, outp = ref, yp
