#!/usr/bin/env python3
"""Unit check: deconv_planes_eval / conv2d_planes_eval against the round-1 eval kernels on random data."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3d_poseestimation_amd")
cv = importlib.import_module("3d_poseestimation_amd.conv")
dev = "cuda"
torch.manual_seed(0)
B, H, W, Cin, Cout = 2, 8, 8, 256, 128
x = torch.randn(B, H, W, Cin, device=dev)
w = torch.randn(Cin, Cout, 4, 4, device=dev) * 0.05
sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
wsub = cv.deconv_subkernels(w)
ref = cv.deconv4x4s2_nhwc(x, wsub, sc, sh, relu=1, arith="bf16x6")
xp = cv._planes_of(x, cv.ACT_PLANE_SCALE)
wsp = cv._planes_of(wsub, 16.0)
y, yp = cv.deconv_planes_eval(xp, wsp, Cout, sc, sh, relu=1, want_f32=True, want_planes=True)
n = y.numel()
p16 = yp.reshape(-1).view(torch.float16)
back = ((p16[:n].float() + p16[n:].float() / 2048) / cv.ACT_PLANE_SCALE).reshape(y.shape)
print("deconv f32 vs ref", float((y - ref).abs().max()), "planes vs f32", float((back - y).abs().max()), "ref max", float(ref.abs().max()))
_, yp2 = cv.deconv_planes_eval(xp, wsp, Cout, sc, sh, relu=1, want_f32=False, want_planes=True)
p16 = yp2.reshape(-1).view(torch.float16)
back2 = ((p16[:n].float() + p16[n:].float() / 2048) / cv.ACT_PLANE_SCALE).reshape(y.shape)
print("planes-only vs ref", float((back2 - ref).abs().max()))
# conv 1x1 with bias, N = 1088
Cin2, Cout2 = 256, 1088
x2 = torch.randn(B, 16, 16, Cin2, device=dev)
w2 = torch.randn(Cout2, 1, 1, Cin2, device=dev) * 0.05
b2 = torch.randn(Cout2, device=dev)
ref2 = cv.conv2d_nhwc(x2, w2, 1, 0, bias=b2, arith="bf16x6")
y2, _ = cv.conv2d_planes_eval(cv._planes_of(x2, cv.ACT_PLANE_SCALE), cv._planes_of(w2, 16.0), w2.shape, 1, 0, bias=b2)
print("final conv vs ref", float((y2 - ref2).abs().max()), float(ref2.abs().max()))
# 3x3 with scale/shift/relu=2/resid, planes out
w3 = torch.randn(128, 3, 3, Cin2, device=dev) * 0.05
sc3, sh3 = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
res = torch.randn(B, 16, 16, 128, device=dev)
ref3 = cv.conv2d_nhwc(x2, w3, 1, 1, sc3, sh3, relu=2, resid=res, arith="bf16x6")
y3, y3p = cv.conv2d_planes_eval(cv._planes_of(x2, cv.ACT_PLANE_SCALE), cv._planes_of(w3, 16.0), w3.shape, 1, 1, sc3, sh3, relu=2, resid=res, want_planes=True)
n3 = y3.numel(); q = y3p.reshape(-1).view(torch.float16)
print("3x3 fused vs ref", float((y3 - ref3).abs().max()), "planes", float((((q[:n3].float() + q[n3:].float() / 2048) / cv.ACT_PLANE_SCALE).reshape(y3.shape) - y3).abs().max()))
