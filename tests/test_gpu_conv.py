"""GPU tests of the NHWC convolution forward (first slice) against stock PyTorch on the
CPU (F.conv2d, fp32 -- the ATen arithmetic the reference's nn.Conv2d / BatchNorm2d dispatch to) at the
phase4 backbone's layer shapes (phase4_joined/Resnet.py: Bottleneck conv2 3x3 with the stage's stride,
1x1 downsample with stride 2, 1x1 conv3 with BN + residual + ReLU)."""
import importlib
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    return ge.build()


def _ref(x_nhwc, w_oihw, stride, pad, scale, shift, bias, relu, resid):
    v = F.conv2d(x_nhwc.permute(0, 3, 1, 2).double(), w_oihw.double(), None if bias is None else bias.double(),
                 stride=stride, padding=pad)
    if scale is not None:
        v = v * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if relu == 1:
        v = v.clamp_min(0)
    if resid is not None:
        v = v + resid.permute(0, 3, 1, 2).double()
    if relu == 2:
        v = v.clamp_min(0)
    return v.permute(0, 2, 3, 1)


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,pad,bn,relu,res,bias", [
    (2, 32, 128, 128, 3, 1, 1, True, 1, False, False),     # layer2 conv2 (3x3, stride 1)
    (2, 32, 128, 128, 3, 2, 1, True, 1, False, False),     # first block of a stage: stride on the 3x3
    (2, 16, 256, 256, 3, 1, 1, True, 2, True, False),      # + residual, ReLU after the add
    (2, 32, 256, 512, 1, 2, 0, True, 0, False, False),     # downsample 1x1 stride 2 + BN
    (2, 16, 512, 128, 1, 1, 0, True, 1, False, False),     # conv1 1x1 = plain GEMM
    (1, 8, 256, 1088, 1, 1, 0, False, 0, False, True),     # final 1x1 with bias, Cout not a tile multiple
    (4, 8, 64, 128, 3, 1, 1, False, 0, False, False),      # borders dominate: 8x8 maps
])
def test_conv2d_nhwc_vs_torch(pkg, B, H, Cin, Cout, k, stride, pad, bn, relu, res, bias):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin + Cout + k + stride)
    x = torch.randn(B, H, H, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    Ho = (H + 2 * pad - k) // stride + 1
    scale = torch.rand(Cout, generator=g) + 0.5 if bn else None
    shift = torch.randn(Cout, generator=g) if bn else None
    bs = torch.randn(Cout, generator=g) if bias else None
    resid = torch.randn(B, Ho, Ho, Cout, generator=g) if res else None
    dv = lambda t: None if t is None else t.to(DEV)
    y = pkg.conv.conv2d_nhwc(dv(x), pkg.conv.to_ohwi(w).to(DEV), stride, pad, dv(scale), dv(shift), dv(bs), relu, dv(resid))
    want = _ref(x, w, stride, pad, scale, shift, bs, relu, resid)
    assert y.shape == want.shape
    # fp32-grade: the error bound of an fp32 dot product of this length, on the pre-epilogue magnitude
    mag = F.conv2d(x.permute(0, 3, 1, 2).abs().double(), w.abs().double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    bound = 4e-6 * mag * (scale.abs().double().view(1, 1, 1, -1) if bn else 1.0) + 1e-6
    err = (y.cpu().double() - want).abs()
    assert bool((err <= bound).all()), float((err / bound).max())


def test_conv2d_nhwc_rejects_what_it_does_not_cover(pkg):
    x = torch.randn(2, 16, 16, 3, device=DEV)
    with pytest.raises(pkg.PoseliftError, match="Cin"):
        pkg.conv.conv2d_nhwc(x, torch.randn(128, 7, 7, 3, device=DEV), 2, 3)          # the 7x7 stem
    with pytest.raises(pkg.PoseliftError):
        pkg.conv.conv2d_nhwc(torch.randn(2, 16, 16, 64), torch.randn(128, 3, 3, 64))  # CPU tensors
