"""GPU tests of the fused softmax + soft-argmax heads against the numpy oracle
(oracle/heads_oracle.py, restated from the reference text -- parity unpinned, see its header) and
against a plain torch fp32 evaluation of the same formulas (autograd for the backward)."""
import numpy as np
import pytest
import torch

from oracle import heads_oracle as ho

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    return ge.build()


def _torch_ref(out, J, D, centred):
    B, C, H, W = out.shape
    hm = torch.softmax(out.reshape(B, J, -1), 2)
    hm = (hm / hm.sum(2, keepdim=True)).reshape(B, J, D, H, W)
    cx = (hm.sum((2, 3)) * torch.arange(W, dtype=out.dtype, device=out.device)).sum(2, keepdim=True)
    cy = (hm.sum((2, 4)) * torch.arange(H, dtype=out.dtype, device=out.device)).sum(2, keepdim=True)
    if centred:
        cz = (hm.sum((3, 4)) * torch.arange(D, dtype=out.dtype, device=out.device)).sum(2, keepdim=True)
        c = torch.cat(((cx / W - 0.5) * 2, (cy / H - 0.5) * 2, (cz / D - 0.5) * 2), 2)
    else:
        c = torch.cat((cx / W, cy / H), 2)
    return c.reshape(B, -1)


@pytest.mark.parametrize("B,D,H,W,scale", [(2, 64, 64, 64, 1.0), (3, 64, 64, 64, 12.0), (5, 8, 16, 32, 4.0),
                                           (1, 64, 64, 64, 60.0)])
def test_soft_argmax_3d(pkg, B, D, H, W, scale):
    torch.manual_seed(B + D)
    J = 17
    out = (torch.randn(B, J * D, H, W) * scale)
    out[0, 5 * D + 3, 7, 9] += 40.0                      # one sharp peak: online-softmax rescale path
    x = out.to(DEV).requires_grad_(True)
    c = pkg.soft_argmax_3d(x, J, D)
    want = ho.soft_argmax(out.numpy(), J, D, True)
    np.testing.assert_allclose(c.detach().cpu().numpy(), want, rtol=0, atol=2e-5)
    assert c.shape == (B, J * 3) and float(c.detach().abs().max()) <= 1.0
    g = torch.randn(B, J * 3)
    c.backward(g.to(DEV))
    xr = out.double().requires_grad_(True)
    _torch_ref(xr, J, D, True).backward(g.double())
    ref = xr.grad.numpy()
    got = x.grad.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-12


@pytest.mark.parametrize("B,H,W", [(4, 64, 64), (1, 32, 48)])
def test_soft_argmax_2d(pkg, B, H, W):
    torch.manual_seed(B)
    J = 17
    out = torch.randn(B, J, H, W) * 5
    x = out.to(DEV).requires_grad_(True)
    c = pkg.soft_argmax_2d(x, J)
    np.testing.assert_allclose(c.detach().cpu().numpy(), ho.soft_argmax(out.numpy(), J, 1, False), rtol=0, atol=2e-5)
    assert float(c.detach().min()) >= 0.0 and float(c.detach().max()) <= 1.0
    g = torch.randn(B, J * 2)
    c.backward(g.to(DEV))
    xr = out.double().requires_grad_(True)
    _torch_ref(xr, J, 1, False).backward(g.double())
    assert np.abs(x.grad.cpu().numpy() - xr.grad.numpy()).max() <= 1e-5 * xr.grad.abs().max().item() + 1e-12


def test_soft_argmax_properties_and_errors(pkg):
    J, D = 17, 64
    # a delta at voxel (d, h, w) puts the expectation exactly there
    out = torch.full((1, J * D, 64, 64), -1e4)
    out[0, 2 * D + 10, 20, 30] = 0.0
    c = pkg.soft_argmax_3d(out.to(DEV), J, D).cpu().reshape(J, 3)
    assert torch.allclose(c[2], torch.tensor([(30 / 64 - 0.5) * 2, (20 / 64 - 0.5) * 2, (10 / 64 - 0.5) * 2]), atol=1e-6)
    # shift invariance of softmax; uniform logits -> centre of the index range
    u = pkg.soft_argmax_3d(torch.zeros(1, J * D, 64, 64, device=DEV) + 3.0, J, D)
    assert torch.allclose(u, torch.full_like(u, (31.5 / 64 - 0.5) * 2), atol=1e-6)
    with pytest.raises(ValueError):
        pkg.soft_argmax_3d(torch.zeros(1, 17 * 63, 64, 64, device=DEV))
    with pytest.raises(pkg.PoseliftError):
        pkg.soft_argmax_2d(torch.zeros(1, 17, 64, 64))              # CPU tensor: no fallback


def test_soft_argmax_3d_nhwc_matches_the_nchw_kernel_and_oracle(pkg):
    """The NHWC-layout forward (what the conv path's final layer writes) against the NCHW kernel and the oracle."""
    from oracle import heads_oracle
    torch.manual_seed(5)
    x = torch.randn(3, 1088, 16, 24) * 3
    want = heads_oracle.soft_argmax(x.double().numpy(), 17, 64, True)
    got = pkg.soft_argmax_3d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV)).cpu().numpy()
    ref = pkg.soft_argmax_3d(x.to(DEV)).cpu().numpy()
    assert np.abs(got - want).max() < 2e-5 and np.abs(got - ref).max() < 2e-5


def test_soft_argmax_3d_nhwc_backward_vs_torch_autograd(pkg):
    """dlogits of the NHWC head against torch autograd (fp64) of the stated softmax + expectation, and against the
    NCHW kernel's backward on the same logits."""
    torch.manual_seed(6)
    B, J, D, H, W = 2, 17, 64, 12, 8
    x = torch.randn(B, J * D, H, W) * 2
    g = torch.randn(B, J * 3)
    xr = x.double().requires_grad_(True)
    hm = torch.softmax(xr.reshape(B, J, -1), 2).reshape(B, J, D, H, W)
    cx = (hm.sum((2, 3)) * torch.arange(W, dtype=torch.float64)).sum(2, keepdim=True)
    cy = (hm.sum((2, 4)) * torch.arange(H, dtype=torch.float64)).sum(2, keepdim=True)
    cz = (hm.sum((3, 4)) * torch.arange(D, dtype=torch.float64)).sum(2, keepdim=True)
    pred = torch.cat(((cx / W - .5) * 2, (cy / H - .5) * 2, (cz / D - .5) * 2), 2).reshape(B, J * 3)
    (pred * g.double()).sum().backward()
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    out = pkg.soft_argmax_3d_nhwc(xn)
    assert float((out.detach().cpu().double() - pred.detach()).abs().max()) < 2e-5
    (out * g.to(DEV)).sum().backward()
    want = xr.grad.permute(0, 2, 3, 1)
    scale = float(want.abs().max())
    assert float((xn.grad.cpu().double() - want).abs().max()) < 2e-5 * scale
    xc = x.to(DEV).requires_grad_(True)
    (pkg.soft_argmax_3d(xc) * g.to(DEV)).sum().backward()
    assert float((xn.grad.permute(0, 3, 1, 2) - xc.grad).abs().max()) < 2e-6 * scale
