"""Fused softmax + integral soft-argmax heads, computed by libposelift.so.

soft_argmax_3d  = the tail of Model_3D.forward   /root/reference/phase4_joined/Model.py:94-133
soft_argmax_2d  = the tail of Model_2D.forward   /root/reference/phase5_loop/Model_2d.py:96-134
Both take the final 1x1-conv output of the reference model and are differentiable; the normalised
heat-map (17.8 MB per frame in 3-D) is never materialised.
"""
import torch

from . import _lib


class _SoftArgmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, BJ, D, H, W, ncoord, centred):
        _lib.require_device_tensor(logits, "heat-map logits")
        coords = torch.empty(BJ, ncoord, dtype=torch.float32, device=logits.device)
        stats = torch.empty(BJ, 5, dtype=torch.float32, device=logits.device)
        with _lib.on_device(logits.device):
            rc = _lib.lib().pl_softargmax_fwd(logits.data_ptr(), BJ, D, H, W, ncoord, centred, coords.data_ptr(),
                                              stats.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_softargmax_fwd")
        ctx.save_for_backward(logits, stats)
        ctx.dims = (BJ, D, H, W, ncoord, centred)
        return coords

    @staticmethod
    def backward(ctx, g):
        logits, stats = ctx.saved_tensors
        BJ, D, H, W, ncoord, centred = ctx.dims
        g = g.contiguous()
        dl = torch.empty_like(logits)
        with _lib.on_device(logits.device):
            rc = _lib.lib().pl_softargmax_bwd(logits.data_ptr(), stats.data_ptr(), g.data_ptr(), BJ, D, H, W, ncoord,
                                              centred, dl.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_softargmax_bwd")
        return dl, None, None, None, None, None, None


def soft_argmax_3d(out, num_joints=17, depth_dim=64):
    """(B, num_joints*depth_dim, H, W) logits -> (B, num_joints*3) coordinates in (-1, 1), (x, y, z) per joint."""
    B, C, H, W = out.shape
    if C != num_joints * depth_dim:
        raise ValueError(f"expected {num_joints * depth_dim} channels, got {C}")
    x = out.contiguous().float()
    return _SoftArgmaxFn.apply(x, B * num_joints, depth_dim, H, W, 3, 1).reshape(B, num_joints * 3)


def soft_argmax_2d(out, num_joints=17):
    """(B, num_joints, H, W) logits -> (B, num_joints*2) coordinates in (0, 1), (x, y) per joint."""
    B, C, H, W = out.shape
    if C != num_joints:
        raise ValueError(f"expected {num_joints} channels, got {C}")
    x = out.contiguous().float()
    return _SoftArgmaxFn.apply(x, B * num_joints, 1, H, W, 2, 0).reshape(B, num_joints * 2)


def _pow2_scale_for_bound(bound):
    """{S, 1/S} on the device: S the power of two that maps `bound` (a 0-d device tensor >= max |value|) into [2^13, 2^14)
    -- the range scale of fp16 operand planes (conv.py PlaneLink); bound 0 / inf / nan -> 1."""
    ok = torch.isfinite(bound) & (bound > 0)
    e = torch.frexp(torch.where(ok, bound, torch.ones_like(bound))).exponent
    S = torch.ldexp(torch.ones_like(bound), 14 - e)
    S = torch.where(ok, S, torch.ones_like(bound))
    return torch.stack([S, 1.0 / S]).float().contiguous()


class _SoftArgmax3dNHWCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, num_joints, link=None):
        ctx.link = link
        B, H, W, _ = x.shape
        coords = torch.empty(B * num_joints, 3, dtype=torch.float32, device=x.device)
        stats = torch.empty(B * num_joints, 5, dtype=torch.float32, device=x.device)
        with _lib.on_device(x.device):
            rc = _lib.lib().pl_softargmax3d_nhwc_fwd(x.data_ptr(), B, num_joints, H, W, coords.data_ptr(),
                                                     stats.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_softargmax3d_nhwc_fwd")
        ctx.save_for_backward(x, stats)
        ctx.num_joints = num_joints
        return coords

    @staticmethod
    def backward(ctx, g):
        x, stats = ctx.saved_tensors
        B, H, W, _ = x.shape
        g = g.contiguous()
        dl = torch.empty_like(x)
        link = ctx.link
        if link is not None:
            # the final convolution runs on the planes GEMM (conv.py): dlogits leave as a carrier of their planes, fp16 ones
            # scaled by a power of two from the bound |dlogit| <= 2 max_(b,j) sum_c |g_c| (softmax weights <= 1)
            link.dz_scale = torch.empty(2, dtype=torch.float32, device=x.device)
            with _lib.on_device(x.device):
                _lib.check(_lib.lib().pl_softargmax_dl_scale(g.data_ptr(), g.numel() // 3, 3, link.dz_scale.data_ptr(),
                                                             _lib.current_stream_ptr()), "pl_softargmax_dl_scale")
                rc = _lib.lib().pl_softargmax3d_nhwc_bwd_ex(x.data_ptr(), stats.data_ptr(), g.data_ptr(), B, ctx.num_joints, H, W,
                                                            None, dl.data_ptr(), link.mode, link.dz_scale.data_ptr(),
                                                            _lib.current_stream_ptr())
            _lib.check(rc, "pl_softargmax3d_nhwc_bwd_ex")
            return dl, None, None
        with _lib.on_device(x.device):
            rc = _lib.lib().pl_softargmax3d_nhwc_bwd(x.data_ptr(), stats.data_ptr(), g.data_ptr(), B, ctx.num_joints,
                                                     H, W, dl.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_softargmax3d_nhwc_bwd")
        return dl, None, None


def soft_argmax_3d_nhwc(out, num_joints=17, link=None):
    """(B, H, W, num_joints*64) NHWC logits (depth_dim 64) -> (B, num_joints*3), differentiable: what Model_3D's
    final 1x1 convolution writes, read in place -- no NHWC <-> NCHW pass in either direction."""
    x = out.contiguous()
    _lib.require_device_tensor(x, "heat-map logits")
    B, H, W, C = x.shape
    if C != num_joints * 64:
        raise ValueError(f"expected {num_joints * 64} channels (depth 64), got {C}")
    return _SoftArgmax3dNHWCFn.apply(x, num_joints, link).reshape(B, num_joints * 3)
