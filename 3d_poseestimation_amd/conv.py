"""NHWC convolution forward on the HIP library (first slice).

    conv2d_nhwc   nn.Conv2d forward with the Bottleneck's eval-mode epilogue folded in
                  (/root/reference/phase4_joined/Resnet.py:51-95, :121-165; Model.py:66-69)
    to_ohwi       nn.Conv2d.weight [Cout][Cin][KH][KW]  ->  the kernel's [Cout][KH][KW][Cin]
    fold_bn       eval-mode BatchNorm2d -> per-channel (scale, shift)

The phase4 model permutes its NHWC input to NCHW for cuDNN/MIOpen (Model.py:88); here activations
stay NHWC end to end: a feature map IS the [B*H*W][C] matrix the MFMA GEMM wants, a 1x1 convolution is
a GEMM, a KxK convolution is the same GEMM with a gathering A loader (implicit GEMM, no im2col buffer).
"""
import torch

from . import _lib


def to_ohwi(weight_oihw):
    """[Cout][Cin][KH][KW] (torch) -> contiguous [Cout][KH][KW][Cin]."""
    return weight_oihw.permute(0, 2, 3, 1).contiguous()


def fold_bn(bn):
    """(scale, shift) of an nn.BatchNorm2d in eval mode: y = x * scale + shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


def conv2d_nhwc(x, w_ohwi, stride=1, padding=0, scale=None, shift=None, bias=None, relu=0, resid=None):
    """x [B,H,W,Cin] fp32 -> [B,Ho,Wo,Cout] fp32.  relu: 0 none, 1 ReLU then + resid, 2 + resid then ReLU."""
    x, w = x.contiguous(), w_ohwi.contiguous()
    _lib.require_device_tensor(x, "x")
    _lib.require_device_tensor(w, "weight")
    if x.dim() != 4 or w.dim() != 4 or w.shape[3] != x.shape[3]:
        raise ValueError(f"conv2d_nhwc: x {tuple(x.shape)} (NHWC) vs weight {tuple(w.shape)} (OHWI)")
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    opt = []
    for name, t, n in (("scale", scale, Cout), ("shift", shift, Cout), ("bias", bias, Cout),
                       ("resid", resid, y.numel())):
        if t is not None:
            t = t.contiguous()
            _lib.require_device_tensor(t, name)
            if t.numel() != n:
                raise ValueError(f"conv2d_nhwc: {name} has {t.numel()} elements, expected {n}")
        opt.append(t)
    ptr = [t.data_ptr() if t is not None else None for t in opt]
    with torch.cuda.device(x.device):
        rc = _lib.lib().pl_conv2d_nhwc_fwd(x.data_ptr(), B, H, W, Cin, w.data_ptr(), Cout, KH, KW, stride, padding,
                                           ptr[0], ptr[1], ptr[2], relu, ptr[3], y.data_ptr(),
                                           _lib.current_stream_ptr())
    _lib.check(rc, "pl_conv2d_nhwc_fwd")
    return y
