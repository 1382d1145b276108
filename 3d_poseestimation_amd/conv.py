"""Convolution-path forward ops on the HIP library (first slice), NHWC throughout.

    conv2d_nhwc          nn.Conv2d forward with the Bottleneck's eval-mode epilogue folded in
                         (/root/reference/phase4_joined/Resnet.py:51-95, :112-118, :151-158; Model.py:66-69)
    maxpool3x3s2_nhwc    nn.MaxPool2d(3, 2, 1)                               (Resnet.py:119)
    deconv4x4s2_nhwc     nn.ConvTranspose2d(4, 2, 1, bias=False) + BN + ReLU (Model.py:47-63)
    nhwc_to_nchw         layout change in front of the soft-argmax
    to_ohwi / deconv_subkernels / fold_bn   weight-layout helpers (host side, once per model)

The phase4 model permutes its NHWC input to NCHW for cuDNN/MIOpen (Model.py:88); here activations stay NHWC
end to end: a feature map IS the [B*H*W][C] matrix the MFMA GEMM wants, a 1x1 convolution is a GEMM, a KxK
convolution is the same GEMM with a gathering A loader (implicit GEMM, no im2col buffer).
"""
import torch

from . import _lib


def to_ohwi(weight_oihw):
    """nn.Conv2d.weight [Cout][Cin][KH][KW] -> contiguous [Cout][KH][KW][Cin]."""
    return weight_oihw.permute(0, 2, 3, 1).contiguous()


def deconv_subkernels(weight_iohw):
    """nn.ConvTranspose2d(k=4, s=2, p=1).weight [Cin][Cout][4][4] -> [4 parities][Cout][2][2][Cin].
    Output row oh = 2*ih - 1 + kh: even rows (oh = 2a) take kh = 3 from input row a-1 and kh = 1 from row a;
    odd rows (oh = 2a+1) take kh = 2 from row a and kh = 0 from row a+1; columns alike."""
    w = weight_iohw
    if w.dim() != 4 or w.shape[2] != 4 or w.shape[3] != 4:
        raise ValueError("deconv_subkernels expects a [Cin][Cout][4][4] ConvTranspose2d weight")
    subs = []
    for ph in (0, 1):
        for pw in (0, 1):
            kh = [2, 0] if ph else [3, 1]
            kw = [2, 0] if pw else [3, 1]
            sub = w[:, :, kh][:, :, :, kw]                     # [Cin][Cout][2][2]
            subs.append(sub.permute(1, 2, 3, 0))               # [Cout][2][2][Cin]
    return torch.stack(subs).contiguous()


def fold_bn(bn):
    """(scale, shift) of an nn.BatchNorm2d in eval mode: y = x * scale + shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


def _opt(t, name, n):
    if t is None:
        return None
    t = t.contiguous()
    _lib.require_device_tensor(t, name)
    if t.numel() != n:
        raise ValueError(f"{name} has {t.numel()} elements, expected {n}")
    return t


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
    opt = [_opt(scale, "scale", Cout), _opt(shift, "shift", Cout), _opt(bias, "bias", Cout),
           _opt(resid, "resid", y.numel())]
    ptr = [t.data_ptr() if t is not None else None for t in opt]
    L = _lib.lib()
    nbytes = L.pl_conv2d_nhwc_scratch_bytes(B, H, W, Cin, Cout, KH, KW, stride, padding)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with torch.cuda.device(x.device):
        rc = L.pl_conv2d_nhwc_fwd(x.data_ptr(), B, H, W, Cin, w.data_ptr(), Cout, KH, KW, stride, padding,
                                  ptr[0], ptr[1], ptr[2], relu, ptr[3], y.data_ptr(),
                                  scratch.data_ptr() if nbytes else None, nbytes, _lib.current_stream_ptr())
    _lib.check(rc, "pl_conv2d_nhwc_fwd")
    return y


def maxpool3x3s2_nhwc(x):
    x = x.contiguous()
    _lib.require_device_tensor(x, "x")
    B, H, W, C = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().pl_maxpool3x3s2_nhwc(x.data_ptr(), B, H, W, C, y.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "pl_maxpool3x3s2_nhwc")
    return y


def deconv4x4s2_nhwc(x, w_sub, scale=None, shift=None, relu=0):
    """x [B,Hi,Wi,Cin] -> [B,2Hi,2Wi,Cout]; w_sub from deconv_subkernels()."""
    x, w = x.contiguous(), w_sub.contiguous()
    _lib.require_device_tensor(x, "x")
    _lib.require_device_tensor(w, "w_sub")
    B, Hi, Wi, Cin = x.shape
    if w.dim() != 5 or w.shape[0] != 4 or w.shape[2:] != (2, 2, Cin):
        raise ValueError(f"deconv4x4s2_nhwc: w_sub {tuple(w.shape)} is not [4][Cout][2][2][{Cin}]")
    Cout = w.shape[1]
    y = torch.empty(B, 2 * Hi, 2 * Wi, Cout, dtype=torch.float32, device=x.device)
    scale, shift = _opt(scale, "scale", Cout), _opt(shift, "shift", Cout)
    L = _lib.lib()
    nbytes = L.pl_deconv4x4s2_nhwc_scratch_bytes(B, Hi, Wi, Cout)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.pl_deconv4x4s2_nhwc_fwd(x.data_ptr(), B, Hi, Wi, Cin, w.data_ptr(), Cout,
                                       scale.data_ptr() if scale is not None else None,
                                       shift.data_ptr() if shift is not None else None, relu, y.data_ptr(),
                                       scratch.data_ptr(), nbytes, _lib.current_stream_ptr())
    _lib.check(rc, "pl_deconv4x4s2_nhwc_fwd")
    return y


def nhwc_to_nchw(x):
    """[B,H,W,C] -> contiguous [B,C,H,W]."""
    x = x.contiguous()
    _lib.require_device_tensor(x, "x")
    B, H, W, C = x.shape
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().pl_nhwc_to_nchw(x.data_ptr(), B, H * W, C, y.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "pl_nhwc_to_nchw")
    return y
