"""Convolution-path ops on the HIP library, NHWC throughout: forward ops and their autograd forms.

    conv2d_nhwc          nn.Conv2d forward with the Bottleneck's eval-mode epilogue folded in
                         (/root/reference/phase4_joined/Resnet.py:51-95, :112-118, :151-158; Model.py:66-69)
    maxpool3x3s2_nhwc    nn.MaxPool2d(3, 2, 1)                               (Resnet.py:119)
    deconv4x4s2_nhwc     nn.ConvTranspose2d(4, 2, 1, bias=False) + BN + ReLU (Model.py:47-63)
    nhwc_to_nchw         layout change for heads that want NCHW (Model_2D)
    *_autograd, batchnorm_relu_train, add_relu   the differentiable forms used in training mode
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
    # tap a of parity ph is kh = 3 - (2a + ph): on the flipped filter the tap index IS 2a + ph, so a reshape of each 4 into
    # (a, ph) and one permuted copy build all four sub-kernels (no index tensors: nothing here touches the host, the whole
    # training step can be captured in a hipGraph)
    cin, cout = w.shape[0], w.shape[1]
    wf = w.flip(2, 3).reshape(cin, cout, 2, 2, 2, 2)           # [Cin][Cout][a][ph][b][pw]
    return wf.permute(3, 5, 1, 2, 4, 0).reshape(4, cout, 2, 2, cin).contiguous()


def fold_bn(bn):
    """(scale, shift) of an nn.BatchNorm2d in eval mode: y = x * scale + shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


def _pgrad(param, make):
    """The gradient of `param` from an autograd node's backward.  make(out) computes it: into `out` (a contiguous tensor of
    the parameter's shape) when given, else into a fresh tensor it returns.
    A parameter managed by arena.ModuleArena (`_pl_grad`: its view of the flat gradient arena) gets its gradient written
    straight into the arena -- the first backward of a step overwrites and attaches the view as .grad, a later one in the
    same step (a second forward pass: the phase5 Flip branch) accumulates -- and autograd is handed None; any other
    parameter's gradient goes to autograd as usual."""
    view = getattr(param, "_pl_grad", None)
    if view is None:
        return make(None)
    if param.grad is None:
        make(view)
        param.grad = view
    else:
        param.grad.add_(make(None).reshape(param.shape))
    return None


def _pgrad2(pa, pb, C, dev, run):
    """_pgrad for two parameters whose gradients ONE kernel writes (BatchNorm's dgamma, dbeta): run(out_a, out_b)."""
    va, vb = getattr(pa, "_pl_grad", None), getattr(pb, "_pl_grad", None)
    if va is not None and vb is not None and pa.grad is None and pb.grad is None:
        run(va, vb)
        pa.grad, pb.grad = va, vb
        return None, None
    a, b = torch.empty(C, device=dev), torch.empty(C, device=dev)
    run(a, b)
    if va is None or vb is None:
        return a, b
    for p, v, t in ((pa, va, a), (pb, vb, b)):
        if p.grad is None:
            v.copy_(t)
            p.grad = v
        else:
            p.grad.add_(t)
    return None, None


def _oihw_from_ohwi(dw_ohwi, out):
    """[O][KH][KW][I] -> the parameter's [O][I][KH][KW] layout (pl_nhwc_to_nchw: rows O, pixels KH*KW, channels I)."""
    O, KH, KW, I = dw_ohwi.shape
    if out is None:
        out = torch.empty(O, I, KH, KW, device=dw_ohwi.device)
    with _lib.on_device(dw_ohwi.device):
        rc = _lib.lib().pl_nhwc_to_nchw(dw_ohwi.data_ptr(), O, KH * KW, I, out.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "pl_nhwc_to_nchw")
    return out


def _opt(t, name, n):
    if t is None:
        return None
    t = t.contiguous()
    _lib.require_device_tensor(t, name)
    if t.numel() != n:
        raise ValueError(f"{name} has {t.numel()} elements, expected {n}")
    return t


# "f16x3": the 1x1 convolutions of a TRAINING step run on the planes GEMM (bottom of this file); everything else -- the
# 3x3 / 7x7 / transposed convolutions, eval mode -- keeps the fp32-grade bf16x6 kernels
ARITH = {"bf16x6": _lib.PL_BF16X6, "bf16": _lib.PL_BF16, "f16x3": _lib.PL_BF16X6, "bf16p": _lib.PL_BF16}


def conv2d_nhwc(x, w_ohwi, stride=1, padding=0, scale=None, shift=None, bias=None, relu=0, resid=None,
                arith="bf16x6"):
    """x [B,H,W,Cin] fp32 -> [B,Ho,Wo,Cout] fp32.  relu: 0 none, 1 ReLU then + resid, 2 + resid then ReLU.
    arith: "bf16x6" (fp32-grade) or "bf16" (operands rounded to bf16: the throughput mode)."""
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
    nbytes = L.pl_conv2d_nhwc_scratch_bytes_ex(B, H, W, Cin, Cout, KH, KW, stride, padding, int(bias is not None),
                                               int(resid is not None), int(relu))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with _lib.on_device(x.device):
        rc = L.pl_conv2d_nhwc_fwd(x.data_ptr(), B, H, W, Cin, w.data_ptr(), Cout, KH, KW, stride, padding,
                                  ptr[0], ptr[1], ptr[2], relu, ptr[3], y.data_ptr(), ARITH[arith],
                                  scratch.data_ptr() if nbytes else None, nbytes, _lib.current_stream_ptr())
    _lib.check(rc, "pl_conv2d_nhwc_fwd")
    return y


def conv2d_nhwc_wgrad(x, dy, kernel_size, stride=1, padding=0, arith="bf16x6"):
    """Weight gradient of conv2d_nhwc: x [B,H,W,Cin], dy [B,Ho,Wo,Cout] -> dw [Cout,KH,KW,Cin] (OHWI)."""
    x, dy = x.contiguous(), dy.contiguous()
    _lib.require_device_tensor(x, "x")
    _lib.require_device_tensor(dy, "dy")
    B, H, W, Cin = x.shape
    KH = KW = int(kernel_size)
    Cout = dy.shape[3]
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    if tuple(dy.shape) != (B, Ho, Wo, Cout):
        raise ValueError(f"conv2d_nhwc_wgrad: dy {tuple(dy.shape)} does not match the forward output {(B, Ho, Wo, Cout)}")
    dw = torch.empty(Cout, KH, KW, Cin, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    nbytes = L.pl_conv2d_nhwc_wgrad_scratch_bytes(B, H, W, Cin, Cout, KH, KW, stride, padding)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with _lib.on_device(x.device):
        rc = L.pl_conv2d_nhwc_wgrad(x.data_ptr(), B, H, W, Cin, dy.data_ptr(), Cout, KH, KW, stride, padding,
                                    dw.data_ptr(), ARITH[arith], scratch.data_ptr() if nbytes else None, nbytes,
                                    _lib.current_stream_ptr())
    _lib.check(rc, "pl_conv2d_nhwc_wgrad")
    return dw


def conv2d_nhwc_dgrad(dy, w_ohwi, in_hw, stride=1, padding=0, arith="bf16x6"):
    """Input gradient of conv2d_nhwc: dy [B,Ho,Wo,Cout], w [Cout,KH,KW,Cin] -> dx [B,H,W,Cin].  No kernel of its
    own: stride 1 is the forward convolution of dy with the flipped filter, channels swapped; stride 2 (3x3 pad 1
    and 1x1 pad 0, the backbone's two cases, even H and W) is the transposed convolution
    dx[2*oh - pad + kh] += dy[oh] * w[kh], i.e. deconv4x4s2_nhwc with the filter placed inside a 4x4 one."""
    Cout, KH, KW, Cin = w_ohwi.shape
    H, W = in_hw
    if stride == 1:
        wf = w_ohwi.flip(1, 2).permute(3, 1, 2, 0).contiguous()            # [Cin][KH][KW][Cout]
        dx = conv2d_nhwc(dy, wf, 1, KH - 1 - padding, arith=arith)
        if dx.shape[1:3] != (H, W):
            raise ValueError("conv2d_nhwc_dgrad: input size does not match")
        return dx
    if stride == 2 and H % 2 == 0 and W % 2 == 0 and KH == 1 and KW == 1 and padding == 0:
        # dy W on the even pixels: one GEMM ([M][Cout] x [Cout][Cin], the OHWI filter read as [Cin][1][1][Cout]^T)
        B, Ho, Wo, _ = dy.shape
        t = conv2d_nhwc(dy, w_ohwi.reshape(Cout, Cin).t().contiguous().reshape(Cin, 1, 1, Cout), 1, 0, arith=arith)
        dx = torch.empty(B, H, W, Cin, dtype=torch.float32, device=dy.device)
        with _lib.on_device(dy.device):
            rc = _lib.lib().pl_upsample2x_zero_nhwc(t.data_ptr(), B, Ho, Wo, Cin, dx.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_upsample2x_zero_nhwc")
        return dx
    if stride == 2 and H % 2 == 0 and W % 2 == 0 and (KH, padding) in ((3, 1), (1, 0)) and KH == KW:
        # ConvTranspose2d weight [in = Cout][out = Cin][4][4]; with padding 1: kh -> kh, with padding 0: kh -> kh + 1
        w4 = torch.zeros(Cout, Cin, 4, 4, dtype=torch.float32, device=w_ohwi.device)
        o = 1 - padding
        w4[:, :, o:o + KH, o:o + KW] = w_ohwi.permute(0, 3, 1, 2)
        return deconv4x4s2_nhwc(dy, deconv_subkernels(w4), arith=arith)
    raise NotImplementedError(f"conv2d_nhwc_dgrad: stride {stride}, kernel {KH}, padding {padding}, input {H}x{W}")


def maxpool3x3s2_nhwc(x):
    x = x.contiguous()
    _lib.require_device_tensor(x, "x")
    B, H, W, C = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C, dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        rc = _lib.lib().pl_maxpool3x3s2_nhwc(x.data_ptr(), B, H, W, C, y.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "pl_maxpool3x3s2_nhwc")
    return y


def deconv4x4s2_nhwc(x, w_sub, scale=None, shift=None, relu=0, arith="bf16x6"):
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
    with _lib.on_device(x.device):
        rc = L.pl_deconv4x4s2_nhwc_fwd(x.data_ptr(), B, Hi, Wi, Cin, w.data_ptr(), Cout,
                                       scale.data_ptr() if scale is not None else None,
                                       shift.data_ptr() if shift is not None else None, relu, y.data_ptr(), ARITH[arith],
                                       scratch.data_ptr(), nbytes, _lib.current_stream_ptr())
    _lib.check(rc, "pl_deconv4x4s2_nhwc_fwd")
    return y


def nhwc_to_nchw(x):
    """[B,H,W,C] -> contiguous [B,C,H,W]."""
    x = x.contiguous()
    _lib.require_device_tensor(x, "x")
    B, H, W, C = x.shape
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        rc = _lib.lib().pl_nhwc_to_nchw(x.data_ptr(), B, H * W, C, y.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "pl_nhwc_to_nchw")
    return y


# ---------------------------------------------------------------------------------------------
# Training-mode building blocks (autograd), NHWC.  The assembly of a trainable ResNet from them -- and the
# transposed-convolution / max-pool backward it also needs -- is the next slice of SURVEY 8f row N2.
# ---------------------------------------------------------------------------------------------
class _Conv2dFn(torch.autograd.Function):
    """conv2d_nhwc with autograd: dgrad and wgrad run on the library too (conv2d_nhwc_dgrad / _wgrad)."""

    @staticmethod
    def forward(ctx, x, w_ohwi, stride, padding, arith):
        ctx.save_for_backward(x, w_ohwi)
        ctx.geom = (stride, padding, arith)
        return conv2d_nhwc(x, w_ohwi, stride, padding, arith=arith)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, arith = ctx.geom
        dy = dy.contiguous()
        dx = conv2d_nhwc_dgrad(dy, w, x.shape[1:3], stride, padding, arith) if ctx.needs_input_grad[0] else None
        dw = conv2d_nhwc_wgrad(x, dy, w.shape[1], stride, padding, arith) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


def conv2d_nhwc_autograd(x, w_ohwi, stride=1, padding=0, arith="bf16x6"):
    """Differentiable conv2d_nhwc (no folded epilogue: in training mode BatchNorm needs batch statistics).
    arith "bf16": forward, dgrad and wgrad round their operands to bf16 while staging (fp32 accumulate/storage)."""
    return _Conv2dFn.apply(x, w_ohwi, stride, padding, arith)


class _BNReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, batches, eps, momentum, relu):
        shape = z.shape
        C = shape[-1]
        z2 = z.contiguous().reshape(-1, C)
        rows = z2.shape[0]
        for name, t in (("z", z2), ("gamma", gamma), ("beta", beta), ("running_mean", running_mean),
                        ("running_var", running_var)):
            _lib.require_device_tensor(t, name)
        _lib.require_device_tensor(batches, "num_batches_tracked", torch.int64)
        dev, L = z2.device, _lib.lib()
        y = torch.empty_like(z2)
        bits = torch.empty(rows, 4 * ((C + 255) // 256), dtype=torch.int64, device=dev)
        mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        scratch = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = L.pl_bn_train_fwd(z2.data_ptr(), rows, C, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                                   running_mean.data_ptr(), running_var.data_ptr(), batches.data_ptr(), int(relu),
                                   y.data_ptr(), bits.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scratch.data_ptr(),
                                   _lib.current_stream_ptr())
        _lib.check(rc, "pl_bn_train_fwd")
        ctx.save_for_backward(z2, bits, mean, rstd, gamma)
        ctx.shape, ctx.gparam, ctx.bparam = shape, gamma, beta
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        z2, bits, mean, rstd, gamma = ctx.saved_tensors
        rows, C = z2.shape
        dy2 = dy.contiguous().reshape(rows, C)
        dev, L = z2.device, _lib.lib()
        dz = torch.empty_like(z2)
        scratch = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=dev)

        def run(dgamma, dbeta):
            with _lib.on_device(dev):
                rc = L.pl_bn_train_bwd(dy2.data_ptr(), bits.data_ptr(), z2.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       gamma.data_ptr(), rows, C, dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                       scratch.data_ptr(), _lib.current_stream_ptr())
            _lib.check(rc, "pl_bn_train_bwd")
        dgamma, dbeta = _pgrad2(ctx.gparam, ctx.bparam, C, dev, run)
        return dz.reshape(ctx.shape), dgamma, dbeta, None, None, None, None, None, None


def batchnorm_relu_train(z, bn, relu=True):
    """Training-mode nn.BatchNorm2d `bn` (+ ReLU) applied to an NHWC feature map z [..., C] (batch statistics over
    every leading dimension); updates bn.running_mean / running_var / num_batches_tracked in place."""
    if bn.momentum is None:
        raise NotImplementedError("cumulative moving average (momentum=None)")
    return _BNReLUFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                           float(bn.eps), float(bn.momentum), bool(relu))


class _AddReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        if a.shape != b.shape:
            raise ValueError("add_relu: shapes differ")
        C = a.shape[-1]
        a2, b2 = a.contiguous().reshape(-1, C), b.contiguous().reshape(-1, C)
        _lib.require_device_tensor(a2, "a")
        _lib.require_device_tensor(b2, "b")
        rows = a2.shape[0]
        out = torch.empty_like(a2)
        bits = torch.empty(rows, 4 * ((C + 255) // 256), dtype=torch.int64, device=a2.device)
        with _lib.on_device(a2.device):
            rc = _lib.lib().pl_add_relu_fwd(a2.data_ptr(), b2.data_ptr(), rows, C, out.data_ptr(), bits.data_ptr(),
                                            _lib.current_stream_ptr())
        _lib.check(rc, "pl_add_relu_fwd")
        ctx.save_for_backward(bits)
        ctx.shape = a.shape
        return out.reshape(a.shape)

    @staticmethod
    def backward(ctx, g):
        (bits,) = ctx.saved_tensors
        C = ctx.shape[-1]
        g2 = g.contiguous().reshape(-1, C)
        dx = torch.empty_like(g2)
        with _lib.on_device(g2.device):
            rc = _lib.lib().pl_mask_by_bits(g2.data_ptr(), bits.data_ptr(), g2.shape[0], C, dx.data_ptr(),
                                            _lib.current_stream_ptr())
        _lib.check(rc, "pl_mask_by_bits")
        dx = dx.reshape(ctx.shape)
        return dx, dx


def add_relu(a, b):
    """relu(a + b), differentiable: the residual join of a Bottleneck (Resnet.py:90-91)."""
    return _AddReLUFn.apply(a, b)


class _MaxPoolFn(torch.autograd.Function):
    """Forward records the winning tap per output element (one byte); backward reads those and dy only."""

    @staticmethod
    def forward(ctx, x):
        _lib.require_device_tensor(x, "x")
        B, H, W, C = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
        idx = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=x.device)
        with _lib.on_device(x.device):
            rc = _lib.lib().pl_maxpool3x3s2_nhwc_idx(x.data_ptr(), B, H, W, C, y.data_ptr(), idx.data_ptr(),
                                                     _lib.current_stream_ptr())
        _lib.check(rc, "pl_maxpool3x3s2_nhwc_idx")
        ctx.save_for_backward(idx)
        ctx.in_shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = dy.contiguous()
        B, H, W, C = ctx.in_shape
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        with _lib.on_device(dy.device):
            rc = _lib.lib().pl_maxpool3x3s2_nhwc_bwd_idx(idx.data_ptr(), dy.data_ptr(), B, H, W, C, dx.data_ptr(),
                                                         _lib.current_stream_ptr())
        _lib.check(rc, "pl_maxpool3x3s2_nhwc_bwd_idx")
        return dx


def maxpool3x3s2_nhwc_autograd(x):
    return _MaxPoolFn.apply(x.contiguous())


class _DeconvFn(torch.autograd.Function):
    """nn.ConvTranspose2d(4, 2, 1, bias=False) on NHWC, weight in torch's [Cin][Cout][4][4] layout.  Its backward
    needs no kernel of its own: y = C^T x for the stride-2 convolution C with the same filter, so dx = C dy (the
    forward convolution kernel, stride 2) and dW = wgrad of C with input dy and output gradient x."""

    @staticmethod
    def forward(ctx, x, weight, arith):
        ctx.save_for_backward(x, weight)
        ctx.arith = arith
        return deconv4x4s2_nhwc(x, deconv_subkernels(weight.detach().float()), arith=arith)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:       # [Cin][4][4][Cout] as OHWI
            dx = conv2d_nhwc(dy, weight.detach().permute(0, 2, 3, 1).contiguous(), 2, 1, arith=ctx.arith)
        if ctx.needs_input_grad[1]:       # [Cin][4][4][Cout] -> [Cin][Cout][4][4]
            dw = conv2d_nhwc_wgrad(dy, x, 4, 2, 1, ctx.arith).permute(0, 3, 1, 2).contiguous()
        return dx, dw, None


def deconv4x4s2_nhwc_autograd(x, weight_iohw, arith="bf16x6"):
    return _DeconvFn.apply(x.contiguous(), weight_iohw, arith)


class _ConvBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ohwi, bias, stride, padding, arith):
        ctx.save_for_backward(x, w_ohwi)
        ctx.geom = (stride, padding, arith)
        return conv2d_nhwc(x, w_ohwi, stride, padding, bias=bias, arith=arith)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, arith = ctx.geom
        dy = dy.contiguous()
        dx = conv2d_nhwc_dgrad(dy, w, x.shape[1:3], stride, padding, arith) if ctx.needs_input_grad[0] else None
        dw = conv2d_nhwc_wgrad(x, dy, w.shape[1], stride, padding, arith) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.needs_input_grad[2]:
            C = dy.shape[-1]
            rows = dy.numel() // C
            db = torch.empty(C, dtype=torch.float32, device=dy.device)
            L = _lib.lib()
            scratch = torch.empty(L.pl_colsum_scratch_bytes(rows, C), dtype=torch.uint8, device=dy.device)
            with _lib.on_device(dy.device):
                rc = L.pl_colsum(dy.data_ptr(), rows, C, db.data_ptr(), scratch.data_ptr(), _lib.current_stream_ptr())
            _lib.check(rc, "pl_colsum")
        return dx, dw, db, None, None, None


def conv2d_bias_nhwc_autograd(x, w_ohwi, bias, stride=1, padding=0, arith="bf16x6"):
    return _ConvBiasFn.apply(x, w_ohwi, bias, stride, padding, arith)


class _ToNCHWFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return nhwc_to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, C, H, W = g.shape
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=g.device)
        with _lib.on_device(g.device):   # [B][C][P] -> [B][P][C]: the same tiled transpose with the roles swapped
            rc = _lib.lib().pl_nhwc_to_nchw(g.data_ptr(), B, C, H * W, out.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_nhwc_to_nchw")
        return out


def nhwc_to_nchw_autograd(x):
    return _ToNCHWFn.apply(x)


# ---------------------------------------------------------------------------------------------
# 1x1 convolutions on the planes GEMM (compute_dtype "f16x3"): the lifter's round-2 design on the conv path.
#
# A 1x1 convolution over an NHWC map IS the GEMM [pixels][Cin] x [Cout][Cin]^T.  Its operands travel as 16-bit operand
# planes written by the kernel that produces the tensor (BatchNorm apply, the residual join, BatchNorm backward), and the
# GEMM stages them by LDS-DMA (csrc/gemm_planes16.h): three fp16 MFMAs per product instead of six bf16 ones and no split
# work in the loop.  Between autograd nodes a tensor that exists ONLY as planes travels in a float32-typed "carrier" of
# the logical shape: two fp16 planes are exactly four bytes per element, so the carrier has the size autograd expects of
# the tensor (or of its gradient) while its bytes are [2][n] fp16.  Carriers are only ever handed to the functions below
# (ResNet._forward_train wires producer to consumer directly); every gradient that reaches torch code is real fp32.
#   reference: phase4_joined/Resnet.py:56-63, 65-93 (the Bottleneck's conv1 / conv3 / downsample and their autograd)
# ---------------------------------------------------------------------------------------------
ACT_PLANE_SCALE = 1.0 / 64   # csrc/pl_internal.h kConvActPlaneScale (pl_conv_act_plane_scale(): checked by the host-logic test)
WEIGHT_PLANE_SCALE = 16.0    # kWeightPlaneScale


class PlaneLink:
    """Connects the BatchNorm behind a planes convolution to that convolution's backward: the BatchNorm backward writes
    dz as planes (fp16, scaled by a power of two chosen on the device) and leaves {S, 1/S} here for the two GEMMs."""
    __slots__ = ("dz_scale", "mode", "stat")

    def __init__(self, mode=_lib.PL_F16X3):
        self.dz_scale = None
        self.stat = None          # forward: the convolution's epilogue statistics [2][groups][C] for the BatchNorm behind it
        self.mode = mode          # PL_F16X3: two fp16 planes (fp32-grade); PL_BF16: one bf16 plane (bf16 STORAGE of the operands)


def _planes_of_strided(base, dims, strides, offset, scale, mode):
    """Planes of the strided view (dims, signed element strides, element offset) of `base` -- pl_planes_split_strided: the layout
    change and the split in one launch."""
    import ctypes
    nd = len(dims)
    d4 = (ctypes.c_int64 * 4)(*((1,) * (4 - nd) + tuple(int(v) for v in dims)))
    s4 = (ctypes.c_int64 * 4)(*((0,) * (4 - nd) + tuple(int(v) for v in strides)))
    out = torch.empty(tuple(int(v) for v in dims), dtype=torch.float32, device=base.device)
    with _lib.on_device(base.device):
        rc = _lib.lib().pl_planes_split_strided(base.data_ptr() + 4 * int(offset), d4, s4, mode, float(scale), out.data_ptr(),
                                                _lib.current_stream_ptr())
    _lib.check(rc, "pl_planes_split_strided")
    return out


def _planes_of_flipped_t(w_ohwi, scale, mode):
    """Planes of w.flip(1, 2).permute(3, 1, 2, 0) -- [Cin][KH][KW][Cout], the stride-1 data gradient's kernel -- straight from
    the (possibly non-contiguous) OHWI view: the flip is a negative stride."""
    cout, kh, kw, cin = w_ohwi.shape
    S = w_ohwi.stride()
    if cout % 4 or w_ohwi.dtype != torch.float32:
        return _planes_of(w_ohwi.flip(1, 2).permute(3, 1, 2, 0), scale, mode)
    return _planes_of_strided(w_ohwi, (cin, kh, kw, cout), (S[3], -S[1], -S[2], S[0]), (kh - 1) * S[1] + (kw - 1) * S[2],
                              scale, mode)


def _planes_of(t, scale, mode=_lib.PL_F16X3):
    """fp32 tensor -> carrier holding its planes (weights: once per step and direction, they are small).  A PL_BF16
    carrier uses the first half of its bytes."""
    if not t.is_contiguous() and 1 <= t.dim() <= 4 and t.shape[-1] % 4 == 0 and t.dtype == torch.float32 and t.numel() > 0:
        return _planes_of_strided(t, t.shape, t.stride(), 0, scale, mode)       # (a permuted / transposed weight view)
    t = t.contiguous()
    out = torch.empty_like(t)
    with _lib.on_device(t.device):
        rc = _lib.lib().pl_planes_split(t.data_ptr(), t.numel(), mode, float(scale), out.data_ptr(),
                                        _lib.current_stream_ptr())
    _lib.check(rc, "pl_planes_split")
    return out


def _stat_buffer(rows, cols, dev):
    """[2][groups][cols]: what a GEMM epilogue writes as training-mode BatchNorm partial statistics of its output."""
    return torch.empty(2, _lib.lib().pl_gemm_stat_groups(rows), cols, device=dev)


def _gemm_planes_raw(layout, a, a_rows_cols, b, b_rows_cols, M, N, K, out_scale, dyn_inv=None, mode=_lib.PL_F16X3, stat=None,
                     out=None):
    """C [M][N] fp32 from two carriers (planes of row-major matrices a_rows_cols / b_rows_cols); out: where to put it."""
    L = _lib.lib()
    dev = a.device
    C = torch.empty(M, N, device=dev) if out is None else out
    splits = L.pl_gemm_planes_splits(M, N, K)
    slabs = torch.empty(splits * M * N, device=dev) if splits > 1 else None
    with _lib.on_device(dev):
        rc = L.pl_gemm_planes_raw(layout, mode, a.data_ptr(), a_rows_cols[0] * a_rows_cols[1], a_rows_cols[1],
                                  b.data_ptr(), b_rows_cols[0] * b_rows_cols[1], b_rows_cols[1], C.data_ptr(), M, N, K,
                                  None, float(out_scale), dyn_inv.data_ptr() if dyn_inv is not None else None,
                                  slabs.data_ptr() if slabs is not None else None,
                                  stat.data_ptr() if stat is not None else None, _lib.current_stream_ptr())
    _lib.check(rc, "pl_gemm_planes_raw")
    return C


def planes_conv_supported(rows, cin, cout):
    """Shapes the planes GEMMs of a 1x1 convolution take: whole 32-k tiles in every contraction (Cin forward, Cout for
    the data gradient, pixels for the weight gradient), 8-element rows, 32-bit byte offsets inside one operand."""
    return cin % 32 == 0 and cout % 32 == 0 and rows % 32 == 0 and rows * max(cin, cout) * 4 < (1 << 31)


class _Conv1x1PlanesFn(torch.autograd.Function):
    """z = x W^T for a 1x1 convolution: xp = carrier of x's planes [rows][Cin], w [Cout][Cin] fp32 -> z [rows][Cout] fp32.
    backward takes dz as a carrier (written by the BatchNorm behind this convolution, see PlaneLink)."""

    @staticmethod
    def forward(ctx, xp, wparam, link):
        rows, cin = xp.shape
        cout = wparam.shape[0]
        w = wparam.float().reshape(cout, cin)
        wp = _planes_of(w, WEIGHT_PLANE_SCALE, link.mode)
        # the BatchNorm behind this convolution reads the epilogue's statistics (no pass over z) -- unless the GEMM is split over K
        stat = (_stat_buffer(rows, cout, xp.device) if _lib.lib().pl_gemm_planes_splits(rows, cout, cin) == 1
                else torch.empty(0, device=xp.device))
        z = _gemm_planes_raw(0, xp, (rows, cin), wp, (cout, cin), rows, cout, cin,
                             1.0 / (ACT_PLANE_SCALE * WEIGHT_PLANE_SCALE), None, link.mode, stat if stat.numel() else None)
        ctx.save_for_backward(xp)
        ctx.link, ctx.wparam = link, wparam
        ctx.mark_non_differentiable(stat)
        ctx.set_materialize_grads(False)       # (no zero tensor the size of `stat` per backward)
        return z, stat

    @staticmethod
    def backward(ctx, dzp, _gstat):
        if dzp is None:
            return None, None, None
        (xp,) = ctx.saved_tensors
        wparam = ctx.wparam
        rows, cin = xp.shape
        cout = wparam.shape[0]
        w = wparam.detach().float().reshape(cout, cin)
        dzp = dzp.contiguous()
        inv, mode = ctx.link.dz_scale[1:], ctx.link.mode   # 1 / S of the dz planes (device scalar; fp16 planes only)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wtp = _planes_of(w.t(), WEIGHT_PLANE_SCALE, mode)  # [Cin][Cout]: the data gradient is NT on W^T
            dx = _gemm_planes_raw(0, dzp, (rows, cout), wtp, (cin, cout), rows, cin, cout, 1.0 / WEIGHT_PLANE_SCALE, inv, mode)
        if ctx.needs_input_grad[1]:
            def make(out):
                o2 = out.reshape(cout, cin) if out is not None else None
                r = _gemm_planes_raw(2, dzp, (rows, cout), xp, (rows, cin), cout, cin, rows, 1.0 / ACT_PLANE_SCALE, inv, mode, out=o2)
                return r.reshape(wparam.shape)
            dw = _pgrad(wparam, make)
        return dx, dw, None


def conv1x1_planes(xp, weight_oihw, link):
    """xp: carrier [B, H, W, Cin] of the input's planes; weight: the nn.Conv2d parameter [Cout][Cin][1][1]."""
    shape = xp.shape
    cout, cin = weight_oihw.shape[0], weight_oihw.shape[1]
    z, stat = _Conv1x1PlanesFn.apply(xp.reshape(-1, cin), weight_oihw, link)
    link.stat = stat if stat.numel() else None
    return z.reshape(*shape[:-1], cout)


class _BNPlanesFn(torch.autograd.Function):
    """_BNReLUFn with the output as fp32 (out_planes False) or ONLY as a carrier of its planes (True: the next op is a
    planes convolution), and -- link given: the convolution that produced z is a planes convolution -- dz returned as a
    carrier of its planes, range-scaled on the device (link.dz_scale)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, batches, eps, momentum, relu, out_planes, link, mode):
        shape = z.shape
        C = shape[-1]
        z2 = z.contiguous().reshape(-1, C)
        rows = z2.shape[0]
        for name, t in (("z", z2), ("gamma", gamma), ("beta", beta), ("running_mean", running_mean),
                        ("running_var", running_var)):
            _lib.require_device_tensor(t, name)
        _lib.require_device_tensor(batches, "num_batches_tracked", torch.int64)
        dev, L = z2.device, _lib.lib()
        y = torch.empty_like(z2)                      # fp32 values, or the carrier
        bits = torch.empty(rows, 4 * ((C + 255) // 256), dtype=torch.int64, device=dev)
        mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        scratch = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=dev)
        gstat = link.stat if link is not None else None
        if link is not None:
            link.stat = None
        with _lib.on_device(dev):
            rc = L.pl_bn_train_fwd_ex(z2.data_ptr(), rows, C, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                                      running_mean.data_ptr(), running_var.data_ptr(), batches.data_ptr(), int(relu),
                                      None if out_planes else y.data_ptr(), bits.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), scratch.data_ptr(), y.data_ptr() if out_planes else None,
                                      mode, gstat.data_ptr() if gstat is not None else None, None, _lib.current_stream_ptr())
        _lib.check(rc, "pl_bn_train_fwd_ex")
        ctx.save_for_backward(z2, bits, mean, rstd, gamma)
        ctx.shape, ctx.link, ctx.gparam, ctx.bparam = shape, link, gamma, beta
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        z2, bits, mean, rstd, gamma = ctx.saved_tensors
        rows, C = z2.shape
        dy2 = dy.contiguous().reshape(rows, C)
        dev, L = z2.device, _lib.lib()
        dz = torch.empty_like(z2)
        scratch = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=dev)
        link = ctx.link
        if link is not None:
            link.dz_scale = torch.empty(2, device=dev)

        def run(dgamma, dbeta):
            with _lib.on_device(dev):
                rc = L.pl_bn_train_bwd_ex(dy2.data_ptr(), bits.data_ptr(), z2.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          gamma.data_ptr(), rows, C, None if link is not None else dz.data_ptr(),
                                          dgamma.data_ptr(), dbeta.data_ptr(), scratch.data_ptr(),
                                          dz.data_ptr() if link is not None else None,
                                          link.mode if link is not None else _lib.PL_F16X3,
                                          link.dz_scale.data_ptr() if link is not None else None, _lib.current_stream_ptr())
            _lib.check(rc, "pl_bn_train_bwd_ex")
        dgamma, dbeta = _pgrad2(ctx.gparam, ctx.bparam, C, dev, run)
        return dz.reshape(ctx.shape), dgamma, dbeta, None, None, None, None, None, None, None, None, None


def batchnorm_relu_train_planes(z, bn, relu=True, out_planes=False, link=None, mode=_lib.PL_F16X3):
    if bn.momentum is None:
        raise NotImplementedError("cumulative moving average (momentum=None)")
    return _BNPlanesFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                             float(bn.eps), float(bn.momentum), bool(relu), bool(out_planes), link,
                             link.mode if link is not None else mode)


class _AddReLUPlanesFn(torch.autograd.Function):
    """relu(a + b) as fp32 (the next join's identity) AND as a carrier of its planes (the next block's 1x1
    convolutions); backward: one masked pass over the SUM of the two gradients (no separate add)."""

    @staticmethod
    def forward(ctx, a, b, mode):
        if a.shape != b.shape:
            raise ValueError("add_relu: shapes differ")
        C = a.shape[-1]
        a2, b2 = a.contiguous().reshape(-1, C), b.contiguous().reshape(-1, C)
        _lib.require_device_tensor(a2, "a")
        _lib.require_device_tensor(b2, "b")
        rows = a2.shape[0]
        out, outp = torch.empty_like(a2), torch.empty_like(a2)
        bits = torch.empty(rows, 4 * ((C + 255) // 256), dtype=torch.int64, device=a2.device)
        with _lib.on_device(a2.device):
            rc = _lib.lib().pl_add_relu_fwd_ex(a2.data_ptr(), b2.data_ptr(), rows, C, out.data_ptr(), bits.data_ptr(),
                                               outp.data_ptr(), mode, _lib.current_stream_ptr())
        _lib.check(rc, "pl_add_relu_fwd_ex")
        ctx.save_for_backward(bits)
        ctx.shape = a.shape
        ctx.set_materialize_grads(False)       # an unused output's gradient arrives as None, not as a zero map
        return out.reshape(a.shape), outp.reshape(a.shape)

    @staticmethod
    def backward(ctx, g, gp):
        (bits,) = ctx.saved_tensors
        C = ctx.shape[-1]
        if g is None:
            g, gp = gp, None
        if g is None:
            return None, None, None
        g2 = g.contiguous().reshape(-1, C)
        gp2 = gp.contiguous().reshape(-1, C) if gp is not None else None
        dx = torch.empty_like(g2)
        with _lib.on_device(g2.device):
            rc = _lib.lib().pl_mask_add_by_bits(g2.data_ptr(), gp2.data_ptr() if gp2 is not None else None, bits.data_ptr(),
                                                g2.shape[0], C, dx.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_mask_add_by_bits")
        dx = dx.reshape(ctx.shape)
        return dx, dx, None


def add_relu_planes(a, b, mode=_lib.PL_F16X3):
    return _AddReLUPlanesFn.apply(a, b, mode)


class _ToPlanesFn(torch.autograd.Function):
    """fp32 tensor -> carrier of its planes (the stem's max-pool output, read by layer1's first 1x1 convolutions)."""

    @staticmethod
    def forward(ctx, x, mode):
        return _planes_of(x, ACT_PLANE_SCALE, mode)

    @staticmethod
    def backward(ctx, g):
        return g, None


def to_planes(x, mode=_lib.PL_F16X3):
    return _ToPlanesFn.apply(x, mode)


def _conv_planes_fwd(xp, x_shape, wp, w_shape, stride, pad, out_scale, dyn_inv=None, mode=_lib.PL_F16X3, stat=None):
    """pl_conv2d_planes_fwd on carriers: xp planes of x [B][H][W][Cin], wp planes of the OHWI kernel w_shape."""
    B, H, W, cin = x_shape
    cout, kh, kw, _ = w_shape
    sh, sw = stride if isinstance(stride, tuple) else (stride, stride)
    ph, pw, pwr = pad if isinstance(pad, tuple) else (pad, pad, pad)          # (above = below, left, right)
    ho, wo = (H + 2 * ph - kh) // sh + 1, (W + pw + pwr - kw) // sw + 1
    y = torch.empty(B, ho, wo, cout, device=xp.device)
    with _lib.on_device(xp.device):
        rc = _lib.lib().pl_conv2d_planes_fwd_hw(mode, xp.data_ptr(), B * H * W * cin, B, H, W, cin, wp.data_ptr(),
                                                cout * kh * kw * cin, cout, kh, kw, sh, sw, ph, pw, pwr, y.data_ptr(), float(out_scale),
                                                dyn_inv.data_ptr() if dyn_inv is not None else None,
                                                stat.data_ptr() if stat is not None else None, _lib.current_stream_ptr())
    _lib.check(rc, "pl_conv2d_planes_fwd")
    return y


class _PlanesTwinFn(torch.autograd.Function):
    """xp (the carrier of x's planes) presented as a function of x, the fp32 twin the same kernel wrote.  A block with a
    downsample branch reads its input's planes twice (conv1, downsample[0]) and its fp32 twin not at all: routing the
    downsample's input gradient to x instead of xp hands the producing join's backward its two gradients separately -- it
    adds them inside its masked pass (pl_mask_add_by_bits) -- where autograd would otherwise sum two maps for xp with a
    pass of its own (1 GB each at layer2.0, B = 256)."""

    @staticmethod
    def forward(ctx, x, xp):
        return xp.view_as(xp)

    @staticmethod
    def backward(ctx, g):
        return g, None


def planes_twin(x, xp):
    return _PlanesTwinFn.apply(x, xp.detach()) if x.requires_grad else xp


class _ConvKxKPlanesFn(torch.autograd.Function):
    """KxK convolution (the Bottleneck's conv2, the stride-2 downsample) on the planes GEMM with the input gathered by the
    loader waves: xp carrier of x's planes [B][H][W][Cin], w OHWI fp32 -> z [B][Ho][Wo][Cout] fp32.  backward takes dz as a
    carrier (PlaneLink): the data gradient is the same kernel on dz with the kernel flipped and transposed (a stride-2
    convolution's dz is first spread over the even pixels of a zero map), the weight gradient the TN GEMM over the output
    pixels with the gathered x."""

    @staticmethod
    def forward(ctx, xp, wparam, stride, pad, link):
        w = wparam.float().permute(0, 2, 3, 1)         # the OHWI kernel as a VIEW of the [Cout][Cin][KH][KW] parameter
        wp = _planes_of(w, WEIGHT_PLANE_SCALE, link.mode)
        B, H, W, _ = xp.shape
        cout, kh, kw, _ = w.shape
        rows = B * ((H + 2 * pad - kh) // stride + 1) * ((W + 2 * pad - kw) // stride + 1)
        stat = _stat_buffer(rows, cout, xp.device)
        z = _conv_planes_fwd(xp, xp.shape, wp, w.shape, stride, pad, 1.0 / (ACT_PLANE_SCALE * WEIGHT_PLANE_SCALE), None, link.mode,
                             stat)
        ctx.save_for_backward(xp)
        ctx.geom, ctx.link, ctx.wparam = (stride, pad), link, wparam
        ctx.mark_non_differentiable(stat)
        ctx.set_materialize_grads(False)
        return z, stat

    @staticmethod
    def backward(ctx, dzp, _gstat):
        if dzp is None:
            return None, None, None, None, None
        (xp,) = ctx.saved_tensors
        wparam = ctx.wparam
        w = wparam.detach().float().permute(0, 2, 3, 1)
        stride, pad = ctx.geom
        B, H, W, cin = xp.shape
        cout, kh, kw, _ = w.shape
        _, ho, wo, _ = dzp.shape
        dzp = dzp.contiguous()
        inv, mode = ctx.link.dz_scale[1:], ctx.link.mode
        npl = 2 if mode == _lib.PL_F16X3 else 1
        dx = dw = None
        if ctx.needs_input_grad[0]:
            even = stride == 2 and H % 2 == 0 and W % 2 == 0 and 2 * ho == H and 2 * wo == W
            if even and kh == 1 and kw == 1 and pad == 0 and planes_conv_supported(B * ho * wo, cin, cout):
                # 1x1 stride 2 (the downsample): dz W at the output resolution, placed on the even pixels of a zero map
                wtp = _planes_of(w.reshape(cout, cin).t(), WEIGHT_PLANE_SCALE, mode)
                low = _gemm_planes_raw(0, dzp, (B * ho * wo, cout), wtp, (cin, cout), B * ho * wo, cin, cout,
                                       1.0 / WEIGHT_PLANE_SCALE, inv, mode)
                dx = torch.empty(B, H, W, cin, device=dzp.device)
                with _lib.on_device(dzp.device):
                    rc = _lib.lib().pl_upsample2x_zero_nhwc(low.data_ptr(), B, ho, wo, cin, dx.data_ptr(), _lib.current_stream_ptr())
                _lib.check(rc, "pl_upsample2x_zero_nhwc")
            elif even and kh == 3 and kw == 3 and pad == 1 and planes_deconv_supported(B, ho, wo, cout, cin):
                # 3x3 stride 2 pad 1: its transpose is ConvTranspose2d(4, 2, 1) with this filter in the top-left 3x3 of a
                # zero 4x4 one (ih = 2 oh - 1 + kh) -- four 2x2-tap gathers at the OUTPUT resolution, 4 taps per input
                # pixel instead of the 9 of the stride-1 gather over a zero-spread map (and no map to fill)
                w4 = torch.zeros(cout, cin, 4, 4, device=w.device)
                w4[:, :, :3, :3] = w.permute(0, 3, 1, 2)
                wsub = _planes_of(deconv_subkernels(w4), WEIGHT_PLANE_SCALE, mode)
                dx = torch.empty(B, H, W, cin, device=dzp.device)
                with _lib.on_device(dzp.device):
                    rc = _lib.lib().pl_deconv4x4s2_planes_fwd(mode, dzp.data_ptr(), B * ho * wo * cout, B, ho, wo, cout,
                                                              wsub.data_ptr(), wsub.numel(), cin, dx.data_ptr(),
                                                              1.0 / WEIGHT_PLANE_SCALE, inv.data_ptr(), _lib.current_stream_ptr())
                _lib.check(rc, "pl_deconv4x4s2_planes_fwd")
            elif stride == 1:
                wf = _planes_of_flipped_t(w, WEIGHT_PLANE_SCALE, mode)                          # [Cin][KH][KW][Cout]
                dx = _conv_planes_fwd(dzp, (B, ho, wo, cout), wf, (cin, kh, kw, cout), 1, kh - 1 - pad, 1.0 / WEIGHT_PLANE_SCALE, inv, mode)
            else:
                wf = _planes_of_flipped_t(w, WEIGHT_PLANE_SCALE, mode)
                # dz at the even pixels of a zero map the size the stride-1 gradient expects (planes are 16-bit: as int16)
                hu, wu = H + 2 * pad - kh + 1, W + 2 * pad - kw + 1
                up = torch.zeros(2, B, hu, wu, cout, dtype=torch.int16, device=dzp.device)
                up[:npl, :, ::stride, ::stride][:, :, :ho, :wo] = \
                    dzp.reshape(-1).view(torch.int16)[:npl * B * ho * wo * cout].reshape(npl, B, ho, wo, cout)
                src, shape = up.reshape(-1).view(torch.float32).reshape(B, hu, wu, cout), (B, hu, wu, cout)
                dx = _conv_planes_fwd(src, shape, wf, (cin, kh, kw, cout), 1, kh - 1 - pad, 1.0 / WEIGHT_PLANE_SCALE, inv, mode)
        if ctx.needs_input_grad[1]:
            L = _lib.lib()
            n = cout * kh * kw * cin
            splits = L.pl_gemm_planes_splits(cout, kh * kw * cin, B * ho * wo)
            slabs = torch.empty(splits * n, device=dzp.device) if splits > 1 else None
            dw = torch.empty(cout, kh, kw, cin, device=dzp.device)
            with _lib.on_device(dzp.device):
                rc = L.pl_conv2d_planes_wgrad(mode, dzp.data_ptr(), B * ho * wo * cout, xp.data_ptr(), B * H * W * cin,
                                              B, H, W, cin, cout, kh, kw, stride, pad, dw.data_ptr(), 1.0 / ACT_PLANE_SCALE,
                                              inv.data_ptr(), slabs.data_ptr() if slabs is not None else None,
                                              _lib.current_stream_ptr())
            _lib.check(rc, "pl_conv2d_planes_wgrad")
            dw_ohwi = dw
            dw = _pgrad(wparam, lambda out: _oihw_from_ohwi(dw_ohwi, out))
        return dx, dw, None, None, None


# ---------------------------------------------------------------------------------------------------------------------
# The stem (phase4_joined/Resnet.py:112-113,137: Conv2d(3, 64, 7, stride 2, padding 3, bias=False)) on the planes GEMM.
# K = 7*7*3 = 147 fits nothing the matrix pipeline stages, so the frame is padded to 4 channels and read as pixel PAIRS
# [B][H][W/2][8]; the window of output (oh, ow) in kernel row kh is then the four pairs ow - 2 .. ow + 1 of input row
# 2 oh - 3 + kh: 32 contiguous values = one k-tile, of which the first pixel (kw = -1) and every fourth channel meet zero
# weights.  A convolution with 7 x 4 taps of 8 channels, stride (2, 1), padding 3 / 3 / 2 left / 1 right: K = 224 (pl_conv2d_planes_fwd_hw;
# the weight gradient is the gathered TN GEMM with the same geometry, its 224 columns folded back to the 147 real ones).
# Round 3; before: a direct fp32 kernel on the vector unit (1.45 ms forward, 1.28 ms weight gradient at B = 256).
# ---------------------------------------------------------------------------------------------------------------------
def stem_planes_supported(B, H, W, cin, cout, kh, kw, stride, pad):
    return (cin == 3 and cout == 64 and kh == 7 and kw == 7 and stride == 2 and pad == 3 and H % 2 == 0 and W % 4 == 0 and
            (B * (H // 2) * (W // 2)) % 256 == 0 and B * H * W * 4 * 2 < (1 << 31))


def stem_input_planes(frames_nhwc, mode=_lib.PL_F16X3):
    """[B, H, W, 3] fp32 frames -> carrier of the planes of the 4-channel pixel-pair view [B, H, W/2, 8]."""
    B, H, W, _ = frames_nhwc.shape
    x4 = torch.nn.functional.pad(frames_nhwc.float(), (0, 1))
    return _planes_of(x4.reshape(B, H, W // 2, 8), ACT_PLANE_SCALE, mode)


def _stem_weight_pairs(w_oihw, ohwi=False):
    """[64][3][7][7] (or, ohwi: [64][7][7][3]) -> the 7 x 4-tap kernel on pixel pairs [64][7][4][8]: pixel p of the 8-pixel
    window is kw = p - 1."""
    cout = w_oihw.shape[0]
    wv = torch.zeros(cout, 7, 8, 4, device=w_oihw.device, dtype=torch.float32)
    wv[:, :, 1:, :3] = w_oihw.float() if ohwi else w_oihw.float().permute(0, 2, 3, 1)
    return wv.reshape(cout, 7, 4, 8)


class _StemPlanesFn(torch.autograd.Function):
    """z [B][H/2][W/2][64] fp32 (+ the epilogue's BatchNorm statistics) from the frame's pair-view planes and the conv1
    parameter; backward: the weight gradient only (frames need none)."""

    @staticmethod
    def forward(ctx, xp, wparam, link):
        B, H, W2, _ = xp.shape
        wp = _planes_of(_stem_weight_pairs(wparam.detach()), WEIGHT_PLANE_SCALE, link.mode)
        rows = B * (H // 2) * W2
        stat = _stat_buffer(rows, 64, xp.device)
        z = _conv_planes_fwd(xp, xp.shape, wp, (64, 7, 4, 8), (2, 1), (3, 2, 1), 1.0 / (ACT_PLANE_SCALE * WEIGHT_PLANE_SCALE), None,
                             link.mode, stat)
        ctx.save_for_backward(xp)
        ctx.link, ctx.wparam = link, wparam
        ctx.mark_non_differentiable(stat)
        ctx.set_materialize_grads(False)
        return z, stat

    @staticmethod
    def backward(ctx, dzp, _gstat):
        if dzp is None or not ctx.needs_input_grad[1]:
            return None, None, None
        (xp,) = ctx.saved_tensors
        wparam = ctx.wparam
        B, H, W2, _ = xp.shape
        _, ho, wo, cout = dzp.shape
        dzp = dzp.contiguous()
        inv, mode = ctx.link.dz_scale[1:], ctx.link.mode
        L = _lib.lib()
        n = cout * 7 * 4 * 8
        splits = L.pl_gemm_planes_splits(cout, 7 * 4 * 8, B * ho * wo)
        slabs = torch.empty(splits * n, device=dzp.device) if splits > 1 else None
        dwp = torch.empty(cout, 7, 8, 4, device=dzp.device)
        with _lib.on_device(dzp.device):
            rc = L.pl_conv2d_planes_wgrad_hw(mode, dzp.data_ptr(), B * ho * wo * cout, xp.data_ptr(), B * H * W2 * 8, B, H, W2, 8,
                                             cout, 7, 4, 2, 1, 3, 2, 1, dwp.data_ptr(), 1.0 / ACT_PLANE_SCALE, inv.data_ptr(),
                                             slabs.data_ptr() if slabs is not None else None, _lib.current_stream_ptr())
        _lib.check(rc, "pl_conv2d_planes_wgrad_hw")

        def make(out):
            g = dwp[:, :, 1:, :3].permute(0, 3, 1, 2)              # the 147 real taps, as OIHW
            if out is None:
                return g.contiguous()
            out.copy_(g)
            return out
        return None, _pgrad(wparam, make), None


def stem_planes(xp, weight_oihw, link):
    """xp: stem_input_planes(frames); weight: the conv1 parameter [64][3][7][7]; z fp32 [B][H/2][W/2][64]."""
    z, stat = _StemPlanesFn.apply(xp, weight_oihw, link)
    link.stat = stat
    return z


def conv_planes(xp, weight_oihw, stride, pad, link):
    """xp: carrier [B, H, W, Cin]; weight: the nn.Conv2d parameter [Cout][Cin][KH][KW]."""
    # (the OHWI kernel as a VIEW of the parameter: its planes -- and the flipped / transposed ones of the backward -- are
    #  gathered by pl_planes_split_strided, no layout copy)
    z, stat = _ConvKxKPlanesFn.apply(xp, weight_oihw, stride, pad, link)
    link.stat = stat
    return z


def planes_convk_supported(B, H, W, cin, cout, k, stride, pad):
    """Whole 32-k tiles in every contraction (Cin per tap forward, Cout per tap for the data gradient, output pixels for the
    weight gradient) and gathered tensors -- x, and the (zero-spread, for stride 2) dz map of the data gradient -- whose
    planes stay below the DMA descriptor's 2 GiB."""
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    hu, wu = (ho, wo) if stride == 1 else (H + 2 * pad - k + 1, W + 2 * pad - k + 1)
    big = max(B * H * W * cin, B * hu * wu * cout)
    return cin % 32 == 0 and cout % 32 == 0 and (B * ho * wo) % 32 == 0 and big * 2 < (1 << 31) - 64


class _DeconvPlanesFn(torch.autograd.Function):
    """nn.ConvTranspose2d(4, 2, 1, bias=False) on the planes GEMM: xp carrier of x's planes [B][H][W][Cin], weight in torch's
    [Cin][Cout][4][4] layout -> y [B][2H][2W][Cout] fp32 (four 2x2-tap gathers, one per output parity).  y = C^T x for the
    stride-2 convolution C with the same filter, so the backward (dy as a carrier, PlaneLink) is dx = C dy -- the gathered
    forward kernel, 4x4 stride 2 -- and dW = the weight gradient of C with the roles of x and dy exchanged."""

    @staticmethod
    def forward(ctx, xp, weight, link):
        B, H, W, cin = xp.shape
        cout = weight.shape[1]
        wsub = _planes_of(deconv_subkernels(weight.float()), WEIGHT_PLANE_SCALE, link.mode)
        y = torch.empty(B, 2 * H, 2 * W, cout, device=xp.device)
        with _lib.on_device(xp.device):
            rc = _lib.lib().pl_deconv4x4s2_planes_fwd(link.mode, xp.data_ptr(), xp.numel(), B, H, W, cin, wsub.data_ptr(),
                                                      wsub.numel(), cout, y.data_ptr(),
                                                      1.0 / (ACT_PLANE_SCALE * WEIGHT_PLANE_SCALE), None,
                                                      _lib.current_stream_ptr())
        _lib.check(rc, "pl_deconv4x4s2_planes_fwd")
        ctx.save_for_backward(xp)
        ctx.link, ctx.wparam = link, weight
        return y

    @staticmethod
    def backward(ctx, dyp):
        (xp,) = ctx.saved_tensors
        weight = ctx.wparam
        B, H, W, cin = xp.shape
        cout = weight.shape[1]
        dyp = dyp.contiguous()
        inv, mode = ctx.link.dz_scale[1:], ctx.link.mode
        dx = dw = None
        if ctx.needs_input_grad[0]:       # C dy: OHWI kernel [Cin][4][4][Cout]
            wc = _planes_of(weight.float().permute(0, 2, 3, 1), WEIGHT_PLANE_SCALE, mode)
            dx = _conv_planes_fwd(dyp, (B, 2 * H, 2 * W, cout), wc, (cin, 4, 4, cout), 2, 1, 1.0 / WEIGHT_PLANE_SCALE, inv, mode)
        if ctx.needs_input_grad[1]:       # wgrad of C: "output gradient" x [B][H][W][Cin], gathered input dy -> [Cin][4][4][Cout]
            L = _lib.lib()
            n = cin * 16 * cout
            splits = L.pl_gemm_planes_splits(cin, 16 * cout, B * H * W)
            slabs = torch.empty(splits * n, device=dyp.device) if splits > 1 else None
            dwc = torch.empty(cin, 4, 4, cout, device=dyp.device)
            with _lib.on_device(dyp.device):
                rc = L.pl_conv2d_planes_wgrad(mode, xp.data_ptr(), xp.numel(), dyp.data_ptr(), dyp.numel(), B, 2 * H, 2 * W,
                                              cout, cin, 4, 4, 2, 1, dwc.data_ptr(), 1.0 / ACT_PLANE_SCALE, inv.data_ptr(),
                                              slabs.data_ptr() if slabs is not None else None, _lib.current_stream_ptr())
            _lib.check(rc, "pl_conv2d_planes_wgrad")
            dw = _pgrad(weight, lambda out: _oihw_from_ohwi(dwc, out))      # [Cin][4][4][Cout] -> [Cin][Cout][4][4]
        return dx, dw, None


def deconv4x4s2_planes(xp, weight_iohw, link):
    return _DeconvPlanesFn.apply(xp, weight_iohw, link)


def planes_deconv_supported(B, H, W, cin, cout):
    return (cin % 32 == 0 and cout % 32 == 0 and (B * H * W) % 32 == 0 and
            max(B * H * W * cin, B * 4 * H * W * cout) * 4 < (1 << 31))


class _ConvBiasPlanesFn(torch.autograd.Function):
    """The head's final 1x1 convolution with bias (Model.py:66-69) on the planes GEMM: xp carrier [rows][Cin], w [Cout][Cin],
    bias [Cout] -> logits [rows][Cout] fp32.  backward takes dlogits as a carrier written by the soft-argmax backward
    (heads.py, PlaneLink): data gradient NT on W^T, weight gradient TN over the pixels, bias gradient = column sums of the
    planes."""

    @staticmethod
    def forward(ctx, xp, wparam, bias, link):
        rows, cin = xp.shape
        cout = wparam.shape[0]
        w = wparam.float().reshape(cout, cin)
        mode = link.mode
        wp = _planes_of(w, WEIGHT_PLANE_SCALE, mode)
        L = _lib.lib()
        y = torch.empty(rows, cout, device=xp.device)
        with _lib.on_device(xp.device):
            rc = L.pl_gemm_planes_raw(0, mode, xp.data_ptr(), rows * cin, cin, wp.data_ptr(), cout * cin, cin, y.data_ptr(), rows,
                                      cout, cin, bias.data_ptr(), 1.0 / (ACT_PLANE_SCALE * WEIGHT_PLANE_SCALE), None, None, None,
                                      _lib.current_stream_ptr())
        _lib.check(rc, "pl_gemm_planes_raw")
        ctx.save_for_backward(xp)
        ctx.link, ctx.wparam, ctx.bparam = link, wparam, bias
        return y

    @staticmethod
    def backward(ctx, dyp):
        (xp,) = ctx.saved_tensors
        wparam, bparam = ctx.wparam, ctx.bparam
        rows, cin = xp.shape
        cout = wparam.shape[0]
        w = wparam.detach().float().reshape(cout, cin)
        dyp = dyp.contiguous()
        inv, mode = ctx.link.dz_scale[1:], ctx.link.mode
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wtp = _planes_of(w.t(), WEIGHT_PLANE_SCALE, mode)
            dx = _gemm_planes_raw(0, dyp, (rows, cout), wtp, (cin, cout), rows, cin, cout, 1.0 / WEIGHT_PLANE_SCALE, inv, mode)
        if ctx.needs_input_grad[1]:
            def make_w(out):
                o2 = out.reshape(cout, cin) if out is not None else None
                r = _gemm_planes_raw(2, dyp, (rows, cout), xp, (rows, cin), cout, cin, rows, 1.0 / ACT_PLANE_SCALE, inv, mode, out=o2)
                return r.reshape(wparam.shape)
            dw = _pgrad(wparam, make_w)
        if ctx.needs_input_grad[2]:
            L = _lib.lib()

            def make_b(out):
                b = torch.empty(cout, device=dyp.device) if out is None else out
                scratch = torch.empty(L.pl_colsum_scratch_bytes(rows, cout), dtype=torch.uint8, device=dyp.device)
                with _lib.on_device(dyp.device):
                    rc = L.pl_colsum_planes(dyp.data_ptr(), mode, rows, cout, inv.data_ptr(), b.data_ptr(), scratch.data_ptr(),
                                            _lib.current_stream_ptr())
                _lib.check(rc, "pl_colsum_planes")
                return b
            db = _pgrad(bparam, make_b)
        return dx, dw, db, None


def conv1x1_bias_planes(xp, weight_oihw, bias, link):
    shape = xp.shape
    cout, cin = weight_oihw.shape[0], weight_oihw.shape[1]
    y = _ConvBiasPlanesFn.apply(xp.reshape(-1, cin), weight_oihw, bias, link)
    return y.reshape(*shape[:-1], cout)


# ---------------------------------------------------------------------------------------------
# Eval mode on the planes GEMM: conv2d_nhwc / deconv4x4s2_nhwc with planes in (and, optionally, out)
# ---------------------------------------------------------------------------------------------
def _epilogue(scale, shift, bias, relu, resid, y_planes):
    ep = _lib.PLPlanesEpilogue()
    ep.bias = bias.data_ptr() if bias is not None else None
    ep.scale = scale.data_ptr() if scale is not None else None
    ep.shift = shift.data_ptr() if shift is not None else None
    ep.resid = resid.data_ptr() if resid is not None else None
    ep.relu = int(relu)
    ep.y_planes = y_planes.data_ptr() if y_planes is not None else None
    return ep


def conv2d_planes_eval(xp, wp, w_shape, stride=1, padding=0, scale=None, shift=None, bias=None, relu=0, resid=None,
                       want_f32=True, want_planes=False, mode=_lib.PL_F16X3):
    """conv2d_nhwc with the input as a carrier of its planes [B,H,W,Cin] and the OHWI kernel's planes wp (w_shape):
    returns (y fp32 or None, carrier of y's planes or None)."""
    import ctypes
    B, H, W, cin = xp.shape
    cout, kh, kw, _ = w_shape
    sh, sw = stride if isinstance(stride, tuple) else (stride, stride)
    ph, pw, pwr = padding if isinstance(padding, tuple) else (padding, padding, padding)     # (above = below, left, right)
    ho, wo = (H + 2 * ph - kh) // sh + 1, (W + pw + pwr - kw) // sw + 1
    y = torch.empty(B, ho, wo, cout, device=xp.device) if want_f32 else None
    yp = torch.empty(B, ho, wo, cout, device=xp.device) if want_planes else None
    opt = [_opt(scale, "scale", cout), _opt(shift, "shift", cout), _opt(bias, "bias", cout),
           _opt(resid, "resid", B * ho * wo * cout)]
    ep = _epilogue(opt[0], opt[1], opt[2], relu, opt[3], yp)
    with _lib.on_device(xp.device):
        rc = _lib.lib().pl_conv2d_planes_fwd_ep_hw(mode, xp.data_ptr(), B * H * W * cin, B, H, W, cin, wp.data_ptr(),
                                                   cout * kh * kw * cin, cout, kh, kw, sh, sw, ph, pw, pwr,
                                                   y.data_ptr() if y is not None else None,
                                                   1.0 / (ACT_PLANE_SCALE * WEIGHT_PLANE_SCALE), ctypes.byref(ep),
                                                   _lib.current_stream_ptr())
    _lib.check(rc, "pl_conv2d_planes_fwd_ep")
    return y, yp


def deconv_planes_eval(xp, wsubp, cout, scale=None, shift=None, relu=0, want_f32=True, want_planes=False, mode=_lib.PL_F16X3):
    """deconv4x4s2_nhwc the same way: wsubp = planes of deconv_subkernels(weight)."""
    import ctypes
    B, H, W, cin = xp.shape
    y = torch.empty(B, 2 * H, 2 * W, cout, device=xp.device) if want_f32 else None
    yp = torch.empty(B, 2 * H, 2 * W, cout, device=xp.device) if want_planes else None
    ep = _epilogue(_opt(scale, "scale", cout), _opt(shift, "shift", cout), None, relu, None, yp)
    with _lib.on_device(xp.device):
        rc = _lib.lib().pl_deconv4x4s2_planes_fwd_ep(mode, xp.data_ptr(), B * H * W * cin, B, H, W, cin, wsubp.data_ptr(),
                                                     16 * cout * cin, cout, y.data_ptr() if y is not None else None,
                                                     1.0 / (ACT_PLANE_SCALE * WEIGHT_PLANE_SCALE), ctypes.byref(ep),
                                                     _lib.current_stream_ptr())
    _lib.check(rc, "pl_deconv4x4s2_planes_fwd_ep")
    return y, yp



class _BNJoinPlanesFn(torch.autograd.Function):
    """bn3 and the residual join of a Bottleneck in ONE pass: x = relu(bn(z) + identity) as fp32 (the next join's identity) AND
    as a carrier of its planes (the next block's convolutions) -- bn3's output is never materialised.  backward: the masked
    sum of the two incoming gradients (pl_mask_add_by_bits) IS both the identity's gradient and bn3's dy; dz leaves as a
    carrier of its planes (link).  Resnet.py:81-91."""

    @staticmethod
    def forward(ctx, z, identity, gamma, beta, running_mean, running_var, batches, eps, momentum, link):
        shape = z.shape
        C = shape[-1]
        z2, id2 = z.contiguous().reshape(-1, C), identity.contiguous().reshape(-1, C)
        rows = z2.shape[0]
        dev, L = z2.device, _lib.lib()
        x, xp = torch.empty_like(z2), torch.empty_like(z2)
        bits = torch.empty(rows, 4 * ((C + 255) // 256), dtype=torch.int64, device=dev)
        mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        scratch = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=dev)
        gstat, link.stat = link.stat, None
        with _lib.on_device(dev):
            rc = L.pl_bn_train_fwd_ex(z2.data_ptr(), rows, C, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                                      running_mean.data_ptr(), running_var.data_ptr(), batches.data_ptr(), 1, x.data_ptr(),
                                      bits.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scratch.data_ptr(), xp.data_ptr(),
                                      link.mode, gstat.data_ptr() if gstat is not None else None, id2.data_ptr(),
                                      _lib.current_stream_ptr())
        _lib.check(rc, "pl_bn_train_fwd_ex")
        ctx.save_for_backward(z2, bits, mean, rstd, gamma)
        ctx.shape, ctx.link, ctx.gparam, ctx.bparam = shape, link, gamma, beta
        ctx.set_materialize_grads(False)
        return x.reshape(shape), xp.reshape(shape)

    @staticmethod
    def backward(ctx, g, gp):
        z2, bits, mean, rstd, gamma = ctx.saved_tensors
        rows, C = z2.shape
        dev, L, link = z2.device, _lib.lib(), ctx.link
        if g is None:
            g, gp = gp, None
        if g is None:
            return (None,) * 10
        g2 = g.contiguous().reshape(rows, C)
        gp2 = gp.contiguous().reshape(rows, C) if gp is not None else None
        dx = torch.empty_like(g2)                    # masked sum: the identity's gradient and bn3's dy
        dz = torch.empty_like(z2)
        scratch = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=dev)
        link.dz_scale = torch.empty(2, device=dev)

        def run(dgamma, dbeta):
            with _lib.on_device(dev):
                if C >= 256:
                    # one pass writes the masked sum and takes BatchNorm-backward's column sums of it (pl_bn_join_bwd)
                    rc = L.pl_bn_join_bwd(g2.data_ptr(), gp2.data_ptr() if gp2 is not None else None, bits.data_ptr(), z2.data_ptr(),
                                          mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), rows, C, dx.data_ptr(), None,
                                          dgamma.data_ptr(), dbeta.data_ptr(), scratch.data_ptr(), dz.data_ptr(), link.mode,
                                          link.dz_scale.data_ptr(), _lib.current_stream_ptr())
                    _lib.check(rc, "pl_bn_join_bwd")
                else:
                    rc = L.pl_mask_add_by_bits(g2.data_ptr(), gp2.data_ptr() if gp2 is not None else None, bits.data_ptr(), rows, C,
                                               dx.data_ptr(), _lib.current_stream_ptr())
                    _lib.check(rc, "pl_mask_add_by_bits")
                    # (dy = dx where the join's bitmap is set: masking the masked sum again changes nothing)
                    rc = L.pl_bn_train_bwd_ex(dx.data_ptr(), bits.data_ptr(), z2.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                              gamma.data_ptr(), rows, C, None, dgamma.data_ptr(), dbeta.data_ptr(), scratch.data_ptr(),
                                              dz.data_ptr(), link.mode, link.dz_scale.data_ptr(), _lib.current_stream_ptr())
                    _lib.check(rc, "pl_bn_train_bwd_ex")
        dgamma, dbeta = _pgrad2(ctx.gparam, ctx.bparam, C, dev, run)
        return dz.reshape(ctx.shape), dx.reshape(ctx.shape), dgamma, dbeta, None, None, None, None, None, None


def bn_join_planes(z, identity, bn, link):
    if bn.momentum is None:
        raise NotImplementedError("cumulative moving average (momentum=None)")
    return _BNJoinPlanesFn.apply(z, identity, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                 float(bn.eps), float(bn.momentum), link)
