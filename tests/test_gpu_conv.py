"""GPU tests of the NHWC convolution forward (first slice) against stock PyTorch on the
CPU (F.conv2d, fp32 -- the ATen arithmetic the reference's nn.Conv2d / BatchNorm2d dispatch to) at the
phase4 backbone's layer shapes (phase4_joined/Resnet.py: Bottleneck conv2 3x3 with the stage's stride,
1x1 downsample with stride 2, 1x1 conv3 with BN + residual + ReLU)."""
import importlib
import os
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
    (2, 64, 3, 64, 7, 2, 3, True, 1, False, False),        # the 7x7 stem (Cin = 3): im2col fallback
    (2, 16, 64, 64, 3, 1, 1, True, 1, False, False),       # layer1 conv2, 64 -> 64: im2col fallback
    (2, 16, 256, 64, 1, 1, 0, True, 1, False, False),      # layer1 conv1 1x1, Cout = 64: edge GEMM
    (3, 10, 32, 40, 3, 2, 1, False, 2, True, True),        # nothing aligned
    (4, 8, 512, 512, 3, 1, 1, True, 2, True, False),       # layer4 conv2: 8 output tiles -> split over K; the reduce
    (4, 8, 512, 512, 3, 1, 1, False, 1, False, True),      #   kernel applies BN / ReLU / residual / bias
    (4, 8, 2048, 512, 1, 1, 0, True, 1, False, False),     # layer4 conv1 (1x1 = plain GEMM), split over K
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


def test_conv_path_has_no_cpu_fallback(pkg):
    with pytest.raises(pkg.PoseliftError):
        pkg.conv.conv2d_nhwc(torch.randn(2, 16, 16, 64), torch.randn(128, 3, 3, 64))  # CPU tensors
    with pytest.raises(pkg.PoseliftError):
        pkg.conv.maxpool3x3s2_nhwc(torch.randn(2, 16, 16, 64))


@pytest.mark.parametrize("B,H,W,C", [(2, 128, 128, 64), (3, 9, 7, 8)])
def test_maxpool_vs_torch(pkg, B, H, W, C):
    x = torch.randn(B, H, W, C)
    want = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    got = pkg.conv.maxpool3x3s2_nhwc(x.to(DEV)).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,Hi,Cin,Cout,bn", [(2, 8, 2048, 256, True), (2, 16, 256, 256, True), (2, 8, 64, 128, False)])
def test_deconv4x4s2_vs_torch(pkg, B, Hi, Cin, Cout, bn):
    """Model.py:47-63 head layers: ConvTranspose2d(k=4, s=2, p=1, bias=False) + BN(eval) + ReLU."""
    g = torch.Generator().manual_seed(Hi + Cin + Cout)
    x = torch.randn(B, Hi, Hi, Cin, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, generator=g) / np.sqrt(Cin * 4)
    scale = torch.rand(Cout, generator=g) + 0.5 if bn else None
    shift = torch.randn(Cout, generator=g) if bn else None
    want = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=2, padding=1)
    mag = F.conv_transpose2d(x.permute(0, 3, 1, 2).abs().double(), w.abs().double(), stride=2, padding=1)
    if bn:
        want = (want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).clamp_min(0)
        mag = mag * scale.double().view(1, -1, 1, 1)
    dv = lambda t: None if t is None else t.to(DEV)
    got = pkg.conv.deconv4x4s2_nhwc(x.to(DEV), pkg.conv.deconv_subkernels(w).to(DEV), dv(scale), dv(shift), 1 if bn else 0)
    assert got.shape == (B, 2 * Hi, 2 * Hi, Cout)
    err = (got.cpu().double() - want.permute(0, 2, 3, 1)).abs()
    bound = 4e-6 * mag.permute(0, 2, 3, 1) + 1e-6
    assert bool((err <= bound).all()), float((err / bound).max())


@pytest.mark.parametrize("shape", [(3, 5, 7, 37),          # scalar 32x32 tiles, ragged everywhere
                                   (2, 8, 12, 68),          # 16-byte path, ragged 64x64 tiles (P = 96, C = 68)
                                   (2, 16, 16, 192)])       # 16-byte path, whole tiles
def test_nhwc_to_nchw(pkg, shape):
    x = torch.randn(*shape)
    assert torch.equal(pkg.conv.nhwc_to_nchw(x.to(DEV)).cpu(), x.permute(0, 3, 1, 2).contiguous())


def test_g9_resnet50_eval_forward_vs_reference(pkg):
    """The phase4 backbone against phase4_joined/Resnet.py imported and run as-is (golden g9): same seeded
    weights and frames (synth.seeded_state / seeded_frames), eval mode, 53 convolutions deep."""
    from conftest import load_golden
    g = load_golden("g9_resnet50_eval.npz")
    net = pkg.ResNet("resnet50").eval()
    net.load_state_dict(pkg.synth.seeded_state(net.state_dict(), int(g["weight_seed"])))
    net = net.to(DEV)
    frames = pkg.synth.seeded_frames(2, int(g["frame_seed"])).to(DEV)
    feat = net(frames)                                        # NHWC [2, 8, 8, 2048]
    assert tuple(feat.shape) == (2, 8, 8, 2048)
    nchw = feat.permute(0, 3, 1, 2).contiguous().cpu()
    assert tuple(nchw.shape) == tuple(g["shape"])
    scale = float(g["abs_max"])
    err = np.abs(nchw.reshape(-1)[::int(g["sample_stride"])].numpy().astype(np.float64) - g["sample"]).max()
    assert err < 2e-4 * scale, (err, scale)                   # fp32 round-off through 53 layers, relative to the peak
    cm = nchw.mean(dim=(0, 2, 3)).numpy()
    assert np.abs(cm - g["channel_mean"]).max() < 2e-4 * scale


@pytest.mark.parametrize("dtype", ["f16x3", "bf16x6"])
def test_g11_resnet50_training_pass_vs_reference(pkg, dtype):
    """ONE training-mode forward + backward of the phase4 backbone against phase4_joined/Resnet.py imported and run
    as-is in float64 (golden g11: 4 frames of 128 x 128, loss = mean(features^2)): features, every parameter's gradient,
    the BatchNorm running statistics after the pass.  Tolerances: the fixture carries, per tensor, how far the
    REFERENCE'S OWN float32 run is from the float64 one ("floor": 1e-5 at the top of the stack, 1-2 % below the
    BatchNorm layers, whose backward leaves what survives a cancellation); this library's fp32-grade arithmetic gets
    1.5 x that + 1e-4 (measured: a twentieth of it).  Running statistics and features do not cancel: 1e-5 / 3 x floor."""
    from conftest import load_golden
    g = load_golden("g11_resnet50_train.npz")
    net = pkg.ResNet("resnet50", compute_dtype=dtype).train()
    net.load_state_dict(pkg.synth.seeded_state(net.state_dict(), int(g["weight_seed"])))
    net = net.to(DEV)
    n, size = (int(v) for v in g["frames"])
    frames = pkg.synth.seeded_frames(n, int(g["frame_seed"]), size).to(DEV)
    feat = net(frames)                                        # NHWC
    loss = feat.pow(2).mean()
    loss.backward()
    nchw = feat.detach().permute(0, 3, 1, 2).contiguous().cpu().double()
    assert tuple(nchw.shape) == tuple(g["feat_shape"])
    fs = g["feat_sample"]
    assert np.abs(nchw.reshape(-1)[::7].numpy() - fs).max() < 3 * float(g["feat_floor"]) * np.abs(fs).max() + 1e-6
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    worst = []
    for k, p in net.named_parameters():
        want = g["gsample:" + k]
        got = p.grad.detach().cpu().double().reshape(-1)
        assert abs(float(got.norm()) - float(g["gnorm:" + k])) <= (1.5 * float(g["gfloor:" + k]) + 1e-4) * float(g["gnorm:" + k]), k
        stride = max(1, got.numel() // 64)
        # the 64 samples carry 64 / numel of the squared norm on average: scale the bound by the samples' own share
        tol = (1.5 * float(g["gfloor:" + k]) + 1e-4) * float(g["gnorm:" + k]) * max(1.0, (64.0 / got.numel()) ** 0.5 * 8)
        err = float(np.abs(got[::stride][:64].numpy() - want).max())
        worst.append((err / tol, k))
        assert err <= tol, (k, err, tol)
    if os.environ.get("POSELIFT_TEST_VERBOSE"):
        print("g11 worst sample error / tolerance:", sorted(worst)[-4:])
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            want = g["stat:" + k]
            got = v.detach().cpu().double().reshape(-1)[::max(1, v.numel() // 32)][:32].numpy()
            assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max()), k
        elif k.endswith("num_batches_tracked"):
            assert int(v) == int(g["stat:" + k]), k


def test_model3d_eval_forward_vs_torch_cpu(pkg):
    """Model_3D.forward (phase4_joined/Model.py:83-137) end to end -- backbone, three deconvolutions, final
    1x1 convolution, integral soft-argmax -- against the same stock nn modules run by PyTorch on the CPU plus
    the soft-argmax oracle.  Model.py itself cannot be imported here (torchvision, pretrained download):
    the head is restated from its text, PARITY UNPINNED beyond the backbone (g9)."""
    from oracle import heads_oracle
    m = pkg.Model_3D().eval()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
    with torch.no_grad():                                     # keep the logits in a range where the softmax is not one-hot
        m.final_layer.weight.mul_(1e-4)
        m.final_layer.bias.mul_(0.1)
    frames = pkg.synth.seeded_frames(2, 32)
    with torch.no_grad():                                     # plain PyTorch on the CPU, module by module
        r = m.preact
        x = frames.permute(0, 3, 1, 2)
        x = F.max_pool2d(F.relu(r.bn1(r.conv1(x))), 3, 2, 1)
        for li in (1, 2, 3, 4):
            for blk in getattr(r, f"layer{li}"):
                idn = x if blk.downsample is None else blk.downsample(x)
                o = F.relu(blk.bn1(blk.conv1(x)))
                o = F.relu(blk.bn2(blk.conv2(o)))
                x = F.relu(blk.bn3(blk.conv3(o)) + idn)
        logits_ref = m.final_layer(m.deconv_layers(x))
    want = heads_oracle.soft_argmax(logits_ref.double().numpy(), 17, 64, True)
    md = m.to(DEV)
    logits = md.heatmap_logits(frames.to(DEV))
    assert tuple(logits.shape) == (2, 1088, 64, 64)
    lscale = float(logits_ref.abs().max())
    assert float((logits.cpu() - logits_ref).abs().max()) < 3e-4 * lscale
    got = md(frames.to(DEV)).cpu().numpy()
    assert got.shape == (2, 51)
    assert np.abs(got - want.reshape(2, 51)).max() < 2e-3     # coordinates in (-1, 1): 64 voxels per unit... 2e-3 = 0.06 voxel


def test_model2d_eval_forward_vs_torch_cpu(pkg):
    """Model_2D.forward (phase5_loop/Model_2d.py:87-136: NCHW frames, 17 heat-maps, coordinates c/64 in (0, 1))
    against the same stock modules on the CPU + the oracle.  Restated from text: PARITY UNPINNED beyond g9."""
    from oracle import heads_oracle
    m = pkg.Model_2D().eval()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 41))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-4)
    frames = pkg.synth.seeded_frames(2, 42).permute(0, 3, 1, 2).contiguous()          # NCHW
    with torch.no_grad():
        r = m.preact
        x = F.max_pool2d(F.relu(r.bn1(r.conv1(frames))), 3, 2, 1)
        for li in (1, 2, 3, 4):
            for blk in getattr(r, f"layer{li}"):
                idn = x if blk.downsample is None else blk.downsample(x)
                o = F.relu(blk.bn1(blk.conv1(x)))
                o = F.relu(blk.bn2(blk.conv2(o)))
                x = F.relu(blk.bn3(blk.conv3(o)) + idn)
        logits_ref = m.final_layer(m.deconv_layers(x))
    want = heads_oracle.soft_argmax(logits_ref.double().numpy(), 17, 1, False)
    got = m.to(DEV)(frames.to(DEV)).cpu().numpy()
    assert got.shape == (2, 34) and np.abs(got - want.reshape(2, 34)).max() < 1e-3
    with pytest.raises(ValueError):
        m(torch.zeros(2, 256, 256, 3, device=DEV))


@pytest.mark.parametrize("B", [1, 3])
def test_model3d_any_batch_size(pkg, B):
    """Batch sizes whose deep layers do not make whole 128-row tiles (B = 1: 64 rows at 8x8) fall back to the
    im2col + edge GEMM path: same result as the rows of a bigger batch."""
    m = pkg.Model_3D().eval()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-4)
    m = m.to(DEV)
    frames = pkg.synth.seeded_frames(4, 77).to(DEV)
    full = m(frames)
    part = m(frames[:B])
    assert part.shape == (B, 51)
    assert float((part - full[:B]).abs().max()) < 2e-4


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,pad", [
    (4, 32, 128, 128, 3, 1, 1),      # layer2 conv2
    (4, 32, 128, 128, 3, 2, 1),      # stride on the 3x3
    (4, 16, 256, 256, 3, 1, 1),
    (2, 32, 256, 512, 1, 2, 0),      # downsample 1x1 stride 2
    (2, 16, 512, 128, 1, 1, 0),      # 1x1
    (2, 16, 64, 128, 3, 1, 1),       # Cin = 64, 576 columns: not whole tiles -> im2col fallback
    (2, 16, 64, 64, 3, 1, 1),        # layer1 conv2
    (2, 32, 3, 64, 7, 2, 3),         # the stem (dgrad not needed: x is the image): stem7x7_c3_wgrad_kernel
    (3, 38, 3, 64, 7, 2, 3),         # the stem on an odd map (Wo = 19: the last pixel pair is half empty)
    (2, 64, 256, 1088, 1, 1, 0),     # the head's final 1x1: ragged Cout tiles, uneven K slices
    (4, 8, 512, 512, 3, 1, 1),       # layer4 conv2: dgrad split over K
])
def test_conv2d_backward_vs_torch_autograd(pkg, B, H, Cin, Cout, k, stride, pad):
    """dgrad and wgrad of the NHWC convolution against torch autograd on the CPU (fp64)."""
    g = torch.Generator().manual_seed(B + H + Cin + Cout + k + stride)
    x = torch.randn(B, H, H, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    y = F.conv2d(xr, wr, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dy_nhwc = dy.permute(0, 2, 3, 1).float().contiguous()
    wo = pkg.conv.to_ohwi(w).to(DEV)
    dw = pkg.conv.conv2d_nhwc_wgrad(x.to(DEV), dy_nhwc.to(DEV), k, stride, pad)
    want_dw = wr.grad.permute(0, 2, 3, 1)
    assert dw.shape == want_dw.shape
    assert float((dw.cpu().double() - want_dw).abs().max()) < 2e-5 * float(want_dw.abs().max())
    if k == 7:
        return
    dx = pkg.conv.conv2d_nhwc_dgrad(dy_nhwc.to(DEV), wo, (H, H), stride, pad)
    want_dx = xr.grad.permute(0, 2, 3, 1)
    assert dx.shape == want_dx.shape
    assert float((dx.cpu().double() - want_dx).abs().max()) < 2e-5 * float(want_dx.abs().max())


@pytest.mark.parametrize("rows,C,relu", [(2 * 16 * 16, 256, True), (3 * 7 * 5, 64, False), (4096, 2048, True)])
def test_batchnorm2d_train_fwd_bwd_vs_torch(pkg, rows, C, relu):
    """Training-mode BatchNorm2d (+ ReLU) over an NHWC map = BatchNorm over the rows of [rows][C]: forward, running
    statistics and all three gradients against torch's batch_norm on the CPU (fp64)."""
    g = torch.Generator().manual_seed(rows + C)
    z = torch.randn(rows, C, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref = torch.nn.BatchNorm2d(C).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    zr = z.double().requires_grad_(True)
    yr = ref.train()(zr.t().reshape(1, C, rows, 1))
    yr = torch.relu(yr) if relu else yr
    dy = torch.randn(rows, C, generator=g)
    yr.backward(dy.double().t().reshape(1, C, rows, 1))
    bnd = bn.to(DEV).train()
    zd = z.to(DEV).requires_grad_(True)
    y = pkg.conv.batchnorm_relu_train(zd.reshape(1, rows, 1, C), bnd, relu)
    y.backward(dy.to(DEV).reshape(1, rows, 1, C))
    want = yr.reshape(C, rows).t()
    assert float((y.detach().reshape(rows, C).cpu().double() - want.detach()).abs().max()) < 2e-5
    assert float((bnd.running_mean.cpu().double() - ref.running_mean).abs().max()) < 1e-6
    assert float((bnd.running_var.cpu().double() - ref.running_var).abs().max()) < 1e-5
    assert int(bnd.num_batches_tracked) == 1
    gz = zr.grad
    assert float((zd.grad.cpu().double() - gz).abs().max()) < 2e-5 * max(1.0, float(gz.abs().max()))
    for ours, theirs in ((bnd.weight.grad, ref.weight.grad), (bnd.bias.grad, ref.bias.grad)):
        assert float((ours.cpu().double() - theirs).abs().max()) < 2e-5 * max(1.0, float(theirs.abs().max()))


def test_bottleneck_train_step_from_the_building_blocks_vs_torch(pkg):
    """One Bottleneck (Resnet.py:51-95) in TRAINING mode assembled from the library's differentiable pieces --
    conv2d_nhwc_autograd, batchnorm_relu_train, add_relu -- against the same stock module under torch autograd on the
    CPU: output and every parameter / input gradient."""
    torch.manual_seed(3)
    blk = pkg.backbone.Bottleneck(256, 64, stride=1).train()
    x = torch.randn(2, 16, 16, 256)
    dy = torch.randn(2, 16, 16, 256)
    # reference: stock PyTorch, NCHW, CPU
    import copy
    ref = copy.deepcopy(blk).double()
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    o = torch.relu(ref.bn1(ref.conv1(xr)))
    o = torch.relu(ref.bn2(ref.conv2(o)))
    out_ref = torch.relu(ref.bn3(ref.conv3(o)) + xr)
    out_ref.backward(dy.permute(0, 3, 1, 2).double())
    # ours
    b = blk.to(DEV)
    ws = {n: pkg.conv.to_ohwi(getattr(b, n).weight.detach()).requires_grad_(True) for n in ("conv1", "conv2", "conv3")}
    xd = x.to(DEV).requires_grad_(True)
    o = pkg.conv.batchnorm_relu_train(pkg.conv.conv2d_nhwc_autograd(xd, ws["conv1"], 1, 0), b.bn1, True)
    o = pkg.conv.batchnorm_relu_train(pkg.conv.conv2d_nhwc_autograd(o, ws["conv2"], 1, 1), b.bn2, True)
    o = pkg.conv.batchnorm_relu_train(pkg.conv.conv2d_nhwc_autograd(o, ws["conv3"], 1, 0), b.bn3, False)
    out = pkg.conv.add_relu(o, xd)
    out.backward(dy.to(DEV))
    close = lambda a, r, tol: float((a.cpu().double() - r).abs().max()) < tol * max(1.0, float(r.abs().max()))
    assert close(out.detach(), out_ref.permute(0, 2, 3, 1), 5e-5)
    assert close(xd.grad, xr.grad.permute(0, 2, 3, 1), 2e-4)
    for n in ("conv1", "conv2", "conv3"):
        assert close(ws[n].grad, getattr(ref, n).weight.grad.permute(0, 2, 3, 1), 2e-4), n
    for n in ("bn1", "bn2", "bn3"):
        assert close(getattr(b, n).weight.grad, getattr(ref, n).weight.grad, 2e-4), n
        assert close(getattr(b, n).bias.grad, getattr(ref, n).bias.grad, 2e-4), n
        assert close(getattr(b, n).running_var, getattr(ref, n).running_var, 1e-5), n


@pytest.mark.parametrize("shape", [(2, 12, 10, 8), (1, 11, 9, 4)])
def test_maxpool_backward_vs_torch(pkg, shape):
    """Both backward entry points (from the recorded tap index = the autograd path, and recomputed from x) against
    torch, bit for bit, ties included."""
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape)
    x[0, 2:4, 2:4] = 1.5                                       # ties inside windows: first maximum wins, as in torch
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(y.shape)
    y.backward(dy)
    want = xr.grad.permute(0, 2, 3, 1)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    yd = pkg.conv.maxpool3x3s2_nhwc_autograd(xd)
    assert torch.equal(yd.detach().cpu(), y.detach().permute(0, 2, 3, 1))
    yd.backward(dyd)
    assert torch.equal(xd.grad.cpu(), want)
    B, H, W, C = shape
    dx = torch.empty(B, H, W, C, device=DEV)
    L = pkg.lib()
    rc = L.pl_maxpool3x3s2_nhwc_bwd(xd.data_ptr(), dyd.data_ptr(), B, H, W, C, dx.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(dx.cpu(), want)


@pytest.mark.parametrize("size,dtype", [(64, "bf16x6"), (256, "bf16x6"), (64, "f16x3"), (256, "f16x3")])
def test_model3d_train_step_vs_torch_autograd(pkg, size, dtype):
    """Model_3D in TRAINING mode end to end on the library's differentiable pieces (53 convolutions with dgrad and
    wgrad, 56 BatchNorms on batch statistics, max-pool, three transposed convolutions, biased 1x1, soft-argmax)
    against the same stock modules under torch autograd on the CPU in fp64: loss and parameter gradients.
    size 256 = BASELINE configs[3]'s real frame size (phase4_joined/train.py:69-89 on 256 x 256 frames: 64 x 64 x 64
    heat-map volumes, 8 x 8 layer4 maps); 64 is the quick case.  dtype "f16x3": the Bottlenecks' 1x1 convolutions (forward,
    data and weight gradients) on the planes GEMM with operand planes written by the BatchNorm / join kernels
    (conv.py, bottom) -- at 256 every block qualifies, at 64 layer3 / layer4 fall back block by block."""
    import copy
    torch.manual_seed(7)
    m = pkg.Model_3D(compute_dtype=dtype).train()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 51))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-3)
    frames = pkg.synth.seeded_frames(2, 52, size=size)
    target = torch.randn(2, 51)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    def torch_step(dtype):
        ref = copy.deepcopy(m).to(dtype)
        r = ref.preact
        x = frames.permute(0, 3, 1, 2).to(dtype)
        x = F.max_pool2d(F.relu(r.bn1(r.conv1(x))), 3, 2, 1)
        for li in (1, 2, 3, 4):
            for blk in getattr(r, f"layer{li}"):
                idn = x if blk.downsample is None else blk.downsample(x)
                o = F.relu(blk.bn1(blk.conv1(x)))
                o = F.relu(blk.bn2(blk.conv2(o)))
                x = F.relu(blk.bn3(blk.conv3(o)) + idn)
        out = ref.final_layer(ref.deconv_layers(x))
        B, _, H, W = out.shape
        hm = torch.softmax(out.reshape(B, 17, -1), 2).reshape(B, 17, 64, H, W)
        cx = (hm.sum((2, 3)) * torch.arange(W, dtype=dtype)).sum(2, keepdim=True)
        cy = (hm.sum((2, 4)) * torch.arange(H, dtype=dtype)).sum(2, keepdim=True)
        cz = (hm.sum((3, 4)) * torch.arange(64, dtype=dtype)).sum(2, keepdim=True)
        pred = torch.cat(((cx / W - .5) * 2, (cy / H - .5) * 2, (cz / 64 - .5) * 2), 2).reshape(B, 51)
        loss = ((pred - target.to(dtype)) ** 2).mean()
        loss.backward()
        return ref, loss

    ref, loss_ref = torch_step(torch.float64)
    ref32, _ = torch_step(torch.float32)        # the noise floor: stock fp32 against its own fp64 (batch statistics over
    r = ref.preact                              # as few as 8 rows amplify round-off through 50 layers)
    md = m.to(DEV)
    pred = md(frames.to(DEV))
    loss = ((pred - target.to(DEV)) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-4 * max(1.0, abs(float(loss_ref.detach())))
    # Round-off is chaotic here (a ReLU at a small map flips and a tiny gradient changes discretely), so every tensor is
    # judged against what stock fp32 does against its own fp64 -- but per tensor and in the relative L2 norm: an indexing
    # bug in ONE layer's weight gradient gives a relative error of order 1 in that tensor, wherever its size puts it
    # among the others, and cannot hide under a bound scaled by the tensor's largest element.
    checked, e2, f2, n2, worst = 0, 0.0, 0.0, 0.0, (0.0, "")
    for (k, p), (_, q), (_, q32) in zip(md.named_parameters(), ref.named_parameters(), ref32.named_parameters()):
        if p.grad is None:
            continue
        gr = q.grad
        norm = float(gr.norm())
        if float(gr.abs().max()) < 1e-12:
            continue
        d = p.grad.cpu().double() - gr
        rel = float(d.norm()) / norm
        floor_rel = float((q32.grad.double() - gr).norm()) / norm
        assert rel < min(max(25 * floor_rel, 5e-3), 0.5), (k, rel, floor_rel)
        worst = max(worst, (rel, k))
        scale = float(gr.abs().max())
        e2 += float((d / scale).pow(2).mean()); f2 += float(((q32.grad.double() - gr) / scale).pow(2).mean()); n2 += 1
        checked += 1
    assert (e2 / n2) ** 0.5 < 4 * (f2 / n2) ** 0.5 + 1e-4, ((e2 / n2) ** 0.5, (f2 / n2) ** 0.5)
    assert checked > 150
    assert float((md.preact.bn1.running_mean.cpu().double() - r.bn1.running_mean).abs().max()) < 1e-5


def test_model3d_train_step_at_config4_size_properties(pkg):
    """BASELINE configs[3] at its full size -- 256 x 256 frames, batch 256 -- through size-independent properties: the
    training step (forward, loss, backward over 161 parameter tensors) is bitwise repeatable, every gradient is finite
    and non-trivial, BatchNorm buffers move; the eval forward treats rows independently."""
    torch.manual_seed(3)
    m = pkg.Model_3D()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 71))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-3)
    m = m.to(DEV).train()
    B = 256
    frames = pkg.synth.seeded_frames(B, 72, size=256).to(DEV)
    target = torch.randn(B, 51, device=DEV) * 0.3
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    runs = []
    for _ in range(2):
        m.load_state_dict(sd0)
        m.zero_grad(set_to_none=True)
        loss = ((m(frames) - target) ** 2).mean()
        loss.backward()
        runs.append((loss.detach().clone(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0])
    n_nonzero = 0
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
        n_nonzero += int(float(a.abs().max()) > 0)
    assert n_nonzero >= len(runs[0][1]) - 2
    assert not torch.equal(m.preact.bn1.running_mean, sd0["preact.bn1.running_mean"])
    m.eval()
    with torch.no_grad():
        y_all = m(frames[:64])
        y_sub = m(frames[8:24])
    assert torch.allclose(y_all[8:24], y_sub, rtol=0, atol=2e-5)      # rows independent (split-K choices may differ by batch)


def test_phase5_cycle_step_runs_and_couples_the_networks(pkg):
    """train_5 copy.py:147-236 assembled from the library's pieces: Model_2D + Model_3D (training mode) on the same
    NHWC frames, the lifter called on the predicted and on the true 2-D pose, the projector LinearModel(51, 34, 64),
    TriangleLoss, one backward, four optimizers.  Checks the plumbing every component test cannot: the lifter's input
    gradient reaches the 2-D network, every model's parameters move, the loss goes down on a repeated batch."""
    torch.manual_seed(0)
    m2 = pkg.Model_2D().train()
    m3 = pkg.Model_3D().train()
    for m, seed in ((m2, 61), (m3, 62)):
        m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), seed))
        with torch.no_grad():
            m.final_layer.weight.mul_(1e-3)
    m2, m3 = m2.to(DEV), m3.to(DEV)
    lift = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.0).to(DEV).train()
    proj = pkg.LinearModel(51, 34, linear_size=64, p_dropout=0.0).to(DEV).train()
    opts = [torch.optim.Adam(m2.parameters(), lr=1e-4), torch.optim.Adam(m3.parameters(), lr=1e-4),
            pkg.FlatAdamW(lift, lr=1e-3), pkg.FlatAdamW(proj, lr=1e-3)]
    frames = pkg.synth.seeded_frames(4, 63, size=64).to(DEV)
    y1, y2 = pkg.synth.synthetic_batch(4, 64, DEV)
    crit = pkg.TriangleLoss(Project=True, era="lifter")
    before = [m2.final_layer.bias.detach().clone(), m3.final_layer.bias.detach().clone(), lift.flat_params.clone(),
              proj.flat_params.clone()]
    losses = []
    for _ in range(4):
        loss, y1_hat, y2_hat = pkg.cycle_step(m2, m3, lift, opts, frames, y1, y2, crit, model_proj=proj)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert y1_hat.shape == (4, 17, 2) and y2_hat.shape == (4, 17, 3)
    after = [m2.final_layer.bias.detach(), m3.final_layer.bias.detach(), lift.flat_params, proj.flat_params]
    assert all(not torch.equal(a, b) for a, b in zip(before, after))
    # the lifter's input gradient is the ONLY path from the lift terms into the 2-D network
    for opt in opts:
        opt.zero_grad()
    y1_hat = m2.predict_nhwc(frames).reshape(4, 17, 2)
    pkg.l1_loss(lift(y1_hat).reshape(4, 17, 3), y2).backward()
    assert float(m2.final_layer.bias.grad.abs().max()) > 0
    assert len(crit.term_means()) == 6


def test_bf16_arithmetic_mode_of_the_conv_path(pkg):
    """arith="bf16" (operands rounded to bf16 while staged, one product): bf16-sized error on a convolution, and
    Model_3D(compute_dtype="bf16") stays within a fraction of a voxel of the fp32-grade model."""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 32, 32, 128, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) / np.sqrt(128 * 9)
    want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    got = pkg.conv.conv2d_nhwc(x.to(DEV), pkg.conv.to_ohwi(w).to(DEV), 1, 1, arith="bf16").cpu().double()
    err = float((got - want).abs().max())
    assert 1e-4 < err < 5e-2, err                                # really bf16, and only bf16
    m = pkg.Model_3D().eval()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-4)
    m = m.to(DEV)
    frames = pkg.synth.seeded_frames(2, 32).to(DEV)
    ref = m(frames)
    m.compute_dtype = m.preact.compute_dtype = "bf16"
    fast = m(frames)
    assert float((fast - ref).abs().max()) < 2e-2                # coordinates in (-1, 1): 0.6 voxel of 64
    with pytest.raises(KeyError):
        pkg.conv.conv2d_nhwc(x.to(DEV), pkg.conv.to_ohwi(w).to(DEV), 1, 1, arith="fp16")


def test_bf16_arithmetic_mode_trains(pkg):
    """compute_dtype="bf16" in TRAINING mode: forward, dgrad and wgrad of every convolution round their operands to
    bf16 while staging (fp32 accumulate and storage; the stem and BatchNorm stay fp32).  A single weight gradient is
    really bf16 and only bf16; the model's loss and gradients stay close to the fp32-grade mode's."""
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 16, 16, 128, generator=g)
    dy = torch.randn(2, 16, 16, 256, generator=g)
    ref = pkg.conv.conv2d_nhwc_wgrad(x.to(DEV), dy.to(DEV), 3, 1, 1)
    fast = pkg.conv.conv2d_nhwc_wgrad(x.to(DEV), dy.to(DEV), 3, 1, 1, arith="bf16")
    rel = float((fast - ref).abs().max() / ref.abs().max())
    assert 1e-5 < rel < 2e-2, rel
    torch.manual_seed(3)
    m = pkg.Model_3D().train()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 61))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-3)
    m = m.to(DEV)
    frames = pkg.synth.seeded_frames(4, 62, size=64).to(DEV)
    target = torch.randn(4, 51, device=DEV)

    def grads(mode):
        m.compute_dtype = m.preact.compute_dtype = mode
        m.zero_grad(set_to_none=True)
        loss = ((m(frames) - target) ** 2).mean()
        loss.backward()
        return float(loss), torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    l0, g0 = grads("bf16x6")
    l1, g1 = grads("bf16")
    assert abs(l1 - l0) < 2e-2 * abs(l0)
    cos = float(torch.dot(g0, g1) / (g0.norm() * g1.norm()))
    assert cos > 0.98, cos                                     # same descent direction; not bit-close by design
    assert float((g0 - g1).norm()) > 0                          # and really a different arithmetic


# ---------------------------------------------------------------------------- convolutions on the planes GEMM
def _planes(pkg, t, scale=1.0):
    from importlib import import_module
    cv = import_module("3d_poseestimation_amd.conv")
    return cv._planes_of(t, scale)


@pytest.mark.parametrize("dtype", ["f16x3", "bf16p"])
def test_graphed_module_step_replays_the_eager_step_bit_for_bit(pkg, dtype):
    """train.GraphedModuleStep: the phase4 step (train.py:69-89: zero_grad, Model_3D forward, MSE, backward, Adam) captured
    once as a hipGraph.  Three replays on three different batches == three eager steps from the same start: losses,
    parameters, BatchNorm buffers and Adam state identical bit for bit; construction (its warm-up steps run for real) leaves
    the model and a fresh optimizer exactly as they were."""
    frames = [pkg.synth.seeded_frames(2, 40 + i, 64).to(DEV) for i in range(3)]
    targets = [torch.randn(2, 51, generator=torch.Generator().manual_seed(50 + i)).to(DEV) for i in range(3)]

    def build():
        m = pkg.Model_3D(compute_dtype=dtype).train()
        m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 31))
        with torch.no_grad():
            m.final_layer.weight.mul_(1e-3)
        m = m.to(DEV)
        return m, torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)

    m0, o0 = build()
    eager = []
    for x, t in zip(frames, targets):
        o0.zero_grad(set_to_none=True)
        loss = F.mse_loss(m0(x), t)
        loss.backward()
        o0.step()
        eager.append(loss.detach().clone())
    m1, o1 = build()
    before = {k: v.detach().clone() for k, v in m1.state_dict().items()}
    step = pkg.GraphedModuleStep(m1, o1, F.mse_loss, frames[2], targets[2])
    torch.cuda.synchronize()
    for k, v in m1.state_dict().items():
        assert torch.equal(v, before[k]), f"construction changed {k}"
    for p in m1.parameters():
        st = o1.state[p]
        assert float(st["step"]) == 0.0 and not bool(st["exp_avg"].any()) and not bool(st["exp_avg_sq"].any())
    for i, (x, t) in enumerate(zip(frames, targets)):
        loss = step(x, t)
        assert torch.equal(loss, eager[i]), (i, float(loss), float(eager[i]))
    assert step.replays == 3
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k
    for p0, p1 in zip(m0.parameters(), m1.parameters()):
        for key in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(o0.state[p0][key], o1.state[p1][key]), key
    with pytest.raises(pkg.PoseliftError, match="capturable"):
        pkg.GraphedModuleStep(m1, torch.optim.Adam(m1.parameters(), lr=1e-3), F.mse_loss, frames[0], targets[0])


@pytest.mark.parametrize("mode", ["f16x3", "bf16p"])
def test_planes_of_strided_views_equal_planes_of_their_copies(pkg, mode):
    """pl_planes_split_strided (conv._planes_of on a non-contiguous view, conv._planes_of_flipped_t): a convolution weight's
    OHWI layout, its transpose and the flipped / transposed kernel of the stride-1 data gradient go from the OIHW parameter to
    operand planes in one launch -- bit-identical to the layout copy followed by pl_planes_split."""
    from importlib import import_module
    cv = import_module("3d_poseestimation_amd.conv")
    lib_mode = {"f16x3": pkg._lib.PL_F16X3, "bf16p": pkg._lib.PL_BF16}[mode]
    w = (torch.randn(72, 32, 3, 3, generator=torch.Generator().manual_seed(5)) * 0.07).to(DEV)        # OIHW
    used = 1 if mode == "f16x3" else 2                                       # (a bf16 carrier uses the first half of its bytes)
    eq = lambda a, b: torch.equal(a.reshape(-1).view(torch.int32)[:a.numel() // used],                # noqa: E731
                                  b.reshape(-1).view(torch.int32)[:b.numel() // used])
    ohwi = w.permute(0, 2, 3, 1)
    assert not ohwi.is_contiguous()
    got = cv._planes_of(ohwi, 16.0, lib_mode)
    want = cv._planes_of(ohwi.contiguous(), 16.0, lib_mode)
    assert got.shape == want.shape and eq(got, want)
    got = cv._planes_of_flipped_t(ohwi, 16.0, lib_mode)                      # from the view ...
    want = cv._planes_of(ohwi.contiguous().flip(1, 2).permute(3, 1, 2, 0).contiguous(), 16.0, lib_mode)
    assert tuple(got.shape) == (32, 3, 3, 72) and eq(got, want)
    assert eq(cv._planes_of_flipped_t(ohwi.contiguous(), 16.0, lib_mode), want)                       # ... and from a copy
    m = w.reshape(72, 288)
    assert eq(cv._planes_of(m.t(), 16.0, lib_mode), cv._planes_of(m.t().contiguous(), 16.0, lib_mode))
    v = torch.randn(8, 40, device=DEV)[:, ::2]                               # 1-D-ish strided rows, innermost stride 2
    assert eq(cv._planes_of(v, 1.0, lib_mode), cv._planes_of(v.contiguous(), 1.0, lib_mode))


@pytest.mark.parametrize("rows,C,two", [(1000, 320, True), (4096, 256, False), (777, 1024, True)])
def test_bn_join_bwd_is_the_masked_sum_plus_batchnorm_backward_bit_for_bit(pkg, rows, C, two):
    """pl_bn_join_bwd (bn3 + residual join backward, the masked sum written by the pass that takes the BatchNorm-backward column
    sums) against the two calls it replaces, pl_mask_add_by_bits + pl_bn_train_bwd_ex: dx, the dz planes, their scale, dgamma
    and dbeta identical bit for bit."""
    L = pkg.lib()
    s = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(rows + C)
    z = torch.randn(rows, C, generator=gen).to(DEV)
    ident = torch.randn(rows, C, generator=gen).to(DEV)
    g = torch.randn(rows, C, generator=gen).to(DEV)
    g2 = torch.randn(rows, C, generator=gen).to(DEV) if two else None
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(DEV), torch.randn(C, generator=gen).to(DEV)
    rm, rv, nb = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    x, xp = torch.empty_like(z), torch.empty_like(z)
    bits = torch.empty(rows, 4 * ((C + 255) // 256), dtype=torch.int64, device=DEV)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    scratch = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=DEV)
    rc = L.pl_bn_train_fwd_ex(z.data_ptr(), rows, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                              nb.data_ptr(), 1, x.data_ptr(), bits.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scratch.data_ptr(),
                              xp.data_ptr(), 3, None, ident.data_ptr(), s)
    assert rc == 0, L.pl_last_error()
    out = []
    for fused in (False, True):
        dx = torch.full_like(z, float("nan"))
        dz = torch.full_like(z, float("nan"))                     # carrier of the dz planes
        dgam, dbet, dzs = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2, device=DEV)
        sc = torch.empty(L.pl_bn_train_scratch_bytes(rows, C), dtype=torch.uint8, device=DEV)
        g2p = g2.data_ptr() if g2 is not None else None
        if fused:
            rc = L.pl_bn_join_bwd(g.data_ptr(), g2p, bits.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                  rows, C, dx.data_ptr(), None, dgam.data_ptr(), dbet.data_ptr(), sc.data_ptr(), dz.data_ptr(), 3,
                                  dzs.data_ptr(), s)
        else:
            rc = L.pl_mask_add_by_bits(g.data_ptr(), g2p, bits.data_ptr(), rows, C, dx.data_ptr(), s)
            assert rc == 0, L.pl_last_error()
            rc = L.pl_bn_train_bwd_ex(dx.data_ptr(), bits.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                      rows, C, None, dgam.data_ptr(), dbet.data_ptr(), sc.data_ptr(), dz.data_ptr(), 3, dzs.data_ptr(), s)
        assert rc == 0, L.pl_last_error()
        torch.cuda.synchronize()
        out.append((dx, dz.view(torch.int32), dgam, dbet, dzs))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    mask = x > 0
    want = torch.where(mask, g + (g2 if g2 is not None else 0), torch.zeros_like(g))
    assert torch.equal(out[1][0], want)


@pytest.mark.parametrize("mode", ["f16x3", "bf16p"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,K,pad,route", [
    (2, 16, 24, 32, 128, 3, 1, "transposed 4x4"),     # layer2.0 / 3.0 / 4.0 conv2 in small
    (2, 8, 8, 128, 96, 1, 0, "low-res GEMM"),          # the stride-2 downsample
    (2, 7, 7, 64, 64, 1, 0, "zero-spread map"),        # odd maps (2 Ho != H): the general route
    (2, 7, 7, 32, 64, 3, 1, "zero-spread map"),
])
def test_conv_planes_stride2_autograd_vs_torch_fp64(pkg, mode, B, H, W, Cin, Cout, K, pad, route):
    """Backward of a stride-2 planes convolution through its autograd node (conv._ConvKxKPlanesFn): dz arrives as a carrier of
    planes, the data gradient takes one of three routes -- ConvTranspose2d(4, 2, 1) gathers with the 3x3 filter in a zero
    4x4 one, a GEMM at the output resolution spread over the even pixels (1x1), or the stride-1 gather over a zero-spread
    map (everything else) -- and all three, with the weight gradient, match torch's conv2d in fp64."""
    from importlib import import_module
    cv = import_module("3d_poseestimation_amd.conv")
    lib_mode = {"f16x3": pkg._lib.PL_F16X3, "bf16p": pkg._lib.PL_BF16}[mode]
    g = torch.Generator().manual_seed(B * 7 + H * 31 + Cin + Cout + K)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) * 0.05                        # the nn.Conv2d parameter
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    dz = torch.randn(B, Ho, Wo, Cout, generator=g)
    link = cv.PlaneLink(lib_mode)
    link.dz_scale = torch.tensor([1.0, 1.0, 1.0], device=DEV)                    # {S, 1/S}: dz planes unscaled here
    xp = cv._planes_of(x.to(DEV), cv.ACT_PLANE_SCALE, lib_mode).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    z = cv.conv_planes(xp, wd, 2, pad, link)
    z.backward(cv._planes_of(dz.to(DEV), 1.0, lib_mode))
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    ref = F.conv2d(x64, w64, stride=2, padding=pad)
    ref.backward(dz.double().permute(0, 3, 1, 2))
    # bf16 storage: operands carry 8 bits; fp32-grade: the bound of an fp32 evaluation
    tol = 2e-2 if mode == "bf16p" else 3e-6
    nterm = (K * K * Cout) ** 0.5
    assert float((z.detach().cpu().double() - ref.detach().permute(0, 2, 3, 1)).abs().max()) <= tol * float(ref.abs().max()) * 4
    want = x64.grad.permute(0, 2, 3, 1)
    got = xp.grad.cpu().double()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= tol * float(want.abs().max()) * nterm, route
    wantw = w64.grad
    assert float((wd.grad.cpu().double() - wantw).abs().max()) <= tol * float(wantw.abs().max()) * (B * Ho * Wo) ** 0.5, route


@pytest.mark.parametrize("mode", ["f16x3", "bf16p"])
@pytest.mark.parametrize("B,H,W", [(4, 64, 64), (1, 32, 64), (2, 128, 64)])
def test_stem_on_the_planes_gemm_vs_torch_fp64(pkg, mode, B, H, W):
    """The 7x7 / stride 2 / pad 3 stem on 3 input channels (Resnet.py:112-113) as a planes GEMM on the frame's pixel-pair view
    (conv.stem_planes: 7 x 4 taps of 8 channels, stride (2, 1), padding (3, 2), zero weights where the pair window overhangs the
    seven real taps): output -- every border pixel included --, the epilogue's BatchNorm statistics and the weight gradient
    (gathered TN GEMM, its 224 columns folded back to the 147 real ones) against torch's conv2d in fp64."""
    from importlib import import_module
    cv = import_module("3d_poseestimation_amd.conv")
    lib_mode = {"f16x3": pkg._lib.PL_F16X3, "bf16p": pkg._lib.PL_BF16}[mode]
    assert cv.stem_planes_supported(B, H, W, 3, 64, 7, 7, 2, 3)
    g = torch.Generator().manual_seed(B * 7 + H + W)
    x = torch.rand(B, H, W, 3, generator=g)                                       # frames in [0, 1)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    dz = torch.randn(B, H // 2, W // 2, 64, generator=g)
    link = cv.PlaneLink(lib_mode)
    link.dz_scale = torch.tensor([1.0, 1.0, 1.0], device=DEV)
    wd = w.to(DEV).requires_grad_(True)
    z = cv.stem_planes(cv.stem_input_planes(x.to(DEV), lib_mode), wd, link)
    stat = link.stat
    z.backward(cv._planes_of(dz.to(DEV), 1.0, lib_mode))
    x64 = x.double().permute(0, 3, 1, 2)
    w64 = w.double().requires_grad_(True)
    ref = F.conv2d(x64, w64, stride=2, padding=3)
    ref.backward(dz.double().permute(0, 3, 1, 2))
    refn = ref.detach().permute(0, 2, 3, 1)
    tol = 2e-2 if mode == "bf16p" else 3e-6
    assert z.shape == refn.shape
    assert float((z.detach().cpu().double() - refn).abs().max()) <= tol * float(refn.abs().max()) * 4
    # the epilogue statistics: sums over 64-row groups of the output
    rows = refn.reshape(-1, 64)
    sums = stat[0].cpu().double().sum(0)
    assert float((sums - rows.sum(0)).abs().max()) <= (tol * 40) * float(rows.abs().sum(0).max())
    wantw = w64.grad
    assert float((wd.grad.cpu().double() - wantw).abs().max()) <= tol * float(wantw.abs().max()) * (B * H * W / 4) ** 0.5


@pytest.mark.parametrize("B,H,W,Cin,Cout,K,stride,pad", [
    (2, 16, 16, 64, 64, 3, 1, 1),        # layer1's conv2 in small: a 64-wide tile hanging over N
    (2, 16, 24, 32, 128, 3, 2, 1),       # stride 2, one 32-k tile per tap
    (2, 8, 8, 128, 72, 1, 2, 0),         # the stride-2 downsample: a row gather
    (3, 9, 7, 64, 96, 3, 1, 1),          # odd map, 189 output pixels: ragged M (forward / data gradient only)
    (1, 32, 32, 96, 256, 3, 1, 1),       # Cin not a power of two: taps of three 32-k tiles
])
def test_conv_planes_fwd_dgrad_wgrad_vs_torch_fp64(pkg, B, H, W, Cin, Cout, K, stride, pad):
    """Implicit-GEMM convolution on the planes GEMM (LDS-DMA gather by the loader waves, zero-filling out-of-range lanes
    for the padding): forward, data gradient (the same kernel on dz with the flipped kernel) and weight gradient (TN with
    the gathered B operand) against torch's conv2d in fp64, inside the bound of an fp32 evaluation."""
    L = pkg.lib()
    g = torch.Generator().manual_seed(B * 131 + H * 17 + Cin + Cout + K)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, K, K, Cin, generator=g) * 0.05
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    dz = torch.randn(B, Ho, Wo, Cout, generator=g)
    xd, wd, dzd = x.to(DEV), w.to(DEV), dz.to(DEV)
    xp, wp, dzp = _planes(pkg, xd, 1.0), _planes(pkg, wd, 16.0), _planes(pkg, dzd, 1.0)
    s = torch.cuda.current_stream().cuda_stream
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device=DEV)
    G = L.pl_gemm_stat_groups(B * Ho * Wo)
    stat = torch.full((2, G, Cout), float("nan"), device=DEV)        # the epilogue's BatchNorm partial statistics of y
    rc = L.pl_conv2d_planes_fwd(3, xp.data_ptr(), x.numel(), B, H, W, Cin, wp.data_ptr(), w.numel(), Cout,
                                K, K, stride, pad, y.data_ptr(), 1.0 / 16.0, None, stat.data_ptr(), s)
    assert rc == 0, L.pl_last_error()
    y2 = y.reshape(-1, Cout).double().cpu()
    for g in range(G):
        blk = y2[64 * g:64 * (g + 1)]
        want_s = blk.sum(0) if len(blk) else torch.zeros(Cout, dtype=torch.float64)
        want_q = ((blk - blk.mean(0)) ** 2).sum(0) if len(blk) else torch.zeros(Cout, dtype=torch.float64)
        assert float((stat[0, g].cpu().double() - want_s).abs().max()) <= 1e-4 * max(1.0, float(want_s.abs().max()))
        assert float((stat[1, g].cpu().double() - want_q).abs().max()) <= 1e-4 * max(1.0, float(want_q.abs().max()))
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(x64, w64, stride=stride, padding=pad)
    ref.backward(dz.double().permute(0, 3, 1, 2))
    mag = F.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs().permute(0, 3, 1, 2), stride=stride, padding=pad)
    err = (y.cpu().double() - ref.detach().permute(0, 2, 3, 1)).abs()
    assert bool((err <= 2e-6 * mag.permute(0, 2, 3, 1) + 1e-7).all()), float(err.max())
    if stride == 1:
        wf = w.flip(1, 2).permute(3, 1, 2, 0).contiguous().to(DEV)            # [Cin][KH][KW][Cout]
        wfp = _planes(pkg, wf, 16.0)
        dx = torch.full((B, H, W, Cin), float("nan"), device=DEV)
        rc = L.pl_conv2d_planes_fwd(3, dzp.data_ptr(), dz.numel(), B, Ho, Wo, Cout, wfp.data_ptr(), wf.numel(),
                                    Cin, K, K, 1, K - 1 - pad, dx.data_ptr(), 1.0 / 16.0, None, None, s)
        assert rc == 0, L.pl_last_error()
        want = x64.grad.permute(0, 2, 3, 1)
        assert float((dx.cpu().double() - want).abs().max()) <= 3e-6 * float(want.abs().max()) * (K * K * Cout) ** 0.5
    if (B * Ho * Wo) % 32 == 0:
        n = Cout * K * K * Cin
        splits = L.pl_gemm_planes_splits(Cout, K * K * Cin, B * Ho * Wo)
        slabs = torch.empty(splits * n, device=DEV) if splits > 1 else None
        dw = torch.full((Cout, K, K, Cin), float("nan"), device=DEV)
        rc = L.pl_conv2d_planes_wgrad(3, dzp.data_ptr(), dz.numel(), xp.data_ptr(), x.numel(), B, H, W, Cin, Cout,
                                      K, K, stride, pad, dw.data_ptr(), 1.0, None, slabs.data_ptr() if slabs is not None else None, s)
        assert rc == 0, L.pl_last_error()
        want = w64.grad.permute(0, 2, 3, 1)
        assert float((dw.cpu().double() - want).abs().max()) <= 3e-6 * float(want.abs().max()) * (B * Ho * Wo) ** 0.5


def test_model3d_bf16_storage_train_step_tracks_the_fp32_grade_one(pkg):
    """compute_dtype "bf16p": the Bottleneck convolutions on ONE bf16 operand plane (bf16 storage of activations, dz and the
    weight shadow; fp32 accumulation, fp32 BatchNorm statistics) and bf16 arithmetic elsewhere -- the throughput mode, never
    the parity-gated one: its loss and gradients must track the fp32-grade step at bf16 accuracy."""
    import copy
    torch.manual_seed(11)
    m = pkg.Model_3D().train()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 61))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-3)
    frames = pkg.synth.seeded_frames(4, 62, size=256).to(DEV)
    target = torch.randn(4, 51, device=DEV)
    out = {}
    for dt in ("bf16x6", "bf16p"):
        md = copy.deepcopy(m).to(DEV)
        md.compute_dtype = md.preact.compute_dtype = dt
        loss = ((md(frames) - target) ** 2).mean()
        loss.backward()
        out[dt] = (float(loss.detach()), torch.cat([p.grad.reshape(-1) for p in md.parameters() if p.grad is not None]).double())
    (l0, g0), (l1, g1) = out["bf16x6"], out["bf16p"]
    assert abs(l1 - l0) < 2e-2 * abs(l0), (l0, l1)
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    assert cos > 0.98 and bool(torch.isfinite(g1).all()), cos


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 8, 8, 64, 128), (1, 6, 10, 96, 64), (3, 4, 4, 256, 256)])
def test_deconv_planes_fwd_dgrad_wgrad_vs_torch_fp64(pkg, B, H, W, Cin, Cout):
    """ConvTranspose2d(4, 2, 1) on the planes GEMM: forward as four 2x2-tap gathers stored in place per output parity, data
    gradient as the gathered 4x4 stride-2 convolution, weight gradient with the roles of x and dy exchanged -- against
    torch's conv_transpose2d in fp64."""
    cv = importlib.import_module("3d_poseestimation_amd.conv")
    L = pkg.lib()
    g = torch.Generator().manual_seed(B + 7 * H + Cin + 3 * Cout)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.05            # torch's ConvTranspose2d layout
    dy = torch.randn(B, 2 * H, 2 * W, Cout, generator=g)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    s = torch.cuda.current_stream().cuda_stream
    xp, dyp = cv._planes_of(xd, 1.0), cv._planes_of(dyd, 1.0)
    wsub = cv._planes_of(cv.deconv_subkernels(wd), 16.0)
    y = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device=DEV)
    rc = L.pl_deconv4x4s2_planes_fwd(3, xp.data_ptr(), x.numel(), B, H, W, Cin, wsub.data_ptr(), wsub.numel(), Cout,
                                     y.data_ptr(), 1.0 / 16.0, None, s)
    assert rc == 0, L.pl_last_error()
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    ref = F.conv_transpose2d(x64, w64, stride=2, padding=1)
    ref.backward(dy.double().permute(0, 3, 1, 2))
    want = ref.detach().permute(0, 2, 3, 1)
    assert float((y.cpu().double() - want).abs().max()) <= 3e-6 * float(want.abs().max()) * (4 * Cin) ** 0.5
    # dx = C dy: the OHWI kernel [Cin][4][4][Cout], stride 2, pad 1
    wc = cv._planes_of(wd.permute(0, 2, 3, 1), 16.0)
    dx = torch.full((B, H, W, Cin), float("nan"), device=DEV)
    rc = L.pl_conv2d_planes_fwd(3, dyp.data_ptr(), dy.numel(), B, 2 * H, 2 * W, Cout, wc.data_ptr(), w.numel(), Cin, 4, 4, 2, 1,
                                dx.data_ptr(), 1.0 / 16.0, None, None, s)
    assert rc == 0, L.pl_last_error()
    wantx = x64.grad.permute(0, 2, 3, 1)
    assert float((dx.cpu().double() - wantx).abs().max()) <= 3e-6 * float(wantx.abs().max()) * (16 * Cout) ** 0.5
    if (B * H * W) % 32 == 0:
        n = Cin * 16 * Cout
        splits = L.pl_gemm_planes_splits(Cin, 16 * Cout, B * H * W)
        slabs = torch.empty(splits * n, device=DEV) if splits > 1 else None
        dwc = torch.full((Cin, 4, 4, Cout), float("nan"), device=DEV)
        rc = L.pl_conv2d_planes_wgrad(3, xp.data_ptr(), x.numel(), dyp.data_ptr(), dy.numel(), B, 2 * H, 2 * W, Cout, Cin, 4, 4,
                                      2, 1, dwc.data_ptr(), 1.0, None, slabs.data_ptr() if slabs is not None else None, s)
        assert rc == 0, L.pl_last_error()
        wantw = w64.grad.permute(0, 2, 3, 1)                         # [Cin][4][4][Cout]
        assert float((dwc.cpu().double() - wantw).abs().max()) <= 3e-6 * float(wantw.abs().max()) * (B * H * W) ** 0.5


def test_softargmax_bwd_planes_and_colsum_planes_match_the_fp32_forms(pkg):
    """The head's last link: dlogits written as fp16 operand planes (scaled by the bound-derived power of two) carry the
    fp32 gradient to 2^-21 of the bound, and their column sums are the bias gradient."""
    heads = importlib.import_module("3d_poseestimation_amd.heads")
    L = pkg.lib()
    B, J, H, W = 3, 17, 8, 8
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(B, H, W, J * 64, generator=g) * 2).to(DEV)
    gc = (torch.randn(B * J, 3, generator=g) * 1e-3).to(DEV)
    coords = torch.empty(B * J, 3, device=DEV); stats = torch.empty(B * J, 5, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    assert L.pl_softargmax3d_nhwc_fwd(logits.data_ptr(), B, J, H, W, coords.data_ptr(), stats.data_ptr(), s) == 0
    dl = torch.empty_like(logits)
    assert L.pl_softargmax3d_nhwc_bwd(logits.data_ptr(), stats.data_ptr(), gc.data_ptr(), B, J, H, W, dl.data_ptr(), s) == 0
    scale = heads._pow2_scale_for_bound(2.0 * gc.abs().sum(1).max())
    bound = float(2.0 * gc.abs().sum(1).max())
    assert 2 ** 13 <= bound * float(scale[0]) < 2 ** 14 and float(scale[0] * scale[1]) == 1.0
    carrier = torch.empty_like(logits)
    rc = L.pl_softargmax3d_nhwc_bwd_ex(logits.data_ptr(), stats.data_ptr(), gc.data_ptr(), B, J, H, W, None, carrier.data_ptr(),
                                       3, scale.data_ptr(), s)
    assert rc == 0, L.pl_last_error()
    n = logits.numel()
    pl16 = carrier.reshape(-1).view(torch.float16)
    back = (pl16[:n].float() + pl16[n:].float() / 2048.0) * scale[1]
    assert float((back - dl.reshape(-1)).abs().max()) <= bound * 2.0 ** -21
    rows, cols = B * H * W, J * 64
    db = torch.empty(cols, device=DEV)
    scratch = torch.empty(L.pl_colsum_scratch_bytes(rows, cols), dtype=torch.uint8, device=DEV)
    rc = L.pl_colsum_planes(carrier.data_ptr(), 3, rows, cols, scale[1:].data_ptr(), db.data_ptr(), scratch.data_ptr(), s)
    assert rc == 0, L.pl_last_error()
    want = dl.reshape(rows, cols).double().sum(0)
    assert float((db.double() - want).abs().max()) <= 1e-5 * float(want.abs().max()) + bound * rows * 2.0 ** -21


# ---------------------------------------------------------------------------- flat arenas + one-launch Adam for the conv models
def _gpu_kernel_names(fn):
    """Names of the GPU kernels one call of fn() launches (torch.profiler)."""
    from torch.profiler import profile, ProfilerActivity
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]


def test_flat_adam_three_steps_of_model3d_vs_torch_adam(pkg):
    """arena.FlatAdam (one pl_adamw_flat launch over the module's flat arenas; parameter gradients written by the library
    straight into the gradient arena) against torch.optim.Adam(model.parameters(), lr) -- phase4_joined/train.py:39,87 -- on
    the same model, same kernels for forward and backward: three steps, parameters within Adam's fp32 round-off (a
    noise-level gradient element may land on the other side of zero: +-lr on that element, as with the lifter's AdamW)."""
    import copy
    torch.manual_seed(5)
    m = pkg.Model_3D()
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 81))
    with torch.no_grad():
        m.final_layer.weight.mul_(1e-2)
    a, b = copy.deepcopy(m).to(DEV).train(), copy.deepcopy(m).to(DEV).train()
    lr = 1e-3
    oa, ob = torch.optim.Adam(a.parameters(), lr=lr), pkg.FlatAdam(b, lr=lr)
    sd_keys = list(b.state_dict().keys())
    assert sd_keys == list(a.state_dict().keys())               # the arena changes no key and no shape
    frames = [pkg.synth.structured_frames(4, 82 + i, size=128).to(DEV) for i in range(3)]
    target = torch.randn(4, 51, device=DEV) * 0.3
    # An untrained network one Adam step (every weight moves by +-lr) away from its seed is chaotic: parameters 1e-7 apart
    # give gradients 10 % apart on the next batch (measured).  So: the first step on each model's OWN backward (identical
    # kernels, hence identical gradients -- this checks the direct-to-arena gradient writes), the next two on the SAME
    # gradients for both optimizers (a's, copied into b's arena) -- this checks three steps of the update arithmetic.
    for i in range(3):
        la = lb = None
        oa.zero_grad()
        la = pkg.mse_loss(a(frames[i]), target)
        la.backward()
        ob.zero_grad()
        if i == 0:
            lb = pkg.mse_loss(b(frames[i]), target)
            lb.backward()
            assert float(la) == float(lb)
            for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
                assert torch.equal(p.grad, q.grad), k
        else:
            for p, q in zip(a.parameters(), b.parameters()):
                q._pl_grad.copy_(p.grad)
                q.grad = q._pl_grad
        oa.step()
        ob.step()
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            d = float((p.detach() - q.detach()).abs().max())
            assert d <= 1e-3 * lr, (i, k, d)                   # measured: 1.2e-4 lr after the first step
    # every gradient sits in the arena (no copy at step time), the optimizer state has the stock Adam layout
    arena = b._pl_arena
    ob.zero_grad()
    pkg.mse_loss(b(frames[0]), target).backward()
    assert all(p.grad is not None and p.grad.data_ptr() == p._pl_grad.data_ptr() for p in arena.params
               if p is not b.preact.conv1.weight)
    st = ob.state_dict()["state"]
    assert set(st[0]) == {"step", "exp_avg", "exp_avg_sq"} and float(st[0]["step"]) == 3.0
    ref = torch.optim.Adam(b.parameters(), lr=lr)
    ref.load_state_dict(ob.state_dict())                        # the stock optimizer can resume from it
    # one training step launches no ATen optimizer / gradient-accumulation kernels any more
    def step():
        ob.zero_grad()
        pkg.mse_loss(b(frames[0]), target).backward()
        ob.step()
    names = _gpu_kernel_names(step)
    aten = [n for n in names if "at::native" in n or "multi_tensor" in n]
    assert not any("multi_tensor" in n for n in aten), aten[:5]
    assert len(aten) <= 30, (len(aten), sorted(set(aten))[:10])      # what is left: a few weight re-layouts (stem, transposed convs)
    assert len(names) > 500


def test_flat_adam_accumulates_a_second_backward_of_the_same_step(pkg):
    """Two forward passes before one optimizer step (the phase5 Flip branch runs every network twice): the first backward of
    a step overwrites the gradient arena, the second accumulates -- equal to autograd's own accumulation."""
    import copy
    torch.manual_seed(6)
    m = pkg.ResNet("resnet50")
    m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), 91))
    a, b = copy.deepcopy(m).to(DEV).train(), copy.deepcopy(m).to(DEV).train()
    pkg.ModuleArena(b)
    x1, x2 = (pkg.synth.structured_frames(4, 92 + i, size=128).to(DEV) for i in range(2))
    for mod in (a, b):
        for p in mod.parameters():
            p.grad = None
        (mod(x1).square().mean() + 0.5 * mod(x2).square().mean()).backward()
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert p.grad is not None and q.grad is not None, k
        den = float(p.grad.abs().max()) + 1e-30
        assert float((p.grad - q.grad).abs().max()) <= 1e-5 * den, (k, float((p.grad - q.grad).abs().max()), den)
