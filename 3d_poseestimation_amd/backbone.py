"""phase4 model, inference path (first slice): ResNet-50 backbone + deconvolution head +
integral soft-argmax, eval-mode forward on the HIP library, NHWC end to end.

    ResNet    /root/reference/phase4_joined/Resnet.py:98-165  (Bottleneck :51-95)
    Model_3D  /root/reference/phase4_joined/Model.py:11-137
    Model_2D  /root/reference/phase5_loop/Model_2d.py:13-138   (depth_dim 1, 17 heat-maps, coordinates in (0, 1))

The modules below are PARAMETER CONTAINERS built from stock nn.Conv2d / nn.BatchNorm2d /
nn.ConvTranspose2d in the reference's construction order, so `state_dict()` has the reference's keys and
shapes (a reference checkpoint loads with load_state_dict) -- pinned by tests/golden/g9: the reference's
ResNet("resnet50") imported and run as-is.  The arithmetic is conv.py's: every convolution with its
BatchNorm (running statistics), ReLU and residual add folded into one launch.  In TRAINING mode the same
containers run the differentiable form of the path (conv.py's differentiable pieces: implicit-GEMM convolution
with dgrad / wgrad on the library, BatchNorm2d on batch statistics with its backward, max-pool, transposed
convolution and residual join with theirs): correct against torch autograd, not yet tuned -- the next slice of
row N2 fuses the statistics into the GEMM epilogue and the BatchNorm backward into dgrad's producer.
Unlike the reference constructor (Model.py:28-38) nothing is downloaded: weights are whatever is loaded.
"""
import torch
import torch.nn as nn

from . import conv
from .heads import soft_argmax_2d, soft_argmax_3d, soft_argmax_3d_nhwc


def _stem_on_planes():
    """POSELIFT_STEM_PLANES=0: the direct fp32 stem kernel of rounds 1-2 (same-box A/B)."""
    import os
    return os.environ.get("POSELIFT_STEM_PLANES", "1") != "0"


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=0.1)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=0.1)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, momentum=0.1)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    """ResNet("resnet50" | "resnet101" | "resnet152"): the Bottleneck architectures of Resnet.py:104-110."""
    LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3], "resnet152": [3, 8, 36, 3]}

    def __init__(self, architecture="resnet50", compute_dtype="f16x3"):
        super().__init__()
        self.compute_dtype = compute_dtype          # "bf16x6" (fp32-grade), "bf16", or "f16x3" (fp32-grade; training: 1x1 convolutions on the planes GEMM)
        if architecture not in self.LAYERS:
            raise ValueError(f"{architecture}: only the Bottleneck ResNets are built (the reference uses resnet50)")
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=1e-5, momentum=0.1, affine=True)
        blocks = self.LAYERS[architecture]
        self.layer1 = self._make_layer(64, blocks[0])
        self.layer2 = self._make_layer(128, blocks[1], stride=2)
        self.layer3 = self._make_layer(256, blocks[2], stride=2)
        self.layer4 = self._make_layer(512, blocks[3], stride=2)
        self._cache = None

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    # ---- folded weights: OHWI kernels and (scale, shift) per BatchNorm, rebuilt when a parameter changes
    def _versions(self):
        return tuple(t._version for t in list(self.parameters()) + list(self.buffers())) + \
            tuple(t.data_ptr() for t in self.parameters())

    def _folded(self):
        v = self._versions()
        if self._cache is None or self._cache[0] != v:
            f = {}
            for name, m in self.named_modules():
                if isinstance(m, nn.Conv2d):
                    f[name] = conv.to_ohwi(m.weight.detach().float())
                elif isinstance(m, nn.BatchNorm2d):
                    f[name] = conv.fold_bn(m)
            f = {k: (tuple(t.detach() for t in x) if isinstance(x, tuple) else x) for k, x in f.items()}
            if self.compute_dtype in ("f16x3", "bf16p") and next(self.parameters()).is_cuda:
                # eval mode on the planes GEMM: the kernels' operand planes, split once per set of weights
                mode = conv._lib.PL_F16X3 if self.compute_dtype == "f16x3" else conv._lib.PL_BF16
                for name in [k for k, x in f.items() if not isinstance(x, tuple) and k != "conv1"]:
                    f[name + "@p"] = conv._planes_of(f[name], conv.WEIGHT_PLANE_SCALE, mode)
                # the stem's 7 x 4-tap kernel on pixel pairs (conv.stem_planes; f["conv1"] is OHWI)
                f["conv1@pairs"] = conv._planes_of(conv._stem_weight_pairs(f["conv1"], ohwi=True), conv.WEIGHT_PLANE_SCALE, mode)
            self._cache = (v, f)
        return self._cache[1]

    def _forward_eval_planes(self, x_nhwc):
        """Eval mode with every convolution but the stem on the planes GEMM (conv.conv2d_planes_eval): the folded BatchNorm /
        ReLU / residual epilogue writes the next convolution's operand planes directly; only the block outputs exist in fp32 as
        well (the next join adds them).  Returns (x fp32, carrier of x's planes).  Resnet.py:135-142, :65-93 under eval()."""
        f = self._folded()
        mode = conv._lib.PL_F16X3 if self.compute_dtype == "f16x3" else conv._lib.PL_BF16
        cpe = conv.conv2d_planes_eval
        with torch.no_grad():
            s, b = f["bn1"]
            Bf, Hf, Wf, _ = x_nhwc.shape
            if _stem_on_planes() and conv.stem_planes_supported(Bf, Hf, Wf, 3, self.conv1.out_channels, 7, 7, 2, 3):
                # the stem as a planes GEMM on the frame's pixel-pair view, folded BatchNorm + ReLU in its epilogue
                x, _ = cpe(conv.stem_input_planes(x_nhwc, mode), f["conv1@pairs"], (self.conv1.out_channels, 7, 4, 8), (2, 1), (3, 2, 1),
                           s, b, relu=1, mode=mode)
            else:
                x = conv.conv2d_nhwc(x_nhwc.float(), f["conv1"], 2, 3, s, b, relu=1, arith=self.compute_dtype)
            x = conv.maxpool3x3s2_nhwc(x)
            xp = conv._planes_of(x, conv.ACT_PLANE_SCALE, mode)
            for li in (1, 2, 3, 4):
                for bi, blk in enumerate(getattr(self, f"layer{li}")):
                    p = f"layer{li}.{bi}"
                    identity = x
                    if blk.downsample is not None:
                        s, b = f[p + ".downsample.1"]
                        identity, _ = cpe(xp, f[p + ".downsample.0@p"], f[p + ".downsample.0"].shape, blk.stride, 0, s, b, mode=mode)
                    s, b = f[p + ".bn1"]
                    _, o = cpe(xp, f[p + ".conv1@p"], f[p + ".conv1"].shape, 1, 0, s, b, relu=1, want_f32=False, want_planes=True, mode=mode)
                    s, b = f[p + ".bn2"]
                    _, o = cpe(o, f[p + ".conv2@p"], f[p + ".conv2"].shape, blk.stride, 1, s, b, relu=1, want_f32=False, want_planes=True, mode=mode)
                    s, b = f[p + ".bn3"]
                    x, xp = cpe(o, f[p + ".conv3@p"], f[p + ".conv3"].shape, 1, 0, s, b, relu=2, resid=identity, want_planes=True, mode=mode)
        return x, xp

    def _planes_eval_ok(self, x_nhwc):
        """Maps large enough for the planes path in every stage (32-bit offsets inside a tensor are the only upper limit)."""
        B, H, W, _ = x_nhwc.shape
        return (self.compute_dtype in ("f16x3", "bf16p") and x_nhwc.is_cuda and H % 32 == 0 and W % 32 == 0 and
                B * (H // 4) * (W // 4) * 256 * 4 < (1 << 31))

    def _forward_train(self, x, want_planes=False):
        """Training mode: batch statistics, running statistics updated, differentiable (Resnet.py:135-142, :65-93)."""
        def w(m):
            return conv.to_ohwi(m.weight.float())
        bnr = conv.batchnorm_relu_train
        ar = self.compute_dtype

        def cv(inp, m, stride, padding):
            return conv.conv2d_nhwc_autograd(inp, w(m), stride, padding, ar)
        planes = self.compute_dtype in ("f16x3", "bf16p")
        Bf, Hf, Wf, _ = x.shape
        if planes and _stem_on_planes() and conv.stem_planes_supported(Bf, Hf, Wf, 3, self.conv1.out_channels, 7, 7, 2, 3):
            # the stem as a planes GEMM on the frame's pixel-pair view (conv.stem_planes), BatchNorm from its epilogue statistics
            mode = conv._lib.PL_F16X3 if self.compute_dtype == "f16x3" else conv._lib.PL_BF16
            link = conv.PlaneLink(mode)
            z = conv.stem_planes(conv.stem_input_planes(x, mode), self.conv1.weight, link)
            x = conv.batchnorm_relu_train_planes(z, self.bn1, True, False, link)
        else:
            x = bnr(cv(x.float(), self.conv1, 2, 3), self.bn1, True)
        x = conv.maxpool3x3s2_nhwc_autograd(x)
        if planes:
            x, xp = self._blocks_train_planes(x, cv, bnr)
            return (x, xp) if want_planes else x
        for li in (1, 2, 3, 4):
            for blk in getattr(self, f"layer{li}"):
                identity = x
                if blk.downsample is not None:
                    identity = bnr(cv(x, blk.downsample[0], blk.stride, 0), blk.downsample[1], False)
                out = bnr(cv(x, blk.conv1, 1, 0), blk.bn1, True)
                out = bnr(cv(out, blk.conv2, blk.stride, 1), blk.bn2, True)
                out = bnr(cv(out, blk.conv3, 1, 0), blk.bn3, False)
                x = conv.add_relu(out, identity)
        return x

    def _blocks_train_planes(self, x, cv, bnr):
        """The Bottleneck stack of a training step on the planes GEMM (conv.py, bottom): every convolution reads operand
        planes written by the kernel that produced its input (residual join / BatchNorm apply) -- the 1x1 ones as plain
        GEMMs, conv2 and the stride-2 downsample with the input gathered by the loader waves -- and every backward reads
        dz planes written by the BatchNorm backward.  Only the block outputs exist in fp32 as well (the next join adds
        them).  Same arithmetic class throughout (fp32-grade products, fp32 accumulation).  Resnet.py:65-93, :139-142."""
        # "f16x3": two fp16 planes per operand (fp32-grade); "bf16p": ONE bf16 plane -- bf16 storage of the GEMM operands
        # (activations, dz, weight shadow), everything off the planes path in bf16 arithmetic too (the throughput mode)
        mode = conv._lib.PL_F16X3 if self.compute_dtype == "f16x3" else conv._lib.PL_BF16
        PlaneLink = lambda: conv.PlaneLink(mode)     # noqa: E731
        bnp = conv.batchnorm_relu_train_planes
        xp = conv.to_planes(x, mode)
        for li in (1, 2, 3, 4):
            for blk in getattr(self, f"layer{li}"):
                B, H, W, cin = x.shape
                mid, cout, st = blk.conv1.out_channels, blk.conv3.out_channels, blk.stride
                ho, wo = (H - 1) // st + 1, (W - 1) // st + 1
                ok = (conv.planes_conv_supported(B * H * W, cin, mid) and conv.planes_conv_supported(B * ho * wo, mid, cout) and
                      conv.planes_convk_supported(B, H, W, mid, mid, 3, st, 1) and
                      (blk.downsample is None or conv.planes_convk_supported(B, H, W, cin, cout, 1, st, 0)))
                if not ok:                     # (tiny maps / odd widths: the plain path, block by block)
                    identity = x
                    if blk.downsample is not None:
                        identity = bnr(cv(x, blk.downsample[0], st, 0), blk.downsample[1], False)
                    out = bnr(cv(x, blk.conv1, 1, 0), blk.bn1, True)
                    out = bnr(cv(out, blk.conv2, st, 1), blk.bn2, True)
                    out = bnr(cv(out, blk.conv3, 1, 0), blk.bn3, False)
                    x, xp = conv.add_relu_planes(out, identity, mode)
                    continue
                identity = x
                if blk.downsample is not None:
                    lk = PlaneLink()
                    xd = conv.planes_twin(x, xp)     # the same planes; the gradient goes to x (conv._PlanesTwinFn)
                    zd = (conv.conv1x1_planes(xd, blk.downsample[0].weight, lk) if st == 1 else
                          conv.conv_planes(xd, blk.downsample[0].weight, st, 0, lk))
                    identity = bnp(zd, blk.downsample[1], False, False, lk)
                l1, l2, l3 = PlaneLink(), PlaneLink(), PlaneLink()
                out = bnp(conv.conv1x1_planes(xp, blk.conv1.weight, l1), blk.bn1, True, True, l1)
                out = bnp(conv.conv_planes(out, blk.conv2.weight, st, 1, l2), blk.bn2, True, True, l2)
                # bn3 and the residual join in one pass: bn3's output is never materialised
                x, xp = conv.bn_join_planes(conv.conv1x1_planes(out, blk.conv3.weight, l3), identity, blk.bn3, l3)
        return x, xp

    def forward(self, x_nhwc):
        """x [B, H, W, 3] fp32 (NHWC, as the phase4 loader delivers frames, Model.py:88) -> [B, H/32, W/32, 2048]."""
        if self.training:
            return self._forward_train(x_nhwc)
        if self._planes_eval_ok(x_nhwc):
            return self._forward_eval_planes(x_nhwc)[0]
        f = self._folded()
        ar = self.compute_dtype
        with torch.no_grad():
            s, b = f["bn1"]
            x = conv.conv2d_nhwc(x_nhwc.float(), f["conv1"], 2, 3, s, b, relu=1, arith=ar)   # Resnet.py:137
            x = conv.maxpool3x3s2_nhwc(x)
            for li in (1, 2, 3, 4):
                for bi, blk in enumerate(getattr(self, f"layer{li}")):
                    p = f"layer{li}.{bi}"
                    identity = x
                    if blk.downsample is not None:                                          # :87-88
                        s, b = f[p + ".downsample.1"]
                        identity = conv.conv2d_nhwc(x, f[p + ".downsample.0"], blk.stride, 0, s, b, arith=ar)
                    s, b = f[p + ".bn1"]
                    out = conv.conv2d_nhwc(x, f[p + ".conv1"], 1, 0, s, b, relu=1, arith=ar)   # :67
                    s, b = f[p + ".bn2"]
                    out = conv.conv2d_nhwc(out, f[p + ".conv2"], blk.stride, 1, s, b, relu=1, arith=ar)   # :69
                    s, b = f[p + ".bn3"]
                    x = conv.conv2d_nhwc(out, f[p + ".conv3"], 1, 0, s, b, relu=2, resid=identity, arith=ar)  # :81-91
        return x


class _HeatmapNet(nn.Module):
    """Backbone + three transposed convolutions + final 1x1 convolution: the part Model_3D and Model_2D share."""

    def __init__(self, depth_dim, architecture="resnet50", compute_dtype="f16x3"):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.deconv_dim = [256, 256, 256]
        self.num_joints, self.depth_dim, self.height_dim, self.width_dim = 17, depth_dim, 64, 64
        self.preact = ResNet(architecture, compute_dtype)
        self.feature_channel = 2048
        layers, cin = [], self.feature_channel
        for cout in self.deconv_dim:                                          # Model.py:47-69 / Model_2d.py:48-71
            layers += [nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1, bias=False),
                       nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
            cin = cout
        self.deconv_layers = nn.Sequential(*layers)
        self.final_layer = nn.Conv2d(self.deconv_dim[2], self.num_joints * self.depth_dim, kernel_size=1)
        self._cache = None

    def _folded(self):
        ts = list(self.deconv_layers.parameters()) + list(self.deconv_layers.buffers()) + list(self.final_layer.parameters())
        v = tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts)
        if self._cache is None or self._cache[0] != v:
            f = {}
            for i in (0, 3, 6):
                f[i] = conv.deconv_subkernels(self.deconv_layers[i].weight.detach().float())
                f[i + 1] = tuple(t.detach() for t in conv.fold_bn(self.deconv_layers[i + 1]))
            f["final"] = conv.to_ohwi(self.final_layer.weight.detach().float())
            if self.compute_dtype in ("f16x3", "bf16p") and self.final_layer.weight.is_cuda:
                mode = conv._lib.PL_F16X3 if self.compute_dtype == "f16x3" else conv._lib.PL_BF16
                for i in (0, 3, 6):
                    f[f"{i}@p"] = conv._planes_of(f[i], conv.WEIGHT_PLANE_SCALE, mode)
                f["final@p"] = conv._planes_of(f["final"], conv.WEIGHT_PLANE_SCALE, mode)
            self._cache = (v, f)
        return self._cache[1]

    def heatmap_logits_nhwc(self, x_nhwc):
        """[B, 256, 256, 3] -> [B, 64, 64, J*depth]: everything in front of the soft-argmax, in the path's layout."""
        f = self._folded()
        if not self.training and self.preact._planes_eval_ok(x_nhwc) and "final@p" in f and f["final"].shape[0] % 8 == 0:
            # the whole eval forward on the planes GEMM: the head's transposed convolutions and final convolution as well
            mode = conv._lib.PL_F16X3 if self.compute_dtype == "f16x3" else conv._lib.PL_BF16
            with torch.no_grad():
                _, outp = self.preact._forward_eval_planes(x_nhwc)
                for i in (0, 3, 6):
                    _, outp = conv.deconv_planes_eval(outp, f[f"{i}@p"], f[i].shape[1], f[i + 1][0], f[i + 1][1], relu=1,
                                                      want_f32=False, want_planes=True, mode=mode)
                y, _ = conv.conv2d_planes_eval(outp, f["final@p"], f["final"].shape, 1, 0, bias=self.final_layer.bias.detach(),
                                               mode=mode)
                return y
        x0 = self.preact(x_nhwc)
        with torch.no_grad():
            out = x0
            for i in (0, 3, 6):
                out = conv.deconv4x4s2_nhwc(out, f[i], f[i + 1][0], f[i + 1][1], relu=1, arith=self.compute_dtype)
            return conv.conv2d_nhwc(out, f["final"], 1, 0, bias=self.final_layer.bias.detach(), arith=self.compute_dtype)

    def heatmap_logits(self, x_nhwc):
        """The same in the reference's layout [B, J*depth, 64, 64] (Model.py:91)."""
        return conv.nhwc_to_nchw(self.heatmap_logits_nhwc(x_nhwc))

    def _heatmap_logits_train(self, x_nhwc, nhwc=False, final_link=None):
        """Training mode, differentiable: [B, H, W, 3] -> [B, J*depth, H/4, W/4] (NCHW for the soft-argmax), or the
        NHWC logits as the final convolution writes them (nhwc=True: the depth-64 head reads them in place)."""
        if self.compute_dtype in ("f16x3", "bf16p"):
            # the head's transposed convolutions on the planes GEMM too (conv.py: _DeconvPlanesFn): the backbone hands over
            # its output as planes, every BatchNorm between two of them writes planes only
            mode = conv._lib.PL_F16X3 if self.compute_dtype == "f16x3" else conv._lib.PL_BF16
            out, outp = self.preact._forward_train(x_nhwc, want_planes=True)
            for i in (0, 3, 6):
                B, H, W, cin = out.shape
                cout = self.deconv_layers[i].weight.shape[1]
                last = i == 6 and final_link is None      # (final_link: the final convolution reads planes as well)
                if outp is not None and conv.planes_deconv_supported(B, H, W, cin, cout):
                    lk = conv.PlaneLink(mode)
                    z = conv.deconv4x4s2_planes(outp, self.deconv_layers[i].weight, lk)
                    out = conv.batchnorm_relu_train_planes(z, self.deconv_layers[i + 1], True, not last, lk)
                    outp = None if last else out
                else:
                    if outp is not None and out is outp:        # a planes-only tensor cannot feed the plain path
                        raise RuntimeError("deconvolution head: map too small for the planes path after a planes-only layer")
                    out = conv.batchnorm_relu_train(
                        conv.deconv4x4s2_nhwc_autograd(out, self.deconv_layers[i].weight, "bf16x6" if mode == conv._lib.PL_F16X3 else "bf16"),
                        self.deconv_layers[i + 1], True)
                    outp = None
            if final_link is not None and outp is not None:
                final_link.mode = mode
                return conv.conv1x1_bias_planes(outp, self.final_layer.weight, self.final_layer.bias, final_link)
            if final_link is not None:
                final_link.mode = None                   # (told the caller: the logits' gradient is wanted in fp32)
            if outp is not None:
                raise RuntimeError("deconvolution head: planes-only activation in front of the plain final convolution")
            out = conv.conv2d_bias_nhwc_autograd(out, conv.to_ohwi(self.final_layer.weight.float()), self.final_layer.bias,
                                                 arith=self.compute_dtype)
            return out if nhwc else conv.nhwc_to_nchw_autograd(out)
        out = self.preact(x_nhwc)
        for i in (0, 3, 6):
            out = conv.batchnorm_relu_train(
                conv.deconv4x4s2_nhwc_autograd(out, self.deconv_layers[i].weight, self.compute_dtype),
                self.deconv_layers[i + 1], True)
        out = conv.conv2d_bias_nhwc_autograd(out, conv.to_ohwi(self.final_layer.weight.float()), self.final_layer.bias,
                                             arith=self.compute_dtype)
        return out if nhwc else conv.nhwc_to_nchw_autograd(out)


    def predict_nhwc(self, x_nhwc):
        """Coordinates from NHWC frames, in whichever mode the module is in (phase5's cycle step feeds BOTH networks
        the same frames; their reference forwards disagree about the input layout, this entry point does not)."""
        if self.training:
            if self.depth_dim == 64:
                if self.compute_dtype in ("f16x3", "bf16p"):
                    # the final convolution on the planes GEMM: its gradient arrives as planes written by the soft-argmax backward
                    lk = conv.PlaneLink()
                    logits = self._heatmap_logits_train(x_nhwc, nhwc=True, final_link=lk)
                    return soft_argmax_3d_nhwc(logits, self.num_joints, lk if lk.mode is not None else None)
                return soft_argmax_3d_nhwc(self._heatmap_logits_train(x_nhwc, nhwc=True), self.num_joints)
            logits = self._heatmap_logits_train(x_nhwc)
            return (soft_argmax_3d(logits, self.num_joints, self.depth_dim) if self.depth_dim > 1
                    else soft_argmax_2d(logits, self.num_joints))
        with torch.no_grad():
            if self.depth_dim == 64:
                return soft_argmax_3d_nhwc(self.heatmap_logits_nhwc(x_nhwc), self.num_joints)
            logits = self.heatmap_logits(x_nhwc)
            return (soft_argmax_3d(logits, self.num_joints, self.depth_dim) if self.depth_dim > 1
                    else soft_argmax_2d(logits, self.num_joints))


class Model_3D(_HeatmapNet):
    def __init__(self, architecture="resnet50", compute_dtype="f16x3"):
        super().__init__(64, architecture, compute_dtype)

    def forward(self, x):
        """x [B, 256, 256, 3] NHWC frames -> [B, 51] (x, y, z) per joint in (-1, 1)  (Model.py:83-137)."""
        if self.training:
            return self.predict_nhwc(x)
        with torch.no_grad():
            return soft_argmax_3d_nhwc(self.heatmap_logits_nhwc(x), self.num_joints)


class Model_2D(_HeatmapNet):
    def __init__(self, architecture="resnet50", compute_dtype="f16x3"):
        super().__init__(1, architecture, compute_dtype)

    def forward(self, x):
        """x [B, 3, 256, 256] NCHW frames (Model_2d.py:91 leaves the permute commented out) -> [B, 34]
        (x, y) per joint in (0, 1)  (Model_2d.py:87-136)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"Model_2D expects NCHW frames [B, 3, H, W], got {tuple(x.shape)}")
        if self.training:
            return soft_argmax_2d(self._heatmap_logits_train(x.permute(0, 2, 3, 1).contiguous()), self.num_joints)
        with torch.no_grad():
            return soft_argmax_2d(self.heatmap_logits(x.permute(0, 2, 3, 1).contiguous()), self.num_joints)
