"""Train / eval step and metrics of the lifter path, mirroring the reference script.

  train_step        /root/reference/phase1_lifting/train_1.py:75-100
  eval_step         /root/reference/phase1_lifting/train_1.py:112-145
  loss_MPJPE        /root/reference/phase1_lifting/train_1.py:19-23
  epoch_mpjpe_mm    /root/reference/phase1_lifting/train_1.py:100-104
  cycle_step        /root/reference/phase5_loop/train_5 copy.py:147-236 (the Triangle branch; Flip=True: :174-199)
All arithmetic is in libposelift.so; tensors must live on the ROCm device.
"""
import torch

from . import _lib


class _MSEFn(torch.autograd.Function):
    """nn.MSELoss(reduction='mean') with the gradient produced in the same pass."""

    @staticmethod
    def forward(ctx, pred, tgt):
        _lib.require_device_tensor(pred, "pred")
        _lib.require_device_tensor(tgt, "target")
        if pred.shape != tgt.shape:
            raise ValueError(f"MSE shapes differ: {tuple(pred.shape)} vs {tuple(tgt.shape)}")
        n = pred.numel()
        need = pred.requires_grad
        dpred = torch.empty_like(pred) if need else None
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        scratch = torch.empty(_lib.lib().pl_mse_scratch_bytes(n), dtype=torch.uint8, device=pred.device)
        with _lib.on_device(pred.device):
            rc = _lib.lib().pl_mse_fwd_bwd(pred.data_ptr(), tgt.data_ptr(), n, 1.0,
                                           dpred.data_ptr() if need else None, loss.data_ptr(),
                                           scratch.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "pl_mse_fwd_bwd")
        ctx.dpred = dpred
        return loss

    @staticmethod
    def backward(ctx, g):
        d = ctx.dpred
        ctx.dpred = None
        return (d.mul_(g) if d is not None else None), None


def mse_loss(pred, tgt):
    """torch.nn.MSELoss(reduction="mean")(pred, tgt)  (train_1.py:37,94)."""
    return _MSEFn.apply(pred.contiguous(), tgt.contiguous())


def loss_MPJPE(prediction, target, out=None):
    """train_1.py:19-23: (B,J,3),(B,J,3) -> (J,) sum over the batch of per-joint L2 errors.
    With `out`, accumulates into it (the reference's `train_metric_3d +=`)."""
    prediction, target = prediction.detach().contiguous(), target.detach().contiguous()
    _lib.require_device_tensor(prediction, "prediction")
    _lib.require_device_tensor(target, "target")
    B, J, d = target.shape
    if d != 3 or prediction.shape != target.shape:
        raise ValueError("loss_MPJPE expects two (B, J, 3) tensors")
    metric = out if out is not None else torch.zeros(J, dtype=torch.float32, device=target.device)
    scratch = torch.empty(_lib.lib().pl_mpjpe_scratch_bytes(B, J), dtype=torch.uint8, device=target.device)
    with _lib.on_device(target.device):
        rc = _lib.lib().pl_mpjpe_accum(prediction.data_ptr(), target.data_ptr(), B, J, metric.data_ptr(),
                                       scratch.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "pl_mpjpe_accum")
    return metric


def _check_pose(data, what):
    _lib.require_device_tensor(data, what)
    if data.dim() != 3 or data.shape[1] != 17 or data.shape[2] not in (2, 3):
        raise ValueError(f"{what}: expects (N, 17, 2) or (N, 17, 3)")


def _flip_ex(src, addend, x_offset, scale):
    out = torch.empty_like(src)
    with _lib.on_device(src.device):
        rc = _lib.lib().pl_flip_pose_ex(src.data_ptr(), addend.data_ptr() if addend is not None else None, out.data_ptr(),
                                        src.shape[0], 17, src.shape[2], float(x_offset), float(scale),
                                        _lib.current_stream_ptr())
    _lib.check(rc, "pl_flip_pose_ex")
    return out


class _FlipAvgFn(torch.autograd.Function):
    """y = (flip_pose(a) + b) * scale [b optional] with its backward (flip_pose is an affine involution: the gradient is
    the joint swap with x negated -- no `1 - x` offset)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a = a.contiguous()
        _check_pose(a, "flip_pose input")
        if b is not None:
            b = b.contiguous()
            _check_pose(b, "flip_pose addend")
            if b.shape != a.shape:
                raise ValueError(f"flip average: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        ctx.scale, ctx.has_b = scale, b is not None
        return _flip_ex(a, b, 1.0 if a.shape[2] == 2 else 0.0, scale)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        da = _flip_ex(g, None, 0.0, ctx.scale) if ctx.needs_input_grad[0] else None
        db = g * ctx.scale if (ctx.has_b and ctx.needs_input_grad[1]) else None
        return da, db, None


def flip_pose(data):
    """utils.py:372-396: horizontal flip of (N, 17, 2|3) poses (x -> 1-x or -x, left/right joints
    swapped); differentiable.  The flip-TTA of train_1.py:128-134 is `(flip_pose(model(flip_pose(x))) + model(x)) / 2`."""
    return _FlipAvgFn.apply(data, None, 1.0)


def flip_average(a, b):
    """(flip_pose(a) + b) / 2 in one pass, differentiable in both: the combination every line of the training-mode Flip
    branch forms (train_5 copy.py:180-196)."""
    return _FlipAvgFn.apply(a, b, 0.5)


def flip_frames_nhwc(frame_nhwc):
    """torch.flip(frame, (3,)) of train_5 copy.py:176 (width of the NCHW frame) on the NHWC frames this path keeps."""
    f = frame_nhwc.contiguous()
    _lib.require_device_tensor(f, "frames")
    if f.dim() != 4:
        raise ValueError("flip_frames_nhwc expects [B, H, W, C]")
    out = torch.empty_like(f)
    with _lib.on_device(f.device):
        rc = _lib.lib().pl_flip_w_nhwc(f.data_ptr(), out.data_ptr(), f.shape[0], f.shape[1], f.shape[2], f.shape[3],
                                       _lib.current_stream_ptr())
    _lib.check(rc, "pl_flip_w_nhwc")
    return out


def epoch_mpjpe_mm(metric_sum, n_samples, num_of_joints=17, zero_centre=True):
    """train_1.py:100-104 reproduced literally: /len(dataset), mean over joints 1..16,
    then *(17/16)*1000 (mm) when the 17-joint root-centred layout is used."""
    m = torch.mean((metric_sum / n_samples)[1:17])
    if num_of_joints == 17 and zero_centre:
        m = m * (17 / 16) * 1000
    return m


def _fusable(model, optimizer, y1, y2):
    from .model import LinearModel
    from .optim import FlatAdamW
    return (isinstance(model, LinearModel) and isinstance(optimizer, FlatAdamW) and optimizer._model is model
            and model.training and torch.is_grad_enabled() and not y1.requires_grad and y1.is_cuda and y2.is_cuda
            and model._inject_keep is None and model._arenas_intact()
            and y2.numel() == y2.shape[0] * model.output_size
            and y1.numel() == y1.shape[0] * model.input_size)


def train_step(model, optimizer, y1, y2, grad_sync=None):
    """One train_1.py:75-100 step: zero_grad, forward, reshape (B,J,3), MSE(mean), backward,
    [gradient all-reduce], optimizer.step.  Returns (loss, y2_hat) as device tensors -- the
    caller decides when to pay the host sync the reference pays every step (train_1.py:98).
    grad_sync: optional callable(model) -> grad_scale, e.g. dp.GradSync; attach it with
    model.set_grad_sync(grad_sync) to have its all-reduce overlapped with the backward pass."""
    y1, y2 = y1.float(), y2.float()
    if _fusable(model, optimizer, y1, y2):
        # the whole step is three library calls: fwd+loss+bwd, [all-reduce], AdamW.  Same kernels,
        # same results as the autograd route below; `zero_grad` is implicit (gradients are overwritten)
        with _lib.on_device(y1.device):
            x2, t2 = y1.reshape(y1.shape[0], -1).contiguous(), y2.reshape(y2.shape[0], -1).contiguous()
            if grad_sync is None and model._grad_sync is None and model.step_carries_adamw(x2.shape[0]):
                # small batches: AdamW rides in the backward launches (pl_lifter_train_step) -- one library call per step
                adamw = optimizer._step_struct(float(optimizer.param_groups[0]["lr"]), None, optimizer._t + 1, None)
                if adamw is not None:
                    loss, y2_hat = model.fused_train_fwd_bwd(x2, t2, None, adamw=adamw)
                    optimizer._advance_host(1)
                    return loss, y2_hat.reshape(y2.shape)
            loss, y2_hat = model.fused_train_fwd_bwd(x2, t2, grad_sync if model._grad_sync is grad_sync else None)
        optimizer.step(grad_scale=grad_sync(model) if grad_sync is not None else 1.0)
        return loss, y2_hat.reshape(y2.shape)
    optimizer.zero_grad()
    y2_hat = model(y1).reshape(y2.shape)
    loss = mse_loss(y2_hat, y2)
    loss.backward()
    if grad_sync is not None:
        optimizer.step(grad_scale=grad_sync(model))
    else:
        optimizer.step()
    return loss, y2_hat


class GraphedTrainStep:
    """The train_1.py:75-100 step captured ONCE as a hipGraph (torch.cuda.graph) and replayed: the ~50 kernel launches
    of a step become one graph launch, which is what a launch-bound step wants (B = 64, BASELINE configs[0]: the
    host needs ~3.5 us per launch, the kernels far less).

        step = GraphedTrainStep(model, optimizer, x_example, y_example)
        loss, y_hat = step(x, y)            # same results, bit for bit, as train_step(model, optimizer, x, y)

    What changes from step to step lives in device memory and is advanced INSIDE the graph: the dropout stream's
    step number and AdamW's t come from one device counter (PLDesc.step_dev, pl_adamw_flat_dev), the learning rate
    from a device scalar the host refreshes when a scheduler changed it.  Single process (a captured RCCL
    all-reduce is not wired); LinearModel + FlatAdamW; fixed batch shape."""

    def __init__(self, model, optimizer, y1, y2):
        y1, y2 = y1.float(), y2.float()
        if not _fusable(model, optimizer, y1, y2) or model._grad_sync is not None:
            raise _lib.PoseliftError("GraphedTrainStep needs a training-mode LinearModel on the GPU with its FlatAdamW "
                                     "and no gradient sync attached")
        self.model, self.opt = model, optimizer
        B = y1.shape[0]
        dev = y1.device
        self._x = y1.reshape(B, -1).contiguous().clone()
        self._y = y2.reshape(B, -1).contiguous().clone()
        self._out_shape = tuple(y2.shape)
        self._tick = torch.zeros(1, dtype=torch.int64, device=dev)          # completed replays
        self._lr = float(optimizer.param_groups[0]["lr"])
        self._lr_dev = torch.full((1,), self._lr, dtype=torch.float32, device=dev)
        optimizer._bind()
        with _lib.on_device(dev):
            # one eager step on a snapshot: every kernel is loaded and the workspace sits in the model's pool before
            # anything is captured; the snapshot is then restored (the step must not count)
            snap = [t.clone() for t in (model.flat_params, model._bn_running, model._bn_batches, optimizer._m, optimizer._v)]
            step0, t0 = model._step, optimizer._t
            train_step(model, optimizer, self._x, self._y.reshape(self._out_shape))
            for dst, src in zip((model.flat_params, model._bn_running, model._bn_batches, optimizer._m, optimizer._v), snap):
                dst.copy_(src)
            model._step, optimizer._t = step0, t0
            optimizer._step_tensor.fill_(float(t0))
            model._ensure_wplanes()          # restoring the snapshot made the persistent weight planes stale
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            adamw = optimizer._step_struct(0.0, self._lr_dev, t0, self._tick) if model.step_carries_adamw(B) else None
            self._in_call = adamw is not None
            with torch.cuda.graph(self.graph):
                if adamw is not None:        # small batches: the optimizer step is inside the same call's launches
                    self.loss, y_hat = model.fused_train_fwd_bwd(self._x, self._y, None, step_dev=self._tick, adamw=adamw)
                else:
                    self.loss, y_hat = model.fused_train_fwd_bwd(self._x, self._y, None, step_dev=self._tick)
                    optimizer._enqueue_dev(self._lr_dev, t0, self._tick)
            self.y_hat = y_hat.reshape(self._out_shape)
            model._step = step0                                              # capturing ran nothing
        # the graph owns the dropout-stream step and AdamW's t (both = the capture-time base + the device counter):
        # what __call__ checks the host-side counters against
        self._step0, self._t0, self._seed0, self.replays = step0, t0, model._seed, 0
        self._ptrs = self._baked_ptrs()

    @property
    def inputs(self):
        """The graph's own input buffers (x [B, in], y [B, out], flat): fill them in place and call step(*step.inputs) to
        replay without the two device-to-device copies of a call with other tensors."""
        return self._x, self._y

    def _baked_ptrs(self):
        return (self.model.flat_params.data_ptr(), self.opt._m.data_ptr() if self.opt._m is not None else 0,
                self.opt._v.data_ptr() if self.opt._v is not None else 0)

    def __call__(self, y1, y2):
        lr = float(self.opt.param_groups[0]["lr"])
        if lr != self._lr:                                                   # ReduceLROnPlateau et al. (train_1.py:106)
            self._lr = lr
            self._lr_dev.fill_(lr)
        # an eager train_step / optimizer.step() / load_state_dict() between replays moves the host-side counters but
        # not the device counter the captured launches read: re-seed it when both moved together, refuse otherwise
        ds, dt = self.model._step - self._step0, self.opt._t - self._t0
        if self.model._seed != self._seed0 or self._baked_ptrs() != self._ptrs:
            raise _lib.PoseliftError("GraphedTrainStep: the dropout seed or an arena address changed after capture "
                                     "(manual_seed, optimizer.load_state_dict, .to()): capture a new GraphedTrainStep")
        if ds != dt or ds < 0:
            raise _lib.PoseliftError(f"GraphedTrainStep: the model ran {ds} steps since capture but the optimizer {dt}: "
                                     "the graph advances both from one device counter -- capture a new one")
        if ds != self.replays:
            self._tick.fill_(ds)
            self.replays = ds
        # (a feeder that writes the next batch straight into `inputs` passes those tensors back: nothing to copy)
        if y1.data_ptr() != self._x.data_ptr():
            self._x.copy_(y1.reshape(self._x.shape))
        if y2.data_ptr() != self._y.data_ptr():
            self._y.copy_(y2.reshape(self._y.shape))
        if not self._in_call:
            self.model._ensure_wplanes()     # parameters changed behind the graph's back (load_state_dict, ...)
        self.graph.replay()
        if self._in_call:
            # a small-batch step neither reads nor refreshes the persistent weight planes: they are stale from here on, and
            # the next call that wants them (an evaluation at a batch on the planes path) refreshes them
            self.model._wplanes_ver = None
        self.model._step += 1
        self.opt._advance_host(1)
        self.replays += 1
        return self.loss, self.y_hat


class GraphedModuleStep:
    """A whole training step of the conv models -- phase4_joined/train.py:69-89: zero_grad, Model_3D forward, loss, backward,
    Adam -- captured ONCE as a hipGraph and replayed.  The step is ~1,100 launches driven from Python autograd nodes; at the
    reference's own batch (8 frames, train.py:187) the host needs 14-17 ms to issue what the GPU finishes in less, and a replay is
    one launch.  Same kernels in the same order: results are bit-identical to the eager step's.

        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        step = GraphedModuleStep(model, opt, torch.nn.functional.mse_loss, frames_example, target_example)
        loss = step(frames, target)         # a device scalar, overwritten by the next call

    Construction runs `warmup` eager steps on the example batch (every kernel loaded, the allocator's pool sized) and then
    puts parameters, buffers and optimizer state back exactly as they were, so the first replay is the model's first step.
    Fixed shapes; single process; the optimizer must keep its step state on the device (torch: capturable=True; give lr as
    a tensor to change it between replays).  forward(*inputs) is whatever the module's forward takes; loss_fn(output,
    target)."""

    def __init__(self, model, optimizer, loss_fn, inputs, target, warmup=2):
        inputs = tuple(inputs) if isinstance(inputs, (tuple, list)) else (inputs,)
        dev = target.device
        if dev.type != "cuda" or any(t.device != dev for t in inputs):
            raise _lib.PoseliftError("GraphedModuleStep: inputs and target must sit on one GPU")
        if not model.training:
            raise _lib.PoseliftError("GraphedModuleStep captures a TRAINING step: call model.train() first")
        for gr in optimizer.param_groups:
            if not gr.get("capturable", False):
                raise _lib.PoseliftError("GraphedModuleStep: the optimizer must be built with capturable=True "
                                         "(its step counter has to live on the device)")
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self._in = tuple(t.detach().clone() for t in inputs)
        self._target = target.detach().clone()
        with _lib.on_device(dev):
            torch.cuda.synchronize(dev)
            state = [t for t in model.state_dict().values()]
            snap = [t.detach().clone() for t in state]
            had = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in optimizer.state.get(p, {}).items()}
                   for gr in optimizer.param_groups for p in gr["params"]}
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, int(warmup))):
                    self._step()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for dst, src in zip(state, snap):
                    dst.copy_(src)
                for p, old in had.items():                                   # optimizer state: back to what it was, IN PLACE
                    for k, v in optimizer.state.get(p, {}).items():          # (a fresh optimizer: zeros, step 0)
                        if torch.is_tensor(v):
                            v.copy_(old[k]) if k in old else v.zero_()
            optimizer.zero_grad(set_to_none=True)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._step()
        self.replays = 0

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(*self._in), self._target)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, inputs, target):
        inputs = tuple(inputs) if isinstance(inputs, (tuple, list)) else (inputs,)
        for dst, src in zip(self._in, inputs):
            dst.copy_(src)
        self._target.copy_(target)
        self.graph.replay()
        self.replays += 1
        return self.loss


@torch.no_grad()
def predict_flip_tta(model, y1, out_dims=3):
    """(flip_pose(model(flip_pose(y1))) + model(y1)) / 2 for a model in eval mode, as ONE forward
    of 2B rows (eval BatchNorm is row-wise): pack [y1; flip(y1)], lift, merge.  The intent of the
    reference's `Flip` branches (train_1.py:128-134, which at HEAD never flips the input;
    phase5_loop/train_5 copy.py:160-171)."""
    if model.training:
        raise ValueError("flip TTA is an eval-mode operation (BatchNorm on running statistics)")
    y1 = y1.float().contiguous()
    _lib.require_device_tensor(y1, "y1")
    if y1.dim() != 3 or y1.shape[1] != 17 or y1.shape[2] not in (2, 3):
        raise ValueError("flip TTA expects (B, 17, 2|3) poses")
    B, L = y1.shape[0], _lib.lib()
    xx = torch.empty((2 * B,) + tuple(y1.shape[1:]), dtype=torch.float32, device=y1.device)
    with _lib.on_device(y1.device):
        _lib.check(L.pl_flip_tta_pack(y1.data_ptr(), xx.data_ptr(), B, 17, y1.shape[2], _lib.current_stream_ptr()),
                   "pl_flip_tta_pack")
        yy = model(xx).reshape(2 * B, 17, out_dims).contiguous()
        out = torch.empty(B, 17, out_dims, dtype=torch.float32, device=y1.device)
        _lib.check(L.pl_flip_tta_merge(yy.data_ptr(), out.data_ptr(), B, 17, out_dims, _lib.current_stream_ptr()),
                   "pl_flip_tta_merge")
    return out


@torch.no_grad()
def eval_step(model, y1, y2, metric_out=None, flip=False):
    """train_1.py:112-145 body (model must be in eval mode): forward, MSE, loss_MPJPE.
    flip=True: flip test-time augmentation (predict_flip_tta)."""
    y1, y2 = y1.float(), y2.float()
    y2_hat = predict_flip_tta(model, y1, y2.shape[-1]).reshape(y2.shape) if flip else model(y1).reshape(y2.shape)
    loss = mse_loss(y2_hat, y2)
    metric = loss_MPJPE(y2_hat, y2, out=metric_out)
    return loss, metric, y2_hat


def cycle_step(model_2d, model_3d, model_lift, optimizers, frame_nhwc, y1, y2, loss_function, model_proj=None, Flip=False):
    """One phase5 cycle step (train_5 copy.py:147-236, `Triangle` on): zero_grad on every optimizer;
    y1^ = model_2d(frame), y2^ = model_3d(frame); the lifter on the predicted AND on the ground-truth 2-D pose (two
    calls in one graph: its input gradient flows back into model_2d); optionally the projector on y2^ and y2;
    [the Flip branch]; TriangleLoss; ONE backward; every optimizer steps.
    Flip=True reproduces train_5 copy.py:174-199 literally: every network runs a SECOND training-mode forward (its
    BatchNorm statistics move twice per step, as in the reference) -- the backbones on the mirrored frames, the lifter
    on the AVERAGED 2-D prediction (:183 feeds `y1_hat` after :180 has overwritten it with the average; it is not
    flipped) and on the flipped ground truth (:184), the projector on the averaged 3-D prediction and on flip(y2) -- and
    each result becomes (flip_pose(second) + first) / 2.
    frame_nhwc [B, H, W, 3]; y1 [B, 17, 2]; y2 [B, 17, 3].
    Returns (loss, y1_hat, y2_hat) as device tensors (the reference's `.cpu().item()` per step is the caller's choice)."""
    for opt in optimizers:
        opt.zero_grad()
    B = y1.shape[0]
    y1, y2 = y1.float(), y2.float()
    frame = frame_nhwc.float()
    y1_hat = model_2d.predict_nhwc(frame).reshape(B, 17, 2)
    y2_hat = model_3d.predict_nhwc(frame).reshape(B, 17, 3)
    lift_2d_pred = model_lift(y1_hat).reshape(B, 17, 3)
    lift_2d_gt = model_lift(y1).reshape(B, 17, 3)
    proj_3d_pred = proj_3d_gt = None
    if model_proj is not None:
        proj_3d_pred = model_proj(y2_hat).reshape(B, 17, 2)
        proj_3d_gt = model_proj(y2).reshape(B, 17, 2)
    if Flip:
        frame_f = flip_frames_nhwc(frame)                                                    # :176
        y1_f = flip_pose(y1)                                                                 # :178
        y1_hat = flip_average(model_2d.predict_nhwc(frame_f).reshape(B, 17, 2), y1_hat)      # :180
        y2_hat = flip_average(model_3d.predict_nhwc(frame_f).reshape(B, 17, 3), y2_hat)      # :181
        lift_2d_pred = flip_average(model_lift(y1_hat).reshape(B, 17, 3), lift_2d_pred)      # :184
        lift_2d_gt = flip_average(model_lift(y1_f).reshape(B, 17, 3), lift_2d_gt)            # :185
        if model_proj is not None:
            y2_f = flip_pose(y2)                                                             # :189
            proj_3d_pred = flip_average(model_proj(y2_hat).reshape(B, 17, 2), proj_3d_pred)  # :191
            proj_3d_gt = flip_average(model_proj(y2_f).reshape(B, 17, 2), proj_3d_gt)        # :192
        # (:194, :197-198 flip y2, the frame and y1 back: out of place here, the originals are still y1 / y2 / frame)
    kw = {}
    if model_proj is not None:
        kw = dict(proj_3d_pred=proj_3d_pred, proj_3d_gt=proj_3d_gt)
    out = loss_function(predicted_2d=y1_hat, predicted_3d=y2_hat, lift_2d_gt=lift_2d_gt, lift_2d_pred=lift_2d_pred,
                        gt_2d=y1, gt_3d=y2, **kw)
    loss = out[0] if isinstance(out, tuple) else out
    loss.backward()
    for opt in optimizers:
        opt.step()
    return loss.detach(), y1_hat.detach(), y2_hat.detach()
