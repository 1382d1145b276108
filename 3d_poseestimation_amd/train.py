"""Train / eval step and metrics of the lifter path, mirroring the reference script.

  train_step        /root/reference/phase1_lifting/train_1.py:75-100
  eval_step         /root/reference/phase1_lifting/train_1.py:112-145
  loss_MPJPE        /root/reference/phase1_lifting/train_1.py:19-23
  epoch_mpjpe_mm    /root/reference/phase1_lifting/train_1.py:100-104
  cycle_step        /root/reference/phase5_loop/train_5 copy.py:147-236 (the Triangle branch, without its Flip pass)
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


def flip_pose(data):
    """utils.py:372-396: horizontal flip of (N, 17, 2|3) poses (x -> 1-x or -x, left/right joints
    swapped).  The flip-TTA of train_1.py:128-134 is `(flip_pose(model(flip_pose(x))) + model(x)) / 2`."""
    data = data.contiguous()
    _lib.require_device_tensor(data, "data")
    if data.dim() != 3 or data.shape[1] != 17 or data.shape[2] not in (2, 3):
        raise ValueError("flip_pose expects (N, 17, 2) or (N, 17, 3)")
    out = torch.empty_like(data)
    with _lib.on_device(data.device):
        rc = _lib.lib().pl_flip_pose(data.data_ptr(), out.data_ptr(), data.shape[0], 17, data.shape[2],
                                     _lib.current_stream_ptr())
    _lib.check(rc, "pl_flip_pose")
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
            loss, y2_hat = model.fused_train_fwd_bwd(y1.reshape(y1.shape[0], -1).contiguous(),
                                                     y2.reshape(y2.shape[0], -1).contiguous(),
                                                     grad_sync if model._grad_sync is grad_sync else None)
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
            with torch.cuda.graph(self.graph):
                self.loss, y_hat = model.fused_train_fwd_bwd(self._x, self._y, None, step_dev=self._tick)
                optimizer._enqueue_dev(self._lr_dev, t0, self._tick)
            self.y_hat = y_hat.reshape(self._out_shape)
            model._step = step0                                              # capturing ran nothing

    def __call__(self, y1, y2):
        lr = float(self.opt.param_groups[0]["lr"])
        if lr != self._lr:                                                   # ReduceLROnPlateau et al. (train_1.py:106)
            self._lr = lr
            self._lr_dev.fill_(lr)
        self._x.copy_(y1.reshape(self._x.shape))
        self._y.copy_(y2.reshape(self._y.shape))
        self.model._ensure_wplanes()         # parameters changed behind the graph's back (load_state_dict, ...)
        self.graph.replay()
        self.model._step += 1
        self.opt._advance_host(1)
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


def cycle_step(model_2d, model_3d, model_lift, optimizers, frame_nhwc, y1, y2, loss_function, model_proj=None):
    """One phase5 cycle step (train_5 copy.py:147-236, `Triangle` on, `Flip` off): zero_grad on every optimizer;
    y1^ = model_2d(frame), y2^ = model_3d(frame); the lifter on the predicted AND on the ground-truth 2-D pose (two
    calls in one graph: its input gradient flows back into model_2d); optionally the projector on y2^ and y2;
    TriangleLoss; ONE backward; every optimizer steps.  frame_nhwc [B, H, W, 3]; y1 [B, 17, 2]; y2 [B, 17, 3].
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
    kw = {}
    if model_proj is not None:
        kw = dict(proj_3d_pred=model_proj(y2_hat).reshape(B, 17, 2), proj_3d_gt=model_proj(y2).reshape(B, 17, 2))
    out = loss_function(predicted_2d=y1_hat, predicted_3d=y2_hat, lift_2d_gt=lift_2d_gt, lift_2d_pred=lift_2d_pred,
                        gt_2d=y1, gt_3d=y2, **kw)
    loss = out[0] if isinstance(out, tuple) else out
    loss.backward()
    for opt in optimizers:
        opt.step()
    return loss.detach(), y1_hat.detach(), y2_hat.detach()
