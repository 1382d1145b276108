"""Data-parallel driver of the lifter step: one process per GPU, RCCL over xGMI.

The reference is single-process; this is the row-(e) extension.  Poses are
independent except through the gradient sum, so each rank runs the whole step on its own
shard of the batch (BatchNorm over the local 4096 rows, the DDP convention) and the only
collective is ONE sum all-reduce of the flat gradient arena (4.3 M fp32 = 17.19 MB); the
1/world_size average is folded into the AdamW kernel (grad_scale), so no extra pass.

`backend="nccl"` is RCCL on ROCm.  The same code runs on `gloo` with CPU tensors, which is
how tests/test_dp_gloo.py covers it without a GPU.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*.
    Returns (rank, local_rank, world_size); world_size 1 without initialising anything."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool's hosts
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("POSELIFT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def local_device(local_rank):
    """cuda device of this rank: LOCAL_RANK, folded onto the visible devices (a 1-GPU box can
    rehearse a multi-rank job with POSELIFT_DIST_BACKEND=gloo, all ranks on cuda:0)."""
    n = torch.cuda.device_count()
    return torch.device("cuda", local_rank % max(n, 1))


def shard_rows(n_rows, rank, world):
    """Contiguous shard [lo, hi) of a global batch: the first n_rows % world ranks get one
    extra row, so every row is owned exactly once."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class GradSync:
    """Sum-all-reduce of a flat gradient arena; returns the scale the optimizer applies.

    bucket_bytes splits the arena into contiguous buckets (default: one bucket -- at 17 MB a
    single RCCL call is latency-optimal on the fully connected xGMI mesh)."""

    def __init__(self, group=None, bucket_bytes=None, overlap=True):
        self.group, self.bucket_bytes = group, bucket_bytes
        self._pending = []
        self._overlap, self._defer = bool(overlap), 0

    # ---- when may backward launch buckets? -----------------------------------------------------
    # Only from the backward that completes the optimizer step's gradient: LinearModel launches them from the
    # backward of the last live graph (two lifter calls in one graph contribute first), and the caller marks
    # gradient-accumulation micro-batches with no_sync(), exactly as with DistributedDataParallel.  Whatever was
    # not overlapped is reduced as one whole-arena all-reduce in __call__.
    def overlap_enabled(self):
        return self._overlap and self._defer == 0

    def no_sync(self):
        """Context manager: backward passes inside it launch no all-reduce (gradient accumulation)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            self._defer += 1
            try:
                yield self
            finally:
                self._defer -= 1
        return _cm()

    def has_pending(self):
        return bool(self._pending)

    def abandon(self):
        """Wait for and forget buckets in flight (their result is unusable)."""
        for w in self._pending:
            w.wait()
        self._pending = []

    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def reduce_flat(self, flat):
        world = self.world()
        if world == 1:
            return 1.0
        if self.bucket_bytes:
            step = max(1, self.bucket_bytes // flat.element_size())
            works = [dist.all_reduce(flat[i:i + step], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                     for i in range(0, flat.numel(), step)]
            for w in works:
                w.wait()
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / world

    # ---- overlapped form: LinearModel.set_grad_sync(self) makes backward call launch_bucket()
    #      as soon as a contiguous part of the gradient arena is final -------------------------
    def launch_bucket(self, flat_slice):
        if self.world() > 1:
            self._pending.append(dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def __call__(self, model):
        """Scale to apply to the summed gradients.  Waits for buckets launched during backward;
        if backward launched none (no overlap attached, no_sync() micro-batches, several lifter calls in
        one graph) the whole arena is reduced here."""
        if self._pending:
            for w in self._pending:
                w.wait()
            self._pending = []
            return 1.0 / self.world()
        return self.reduce_flat(model.flat_grads)


def all_gather_slabs(buf, group=None):
    """buf [world][n] on the device with this rank's slab filled: fill the other slabs (the
    PLSync gather contract of include/poselift.h).  RCCL gathers in place on its own stream,
    ordered against the current stream by torch.distributed."""
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(buf, buf[rank], group=group)
    else:
        parts = [torch.empty_like(buf[0]) for _ in range(buf.shape[0])]
        dist.all_gather(parts, buf[rank].clone(), group=group)
        for r, part in enumerate(parts):
            if r != rank:
                buf[r].copy_(part)


def broadcast_model(model, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and BatchNorm buffers."""
    if not dist.is_initialized():
        return
    dist.broadcast(model.flat_params, src, group=group)
    dist.broadcast(model._bn_running, src, group=group)
    dist.broadcast(model._bn_batches, src, group=group)


class FlatGrads:
    """One flat gradient buffer for any module (the conv models keep ordinary torch parameters): every p.grad is a
    view into it, autograd accumulates straight into the views, and the data-parallel step is ONE sum all-reduce of
    the buffer (or a few contiguous buckets) instead of one collective per tensor -- the same arena idea as the
    lifter's (BASELINE configs[4]: phase5 on 8 GPUs, data parallel).

        flat = FlatGrads(model);  ...  flat.zero(); loss.backward(); flat.all_reduce_mean(); optimizer.step()
    Optimizers must not replace .grad (use zero_grad(set_to_none=False) or flat.zero())."""

    def __init__(self, module, group=None, bucket_bytes=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGrads: the module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGrads: parameters must share one device and dtype")
        sizes = [(p.numel() + 63) // 64 * 64 for p in self.params]          # 64-element aligned slots
        self.flat = torch.zeros(sum(sizes), dtype=dt, device=dev)
        self.group, self.bucket_bytes = group, bucket_bytes
        off = 0
        for p, n in zip(self.params, sizes):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def attached(self):
        """True while every p.grad still is its view of the flat buffer."""
        base = self.flat.data_ptr()
        end = base + self.flat.numel() * self.flat.element_size()
        return all(p.grad is not None and base <= p.grad.data_ptr() < end for p in self.params)

    def all_reduce_mean(self):
        """Sum over the ranks, divide by the world size; a no-op without an initialised process group."""
        if not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return
        if not self.attached():
            raise RuntimeError("FlatGrads: a .grad was replaced (zero_grad(set_to_none=True)?): gradients are not in the buffer")
        world = dist.get_world_size(self.group)
        if self.bucket_bytes:
            step = max(1, self.bucket_bytes // self.flat.element_size())
            works = [dist.all_reduce(self.flat[i:i + step], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                     for i in range(0, self.flat.numel(), step)]
            for w in works:
                w.wait()
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(world)


class SyncedOptimizer:
    """An optimizer facade for steps that drive SEVERAL models with one backward (train.cycle_step: Model_2D,
    Model_3D, lifter, projector -- BASELINE configs[4], phase5 data parallel): zero_grad() / step() as the step calls
    them, with the gradient average in between.

      * a FlatAdamW (lifter / projector): ONE sum all-reduce of the model's flat gradient arena, the 1/world average
        folded into the AdamW launch (grad_scale) -- GradSync, no overlap (two lifter calls per graph contribute first);
      * an arena.FlatAdam (the conv models): the same on the module's flat gradient arena;
      * any torch optimizer over a module with ordinary parameters: the module's gradients live in one FlatGrads
        buffer, zero_grad() zeroes it in place, step() all-reduces it once and steps.
    Without an initialised process group (or world 1) it is the optimizer itself."""

    def __init__(self, optimizer, module, group=None, bucket_bytes=None):
        from .arena import FlatAdam
        from .optim import FlatAdamW
        self.optimizer, self.module, self.group = optimizer, module, group
        self.param_groups = optimizer.param_groups
        self._flat_adamw = isinstance(optimizer, FlatAdamW)
        self._flat_adam = isinstance(optimizer, FlatAdam)
        self._sync = GradSync(group, bucket_bytes, overlap=False) if (self._flat_adamw or self._flat_adam) else None
        self._flat = None if (self._flat_adamw or self._flat_adam) else FlatGrads(module, group, bucket_bytes)

    def zero_grad(self, set_to_none=False):
        if self._flat_adamw or self._flat_adam:
            self.optimizer.zero_grad()
        else:
            if not self._flat.attached():                 # (someone set the grads to None: re-attach the views)
                self._flat = FlatGrads(self.module, self.group, self._flat.bucket_bytes)
            self._flat.zero()

    def step(self):
        if self._flat_adamw:
            return self.optimizer.step(grad_scale=self._sync(self.module))
        if self._flat_adam:
            arena = self.optimizer.arena
            arena.gather_grads()                          # (a gradient an op outside conv.py handed to autograd)
            return self.optimizer.step(grad_scale=self._sync.reduce_flat(arena.grad))
        self._flat.all_reduce_mean()
        return self.optimizer.step()

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, sd):
        return self.optimizer.load_state_dict(sd)
