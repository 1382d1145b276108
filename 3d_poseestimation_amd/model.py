"""LinearModel: drop-in for the reference's 2D->3D lifter, computed by libposelift.so.

Same constructor, forward contract, .train()/.eval() behaviour and state_dict() keys as
/root/reference/phase1_lifting/baselineModel.py:50-102 (residual block :14-47), so
`train_1.py`-style code can swap the import and keep running -- on an MI355X.  The
containers are stock nn.Linear / nn.BatchNorm1d objects (so `torch.manual_seed(s)` gives
the reference's initial weights, and checkpoints load unchanged), but they are parameter
holders only: their tensors are views into flat arenas and every FLOP of forward and
backward runs in the HIP library.  There is no CPU or eager fallback.
"""
import ctypes
import os
import weakref

import torch
from torch import nn

from . import _lib
from .layout import bitmap_words_per_row, hidden_layer_prefixes, param_slots


class Linear(nn.Module):
    """Residual block (baselineModel.py:14-47).  A container: LinearModel.forward does the math."""

    def __init__(self, linear_size, p_dropout=0.5, BN=True):
        super().__init__()
        self.l_size = linear_size
        self.w1 = nn.Linear(linear_size, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.w2 = nn.Linear(linear_size, linear_size)
        self.batch_norm2 = nn.BatchNorm1d(linear_size)
        self.BN = BN

    def forward(self, x):
        raise _lib.PoseliftError("residual blocks are evaluated by LinearModel.forward as one fused path")


class _LifterEvalFn(torch.autograd.Function):
    """model.eval() inside an autograd graph (phase5_loop/train_5.py:120 runs the lifter in eval mode with gradients
    flowing through it into Model_2D): pl_lifter_fwd_eval_saved / pl_lifter_bwd_eval -- BatchNorm on the running
    statistics, Dropout the identity, dx and every parameter gradient as torch computes them in eval mode."""

    @staticmethod
    def forward(ctx, x2, model, ws, *params):
        B = x2.shape[0]
        y = torch.empty(B, model.output_size, dtype=torch.float32, device=x2.device)
        try:
            model._mark_wplanes()
            _lib.check(_lib.lib().pl_lifter_fwd_eval_saved(
                ctypes.byref(model._desc), x2.data_ptr(), y.data_ptr(), B, ws["buf"].data_ptr(), ws["bytes"],
                _lib.current_stream_ptr()), "pl_lifter_fwd_eval_saved")
        except BaseException:
            model._release_workspace(ws, ws["busy"])
            raise
        ctx.model, ctx.ws, ctx.token = model, ws, ws["busy"]
        ctx.ticket = _GraphTicket(model)
        weakref.finalize(ctx, LinearModel._release_workspace, ws, ws["busy"])
        weakref.finalize(ctx, ctx.ticket.close)
        ctx.save_for_backward(x2)
        ctx.need_dx = x2.requires_grad
        model.last_workspace = ws
        return y

    @staticmethod
    def backward(ctx, gy):
        (x2,) = ctx.saved_tensors
        model, ws = ctx.model, ctx.ws
        if not ctx.ticket.open:
            raise _lib.PoseliftError("second backward through the same LinearModel forward (retain_graph): the saved "
                                     "activations were released after the first one -- run the forward again")
        try:
            dx = model._run_bwd(x2, gy.contiguous(), ws, ctx.need_dx, last_graph=False, eval_mode=True)
        finally:
            ctx.ticket.close()
            model._release_workspace(ws, ctx.token)
        return (dx, None, None) + (None,) * len(model._param_list)


def _overlap_ok(sync):
    """dp.GradSync.overlap_enabled(); any other object with launch_bucket()/world() overlaps unconditionally."""
    fn = getattr(sync, "overlap_enabled", None)
    return True if fn is None else bool(fn())


class _GraphTicket:
    """One training forward whose backward has not run yet.  LinearModel counts the live ones: the
    data-parallel overlap launches its all-reduce buckets only from the backward of the LAST live graph
    (two lifter calls in one graph -- the phase5 cycle -- must both have contributed first)."""

    def __init__(self, model):
        self.model, self.open = weakref.ref(model), True
        model._live_graphs += 1

    def close(self):
        if self.open:
            self.open = False
            m = self.model()
            if m is not None:
                m._live_graphs -= 1


class _LifterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, model, ws, *params):
        try:
            y = model._run_fwd_train(x2, ws)
        except BaseException:
            model._release_workspace(ws, ws["busy"])     # a failed library call must not leak the workspace
            raise
        ctx.model, ctx.ws, ctx.token = model, ws, ws["busy"]
        ctx.ticket = _GraphTicket(model)
        weakref.finalize(ctx, LinearModel._release_workspace, ws, ws["busy"])
        weakref.finalize(ctx, ctx.ticket.close)          # a graph dropped without backward
        ctx.save_for_backward(x2)
        ctx.need_dx = x2.requires_grad
        return y

    @staticmethod
    def backward(ctx, gy):
        (x2,) = ctx.saved_tensors
        model, ws = ctx.model, ctx.ws
        if not ctx.ticket.open:
            # the workspace was handed back after the first backward and may hold a later forward by now
            raise _lib.PoseliftError("second backward through the same LinearModel forward (retain_graph): the saved "
                                     "activations were released after the first one -- run the forward again")
        last = model._live_graphs == 1
        try:
            dx = model._run_bwd(x2, gy.contiguous(), ws, ctx.need_dx, last_graph=last)
        finally:
            ctx.ticket.close()
            model._release_workspace(ws, ctx.token)
        # parameter gradients were written straight into the flat gradient arena and
        # attached as .grad views (one arena = one all-reduce, one fused AdamW launch)
        return (dx, None, None) + (None,) * len(model._param_list)


class LinearModel(nn.Module):
    """compute_dtype: "fp32" (exact fp32 MFMA), "bf16x6" and "f16x3" (fp32-grade: meet the 1e-3 mm gate), "bf16" (bf16 operand
    storage, ~1 mm).  Range contract of "f16x3": the 1024-wide layers' activations are stored as fp16 planes at scale 1, so a
    hidden activation must stay below 65504 in magnitude (the conv path stores at 1/64: 4.2e6).  A training-mode BatchNorm
    output is at most gamma * sqrt(B) + |beta| and the block adds two of them, so only weights in the thousands can get there;
    nothing checks it at run time (an overflowing value becomes inf where "fp32" / "bf16x6" stay finite) -- use one of those
    for models outside the contract.  Gradients have no such limit: dz is range-scaled on the device per layer and step."""

    def __init__(self, i_dim, o_dim, linear_size=1024, num_stage=2, p_dropout=0.5, BN=True,
                 compute_dtype="fp32"):
        super().__init__()
        if linear_size % 4:
            raise ValueError("linear_size must be a multiple of 4 for the HIP path")
        self.linear_size, self.p_dropout, self.num_stage = linear_size, p_dropout, num_stage
        self.input_size, self.output_size = i_dim, o_dim
        self.w1 = nn.Linear(i_dim, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.linear_stages = nn.ModuleList(Linear(linear_size, p_dropout, BN) for _ in range(num_stage))
        self.w2 = nn.Linear(linear_size, o_dim)
        self.BN = BN
        self.compute_dtype = {"fp32": _lib.PL_F32, "bf16": _lib.PL_BF16, "bf16x6": _lib.PL_BF16X6,
                              "f16x3": _lib.PL_F16X3}[compute_dtype]
        self._slots, self._arena_floats = param_slots(i_dim, linear_size, o_dim, num_stage)
        self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self._step = 0
        self._inject_keep = None
        self._grad_sync = None
        self._dp_cuts = None
        self._live_graphs = 0
        self._sync_struct = self._gather_cb = self._sync_group = self._active_ws = self._cb_error = None
        self._flat = self._flat_grad = self._flat_grad_tmp = None
        self._ws_pool, self._ws_token = {}, 0
        self._flatten()

    # ------------------------------------------------------------------ arenas
    def _named_holders(self):
        mods = dict(self.named_modules())
        return [(mods[lin], mods[bn]) for lin, bn in hidden_layer_prefixes(self.num_stage)]

    def _flatten(self):
        """(Re)build the flat arenas on the parameters' current device and re-point every
        parameter / BatchNorm buffer at its slot."""
        named = dict(self.named_parameters())
        dev = self.w1.weight.device
        flat = torch.zeros(self._arena_floats, dtype=torch.float32, device=dev)
        for s in self._slots:
            p = named[s.name]
            view = flat[s.offset:s.offset + s.numel].view(s.shape)
            view.copy_(p.data.to(torch.float32))
            p.data = view
            p.grad = None
        holders = self._named_holders()
        L, H = len(holders), self.linear_size
        running = torch.zeros(L, 2, H, dtype=torch.float32, device=dev)
        batches = torch.zeros(L, dtype=torch.int64, device=dev)
        for l, (_, bn) in enumerate(holders):
            running[l, 0].copy_(bn.running_mean)
            running[l, 1].copy_(bn.running_var)
            batches[l] = bn.num_batches_tracked
            bn.running_mean.data = running[l, 0]
            bn.running_var.data = running[l, 1]
            bn.num_batches_tracked.data = batches[l]
        self._flat, self._bn_running, self._bn_batches = flat, running, batches
        self._flat_grad = self._flat_grad_tmp = None
        self._param_list = [named[s.name] for s in self._slots]
        self._ws_pool = {}
        self._desc = _lib.PLDesc(
            in_dim=self.input_size, hidden=H, out_dim=self.output_size, num_stage=self.num_stage,
            bn=int(bool(self.BN)), dtype=self.compute_dtype, p_dropout=float(self.p_dropout),
            bn_eps=float(self.batch_norm1.eps), bn_momentum=float(self.batch_norm1.momentum), reserved=0,
            params=flat.data_ptr(), bn_running=running.data_ptr(), bn_batches=batches.data_ptr())
        if self._sync_struct is not None:
            self._desc.sync = ctypes.pointer(self._sync_struct)
        # GEMM operand planes of the 1024-wide weights (PL_F16X3 / PL_BF16), kept across calls: FlatAdamW refreshes them
        # while it updates the parameters, so the forward has no weight-split pass.  _wplanes_ver = _planes_key() of the
        # parameters they were derived from (any in-place torch op on a parameter changes the key: then they are stale
        # and the next forward refreshes them first).
        self._wplanes, self._wplanes_ver = None, None
        nbytes = 0
        if flat.is_cuda and os.environ.get("POSELIFT_NO_WPLANES") != "1":     # (=1: same-box A/B of the per-call split)
            try:
                nbytes = _lib.lib().pl_wplanes_bytes(ctypes.byref(self._desc))
            except _lib.PoseliftError:
                nbytes = 0
        if nbytes:
            self._wplanes = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._desc.wplanes = self._wplanes.data_ptr()

    def _planes_key(self):
        """What the persistent weight planes were derived from.  The parameters are attached to the arena with
        `p.data = view`, so every Parameter keeps its OWN version counter: an in-place torch op on a parameter
        (load_state_dict, nn.init.*, torch.optim.*.step) bumps p._version, never the arena's -- the key therefore holds
        every parameter's counter next to the arena's (and the arena's address: .to() / re-flattening)."""
        return (self._flat.data_ptr(), self._flat._version) + tuple(p._version for p in self._param_list)

    def _mark_wplanes(self):
        """Before a library call that runs a forward: make the persistent weight planes current (an explicit
        pl_wplanes_refresh when any parameter changed since they were written) and say so.  The library itself only
        re-splits on calls that take the planes path (whole 128-row tiles, local statistics), so "the call will
        refresh them" is not something the host can assume for an arbitrary batch."""
        if self._wplanes is not None:
            self._ensure_wplanes()
            self._desc.wplanes_valid = 1

    def _ensure_wplanes(self):
        if self._wplanes is None:
            return
        key = self._planes_key()
        if self._wplanes_ver != key:
            _lib.check(_lib.lib().pl_wplanes_refresh(ctypes.byref(self._desc), _lib.current_stream_ptr()), "pl_wplanes_refresh")
            self._wplanes_ver = key

    def adamw_plane_segments(self):
        """PLAdamWPlanes for pl_adamw_flat_planes, or None: where the weight planes of each 1024-wide Linear go."""
        if self._wplanes is None:
            return None
        L = _lib.lib()
        per = L.pl_wplanes_layer_bytes(ctypes.byref(self._desc))
        H = self.linear_size
        kind = 2 if self.compute_dtype == _lib.PL_F16X3 else 1
        pl = _lib.PLAdamWPlanes(nseg=0, kind=kind, scale=L.pl_weight_plane_scale(), reserved=0)
        n_hidden = len(self._named_holders())
        if n_hidden - 1 > _lib.ADAMW_MAX_SEGS:
            return None
        for l in range(1, n_hidden):
            s = self._slots[4 * l]
            base = self._wplanes.data_ptr() + (l - 1) * per
            pl.seg[l - 1] = _lib.PLAdamWSeg(offset=s.offset, numel=s.numel, h=base, l=base + 2 * H * H if kind == 2 else None)
        pl.nseg = n_hidden - 1
        return pl

    # ctypes descriptors and device workspaces are rebuilt, not copied (copy.deepcopy / torch.save(model))
    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("_desc", "_ws_pool", "_flat_grad", "_flat_grad_tmp", "_grad_sync", "last_workspace", "_inject_keep",
                  "_wplanes", "_wplanes_ver",
                  "_sync_struct", "_gather_cb", "_sync_group", "_active_ws", "_cb_error"):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._grad_sync = self._inject_keep = self._dp_cuts = None
        self._live_graphs = 0
        self._sync_struct = self._gather_cb = self._sync_group = self._active_ws = self._cb_error = None
        self._ws_pool, self._ws_token = {}, 0
        self._flatten()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def _arenas_intact(self):
        s = self._slots[-1]
        return self._param_list[-1].data_ptr() == self._flat.data_ptr() + 4 * s.offset

    @property
    def flat_params(self):
        """The flat fp32 parameter arena (22 tensors + alignment padding)."""
        return self._flat

    @property
    def flat_grads(self):
        """The flat gradient arena (same layout); allocated on first use."""
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self._flat)
        return self._flat_grad

    def manual_seed(self, seed, step=0):
        """Key of the Philox dropout stream (csrc/philox.h); give each DP rank its own."""
        self._seed, self._step = int(seed) & 0xFFFFFFFFFFFFFFFF, int(step)
        return self

    def set_grad_sync(self, sync, cuts=None):
        """Attach a dp.GradSync: backward then all-reduces the upper layers' gradients while the
        lower layers are still being computed (see pl_lifter_bwd_layers).

        cuts: descending hidden-layer indices at which backward is cut; each cut c closes a bucket
        = the gradient arena from layer c's first tensor up to the previous cut.  Default: the
        output layer with the two top hidden layers, then one bucket per layer -- every bucket
        except the last (the 34-wide first layer, 150 KB) has a full layer of backward compute
        behind which to hide."""
        self._grad_sync = sync
        self._dp_cuts = self._check_cuts(cuts)

    # ------------------------------------------------------------------ SyncBN
    def set_sync_bn(self, enabled=True, group=None):
        """Training-mode BatchNorm over the GLOBAL batch of a data-parallel job (every rank must feed
        the same number of rows): the per-layer partial statistics are all-gathered (one small
        collective per hidden layer in forward, one in backward -- PLSync in include/poselift.h).
        The result does not depend on how the global batch is cut into ranks.  Off by default: the
        reference is single-process, and plain DP (statistics per shard) is the DDP convention."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if not enabled or world == 1:
            return self._enable_sync(1, 0, None)
        from . import dp
        return self._enable_sync(world, dist.get_rank(group), lambda slabs: dp.all_gather_slabs(slabs, group))

    def _enable_sync(self, world, rank, gather_impl):
        """gather_impl(slabs [world][n] device tensor, slab `rank` filled) fills the other slabs."""
        if world == 1:
            self._sync_struct = self._gather_cb = self._sync_group = None
            self._desc.sync = None
        else:
            self._sync_group = gather_impl
            self._gather_cb = _lib.GATHER_FN(self._on_gather)
            self._sync_struct = _lib.PLSync(world, rank, self._gather_cb, None)
            self._desc.sync = ctypes.pointer(self._sync_struct)
        self._ws_pool = {}                    # the workspace holds world-sized gather buffers
        return self

    @property
    def sync_bn(self):
        return self._sync_struct is not None

    def _on_gather(self, _user, buf, floats_per_rank, _stream):
        """PLGatherFn: called by the library between two of its own launches.  Exceptions cannot
        cross the C frames: they are parked and re-raised by _guarded()."""
        try:
            ws, world = self._active_ws, self._sync_struct.world
            off, nbytes = buf - ws["buf"].data_ptr(), world * floats_per_rank * 4
            if off < 0 or off + nbytes > ws["bytes"]:
                raise _lib.PoseliftError("gather buffer outside the active workspace")
            self._sync_group(ws["buf"][off:off + nbytes].view(torch.float32).view(world, floats_per_rank))
            return 0
        except BaseException as e:            # noqa: BLE001 -- re-raised on the Python side of the call
            self._cb_error = e
            return 1

    def _guarded(self, ws, rc_fn, what):
        """Run one library call that may call back into _on_gather."""
        self._active_ws, self._cb_error = ws, None
        try:
            rc = rc_fn()
        finally:
            self._active_ws = None
        if self._cb_error is not None:
            err, self._cb_error = self._cb_error, None
            raise err
        _lib.check(rc, what)

    def _check_cuts(self, cuts):
        n = len(self._named_holders())
        if cuts is None:
            cuts = list(range(max(n - 2, 0), -1, -1))
        cuts = [int(c) for c in cuts]
        if cuts != sorted(set(cuts), reverse=True) or (cuts and (cuts[0] > n or cuts[-1] < 0)):
            raise ValueError(f"cuts must be strictly descending layer indices within 0..{n}: {cuts}")
        if not cuts or cuts[-1] != 0:
            cuts.append(0)
        return cuts

    def _bwd_ranges(self):
        """[(hi, lo, arena_lo, arena_hi)]: layer ranges of a cut backward and the arena slice that
        becomes final with each."""
        n = len(self._named_holders())
        out, hi, end = [], n, self._flat.numel()
        for c in (self._dp_cuts or self._check_cuts(None)):
            off = self._slots[4 * c].offset
            out.append((hi, c, off, end))
            hi, end = c - 1, off
        return out

    def debug_inject_keep(self, keep_bitmaps):
        """Parity mode: the next training forward takes its dropout keep decisions from
        `keep_bitmaps` (int64/uint64 tensor [n_hidden][B][words], layout.pack_keep_bitmap)."""
        self._inject_keep = keep_bitmaps

    # ------------------------------------------------------------------ workspaces
    def _acquire_workspace(self, B):
        pool = self._ws_pool.setdefault(B, [])
        self._ws_token += 1
        for ws in pool:
            if ws["busy"] is None:
                ws["busy"] = self._ws_token
                return ws
        nbytes = _lib.lib().pl_workspace_bytes(ctypes.byref(self._desc), B)
        if nbytes == 0:
            _lib.check(-1, "pl_workspace_bytes")
        ws = {"buf": torch.empty(nbytes, dtype=torch.uint8, device=self._flat.device), "bytes": nbytes,
              "busy": self._ws_token, "B": B}
        pool.append(ws)
        return ws

    @staticmethod
    def _release_workspace(ws, token=None):
        """Free a workspace for reuse.  With a token, only if that acquisition still owns it
        (a graph freed without backward releases through a finalizer, possibly late)."""
        if token is None or ws["busy"] == token:
            ws["busy"] = None

    def workspace_view(self, ws, which, layer):
        """Debug/test: a saved tensor of the last training forward (pl_workspace_view)."""
        off, size = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(_lib.lib().pl_workspace_view(ctypes.byref(self._desc), ws["B"], which, layer,
                                                ctypes.byref(off), ctypes.byref(size)), "pl_workspace_view")
        raw = ws["buf"][off.value:off.value + size.value]
        H = self.linear_size
        if which in (0, 1):
            return raw.view(torch.float32).view(ws["B"], H)
        if which == 2:
            # always handed out in the row format (layout.unpack_bitmap); small batches keep a tile format on the device
            fmt = _lib.lib().pl_workspace_bitmap_format(ctypes.byref(self._desc), ws["B"], layer)
            _lib.check(min(fmt, 0), "pl_workspace_bitmap_format")
            words = raw.view(torch.int64)
            if fmt == 1:
                from .layout import pack_keep_bitmap, unpack_bitmap_tile
                on = unpack_bitmap_tile(words[:H].cpu().numpy().view("uint64"), ws["B"], H)
                return torch.from_numpy(pack_keep_bitmap(on).view("int64")).to(raw.device)
            return words[:ws["B"] * bitmap_words_per_row(H)].view(ws["B"], bitmap_words_per_row(H))
        return raw.view(torch.float32)

    # ------------------------------------------------------------------ launches
    def _run_fwd_train(self, x2, ws):
        B = x2.shape[0]
        y = torch.empty(B, self.output_size, dtype=torch.float32, device=x2.device)
        inj = self._inject_keep
        self._inject_keep = None
        if inj is not None:
            _lib.require_device_tensor(inj, "inject_keep", inj.dtype)
            want = (len(self._named_holders()), B, bitmap_words_per_row(self.linear_size))
            if tuple(inj.shape) != want or inj.element_size() != 8:
                raise _lib.PoseliftError(f"inject_keep must be 64-bit words of shape {want}")
        self._step += 1
        self._mark_wplanes()
        self._guarded(ws, lambda: _lib.lib().pl_lifter_fwd_train(
            ctypes.byref(self._desc), x2.data_ptr(), y.data_ptr(), B, ws["buf"].data_ptr(), ws["bytes"],
            self._seed, self._step, inj.data_ptr() if inj is not None else None,
            _lib.current_stream_ptr()), "pl_lifter_fwd_train")
        self.last_workspace = ws
        return y

    def _run_bwd(self, x2, gy, ws, need_dx, last_graph=True, eval_mode=False):
        B = x2.shape[0]
        accumulate = any(p.grad is not None for p in self._param_list)
        sync = self._grad_sync
        if accumulate and sync is not None and getattr(sync, "has_pending", lambda: False)():
            # an earlier backward of this optimizer step already all-reduced its buckets: adding local gradients
            # to summed ones would make the replicas diverge silently
            sync.abandon()
            raise _lib.PoseliftError(
                "gradient accumulation with overlapped all-reduce: wrap every backward but the last of an optimizer "
                "step in `with sync.no_sync():` (dp.GradSync), as with DistributedDataParallel")
        if accumulate:
            if self._flat_grad_tmp is None:
                self._flat_grad_tmp = torch.zeros_like(self._flat)
            target = self._flat_grad_tmp
        else:
            target = self.flat_grads
        dx = torch.empty_like(x2) if need_dx else None
        args = (ctypes.byref(self._desc), x2.data_ptr(), gy.data_ptr(), B, ws["buf"].data_ptr(), ws["bytes"],
                dx.data_ptr() if need_dx else None, target.data_ptr())
        if sync is not None and not accumulate and last_graph and _overlap_ok(sync) and sync.world() > 1:
            # data-parallel overlap: after each layer range the tail of the arena down to that
            # range's lowest layer is final and is all-reduced while the layers below compute
            for hi, lo, a_lo, a_hi in self._bwd_ranges():
                self._guarded(ws, lambda: _lib.lib().pl_lifter_bwd_layers(*args, hi, lo, _lib.current_stream_ptr()),
                              "pl_lifter_bwd_layers")
                sync.launch_bucket(target[a_lo:a_hi])
        elif eval_mode:
            self._guarded(ws, lambda: _lib.lib().pl_lifter_bwd_eval(*args, _lib.current_stream_ptr()), "pl_lifter_bwd_eval")
        else:
            self._guarded(ws, lambda: _lib.lib().pl_lifter_bwd(*args, _lib.current_stream_ptr()), "pl_lifter_bwd")
        if accumulate:
            for s, p in zip(self._slots, self._param_list):
                gview = target[s.offset:s.offset + s.numel].view(s.shape)
                if p.grad is None:
                    p.grad = gview.clone()
                else:
                    p.grad.add_(gview)
        else:
            for s, p in zip(self._slots, self._param_list):
                if self.BN or "batch_norm" not in s.name:
                    p.grad = target[s.offset:s.offset + s.numel].view(s.shape)
        return dx

    # ------------------------------------------------------------------ fused train step
    def step_carries_adamw(self, B):
        """Does a fused step of B rows carry the AdamW update inside its backward launches (pl_lifter_train_step: batches of
        <= 64 rows whose hidden layers run as one launch each)?"""
        return _lib.lib().pl_lifter_step_carries_adamw(ctypes.byref(self._desc), int(B)) == 1

    def fused_train_fwd_bwd(self, x2, target, sync=None, step_dev=None, adamw=None):
        """forward + MSE(mean) + backward in one library call (two when a data-parallel sync wants
        the upper layers' gradients early).  Returns (loss, y) device tensors; parameter gradients
        land in the flat arena and stay attached as .grad views.
        step_dev (graph capture, train.GraphedTrainStep): device counter of completed steps; the dropout stream
        of the captured launches uses (the step number at capture) + *step_dev.
        adamw (a _lib.PLAdamWStep from FlatAdamW._step_struct; no sync): the optimizer step as well, in the same call
        (pl_lifter_train_step) -- the parameters are updated in place."""
        B = x2.shape[0]
        ws = self._acquire_workspace(B)
        try:
            y = torch.empty(B, self.output_size, dtype=torch.float32, device=x2.device)
            loss = torch.empty((), dtype=torch.float32, device=x2.device)
            grads = self.flat_grads
            self._step += 1
            L = _lib.lib()
            self._desc.step_dev = step_dev.data_ptr() if step_dev is not None else None
            if adamw is not None and self.step_carries_adamw(B):
                # (a batch of <= 64 rows never takes the operand-planes path: no refresh before it; the update below leaves
                #  the persistent planes stale and the next call that wants them refreshes them)
                self._desc.wplanes_valid = 0
            else:
                self._mark_wplanes()

            def call(hi, lo):
                self._guarded(ws, lambda: L.pl_lifter_train_fwd_bwd(
                    ctypes.byref(self._desc), x2.data_ptr(), target.data_ptr(), B, ws["buf"].data_ptr(), ws["bytes"],
                    self._seed, self._step, y.data_ptr(), loss.data_ptr(), grads.data_ptr(), hi, lo,
                    _lib.current_stream_ptr()), "pl_lifter_train_fwd_bwd")
            if adamw is not None:
                if sync is not None:
                    raise _lib.PoseliftError("fused_train_fwd_bwd: the in-call AdamW step and a gradient sync exclude each other")
                self._guarded(ws, lambda: L.pl_lifter_train_step(
                    ctypes.byref(self._desc), x2.data_ptr(), target.data_ptr(), B, ws["buf"].data_ptr(), ws["bytes"],
                    self._seed, self._step, y.data_ptr(), loss.data_ptr(), grads.data_ptr(), ctypes.byref(adamw),
                    _lib.current_stream_ptr()), "pl_lifter_train_step")
                self._wplanes_ver = None          # the parameters changed under the persistent weight planes
            elif sync is not None and sync.world() > 1 and _overlap_ok(sync):
                for hi, lo, a_lo, a_hi in self._bwd_ranges():
                    call(hi, lo)
                    sync.launch_bucket(grads[a_lo:a_hi])
            else:
                call(len(self._named_holders()), 0)
        finally:
            self._desc.step_dev = None
            self._release_workspace(ws)
        if self._param_list[-1].grad is None or self._param_list[-1].grad.data_ptr() != \
                grads.data_ptr() + 4 * self._slots[-1].offset:
            for s, p in zip(self._slots, self._param_list):
                p.grad = grads[s.offset:s.offset + s.numel].view(s.shape) if (self.BN or "batch_norm" not in s.name) else None
        self.last_workspace = ws
        return loss, y

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not self._arenas_intact():
            self._flatten()
        B = x.shape[0]
        x2 = x.reshape(B, -1)
        if x2.shape[1] != self.input_size:
            raise ValueError(f"expected {self.input_size} input features, got {x2.shape[1]}")
        if x2.dtype != torch.float32:
            x2 = x2.float()
        x2 = x2.contiguous()
        _lib.require_device_tensor(x2, "x")
        _lib.require_device_tensor(self._flat, "parameters")
        if x2.device != self._flat.device:
            raise _lib.PoseliftError(f"x on {x2.device}, parameters on {self._flat.device}")
        with _lib.on_device(x2.device):
            if self.training:
                ws = self._acquire_workspace(B)
                needs_graph = torch.is_grad_enabled() and (
                    x2.requires_grad or any(p.requires_grad for p in self._param_list))
                if needs_graph:
                    return _LifterFn.apply(x2, self, ws, *self._param_list)
                try:
                    return self._run_fwd_train(x2, ws)
                finally:
                    self._release_workspace(ws)
            ws = self._acquire_workspace(B)
            if torch.is_grad_enabled() and (x2.requires_grad or any(p.requires_grad for p in self._param_list)):
                return _LifterEvalFn.apply(x2, self, ws, *self._param_list)
            try:
                y = torch.empty(B, self.output_size, dtype=torch.float32, device=x2.device)
                self._mark_wplanes()
                rc = _lib.lib().pl_lifter_fwd_eval(
                    ctypes.byref(self._desc), x2.data_ptr(), y.data_ptr(), B, ws["buf"].data_ptr(),
                    ws["bytes"], _lib.current_stream_ptr())
                _lib.check(rc, "pl_lifter_fwd_eval")
            finally:
                self._release_workspace(ws)
            return y


def weight_init(m):
    """baselineModel.py:10-12: Kaiming-normal on every nn.Linear weight (biases untouched)."""
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight)
