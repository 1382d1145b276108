"""GPU tests of the data-parallel step.  A 1-GPU box cannot run RCCL across devices, so
 (a) the two-part backward used for overlap is checked bit for bit against the one-call backward, and
 (b) a 2-rank job is rehearsed with both ranks on cuda:0 over gloo (POSELIFT_DIST_BACKEND=gloo): the
     overlapped, bucketed all-reduce must give exactly the parameters of the plain one, every rank must
     hold the same parameters, and two ranks on half batches (BatchNorm off) must match one rank on the
     whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


class _FakeSync:
    def __init__(self):
        self.buckets = []

    def world(self):
        return 2

    def launch_bucket(self, t):
        self.buckets.append((t.data_ptr(), t.numel()))


@pytest.mark.parametrize("cuts", [None, [3], [4, 2], [5, 4, 3, 2, 1, 0], [0]])
def test_cut_backward_is_bitwise_the_one_call_backward(cuts):
    import __graft_entry__ as ge
    pkg = ge.build()
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.5).to("cuda:0").train()
    x, y = pkg.synth.synthetic_batch(384, 3, "cuda:0")
    grads = []
    for sync in (None, _FakeSync()):
        m.set_grad_sync(sync, cuts)
        m.zero_grad(set_to_none=True)
        m.manual_seed(4, step=0)
        xr = x.clone().requires_grad_(True)
        pkg.mse_loss(m(xr).reshape(y.shape), y).backward()
        grads.append((m.flat_grads.clone(), xr.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    # the buckets tile the arena from its end (output layer) down to its start, in launch order
    base, end = m.flat_grads.data_ptr(), m.flat_grads.numel()
    want = [c for c in (cuts if cuts is not None else [3, 2, 1, 0])]
    if want[-1] != 0:
        want.append(0)
    assert len(sync.buckets) == len(want)
    for (p, n), c in zip(sync.buckets, want):
        off = m._slots[4 * c].offset
        assert p == base + 4 * off and n == end - off
        end = off
    assert end == 0
    # the fused step cut the same way gives the same gradients as the uncut fused step
    fused = []
    for sync in (None, _FakeSync()):
        m.set_grad_sync(sync, cuts)
        m.manual_seed(4, step=0)
        m.fused_train_fwd_bwd(x.reshape(384, -1), y.reshape(384, -1), sync)
        fused.append(m.flat_grads.clone())
    assert torch.equal(fused[0], fused[1]) and torch.equal(fused[0], grads[0][0])
    with pytest.raises(ValueError):
        m.set_grad_sync(None, [2, 3])
    with pytest.raises(ValueError):
        m.set_grad_sync(None, [7])


@pytest.mark.parametrize("cuts", [[3], [4, 2], [5, 4, 3, 2, 1, 0], [0]])
def test_cut_backward_of_a_small_batch(cuts):
    """B <= 64: a hidden layer's dX GEMM and the BatchNorm backward of the layer below share a launch (small_layer.hip)
    unless a cut separates them -- then the stand-alone kernels run.  Same gradients to round-off wherever the cuts are."""
    import __graft_entry__ as ge
    pkg = ge.build()
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.5).to("cuda:0").train()
    x, y = pkg.synth.synthetic_batch(48, 3, "cuda:0")
    grads = []
    for sync in (None, _FakeSync()):
        m.set_grad_sync(sync, cuts)
        m.zero_grad(set_to_none=True)
        m.manual_seed(4, step=0)
        xr = x.clone().requires_grad_(True)
        pkg.mse_loss(m(xr).reshape(y.shape), y).backward()
        grads.append((m.flat_grads.clone(), xr.grad.clone()))
    for a, b in zip(grads[0], grads[1]):
        assert (a - b).norm() <= 2e-5 * a.norm(), ((a - b).norm() / a.norm()).item()
    for s, p in zip(m._slots, m._param_list):          # per tensor, the pre-BatchNorm biases (true gradient 0) aside
        a, b = grads[0][0][s.offset:s.offset + s.numel], grads[1][0][s.offset:s.offset + s.numel]
        if a.norm() > 1e-6 * grads[0][0].norm():
            assert (a - b).norm() <= 5e-5 * a.norm(), (s.name, ((a - b).norm() / a.norm()).item())


@pytest.mark.parametrize("rank", [0, 1])
@pytest.mark.parametrize("dtype", ["fp32", "bf16x6", "f16x3", "bf16"])
def test_sync_bn_with_a_mirror_rank_is_the_doubled_batch(rank, dtype):
    """PLSync in one process: a fake 2-rank world whose other rank holds the same rows (the gather
    copies this rank's slab into the other).  Global-batch statistics of [x; x] -- the forward
    must be bitwise what one process computes on the concatenated batch (B % 128 == 0), running
    statistics included; gradients agree to round-off once the mirror's share is added.
    f16x3 / bf16 (round 3): the operand-planes kernels under SyncBN -- every rank's {max|dy|, max|zhat|} travel in the
    gathered slab, so the range scale of the fp16 dz planes is the one the doubled batch computes."""
    import __graft_entry__ as ge
    pkg = ge.build()
    B = 256
    x, y = pkg.synth.synthetic_batch(B, 8, "cuda:0")
    torch.manual_seed(0)
    ref = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.0, compute_dtype=dtype).to("cuda:0").train()
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.0, compute_dtype=dtype).to("cuda:0").train()
    calls = []

    def mirror(slabs):
        calls.append(tuple(slabs.shape))
        slabs[1 - rank].copy_(slabs[rank])
    m._enable_sync(2, rank, mirror)
    assert m.sync_bn

    xx, yy = torch.cat([x, x]), torch.cat([y, y])
    out_ref = ref(xx)
    pkg.mse_loss(out_ref.reshape(yy.shape), yy).backward()
    out = m(x)
    pkg.mse_loss(out.reshape(y.shape), y).backward()
    assert torch.equal(out, out_ref[:B]) and torch.equal(out, out_ref[B:])
    assert torch.equal(m._bn_running, ref._bn_running) and torch.equal(m._bn_batches, ref._bn_batches)
    assert len(calls) == 10 and all(c[0] == 2 for c in calls)          # 5 layers x (forward + backward)
    # DP averaging: (g_rank + g_mirror) / 2 = g_rank here; the full-batch loss is the mean over 2B rows
    for s in m._slots:
        if s.name.endswith(".bias") and "batch_norm" not in s.name and s.name != "w2.bias":
            continue                                                   # pre-BN biases: zero true gradient
        a = m.flat_grads[s.offset:s.offset + s.numel].double()
        b = ref.flat_grads[s.offset:s.offset + s.numel].double()
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 2e-5, s.name
    # the fused step takes the same route
    m.manual_seed(0, step=0)
    g_autograd = m.flat_grads.clone()
    m.fused_train_fwd_bwd(x.reshape(B, -1), y.reshape(B, -1), None)
    assert torch.equal(m.flat_grads, g_autograd)
    # an exception inside the gather surfaces as itself, not as a status code
    def broken(_):
        raise RuntimeError("link down")
    m._enable_sync(2, rank, broken)
    with pytest.raises(RuntimeError, match="link down"):
        m(x)
    m.set_sync_bn(False)
    assert not m.sync_bn
    m(x)                                                               # local statistics again


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), POSELIFT_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import importlib
    import torch.distributed as dist
    pkg = importlib.import_module("3d_poseestimation_amd")
    try:
        r, local, w = pkg.dp.init_from_env()
    except Exception as e:                      # the communicator could not be built HERE (driver / IPC / fabric): not this
        print(f"rank {rank}: {backend} did not initialise: {e!r}", flush=True)      # library's arithmetic -- the caller skips
        os._exit(77)
    dev = pkg.dp.local_device(local)
    torch.cuda.set_device(dev)
    B = 256
    xs, ys = pkg.synth.synthetic_batch(B, 21, dev)
    lo, hi = pkg.dp.shard_rows(B, rank, world)

    def run(overlap, bn, shard):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=128, p_dropout=0.0, BN=bn).to(dev).train()
        pkg.dp.broadcast_model(m)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        sync = pkg.dp.GradSync()
        m.set_grad_sync(sync if overlap else None)
        for _ in range(3):
            if shard:
                pkg.train_step(m, opt, xs[lo:hi], ys[lo:hi], grad_sync=sync)
            else:
                pkg.train_step(m, opt, xs, ys, grad_sync=sync)
        torch.cuda.synchronize()
        return m.flat_params.clone()

    a = run(True, True, True)
    b = run(False, True, True)
    assert torch.equal(a, b), "overlapped bucketed all-reduce changed the result"
    gathered = [torch.zeros_like(a) for _ in range(world)]
    dist.all_gather(gathered, a)
    assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
    # BatchNorm off: mean of the two half-batch gradients == full-batch gradient (MSE mean)
    c = run(True, False, True)
    d = run(True, False, False)          # every rank the whole batch: the average is the same gradient
    assert torch.allclose(c, d, rtol=1e-4, atol=1e-6), float((c - d).abs().max())
    # SyncBN (PLSync over a real all-gather): 2 x half batch with global statistics == one process
    # on the whole batch -- forward and running statistics bitwise (128-row shards), gradients to
    # round-off after the DP average
    torch.manual_seed(0)
    ref = pkg.LinearModel(34, 51, linear_size=128, p_dropout=0.0).to(dev).train()
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=128, p_dropout=0.0).to(dev).train().set_sync_bn(True)
    assert m.sync_bn
    out_ref = ref(xs)
    pkg.mse_loss(out_ref.reshape(ys.shape), ys).backward()
    out = m(xs[lo:hi])
    pkg.mse_loss(out.reshape(ys[lo:hi].shape), ys[lo:hi]).backward()
    assert torch.equal(out, out_ref[lo:hi]), "SyncBN forward differs from the single-process global batch"
    assert torch.equal(m._bn_running, ref._bn_running)
    gsum = m.flat_grads.clone()
    dist.all_reduce(gsum)
    gsum /= world
    for s in m._slots:
        if s.name.endswith(".bias") and "batch_norm" not in s.name and s.name != "w2.bias":
            continue
        a_, b_ = gsum[s.offset:s.offset + s.numel].double(), ref.flat_grads[s.offset:s.offset + s.numel].double()
        assert float((a_ - b_).norm() / (b_.norm() + 1e-30)) < 2e-5, s.name
    # and through train_step with the overlapped gradient all-reduce
    opt = pkg.FlatAdamW(m, lr=1e-3)
    sync = pkg.dp.GradSync()
    m.set_grad_sync(sync)
    pkg.train_step(m, opt, xs[lo:hi], ys[lo:hi], grad_sync=sync)
    torch.cuda.synchronize()
    gathered = [torch.zeros_like(m.flat_params) for _ in range(world)]
    dist.all_gather(gathered, m.flat_params)
    assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged under SyncBN"

    # ---- more than one backward per optimizer step, with the overlapped all-reduce attached -------------
    def same_on_all_ranks(t, what):
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        assert all(torch.equal(got[0], g) for g in got), what

    def accumulate(overlap):
        """two micro-batches: every backward but the last inside no_sync(), as with DistributedDataParallel"""
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=128, p_dropout=0.0).to(dev).train()
        pkg.dp.broadcast_model(m)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        sync = pkg.dp.GradSync(overlap=overlap)
        m.set_grad_sync(sync)
        half = (hi - lo) // 2
        opt.zero_grad()
        with sync.no_sync():
            pkg.mse_loss(m(xs[lo:lo + half]).reshape(-1, 17, 3), ys[lo:lo + half]).backward()
        pkg.mse_loss(m(xs[lo + half:hi]).reshape(-1, 17, 3), ys[lo + half:hi]).backward()
        opt.step(grad_scale=sync(m))
        torch.cuda.synchronize()
        return m.flat_params.clone()
    p_acc = accumulate(True)
    same_on_all_ranks(p_acc, "ranks diverged under gradient accumulation")
    assert torch.equal(p_acc, accumulate(False)), "accumulation with the overlap attached changed the result"

    # without no_sync the second backward would add local gradients to already summed ones: loud, not silent
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=128, p_dropout=0.0).to(dev).train()
    sync = pkg.dp.GradSync()
    m.set_grad_sync(sync)
    pkg.mse_loss(m(xs[lo:hi]).reshape(-1, 17, 3), ys[lo:hi]).backward()
    try:
        pkg.mse_loss(m(xs[lo:hi]).reshape(-1, 17, 3), ys[lo:hi]).backward()
        raise AssertionError("second backward with buckets in flight did not raise")
    except pkg.PoseliftError as e:
        assert "no_sync" in str(e)
    assert not sync.has_pending()

    def cycle(overlap):
        """the phase5 pattern: the lifter twice in ONE graph (train_5 copy.py:167-168, 184-185)"""
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=128, p_dropout=0.0).to(dev).train()
        pkg.dp.broadcast_model(m)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        sync = pkg.dp.GradSync(overlap=overlap)
        m.set_grad_sync(sync)
        opt.zero_grad()
        xa = xs[lo:hi].clone().requires_grad_(True)
        la = pkg.mse_loss(m(xa).reshape(-1, 17, 3), ys[lo:hi])
        lb = pkg.mse_loss(m(xs[lo:hi] * 0.9).reshape(-1, 17, 3), ys[lo:hi])
        (la + lb).backward()
        assert m._live_graphs == 0
        opt.step(grad_scale=sync(m))
        torch.cuda.synchronize()
        return m.flat_params.clone()
    p_cyc = cycle(True)
    same_on_all_ranks(p_cyc, "ranks diverged with two lifter calls in one graph")
    assert torch.equal(p_cyc, cycle(False))
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_two_rank_rccl_on_two_gpus():
    """The same worker over RCCL (backend "nccl"), one GPU per rank: overlapped buckets == plain all-reduce bit for
    bit, replicas identical, SyncBN == the concatenated batch, accumulation / two-calls-per-graph.  Needs two visible
    GPUs: skipped on the 1-GPU test box, runs wherever the driver has a multi-GPU node."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:                             # (a rank left waiting for a peer that gave up)
        if p.is_alive():
            p.terminate()
            p.join(30)
    if any(p.exitcode == 77 for p in procs):
        pytest.skip("RCCL could not build a two-rank communicator on this node")
    for p in procs:
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _rccl_one_rank_worker(port, q):
    """Every torch.distributed call the data-parallel step makes, on the REAL RCCL backend with a one-rank group (what a
    1-GPU box can run): communicator creation with device_id, async all-reduce of arena slices launched between the
    backward's layer ranges, wait(), in-place all_gather_into_tensor of the SyncBN slabs, broadcast, barrier."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import importlib
    import torch.distributed as dist
    pkg = importlib.import_module("3d_poseestimation_amd")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    except Exception as e:
        print(f"RCCL did not initialise: {e!r}", flush=True)
        os._exit(77)

    class PretendPeer(pkg.dp.GradSync):
        """reports a second rank so that backward takes the bucketed route; the sum over one rank is the identity"""
        launched = 0

        def world(self):
            return 2

        def launch_bucket(self, flat_slice):
            PretendPeer.launched += 1
            super().launch_bucket(flat_slice)

        def __call__(self, model):
            super().__call__(model)
            return 1.0

    xs, ys = pkg.synth.synthetic_batch(512, 21, dev)

    def run(sync):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=1024, p_dropout=0.5).to(dev).train()
        m.manual_seed(7)
        pkg.dp.broadcast_model(m)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        m.set_grad_sync(sync)
        for _ in range(3):
            pkg.train_step(m, opt, xs, ys, grad_sync=sync)
        # and the autograd route (backward launches the buckets from pl_lifter_bwd_layers)
        opt.zero_grad()
        pkg.mse_loss(m(xs).reshape(-1, 17, 3), ys).backward()
        opt.step(grad_scale=sync(m) if sync is not None else 1.0)
        torch.cuda.synchronize()
        return m.flat_params.clone()

    a = run(PretendPeer(overlap=True))
    assert PretendPeer.launched >= 8, PretendPeer.launched      # 4 steps x (at least) 2 buckets
    b = run(PretendPeer(overlap=False))
    c = run(None)
    assert torch.equal(a, c) and torch.equal(b, c), "RCCL buckets on arena slices changed the single-rank result"
    slabs = torch.arange(2048, dtype=torch.float32, device=dev).view(1, 2048).clone()
    keep = slabs.clone()
    pkg.dp.all_gather_slabs(slabs)                     # in place: input is slab `rank` of the output
    torch.cuda.synchronize()
    assert torch.equal(slabs, keep)
    t = torch.ones(18, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    assert float(t.sum()) == 18.0
    dist.destroy_process_group()
    q.put(0)


def test_one_rank_rccl_api_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    p.join(300)
    if p.exitcode == 77:
        pytest.skip("RCCL could not build a communicator on this node")
    assert p.exitcode == 0
    assert q.get(timeout=5) == 0


def test_two_rank_rehearsal_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
