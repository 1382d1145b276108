"""GPU tests of the operand-planes GEMM path (csrc/gemm_planes.h, gemm_planes.hip): PL_F16X3 -- fp32-grade products
from two fp16 planes per operand written by the producing kernels, three MFMAs per product term -- and the planes
GEMM itself in all three layouts.  The golden-vector / oracle gates of this mode are parametrisations of the tests in
test_gpu_parity.py: g1 eval forward, g2 'full' train step, the B = 1024 train step against the oracle, and (round 3)
test_full_size_train_then_eval_vs_oracle[f16x3] -- B = 4096, H = 1024, the bench's shape and arithmetic.  Here: the GEMM
against fp64, tensor scales, properties at BASELINE.json's full size (bitwise repeatability and a comparison with this
library's own fp32 path -- a self-comparison, not an oracle gate), the fallbacks, and the freshness of the persistent
weight planes."""
import numpy as np
import pytest
import torch

from oracle import lifter_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PL_BF16, PL_F16X3 = 1, 3


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    p = ge.build()
    assert torch.cuda.is_available()
    return p


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _planes_gemm(pkg, layout, mode, a, b, bias=None, sa=1.0, sb=1.0):
    """a [M][K], b [K][N] (numpy) -> C on the device through pl_gemm_planes."""
    M, K = a.shape
    N = b.shape[1]
    A = _t(a if layout != 2 else a.T)
    Bm = _t(b.T if layout == 0 else b)
    C = torch.full((M, N), float("nan"), device=DEV)
    L = pkg.lib()
    scratch = torch.empty(L.pl_gemm_planes_scratch_bytes(M, N, K), dtype=torch.uint8, device=DEV)
    bt = _t(bias) if bias is not None else None
    rc = L.pl_gemm_planes(layout, mode, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K,
                          bt.data_ptr() if bt is not None else None, sa, sb, scratch.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.pl_last_error()
    return C.cpu().numpy()


@pytest.mark.parametrize("mode,tol", [(PL_F16X3, 2e-6), (PL_BF16, 2e-2)])
@pytest.mark.parametrize("layout,M,N,K", [(0, 4096, 1024, 1024), (1, 4096, 1024, 1024), (2, 1024, 1024, 4096),
                                          (0, 128, 128, 32), (0, 256, 128, 96), (1, 128, 384, 64), (2, 128, 256, 512),
                                          (2, 256, 128, 160),
                                          # tiles that hang over the matrix edge (the conv path's 64-wide layers, ragged
                                          # pixel counts): clamped sources, guarded epilogue
                                          (0, 200, 192, 64), (0, 1000, 64, 256), (0, 1, 8, 32), (1, 130, 64, 128),
                                          (1, 300, 320, 96), (2, 64, 256, 512), (2, 200, 72, 256)])
def test_planes_gemm_vs_fp64(pkg, mode, tol, layout, M, N, K):
    """Every layout (k-contiguous rows through ds_read_b128, k-strided through the transposing LDS read), K from one
    tile to 128 tiles (prologue / steady state / tail of the DMA pipeline), with a bias in the epilogue.  f16x3 must
    stay inside the bound of an fp32 evaluation; bf16 shows bf16-sized error."""
    rng = np.random.default_rng(M + 3 * N + 7 * K + layout)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    got = _planes_gemm(pkg, layout, mode, a, b, bias, sa=1.0, sb=16.0)
    want = a.astype(np.float64) @ b.astype(np.float64) + bias
    bound = tol * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64) + np.abs(bias)) + 1e-7
    err = np.abs(got - want)
    assert np.all(err <= bound), float((err / bound).max())
    if mode == PL_BF16:
        assert err.max() > 1e-4                       # really one bf16 product


def test_planes_gemm_asymmetric_operands_catch_transposes(pkg):
    """A = I with an asymmetric B returns B itself: a swapped fragment or accumulator map cannot hide."""
    n = 256
    b = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 251) / 16.0
    eye = np.eye(n, dtype=np.float32)
    for layout in (0, 1, 2):
        got = _planes_gemm(pkg, layout, PL_F16X3, eye, b)
        np.testing.assert_array_equal(got, b)
        got = _planes_gemm(pkg, layout, PL_F16X3, b, eye)
        np.testing.assert_array_equal(got, b)


@pytest.mark.parametrize("mag,scale", [(3e-7, 2.0 ** 34), (5e-3, 2.0 ** 20), (300.0, 2.0 ** 5)])
def test_f16x3_tensor_scale_keeps_small_and_large_operands_fp32_grade(pkg, mag, scale):
    """Gradient-sized (1e-7) and large operands: with the power-of-two tensor scale the fp16 planes carry 22+ bits."""
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((256, 512)) * mag).astype(np.float32)
    b = (rng.standard_normal((512, 128)) * 0.03).astype(np.float32)
    for layout in (1, 2):
        got = _planes_gemm(pkg, layout, PL_F16X3, a, b, sa=scale, sb=16.0)
        want = a.astype(np.float64) @ b.astype(np.float64)
        bound = 2e-6 * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64))
        assert np.all(np.abs(got - want) <= bound)


def test_planes_gemm_rejects_what_it_cannot_tile(pkg):
    L = pkg.lib()
    one = torch.zeros(130 * 68, device=DEV)
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    for M, N, K in ((130, 64, 40), (130, 68, 32)):            # K not whole 32-k tiles; N not a multiple of 8
        rc = L.pl_gemm_planes(0, PL_F16X3, one.data_ptr(), one.data_ptr(), one.data_ptr(), M, N, K, None, 1.0, 1.0,
                              scratch.data_ptr(), None)
        assert rc != 0 and b"unsupported" in L.pl_last_error()
    assert L.pl_gemm_planes(0, 2, one.data_ptr(), one.data_ptr(), one.data_ptr(), 128, 128, 32, None, 1.0, 1.0,
                            scratch.data_ptr(), None) != 0                     # PL_BF16X6 has no planes form


@pytest.fixture(scope="module")
def full(pkg):
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=1024, num_stage=2, p_dropout=0.5, BN=True, compute_dtype="f16x3").to(DEV)
    x, y = pkg.synth.synthetic_batch(4096, 1234, DEV)
    return m, x, y


def test_f16x3_full_size_properties(pkg, full):
    """BASELINE.json's size (B = 4096, H = 1024) on the planes path: size-independent properties."""
    m, x, y = full
    m.eval()
    with torch.no_grad():
        y_all = m(x)
        y_head = m(x[:1024])
    # rows are independent and the contraction order does not depend on the batch: bit-exact
    assert torch.equal(y_all[:1024], y_head)
    # against the fp32 arithmetic of the same library on the same weights: both inside the gate of each other
    m32 = pkg.LinearModel(34, 51, linear_size=1024, p_dropout=0.5, compute_dtype="fp32").to(DEV).eval()
    m32.load_state_dict(m.state_dict())
    with torch.no_grad():
        assert orc.mpjpe_mm(y_all.cpu().numpy(), m32(x).cpu().numpy()) < 1e-3
    # the training step is bitwise reproducible for a fixed (seed, step)
    outs = []
    for _ in range(2):
        m.train().manual_seed(11, step=3)
        m.zero_grad(set_to_none=True)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        pred = m(x)
        pkg.mse_loss(pred.reshape(-1, 17, 3), y).backward()
        outs.append((pred.detach().clone(), m.flat_grads.clone()))
        m.load_state_dict(sd0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][1]).all()
    # BatchNorm really normalised over the batch
    ws = m.last_workspace
    z = m.workspace_view(ws, 0, 2).double()
    mean, rstd = m.workspace_view(ws, 3, 2).double(), m.workspace_view(ws, 4, 2).double()
    zhat = (z - mean) * rstd
    assert zhat.mean(0).abs().max() < 1e-5 and (zhat.var(0, unbiased=False) - 1).abs().max() < 1e-3
    # odd hidden layers feed GEMMs only: they exist as fp16 planes, the fp32 view says so
    with pytest.raises(pkg.PoseliftError, match="planes"):
        m.workspace_view(ws, 1, 1)
    assert m.workspace_view(ws, 1, 2).shape == (4096, 1024)
    # gradients of the planes path against the fp32-MFMA path of the same library, same dropout stream
    m32.train().manual_seed(11, step=3)
    m32.zero_grad(set_to_none=True)
    pkg.mse_loss(m32(x).reshape(-1, 17, 3), y).backward()
    g16, g32 = outs[0][1].double(), m32.flat_grads.double()
    assert float((g16 - g32).norm() / g32.norm()) < 2e-4


def test_f16x3_gradient_scale_invariance(pkg, full):
    """The dz planes are range-scaled per layer from a bound computed in BatchNorm-backward pass 1: a loss 1e4 x
    larger or smaller gives gradients 1e4 x larger or smaller (to fp32 round-off), never an fp16 overflow or flush."""
    m, x, y = full
    grads = {}
    for fac in (1.0, 1e4, 1e-4):
        m.train().manual_seed(5, step=1)
        m.zero_grad(set_to_none=True)
        (pkg.mse_loss(m(x).reshape(-1, 17, 3), y) * fac).backward()
        grads[fac] = m.flat_grads.double().clone()
        assert torch.isfinite(grads[fac]).all()
    for fac in (1e4, 1e-4):
        rel = float((grads[fac] / fac - grads[1.0]).norm() / grads[1.0].norm())
        assert rel < 1e-5, (fac, rel)


@pytest.mark.parametrize("B,bn", [(100, True), (4097, True), (256, False)])
def test_f16x3_falls_back_to_the_fp32_grade_kernels_off_the_tile_grid(pkg, B, bn):
    """Ragged batches and BN=False do not meet the planes path's conditions: the same request runs PL_BF16X6 / fp32-MFMA
    arithmetic on the round-1 kernels (never a lower precision, never a CPU path) and still matches the oracle."""
    torch.manual_seed(1)
    H = 256
    m = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, BN=bn, compute_dtype="f16x3").to(DEV).train()
    st = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    x, y = pkg.synth.synthetic_batch(B, 3, DEV)
    pred = m(x).reshape(B, 17, 3)
    pkg.mse_loss(pred, y).backward()
    opred, cache = orc.forward(st, x.cpu().numpy(), num_stage=2, train=True, use_bn=bn, p_dropout=0.0)
    p64, _ = orc.forward({k: v.copy() for k, v in st.items()}, x.cpu().numpy(), num_stage=2, train=True, use_bn=bn,
                         p_dropout=0.0, dtype=np.float64)
    e_gpu, e_ref = orc.mpjpe_mm(pred.detach().cpu().numpy(), p64), orc.mpjpe_mm(opred, p64)
    assert e_gpu <= 3 * e_ref + 1e-4, (e_gpu, e_ref)
    assert torch.isfinite(m.flat_grads).all()


@pytest.mark.parametrize("dtype,B,H", [("f16x3", 256, 256), ("fp32", 64, 128), ("f16x3", 4096, 1024)])
def test_graphed_train_step_is_bitwise_the_eager_one(pkg, dtype, B, H):
    """The whole train_1.py step (forward, MSE, backward, AdamW) captured once as a hipGraph and replayed: dropout
    masks, AdamW bias corrections and the learning rate advance inside the graph (device step counter, device lr), so
    every replay equals the eager step of the same number bit for bit -- losses, outputs, parameters, BN buffers."""
    def make():
        torch.manual_seed(3)
        m = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.5, compute_dtype=dtype).to(DEV).train()
        m.manual_seed(99, step=0)
        return m, pkg.FlatAdamW(m, lr=1e-3)
    batches = [pkg.synth.synthetic_batch(B, 40 + i, DEV) for i in range(4)]
    me, oe = make()
    eager = []
    for i, (x, y) in enumerate(batches):
        if i == 2:
            oe.param_groups[0]["lr"] = 5e-4                      # a scheduler stepping between two steps
        loss, yh = pkg.train_step(me, oe, x, y)
        eager.append((loss.clone(), yh.clone()))
    mg, og = make()
    step = pkg.GraphedTrainStep(mg, og, *batches[0])
    assert torch.equal(mg.flat_params, make()[0].flat_params)   # building the graph trained nothing
    for i, (x, y) in enumerate(batches):
        if i == 2:
            og.param_groups[0]["lr"] = 5e-4
        loss, yh = step(x, y)
        assert torch.equal(loss, eager[i][0]) and torch.equal(yh, eager[i][1]), i
    assert torch.equal(mg.flat_params, me.flat_params)
    assert torch.equal(mg._bn_running, me._bn_running) and torch.equal(mg._bn_batches, me._bn_batches)
    assert torch.equal(og._m, oe._m) and torch.equal(og._v, oe._v) and og._t == oe._t == 4
    # and the eager path carries on from where the graph left off
    l1, _ = pkg.train_step(mg, og, *batches[0])
    l2, _ = pkg.train_step(me, oe, *batches[0])
    assert torch.equal(l1, l2)


@pytest.mark.parametrize("dtype,B,H", [("fp32", 96, 128), ("f16x3", 256, 256), ("bf16x6", 160, 256)])
def test_eval_mode_backward_vs_torch_twin(pkg, dtype, B, H):
    """model.eval() inside an autograd graph -- phase5_loop/train_5.py:120 runs the lifter in eval mode with gradients
    flowing through it (twice per graph) into Model_2D.  BatchNorm on non-trivial running statistics, Dropout the
    identity: outputs, input gradient and every parameter gradient (dgamma / dbeta and the pre-BN biases included:
    in eval mode they are ordinary affine parameters) against the stock-PyTorch twin in fp64 on the CPU; the running
    statistics must not move."""
    from oracle.torch_twin import TwinLifter
    torch.manual_seed(17)
    st = orc.init_state(34, 51, H, 2, rng=np.random.default_rng(5), nontrivial_bn=True)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in st.items()}
    m = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.5, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd)
    m.eval()
    tw = TwinLifter(34, 51, linear_size=H, p_dropout=0.5).double()
    tw.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    tw.eval()
    xa, xb = torch.rand(B, 34), torch.rand(B, 34)
    ta, tb = torch.rand(B, 51) - 0.5, torch.rand(B, 51) - 0.5
    outs = []
    for model, dev, dt in ((m, DEV, torch.float32), (tw, "cpu", torch.float64)):
        a = xa.clone().to(dev, dt).requires_grad_(True)
        ya, yb = model(a), model(xb.to(dev, dt))                 # two calls in one graph, the second without dx
        loss = ((ya - ta.to(dev, dt)) ** 2).mean() + 0.5 * ((yb - tb.to(dev, dt)) ** 2).mean()
        loss.backward()
        outs.append((ya.detach().cpu().double().numpy(), a.grad.cpu().double().numpy(),
                     {k: p.grad.cpu().double().numpy() for k, p in model.named_parameters()}))
    (yg, dxg, gg), (yc, dxc, gc) = outs
    assert orc.mpjpe_mm(yg, yc) < 1e-3
    with torch.no_grad():
        assert orc.mpjpe_mm(m(xa.to(DEV)).cpu().numpy(), yg) < 1e-3     # the nothing-saved eval forward agrees
    assert np.abs(dxg - dxc).max() <= 5e-4 * np.abs(dxc).max()
    for k, v in gc.items():
        rel = np.linalg.norm(gg[k] - v) / (np.linalg.norm(v) + 1e-30)
        assert rel < 2e-4, (k, rel)
    after = m.state_dict()
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            assert torch.equal(after[k].cpu(), v), k
    assert m._live_graphs == 0


# ---------------------------------------------------------------------------- freshness of the persistent weight planes
@pytest.mark.parametrize("dtype", ["f16x3", "bf16"])
def test_weight_planes_follow_load_state_dict_and_torch_optimizers(pkg, dtype):
    """The 1024-wide weights' operand planes persist across calls.  Parameters are attached to the arena with
    `p.data = view`, so an in-place torch op (load_state_dict, nn.init, torch.optim.*.step) bumps only that Parameter's
    version counter: the planes must be re-derived all the same (they were not)."""
    H, B = 256, 256
    tol = 1e-3 if dtype == "f16x3" else 30.0                  # bf16 storage: ~1 mm-class, only staleness matters
    torch.manual_seed(1)
    a = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype=dtype).to(DEV).eval()
    torch.manual_seed(2)
    b = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype=dtype).to(DEV).eval()
    ref = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype="fp32").to(DEV).eval()
    x, y = pkg.synth.synthetic_batch(B, 9, DEV)
    with torch.no_grad():
        ya = a(x)                                             # planes of a's weights are now valid
        a.load_state_dict(b.state_dict())                     # ... and now stale
        ref.load_state_dict(b.state_dict())
        got, want, fresh = a(x), ref(x), b(x)
    assert not torch.equal(ya, got)
    assert torch.equal(got, fresh)                            # same weights, same kernels: bit-identical
    assert orc.mpjpe_mm(got.cpu().numpy(), want.cpu().numpy()) < tol
    # weight_init (nn.init on the Linear weights) after a forward
    torch.manual_seed(3)
    a.apply(pkg.weight_init)
    ref.load_state_dict(a.state_dict())
    with torch.no_grad():
        assert orc.mpjpe_mm(a(x).cpu().numpy(), ref(x).cpu().numpy()) < tol
    # a stock torch optimizer stepping the parameters (what cycle_step's Adam does, train_5 copy.py:105-109): after every
    # step the model must compute with the NEW weights -- bit for bit what a fresh model loaded with them computes
    # (two arithmetics' Adam trajectories diverge on their own: a same-arithmetic twin is the staleness check)
    a.train()
    oa = torch.optim.Adam(a.parameters(), lr=1e-2)
    for _ in range(3):
        oa.zero_grad()
        pkg.mse_loss(a(x).reshape(B, 17, 3), y).backward()
        oa.step()
        fresh = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype=dtype).to(DEV).train()
        fresh.load_state_dict(a.state_dict())
        sd = {k: v.clone() for k, v in a.state_dict().items()}
        assert torch.equal(a(x), fresh(x))                    # (training-mode forward: moves the running statistics)
        a.load_state_dict(sd)
    a.eval(); fresh.eval()
    fresh.load_state_dict(a.state_dict())
    with torch.no_grad():
        assert torch.equal(a(x), fresh(x))


def test_weight_planes_are_written_before_the_first_planes_call_whatever_came_first(pkg):
    """A call off the planes path (ragged batch) leaves the planes untouched: the next whole-tile call must not find
    them marked valid (fresh model: the buffer is uninitialised memory).  ADVICE r02."""
    H = 256
    torch.manual_seed(4)
    m = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype="f16x3").to(DEV).eval()
    m._wplanes.fill_(0xFF)                                    # poison: NaN planes if they were ever read as they are
    ref = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype="bf16x6").to(DEV).eval()
    ref.load_state_dict(m.state_dict())
    x, _ = pkg.synth.synthetic_batch(128, 5, DEV)
    with torch.no_grad():
        y100 = m(x[:100])                                     # round-1 kernels
        y128 = m(x)                                           # planes path
        r100, r128 = ref(x[:100]), ref(x)
    assert torch.isfinite(y128).all()
    assert orc.mpjpe_mm(y100.cpu().numpy(), r100.cpu().numpy()) < 1e-3
    assert orc.mpjpe_mm(y128.cpu().numpy(), r128.cpu().numpy()) < 1e-3
    # flip TTA at B = 64 runs 2B = 128 rows (the ADVICE example)
    m2 = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype="f16x3").to(DEV).eval()
    m2._wplanes.fill_(0xFF)
    m2.load_state_dict(m.state_dict())
    with torch.no_grad():
        m2(x[:64])
        t = pkg.train.predict_flip_tta(m2, x[:64])
        tr = pkg.train.predict_flip_tta(ref, x[:64])
    assert orc.mpjpe_mm(t.cpu().numpy().reshape(64, -1), tr.cpu().numpy().reshape(64, -1)) < 1e-3


@pytest.mark.parametrize("H,B", [(512, 128 * 17), (256, 128 * 5)])
def test_f16x3_shapes_whose_batch_does_not_split_into_whole_k_tiles_fall_back(pkg, H, B):
    """H = 512 with B = 128 * 17: the weight-gradient GEMM would split K = B sixteen ways, not whole 32-k tiles -- such
    a batch must run on the round-1 kernels (as fp32 / bf16x6 do), not fail with PL_ESHAPE.  ADVICE r02."""
    torch.manual_seed(6)
    m = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype="f16x3").to(DEV).train()
    r = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.0, compute_dtype="bf16x6").to(DEV).train()
    r.load_state_dict(m.state_dict())
    x, y = pkg.synth.synthetic_batch(B, 8, DEV)
    for mm in (m, r):
        pkg.mse_loss(mm(x).reshape(B, 17, 3), y).backward()
    assert torch.isfinite(m.flat_grads).all()
    rel = float((m.flat_grads.double() - r.flat_grads.double()).norm() / r.flat_grads.double().norm())
    assert rel < 2e-4, rel


def test_graphed_train_step_follows_or_refuses_host_side_counter_changes(pkg):
    """The graph owns the dropout-stream step and AdamW's t (capture-time base + a device counter).  An eager
    train_step between replays moves the host counters: the next replay re-seeds the device counter and stays bitwise
    the eager sequence; counters that drift apart, a new dropout seed or re-bound optimizer arenas are refused."""
    def make():
        torch.manual_seed(3)
        m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.5, compute_dtype="f16x3").to(DEV).train()
        m.manual_seed(99, step=0)
        return m, pkg.FlatAdamW(m, lr=1e-3)
    batches = [pkg.synth.synthetic_batch(256, 60 + i, DEV) for i in range(4)]
    me, oe = make()
    eager = [pkg.train_step(me, oe, x, y)[0].clone() for x, y in batches]
    mg, og = make()
    step = pkg.GraphedTrainStep(mg, og, *batches[0])
    l0 = step(*batches[0])[0].clone()
    l1 = pkg.train_step(mg, og, *batches[1])[0].clone()       # an eager step in between
    l2 = step(*batches[2])[0].clone()
    l3 = step(*batches[3])[0].clone()
    for got, want in zip((l0, l1, l2, l3), eager):
        assert torch.equal(got, want)
    assert torch.equal(mg.flat_params, me.flat_params)
    og._advance_host(1)                                       # optimizer ahead of the model
    with pytest.raises(pkg.PoseliftError, match="capture a new"):
        step(*batches[0])
    og._advance_host(-1)
    mg.manual_seed(5, step=mg._step)
    with pytest.raises(pkg.PoseliftError, match="capture a new"):
        step(*batches[0])


@pytest.mark.gpu
@pytest.mark.parametrize("graphed", [False, True])
def test_weight_planes_follow_small_batch_steps(graphed):
    """Training at the reference's batch of 64 (AdamW inside the backward launches: pl_lifter_train_step) never touches the
    persistent weight planes; an evaluation at a batch on the planes path afterwards must see the NEW weights -- eager steps
    and replayed ones (a replay bumps no tensor version: GraphedTrainStep says so itself)."""
    import __graft_entry__ as ge
    pkg = ge.build()
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, p_dropout=0.5, compute_dtype="f16x3").to("cuda:0").train()
    opt = pkg.FlatAdamW(m, lr=1e-2)
    x, y = pkg.synth.synthetic_batch(64, 5, "cuda:0")
    xe, _ = pkg.synth.synthetic_batch(256, 6, "cuda:0")
    m.eval()
    with torch.no_grad():
        y0 = m(xe).clone()                      # (writes the planes of the initial weights)
    m.train()
    step = pkg.GraphedTrainStep(m, opt, x, y) if graphed else (lambda a, b: pkg.train_step(m, opt, a, b))
    for _ in range(3):
        step(x, y)
    m.eval()
    with torch.no_grad():
        y1 = m(xe).clone()
        m._wplanes_ver = None                   # force a refresh: what the planes must have been
        y2 = m(xe).clone()
    assert torch.equal(y1, y2)
    assert (y1 - y0).abs().max().item() > 1e-3  # the three steps moved the weights
