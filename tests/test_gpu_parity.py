"""GPU parity tests: the HIP path (through the C ABI, via the Python host) against
 (a) the golden vectors recorded from the reference model (tests/golden/, tools/make_golden.py),
 (b) the numpy oracle on the same seeded inputs,
 (c) size-independent properties at BASELINE.json's full size (B = 4096, H = 1024).
Tolerances: fp32 path.  Forward gate = BASELINE.json's 1e-3 mm MPJPE; gradients to 2e-5 of
each tensor's max (5e-4 for the 1024-wide model: a ReLU input within round-off of zero may
flip between two summation orders, see tests/test_oracle_golden.py)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_state, load_golden
from oracle import lifter_oracle as orc
from oracle import philox

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    p = ge.build()
    assert torch.cuda.is_available()
    return p


DEV = "cuda:0"


def _model_from_state(pkg, st, hidden, S, p, bn, dtype="fp32"):
    m = pkg.LinearModel(34, 51, linear_size=hidden, num_stage=S, p_dropout=p, BN=bn, compute_dtype=dtype).to(DEV)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    return m


def _close(a, b, rtol, atol):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _grads(model):
    return {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}


def _close_most(a, b, atol, frac=0.002, cap=0.1):
    """|a-b| <= atol for all but a fraction `frac` of the elements (at least one allowed), and
    never beyond `cap`.  A ReLU input within round-off of zero can flip between two correct
    fp32 evaluations; the sample it belongs to then enters or leaves one column sum, moving
    that single gradient element by ~1/B of its magnitude (seen: 1 of 1024 elements)."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    bad = int((d > atol).sum())
    assert bad <= max(1, int(frac * d.size)) and d.max() <= cap, (bad, d.size, d.max())


def _check_grads(got, want, bn, tol=2e-5, flips=False):
    for k, v in want.items():
        scale = np.abs(v).max() + 1e-30
        t = tol
        if bn and k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias":
            # zero-true-gradient bias: BOTH sides are round-off of a cancelling sum, amplified by gamma * rstd of the column
            # (a near-constant feature: rstd in the hundreds).  Judged on the scale of the layer's weight gradient, with
            # room for two different summation orders (round 3's one-launch small-batch backward sums in another order
            # than the chunked passes: 8.6e-5 of that scale between the two, one column)
            scale = np.abs(want[k[:-4] + "weight"]).max()
            t = 3 * tol
        if flips:
            _close_most(got[k] / scale, v / scale, t)
        else:
            _close(got[k] / scale, v / scale, 0, t)


def _assert_train_fwd(pred_gpu, pred_ref32, pred_fp64):
    """Training-mode forward: batch statistics amplify fp32 round-off, so two correct fp32
    implementations differ by 1e-3..1e-2 mm (reference fp32 vs the same model in fp64:
    1.6e-3 mm at B=128, 1.1e-2 mm at B=4096 -- DESIGN.md 'noise floors').  The HIP path must be
    as close to the fp64 result as the fp32 reference is (factor 3), not closer than possible.
    The strict 1e-3 mm gate of BASELINE.json applies to the eval forward (tests below)."""
    e_gpu = orc.mpjpe_mm(pred_gpu, pred_fp64)
    e_ref = orc.mpjpe_mm(pred_ref32, pred_fp64)
    assert e_gpu <= 3 * e_ref + 1e-4, (e_gpu, e_ref)


def _gpu_decisions(pkg, m, n_layers, H):
    """The HIP path's own positive&kept decisions of the last training forward (bitmaps)."""
    ws = m.last_workspace
    return [pkg.layout.unpack_bitmap(m.workspace_view(ws, 2, l).cpu().numpy().view(np.uint64), H)
            for l in range(n_layers)]


def _assert_decisions_consistent(cache, tol=1e-4):
    """Every forced decision that differs from the oracle's own sits on a round-off-sized
    pre-activation (BN outputs are O(1)), and there are few of them."""
    for c in cache["layers"]:
        d = c["on_disagree"]
        assert d.size <= 1e-4 * c["z"].size + 2 and (d.size == 0 or d.max() < tol), (d.size, d.max() if d.size else 0)


# ---------------------------------------------------------------------------- GEMM block
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 96), (130, 51, 34), (64, 1024, 51),
                                   (1024, 34, 300), (4096, 1024, 1024), (1, 7, 5), (257, 129, 1030)])
def test_gemm_layouts(pkg, layout, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K + layout)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if layout != 2 else None
    A = _t(a if layout != 2 else a.T)
    Bm = _t(b.T if layout == 0 else b)
    C = torch.full((M, N), float("nan"), device=DEV)
    bt = _t(bias) if bias is not None else None
    rc = pkg.lib().pl_gemm_f32(layout, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K,
                               bt.data_ptr() if bt is not None else None, 1, None,
                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0, pkg.lib().pl_last_error()
    want = a.astype(np.float64) @ b.astype(np.float64) + (bias if bias is not None else 0)
    got = C.cpu().numpy()
    # k-ordered fp32 fma chain: typical error ~1e-7 * sum|a||b|; 2e-6 bounds the max over 4M outputs
    bound = 2e-6 * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) + 1e-6
    assert np.all(np.abs(got - want) <= bound)


@pytest.mark.parametrize("M,N,K,splits", [(1024, 1024, 4096, 4), (51, 1024, 4096, 32), (1024, 34, 777, 5)])
def test_gemm_tn_split_k(pkg, M, N, K, splits):
    rng = np.random.default_rng(K + splits)
    a = rng.standard_normal((K, M)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    A, Bm = _t(a), _t(b)
    C = torch.full((M, N), float("nan"), device=DEV)
    slabs = torch.full((splits, M, N), float("nan"), device=DEV)
    rc = pkg.lib().pl_gemm_f32(2, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, None, splits,
                               slabs.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, pkg.lib().pl_last_error()
    want = a.T.astype(np.float64) @ b.astype(np.float64)
    bound = 2e-6 * (np.abs(a.T).astype(np.float64) @ np.abs(b).astype(np.float64)) + 1e-6
    assert np.all(np.abs(C.cpu().numpy() - want) <= bound)


@pytest.mark.parametrize("arith,tol", [(1, 2e-2), (2, 2e-6)])
@pytest.mark.parametrize("layout,M,N,K,splits", [(0, 4096, 1024, 1024, 1), (1, 4096, 1024, 1024, 1),
                                                 (2, 1024, 1024, 4096, 4), (0, 256, 128, 96, 1), (2, 128, 256, 512, 1)])
def test_gemm_bf16_arithmetic_modes(pkg, arith, tol, layout, M, N, K, splits):
    """PL_BF16 (1): bf16-sized error.  PL_BF16X6 (2): three-way split, must stay inside the fp32 bound."""
    rng = np.random.default_rng(M + N + K + layout)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    A = _t(a if layout != 2 else a.T)
    Bm = _t(b.T if layout == 0 else b)
    C = torch.full((M, N), float("nan"), device=DEV)
    slabs = torch.empty(splits, M, N, device=DEV) if splits > 1 else None
    rc = pkg.lib().pl_gemm_arith(layout, arith, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, None, splits,
                                 slabs.data_ptr() if splits > 1 else None, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, pkg.lib().pl_last_error()
    want = a.astype(np.float64) @ b.astype(np.float64)
    bound = tol * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) + 1e-6
    err = np.abs(C.cpu().numpy() - want)
    assert np.all(err <= bound), float((err / bound).max())
    if arith == 1:
        assert err.max() > 1e-4          # really bf16


@pytest.mark.parametrize("layout,M,N,K,splits", [(0, 4096, 1024, 1024, 1), (1, 4096, 1024, 1024, 1),
                                                 (2, 1024, 1024, 4096, 4), (0, 256, 128, 96, 1), (2, 128, 256, 512, 1),
                                                 (1, 128, 384, 64, 1), (2, 256, 128, 256, 2)])
def test_bf16x6_planes_pipeline_is_bitwise_the_fragment_split(pkg, layout, M, N, K, splits):
    """The two PL_BF16X6 main loops (split at staging into bf16 planes in LDS = 5, split per fragment = 6)
    run the same splits and the same MFMA order: identical bits, with a bias in the epilogue too."""
    rng = np.random.default_rng(M * 3 + N + K + layout)
    A = _t(rng.standard_normal((K, M) if layout == 2 else (M, K)).astype(np.float32))
    Bm = _t(rng.standard_normal((N, K) if layout == 0 else (K, N)).astype(np.float32))
    bias = _t(rng.standard_normal(N).astype(np.float32)) if splits == 1 else None
    outs = []
    for arith in (5, 6, 2):
        C = torch.full((M, N), float("nan"), device=DEV)
        slabs = torch.empty(splits, M, N, device=DEV) if splits > 1 else None
        rc = pkg.lib().pl_gemm_arith(layout, arith, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K,
                                     bias.data_ptr() if bias is not None else None, splits,
                                     slabs.data_ptr() if splits > 1 else None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, pkg.lib().pl_last_error()
        outs.append(C)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[0]).all()


# ---------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("dtype", ["fp32", "bf16x6", "f16x3"])
def test_g1_eval_forward_vs_reference(pkg, dtype):
    """The BASELINE.json gate.  bf16x6 (three-way bf16 split, six MFMAs per product) and f16x3 (two fp16 planes
    written by the producers, three MFMAs) must meet it too; the batch is tiled to 128 rows so that their
    whole-tile MFMA paths are the ones that run."""
    g = load_golden("g1_eval_full.npz")
    st = orc.init_state(34, 51, 1024, 2, rng=np.random.default_rng(int(g["weight_seed"])), nontrivial_bn=True)
    m = pkg.LinearModel(34, 51, linear_size=1024, compute_dtype=dtype).to(DEV)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    m.eval()
    with torch.no_grad():
        y = m(_t(np.concatenate([g["x"], g["x"]]))).cpu().numpy()
    assert np.array_equal(y[:64], y[64:])
    y = y[:64]
    assert orc.mpjpe_mm(y, g["y"]) < 1e-3            # BASELINE.json parity gate, reference fp32 forward
    assert orc.mpjpe_mm(y, g["y_fp64"]) < 1e-3       # and against the reference run in fp64


@pytest.mark.parametrize("tag", ["small", "nobn", "s3", "full", "full-f16x3"])
def test_g2_train_nodrop_vs_reference(pkg, tag):
    dtype = "fp32"
    if tag == "full-f16x3":          # B = 128, H = 1024: the fp16-planes GEMM path, against the same reference outputs
        tag, dtype = "full", "f16x3"
    g = load_golden(f"g2_train_nodrop_{tag}.npz")
    H, S, bn = int(g["hidden"]), int(g["num_stage"]), bool(g["bn"])
    if tag == "full":
        st = orc.init_state(34, 51, H, S, rng=np.random.default_rng(int(g["weight_seed"])), nontrivial_bn=True)
    else:
        st = golden_state(g)
    m = _model_from_state(pkg, st, H, S, 0.0, bn, dtype).train()
    x = _t(g["x"]).requires_grad_(True)
    pred = m(x).reshape(g["pred"].shape)
    loss = pkg.mse_loss(pred, _t(g["t"]))
    loss.backward()
    p64, _ = orc.forward({k: v.copy() for k, v in st.items()}, g["x"], num_stage=S, train=True, use_bn=bn,
                         p_dropout=0.0, dtype=np.float64)
    _assert_train_fwd(pred.detach().cpu().numpy(), g["pred"], p64)
    _close(loss.item(), g["loss"], 2e-5, 0)
    got = _grads(m)
    if tag == "full":
        for k in orc.param_names(2):
            flat = got[k].reshape(-1)
            pre_bn_bias = k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias"
            if pre_bn_bias:
                scale = np.abs(g["gval:" + k[:-4] + "weight"]).max()
            else:
                scale = np.abs(g["gval:" + k]).max()
                _close(np.linalg.norm(flat.astype(np.float64)), float(g["gnorm:" + k]), 1e-3, 0)
            if dtype == "f16x3":
                # B = 128 on the operand-planes path: the forward Linears run on the layer kernels' contraction (another
                # summation order than the tile GEMM's and the reference's), and a ReLU decision that flips on a
                # round-off-sized pre-activation moves the samples of the rows it touches by 1 / B of their size.  With the
                # decisions forced (tools/r3_mid_accuracy.py, same shape) every gradient tensor is within 7e-7 of the fp64
                # oracle in relative L2 -- the fp32 oracle itself: 1.3e-6; the tile GEMM: 8e-7 -- and
                # test_ragged_shapes_vs_oracle[128-1024-2-...-f16x3] holds that in the suite.  Here, against the reference's own
                # decisions: the typical sample within 1e-3 of the scale, none beyond 2 %.
                dv = np.abs(flat[g["gidx:" + k]].astype(np.float64) - g["gval:" + k]) / scale
                assert np.median(dv) < 1e-3 and np.mean(dv > 2e-3) < 0.2 and dv.max() < 2e-2, (k, np.median(dv), dv.max())
            else:
                _close(flat[g["gidx:" + k]] / scale, g["gval:" + k] / scale, 0, 5e-4)
    else:
        want = golden_state(g, "grad:")
        _check_grads(got, want, bn)
        if not bn:
            assert m.batch_norm1.weight.grad is None      # unused parameters keep grad None (as in torch)
    sdx = np.abs(g["dx"]).max()
    _close(x.grad.cpu().numpy().reshape(g["dx"].shape) / sdx, g["dx"] / sdx, 0,
           (2e-2 if dtype == "f16x3" else 5e-4) if tag == "full" else 2e-5)
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in golden_state(g, "after:").items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v)
        else:
            _close(sd[k], v, 1e-5, 1e-6)


def test_g3_train_with_reference_dropout_masks(pkg):
    g = load_golden("g3_train_masks_small.npz")
    st = golden_state(g)
    m = _model_from_state(pkg, st, 64, 2, 0.5, True).train()
    words = np.stack([pkg.layout.pack_keep_bitmap(k) for k in g["masks"].astype(bool)])
    m.debug_inject_keep(_t(words.view(np.int64)))
    pred = m(_t(g["x"])).reshape(g["pred"].shape)
    loss = pkg.mse_loss(pred, _t(g["t"]))
    loss.backward()
    p64, _ = orc.forward({k: v.copy() for k, v in st.items()}, g["x"], num_stage=2, train=True, p_dropout=0.5,
                         keep_masks=[k.astype(bool) for k in g["masks"]], dtype=np.float64)
    _assert_train_fwd(pred.detach().cpu().numpy(), g["pred"], p64)
    _close(loss.item(), g["loss"], 2e-5, 0)
    _check_grads(_grads(m), golden_state(g, "grad:"), True)


@pytest.mark.parametrize("how", ["injected masks", "p = 0", "p = 1"])
def test_small_batch_layer_kernels_dropout_modes_vs_oracle(pkg, how):
    """The column-owning layer kernels (csrc/small_layer.hip: B <= 64, H % 256 == 0) under the dropout modes the Philox tests
    do not reach: keep decisions injected as bitmaps (the reference-mask parity mode), Dropout(0) and Dropout(1) -- forward,
    every gradient and dx against the numpy oracle with the same masks."""
    B, H, S = 24, 256, 2
    p = {"injected masks": 0.5, "p = 0": 0.0, "p = 1": 1.0}[how]
    torch.manual_seed(3)
    m = pkg.LinearModel(34, 51, linear_size=H, num_stage=S, p_dropout=p).to(DEV).train()
    st = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 34, generator=g).to(DEV).requires_grad_(True)
    t = (torch.rand(B, 51, generator=g) - 0.5).to(DEV)
    L = 1 + 2 * S
    rng = np.random.default_rng(5)
    if how == "injected masks":
        masks = [rng.random((B, H)) < 0.5 for _ in range(L)]
        words = np.stack([pkg.layout.pack_keep_bitmap(k) for k in masks])
        m.debug_inject_keep(_t(words.view(np.int64)))
    else:
        masks = [np.full((B, H), p == 0.0) for _ in range(L)]
    pred = m(x)
    loss = pkg.mse_loss(pred, t)
    loss.backward()
    opred, cache = orc.forward(st, x.detach().cpu().numpy(), num_stage=S, train=True, use_bn=True, p_dropout=p if p < 1 else 0.5,
                               keep_masks=masks, on_masks=_gpu_decisions(pkg, m, L, H))
    _assert_decisions_consistent(cache, 1e-3)
    oloss, dpred = orc.mse_loss(opred, t.cpu().numpy())
    ograds, odx = orc.backward(st, cache, dpred)
    scale = max(np.abs(opred).max(), 1e-6)
    _close(pred.detach().cpu().numpy() / scale, opred / scale, 0, 2e-5)
    _close(loss.item(), oloss, 2e-5, 0)
    got = _grads(m)
    for k, v in ograds.items():
        if (k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias") or np.linalg.norm(v) == 0:
            assert np.linalg.norm(got[k]) <= 1e-4 * (1 + np.linalg.norm(v)) or not np.linalg.norm(v) == 0
            continue
        rel = np.linalg.norm((got[k] - v).astype(np.float64)) / (np.linalg.norm(v.astype(np.float64)) + 1e-30)
        assert rel < 2e-4, (k, rel)
    if np.linalg.norm(odx) > 0:
        rel = np.linalg.norm((x.grad.cpu().numpy() - odx).astype(np.float64)) / np.linalg.norm(odx.astype(np.float64))
        assert rel < 2e-4, ("dx", rel)


def test_g4_three_adamw_steps_vs_reference(pkg):
    g = load_golden("g4_adamw_small.npz")
    m = _model_from_state(pkg, golden_state(g), 64, 2, 0.0, True).train()
    opt = pkg.FlatAdamW(m, lr=float(g["lr"]), weight_decay=float(g["wd"]))
    for i in range(3):
        loss, _ = pkg.train_step(m, opt, _t(g["xs"][i]), _t(g["ts"][i]))
        _close(loss.item(), g["losses"][i], 2e-5, 0)
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k, v in golden_state(g, "final:").items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v)
        elif (k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias") or k.endswith("running_mean"):
            _close(sd[k], v, 0, 2 * 3 * float(g["lr"]) + 1e-6)   # Adam on round-off noise, see oracle test
        else:
            _close(sd[k], v, 1e-5, 2e-7)


def test_g5_mpjpe_and_mse(pkg):
    g = load_golden("g5_mpjpe.npz")
    a, b = _t(g["pred"]), _t(g["tgt"])
    metric = pkg.loss_MPJPE(a, b)
    _close(metric.cpu().numpy(), g["metric"], 1e-5, 1e-6)
    assert float(metric[0]) == 0.0
    pkg.loss_MPJPE(a, b, out=metric)                      # accumulates like `train_metric_3d +=`
    _close(metric.cpu().numpy(), 2 * g["metric"], 1e-5, 1e-6)
    assert abs(float(pkg.epoch_mpjpe_mm(metric / 2, 48)) - float(g["epoch_mm"])) < 1e-2
    want, dwant = orc.mse_loss(g["pred"], g["tgt"], np.float64)
    ar = a.clone().requires_grad_(True)
    loss = pkg.mse_loss(ar, b)
    loss.backward()
    _close(loss.item(), want, 1e-6, 0)
    _close(ar.grad.cpu().numpy(), dwant, 1e-6, 1e-12)


# ---------------------------------------------------------------------------- oracle, same inputs
def test_philox_dropout_bits_exact(pkg):
    """With BN off, zero weights and a positive bias every pre-activation is > 0, so the
    stored keep&relu bitmap IS the Philox keep mask: compare it word for word."""
    H, B = 320, 37          # two 256-column strips, the second one partial
    m = pkg.LinearModel(34, 51, linear_size=H, num_stage=1, p_dropout=0.5, BN=False).to(DEV).train()
    with torch.no_grad():
        for p in m.parameters():
            p.zero_()
        m.w1.bias.fill_(1.0), m.linear_stages[0].w1.bias.fill_(1.0), m.linear_stages[0].w2.bias.fill_(1.0)
    m.manual_seed(0x1234_5678_9ABC_DEF0, step=41)
    x = torch.rand(B, 17, 2, device=DEV)
    with torch.no_grad():
        m(x)
    ws = m.last_workspace
    for layer in range(3):
        got = pkg.layout.unpack_bitmap(m.workspace_view(ws, 2, layer).cpu().numpy().view(np.uint64), H)
        want = philox.dropout_keep_mask(0x1234_5678_9ABC_DEF0, 42, layer, B, H, 0.5)
        assert (got == want).all()
        act = m.workspace_view(ws, 1, layer).cpu().numpy()
        base = 1.0 if layer < 2 else None
        if base is not None:
            assert np.array_equal(act, np.where(want, 2.0, 0.0).astype(np.float32))


@pytest.mark.parametrize("p", [0.5, 0.25])
def test_train_step_with_philox_dropout_vs_oracle(pkg, p):
    torch.manual_seed(3)
    H, S, B = 128, 2, 96
    m = pkg.LinearModel(34, 51, linear_size=H, num_stage=S, p_dropout=p, BN=True).to(DEV).train()
    st = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    m.manual_seed(99, step=6)
    x, y = pkg.synth.synthetic_batch(B, 5, DEV)
    pred = m(x).reshape(B, 17, 3)
    loss = pkg.mse_loss(pred, y)
    loss.backward()
    masks = [philox.dropout_keep_mask(99, 7, l, B, H, p) for l in range(1 + 2 * S)]
    p64, _ = orc.forward({k: v.copy() for k, v in st.items()}, x.cpu().numpy(), num_stage=S, train=True,
                         p_dropout=p, keep_masks=masks, dtype=np.float64)
    # same Philox masks, the HIP path's own ReLU decisions (must agree up to round-off ties)
    opred, cache = orc.forward(st, x.cpu().numpy(), num_stage=S, train=True, p_dropout=p, keep_masks=masks,
                               on_masks=_gpu_decisions(pkg, m, 1 + 2 * S, H))
    _assert_decisions_consistent(cache)
    oloss, dpred = orc.mse_loss(opred, y.cpu().numpy().reshape(B, -1))
    ograds, _ = orc.backward(st, cache, dpred)
    _assert_train_fwd(pred.detach().cpu().numpy(), opred, p64)
    _close(loss.item(), oloss, 2e-5, 0)
    _check_grads(_grads(m), ograds, True, tol=1e-4)


def test_flat_adamw_matches_oracle_and_torch_state_layout(pkg):
    torch.manual_seed(5)
    m = pkg.LinearModel(34, 51, linear_size=64, num_stage=1, p_dropout=0.0).to(DEV).train()
    opt = pkg.FlatAdamW(m, lr=3e-4, weight_decay=0.02)
    p0 = m.flat_params.clone()
    rng = np.random.default_rng(0)
    mo = np.zeros(p0.numel(), np.float32)
    vo = np.zeros_like(mo)
    po = p0.cpu().numpy()
    for t in range(1, 4):
        gnp = rng.standard_normal(p0.numel()).astype(np.float32) * 1e-2
        m.flat_grads.copy_(_t(gnp))
        opt.step()
        po, mo, vo = orc.adamw_step(po, gnp, mo, vo, t, lr=3e-4, wd=0.02)
    real = np.zeros(p0.numel(), bool)                      # alignment padding is not a parameter
    for sl in m._slots:
        real[sl.offset:sl.offset + sl.numel] = True
    _close(m.flat_params.cpu().numpy()[real], po[real], 1e-6, 1e-8)
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and len(sd["state"]) == 14
    assert float(sd["state"][0]["step"]) == 3.0
    # the stock optimiser can resume from it, and ours from the stock one's
    ref = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=0.02)
    ref.load_state_dict(sd)
    opt2 = pkg.FlatAdamW(m, lr=3e-4, weight_decay=0.02)
    opt2.load_state_dict(ref.state_dict())
    _close(opt2._m.cpu().numpy()[real], mo[real], 1e-6, 1e-9)


# ---------------------------------------------------------------------------- full size properties
@pytest.fixture(scope="module")
def full(pkg):
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=1024, num_stage=2, p_dropout=0.5, BN=True).to(DEV)
    x, y = pkg.synth.synthetic_batch(4096, 1234, DEV)
    return m, x, y


@pytest.mark.parametrize("compute_dtype", ["fp32", "f16x3"])
def test_full_size_train_then_eval_vs_oracle(pkg, full, compute_dtype):
    """B = 4096, H = 1024 (BASELINE configs[1]) against the oracle, in the exact-fp32 arithmetic and in the arithmetic
    bench.py's headline runs (f16x3: fp16 operand planes, three MFMAs per product)."""
    m, x, y = full
    if compute_dtype != "fp32":
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=1024, num_stage=2, p_dropout=0.5, BN=True,
                            compute_dtype=compute_dtype).to(DEV)
    st = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    m.train().manual_seed(7, step=0)
    pred = m(x).reshape(-1, 17, 3)
    loss = pkg.mse_loss(pred, y)
    loss.backward()
    masks = [philox.dropout_keep_mask(7, 1, l, 4096, 1024, 0.5) for l in range(5)]
    p64, _ = orc.forward({k: v.copy() for k, v in st.items()}, x.cpu().numpy(), num_stage=2, train=True,
                         p_dropout=0.5, keep_masks=masks, dtype=np.float64)
    opred, cache = orc.forward(st, x.cpu().numpy(), num_stage=2, train=True, p_dropout=0.5, keep_masks=masks,
                               on_masks=_gpu_decisions(pkg, m, 5, 1024))
    _assert_decisions_consistent(cache)
    oloss, dpred = orc.mse_loss(opred, y.cpu().numpy().reshape(4096, -1))
    _assert_train_fwd(pred.detach().cpu().numpy(), opred, p64)
    _close(loss.item(), oloss, 2e-5, 0)
    ograds, _ = orc.backward(st, cache, dpred)
    got = _grads(m)
    for k, v in ograds.items():
        pre_bn_bias = k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias"
        scale = np.abs(ograds[k[:-4] + "weight"]).max() if pre_bn_bias else np.abs(v).max()
        # w1.weight = dz0^T x with sum_b dz0 = 0 (BatchNorm) and x ~ 0.5 +- 0.1: the 0.5 cancels,
        # which costs ~1.5 digits in any fp32 evaluation (reference fp32 vs oracle: 1e-4 at B=128)
        # (pre-BN biases: true gradient 0, both sides are the round-off of a 4096-term sum)
        _close(got[k] / scale, v / scale, 0, 2e-3 if pre_bn_bias else 1e-3 if k == "w1.weight" else 5e-5)
        if not pre_bn_bias:
            _close(np.linalg.norm(got[k].astype(np.float64)), np.linalg.norm(v.astype(np.float64)), 1e-3, 0)
    # running statistics moved exactly as nn.BatchNorm1d moves them
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    for k in st:
        if "running" in k:
            _close(sd[k], st[k], 1e-5, 1e-6)
        if "num_batches" in k:
            assert int(sd[k]) == int(st[k]) == 1
    # eval forward on the updated statistics: the BASELINE.json gate at the bench size
    m.eval()
    with torch.no_grad():
        ye = m(x).cpu().numpy()
    yo, _ = orc.forward(st, x.cpu().numpy(), num_stage=2, train=False)
    assert orc.mpjpe_mm(ye, yo) < 1e-3
    yo64, _ = orc.forward(st, x.cpu().numpy(), num_stage=2, train=False, dtype=np.float64)
    assert orc.mpjpe_mm(ye, yo64) < 1e-3


def test_full_size_properties(pkg, full):
    m, x, y = full
    m.eval()
    with torch.no_grad():
        y_all = m(x)
        y_head = m(x[:1000])
        y_one = m(x[77:78])
        y_ten = m(x[70:80])
        y_300 = m(x[:300])
    # rows are independent and the contraction order does not depend on the batch: bit-exact -- within each of the two
    # contraction orders the library has: the 128x128-tile kernels (whole tiles and ragged M > 512, sequential K) and the
    # small-batch split-K kernels (ragged M <= 512, gemm_thin.hip: K slices fixed by (N, K) alone); across the two the
    # outputs agree to fp32 rounding
    assert torch.equal(y_all[:1000], y_head)
    assert torch.equal(y_ten[7:8], y_one) and torch.equal(y_300[77:78], y_one)
    assert (y_all[77:78] - y_one).abs().max() <= 1e-5 * y_all.abs().max()
    # training step is bitwise reproducible for a fixed (seed, step)
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
    # BatchNorm really normalised over the batch: column mean 0 / variance 1 of z-hat
    ws = m.last_workspace
    z = m.workspace_view(ws, 0, 2).double()
    mean, rstd = m.workspace_view(ws, 3, 2).double(), m.workspace_view(ws, 4, 2).double()
    zhat = (z - mean) * rstd
    assert zhat.mean(0).abs().max() < 1e-5 and (zhat.var(0, unbiased=False) - 1).abs().max() < 1e-3
    # dropout keeps ~half, and survivors are scaled by exactly 2
    bits = pkg.layout.unpack_bitmap(m.workspace_view(ws, 2, 1).cpu().numpy().view(np.uint64), 1024)
    act = m.workspace_view(ws, 1, 1).cpu().numpy()
    assert ((act > 0) == bits).all() and 0.2 < bits.mean() < 0.3
    # gradient accumulation across two backward calls equals twice one call
    m.train().manual_seed(11, step=3)
    m.zero_grad(set_to_none=True)
    for _ in range(2):
        m.manual_seed(11, step=3)
        pkg.mse_loss(m(x).reshape(-1, 17, 3), y).backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    m.zero_grad(set_to_none=True)
    m.manual_seed(11, step=3)
    pkg.mse_loss(m(x).reshape(-1, 17, 3), y).backward()
    g1 = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(g2, 2 * g1, rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------------------- ragged / odd shapes
@pytest.mark.parametrize("B,H,S,i_dim,o_dim,bn,dtype", [
    (2, 64, 2, 34, 51, True, "fp32"),        # smallest batch BatchNorm accepts
    (3, 128, 1, 34, 51, True, "fp32"),
    (63, 128, 2, 34, 51, True, "fp32"),      # one short statistics group
    (64, 1024, 2, 34, 51, True, "fp32"),     # the reference's own batch (train_1.py:194): one launch per hidden layer and
    (37, 256, 2, 34, 51, True, "fp32"),      #   direction (small_layer.hip), ragged rows,
    (2, 256, 1, 34, 51, True, "fp32"),       #   two rows,
    (5, 256, 1, 34, 51, True, "fp32"),
    (32, 256, 1, 300, 70, True, "fp32"),     #   300 inputs, 70 outputs: first / output layer NOT on the layer kernels (row-format
                                             #   bitmap for layer 0, tile format above it; generic GEMMs at both ends)
    (16, 512, 3, 51, 34, True, "f16x3"),     #   three stages, the other K split; f16x3: the forward launches contract on
    (64, 1024, 2, 34, 51, True, "f16x3"),    #   fp16 planes (three MFMAs per product) written by the launch before
    (65, 256, 2, 34, 51, True, "fp32"),      # 64 + 1 rows
    (100, 64, 2, 51, 34, True, "fp32"),      # the phase5 projector LinearModel(51, 34, linear_size=64)
    (129, 1024, 2, 34, 51, True, "fp32"),    # full width, ragged rows: whole-tile GEMMs fall back to edge path
    (1000, 128, 3, 34, 51, False, "fp32"),   # no BatchNorm, three stages
    (4097, 1024, 2, 34, 51, True, "fp32"),   # max bench size + 1
    (200, 36, 0, 20, 7, True, "fp32"),       # generic dims: nothing specialised applies
    (128, 1024, 2, 34, 51, True, "f16x3"),   # whole tiles, 128 ... 512 rows: the forward Linears on the layer kernels'
    (384, 256, 2, 34, 51, True, "f16x3"),    #   contraction (launch_small_linear_stats), everything else on the planes path
    (512, 512, 1, 51, 34, True, "f16x3"),
    (384, 256, 2, 34, 51, True, "bf16"),     # bf16 arithmetic, whole tiles
    (300, 256, 2, 34, 51, True, "bf16"),     # bf16 requested but ragged -> fp32 edge arithmetic
])
def test_ragged_shapes_vs_oracle(pkg, B, H, S, i_dim, o_dim, bn, dtype):
    torch.manual_seed(B + H)
    m = pkg.LinearModel(i_dim, o_dim, linear_size=H, num_stage=S, p_dropout=0.5, BN=bn,
                        compute_dtype=dtype).to(DEV).train()
    st = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, i_dim, generator=g).to(DEV).requires_grad_(True)
    t = (torch.rand(B, o_dim, generator=g) - 0.5).to(DEV)
    m.manual_seed(17, step=2)
    pred = m(x)
    loss = pkg.mse_loss(pred, t)
    loss.backward()
    L = 1 + 2 * S
    masks = [philox.dropout_keep_mask(17, 3, l, B, H, 0.5) for l in range(L)]
    xn = x.detach().cpu().numpy()
    opred, cache = orc.forward(st, xn, num_stage=S, train=True, use_bn=bn, p_dropout=0.5, keep_masks=masks,
                               on_masks=_gpu_decisions(pkg, m, L, H))
    bf = dtype == "bf16" and B % 128 == 0
    for c in cache["layers"]:
        d = c["on_disagree"]
        assert d.size <= (0.02 if bf else 1e-3) * c["z"].size + 2 and (d.size == 0 or d.max() < (0.2 if bf else 1e-3))
    oloss, dpred = orc.mse_loss(opred, t.cpu().numpy())
    ograds, odx = orc.backward(st, cache, dpred)
    tol = 8e-2 if bf else 2e-4      # bf16-sized: the worst tensor (a BatchNorm weight gradient) sits at 4-7 %
    scale = np.abs(opred).max()
    # (two rows of 256 columns: BatchNorm maps z0, z1 to +-d / sqrt(d^2 + eps) with d = (z0 - z1) / 2 -- where d is small the
    #  round-off of z is amplified by up to 1 / sqrt(eps) = 316, once per BatchNorm on the way: the stand-alone kernels sit at
    #  7e-5 on this case's forward, too; the case is here for the two-row code paths, (5, 256) for their accuracy)
    two_rows = 30.0 if (B == 2 and H >= 256) else 1.0
    tol *= two_rows
    _close(pred.detach().cpu().numpy() / scale, opred / scale, 0, 3e-2 if bf else 2e-5 * two_rows)
    _close(loss.item(), oloss, 2e-2 if bf else 2e-5 * two_rows, 0)
    got = _grads(m)
    for k, v in ograds.items():
        if bn and k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias":
            continue
        rel = np.linalg.norm((got[k] - v).astype(np.float64)) / (np.linalg.norm(v.astype(np.float64)) + 1e-30)
        assert rel < tol, (k, rel)
    rel = np.linalg.norm((x.grad.cpu().numpy() - odx).astype(np.float64)) / np.linalg.norm(odx.astype(np.float64))
    assert rel < tol, ("dx", rel)
    # eval forward on the same ragged batch
    m.eval()
    with torch.no_grad():
        ye = m(x.detach()).cpu().numpy()
    st2 = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    yo, _ = orc.forward(st2, xn, num_stage=S, train=False, use_bn=bn)
    _close(ye / scale, yo / scale, 0, 3e-2 if bf else 2e-5 * two_rows)


# ---------------------------------------------------------------------------- bf16 arithmetic mode
def test_bf16_mode_train_and_eval_vs_oracle(pkg):
    """PL_BF16: the 1024-wide GEMMs round their operands to bf16 (fp32 accumulate, fp32 storage).
    It is the throughput mode of BASELINE configs[1]; it cannot meet the 1e-3 mm gate (SURVEY 7.2:
    bf16 forward differs by ~1.2 mm) -- the tolerances here are bf16-sized and the test also checks
    that the mode really changes the arithmetic."""
    torch.manual_seed(2)
    H, B = 256, 512                      # whole 128x128 tiles -> the bf16 MFMA path is taken
    m32 = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.5).to(DEV).train()
    m16 = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.5, compute_dtype="bf16").to(DEV).train()
    m16.load_state_dict(m32.state_dict())
    st = {k: v.detach().cpu().numpy().copy() for k, v in m32.state_dict().items()}
    x, y = pkg.synth.synthetic_batch(B, 9, DEV)
    outs = {}
    for name, m in (("f32", m32), ("bf16", m16)):
        m.manual_seed(5, step=0)
        pred = m(x).reshape(B, 17, 3)
        pkg.mse_loss(pred, y).backward()
        outs[name] = (pred.detach().cpu().numpy(), _grads(m))
    masks = [philox.dropout_keep_mask(5, 1, l, B, H, 0.5) for l in range(5)]
    opred, _ = orc.forward({k: v.copy() for k, v in st.items()}, x.cpu().numpy(), num_stage=2, train=True,
                           p_dropout=0.5, keep_masks=masks)
    e16 = orc.mpjpe_mm(outs["bf16"][0], opred)
    e32 = orc.mpjpe_mm(outs["f32"][0], opred)
    assert e32 < 2e-2 and 1e-2 < e16 < 20.0, (e32, e16)          # bf16-sized, and really bf16
    # gradients: oracle evaluated on the bf16 path's own ReLU decisions (bf16 rounding moves
    # pre-activations by ~1e-2, so far more of them change side than in fp32)
    opred2, cache = orc.forward(st, x.cpu().numpy(), num_stage=2, train=True, p_dropout=0.5, keep_masks=masks,
                                on_masks=_gpu_decisions(pkg, m16, 5, H))
    for c in cache["layers"]:
        assert c["on_disagree"].size < 0.02 * c["z"].size and (c["on_disagree"].size == 0 or c["on_disagree"].max() < 0.2)
    _, dpred = orc.mse_loss(opred2, y.cpu().numpy().reshape(B, -1))
    ograds, _ = orc.backward(st, cache, dpred)
    for k, v in ograds.items():
        if k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias":
            continue                                                # zero-true-gradient biases: noise
        got = outs["bf16"][1][k]
        rel = np.linalg.norm((got - v).astype(np.float64)) / (np.linalg.norm(v.astype(np.float64)) + 1e-30)
        assert rel < 8e-2, (k, rel)          # bf16-sized (worst tensor here: a BatchNorm weight gradient, 4-7 %)
    m16.eval()
    with torch.no_grad():
        ye = m16(x).cpu().numpy()
    st2 = {k: v.detach().cpu().numpy() for k, v in m16.state_dict().items()}
    yo, _ = orc.forward(st2, x.cpu().numpy(), num_stage=2, train=False)
    assert orc.mpjpe_mm(ye, yo) < 20.0


def test_eval_forward_is_graph_capturable(pkg):
    """The library only enqueues on the caller's stream (no allocation, no sync), so a forward can be
    captured into a HIP graph and replayed on new inputs."""
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, linear_size=256).to(DEV).eval()
    x = torch.rand(512, 17, 2, device=DEV)
    with torch.no_grad():
        want = m(x).clone()                               # also allocates the workspace up front
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(x)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        xs = x.clone()
        with torch.cuda.graph(g):
            ys = m(xs)
        x2 = torch.rand(512, 17, 2, device=DEV)
        xs.copy_(x2)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(ys, m(x2))
        xs.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(ys, want)


@pytest.mark.parametrize("B,H,S", [(64, 1024, 2), (37, 256, 2), (9, 512, 0)])
def test_small_batch_fused_step_vs_autograd_route(pkg, monkeypatch, B, H, S):
    """B <= 64 (the reference's batch, train_1.py:194): the fused step runs one launch per hidden layer and direction and has
    no launch for the output Linear (its slabs come from the last hidden layer's launch, csrc/small_layer.hip); the autograd
    route runs the output layer and the loss as kernels of their own.  Same step to round-off: loss, prediction, every
    gradient, the BatchNorm running statistics -- and the device step counter ticks once per step under graph replay."""
    results = []
    for fused in (True, False):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=H, num_stage=S, p_dropout=0.5).to(DEV).train()
        m.manual_seed(3)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        if not fused:
            monkeypatch.setattr(pkg.train, "_fusable", lambda *a: False)
        x, y = pkg.synth.synthetic_batch(B, 40, DEV)
        loss, pred = pkg.train_step(m, opt, x, y)
        results.append((loss.clone(), pred.clone(), m.flat_grads.clone(), m._bn_running.clone(), m))
    (l0, p0, g0, r0, m0), (l1, p1, g1, r1, _) = results
    assert abs(l0.item() - l1.item()) <= 2e-6 * abs(l1.item())
    assert (p0 - p1).abs().max().item() <= 1e-5 * p1.abs().max().item()
    assert torch.equal(r0, r1)                                   # (the hidden layers run the same kernels on both routes)
    for sl, prm in zip(m0._slots, m0._param_list):
        a, b = g0[sl.offset:sl.offset + sl.numel], g1[sl.offset:sl.offset + sl.numel]
        if b.norm() > 1e-6 * g1.norm():                          # (pre-BatchNorm biases: true gradient 0)
            assert (a - b).norm() <= 2e-5 * b.norm(), (sl.name, ((a - b).norm() / b.norm()).item())
    # captured: the replayed steps draw new dropout masks and step Adam on (the device counter ticks in the backward's top
    # launch at this size): three replays follow three eager steps of a twin model
    monkeypatch.undo()
    losses = []
    for graphed in (True, False):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=H, num_stage=S, p_dropout=0.5).to(DEV).train()
        m.manual_seed(3)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        x, y = pkg.synth.synthetic_batch(B, 40, DEV)
        step = pkg.GraphedTrainStep(m, opt, x, y) if graphed else (lambda a, b: pkg.train_step(m, opt, a, b))
        losses.append([step(x, y)[0].item() for _ in range(3)])
    assert len(set(losses[0])) == 3
    np.testing.assert_allclose(losses[0], losses[1], rtol=1e-5)


@pytest.mark.parametrize("B,H,S,dtype", [(64, 1024, 2, "fp32"), (20, 256, 1, "fp32"), (9, 512, 0, "fp32"), (64, 1024, 2, "f16x3")])
def test_small_batch_adamw_inside_the_backward_launches_is_bitwise_the_separate_launch(pkg, B, H, S, dtype):
    """pl_lifter_train_step at B <= 64: each backward launch carries a slice of the AdamW step on spare workgroups and one
    small launch updates the bottom of the arena.  Against fused_train_fwd_bwd + optimizer.step() (one AdamW launch over the
    whole arena): the same kernels produce the gradients, the element arithmetic is the same -- parameters, both moment
    arenas and the losses of three steps are equal bit for bit."""
    out = []
    for in_call in (True, False):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=H, num_stage=S, p_dropout=0.5, compute_dtype=dtype).to(DEV).train()
        m.manual_seed(3)
        opt = pkg.FlatAdamW(m, lr=1e-3, weight_decay=0.02)
        assert m.step_carries_adamw(B) and not m.step_carries_adamw(128)
        losses = []
        for i in range(3):
            x, y = pkg.synth.synthetic_batch(B, 40 + i, DEV)
            if in_call:
                loss, _ = pkg.train_step(m, opt, x, y)
            else:
                loss, _ = m.fused_train_fwd_bwd(x.reshape(B, -1).contiguous(), y.reshape(B, -1).contiguous())
                opt.step()
            losses.append(loss.clone())
        assert opt.state_dict()["state"][0]["step"] == 3
        out.append((torch.stack(losses), m.flat_params.clone(), opt._m.clone(), opt._v.clone(), m.flat_grads.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert not torch.equal(out[0][1], pkg.LinearModel(34, 51, linear_size=H, num_stage=S, compute_dtype=dtype).to(DEV).flat_params)


@pytest.mark.parametrize("dtype", ["fp32", "f16x3"])
def test_train_step_entry_point_at_a_large_batch_is_fwd_bwd_plus_one_adamw_launch(pkg, dtype):
    """pl_lifter_train_step where no backward launch has room for the optimizer (B = 256: tile GEMMs on every CU): forward,
    loss, backward, then ONE pl_adamw_flat launch -- bitwise fused_train_fwd_bwd + optimizer.step(), and an evaluation on the
    planes path afterwards sees the new weights (the call leaves the persistent weight planes stale and says so)."""
    out = []
    for in_call in (True, False):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.5, compute_dtype=dtype).to(DEV).train()
        m.manual_seed(3)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        assert not m.step_carries_adamw(256)
        x, y = pkg.synth.synthetic_batch(256, 41, DEV)
        x2, y2 = x.reshape(256, -1).contiguous(), y.reshape(256, -1).contiguous()
        for i in range(2):
            if in_call:
                loss, _ = m.fused_train_fwd_bwd(x2, y2, adamw=opt._step_struct(1e-3, None, opt._t + 1, None))
                opt._advance_host(1)
            else:
                loss, _ = m.fused_train_fwd_bwd(x2, y2)
                opt.step()
        m.eval()
        with torch.no_grad():
            ye = m(x).clone()
        out.append((loss.clone(), m.flat_params.clone(), opt._m.clone(), opt._v.clone(), ye))
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_fused_train_step_is_bitwise_the_autograd_route(pkg, monkeypatch):
    """train_step's fast path (pl_lifter_train_fwd_bwd + pl_adamw_flat) and the autograd route
    (LinearModel.forward -> mse_loss -> backward -> optimizer.step) run the same kernels."""
    results = []
    for fused in (True, False):
        torch.manual_seed(0)
        m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.5).to(DEV).train()
        m.manual_seed(3)
        opt = pkg.FlatAdamW(m, lr=1e-3)
        if not fused:
            monkeypatch.setattr(pkg.train, "_fusable", lambda *a: False)
        losses = []
        for i in range(3):
            x, y = pkg.synth.synthetic_batch(384, 40 + i, DEV)
            loss, pred = pkg.train_step(m, opt, x, y)
            losses.append(loss.clone())
        results.append((torch.stack(losses), pred.clone(), m.flat_params.clone(), m.flat_grads.clone(),
                        m._bn_running.clone(), opt._m.clone()))
        assert m.w2.weight.grad is not None and m.w2.weight.grad.data_ptr() == \
            m.flat_grads.data_ptr() + 4 * m._slots[-2].offset
        assert opt.state_dict()["state"][0]["step"] == 3
    for a, b in zip(*results):
        assert torch.equal(a, b)


def test_phase5_cycle_usage_pattern_vs_torch_twin(pkg):
    """phase5_loop/train_5 copy.py:164-168,190-199: the lifter is called TWICE inside one autograd graph
    (on a prediction that needs its input gradient, and on the ground truth), L1 losses, one backward;
    the projector is LinearModel(51, 34, linear_size=64).  Checked against the stock-PyTorch twin on
    CPU with the same weights (p_dropout = 0: torch's dropout stream cannot be matched)."""
    from oracle.torch_twin import TwinLifter
    torch.manual_seed(11)
    B = 160
    for (i_dim, o_dim, H) in ((34, 51, 256), (51, 34, 64)):
        m = pkg.LinearModel(i_dim, o_dim, linear_size=H, p_dropout=0.0).to(DEV).train()
        tw = TwinLifter(i_dim, o_dim, linear_size=H, p_dropout=0.0).train()
        tw.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
        pred_in = torch.rand(B, i_dim)                 # stand-in for Model_2D(frame): needs a gradient
        gt_in, tgt_a, tgt_b = torch.rand(B, i_dim), torch.rand(B, o_dim) - 0.5, torch.rand(B, o_dim) - 0.5
        outs = []
        for model, dev in ((m, DEV), (tw, "cpu")):
            a = pred_in.clone().to(dev).requires_grad_(True)
            l1 = torch.nn.L1Loss()
            loss = l1(model(a), tgt_a.to(dev)) + l1(model(gt_in.to(dev)), tgt_b.to(dev))
            loss.backward()
            outs.append((loss.item(), a.grad.cpu().numpy(),
                         {k: p.grad.cpu().numpy() for k, p in model.named_parameters()},
                         {k: v.cpu().numpy() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}))
        (lg, dxg, gg, bg), (lc, dxc, gc, bc) = outs
        _close(lg, lc, 1e-5, 0)
        sdx = np.abs(dxc).max()
        _close(dxg / sdx, dxc / sdx, 0, 5e-4)
        for k, v in gc.items():
            if k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias":
                continue
            rel = np.linalg.norm((gg[k] - v).astype(np.float64)) / (np.linalg.norm(v.astype(np.float64)) + 1e-30)
            assert rel < 2e-3, (k, rel)       # L1's sign() gradient flips on ties: norm-wise comparison
        for k, v in bc.items():               # two forwards -> two running-stat updates, as in torch
            _close(bg[k], v, 1e-5, 1e-6)


def test_flip_pose_vs_oracle(pkg):
    rng = np.random.default_rng(4)
    for D in (2, 3):
        a = rng.random((77, 17, D)).astype(np.float32)
        got = pkg.flip_pose(_t(a)).cpu().numpy()
        assert np.array_equal(got, orc.flip_pose(a))
        assert np.allclose(pkg.flip_pose(_t(got)).cpu().numpy(), a, atol=1e-7)      # an involution
    with pytest.raises(ValueError):
        pkg.flip_pose(torch.zeros(4, 16, 3, device=DEV))


def test_flip_tta_eval_vs_oracle(pkg):
    """(flip(model(flip(x))) + model(x)) / 2 in one 2B-row forward == the oracle's two passes."""
    torch.manual_seed(9)
    m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.5).to(DEV)
    x, y = pkg.synth.synthetic_batch(200, 6, DEV)
    m.train()
    for _ in range(3):                                     # non-trivial running statistics
        m(x)
    m.eval()
    st = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    xn = x.cpu().numpy()
    plain, _ = orc.forward(st, xn, num_stage=2, train=False)
    flipped, _ = orc.forward(st, orc.flip_pose(xn), num_stage=2, train=False)
    want = (orc.flip_pose(flipped.reshape(-1, 17, 3)) + plain.reshape(-1, 17, 3)) / 2
    got = pkg.predict_flip_tta(m, x)
    assert orc.mpjpe_mm(got.cpu().numpy().reshape(-1, 51), want.reshape(-1, 51)) < 1e-3
    with torch.no_grad():                                  # (with autograd on, eval mode runs the saved-state forward)
        two_pass = (pkg.flip_pose(m(pkg.flip_pose(x)).reshape(-1, 17, 3)) + m(x).reshape(-1, 17, 3)) / 2
    assert torch.equal(got, two_pass)                      # one 2B forward == two B forwards, bitwise
    loss, metric, y_hat = pkg.eval_step(m, x, y, flip=True)
    assert torch.equal(y_hat, got)
    _close(loss.item(), float(np.mean((want - y.cpu().numpy()) ** 2)), 1e-5, 0)
    with pytest.raises(ValueError):
        pkg.predict_flip_tta(m.train(), x)


@pytest.mark.parametrize("dtype", ["bf16x6", "f16x3"])
def test_bf16x6_mode_is_fp32_grade(pkg, dtype):
    """PL_BF16X6 / PL_F16X3 at the bench size: train fwd/bwd against the oracle with the fp32 tolerances."""
    torch.manual_seed(0)
    B, H = 1024, 1024
    m = pkg.LinearModel(34, 51, linear_size=H, p_dropout=0.5, compute_dtype=dtype).to(DEV).train()
    st = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    x, y = pkg.synth.synthetic_batch(B, 31, DEV)
    m.manual_seed(8, step=0)
    pred = m(x).reshape(B, 17, 3)
    pkg.mse_loss(pred, y).backward()
    masks = [philox.dropout_keep_mask(8, 1, l, B, H, 0.5) for l in range(5)]
    p64, _ = orc.forward({k: v.copy() for k, v in st.items()}, x.cpu().numpy(), num_stage=2, train=True,
                         p_dropout=0.5, keep_masks=masks, dtype=np.float64)
    opred, cache = orc.forward(st, x.cpu().numpy(), num_stage=2, train=True, p_dropout=0.5, keep_masks=masks,
                               on_masks=_gpu_decisions(pkg, m, 5, H))
    _assert_decisions_consistent(cache)
    _assert_train_fwd(pred.detach().cpu().numpy(), opred, p64)
    _, dpred = orc.mse_loss(opred, y.cpu().numpy().reshape(B, -1))
    ograds, _ = orc.backward(st, cache, dpred)
    got = _grads(m)
    for k, v in ograds.items():
        pre_bn_bias = k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias"
        scale = np.abs(ograds[k[:-4] + "weight"]).max() if pre_bn_bias else np.abs(v).max()
        _close(got[k] / scale, v / scale, 0, 2e-3 if pre_bn_bias else 1e-3 if k == "w1.weight" else 5e-5)
    m.eval()
    with torch.no_grad():
        ye = m(x).cpu().numpy()
    st2 = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    yo64, _ = orc.forward(st2, x.cpu().numpy(), num_stage=2, train=False, dtype=np.float64)
    assert orc.mpjpe_mm(ye, yo64) < 1e-3


# ---------------------------------------------------------------------------- API behaviour
def test_errors_are_loud(pkg):
    m = pkg.LinearModel(34, 51, linear_size=64).to(DEV).train()
    with pytest.raises(pkg.PoseliftError, match="more than 1 value per channel"):
        m(torch.rand(1, 17, 2, device=DEV))
    with pytest.raises(pkg.PoseliftError, match="no CPU path"):
        m(torch.rand(4, 17, 2))
    with pytest.raises(ValueError):
        m(torch.rand(4, 16, 2, device=DEV))
    bad = pkg.LinearModel(34, 51, linear_size=64).to(DEV).eval()
    bad._desc.dtype = 7
    with pytest.raises(pkg.PoseliftError, match="PLDtype"):
        bad(torch.rand(4, 17, 2, device=DEV))
    m.eval()
    out = m(torch.rand(4, 17, 2, device=DEV))              # eval with grad enabled: a differentiable forward
    out.sum().backward(retain_graph=True)
    with pytest.raises(pkg.PoseliftError, match="second backward"):
        out.sum().backward()                               # its workspace was handed back: loud, not stale data
    d = pkg._lib.PLDesc(in_dim=34, hidden=63, out_dim=51, num_stage=2, bn=1, dtype=0, p_dropout=0.5,
                        bn_eps=1e-5, bn_momentum=0.1)
    assert pkg.lib().pl_workspace_bytes(ctypes.byref(d), 8) == 0
    assert b"multiple of 4" in pkg.lib().pl_last_error()


def test_checkpoint_envelope_roundtrip_and_stock_optimizer(pkg, tmp_path):
    """train_1.py:186 saves {'epoch','batch_size','model','optimizer'}; :43-46 resumes."""
    torch.manual_seed(1)
    m = pkg.LinearModel(34, 51, linear_size=64, p_dropout=0.0).to(DEV).train()
    stock = torch.optim.AdamW(m.parameters(), lr=1e-4)      # the reference's optimiser works on the views
    ours_m = pkg.LinearModel(34, 51, linear_size=64, p_dropout=0.0).to(DEV).train()
    ours_m.load_state_dict(m.state_dict())
    ours = pkg.FlatAdamW(ours_m, lr=1e-4)
    x, y = pkg.synth.synthetic_batch(64, 2, DEV)
    for _ in range(2):
        l1, _ = pkg.train_step(m, stock, x, y)
        l2, _ = pkg.train_step(ours_m, ours, x, y)
        _close(l1.item(), l2.item(), 1e-6, 0)
    for (k, a), (_, b) in zip(m.state_dict().items(), ours_m.state_dict().items()):
        if k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias" or k.endswith("running_mean"):
            continue
        _close(a.cpu().numpy(), b.cpu().numpy(), 1e-5, 1e-7)
    path = tmp_path / "ckpt.pt"
    torch.save({"epoch": 3, "batch_size": 64, "model": ours_m.state_dict(), "optimizer": ours.state_dict()}, path)
    ck = torch.load(path, weights_only=True)
    fresh = pkg.LinearModel(34, 51, linear_size=64, p_dropout=0.0).to(DEV)
    fresh.load_state_dict(ck["model"])
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), ours_m.state_dict().values()))
    assert fresh._arenas_intact()


def test_reduce_lr_on_plateau_drives_the_flat_optimizer(pkg):
    """train_1.py:41,106: ReduceLROnPlateau(factor .7, patience 3, cooldown 2, min_lr 5e-6) stepped with the
    last batch's train loss.  FlatAdamW is a torch Optimizer: the scheduler rewrites param_groups[0]['lr']
    and the next kernel launch must use it (checked against the oracle's AdamW at the scheduled rate)."""
    torch.manual_seed(2)
    m = pkg.LinearModel(34, 51, linear_size=64, p_dropout=0.0).to(DEV).train()
    opt = pkg.FlatAdamW(m, lr=1e-2)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.7, patience=3, cooldown=2, min_lr=5e-6)
    x, y = pkg.synth.synthetic_batch(64, 5, DEV)
    pkg.train_step(m, opt, x, y)
    for _ in range(5):                     # a loss that never improves: patience 3 -> one reduction
        sched.step(1.0)
    assert abs(opt.param_groups[0]["lr"] - 0.7e-2) < 1e-12
    before = m.flat_params.clone()
    names = [k for k, _ in m.named_parameters()]
    p0 = {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
    m0 = {k: opt.state[p]["exp_avg"].cpu().numpy().copy() for k, p in m.named_parameters()}
    v0 = {k: opt.state[p]["exp_avg_sq"].cpu().numpy().copy() for k, p in m.named_parameters()}
    pkg.train_step(m, opt, x, y)
    for k, p in m.named_parameters():
        want, _, _ = orc.adamw_step(p0[k], p.grad.cpu().numpy(), m0[k], v0[k], 2, lr=0.7e-2)
        _close(p.detach().cpu().numpy(), want, 1e-5, 1e-7)
    assert names and not torch.equal(before, m.flat_params)


@pytest.mark.parametrize("resident", [True, False])
def test_pose_feeder_batches_are_the_permuted_rows(pkg, resident):
    """data.PoseFeeder (the DataLoader hop of train_1.py:26-31,75-81): resident tables + device
    gather, or pinned tables + prefetched H2D -- the same batches, each pose once per epoch."""
    rng = np.random.default_rng(3)
    N, bs = 1000, 96
    x = rng.random((N, 17, 2))                       # float64 on purpose: the feeder does .float()
    y = rng.standard_normal((N, 17, 3)).astype(np.float32)
    fd = pkg.PoseFeeder(x, y, bs, device=DEV, seed=7, resident=resident)
    assert len(fd) == 11
    for epoch in (0, 1):
        fd.set_epoch(epoch)
        want = pkg.epoch_indices(N, bs, epoch=epoch, seed=7)
        seen = 0
        for (a, b), idx in zip(fd, want):
            assert a.shape == (idx.numel(), 17, 2) and b.shape == (idx.numel(), 17, 3) and a.dtype == torch.float32
            assert np.array_equal(a.cpu().numpy(), x[idx.numpy()].astype(np.float32))
            assert np.array_equal(b.cpu().numpy(), y[idx.numpy()])
            seen += idx.numel()
        assert seen == N
    # feeds the train step directly; two DP ranks cut each global batch in two
    m = pkg.LinearModel(34, 51, linear_size=64).to(DEV).train()
    opt = pkg.FlatAdamW(m, lr=1e-3)
    for a, b in pkg.PoseFeeder(x, y, 128, device=DEV, resident=resident, drop_last=True):
        loss, _ = pkg.train_step(m, opt, a, b)
    assert torch.isfinite(loss)
    r0 = [a for a, _ in pkg.PoseFeeder(x, y, 48, device=DEV, seed=7, rank=0, world=2, resident=resident)]
    r1 = [a for a, _ in pkg.PoseFeeder(x, y, 48, device=DEV, seed=7, rank=1, world=2, resident=resident)]
    fd.set_epoch(0)
    for (a, _), p, q in zip(fd, r0, r1):
        assert torch.equal(a, torch.cat([p, q]))
    with pytest.raises(pkg.PoseliftError):
        pkg.PoseFeeder(x, y, 8, device="cpu")


def test_bf16x6_loop_variants_are_bitwise_identical_over_a_train_step(pkg, tmp_path):
    """PL_BF16X6 has two equivalent main loops: split at staging into bf16 planes in LDS (the default,
    single launches and the backward dual launch) and the fragment-time split (POSELIFT_X6_FRAG=1).  Same
    split chain, same MFMA order: a training step must produce identical bits (the variant is chosen per
    process through the environment, so both run in child processes)."""
    import subprocess
    script = tmp_path / "run.py"
    script.write_text(
        "import importlib, sys, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "pkg = importlib.import_module('3d_poseestimation_amd')\n"
        "torch.manual_seed(3)\n"
        "m = pkg.LinearModel(34, 51, linear_size=256, p_dropout=0.5, compute_dtype='bf16x6').to('cuda:0').train()\n"
        "m.manual_seed(11)\n"
        "x, y = pkg.synth.synthetic_batch(512, 4, 'cuda:0')\n"
        "xr = x.clone().requires_grad_(True)\n"
        "out = m(xr)\n"
        "pkg.mse_loss(out.reshape(y.shape), y).backward()\n"
        "torch.save({'out': out.detach().cpu(), 'g': m.flat_grads.cpu(), 'dx': xr.grad.cpu()}, sys.argv[1])\n")
    res = {}
    for tag, env in (("planes", {}), ("fragment", {"POSELIFT_X6_FRAG": "1"})):
        out = tmp_path / f"{tag}.pt"
        subprocess.run([sys.executable, str(script), str(out)], check=True, env={**os.environ, **env}, timeout=300)
        res[tag] = torch.load(out, weights_only=True)
    for k in ("out", "g", "dx"):
        assert torch.equal(res["planes"][k], res["fragment"][k]), k


def test_triangle_loss_vs_reference_golden_and_oracle(pkg):
    """TriangleLoss on the HIP L1 kernel (all terms of a call in one launch pair).  era 'model2d' against
    phase5_loop/losses.py run as-is (golden g8: values and gradients, with and without the projector
    term); era 'lifter' (the LinearModel-era copy in train_5 copy.py, restated from its text) against the
    oracle; l1_loss against torch's L1Loss on the CPU."""
    g = load_golden("g8_triangle_loss.npz")
    ins = {k[3:]: g[k] for k in g if k.startswith("in:")}
    for tag, project in (("noproj", False), ("proj", True)):
        t = {k: _t(v).requires_grad_(k in ("p2d", "p3d", "lpred", "proj")) for k, v in ins.items()}
        out = pkg.TriangleLoss(Project=project, era="model2d")(t["p2d"], t["p3d"], t["lgt"], t["lpred"], t["g2d"],
                                                                t["g3d"], proj_3d_pred=t["proj"])
        out[0].backward()
        want = g[f"{tag}:losses"]
        _close(np.array([float(o.detach()) if torch.is_tensor(o) else float(o) for o in out]), want, 2e-6, 1e-7)
        for k in ("p2d", "p3d", "lpred") + (("proj",) if project else ()):
            _close(t[k].grad.cpu().numpy(), g[f"{tag}:grad:{k}"], 1e-5, 1e-9)
        assert project or t["proj"].grad is None
    # LinearModel-era variant: four terms, returns their sum; gradients also reach lift(y2d)
    t = {k: _t(v).requires_grad_(k in ("p2d", "p3d", "lpred", "lgt")) for k, v in ins.items()}
    crit = pkg.TriangleLoss(era="lifter")
    loss = crit(t["p2d"], t["p3d"], t["lgt"], t["lpred"], t["g2d"], t["g3d"])
    loss.backward()
    terms, grads = orc.triangle_loss(ins["p2d"], ins["p3d"], ins["lgt"], ins["lpred"], ins["g2d"], ins["g3d"], era="lifter")
    _close(loss.item(), sum(terms), 2e-6, 0)
    for k, name in (("p2d", "p2d"), ("p3d", "p3d"), ("lpred", "lift_pred"), ("lgt", "lift_gt")):
        _close(t[k].grad.cpu().numpy(), np.broadcast_to(grads[name], ins[k].shape), 1e-5, 1e-9)
    _close(np.array(crit.term_means()), np.array(terms), 2e-6, 0)
    a, b = torch.randn(1000, 51), torch.randn(1000, 51)
    _close(pkg.l1_loss(a.to(DEV), b.to(DEV)).item(), torch.nn.L1Loss()(a, b).item(), 2e-6, 0)
    with pytest.raises(pkg.PoseliftError):
        pkg.l1_loss(a, b)                      # CPU tensors: no fallback
