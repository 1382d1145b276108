"""Pin the CPU oracle against vectors recorded from the reference model
(tools/make_golden.py, which imports /root/reference/phase1_lifting/baselineModel.py).
Runs on CPU; nothing here reads /root/reference."""
import numpy as np
import pytest

from conftest import golden_state, load_golden
from oracle import lifter_oracle as orc
from oracle import philox


def _close(a, b, rtol, atol):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def _check_grads(grads, want, bn):
    """Every gradient to 2e-5 of its own max.  The bias of a Linear that feeds a
    training-mode BatchNorm has an exactly-zero true gradient (BN removes the column
    mean), so both sides hold round-off noise: compare it on the scale of that
    layer's weight gradient instead."""
    for k, v in want.items():
        scale = np.abs(v).max() + 1e-30
        if bn and k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias":
            scale = np.abs(want[k[:-4] + "weight"]).max()
        _close(grads[k] / scale, v / scale, 0, 2e-5)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        got = philox.philox4x32_10(*[np.array([v], np.uint64) for v in c], *k)
        assert tuple(int(g[0]) for g in got) == want


def test_dropout_mask_stream_properties():
    m = philox.dropout_keep_mask(1234, 5, 2, 257, 64, 0.5)
    assert m.shape == (257, 64) and 0.47 < m.mean() < 0.53
    assert philox.dropout_keep_mask(1234, 5, 2, 8, 64, 0.0).all()
    assert not philox.dropout_keep_mask(1234, 5, 2, 8, 64, 1.0).any()
    assert (m != philox.dropout_keep_mask(1234, 6, 2, 257, 64, 0.5)).any()
    assert (m != philox.dropout_keep_mask(1234, 5, 3, 257, 64, 0.5)).any()
    assert 0.22 < 1 - philox.dropout_keep_mask(9, 0, 0, 512, 64, 0.25).mean() < 0.28


def test_g1_eval_forward_full():
    g = load_golden("g1_eval_full.npz")
    st = orc.init_state(34, 51, 1024, 2, rng=np.random.default_rng(int(g["weight_seed"])), nontrivial_bn=True)
    y, _ = orc.forward(st, g["x"], num_stage=2, train=False)
    # parity gate of BASELINE.json: <= 1e-3 mm MPJPE against the reference forward
    assert orc.mpjpe_mm(y, g["y"]) < 1e-3
    y64, _ = orc.forward(st, g["x"], num_stage=2, train=False, dtype=np.float64)
    assert orc.mpjpe_mm(y64, g["y_fp64"]) < 1e-6
    assert orc.mpjpe_mm(y, g["y_fp64"]) < 1e-3


@pytest.mark.parametrize("tag", ["small", "nobn", "s3"])
def test_g2_train_nodrop(tag):
    g = load_golden(f"g2_train_nodrop_{tag}.npz")
    st = golden_state(g)
    S, bn = int(g["num_stage"]), bool(g["bn"])
    pred, cache = orc.forward(st, g["x"], num_stage=S, train=True, use_bn=bn, p_dropout=0.0)
    _close(pred.reshape(g["pred"].shape), g["pred"], 1e-4, 1e-5)
    loss, dpred = orc.mse_loss(pred, g["t"])
    _close(loss, g["loss"], 1e-5, 0)
    grads, dx = orc.backward(st, cache, dpred)
    sdx = np.abs(g["dx"]).max()
    _close(dx.reshape(g["dx"].shape) / sdx, g["dx"] / sdx, 0, 2e-5)
    want = golden_state(g, "grad:")
    assert set(want) <= set(grads)
    _check_grads(grads, want, bn)
    for k, v in golden_state(g, "after:").items():
        if "num_batches" in k:
            assert int(st[k]) == int(v)
        else:
            _close(st[k], v, 1e-5, 1e-7)


def test_g2_train_nodrop_full():
    g = load_golden("g2_train_nodrop_full.npz")
    st = orc.init_state(34, 51, 1024, 2, rng=np.random.default_rng(int(g["weight_seed"])), nontrivial_bn=True)
    pred, cache = orc.forward(st, g["x"], num_stage=2, train=True, p_dropout=0.0)
    _close(pred.reshape(g["pred"].shape), g["pred"], 1e-4, 1e-5)
    loss, dpred = orc.mse_loss(pred, g["t"])
    _close(loss, g["loss"], 1e-5, 0)
    grads, _ = orc.backward(st, cache, dpred)
    for k in orc.param_names(2):
        flat = grads[k].reshape(-1)
        norm = float(g["gnorm:" + k])
        pre_bn_bias = k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias"
        if pre_bn_bias:      # true gradient is exactly 0: noise on the layer's weight-grad scale
            scale = np.abs(g["gval:" + k[:-4] + "weight"]).max()
        else:
            scale = np.abs(g["gval:" + k]).max()
            _close(np.linalg.norm(flat.astype(np.float64)), norm, 1e-3, 0)
        # 5e-4 of max: a ReLU input within round-off of 0 may flip between two fp32
        # summation orders (the fp64 run of the same model differs by 5e-3 for that reason)
        _close(flat[g["gidx:" + k]] / scale, g["gval:" + k] / scale, 0, 5e-4)
    for k, v in golden_state(g, "after:").items():
        if "num_batches" in k:
            assert int(st[k]) == int(v)
        else:
            _close(st[k], v, 1e-5, 1e-6)


def test_g3_train_with_reference_masks():
    g = load_golden("g3_train_masks_small.npz")
    st = golden_state(g)
    masks = [m.astype(bool) for m in g["masks"]]
    pred, cache = orc.forward(st, g["x"], num_stage=2, train=True, p_dropout=0.5, keep_masks=masks)
    _close(pred.reshape(g["pred"].shape), g["pred"], 1e-4, 1e-5)
    loss, dpred = orc.mse_loss(pred, g["t"])
    _close(loss, g["loss"], 1e-5, 0)
    grads, _ = orc.backward(st, cache, dpred)
    _check_grads(grads, golden_state(g, "grad:"), True)


def test_g4_three_adamw_steps():
    g = load_golden("g4_adamw_small.npz")
    st = golden_state(g)
    opt = {"t": 0, "m": {}, "v": {}}
    for i in range(3):
        loss, _, _ = orc.train_step(st, opt, g["xs"][i], g["ts"][i], num_stage=2, p_dropout=0.0,
                                    lr=float(g["lr"]), wd=float(g["wd"]))
        _close(loss, g["losses"][i], 2e-5, 0)
    for k, v in golden_state(g, "final:").items():
        if "num_batches" in k:
            assert int(st[k]) == int(v)
        elif (k.endswith(".bias") and "batch_norm" not in k and k != "w2.bias") or k.endswith("running_mean"):
            # zero-true-gradient bias (see _check_grads): Adam normalises the round-off
            # noise, so each side random-walks by up to lr per step (and running_mean,
            # the EMA of mean(z) = mean(a W^T) + b, follows that bias)
            _close(st[k], v, 0, 2 * 3 * float(g["lr"]) + 1e-6)
        else:
            _close(st[k], v, 1e-5, 2e-7)


def test_g5_mpjpe():
    g = load_golden("g5_mpjpe.npz")
    metric = orc.loss_mpjpe(g["pred"], g["tgt"])
    _close(metric, g["metric"], 1e-5, 1e-6)
    assert abs(orc.epoch_mpjpe_mm(metric, 48) - float(g["epoch_mm"])) < 1e-2
    assert metric[0] == 0.0          # root joint is identically zero (H36_dataset.py:209-211)


def test_batchnorm_rejects_single_row():
    st = orc.init_state(34, 51, 64, 2, rng=np.random.default_rng(0))
    with pytest.raises(ValueError):
        orc.forward(st, np.zeros((1, 17, 2), np.float32), train=True, p_dropout=0.0)


def test_g8_triangle_loss_oracle_vs_reference():
    """oracle.triangle_loss (era 'model2d') against phase5_loop/losses.py run as-is: term values and the
    gradients of the sum w.r.t. every predicted tensor, with and without the projector term."""
    g = load_golden("g8_triangle_loss.npz")
    ins = {k[3:]: g[k] for k in g if k.startswith("in:")}
    for tag, project in (("noproj", False), ("proj", True)):
        terms, grads = orc.triangle_loss(ins["p2d"], ins["p3d"], ins["lgt"], ins["lpred"], ins["g2d"], ins["g3d"],
                                         proj_pred=ins["proj"], project=project)
        want = g[f"{tag}:losses"]
        assert np.allclose(sum(terms), want[0], rtol=1e-6)
        assert np.allclose(terms[:3], want[1:4], rtol=1e-6)
        if project:
            assert np.allclose(terms[3], want[4], rtol=1e-6)
        for name, key in (("p2d", "p2d"), ("p3d", "p3d"), ("lift_pred", "lpred"), ("proj", "proj")):
            got = np.broadcast_to(np.asarray(grads[name], np.float64), g[f"{tag}:grad:{key}"].shape)
            assert np.allclose(got, g[f"{tag}:grad:{key}"], rtol=1e-5, atol=1e-9), (tag, name)
