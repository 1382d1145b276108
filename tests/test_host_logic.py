"""CPU tests of the host side: the C-ABI library loads and exports every symbol declared in
include/poselift.h, the Python arena layout agrees with the C one, the module mirrors the
reference's interface (names, order, shapes), and nothing computes without a GPU."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_state, load_golden


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    return ge.build()          # compiles libposelift.so for gfx950 if missing (no GPU needed)


def test_library_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, "include", "poselift.h")).read()
    declared = set(re.findall(r"\b(pl_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 18
    raw = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/poselift.h but not exported"
    assert declared == set(pkg._lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert pkg.lib().pl_version() == int(re.search(r"#define PL_VERSION (\d+)", header).group(1)) >= 105
    conv = importlib.import_module("3d_poseestimation_amd.conv")
    assert pkg.lib().pl_conv_act_plane_scale() == conv.ACT_PLANE_SCALE      # one constant on both sides of the C ABI


def test_entry_points_reject_bad_arguments_before_touching_a_device(pkg):
    """Argument validation of the conv-path / head entry points returns an error code and a message before any HIP
    call -- checkable here without a GPU (null pointers, bad geometry, bad arithmetic codes), and the scratch-size
    helpers are pure host functions."""
    L = pkg.lib()
    one = ctypes.c_void_p(16)            # a non-null, 16-byte aligned dummy: never dereferenced on these paths
    assert L.pl_conv2d_nhwc_fwd(None, 1, 8, 8, 32, one, 64, 3, 3, 1, 1, None, None, None, 0, None, one, 2, None, 0, None) != 0
    assert b"null" in L.pl_last_error()
    assert L.pl_conv2d_nhwc_fwd(one, 1, 8, 8, 32, one, 64, 3, 3, 1, 1, None, None, None, 0, None, one, 9, None, 0, None) != 0
    assert b"arith" in L.pl_last_error()
    assert L.pl_conv2d_nhwc_wgrad(one, 1, 8, 8, 32, one, 64, 3, 3, 1, 1, one, 9, None, 0, None) != 0
    assert b"arith" in L.pl_last_error()
    assert L.pl_maxpool3x3s2_nhwc_idx(one, 1, 8, 8, 6, one, one, None) != 0          # C % 4
    assert L.pl_maxpool3x3s2_nhwc_bwd_idx(None, one, 1, 8, 8, 8, one, None) != 0
    assert L.pl_softargmax3d_nhwc_bwd(one, one, one, 0, 17, 64, 64, one, None) != 0  # B = 0
    assert L.pl_softargmax3d_nhwc_bwd(one, one, one, 1, 17, 64, 64, ctypes.c_void_p(20), None) != 0
    assert b"aligned" in L.pl_last_error()
    assert L.pl_nhwc_to_nchw(one, 1, 16, 8, one, None) != 0                          # in == out
    # scratch sizes: the stem's weight gradient needs its per-workgroup partials, a 1x1 nothing unless it is split
    assert L.pl_conv2d_nhwc_wgrad_scratch_bytes(2, 64, 64, 3, 64, 7, 7, 2, 3) == 512 * 64 * 147 * 4
    assert L.pl_conv2d_nhwc_scratch_bytes(64, 64, 64, 256, 1024, 1, 1, 1, 0) == 0       # 2048 x 8 tiles: no split
    # the 7x7 stem: scratch-free with the epilogue its own kernel folds, im2col with a bias / residual / relu 2 -- the
    # plain query answers for the worst case, so a caller that sized scratch through it never sees PL_EWORKSPACE
    stem = (2, 256, 256, 3, 64, 7, 7, 2, 3)
    im2col = (2 * 128 * 128 + 64) * 160 * 4
    assert L.pl_conv2d_nhwc_scratch_bytes_ex(*stem, 0, 0, 1) == 0
    assert L.pl_conv2d_nhwc_scratch_bytes_ex(*stem, 1, 0, 1) == im2col == L.pl_conv2d_nhwc_scratch_bytes_ex(*stem, 0, 0, 2)
    assert L.pl_conv2d_nhwc_scratch_bytes(*stem) == im2col
    split = L.pl_conv2d_nhwc_scratch_bytes(4, 8, 8, 512, 512, 3, 3, 1, 1)              # layer4 conv2: 8 tiles
    assert split > 0 and split % (256 * 512 * 4) == 0
    assert L.pl_bn_train_scratch_bytes(0, 64) == 0 and L.pl_bn_train_scratch_bytes(4096, 64) > 0


def test_python_layout_matches_c_layout(pkg):
    for (i, h, o, s) in [(34, 1024, 51, 2), (34, 64, 51, 2), (51, 64, 34, 2), (34, 32, 51, 3), (10, 4, 3, 0)]:
        d = pkg._lib.PLDesc(in_dim=i, hidden=h, out_dim=o, num_stage=s, bn=1, dtype=0, p_dropout=0.5,
                            bn_eps=1e-5, bn_momentum=0.1)
        slots, total = pkg.layout.param_slots(i, h, o, s)
        L = pkg.lib()
        assert L.pl_param_tensors(ctypes.byref(d)) == len(slots) == 4 * (1 + 2 * s) + 2
        assert L.pl_param_arena_floats(ctypes.byref(d)) == total
        for k, sl in enumerate(slots):
            assert L.pl_param_offset(ctypes.byref(d), k) == sl.offset and sl.offset % 64 == 0
            assert L.pl_param_numel(ctypes.byref(d), k) == sl.numel
        assert L.pl_workspace_bytes(ctypes.byref(d), 64) > 0
    assert pkg.lib().pl_param_offset(ctypes.byref(d), 999) < 0
    assert b"out of range" in pkg.lib().pl_last_error()


def test_module_mirrors_reference_interface(pkg):
    g = load_golden("g2_train_nodrop_small.npz")          # state_dict recorded from the reference
    ref_keys = list(golden_state(g))
    m = pkg.LinearModel(34, 51, linear_size=64, num_stage=2, p_dropout=0.5, BN=True)
    sd = m.state_dict()
    assert list(sd) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == tuple(g["state:" + k].shape), k
    assert sd["batch_norm1.num_batches_tracked"].dtype == torch.int64
    full = pkg.LinearModel(34, 51)
    assert len(list(full.parameters())) == 22
    assert sum(p.numel() for p in full.parameters()) == 4_296_755
    assert [n for n, _ in full.named_parameters()] == [s.name for s in full._slots]
    # parameters are views into one arena; loading a checkpoint writes through to it
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in golden_state(g).items()})
    s = m._slots[4]
    assert torch.equal(m.flat_params[s.offset:s.offset + s.numel].view(s.shape), m.linear_stages[0].w1.weight)
    assert m._arenas_intact()
    np.testing.assert_array_equal(m.w2.bias.detach().numpy(), g["state:w2.bias"])
    m2 = m.double().float()                                 # _apply re-flattens
    assert m2._arenas_intact() and torch.equal(m2.w2.bias, torch.from_numpy(g["state:w2.bias"]))


def test_same_seed_gives_reference_initialisation(pkg):
    """The containers are stock nn.Linear objects created in the reference's order, so
    torch.manual_seed(s) reproduces the reference's initial weights (values recorded from the
    reference by tools/make_golden.py)."""
    g = load_golden("g7_init_seed0.npz")
    torch.manual_seed(0)
    m = pkg.LinearModel(34, 51, p_dropout=0.5, linear_size=1024, BN=True)
    sd = m.state_dict()
    for k in [k[5:] for k in g if k.startswith("head:")]:
        np.testing.assert_array_equal(sd[k].reshape(-1)[:16].numpy(), g["head:" + k])
        assert abs(float(sd[k].double().sum()) - float(g["sum:" + k])) < 1e-6 * max(1.0, abs(float(g["sum:" + k])))


def test_weight_init_vs_reference_golden(pkg):
    """weight_init (baselineModel.py:10-12, applied as main.py:396 `model.apply(weight_init)`): golden g10 = the
    reference's function run under a seed.  Linear weights are re-drawn Kaiming-normal (the same stream: bit-equal),
    biases and BatchNorm tensors stay untouched, and the parameters stay views into the flat arena."""
    g = load_golden("g10_weight_init.npz")
    torch.manual_seed(int(g["model_seed"]))
    m = pkg.LinearModel(34, 51, p_dropout=0.5, linear_size=int(g["hidden"]), BN=True)
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), g["before:" + k], err_msg=k)
    torch.manual_seed(int(g["init_seed"]))
    m.apply(pkg.weight_init)
    changed = 0
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), g["after:" + k], err_msg=k)
        changed += int(not np.array_equal(g["after:" + k], g["before:" + k]))
    assert changed == 6                       # the six nn.Linear weights, nothing else
    assert m._arenas_intact()
    s = m._slots[0]
    assert torch.equal(m.flat_params[s.offset:s.offset + s.numel].view(s.shape), m.w1.weight)
    w = torch.from_numpy(g["after:linear_stages.0.w1.weight"])
    assert abs(float(w.std()) - (2.0 / w.shape[1]) ** 0.5) < 0.05 * (2.0 / w.shape[1]) ** 0.5   # fan_in, gain sqrt(2)


def test_no_cpu_fallback(pkg):
    m = pkg.LinearModel(34, 51, linear_size=64)
    with pytest.raises(pkg.PoseliftError, match="no CPU path"):
        m(torch.zeros(4, 17, 2))
    with pytest.raises(pkg.PoseliftError):
        pkg.mse_loss(torch.zeros(4, 3), torch.zeros(4, 3))
    with pytest.raises(pkg.PoseliftError):
        pkg.loss_MPJPE(torch.zeros(4, 17, 3), torch.zeros(4, 17, 3))
    with pytest.raises(pkg.PoseliftError):
        pkg.FlatAdamW(m).step()


def test_bitmap_pack_roundtrip(pkg):
    rng = np.random.default_rng(0)
    for H in (32, 64, 256, 320, 1024):
        keep = rng.random((9, H)) < 0.5
        words = pkg.layout.pack_keep_bitmap(keep)
        assert words.shape == (9, pkg.layout.bitmap_words_per_row(H)) and words.dtype == np.uint64
        assert (pkg.layout.unpack_bitmap(words, H) == keep).all()
    # documented bit position: column 256*q + 4*l + j  ->  word 4*q + j, bit l
    keep = np.zeros((1, 512), bool)
    keep[0, 256 + 4 * 5 + 2] = True
    words = pkg.layout.pack_keep_bitmap(keep)
    assert words[0, 4 + 2] == np.uint64(1) << np.uint64(5) and words.sum() == words[0, 6]


def test_synthetic_batches_are_h36m_shaped(pkg, h36m_stats):
    x, y = pkg.synth.synthetic_batch(2048, 1234)
    assert x.shape == (2048, 17, 2) and y.shape == (2048, 17, 3) and x.dtype == torch.float32
    assert float(x.min()) >= 0 and float(x.max()) <= 1 and torch.all(y[:, 0] == 0)
    np.testing.assert_allclose(x.mean(0).numpy(), h36m_stats["mean_train_2d"], atol=0.02)
    np.testing.assert_allclose(y.std(0).numpy()[1:], h36m_stats["std_train_3d"][1:], rtol=0.1)
    x2, _ = pkg.synth.synthetic_batch(2048, 1234)
    assert torch.equal(x, x2)


def test_shard_rows_partitions_exactly(pkg):
    for n, w in [(32768, 8), (4096, 1), (10, 4), (7, 8)]:
        spans = [pkg.dp.shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_deepcopy_and_pickle_rebuild_the_arenas(pkg, tmp_path):
    import copy
    torch.manual_seed(3)
    m = pkg.LinearModel(34, 51, linear_size=64)
    for clone in (copy.deepcopy(m), torch.load(_save(m, tmp_path / "m.pt"), weights_only=False)):
        assert clone._arenas_intact() and clone.flat_params.data_ptr() != m.flat_params.data_ptr()
        assert all(torch.equal(a, b) for a, b in zip(clone.state_dict().values(), m.state_dict().values()))
        with torch.no_grad():
            clone.w2.bias.add_(1.0)                     # writes through to the clone's own arena only
        s = clone._slots[-1]
        assert torch.equal(clone.flat_params[s.offset:s.offset + s.numel], clone.w2.bias)
        assert not torch.equal(clone.w2.bias, m.w2.bias)


def _save(obj, path):
    torch.save(obj, path)        # a file this test wrote itself
    return path


def test_reduce_lr_on_plateau_drives_flat_adamw(pkg):
    """train_1.py:41,106: ReduceLROnPlateau(factor .7, patience 3, cooldown 2, min_lr 5e-6) stepped with
    the last batch's loss.  (verbose=True of the reference is a TypeError on torch 2.10: dropped.)"""
    m = pkg.LinearModel(34, 51, linear_size=64)
    opt = pkg.FlatAdamW(m, lr=1e-4)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.7, patience=3, cooldown=2, min_lr=5e-6)
    lrs = []
    for _ in range(12):
        sch.step(1.0)                                    # a loss that never improves
        lrs.append(opt.param_groups[0]["lr"])
    assert lrs[3] == pytest.approx(1e-4) and lrs[4] == pytest.approx(7e-5) and min(lrs) >= 5e-6
    assert lrs[-1] < lrs[4]
    sd = opt.state_dict()
    assert sd["param_groups"][0]["lr"] == lrs[-1] and len(sd["param_groups"][0]["params"]) == 22


def test_feeder_epoch_indices_partition_the_table():
    """data.epoch_indices: every pose exactly once per epoch over all ranks and batches, global
    batches cut like dp.shard_rows, fresh order per epoch, drop_last drops only the short batch."""
    import importlib
    import torch
    data = importlib.import_module("3d_poseestimation_amd.data")
    N, bs, world = 1003, 64, 3
    per_rank = [data.epoch_indices(N, bs, epoch=2, seed=5, rank=r, world=world) for r in range(world)]
    allidx = torch.cat([torch.cat(b) for b in per_rank])
    tail = N % (bs * world)
    assert len(set(allidx.tolist())) == allidx.numel() == N - tail % world      # the tail is trimmed to equal shards
    n_batches = len(per_rank[0])
    assert n_batches == (N + bs * world - 1) // (bs * world) == data.epoch_steps(N, bs, world=world)
    assert all(len(per_rank[r]) == n_batches for r in range(world))             # same number of steps on every rank
    for k in range(n_batches - 1):
        assert all(per_rank[r][k].numel() == bs for r in range(world))
    assert all(per_rank[r][-1].numel() == tail // world for r in range(world))
    # one process sees the same global batches as the ranks together (up to the trimmed rows of the tail)
    whole = data.epoch_indices(N, bs * world, epoch=2, seed=5)
    for k in range(n_batches):
        together = torch.cat([per_rank[r][k] for r in range(world)])
        assert torch.equal(whole[k][:together.numel()], together)
    # a tail that would leave a rank with fewer than two rows (BatchNorm raises on one row; an empty shard would skip
    # the collective the other ranks enter) is dropped on EVERY rank
    for n_small in (bs * world * 2 + 1, bs * world * 2 + world, bs * world * 2 + 2 * world - 1):
        counts = [len(data.epoch_indices(n_small, bs, epoch=0, seed=1, rank=r, world=world)) for r in range(world)]
        assert counts == [2] * world == [data.epoch_steps(n_small, bs, world=world)] * world
    assert data.epoch_steps(bs * world * 2 + 2 * world, bs, world=world) == 3
    other = data.epoch_indices(N, bs, epoch=3, seed=5, rank=0, world=world)
    assert not torch.equal(other[0], per_rank[0][0])
    dropped = data.epoch_indices(N, bs, epoch=2, seed=5, rank=0, world=world, drop_last=True)
    assert len(dropped) == N // (bs * world) and all(t.numel() == bs for t in dropped)
    plain = data.epoch_indices(10, 4, shuffle=False)
    assert [t.tolist() for t in plain] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_deconv_subkernels_match_their_definition():
    """conv.deconv_subkernels (one flip + one permuted copy, nothing on the host: capturable in a hipGraph) against the
    definition it replaces: output row 2a + ph of ConvTranspose2d(4, 2, 1) takes kh = [3, 1][a] (ph = 0) or [2, 0][a]."""
    import torch
    cv = importlib.import_module("3d_poseestimation_amd.conv")
    w = torch.randn(5, 7, 4, 4, generator=torch.Generator().manual_seed(4))
    subs = []
    for ph in (0, 1):
        for pw in (0, 1):
            kh = [2, 0] if ph else [3, 1]
            kw = [2, 0] if pw else [3, 1]
            subs.append(w[:, :, kh][:, :, :, kw].permute(1, 2, 3, 0))
    assert torch.equal(torch.stack(subs).contiguous(), cv.deconv_subkernels(w))
