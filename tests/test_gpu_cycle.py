"""GPU tests of the phase5 cycle step at BASELINE configs[4]'s per-GPU shape:
/root/reference/phase5_loop/train_5 copy.py:147-236 -- Model_2D + Model_3D on 256 x 256 frames, the lifter
LinearModel(34, 51, 1024) called on the predicted and on the true 2-D pose, the projector LinearModel(51, 34, 64),
TriangleLoss, ONE backward -- with and without the training-mode Flip branch (:174-199).

Oracle: oracle/cycle_twin.py, the same step on stock torch.nn modules under torch autograd, float64 on the CPU.  The
heat-map networks' reference files are not importable and hold no fixtures: beyond the ResNet backbone and
TriangleLoss (goldens g9 / g11 / g8) this is PARITY UNPINNED against the reference itself -- the twin restates
Model.py / Model_2d.py / train_5 copy.py from their text.
"""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import cycle_twin as twin
from oracle.torch_twin import TwinLifter

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    p = ge.build()
    assert torch.cuda.is_available()
    return p


def _models(pkg, lift_width=1024, p_dropout=0.0):
    torch.manual_seed(0)
    m2, m3 = pkg.Model_2D().train(), pkg.Model_3D().train()          # default arithmetic: f16x3 operand planes
    for m, seed in ((m2, 61), (m3, 62)):
        m.load_state_dict(pkg.synth.seeded_state(m.state_dict(), seed))
        with torch.no_grad():
            # (logit scales that keep the step well-conditioned: the 3-D soft-argmax spans 262,144 voxels)
            m.final_layer.weight.mul_(1.0 if m is m2 else 0.1)
    lift = pkg.LinearModel(34, 51, linear_size=lift_width, p_dropout=p_dropout, compute_dtype="f16x3").train()
    proj = pkg.LinearModel(51, 34, linear_size=64, p_dropout=p_dropout, compute_dtype="f16x3").train()
    return m2, m3, lift, proj


def _twin_of_lifter(m, dtype):
    t = TwinLifter(m.input_size, m.output_size, linear_size=m.linear_size, p_dropout=m.p_dropout).to(dtype)
    t.load_state_dict({k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu())
                       for k, v in m.state_dict().items()})
    return t.train()


def test_flip_pieces_vs_the_torch_composition(pkg):
    """flip_pose / flip_average / flip_frames_nhwc, values and gradients, against the restated utils.py:372-396 and
    torch.flip -- bit-exact (a permutation, one subtraction, one halving)."""
    g = torch.Generator().manual_seed(3)
    for D in (2, 3):
        a = torch.rand(9, 17, D, generator=g)
        b = torch.rand(9, 17, D, generator=g)
        w = torch.rand(9, 17, D, generator=g)
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        got = pkg.flip_average(ad, bd)
        want = (twin.flip_pose(ac) + bc) / 2
        assert torch.equal(got.detach().cpu(), want.detach())
        (got * w.to(DEV)).sum().backward()
        (want * w).sum().backward()
        assert torch.equal(ad.grad.cpu(), ac.grad) and torch.equal(bd.grad.cpu(), bc.grad)
        f = pkg.flip_pose(a.to(DEV))
        assert torch.equal(f.cpu(), twin.flip_pose(a))
        assert torch.allclose(pkg.flip_pose(f).cpu(), a, atol=1e-7)                    # an involution
    fr = torch.rand(3, 8, 12, 3, generator=g)
    assert torch.equal(pkg.flip_frames_nhwc(fr.to(DEV)).cpu(), torch.flip(fr, (2,)))
    fr = torch.rand(2, 4, 6, 8, generator=g)
    assert torch.equal(pkg.flip_frames_nhwc(fr.to(DEV)).cpu(), torch.flip(fr, (2,)))


@pytest.mark.parametrize("Flip", [False, True])
def test_cycle_step_at_config4_shape_vs_torch_autograd_fp64(pkg, Flip):
    """One cycle step on 256 x 256 frames with the config's own networks (1024-wide lifter, 64-wide projector, f16x3)
    against oracle/cycle_twin.py in float64: the loss, the lifter's and the projector's 22 gradients each, every
    gradient of both heat-map networks, and the BatchNorm statistics (which the Flip branch moves twice).
    Batch 4, not 2: BatchNorm1d over two rows is the degenerate +-1 case (every gradient through it is round-off).
    Dropout off on both sides (torch's CPU Bernoulli stream cannot be reproduced; masks are covered by g3)."""
    B = 4
    m2, m3, lift, proj = _models(pkg)
    # frames that differ from each other: untrained networks map pure-noise frames to ONE pose, and BatchNorm1d over
    # near-identical rows turns every gradient behind the lifter into amplified round-off (synth.structured_frames)
    frames = pkg.synth.structured_frames(B, 63, size=256)
    y1, y2 = pkg.synth.synthetic_batch(B, 64)
    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def torch_run(dtype):
        t2, t3 = copy.deepcopy(m2).to(dtype), copy.deepcopy(m3).to(dtype)
        tl, tp = _twin_of_lifter(lift, dtype), _twin_of_lifter(proj, dtype)
        loss, p2, p3 = twin.cycle_loss(t2, t3, tl, tp, frames.permute(0, 3, 1, 2).to(dtype), y1.to(dtype), y2.to(dtype), Flip)
        loss.backward()
        return (t2, t3, tl, tp), loss.detach(), p2.detach(), p3.detach()

    ref, loss_ref, p2_ref, p3_ref = torch_run(torch.float64)
    ref32, loss32, _, _ = torch_run(torch.float32)
    d2, d3, dl, dp = m2.to(DEV), m3.to(DEV), lift.to(DEV), proj.to(DEV)
    crit = pkg.TriangleLoss(Project=True, era="lifter")
    loss, p2, p3 = pkg.cycle_step(d2, d3, dl, [], frames.to(DEV), y1.to(DEV), y2.to(DEV), crit, model_proj=dp, Flip=Flip)
    floor = abs(float(loss32) - float(loss_ref))
    assert abs(float(loss) - float(loss_ref)) < max(10 * floor, 1e-5 * abs(float(loss_ref))), (float(loss), float(loss_ref), floor)
    assert float((p2.cpu().double() - p2_ref).abs().max()) < 1e-4 and float((p3.cpu().double() - p3_ref).abs().max()) < 1e-4
    # per tensor, relative L2, against what stock fp32 does against its own fp64 (test_model3d_train_step_vs_torch_autograd)
    worst = {}
    for name, ours, r64, r32 in (("model_2d", d2, ref[0], ref32[0]), ("model_3d", d3, ref[1], ref32[1]),
                                 ("lifter", dl, ref[2], ref32[2]), ("projector", dp, ref[3], ref32[3])):
        checked, e2, f2 = 0, 0.0, 0.0
        for (k, p), (_, q), (_, q32) in zip(ours.named_parameters(), r64.named_parameters(), r32.named_parameters()):
            assert p.grad is not None, (name, k)
            gr = q.grad
            if float(gr.abs().max()) < 1e-12:
                continue
            norm = float(gr.norm())
            rel = float((p.grad.cpu().double() - gr).norm()) / norm
            floor_rel = float((q32.grad.double() - gr).norm()) / norm
            assert rel < min(max(25 * floor_rel, 5e-3), 0.5), (name, k, rel, floor_rel)
            worst[name] = max(worst.get(name, (0.0, "")), (rel, k))
            e2 += rel * rel
            f2 += floor_rel * floor_rel
            checked += 1
        # over the model's tensors together: no further from fp64 than 4 x what stock fp32 is
        assert (e2 / checked) ** 0.5 < 4 * (f2 / checked) ** 0.5 + 1e-4, (name, (e2 / checked) ** 0.5, (f2 / checked) ** 0.5)
        assert checked >= (150 if name.startswith("model") else 17), (name, checked)      # (the 5 pre-BatchNorm biases: zero gradient)
    # BatchNorm statistics: moved once without Flip, twice with it -- as the reference's second training-mode forward does
    for ours, r64 in ((d2, ref[0]), (d3, ref[1])):
        assert float((ours.preact.bn1.running_mean.cpu().double() - r64.preact.bn1.running_mean).abs().max()) < 1e-5
        assert int(ours.preact.bn1.num_batches_tracked) == int(r64.preact.bn1.num_batches_tracked)
    sd, rd = dl.state_dict(), ref[2].state_dict()
    for k in ("batch_norm1.running_mean", "linear_stages.1.batch_norm2.running_var"):
        assert float((sd[k].cpu().double() - rd[k]).abs().max()) < 1e-4 * max(1.0, float(rd[k].abs().max())), k
    assert int(sd["batch_norm1.num_batches_tracked"]) == int(rd["batch_norm1.num_batches_tracked"]) == (4 if Flip else 2)


@pytest.mark.parametrize("Flip", [False, True])
def test_cycle_step_at_config4_per_gpu_batch_properties(pkg, Flip):
    """BASELINE configs[4] on one GPU: batch 128 (1024 over 8 GPUs), 256 x 256 frames, lifter 1024 + projector 64,
    dropout 0.5 as the reference builds them (train_5 copy.py:94-96), Adam on all four models -- through
    size-independent properties: the step is bitwise repeatable from equal state, every gradient is finite, every
    model's parameters move, the loss falls on a repeated batch."""
    B = 128
    m2, m3, lift, proj = _models(pkg, p_dropout=0.5)
    m2, m3, lift, proj = m2.to(DEV), m3.to(DEV), lift.to(DEV), proj.to(DEV)
    frames = pkg.synth.structured_frames(B, 65, size=256).to(DEV)
    y1, y2 = pkg.synth.synthetic_batch(B, 66, DEV)
    models = (m2, m3, lift, proj)
    sd0 = [{k: v.clone() for k, v in m.state_dict().items()} for m in models]

    def fresh_opts():
        # (arena.FlatAdam: the conv models' gradients go straight into flat arenas, one Adam launch each; under Flip the
        #  second backward of a step accumulates into them)
        return [pkg.FlatAdam(m2, lr=1e-4), pkg.FlatAdam(m3, lr=1e-4),
                pkg.FlatAdamW(lift, lr=1e-3, weight_decay=0.0), pkg.FlatAdamW(proj, lr=1e-3, weight_decay=0.0)]

    runs = []
    for _ in range(2):
        for m, sd in zip(models, sd0):
            m.load_state_dict(sd)
        lift.manual_seed(11, step=0)
        proj.manual_seed(12, step=0)
        crit = pkg.TriangleLoss(Project=True, era="lifter")
        loss, y1_hat, y2_hat = pkg.cycle_step(m2, m3, lift, fresh_opts(), frames, y1, y2, crit, model_proj=proj, Flip=Flip)
        grads = [p.grad.clone() for m in models for p in m.parameters()]
        runs.append((loss.clone(), y1_hat.clone(), y2_hat.clone(), grads, [m.state_dict()[k].clone() for m in models
                                                                           for k in list(m.state_dict())[:3]]))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    nz = 0
    for a, b in zip(runs[0][3], runs[1][3]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
        nz += int(float(a.abs().max()) > 0)
    assert nz >= len(runs[0][3]) - 4
    for a, b in zip(runs[0][4], runs[1][4]):
        assert torch.equal(a, b)
    for m, sd in zip(models, sd0):                                  # every model moved
        k = [k for k in sd if k.endswith("weight")][-1]
        assert not torch.equal(m.state_dict()[k], sd[k]), k
    assert y1_hat.shape == (B, 17, 2) and y2_hat.shape == (B, 17, 3)
    # a few more steps on the same batch: the loss comes down
    opts = fresh_opts()
    crit = pkg.TriangleLoss(Project=True, era="lifter")
    losses = [float(pkg.cycle_step(m2, m3, lift, opts, frames, y1, y2, crit, model_proj=proj, Flip=Flip)[0]) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert len(crit.term_means()) == 6
