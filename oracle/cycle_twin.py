"""Stock-PyTorch restatement of the phase5 cycle step.  TEST INFRASTRUCTURE ONLY.

  /root/reference/phase5_loop/train_5 copy.py:147-236   the step (Triangle on; Flip: :174-199; Project)
  /root/reference/phase5_loop/train_5 copy.py:34-86     TriangleLoss (LinearModel era)
  /root/reference/phase5_loop/Model_2d.py:87-136        Model_2D forward (2-D soft-argmax, coordinates in (0, 1))
  /root/reference/phase4_joined/Model.py:83-137         Model_3D forward (3-D soft-argmax, coordinates in (-1, 1))
  /root/reference/phase3_direct/my_HybrIK/utils.py:372-396   flip_pose

PARITY UNPINNED for the two heat-map networks: Model.py / Model_2d.py / train_5 copy.py cannot be imported here (cv2,
torchvision, wandb absent; the constructors fetch pretrained weights) and the reference holds no fixture for them, so
this file restates them from their text on stock torch.nn modules.  The backbone (Resnet.py) and TriangleLoss of
losses.py ARE pinned elsewhere (goldens g9 / g11 / g8).  Everything here runs under torch autograd in whatever dtype
the modules are in: the GPU tests compare the HIP path with it in float64 and take float32-vs-float64 of this same
code as the noise floor; bench.py's `--workload cycle` times it on the host cores as the CPU baseline.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import torch
import torch.nn.functional as F

LEFT, RIGHT = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]


def flip_pose(t):
    """utils.py:372-396 on an (N, 17, D) tensor, out of place and differentiable."""
    x0 = (1 - t[..., :1]) if t.shape[-1] == 2 else -t[..., :1]
    f = torch.cat((x0, t[..., 1:]), dim=-1)
    idx = list(range(17))
    for a, b in zip(LEFT, RIGHT):
        idx[a], idx[b] = b, a
    return f[..., idx, :]


def heatmap_logits(m, x_nchw):
    """preact -> deconv_layers -> final_layer of a Model_2D / Model_3D container (stock nn modules inside)."""
    r = m.preact
    x = F.max_pool2d(F.relu(r.bn1(r.conv1(x_nchw))), 3, 2, 1)
    for li in (1, 2, 3, 4):
        for blk in getattr(r, f"layer{li}"):
            idn = x if blk.downsample is None else blk.downsample(x)
            o = F.relu(blk.bn1(blk.conv1(x)))
            o = F.relu(blk.bn2(blk.conv2(o)))
            x = F.relu(blk.bn3(blk.conv3(o)) + idn)
    return m.final_layer(m.deconv_layers(x))


def model_3d(m, x_nchw):
    out = heatmap_logits(m, x_nchw)
    B, _, H, W = out.shape
    hm = torch.softmax(out.reshape(B, 17, -1), 2)
    hm = (hm / hm.sum(2, keepdim=True)).reshape(B, 17, 64, H, W)
    dt = out.dtype
    cx = (hm.sum((2, 3)) * torch.arange(W, dtype=dt, device=out.device)).sum(2, keepdim=True)
    cy = (hm.sum((2, 4)) * torch.arange(H, dtype=dt, device=out.device)).sum(2, keepdim=True)
    cz = (hm.sum((3, 4)) * torch.arange(64, dtype=dt, device=out.device)).sum(2, keepdim=True)
    return torch.cat(((cx / W - .5) * 2, (cy / H - .5) * 2, (cz / 64 - .5) * 2), 2)          # (B, 17, 3)


def model_2d(m, x_nchw):
    out = heatmap_logits(m, x_nchw)
    B, _, H, W = out.shape
    hm = torch.softmax(out.reshape(B, 17, -1), 2)
    hm = (hm / hm.sum(2, keepdim=True)).reshape(B, 17, H, W)
    dt = out.dtype
    cx = (hm.sum(2) * torch.arange(W, dtype=dt, device=out.device)).sum(2, keepdim=True)
    cy = (hm.sum(3) * torch.arange(H, dtype=dt, device=out.device)).sum(2, keepdim=True)
    return torch.cat((cx / W, cy / H), 2)                                                      # (B, 17, 2)


def _centre_on_first(t):
    """`t[1:] -= t[0]` (train_5 copy.py:58-61): batch entries 1.. relative to batch entry 0."""
    return torch.cat([t[:1], t[1:] - t[0]], dim=0)


def triangle_loss(predicted_2d, predicted_3d, lift_2d_gt, lift_2d_pred, gt_2d, gt_3d, proj_3d_pred=None, proj_3d_gt=None):
    """train_5 copy.py:47-70 (L1Loss terms; Project when the projector outputs are given)."""
    loss = (F.l1_loss(predicted_2d, gt_2d) + F.l1_loss(predicted_3d, gt_3d) + F.l1_loss(lift_2d_gt, gt_3d)
            + F.l1_loss(lift_2d_pred, lift_2d_gt))
    if proj_3d_pred is not None:
        pp, pg = _centre_on_first(proj_3d_pred), _centre_on_first(proj_3d_gt)
        loss = loss + F.l1_loss(pg, _centre_on_first(gt_2d)) + F.l1_loss(pp, pg)
    return loss


def cycle_loss(m2, m3, lift, proj, frame_nchw, y1, y2, Flip=False):
    """Forward of train_5 copy.py:159-216 (Triangle on).  Returns (loss, y1_hat, y2_hat)."""
    B = y1.shape[0]
    y1_hat = model_2d(m2, frame_nchw)
    y2_hat = model_3d(m3, frame_nchw)
    lift_pred = lift(y1_hat).reshape(B, 17, 3)
    lift_gt = lift(y1).reshape(B, 17, 3)
    pp = pg = None
    if proj is not None:
        pp = proj(y2_hat).reshape(B, 17, 2)
        pg = proj(y2).reshape(B, 17, 2)
    if Flip:
        ff = torch.flip(frame_nchw, (3,))
        y1f = flip_pose(y1)
        y1_hat = (flip_pose(model_2d(m2, ff)) + y1_hat) / 2
        y2_hat = (flip_pose(model_3d(m3, ff)) + y2_hat) / 2
        lift_pred = (flip_pose(lift(y1_hat).reshape(B, 17, 3)) + lift_pred) / 2
        lift_gt = (flip_pose(lift(y1f).reshape(B, 17, 3)) + lift_gt) / 2
        if proj is not None:
            y2f = flip_pose(y2)
            pp = (flip_pose(proj(y2_hat).reshape(B, 17, 2)) + pp) / 2
            pg = (flip_pose(proj(y2f).reshape(B, 17, 2)) + pg) / 2
    return triangle_loss(y1_hat, y2_hat, lift_gt, lift_pred, y1, y2, pp, pg), y1_hat, y2_hat


def cycle_step(m2, m3, lift, proj, optimizers, frame_nchw, y1, y2, Flip=False):
    """train_5 copy.py:147-236: zero_grad, forward, TriangleLoss, one backward, every optimizer steps."""
    for o in optimizers:
        o.zero_grad()
    loss, y1_hat, y2_hat = cycle_loss(m2, m3, lift, proj, frame_nchw, y1, y2, Flip)
    loss.backward()
    for o in optimizers:
        o.step()
    return loss.detach(), y1_hat.detach(), y2_hat.detach()
