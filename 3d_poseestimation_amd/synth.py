"""H3.6M-shaped synthetic batches: the dataset itself
(/root/reference/phase3_direct/my_HybrIK/H36_dataset.py) is not redistributable, so the
bench and tests draw inputs with the per-joint statistics the reference ships
(phase1_lifting/{mean,std}_train_{2d,3d}.npy, stored as data in data/h36m_stats.npz):
  x2d (N,17,2) ~ N(mean_2d, std_2d) clipped to [0,1]   image-normalised keypoints
  y3d (N,17,3) ~ N(0, std_3d), root joint row = 0       root-relative metres
"""
import os

import numpy as np
import torch

_STATS = None


def h36m_stats():
    global _STATS
    if _STATS is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "h36m_stats.npz")
        with np.load(path, allow_pickle=False) as z:
            _STATS = {k: z[k].astype(np.float32) for k in z.files}
    return _STATS


def synthetic_batch(n, seed, device="cpu"):
    st = h36m_stats()
    g = torch.Generator().manual_seed(int(seed))
    x = torch.randn(n, 17, 2, generator=g) * torch.from_numpy(st["std_train_2d"]) + torch.from_numpy(st["mean_train_2d"])
    x = x.clamp_(0.0, 1.0)
    y = torch.randn(n, 17, 3, generator=g) * torch.from_numpy(st["std_train_3d"])
    y[:, 0, :] = 0
    return x.to(device), y.to(device)


def seeded_state(template, seed):
    """Deterministic stand-in for a checkpoint: a state_dict with the template's keys and shapes, every tensor
    drawn from its own torch CPU generator (seed, position).  Used where a fixture would otherwise have to carry
    the weights themselves (ResNet-50 = 94 MB): tools/make_golden.py fills the REFERENCE model with it, the tests
    fill ours, and only the reference's outputs are committed.  Conv / linear weights N(0, 2/fan_in), BatchNorm
    gamma and running_var U(0.5, 1.5), beta / running_mean / biases N(0, 0.1^2)."""
    out = {}
    for idx, (k, v) in enumerate(template.items()):
        g = torch.Generator().manual_seed(int(seed) * 100003 + idx)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(7, dtype=v.dtype)
        elif k.endswith("running_var") or (v.dim() == 1 and k.endswith(".weight")):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        elif v.dim() == 1:
            out[k] = torch.randn(v.shape, generator=g) * 0.1
        else:
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
    return out


def seeded_frames(batch, seed, size=256):
    """[batch, size, size, 3] NHWC frames in [0, 1) (the phase4 loader divides by 256)."""
    g = torch.Generator().manual_seed(int(seed))
    return torch.rand(batch, size, size, 3, generator=g)


def structured_frames(batch, seed, size=256):
    """[batch, size, size, 3] NHWC frames in [0, 1) that DIFFER from each other at low frequency: 30 % noise plus a
    Gaussian blob whose position, width and colour depend on the frame.  Untrained heat-map networks predict the same
    pose for every frame of pure noise; the lifter's BatchNorm1d then normalises a batch of near-identical rows and
    every gradient behind it is amplified round-off (fp32 vs fp64 of stock torch differ by 80 %).  With these frames the
    predictions spread (2-D: 0.05) and the cycle step is a well-conditioned test problem."""
    g = torch.Generator().manual_seed(int(seed))
    frames = torch.rand(batch, size, size, 3, generator=g)
    lin = torch.linspace(0, 1, size)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    for b in range(batch):
        k = b % 4
        cx, cy = 0.2 + 0.2 * k + 0.013 * (b // 4), 0.8 - 0.15 * k - 0.011 * (b // 4)
        blob = torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (0.02 + 0.01 * k))
        frames[b] = 0.3 * frames[b] + 0.7 * blob[..., None] * torch.tensor([1.0, 0.5 + 0.1 * k, 0.2 * k])
    return frames
