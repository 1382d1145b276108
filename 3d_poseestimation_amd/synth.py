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
