"""numpy restatement of the soft-argmax heads.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: /root/reference/phase4_joined/Model.py and /root/reference/phase5_loop/Model_2d.py
cannot be imported here (they import matplotlib/cv2 helpers and fetch torchvision weights), and the
reference holds no fixture for these functions.  This file restates Model.py:72-81,94-133 and
Model_2d.py:96-134 from their text; tests additionally check it against a plain torch fp32
evaluation of the same formulas (autograd for the backward).
"""
import numpy as np


def soft_argmax(out, num_joints, depth_dim, centred, dtype=np.float64):
    """out (B, J*depth, H, W) -> (B, J*ncoord).  Literal order of operations of the reference:
    reshape (B,J,-1) -> softmax(dim 2) -> / sum -> reshape (B,J,D,H,W) -> marginal sums ->
    * arange -> sum -> scale."""
    out = np.asarray(out, dtype=dtype)
    B, C, H, W = out.shape
    D = depth_dim
    hm = out.reshape(B, num_joints, -1)
    hm = np.exp(hm - hm.max(axis=2, keepdims=True))
    hm = hm / hm.sum(axis=2, keepdims=True)
    hm = hm / hm.sum(axis=2, keepdims=True)
    hm = hm.reshape(B, num_joints, D, H, W)
    cx = (hm.sum(axis=(2, 3)) * np.arange(W, dtype=dtype)).sum(axis=2, keepdims=True)
    cy = (hm.sum(axis=(2, 4)) * np.arange(H, dtype=dtype)).sum(axis=2, keepdims=True)
    if centred:
        cz = (hm.sum(axis=(3, 4)) * np.arange(D, dtype=dtype)).sum(axis=2, keepdims=True)
        c = np.concatenate(((cx / W - 0.5) * 2, (cy / H - 0.5) * 2, (cz / D - 0.5) * 2), axis=2)
    else:
        c = np.concatenate((cx / W, cy / H), axis=2)
    return c.reshape(B, -1)
