"""Philox4x32-10 counter-based RNG (Salmon et al., SC'11) in numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference uses torch.nn.Dropout (baselineModel.py:20-21,37,43,81,94),
whose CPU Bernoulli stream cannot be reproduced on a GPU.  The HIP path
therefore defines its own dropout stream; this file is the bit-exact CPU
statement of that definition, so masks can be compared word for word.

Stream definition (shared with csrc/philox.h):
  element (row r, col c) of hidden layer `layer`, H columns:
      e = r*H + c;  g = e >> 2;  j = e & 3
      counter = (g & 0xffffffff, g >> 32, layer, step & 0xffffffff)
      key     = (seed & 0xffffffff, (seed >> 32) ^ (step >> 32))
      u       = philox4x32_10(counter, key)[j]
      keep    = u >= thr,  thr = min(2^32-1, floor(p * 2^32))   (p < 1)
  p >= 1 drops everything, p == 0 keeps everything.
"""
import math

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_SH = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays/ints.

    Returns four uint32 arrays.
    """
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _SH, p0 & _MASK
        hi1, lo1 = p1 >> _SH, p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def dropout_threshold(p):
    """uint32 threshold: keep iff u >= thr."""
    p = float(np.float32(p))
    return int(min(4294967295.0, math.floor(p * 4294967296.0)))


def dropout_keep_mask(seed, step, layer, rows, cols, p):
    """Boolean keep mask (rows, cols) of the stream defined in the module doc."""
    if p >= 1.0:
        return np.zeros((rows, cols), dtype=bool)
    if p <= 0.0:
        return np.ones((rows, cols), dtype=bool)
    n = rows * cols
    ngroups = (n + 3) // 4
    g = np.arange(ngroups, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    step = int(step) & 0xFFFFFFFFFFFFFFFF
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) ^ (step >> 32)) & 0xFFFFFFFF
    r = philox4x32_10(g & _MASK, g >> _SH, layer, step & 0xFFFFFFFF, k0, k1)
    u = np.stack(r, axis=1).reshape(-1)[:n]
    thr = np.uint32(dropout_threshold(p))
    return (u >= thr).reshape(rows, cols)
