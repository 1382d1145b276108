"""Flat-arena layout of the lifter's parameters and BatchNorm buffers (host logic only).

Mirrors param_layout() in csrc/api.hip; tests/test_host_logic.py checks the two agree.
Names and order are those of the reference module's state_dict()/parameters()
(/root/reference/phase1_lifting/baselineModel.py:14-30,50-85).
"""
from dataclasses import dataclass
from typing import List, Tuple

ALIGN = 64  # floats


def hidden_layer_prefixes(num_stage: int) -> List[Tuple[str, str]]:
    names = [("w1", "batch_norm1")]
    for s in range(num_stage):
        names.append((f"linear_stages.{s}.w1", f"linear_stages.{s}.batch_norm1"))
        names.append((f"linear_stages.{s}.w2", f"linear_stages.{s}.batch_norm2"))
    return names


@dataclass(frozen=True)
class TensorSlot:
    name: str
    shape: Tuple[int, ...]
    offset: int     # floats from the start of the arena

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


def _align(v):
    return (v + ALIGN - 1) // ALIGN * ALIGN


def param_slots(in_dim, hidden, out_dim, num_stage) -> Tuple[List[TensorSlot], int]:
    """Slots in parameters() order and the padded arena length in floats."""
    slots, off = [], 0

    def add(name, shape):
        nonlocal off
        s = TensorSlot(name, tuple(shape), off)
        slots.append(s)
        off = _align(off + s.numel)

    for i, (lin, bn) in enumerate(hidden_layer_prefixes(num_stage)):
        add(lin + ".weight", (hidden, in_dim if i == 0 else hidden))
        add(lin + ".bias", (hidden,))
        add(bn + ".weight", (hidden,))
        add(bn + ".bias", (hidden,))
    add("w2.weight", (out_dim, hidden))
    add("w2.bias", (out_dim,))
    return slots, off


def bitmap_words_per_row(hidden):
    return ((hidden + 255) // 256) * 4


def pack_keep_bitmap(keep):
    """(B, H) bool -> (B, words) uint64 in the library's bitmap layout: 256-column strip q,
    word j (0..3), bit l  <->  column 256*q + 4*l + j.  Host helper for parity tests that
    inject dropout masks (pl_lifter_fwd_train inject_keep)."""
    import numpy as np
    keep = np.asarray(keep, dtype=bool)
    B, H = keep.shape
    strips = (H + 255) // 256
    padded = np.zeros((B, strips * 256), dtype=bool)
    padded[:, :H] = keep
    v = padded.reshape(B, strips, 64, 4).astype(np.uint64)          # [b][q][l][j]
    weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))[None, None, :, None]
    words = (v * weights).sum(axis=2, dtype=np.uint64)              # [b][q][j]
    return words.reshape(B, strips * 4)


def unpack_bitmap(words, hidden):
    import numpy as np
    words = np.asarray(words, dtype=np.uint64)
    B = words.shape[0]
    strips = (hidden + 255) // 256
    w = words.reshape(B, strips, 1, 4)
    l = np.arange(64, dtype=np.uint64)[None, None, :, None]
    bits = ((w >> l) & np.uint64(1)).astype(bool)                   # [b][q][l][j]
    return bits.reshape(B, strips * 256)[:, :hidden]


def unpack_bitmap_tile(words, rows, hidden):
    """The tile-format bitmap of the small-batch layer kernels (include/poselift.h, pl_workspace_bitmap_format = 1):
    `hidden` words -> (rows, hidden) bool."""
    import numpy as np
    words = np.asarray(words, dtype=np.uint64).reshape(-1)
    r = np.arange(rows)[:, None]
    c = np.arange(hidden)[None, :]
    w = words[(c >> 4) * 16 + (r >> 4) * 4 + (c & 3)]
    bit = ((r & 15) * 4 + ((c >> 2) & 3)).astype(np.uint64)
    return ((w >> bit) & np.uint64(1)).astype(bool)
