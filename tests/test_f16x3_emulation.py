"""CPU emulation of the GEMM arithmetic modes on the golden eval forward (g1, the reference's fp64 output): what the
operand REPRESENTATION of each mode costs, with the products summed in fp64 (so accumulation order plays no part).
Pins the design decision of PL_F16X3: two fp16 planes with the low plane stored times 2^11 are as good
as exact fp32 products; without that scale the low plane falls into fp16's subnormals and the error is 5x larger; one
bf16 plane is 1000x outside the gate."""
import numpy as np
import torch

from conftest import load_golden
from oracle import lifter_oracle as orc


def _bf16(v):
    return torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _f16(v):
    return v.astype(np.float16).astype(np.float32)


def _mm_exact(a, w):
    return (a.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)


def _mm_bf16x6(a, w):
    def split(v):
        v = v.astype(np.float32)
        p0 = _bf16(v)
        r = v - p0
        p1 = _bf16(r)
        return [t.astype(np.float64) for t in (p0, p1, _bf16(r - p1))]
    a0, a1, a2 = split(a)
    w0, w1, w2 = split(w)
    return (a0 @ w0.T + a0 @ w1.T + a1 @ w0.T + a0 @ w2.T + a1 @ w1.T + a2 @ w0.T).astype(np.float32)


def _mm_f16x3(a, w, sa, sw, lo_scale):
    def split(v, s):
        v = v.astype(np.float32) * np.float32(s)
        h = _f16(v)
        return h.astype(np.float64), _f16((v - h) * np.float32(lo_scale)).astype(np.float64)
    ah, al = split(a, sa)
    wh, wl = split(w, sw)
    return ((ah @ wh.T + (ah @ wl.T + al @ wh.T) / lo_scale) / (sa * sw)).astype(np.float32)


def _mm_bf16(a, w):
    return (_bf16(a).astype(np.float64) @ _bf16(w).astype(np.float64).T).astype(np.float32)


def _eval_forward(st, x, mm):
    """LinearModel.forward in eval mode (baselineModel.py:87-102) with the 1024-wide GEMMs through `mm`."""
    names = orc.hidden_layer_names(2)
    a = x.reshape(x.shape[0], -1).astype(np.float32)

    def hidden(lin, bnp, t, first=False):
        z = (_mm_exact if first else mm)(t, st[lin + ".weight"]) + st[lin + ".bias"]
        rstd = np.float32(1) / np.sqrt(st[bnp + ".running_var"] + np.float32(1e-5))
        y = (z - st[bnp + ".running_mean"]) * rstd * st[bnp + ".weight"] + st[bnp + ".bias"]
        return np.maximum(y, 0).astype(np.float32)
    h = hidden(*names[0], a, True)
    for s in range(2):
        h = h + hidden(*names[2 + 2 * s], hidden(*names[1 + 2 * s], h))
    return _mm_exact(h, st["w2.weight"]) + st["w2.bias"]


def test_f16x3_planes_represent_fp32_operands_as_well_as_exact_products():
    g = load_golden("g1_eval_full.npz")
    st = orc.init_state(34, 51, 1024, 2, rng=np.random.default_rng(int(g["weight_seed"])), nontrivial_bn=True)
    err = {}
    for name, mm in (("exact", _mm_exact), ("bf16x6", _mm_bf16x6),
                     ("f16x3", lambda a, w: _mm_f16x3(a, w, 1.0, 16.0, 2048.0)),       # the library's scales
                     ("f16x3 unscaled low plane", lambda a, w: _mm_f16x3(a, w, 1.0, 1.0, 1.0)),
                     ("bf16", _mm_bf16)):
        err[name] = orc.mpjpe_mm(_eval_forward(st, g["x"], mm), g["y_fp64"])
    assert err["exact"] < 1e-4 and err["bf16x6"] < 1.1 * err["exact"] + 1e-6
    assert err["f16x3"] < 1.15 * err["exact"] + 1e-6, err          # measured 6.4e-5 vs 6.0e-5 mm
    assert err["f16x3 unscaled low plane"] > 3 * err["f16x3"], err  # 3.3e-4 mm: fp16 subnormals
    assert 0.1 < err["bf16"] < 5.0, err                            # ~0.8 mm: a thousand times the gate
    assert orc.mpjpe_mm(g["y"], g["y_fp64"]) < 1e-3                 # the reference's own fp32 forward, for scale
