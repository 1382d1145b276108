"""numpy restatement of the lifter hot path.  TEST INFRASTRUCTURE ONLY.

Follows, formula by formula:
  * LinearModel.forward           /root/reference/phase1_lifting/baselineModel.py:87-102
  * Linear.forward (resid. block) /root/reference/phase1_lifting/baselineModel.py:32-47
  * train step (MSE mean, AdamW)  /root/reference/phase1_lifting/train_1.py:37,39,75-100
  * loss_MPJPE + epoch reduction  /root/reference/phase1_lifting/train_1.py:19-23,100-104
torch semantics restated (BatchNorm1d training/eval, Dropout scaling, AdamW
single-tensor update order) are the documented ATen ones for torch 2.10.

Everything is written with explicit loops over layers and plain matmuls so
that it can run in float32 (parity reference) or float64 (noise floor).
State is a dict keyed by the reference's state_dict() names.
"""
import numpy as np

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# naming: hidden layer index -> (linear prefix, batch-norm prefix)
# --------------------------------------------------------------------------
def hidden_layer_names(num_stage):
    """Order of the 1+2*num_stage hidden Linear/BN pairs (baselineModel.py:67-74)."""
    names = [("w1", "batch_norm1")]
    for s in range(num_stage):
        names.append((f"linear_stages.{s}.w1", f"linear_stages.{s}.batch_norm1"))
        names.append((f"linear_stages.{s}.w2", f"linear_stages.{s}.batch_norm2"))
    return names


def param_names(num_stage, bn_params=True):
    """parameters() order of the reference module (registration order)."""
    out = []
    for lin, bn in hidden_layer_names(num_stage):
        out += [lin + ".weight", lin + ".bias"]
        if bn_params:
            out += [bn + ".weight", bn + ".bias"]
    out += ["w2.weight", "w2.bias"]
    return out


def init_state(in_dim, out_dim, hidden, num_stage, rng, dtype=np.float32,
               nontrivial_bn=False):
    """Deterministic state with torch-default-like scales: U(+-1/sqrt(fan_in))
    for Linear weight and bias (baselineModel.py:67,72,77 -> nn.Linear default),
    gamma=1, beta=0, running_mean=0, running_var=1.

    nontrivial_bn=True perturbs gamma/beta/running stats so eval-mode parity is
    exercised on a non-identity BatchNorm.
    """
    st = {}

    def lin(prefix, fan_out, fan_in):
        bound = 1.0 / np.sqrt(fan_in)
        st[prefix + ".weight"] = rng.uniform(-bound, bound, (fan_out, fan_in)).astype(dtype)
        st[prefix + ".bias"] = rng.uniform(-bound, bound, (fan_out,)).astype(dtype)

    def bn(prefix):
        if nontrivial_bn:
            st[prefix + ".weight"] = rng.uniform(0.5, 1.5, (hidden,)).astype(dtype)
            st[prefix + ".bias"] = rng.uniform(-0.3, 0.3, (hidden,)).astype(dtype)
            st[prefix + ".running_mean"] = rng.uniform(-0.5, 0.5, (hidden,)).astype(dtype)
            st[prefix + ".running_var"] = rng.uniform(0.2, 1.5, (hidden,)).astype(dtype)
            st[prefix + ".num_batches_tracked"] = np.array(7, dtype=np.int64)
        else:
            st[prefix + ".weight"] = np.ones((hidden,), dtype)
            st[prefix + ".bias"] = np.zeros((hidden,), dtype)
            st[prefix + ".running_mean"] = np.zeros((hidden,), dtype)
            st[prefix + ".running_var"] = np.ones((hidden,), dtype)
            st[prefix + ".num_batches_tracked"] = np.array(0, dtype=np.int64)

    names = hidden_layer_names(num_stage)
    for i, (l, b) in enumerate(names):
        lin(l, hidden, in_dim if i == 0 else hidden)
        bn(b)
    lin("w2", out_dim, hidden)
    return st


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
def _hidden_fwd(st, lin, bnp, a_in, train, use_bn, keep, p, dtype, update_running, on=None):
    """Linear -> [BN] -> ReLU -> Dropout   (baselineModel.py:33-37 / 90-94).

    on: optional boolean (B,H) "positive and kept" decisions taken from the implementation
    under test.  ReLU is discontinuous in its derivative: a pre-activation within round-off
    of zero may land on either side in two correct fp32 evaluations, and the sample then
    enters or leaves whole rows of dW.  Forcing the decisions removes that from a comparison;
    the caller checks separately that every forced decision that differs from the oracle's
    own sits on such a round-off-sized pre-activation (cache['on_disagree'])."""
    W = st[lin + ".weight"].astype(dtype)
    b = st[lin + ".bias"].astype(dtype)
    z = a_in @ W.T + b
    c = {"a_in": a_in, "z": z}
    B = z.shape[0]
    if use_bn:
        gamma = st[bnp + ".weight"].astype(dtype)
        beta = st[bnp + ".bias"].astype(dtype)
        if train:
            if B < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            mean = z.mean(axis=0, dtype=dtype)
            var = ((z - mean) ** 2).mean(axis=0, dtype=dtype)          # biased
            if update_running:
                m = dtype(BN_MOMENTUM)
                rm = st[bnp + ".running_mean"].astype(dtype)
                rv = st[bnp + ".running_var"].astype(dtype)
                unbiased = var * dtype(B) / dtype(B - 1)
                st[bnp + ".running_mean"] = ((dtype(1) - m) * rm + m * mean).astype(
                    st[bnp + ".running_mean"].dtype)
                st[bnp + ".running_var"] = ((dtype(1) - m) * rv + m * unbiased).astype(
                    st[bnp + ".running_var"].dtype)
                st[bnp + ".num_batches_tracked"] = np.array(
                    int(st[bnp + ".num_batches_tracked"]) + 1, dtype=np.int64)
        else:
            mean = st[bnp + ".running_mean"].astype(dtype)
            var = st[bnp + ".running_var"].astype(dtype)
        rstd = dtype(1) / np.sqrt(var + dtype(BN_EPS))
        zhat = (z - mean) * rstd
        y = zhat * gamma + beta
        c.update(mean=mean, var=var, rstd=rstd, zhat=zhat, bn_train=train)
    else:
        y = z
    rmask = y > 0
    if on is not None:
        scale = dtype(1)
        if train and 0 < p < 1:
            scale = dtype(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
        own = rmask if keep is None else (rmask & keep)
        c["on_disagree"] = np.abs(y)[own != on]
        c.update(keep=None, scale=scale, rmask=on, a_out=np.where(on, y * scale, dtype(0)))
        # backward reads rmask (decision) and scale through the keep=None branch below
        c["forced_scale"] = scale
        return c["a_out"], c
    y = np.where(rmask, y, dtype(0))
    if train and p > 0:
        if p >= 1:
            scale = dtype(0)
            keep = np.zeros_like(rmask)
        else:
            scale = dtype(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
        y = np.where(keep, y * scale, dtype(0))
        c.update(keep=keep, scale=scale)
    else:
        c.update(keep=None, scale=dtype(1))
    c["rmask"] = rmask
    c["a_out"] = y
    return y, c


def forward(st, x, *, num_stage=2, train=False, use_bn=True, p_dropout=0.5,
            keep_masks=None, dtype=np.float32, update_running=True, on_masks=None):
    """LinearModel.forward (baselineModel.py:87-102).

    keep_masks: list of 1+2*num_stage boolean (B,H) keep masks, required when
    train and 0 < p_dropout < 1.  Returns (y (B,out_dim), cache).
    """
    x = np.asarray(x)
    B = x.shape[0]
    a = x.reshape(B, -1).astype(dtype)                       # nn.Flatten, :89
    names = hidden_layer_names(num_stage)
    need_masks = train and 0 < p_dropout < 1
    if need_masks and keep_masks is None and on_masks is None:
        raise ValueError("train-mode dropout needs explicit keep masks")
    caches = []

    def km(i):
        return keep_masks[i] if (need_masks and keep_masks is not None) else None

    def om(i):
        return on_masks[i] if on_masks is not None else None

    h, c = _hidden_fwd(st, *names[0], a, train, use_bn, km(0), p_dropout, dtype, update_running, om(0))
    caches.append(c)
    for s in range(num_stage):                                # :97-98
        i1, i2 = 1 + 2 * s, 2 + 2 * s
        y, c1 = _hidden_fwd(st, *names[i1], h, train, use_bn, km(i1), p_dropout, dtype, update_running, om(i1))
        y, c2 = _hidden_fwd(st, *names[i2], y, train, use_bn, km(i2), p_dropout, dtype, update_running, om(i2))
        caches += [c1, c2]
        h = h + y                                             # :45
    W = st["w2.weight"].astype(dtype)
    b = st["w2.bias"].astype(dtype)
    out = h @ W.T + b                                         # :100
    return out, {"layers": caches, "h_last": h, "num_stage": num_stage,
                 "use_bn": use_bn, "dtype": dtype}


# --------------------------------------------------------------------------
# backward
# --------------------------------------------------------------------------
def _hidden_bwd(st, lin, bnp, c, g, use_bn, dtype):
    dy = np.where(c["rmask"], g, dtype(0))
    if c["keep"] is not None:
        dy = np.where(c["keep"], dy * c["scale"], dtype(0))
    elif "forced_scale" in c:
        dy = dy * c["forced_scale"]
    grads = {}
    B = dy.shape[0]
    if use_bn:
        gamma = st[bnp + ".weight"].astype(dtype)
        zhat = c["zhat"]
        dgamma = (dy * zhat).sum(axis=0, dtype=dtype)
        dbeta = dy.sum(axis=0, dtype=dtype)
        if c["bn_train"]:
            dz = gamma * c["rstd"] * (dy - dbeta / dtype(B) - zhat * (dgamma / dtype(B)))
        else:
            dz = dy * (gamma * c["rstd"])
        grads[bnp + ".weight"] = dgamma
        grads[bnp + ".bias"] = dbeta
    else:
        dz = dy
    W = st[lin + ".weight"].astype(dtype)
    grads[lin + ".weight"] = dz.T @ c["a_in"]
    grads[lin + ".bias"] = dz.sum(axis=0, dtype=dtype)
    return dz @ W, grads


def backward(st, cache, dout):
    """Gradients of every parameter and of the (flattened) input."""
    dtype = cache["dtype"]
    S, use_bn = cache["num_stage"], cache["use_bn"]
    names = hidden_layer_names(S)
    L = cache["layers"]
    dout = np.asarray(dout).astype(dtype)
    grads = {}
    W = st["w2.weight"].astype(dtype)
    grads["w2.weight"] = dout.T @ cache["h_last"]
    grads["w2.bias"] = dout.sum(axis=0, dtype=dtype)
    g = dout @ W
    for s in reversed(range(S)):
        i1, i2 = 1 + 2 * s, 2 + 2 * s
        gy, gr = _hidden_bwd(st, *names[i2], L[i2], g, use_bn, dtype)
        grads.update(gr)
        gh, gr = _hidden_bwd(st, *names[i1], L[i1], gy, use_bn, dtype)
        grads.update(gr)
        g = g + gh
    dx, gr = _hidden_bwd(st, *names[0], L[0], g, use_bn, dtype)
    grads.update(gr)
    return grads, dx


# --------------------------------------------------------------------------
# loss / metric / optimiser
# --------------------------------------------------------------------------
def mse_loss(pred, tgt, dtype=np.float32):
    """torch.nn.MSELoss(reduction='mean') and d loss / d pred  (train_1.py:37,94)."""
    pred = np.asarray(pred).astype(dtype)
    tgt = np.asarray(tgt).astype(dtype).reshape(pred.shape)
    d = pred - tgt
    n = d.size
    loss = (d * d).sum(dtype=dtype) / dtype(n)
    return loss, d * (dtype(2) / dtype(n))


def loss_mpjpe(pred, tgt, dtype=np.float32):
    """loss_MPJPE (train_1.py:19-23): per-joint sum over the batch of L2 errors."""
    pred = np.asarray(pred).astype(dtype)
    tgt = np.asarray(tgt).astype(dtype)
    err = np.sqrt(((pred - tgt) ** 2).sum(axis=-1, dtype=dtype))
    return err.sum(axis=0, dtype=dtype)


def epoch_mpjpe_mm(metric_sum, n_samples):
    """train_1.py:100-104: accumulate /len(dataset); mean over joints 1..16;
    * (17/16) * 1000 -> millimetres (reproduced literally)."""
    m = np.asarray(metric_sum, dtype=np.float64) / float(n_samples)
    return float(m[1:17].mean() * (17.0 / 16.0) * 1000.0)


def mpjpe_mm(pred, ref):
    """Mean per-joint position error in mm between two (B,17,3) predictions in metres
    (the parity gate of BASELINE.json: <= 1e-3 mm)."""
    pred = np.asarray(pred, dtype=np.float64).reshape(-1, 17, 3)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1, 17, 3)
    return float(np.sqrt(((pred - ref) ** 2).sum(-1)).mean() * 1000.0)


def adamw_step(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update (train_1.py:39,96), in the dtype of p.
    t is the 1-based step count.  Returns new (p, m, v)."""
    dt = p.dtype.type
    p = p * dt(1.0 - lr * wd)
    m = m + (g - m) * dt(1.0 - b1)
    v = v * dt(b2) + (g * g) * dt(1.0 - b2)
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    step_size = lr / bc1
    denom = np.sqrt(v) / dt(np.sqrt(bc2)) + dt(eps)
    p = p - dt(step_size) * (m / denom)
    return p.astype(dt), m.astype(dt), v.astype(dt)


def train_step(st, opt, x, y, *, num_stage=2, use_bn=True, p_dropout=0.5,
               keep_masks=None, lr=1e-4, wd=0.01, dtype=np.float32):
    """One train_1.py:75-100 step on numpy state.  opt = {'t', 'm':{}, 'v':{}}."""
    pred, cache = forward(st, x, num_stage=num_stage, train=True, use_bn=use_bn,
                          p_dropout=p_dropout, keep_masks=keep_masks, dtype=dtype)
    loss, dpred = mse_loss(pred, np.asarray(y).reshape(pred.shape), dtype)
    grads, _ = backward(st, cache, dpred)
    opt["t"] += 1
    for k, g in grads.items():
        if k not in opt["m"]:
            opt["m"][k] = np.zeros_like(st[k])
            opt["v"][k] = np.zeros_like(st[k])
        st[k], opt["m"][k], opt["v"][k] = adamw_step(
            st[k], g.astype(st[k].dtype), opt["m"][k], opt["v"][k], opt["t"], lr=lr, wd=wd)
    return loss, pred, grads


def flip_pose(data):
    """flip_pose restated from the text of /root/reference/phase3_direct/my_HybrIK/utils.py:372-396
    (that module imports cv2/seaborn and cannot be imported here): x -> 1-x for 2-D
    image-normalised keypoints, x -> -x for 3-D, then left joints [4,5,6,11,12,13] and right joints
    [1,2,3,14,15,16] trade places."""
    left, right = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    out = np.array(data, copy=True)
    if out.shape[-1] == 2:
        out[..., 0] = 1 - out[..., 0]
    elif out.shape[-1] == 3:
        out[..., 0] *= -1
    out[..., left + right, :] = out[..., right + left, :]
    return out


# ---------------------------------------------------------------------------------------------
# TriangleLoss of the phase5 cycle step
# ---------------------------------------------------------------------------------------------
def l1_mean(a, b):
    """torch.nn.L1Loss(reduction='mean'): value and the gradient w.r.t. a (the one w.r.t. b is its negative)."""
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return np.abs(d).mean(), np.sign(d) / d.size


def _centre_on_first(t):
    out = np.array(t, dtype=np.float64, copy=True)
    out[1:] -= out[0]
    return out


def _centre_on_first_bwd(g):
    """Gradient of `t[1:] -= t[0]`: entry 0 also collects minus the sum of the others."""
    out = np.array(g, copy=True)
    out[0] -= g[1:].sum(axis=0)
    return out


def triangle_loss(p2d, p3d, lift_gt, lift_pred, g2d, g3d, proj_pred=None, project=False, era="model2d"):
    """/root/reference/phase5_loop/losses.py:24-53 (era 'model2d') and the LinearModel-era copy
    /root/reference/phase5_loop/train_5 copy.py:50-70 (era 'lifter', without its projector branch).
    Returns (terms, grads) with grads for p2d, p3d, lift_pred, lift_gt and proj_pred."""
    grads = {k: 0.0 for k in ("p2d", "p3d", "lift_pred", "lift_gt", "proj")}
    l2, g = l1_mean(p2d, g2d); grads["p2d"] = grads["p2d"] + g
    l3, g = l1_mean(p3d, g3d); grads["p3d"] = grads["p3d"] + g
    if era == "model2d":
        ll, g = l1_mean(lift_pred, p3d)
        grads["lift_pred"] = grads["lift_pred"] + g
        grads["p3d"] = grads["p3d"] - g
        terms = [l2, l3, ll]
        if project:
            lp, g = l1_mean(_centre_on_first(proj_pred), _centre_on_first(p2d))
            grads["proj"] = grads["proj"] + _centre_on_first_bwd(g)
            grads["p2d"] = grads["p2d"] - _centre_on_first_bwd(g)
            terms.append(lp)
    else:
        ll, g = l1_mean(lift_gt, g3d); grads["lift_gt"] = grads["lift_gt"] + g
        lg, g = l1_mean(lift_pred, lift_gt)
        grads["lift_pred"] = grads["lift_pred"] + g
        grads["lift_gt"] = grads["lift_gt"] - g
        terms = [l2, l3, ll, lg]
    return terms, grads
