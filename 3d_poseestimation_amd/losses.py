"""L1 loss and the TriangleLoss of the phase5 cycle step, on the HIP library.

  l1_loss        torch.nn.L1Loss(reduction="mean")
  TriangleLoss   /root/reference/phase5_loop/losses.py:10-62            (era="model2d")
                 /root/reference/phase5_loop/train_5 copy.py:34-86      (era="lifter", the LinearModel-era
                 copy: |y2d^ - y2d| + |y3d^ - y3d| + |lift(y2d) - y3d| + |lift(y2d^) - lift(y2d)|)
All L1 terms of one TriangleLoss call -- value and gradient -- are ONE pl_l1_terms_fwd_bwd launch pair.
The reference reads four scalars back to the host in every call (`.cpu().item()`, losses.py:50-53);
here the per-term values stay on the device until report_losses() asks for them.
"""
import ctypes

import torch

from . import _lib


class _L1TermsFn(torch.autograd.Function):
    """losses[t] = mean |a_t - b_t| for t < len(tensors) / 2; inputs come as a0, b0, a1, b1, ..."""

    @staticmethod
    def forward(ctx, *tensors):
        nt = len(tensors) // 2
        if len(tensors) % 2 or not 1 <= nt <= _lib.L1_MAX_TERMS:
            raise ValueError(f"L1 terms: need 1..{_lib.L1_MAX_TERMS} (a, b) pairs")
        dev = tensors[0].device
        terms = (_lib.PLL1Term * nt)()
        keep, grads = [], []
        for t in range(nt):
            a, b = tensors[2 * t], tensors[2 * t + 1]
            if a.shape != b.shape:
                raise ValueError(f"L1 term {t}: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
            a, b = a.contiguous(), b.contiguous()
            _lib.require_device_tensor(a, f"term {t} input")
            _lib.require_device_tensor(b, f"term {t} target")
            da = torch.empty_like(a) if ctx.needs_input_grad[2 * t] else None
            db = torch.empty_like(b) if ctx.needs_input_grad[2 * t + 1] else None
            terms[t] = _lib.PLL1Term(a.data_ptr(), b.data_ptr(), a.numel(),
                                     da.data_ptr() if da is not None else None,
                                     db.data_ptr() if db is not None else None)
            keep += [a, b]
            grads += [da, db]
        losses = torch.empty(nt, dtype=torch.float32, device=dev)
        scratch = torch.empty(_lib.lib().pl_l1_scratch_bytes(nt), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = _lib.lib().pl_l1_terms_fwd_bwd(terms, nt, 1.0, losses.data_ptr(), scratch.data_ptr(),
                                                _lib.current_stream_ptr())
        _lib.check(rc, "pl_l1_terms_fwd_bwd")
        ctx.grads = grads
        return losses

    @staticmethod
    def backward(ctx, g):
        out = []
        for k, d in enumerate(ctx.grads):
            out.append(d.mul_(g[k // 2]) if d is not None else None)
        ctx.grads = None
        return tuple(out)


def l1_terms(*pairs):
    """Mean-absolute-error of several (a, b) pairs in one launch; returns a [len(pairs)] tensor."""
    flat = []
    for a, b in pairs:
        flat += [a, b]
    return _L1TermsFn.apply(*flat)


def l1_loss(pred, tgt):
    """torch.nn.L1Loss(reduction="mean")(pred, tgt)."""
    return l1_terms((pred, tgt))[0]


def _centre_on_first(t):
    """`t[1:] -= t[0]` as the reference writes it (losses.py:33-36): every batch entry but the first is
    made relative to the first one.  Out of place here; same values."""
    return torch.cat([t[:1], t[1:] - t[0]], dim=0)


class TriangleLoss(torch.nn.Module):
    def __init__(self, Project=False, era="lifter"):
        super().__init__()
        if era not in ("lifter", "model2d"):
            raise ValueError("era is 'lifter' (train_5 copy.py:34-86) or 'model2d' (losses.py:10-62)")
        self.Project, self.era = Project, era
        self._log = []            # per call: the device tensor of term values (no host sync here)

    def forward(self, predicted_2d, predicted_3d, lift_2d_gt, lift_2d_pred, gt_2d, gt_3d,
                proj_3d_pred=None, proj_3d_gt=None):
        if self.era == "lifter":
            # train_5 copy.py:50-54: 2-D, 3-D, domain gap lift(y2d^) vs lift(y2d), lift(y2d) vs y3d
            pairs = [(predicted_2d, gt_2d), (predicted_3d, gt_3d), (lift_2d_gt, gt_3d), (lift_2d_pred, lift_2d_gt)]
            if self.Project:      # :56-68
                pp, pg = _centre_on_first(proj_3d_pred), _centre_on_first(proj_3d_gt)
                pairs += [(pg, _centre_on_first(gt_2d)), (pp, pg)]
            terms = l1_terms(*pairs)
            self._log.append(terms.detach())
            return terms.sum()
        # losses.py:24-53: 2-D, 3-D, lift(y2d^) vs y3d^ (+ projector vs the centred 2-D prediction)
        pairs = [(predicted_2d, gt_2d), (predicted_3d, gt_3d), (lift_2d_pred, predicted_3d)]
        if self.Project:
            pairs.append((_centre_on_first(proj_3d_pred), _centre_on_first(predicted_2d)))
        terms = l1_terms(*pairs)
        self._log.append(terms.detach())
        loss_proj = terms[3] if self.Project else 0
        return terms.sum(), terms[0], terms[1], terms[2], loss_proj

    def term_means(self):
        """Mean of every logged term since the last report (one host read), as a list."""
        if not self._log:
            return []
        return torch.stack(self._log).mean(dim=0).cpu().tolist()

    def report_losses(self):
        m = self.term_means()
        if m:
            print(*m[:4 if self.era == "lifter" else 3])
        self._log = []
