"""Flat arenas and a one-launch Adam for modules with ordinary parameters (the conv models).

Replaces `torch.optim.Adam(model_direct.parameters(), lr)` + `optimizer.step()`
(/root/reference/phase4_joined/train.py:39,87; phase5_loop/train_5 copy.py:105-109,219-224) for Model_3D / Model_2D /
ResNet, the way optim.FlatAdamW does for the lifter:

  * ModuleArena: every parameter of the module becomes a view into ONE fp32 parameter arena (64-float aligned slots, the
    module's `parameters()` order), with gradient / Adam m / Adam v arenas of the same layout beside it.  state_dict() keys
    and shapes do not change (checkpoints load as before: load_state_dict copies into the views).
  * Parameter gradients are written by the library straight into the gradient arena: the autograd nodes of conv.py hand
    their weight / BatchNorm gradient kernels the arena view as output and return None to autograd (`p._pl_grad`, conv._pgrad);
    the first backward of a step overwrites, a second one in the same step (two forward passes: the phase5 Flip branch)
    accumulates.  No `p.grad += dw` launches, no layout copies.
  * FlatAdam.step(): ONE pl_adamw_flat launch over the whole arena (weight_decay 0 = torch.optim.Adam's default; Adam's
    coupled L2 decay is not implemented), in torch's single-tensor update order; the 1/world average of a data-parallel sum
    folds into it (grad_scale).  zero_grad() sets .grad to None and launches nothing.
A torch.optim.Optimizer subclass: LR schedulers and the stock Adam state_dict layout ('step', 'exp_avg', 'exp_avg_sq' per
parameter) work unchanged; capturable=True keeps the step count on the device for train.GraphedModuleStep.
"""
import torch

from . import _lib


class ModuleArena:
    def __init__(self, module):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("ModuleArena: the module has no parameters")
        dev = params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in params):
            raise ValueError("ModuleArena: parameters must be float32 on one device (build it after .to(device))")
        _lib.require_device_tensor(params[0].data.contiguous(), "parameters")
        self.params, self.offsets = params, []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(params, self.offsets):
            view = self.flat[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = None
            p._pl_grad = self.grad[o:o + p.numel()].view(p.shape)
        module._pl_arena = self

    def intact(self):
        """Every parameter still is its view of the arena (no .to() / .data reassignment since)."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def gather_grads(self):
        """Before the optimizer reads the gradient arena: a gradient some autograd node handed to autograd instead of writing
        it into the arena (an op outside conv.py) is copied in.  Returns the [lo, hi) arena ranges that hold a gradient."""
        runs = []
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None:
                continue
            if g.data_ptr() != p._pl_grad.data_ptr():
                p._pl_grad.copy_(g)
                p.grad = p._pl_grad
            end = o + (p.numel() + 63) // 64 * 64
            if runs and runs[-1][1] == o:
                runs[-1] = (runs[-1][0], end)
            else:
                runs.append((o, end))
        return runs


def arena_of(module):
    a = getattr(module, "_pl_arena", None)
    if a is None or not a.intact():
        a = ModuleArena(module)
    return a


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(module.parameters(), lr, betas, eps) as one library launch over the module's flat arenas."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if weight_decay != 0.0:
            raise NotImplementedError("FlatAdam: Adam's coupled L2 weight decay is not implemented (the reference uses 0)")
        self.arena = arena_of(module)
        super().__init__(self.arena.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0, capturable=bool(capturable)))
        a = self.arena
        self._m, self._v = torch.zeros_like(a.flat), torch.zeros_like(a.flat)
        self._t = 0
        dev = a.flat.device
        self._t_dev = torch.zeros(1, dtype=torch.int64, device=dev)           # completed steps (capturable)
        self._lr_dev = torch.full((1,), float(lr) if not torch.is_tensor(lr) else float(lr), dtype=torch.float32, device=dev)
        self._lr_host = None
        step = self._t_dev if capturable else torch.tensor(0.0)
        self._step_tensor = step
        for p, o in zip(a.params, a.offsets):
            n = p.numel()
            self.state[p] = {"step": step, "exp_avg": self._m[o:o + n].view(p.shape), "exp_avg_sq": self._v[o:o + n].view(p.shape)}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        a = self.arena
        t = 0
        for p, o in zip(a.params, a.offsets):               # loaded tensors are copies: back into the arenas
            st, n = self.state[p], p.numel()
            mv, vv = self._m[o:o + n].view(p.shape), self._v[o:o + n].view(p.shape)
            if "exp_avg" in st:
                mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
                t = max(t, int(st["step"]))
            st["exp_avg"], st["exp_avg_sq"], st["step"] = mv, vv, self._step_tensor
        self._t = t
        self._step_tensor.fill_(t)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        a, g = self.arena, self.param_groups[0]
        if not a.intact():
            raise _lib.PoseliftError("FlatAdam: the module's parameters left their arena (.to() / .data reassigned): "
                                     "build the optimizer after moving the module")
        runs = a.gather_grads()
        L, lr = _lib.lib(), g["lr"]
        cap = bool(g.get("capturable", False))
        with _lib.on_device(a.flat.device):
            if cap:
                if torch.is_tensor(lr):
                    lr_ptr = lr.data_ptr()                  # (a device scalar the caller changes between replays)
                else:
                    if self._lr_host != float(lr) and not torch.cuda.is_current_stream_capturing():
                        self._lr_dev.fill_(float(lr)); self._lr_host = float(lr)
                    lr_ptr = self._lr_dev.data_ptr()
                for lo, hi in runs:
                    _lib.check(L.pl_adamw_flat_dev(a.flat.data_ptr() + 4 * lo, a.grad.data_ptr() + 4 * lo, self._m.data_ptr() + 4 * lo,
                                                   self._v.data_ptr() + 4 * lo, hi - lo, lr_ptr, float(g["betas"][0]),
                                                   float(g["betas"][1]), float(g["eps"]), 0.0, 1, self._t_dev.data_ptr(),
                                                   float(grad_scale), _lib.current_stream_ptr()), "pl_adamw_flat_dev")
                _lib.check(L.pl_counter_add(self._t_dev.data_ptr(), 1, _lib.current_stream_ptr()), "pl_counter_add")
            else:
                self._t += 1
                for lo, hi in runs:
                    _lib.check(L.pl_adamw_flat(a.flat.data_ptr() + 4 * lo, a.grad.data_ptr() + 4 * lo, self._m.data_ptr() + 4 * lo,
                                               self._v.data_ptr() + 4 * lo, hi - lo, float(lr), float(g["betas"][0]),
                                               float(g["betas"][1]), float(g["eps"]), 0.0, self._t, float(grad_scale),
                                               _lib.current_stream_ptr()), "pl_adamw_flat")
                self._step_tensor.fill_(float(self._t))
        return loss
