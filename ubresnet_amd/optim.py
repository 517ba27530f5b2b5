"""Flat optimizers: the whole network's parameters in ONE fp32 buffer, one HIP launch per step.

The reference trains with ``torch.optim.Adam(model.parameters(), lr=1e-5, weight_decay=1e-4)``
(training/train_ubresnet2018_wlarcv2.py:155-157) and, in the LArCV1 scripts, ``torch.optim.SGD(..., momentum=0.9,
weight_decay=1e-4)`` (training/train_ubresnet2018_wlarcv1.py:127-129).  The backward pass of this package already
leaves every gradient as a view of one flat buffer (ordered by completion time, so data-parallel buckets can leave
early); ``FlatAdam`` / ``FlatSGD`` re-point every ``parameter.data`` at a view of a parameter buffer with the SAME
layout, so an optimizer step is a single streaming kernel over (param, grad, state) -- ``ubr_adam_step`` /
``ubr_sgd_step`` -- instead of a multi-tensor launch sequence over 165 tensors.

    opt = FlatAdam(model, lr=1e-5, weight_decay=1e-4)        # after model.to(device)
    loss.backward(); reducer.finish(); opt.step(); opt.zero_grad()

Arithmetic is torch.optim's (L2 weight decay added to the gradient; Adam bias corrections; SGD's first step copies
the gradient into the momentum buffer).  ``state_dict()`` / ``load_state_dict()`` use torch.optim's layout (per
parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` or ``momentum_buffer``, parameters numbered in
``model.parameters()`` order), so optimizer state in a reference checkpoint
(``{"iter","epoch","state_dict","best_prec1","optimizer"}``, wlarcv2.py:474-479) loads and saves unchanged.
"""
from __future__ import annotations

import torch

from . import _lib as L

__all__ = ["FlatAdam", "FlatSGD"]


def _kind_of(model) -> str:
    return "aspp" if hasattr(model, "ASPP_layer_enc3") else "uresnet"


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, model, defaults):
        from .autograd_fn import _engine
        params = list(model.parameters())
        if not params:
            raise ValueError("optimizer got a model without parameters")
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda:
                raise RuntimeError("ubresnet_amd.optim: parameters must be float32 on one ROCm device (move the model first)")
        super().__init__(params, defaults)
        self.model = model
        eng = _engine(model, _kind_of(model))
        self._layout = [(name, p, eng.grad_offsets[name]) for name, p in eng.grad_order]
        self._numel = eng.grad_numel
        if len(self._layout) != len(params):
            raise RuntimeError("ubresnet_amd.optim: gradient layout does not cover every parameter")
        self._index = {id(p): i for i, p in enumerate(params)}          # torch.optim numbering (state_dict)
        self.flat = torch.zeros(self._numel, dtype=torch.float32, device=dev)
        self._adopt()
        self._scratch = None
        self.steps = 0

    # parameters become views of self.flat (values preserved)
    def _adopt(self):
        with torch.no_grad():
            for _, p, o in self._layout:
                n = p.numel()
                v = self.flat[o:o + n].view(p.shape)
                if p.data_ptr() != v.data_ptr():
                    v.copy_(p.data)
                    p.data = v

    def _flat_grad(self):
        """the flat gradient buffer of the last backward if every .grad is still its view, else a gathered copy"""
        g = self.model.__dict__.get("_ubr_flat_grad")
        ok = g is not None and g.numel() == self._numel and g.device == self.flat.device
        if ok:
            base = g.data_ptr()
            for _, p, o in self._layout:
                if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                    ok = False
                    break
        if ok:
            return g
        have = [p.grad is not None for _, p, _ in self._layout]
        if not any(have):
            return None          # torch.optim skips parameters without a gradient: nothing to do
        if not all(have):
            raise RuntimeError("ubresnet_amd.optim: some parameters have no gradient (frozen / requires_grad=False); the flat "
                               "one-launch step updates every parameter -- use torch.optim for partially frozen models")
        if self._scratch is None:
            self._scratch = torch.zeros_like(self.flat)
        else:
            self._scratch.zero_()
        for _, p, o in self._layout:
            if p.grad is not None:
                self._scratch[o:o + p.numel()].copy_(p.grad.reshape(-1))
        return self._scratch

    def _check_views(self):
        base = self.flat.data_ptr()
        for _, p, o in self._layout:
            if p.data_ptr() != base + 4 * o:
                self._adopt()            # e.g. load_state_dict / .to() replaced parameter storage
                return

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)

    # ---- torch.optim-compatible state (per-parameter views of the flat state buffers) ----
    def _state_views(self, buf):
        return {self._index[id(p)]: buf[o:o + p.numel()].view(p.shape) for _, p, o in self._layout}


class FlatAdam(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        self._check_views()
        g = self._flat_grad()
        if g is None:
            return loss
        grp = self.param_groups[0]
        self.steps += 1
        L.check(L.lib().ubr_adam_step(self.flat.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                      self._numel, float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                                      float(grp["weight_decay"]), self.steps, float(grad_scale), L.stream_ptr()), "adam_step")
        return loss

    def state_dict(self):
        m, v = self._state_views(self.exp_avg), self._state_views(self.exp_avg_sq)
        n = len(self._index)
        state = {}
        if self.steps > 0:
            state = {i: {"step": torch.tensor(float(self.steps)), "exp_avg": m[i].clone(), "exp_avg_sq": v[i].clone()} for i in range(n)}
        grp = {k: v_ for k, v_ in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(n))
        return {"state": state, "param_groups": [grp]}

    def load_state_dict(self, sd):
        grp = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in grp:
                self.param_groups[0][k] = grp[k]
        m, v = self._state_views(self.exp_avg), self._state_views(self.exp_avg_sq)
        self.steps = 0
        for i, st in sd.get("state", {}).items():
            i = int(i)
            m[i].copy_(st["exp_avg"])
            v[i].copy_(st["exp_avg_sq"])
            self.steps = max(self.steps, int(float(st["step"])))


class FlatSGD(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(model, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))
        self.momentum_buffer = torch.zeros_like(self.flat) if momentum != 0 else None

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        self._check_views()
        g = self._flat_grad()
        if g is None:
            return loss
        grp = self.param_groups[0]
        first = self.steps == 0
        self.steps += 1
        L.check(L.lib().ubr_sgd_step(self.flat.data_ptr(), g.data_ptr(), L.ptr(self.momentum_buffer), self._numel, float(grp["lr"]),
                                     float(grp["momentum"]), float(grp["dampening"]), float(grp["weight_decay"]),
                                     1 if grp["nesterov"] else 0, 1 if first else 0, float(grad_scale), L.stream_ptr()), "sgd_step")
        return loss

    def state_dict(self):
        n = len(self._index)
        state = {}
        if self.momentum_buffer is not None and self.steps > 0:
            b = self._state_views(self.momentum_buffer)
            state = {i: {"momentum_buffer": b[i].clone()} for i in range(n)}
        grp = {k: v_ for k, v_ in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(n))
        return {"state": state, "param_groups": [grp]}

    def load_state_dict(self, sd):
        grp = sd["param_groups"][0]
        for k in ("lr", "momentum", "dampening", "weight_decay", "nesterov"):
            if k in grp:
                self.param_groups[0][k] = grp[k]
        if self.momentum_buffer is not None:
            b = self._state_views(self.momentum_buffer)
            for i, st in sd.get("state", {}).items():
                if st.get("momentum_buffer") is not None:
                    b[int(i)].copy_(st["momentum_buffer"])
                    self.steps = max(self.steps, 1)
