"""autograd / dtype plumbing between the mirror nn.Modules and the HIP graph executor."""
from __future__ import annotations

import torch

from .engine import Engine


def _engine(model, kind) -> Engine:
    eng = model.__dict__.get("_ubr_engine")
    # nn.DataParallel replicas are shallow copies: rebuild when the cached executor belongs to another module
    if eng is None or eng.model is not model:
        eng = Engine(model, kind)
        model.__dict__["_ubr_engine"] = eng
    return eng


def compute_dtype(model) -> torch.dtype:
    dt = getattr(model, "compute_dtype", None)
    if dt is not None:
        return dt
    if torch.is_autocast_enabled():
        return torch.get_autocast_gpu_dtype()
    return torch.float32


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are explicit kernel schedules."""

    @staticmethod
    def forward(ctx, x, model, eng, dt, *params):
        out, sv = eng.forward(x, model.training, dt, True)
        ctx.eng, ctx.sv, ctx.params, ctx.model = eng, sv, params, model
        return out

    @staticmethod
    def backward(ctx, g_out):
        if ctx.sv is None:
            raise RuntimeError("ubresnet_amd: backward called twice on the same forward pass (saved activations were released)")
        hook = getattr(ctx.model, "_grad_ready_hook", None)
        flat, views = ctx.eng.backward(ctx.sv, g_out, hook)
        ctx.sv = None
        ctx.model.__dict__["_ubr_flat_grad"] = flat
        # Parameter gradients are installed directly as views of the flat buffer (zero copy, and the
        # data-parallel all-reduce of `flat` IS the all-reduce of every .grad).  Handing them to autograd's
        # AccumulateGrad instead would clone each of the 165 tensors (it only steals unreferenced tensors).
        # Accumulation semantics are kept: an existing .grad is added to, never overwritten.
        for p in ctx.params:
            if not p.requires_grad:
                continue
            g = views[id(p)]
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        return (None, None, None, None) + (None,) * len(ctx.params)


def run_network(model, kind: str, x: torch.Tensor) -> torch.Tensor:
    eng = _engine(model, kind)
    dt = compute_dtype(model)
    params = tuple(p for _, p in eng.grad_order)
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if need_grad:
        return _NetFn.apply(x, model, eng, dt, *params)
    out, _ = eng.forward(x, model.training, dt, False)
    return out
