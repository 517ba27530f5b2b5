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
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return torch.float32


class _PassToken:
    """lives exactly as long as the autograd node of one forward pass (launch plans learn from it that a pass was dropped)"""
    __slots__ = ("__weakref__",)


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are explicit kernel schedules."""

    @staticmethod
    def forward(ctx, x, model, eng, dt, compat, *params):
        out, sv = eng.forward(x, model.training, dt, True)
        ctx.eng, ctx.sv, ctx.params, ctx.model, ctx.compat = eng, sv, params, model, compat
        plan = getattr(sv, "plan", None)
        if plan is not None:
            import weakref
            ctx.tok = _PassToken()
            plan.live = weakref.ref(ctx.tok)
        return out

    @staticmethod
    def backward(ctx, g_out):
        if ctx.sv is None:
            raise RuntimeError("ubresnet_amd: backward called twice on the same forward pass (saved activations were released)")
        if ctx.compat:
            # nn.DataParallel replica (parameters are non-leaf broadcast copies) or a DistributedDataParallel wrapper: the
            # gradients go back THROUGH autograd, so Broadcast's backward reduces them onto the master parameters and DDP's
            # AccumulateGrad hooks see them.  Slow path (autograd handles 165 tensors one by one, the pass is scheduled from
            # Python into a fresh flat buffer so nothing a later pass overwrites is handed out); GradAllReducer is the fast one.
            flat, views = ctx.eng.backward(ctx.sv, g_out, None, allow_plan=False)
            ctx.sv = None
            return (None, None, None, None, None) + tuple(views[id(p)] if p.requires_grad else None for p in ctx.params)
        hook = getattr(ctx.model, "_grad_ready_hook", None)
        accumulating = any(p.requires_grad and p.grad is not None for p in ctx.params)
        if hook is not None and accumulating:
            # The data-parallel reducer all-reduces ranges of the flat buffer in place WHILE backward is still running;
            # with an existing .grad the flat buffer would be added into .grad on the compute stream at the same time
            # and .grad would no longer alias what was reduced (ranks diverge).  Accumulated passes therefore skip the
            # early exchange: GradAllReducer.finish() reduces the accumulated .grad tensors instead.
            hook = None
            pending = getattr(ctx.model, "_grad_accum_pending", None)
            if pending is not None:
                pending()
        flat, views = ctx.eng.backward(ctx.sv, g_out, hook, allow_plan=not accumulating)
        ctx.sv = None
        # Parameter gradients are installed directly as views of the flat buffer (zero copy, and the
        # data-parallel all-reduce of `flat` IS the all-reduce of every .grad).  Handing them to autograd's
        # AccumulateGrad instead would clone each of the 165 tensors (it only steals unreferenced tensors).
        # Accumulation semantics are kept: an existing .grad is added to, never overwritten.
        if not accumulating:
            ctx.model.__dict__["_ubr_flat_grad"] = flat
        for p in ctx.params:
            if not p.requires_grad:
                continue
            g = views[id(p)]
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        return (None, None, None, None, None) + (None,) * len(ctx.params)


_warned = set()


def _wrapper_mode(params, x, model):
    """True when the reference's own wrap is in use (training/train_ubresnet2018_wlarcv2.py:99,103: nn.DataParallel; or a
    DistributedDataParallel wrapper): parameter gradients must then travel through autograd instead of being installed as
    views of the flat buffer -- the wrappers' reduction hooks would otherwise see nothing and the master parameters would
    silently train on no gradients.  The path works, at the cost the fused node exists to avoid; a one-time warning names the
    fast path (one process per GPU with ubresnet_amd.dist.GradAllReducer)."""
    if x.requires_grad:
        raise RuntimeError("ubresnet_amd: the input requires grad, but the HIP path computes no input gradient (no caller "
                           "of the reference needs one); detach the input")
    kind = None
    if any(p.requires_grad and not p.is_leaf for p in params):
        kind = "nn.DataParallel replica"
    elif getattr(torch.nn.parallel.DistributedDataParallel, "_active_ddp_module", None) is not None:
        kind = "DistributedDataParallel"
    if kind is not None and kind not in _warned:
        _warned.add(kind)
        import warnings
        warnings.warn("ubresnet_amd: running under %s: gradients take the autograd compatibility path (one tensor at a time, no "
                      "launch plans, no overlap of the exchange with backward).  The fast data-parallel path is one process "
                      "per GPU with ubresnet_amd.dist.GradAllReducer(model) -- see INTEGRATION.md." % kind)
    return kind is not None


def run_network(model, kind: str, x: torch.Tensor) -> torch.Tensor:
    eng = _engine(model, kind)
    dt = compute_dtype(model)
    params = tuple(p for _, p in eng.grad_order)
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if need_grad:
        compat = _wrapper_mode(params, x, model)
        return _NetFn.apply(x, model, eng, dt, compat, *params)
    out, _ = eng.forward(x, model.training, dt, False)
    return out
