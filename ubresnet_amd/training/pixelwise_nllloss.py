"""PixelWiseNLLLoss with the reference's signature (training/pixelwise_nllloss.py:34-61), HIP kernels.

    crit = PixelWiseNLLLoss(weight=None, size_average=True, ignore_index=-100)
    loss = crit.forward(predict, target, pixelweights)      # or crit(...)

predict: (b,c,h,w) float32 log-softmax; target: (b,h,w) int64; pixelweights: (b,h,w) float32.
loss = mean over ALL b*h*w pixels of  -predict[b,target,h,w] * weight[target] * pixelweights
(ignore_index pixels contribute zero but stay in the denominator, as the reference's
reduce=False + torch.mean does, :51,:59).
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

# One module object for both import styles of the reference (`sys.path += $UBRESNET_MODELDIR; import pixelwise_nllloss`
# and `import ubresnet_amd.training.pixelwise_nllloss`): a top-level import re-binds itself to the package module.
if __name__ != "ubresnet_amd.training.pixelwise_nllloss":
    import importlib as _il
    sys.modules[__name__] = _il.import_module("ubresnet_amd.training.pixelwise_nllloss")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from ubresnet_amd import _lib as L  # noqa: E402
from ubresnet_amd import ops  # noqa: E402


def _assert_no_grad(variable):
    assert not variable.requires_grad, \
        "nn criterions don't compute the gradient w.r.t. targets - please " \
        "mark these variables as not requiring gradients"


class _LabelCheck:
    """F.nll_loss raises (device assert) for a target outside [0,C) other than ignore_index
    (training/pixelwise_nllloss.py:51); the HIP kernel counts such labels instead.  The count is copied to pinned host
    memory asynchronously and examined when it has landed -- at the next loss call (no host/device sync is added to
    the step), or immediately with UBR_CHECK_LABELS=sync; UBR_CHECK_LABELS=0 disables the report."""

    def __init__(self):
        self.pending = []
        self.free = []         # recycled (event, pinned int64) pairs: pinning memory is far too slow to do per step
        self.mode = os.environ.get("UBR_CHECK_LABELS", "async")

    def _raise(self, n, C, ign):
        raise RuntimeError("PixelWiseNLLLoss: %d target label(s) outside [0, %d) and different from ignore_index=%d "
                           "(F.nll_loss would assert); check the dataset's class ids" % (n, C, ign))

    def watch(self, bad_dev, C, ign):
        if self.mode == "0" or torch.cuda.is_current_stream_capturing():
            return
        ev, host = self.free.pop() if self.free else (torch.cuda.Event(), torch.empty(1, dtype=torch.int64).pin_memory())
        host.copy_(bad_dev.view(torch.int64), non_blocking=True)
        ev.record()
        self.pending.append((ev, host, int(C), int(ign)))
        if self.mode == "sync":
            ev.synchronize()
            self.poll()

    def flush(self):
        """wait for every outstanding count and report it now (end of an epoch, before a checkpoint, after validation)"""
        if torch.cuda.is_current_stream_capturing():
            return
        for ev, _, _, _ in self.pending:
            ev.synchronize()
        self.poll()

    def poll(self):
        if self.pending and torch.cuda.is_current_stream_capturing():
            return
        while self.pending and self.pending[0][0].query():
            ev, host, C, ign = self.pending.pop(0)
            n = int(host.item())
            self.free.append((ev, host))
            if n:
                self.pending.clear()
                self._raise(n, C, ign)


_label_check = _LabelCheck()


class _PixelNLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predict, target, pixelweights, classw, ignore_index, report_now=False):
        L.require_cuda(predict, "predict")
        if predict.dtype != torch.float32 or pixelweights.dtype != torch.float32 or target.dtype != torch.int64:
            raise RuntimeError("PixelWiseNLLLoss: expected predict/pixelweights float32 and target int64, got %s/%s/%s"
                               % (predict.dtype, pixelweights.dtype, target.dtype))
        if predict.dim() != 4 or tuple(target.shape) != (predict.shape[0], predict.shape[2], predict.shape[3]) \
                or tuple(pixelweights.shape) != tuple(target.shape):
            raise RuntimeError("PixelWiseNLLLoss: shape mismatch predict %s target %s pixelweights %s"
                               % (tuple(predict.shape), tuple(target.shape), tuple(pixelweights.shape)))
        predict, target, pixelweights = predict.contiguous(), target.contiguous(), pixelweights.contiguous()
        _label_check.poll()
        acc = torch.empty(L.STAT_SLOTS + 1, dtype=torch.float64, device=predict.device)   # [slots of the sum | bad-label count (u64)]
        ops.zero_(acc)
        ops.pixelwise_nll_fwd(predict, target, pixelweights, classw, ignore_index, acc, bad=acc[L.STAT_SLOTS:])
        _label_check.watch(acc[L.STAT_SLOTS:], predict.shape[1], ignore_index)
        if report_now:
            # a loss nobody back-propagates (validation, a one-off evaluation): there may be no "next call" to report at, and the
            # step-time argument for the asynchronous check does not apply -- report now, as F.nll_loss would
            _label_check.flush()
        loss = torch.empty((), dtype=torch.float32, device=predict.device)
        ops.cast_f64_to_f32(acc, loss, 1, 1.0 / float(target.numel()), stride=1)
        ctx.save_for_backward(target, pixelweights)
        ctx.classw, ctx.ignore_index, ctx.shape = classw, ignore_index, tuple(predict.shape)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        target, pixelweights = ctx.saved_tensors
        g = torch.empty(ctx.shape, dtype=torch.float32, device=target.device)
        g_loss = g_loss.contiguous().to(torch.float32)
        ops.pixelwise_nll_bwd(g_loss, target, pixelweights, ctx.classw, ctx.ignore_index, ctx.shape, g)
        return g, None, None, None, None, None


class PixelWiseNLLLoss(nn.modules.loss._WeightedLoss):
    def __init__(self, weight=None, size_average=True, ignore_index=-100):
        # modern torch folds size_average into `reduction`; keep the attributes the reference reads (:51)
        super(PixelWiseNLLLoss, self).__init__(weight, None, None, "mean")
        self.size_average = size_average
        self.ignore_index = ignore_index
        self.reduce = False

    @staticmethod
    def flush():
        """Out-of-range target labels are counted on the device and reported one loss call later (no host/device sync in the
        train step; a loss computed without gradients reports immediately).  Call this at the end of an epoch or before a
        checkpoint to raise for the last batches too.  The reference's F.nll_loss asserts immediately
        (training/pixelwise_nllloss.py:51)."""
        _label_check.flush()

    def forward(self, predict, target, pixelweights):
        """
        predict: (b,c,h,w) tensor with output from logsoftmax
        target:  (b,h,w) tensor with correct class
        pixelweights: (b,h,w) tensor with weights for each pixel
        """
        _assert_no_grad(target)
        _assert_no_grad(pixelweights)
        classw = self.weight
        if classw is not None:
            classw = classw.to(device=predict.device, dtype=torch.float32).contiguous()
        report_now = not (torch.is_grad_enabled() and predict.requires_grad)      # (decided here: grad mode is off inside Function.forward)
        return _PixelNLLFn.apply(predict, target, pixelweights, classw, self.ignore_index, report_now)
