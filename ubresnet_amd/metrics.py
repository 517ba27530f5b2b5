"""Segmentation metrics on the device: accuracy() of the reference
(training/train_ubresnet2018_wlarcv2.py:509-566) and per-class pixel IoU, both from ONE pass
that builds the C x C confusion matrix (no per-class reductions, one host sync)."""
import torch

from . import _lib as L
from . import ops


def confusion_matrix(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """output: [B,C,H,W] float32 (log-)scores, target [B,H,W] int64 -> int64 [C,C] counts [true, pred].
    pred = first arg-max over channels (the tie-break of ``output.max(1)``)."""
    L.require_cuda(output, "output")
    if output.dtype != torch.float32 or target.dtype != torch.int64:
        raise RuntimeError("confusion_matrix: expected float32 scores and int64 target")
    C = output.shape[1]
    cm = torch.empty(C * C, dtype=torch.int64, device=output.device)
    ops.zero_(cm)
    ops.confusion(output.contiguous(), target.contiguous(), cm)
    return cm.view(C, C)


def accuracy(output: torch.Tensor, target: torch.Tensor):
    """Same return value as the reference's accuracy(): per-class recall in percent, then the total."""
    cm = confusion_matrix(output, target).cpu().double()   # the single host sync
    res = []
    for c in range(cm.shape[0]):
        n = cm[c].sum().item()
        res.append(100.0 * cm[c, c].item() / n if n > 0 else 0.0)
    tot = cm.sum().item()
    res.append(100.0 * cm.diag().sum().item() / tot if tot > 0 else 0.0)
    return res


def iou(cm: torch.Tensor):
    """per-class IoU_c = |A_c & B_c| / |A_c | B_c| from a confusion matrix"""
    cm = cm.cpu().double()
    out = []
    for c in range(cm.shape[0]):
        inter = cm[c, c]
        union = cm[c, :].sum() + cm[:, c].sum() - inter
        out.append(float(inter / union) if union > 0 else float("nan"))
    return out
