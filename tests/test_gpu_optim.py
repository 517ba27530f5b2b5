"""Flat optimizers (ubresnet_amd/optim.py) against torch.optim on the same model, gradients and hyper-parameters
(reference: Adam(lr 1e-5, wd 1e-4) wlarcv2.py:155-157; SGD(momentum 0.9, wd 1e-4) wlarcv1.py:127-129)."""
import copy

import numpy as np
import pytest
import torch

import oracle.uresnet_oracle as O
from ubresnet_amd import synthetic

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ubresnet_amd.models.ub_uresnet import UResNet
    from ubresnet_amd.optim import FlatAdam, FlatSGD
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss


def _pair(num_classes=3):
    sd = O.seeded_state_dict(O.uresnet_schema(num_classes, 1, 16, 16), 42)
    ms = []
    for _ in range(2):
        m = UResNet(num_classes=num_classes, input_channels=1, inplanes=16)
        m.load_state_dict(sd)
        ms.append(m.cuda().train())
    return ms


def _batch(i):
    x, lab, wgt = synthetic.make_batch(2, 64, 64, 1000 + 7 * i)
    return torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda()


def _maxrel(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


@pytest.mark.parametrize("which", ["adam", "sgd", "sgd_nesterov"])
def test_flat_optimizer_tracks_torch_optim(which):
    ma, mb = _pair()
    crit = PixelWiseNLLLoss()
    if which == "adam":
        # a larger lr than the reference's 1e-5 so that three steps move the weights well above fp32 rounding
        oa = FlatAdam(ma, lr=1e-3, weight_decay=1e-4)
        ob = torch.optim.Adam(mb.parameters(), lr=1e-3, weight_decay=1e-4)
    else:
        nest = which == "sgd_nesterov"
        oa = FlatSGD(ma, lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=nest)
        ob = torch.optim.SGD(mb.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=nest)
    # parameters are now views of one buffer, values unchanged
    for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(p, q), n
    assert all(p.data_ptr() >= oa.flat.data_ptr() and p.data_ptr() < oa.flat.data_ptr() + 4 * oa.flat.numel() for p in ma.parameters())
    for i in range(3):
        x, lab, wgt = _batch(i)
        oa.zero_grad()
        crit(ma(x), lab, wgt).backward()
        # both optimizers see the SAME gradients (Adam divides by sqrt(v): with independently computed gradients a
        # rounding-level difference on a near-zero entry becomes an lr-sized difference in the update)
        for p, q in zip(ma.parameters(), mb.parameters()):
            q.grad = p.grad.detach().clone()
        oa.step()
        ob.step()
        torch.cuda.synchronize()
        worst = max(_maxrel(p.detach(), q.detach()) for p, q in zip(ma.parameters(), mb.parameters()))
        assert worst <= 2e-5, "step %d: parameters drift from torch.optim by %.3e" % (i, worst)
    # the flat gradient buffer was used directly (no gather): every .grad is a view of the model's flat buffer
    assert oa._flat_grad() is ma.__dict__["_ubr_flat_grad"]


def test_flat_adam_state_dict_interchanges_with_torch_adam():
    ma, mb = _pair()
    crit = PixelWiseNLLLoss()
    oa = FlatAdam(ma, lr=1e-3, weight_decay=1e-4)
    x, lab, wgt = _batch(0)
    crit(ma(x), lab, wgt).backward()
    oa.step()
    sd = oa.state_dict()
    # torch.optim.Adam accepts it ...
    mb.load_state_dict(ma.state_dict())
    ob = torch.optim.Adam(mb.parameters(), lr=1e-3, weight_decay=1e-4)
    ob.load_state_dict(copy.deepcopy(sd))
    # ... and after one more identical step both agree
    x, lab, wgt = _batch(1)
    oa.zero_grad()
    crit(ma(x), lab, wgt).backward()
    for p, q in zip(ma.parameters(), mb.parameters()):
        q.grad = p.grad.detach().clone()
    oa.step()
    ob.step()
    torch.cuda.synchronize()
    worst = max(_maxrel(p.detach(), q.detach()) for p, q in zip(ma.parameters(), mb.parameters()))
    assert worst <= 2e-5, worst
    # and the other way: a torch.optim.Adam state loads into FlatAdam
    mc = _pair()[0]
    mc.load_state_dict(mb.state_dict())
    oc = FlatAdam(mc, lr=1e-3, weight_decay=1e-4)
    oc.load_state_dict(ob.state_dict())
    assert oc.steps == 2
    x, lab, wgt = _batch(2)
    oc.zero_grad()
    crit(mc(x), lab, wgt).backward()
    for p, q in zip(mc.parameters(), mb.parameters()):
        q.grad = p.grad.detach().clone()
    oc.step()
    ob.step()
    torch.cuda.synchronize()
    worst = max(_maxrel(p.detach(), q.detach()) for p, q in zip(mc.parameters(), mb.parameters()))
    assert worst <= 2e-5, worst


def test_flat_adam_gathers_when_grads_are_not_the_flat_views():
    ma, mb = _pair()
    crit = PixelWiseNLLLoss()
    oa = FlatAdam(ma, lr=1e-3)
    ob = torch.optim.Adam(mb.parameters(), lr=1e-3)
    x, lab, wgt = _batch(0)
    crit(ma(x), lab, wgt).backward()
    first = ma.__dict__["_ubr_flat_grad"]
    g1 = first.clone()
    crit(ma(x), lab, wgt).backward()              # second backward ACCUMULATES into the first pass's buffers
    # the .grad tensors still alias the first pass's flat buffer, which now holds the accumulated sum: used as is
    assert oa._flat_grad() is first and torch.allclose(first, 2 * g1, rtol=1e-5, atol=1e-9)
    # gradients replaced by foreign tensors (e.g. clipped copies) are gathered into a scratch buffer instead
    ma.conv11.weight.grad = ma.conv11.weight.grad.clone()
    assert oa._flat_grad() is not first
    for p, q in zip(ma.parameters(), mb.parameters()):
        q.grad = p.grad.detach().clone()
    oa.step()
    ob.step()
    torch.cuda.synchronize()
    worst = max(_maxrel(p.detach(), q.detach()) for p, q in zip(ma.parameters(), mb.parameters()))
    assert worst <= 2e-5, worst


def test_step_without_gradients_changes_nothing():
    """torch.optim skips parameters that have no gradient: a step() with no backward before it must not decay weights or
    advance the moments (and a partially frozen model is refused, the flat kernel cannot skip ranges)"""
    ma, _ = _pair()
    for opt in (FlatAdam(ma, lr=1e-2, weight_decay=1e-1), FlatSGD(ma, lr=1e-2, momentum=0.9, weight_decay=1e-1)):
        before = opt.flat.clone()
        opt.zero_grad()
        opt.step()
        torch.cuda.synchronize()
        assert opt.steps == 0 and torch.equal(opt.flat, before)
    x, lab, wgt = _batch(0)
    PixelWiseNLLLoss()(ma(x), lab, wgt).backward()
    ma.conv11.bias.grad = None
    with pytest.raises(RuntimeError, match="no gradient"):
        opt.step()
