"""ASPP_ResNet on the HIP path vs the reference fixture (tests/golden/aspp_ip16_1x3x64x96.npz, produced by
the reference's own models/ASPP_ResNet.py) and the CPU oracle.  Tolerances as in test_gpu_uresnet.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import uresnet_oracle as O
from ubresnet_amd import synthetic

if torch.cuda.is_available():
    from ubresnet_amd.models.ASPP_ResNet import ASPP_ResNet
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
    from test_gpu_uresnet import _grad_row, _grad_verdict, _rel


def _setup(golden_dir):
    g = np.load(os.path.join(golden_dir, "aspp_ip16_1x3x64x96.npz"))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.aspp_resnet_schema(3, C, 16), wseed)
    x, lab, wgt = synthetic.make_batch(B, H, W, seed0, planes=C)
    m = ASPP_ResNet(num_classes=3, in_channels=C, inplanes=16, showsizes=False)
    m.load_state_dict(sd)
    return g, sd, torch.from_numpy(x), torch.from_numpy(lab), torch.from_numpy(wgt), m.cuda()


def test_aspp_eval_forward(golden_dir):
    g, sd, xt, lt, wt, m = _setup(golden_dir)
    m.eval()
    with torch.no_grad():
        out = m(xt.cuda()).cpu()
    e = _rel(out, torch.from_numpy(g["logp_eval"]))
    print("aspp eval rel err", e)
    assert e <= 1e-3


def test_aspp_train_step(golden_dir):
    g, sd, xt, lt, wt, m = _setup(golden_dir)
    m.train()
    out = m.forward(xt.cuda())
    loss = PixelWiseNLLLoss()(out, lt.cuda(), wt.cuda())
    loss.backward()
    e = _rel(out.detach().cpu(), torch.from_numpy(g["logp_train"]))
    print("aspp train rel err", e)
    assert e <= 1e-3
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    _, g32, _, _ = O.train_step_grads(O.aspp_resnet_forward, sd, xt, lt, wt)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, g64, _, _ = O.train_step_grads(O.aspp_resnet_forward, sd64, xt.double(), lt, wt.double())
    params = dict(m.named_parameters())
    rows, fails = [], []
    zero_bias = {"conv1.bias", "conv10.bias"} | {n for n in g64 if n.endswith("_conv.bias")}   # each followed by a BatchNorm
    for n in g64:
        gv = params[n].grad.detach().cpu().double()
        if n in zero_bias:
            assert gv.abs().max().item() <= 1e-4, n
            continue
        rows.append(_grad_row(n, gv, g32[n].double(), g64[n]))
        fails += _grad_verdict(rows[-1], cos_min=0.999, l2_max=5e-2)   # measured: cos 0.99957, L2 2.9e-2 (67 BatchNorms + stride-1 max-pool arg-max ties: more discontinuities than UResNet)
    print("aspp worst grad (max-abs rel, l2 rel, min cos):", max(r[1] for r in rows), max(r[3] for r in rows), min(r[5] for r in rows))
    assert not fails, "; ".join(fails[:8])


def test_aspp_bf16_runs(golden_dir):
    g, sd, xt, lt, wt, m = _setup(golden_dir)
    m.eval()
    with torch.no_grad():
        ref = m(xt.cuda())
        m.compute_dtype = torch.bfloat16
        out = m(xt.cuda())
    agree = float((ref.argmax(1) == out.argmax(1)).float().mean())
    print("aspp bf16 pixel agreement", agree)
    assert agree >= 0.97


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_aspp_two_stream_backward_equals_single_stream(dt, monkeypatch):
    """BASELINE config 4's shape per rank (3x512x832, batch 2): the first backward of a fresh ASPP-ResNet on the
    two-stream schedule gives bitwise the single-stream gradients, run to run (dilated / wide-slot weight-gradient
    variants, the affine arena and the stride-1 pools included)."""
    sd = O.seeded_state_dict(O.aspp_resnet_schema(3, 3, 16), 42)
    x, lab, wgt = synthetic.make_batch(2, 512, 832, 1000, planes=3)
    xt, lt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda()
    crit = PixelWiseNLLLoss()
    res = {}
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("UBR_WGRAD_STREAM", mode)
        mm = ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False)
        mm.load_state_dict(sd)
        mm = mm.cuda().train()
        mm.compute_dtype = dt
        crit(mm(xt), lt, wt).backward()
        torch.cuda.synchronize()
        g = [p.grad.clone() for p in mm.parameters()]
        for n_, t in zip([n for n, _ in mm.named_parameters()], g):
            assert torch.isfinite(t).all(), n_
        if mode in res:
            for (n, _), a, b in zip(mm.named_parameters(), res[mode], g):
                assert torch.equal(a, b), "two-stream gradient of %s differs run to run" % n
        res[mode] = g
    for (n, _), a, b in zip(mm.named_parameters(), res["0"], res["1"]):
        assert torch.equal(a, b), "gradient of %s: two-stream schedule differs from the single-stream one" % n
