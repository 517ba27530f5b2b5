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
    # every log-probability element (train-mode BatchNorm keeps them O(10)): |a - b| <= 1e-3 |b| + 1e-4
    ref = torch.from_numpy(g["logp_train"])
    worst = ((out.detach().cpu() - ref).abs() - 1e-3 * ref.abs()).max().item()
    assert worst <= 1e-4, "aspp train log-probabilities: worst |a-b| - 1e-3|b| = %.3e" % worst
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


def test_aspp_eval_normalised_fixture(golden_dir):
    """eval forward on reference-calibrated running statistics (tests/golden/make_golden.py --only aspp): with normalised
    activations every element is held to |a - b| <= 1e-3 |b| + 1e-4 in fp32 (the un-normalised eval fixture above reaches
    1e5 in its logits and can only be compared against its global scale); bf16 / fp16 storage against the fp32 class map."""
    g = np.load(os.path.join(golden_dir, "aspp_ip16_norm_1x3x64x96.npz"))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.state_dict_with_bn_stats(O.seeded_state_dict(O.aspp_resnet_schema(3, C, 16), wseed), g["bn_keys"], g["bn_stats"])
    x = torch.from_numpy(synthetic.make_batch(B, H, W, seed0, planes=C)[0]).cuda()
    m = ASPP_ResNet(num_classes=3, in_channels=C, inplanes=16, showsizes=False)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    ref = torch.from_numpy(g["logp_eval"])
    with torch.no_grad():
        out = m(x).cpu()
    worst = ((out - ref).abs() - 1e-3 * ref.abs()).max().item()
    print("aspp normalised eval: max abs err %.3e, worst |a-b| - 1e-3|b| %.3e" % ((out - ref).abs().max().item(), worst))
    assert worst <= 1e-4
    top2 = torch.topk(ref, 2, dim=1)[0]
    for dt, tol in ((torch.float16, 0.1), (torch.bfloat16, 1.0)):      # measured 0.03 / 0.62 on log-probabilities up to 22.8
        m.compute_dtype = dt
        with torch.no_grad():
            o = m(x).cpu()
        err = (o - ref).abs().max().item()
        safe = (top2[:, 0] - top2[:, 1]) > 2 * tol
        assert err <= tol, "%s eval log-probabilities: max abs err %.3f" % (dt, err)
        assert torch.equal(o.argmax(1)[safe], ref.argmax(1)[safe])


def test_aspp_full_size_matches_reference_summary(golden_dir):
    """BASELINE.json configs[3] at its real per-image size (ASPP_ResNet ip16, 1 x 3 x 512 x 832, fp32) against what the
    reference's own models/ASPP_ResNet.py produced there: sampled log-probabilities (eval and train mode), class counts,
    the loss, every gradient tensor's norm and sampled entries."""
    import hashlib
    g = np.load(os.path.join(golden_dir, "aspp_ip16_1x3x512x832_summary.npz"))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.aspp_resnet_schema(3, C, 16), wseed)
    x, lab, wgt = synthetic.make_batch(B, H, W, seed0, planes=C)
    m = ASPP_ResNet(num_classes=3, in_channels=C, inplanes=16, showsizes=False)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    idx = g["sample_idx"]
    with torch.no_grad():
        ev = m(torch.from_numpy(x).cuda()).cpu().numpy()
    # eval mode on seeded (un-normalising) running statistics: logits reach 5e5, errors scale with them
    err = float(np.abs(ev.reshape(-1)[idx] - g["sample_logp_eval"]).max()) / float(g["absmax_eval"])
    assert err <= 1e-4, "eval samples: %.3e of the largest log-probability" % err
    am = ev.argmax(1).astype(np.uint8)
    nlow = len(g["low_margin_idx_eval"])
    if nlow == 0:
        assert hashlib.sha256(am.tobytes()).hexdigest() == str(g["argmax_sha256_eval"])
    assert np.abs(np.bincount(am.reshape(-1), minlength=3) - g["class_counts_eval"]).sum() <= 2 * nlow
    m.train()
    out = m.forward(torch.from_numpy(x).cuda())
    loss = PixelWiseNLLLoss().forward(out, torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda())
    loss.backward()
    torch.cuda.synchronize()
    got, ref_t = out.detach().cpu().numpy().reshape(-1)[idx], g["sample_logp_train"]
    worst = float((np.abs(got - ref_t) - 1e-3 * np.abs(ref_t)).max())
    assert worst <= 1e-4, "train log-probability samples: worst |a-b| - 1e-3|b| = %.3e" % worst
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    am_t = out.detach().argmax(1).cpu().numpy().astype(np.uint8)
    nlow_t = len(g["low_margin_idx_train"])
    assert np.abs(np.bincount(am_t.reshape(-1), minlength=3) - g["class_counts_train"]).sum() <= 2 * nlow_t + 2
    params = dict(m.named_parameters())
    bad, worst_norm = [], 0.0
    for name, nref in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        gv = params[name].grad.detach().cpu().double().reshape(-1)
        if name in ("conv1.bias", "conv10.bias") or name.endswith("_conv.bias"):          # analytically zero (a BatchNorm follows)
            assert gv.abs().max().item() <= 1e-3 * max(1.0, float(nref)), name
            continue
        nrm = float(torch.sqrt((gv * gv).sum()))
        worst_norm = max(worst_norm, abs(nrm - float(nref)) / float(nref))
        if abs(nrm - float(nref)) > 2e-2 * float(nref):
            bad.append("%s norm %.6e vs %.6e" % (name, nrm, float(nref)))
        rs = np.random.RandomState(7)
        sidx = np.sort(rs.choice(gv.numel(), size=min(16, gv.numel()), replace=False))
        sref = g["gs__" + name].astype(np.float64)
        tol_abs = 5e-2 * max(float(np.abs(sref).max()), float(nref) / np.sqrt(gv.numel()))
        if np.abs(gv.numpy()[sidx] - sref).max() > tol_abs:
            bad.append("%s samples max err %.3e (tol %.3e)" % (name, np.abs(gv.numpy()[sidx] - sref).max(), tol_abs))
    print("aspp full size: eval sample err %.2e of scale, train worst %.2e, worst grad-norm rel %.2e" % (err, worst, worst_norm))
    assert not bad, "; ".join(bad[:6])


def test_aspp_level_backward_tie_free():
    """ASPP + ASPP_post of one encoder level (engine.aspp_level_fwd / aspp_level_bwd: four dilated conv branches, the
    stride-1 max pool, the 1x1 over the virtual five-way concat) on a DENSE random input: no two window entries are equal, so
    the pool's arg-max is unambiguous and every gradient must agree with the oracle's autograd to fp32 accuracy.  (On the
    network's own post-ReLU activations the pool windows are full of exact ties and near-ties, which is what loosens the
    whole-network gradient gate of test_aspp_train_step.)"""
    from collections import OrderedDict
    import torch.nn as nn
    from ubresnet_amd.engine import Engine, Saved
    from ubresnet_amd.models.ASPP_ResNet import ASPP, ASPP_post
    Cn, N, h, w = 128, 2, 24, 40

    class Level(nn.Module):
        def __init__(self):
            super().__init__()
            self.layer, self.post = ASPP(Cn), ASPP_post(64 + Cn, Cn)

        def _grad_completion_order(self):
            return self.post._grad_completion_order("post.") + self.layer._grad_completion_order("layer.")

    torch.manual_seed(5)
    lvl = Level()
    with torch.no_grad():
        for mod in lvl.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.3, 0.3)
    sd = OrderedDict((k, v.detach().clone()) for k, v in lvl.state_dict().items())
    lvl = lvl.cuda().train()
    dev = torch.device("cuda", 0)
    eng = Engine(lvl, "custom")
    sv = Saved()
    eng._alloc_pass_workspaces(sv, dev, True)
    eng._save = True
    eng.pack_all(torch.float32, dev, "fwd")
    eng.pack_all(torch.float32, dev, "bwd")
    arena, offs = eng._affine_arena(sv, dev, [64 + Cn], [(0, 64)])
    for b, (_, bn, _, _) in enumerate(lvl.layer.branches()):
        eng._bind_site(eng.bn(bn), arena, offs[0] + 16 * b)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(N, Cn, h, w, generator=gen)
    g_post, g_base = torch.randn(N, Cn, h, w, generator=gen), torch.randn(N, Cn, h, w, generator=gen)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    e, cpost = nh(x), torch.empty((N, h, w, Cn), device=dev)
    rec = eng.aspp_level_fwd(lvl.layer, lvl.post, e, cpost, arena, offs[0], True, torch.float32)
    flat = torch.zeros(eng.grad_numel, device=dev)
    views = {}
    for name, p in eng.grad_order:
        o = eng.grad_offsets[name]
        views[id(p)] = flat[o:o + p.numel()].view(p.shape)
    g_e = eng.aspp_level_bwd(rec, nh(g_post), nh(g_base), lambda p: views[id(p)])
    eng._wg_flush()        # (a whole backward pass sums the weight-gradient slabs once per stage: Engine._stage_notifier)
    torch.cuda.synchronize()
    # oracle
    p = OrderedDict((k, v.clone().requires_grad_(True) if O.is_param_key(k) else v.clone()) for k, v in sd.items())
    xr = x.clone().requires_grad_(True)
    y = O.aspp_post(p, "post", O.aspp(p, "layer", xr, True, None), True, None)
    ((y * g_post).sum() + (xr * g_base).sum()).backward()
    ps = eng.bn(lvl.post.ASPP_bn)
    act = torch.relu((cpost - ps.mean) * ps.scale + ps.shift).permute(0, 3, 1, 2).cpu()
    assert _rel(act, y.detach()) <= 1e-4, "ASPP level forward"
    ge = g_e.permute(0, 3, 1, 2).cpu()
    assert _rel(ge, xr.grad) <= 5e-4, "ASPP level input gradient: %.3e" % _rel(ge, xr.grad)
    for k, v in p.items():
        if not O.is_param_key(k):
            continue
        got = views[id(dict(lvl.named_parameters())[k])].cpu()
        if k.endswith("_conv.bias"):
            assert got.abs().max().item() <= 1e-3 * max(1.0, float(g_post.abs().sum()) ** 0.5)        # analytically zero (a BatchNorm follows)
            continue
        assert _rel(got, v.grad) <= 5e-4, "gradient of %s: %.3e" % (k, _rel(got, v.grad))
