"""Whole-network parity (GPU): UResNet on the HIP path vs (a) the golden fixtures produced by the
reference's own code and (b) the CPU oracle on the same seeded weights and inputs.

Tolerances (BASELINE.json north_star): fp32 log-probabilities within 1e-3 relative, class-map
arg-max bit-exact wherever the top-2 margin exceeds the tolerance; gradients within 2e-3 of
each tensor's scale (fp32, different summation order over up to 8k pixels x 52 BatchNorms).
bf16: reported/loosely bounded (storage rounding), IoU vs the fp32 class map >= 0.98.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import uresnet_oracle as O
from ubresnet_amd import synthetic

if torch.cuda.is_available():
    from ubresnet_amd.models.ub_uresnet import UResNet
    from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss
    from ubresnet_amd import metrics

torch.set_num_threads(min(16, os.cpu_count() or 1))


def _model(sd, num_classes=3, cin=1, ip=16):
    m = UResNet(num_classes=num_classes, input_channels=cin, inplanes=ip)
    m.load_state_dict(sd)
    return m.to("cuda")


def _rel(a, b):
    return float((a - b).abs().max() / max(b.abs().max().item(), 1e-12))


@pytest.mark.parametrize("tag", ["2x1x64x64", "1x1x96x128"])
def test_forward_matches_reference_fixture(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_%s.npz" % tag))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.uresnet_schema(3, C, 16, 16), wseed)
    x, lab, wgt = synthetic.make_batch(B, H, W, seed0)
    m = _model(sd)
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(x).cuda()).cpu()
    ref = torch.from_numpy(g["logp_eval"])
    assert _rel(out, ref) <= 1e-3, "eval log-probs rel err %.3e" % _rel(out, ref)
    top2 = torch.topk(ref, 2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 2e-3 * ref.abs().max()
    assert torch.equal(out.argmax(1)[safe], ref.argmax(1)[safe]), "class map differs where the margin is safe"
    print("eval rel err", tag, _rel(out, ref))


@pytest.mark.parametrize("tag", ["2x1x64x64", "1x1x96x128"])
def test_train_step_matches_reference_fixture_and_oracle(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_%s.npz" % tag))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.uresnet_schema(3, C, 16, 16), wseed)
    x, lab, wgt = synthetic.make_batch(B, H, W, seed0)
    xt, lt, wt = torch.from_numpy(x), torch.from_numpy(lab), torch.from_numpy(wgt)
    m = _model(sd)
    m.train()
    crit = PixelWiseNLLLoss()
    out = m.forward(xt.cuda())
    loss = crit.forward(out, lt.cuda(), wt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    ref_out = torch.from_numpy(g["logp_train"])
    e = _rel(out.detach().cpu(), ref_out)
    assert e <= 1e-3, "train log-probs rel err %.3e" % e
    # per-element bar (train-mode log-probabilities are O(10), so this is the meaningful form of "1e-3 relative")
    d = (out.detach().cpu() - ref_out).abs()
    assert bool((d <= 1e-3 * ref_out.abs() + 1e-4).all()), "train log-probs: worst per-element excess %.3e" % float((d - 1e-3 * ref_out.abs()).max())
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    # running statistics follow nn.BatchNorm2d (momentum 0.1, unbiased variance)
    after = m.state_dict()
    for k in ("bn1", "bn10"):
        assert _rel(after[k + ".running_mean"].cpu(), torch.from_numpy(g[k + "_running_mean"])) <= 1e-4
        assert _rel(after[k + ".running_var"].cpu(), torch.from_numpy(g[k + "_running_var"])) <= 1e-4
    assert int(after["bn1.num_batches_tracked"]) == int(g["nbt"])
    # Gradients.  ReLU masks make the gradient a discontinuous function of the activations: a pixel whose
    # BatchNorm output is within ~1e-6 of zero flips its mask under ANY change of summation order, and
    # every flip perturbs all upstream gradients.  The fp32 CPU oracle itself is 2e-3..1e-2 away from its
    # own fp64 evaluation on these problems (fixture 2x1x64x64: 5e-3 at dec_layer2.res.res1, where the HIP
    # path is within 1.3e-4 of fp64).  The HIP path is therefore judged against the fp64 oracle with the
    # rule in _grad_verdict below.
    _, g32, _, _ = O.train_step_grads(O.uresnet_forward, sd, xt, lt, wt)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, g64, _, _ = O.train_step_grads(O.uresnet_forward, sd64, xt.double(), lt, wt.double())
    params = dict(m.named_parameters())
    rows, fails = [], []
    for n in g64:
        gv = params[n].grad.detach().cpu().double()
        if n in ("conv1.bias", "conv10.bias"):      # followed by BatchNorm: analytically zero gradient, pure rounding noise
            assert gv.abs().max().item() <= 1e-5
            continue
        rows.append(_grad_row(n, gv, g32[n].double(), g64[n]))
        fails += _grad_verdict(rows[-1])
    out_dir = os.environ.get("UBR_TEST_OUT", "")
    if out_dir:
        with open(os.path.join(out_dir, "grad_errors_%s.txt" % tag), "w") as f:
            f.write("train logp rel err %.3e\n" % e)
            for n, emax, fmax, el2, fl2, cos in rows:
                f.write("%-40s max %.3e (floor %.3e)  l2 %.3e (floor %.3e)  cos %.6f\n" % (n, emax, fmax, el2, fl2, cos))
    print("worst grad (max-abs rel, l2 rel):", max(r[1] for r in rows), max(r[3] for r in rows))
    assert not fails, "; ".join(fails[:8])
    # the reference fixture's own per-tensor norms (an fp32 run of the reference code)
    for n, ref_norm in zip([str(v) for v in g["grad_names"]], g["grad_norms"]):
        if n in ("conv1.bias", "conv10.bias"):
            continue
        norm = float(params[n].grad.double().norm())
        assert abs(norm - ref_norm) <= 5e-2 * ref_norm + 1e-6, "grad norm %s: %g vs reference %g" % (n, norm, ref_norm)


def _grad_row(n, gv, g32, g64):
    scale = max(g64.abs().max().item(), 1e-12)
    nrm = max(g64.norm().item(), 1e-12)
    emax = (gv - g64).abs().max().item() / scale
    fmax = (g32 - g64).abs().max().item() / scale
    el2 = (gv - g64).norm().item() / nrm
    fl2 = (g32 - g64).norm().item() / nrm
    cos = float(torch.nn.functional.cosine_similarity(gv.reshape(1, -1), g64.reshape(1, -1)))
    return (n, emax, fmax, el2, fl2, cos)


def _grad_verdict(row, cos_min=0.9999, l2_max=2e-2):
    """A gradient passes if it is as close to the fp64 truth as PyTorch-CPU fp32 is (10x its floor + 5e-4),
    or -- when a ReLU mask flipped somewhere (a legitimate fp32 outcome, see the docstring above) -- if it
    still points the same way: cosine >= 0.9999 and relative L2 error <= 2e-2 (the measured envelope is
    L2 <= 1e-2, cosine >= 0.99996).  Wiring or indexing bugs give O(1) errors and fail both."""
    n, emax, fmax, el2, fl2, cos = row
    if emax <= 10 * fmax + 5e-4:
        return []
    if cos >= cos_min and el2 <= l2_max:
        return []
    return ["grad %s: max-rel %.3e (fp32 floor %.3e), l2-rel %.3e, cos %.5f" % (n, emax, fmax, el2, cos)]


def test_full_size_config1_matches_reference_summary(golden_dir):
    """BASELINE.json configs[0] at its real size (UResNet ip16, B=2, 1x512x512, fp32): the fixture holds what the
    reference's own code produced there -- 4096 sampled log-probabilities (eval and train mode), class counts and an
    argmax hash, the loss, and for each of the 165 gradient tensors its L2 norm and 16 sampled entries."""
    import hashlib
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_2x1x512x512_summary.npz"))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.uresnet_schema(3, C, 16, 16), wseed)
    x, lab, wgt = synthetic.make_batch(B, H, W, seed0)
    m = _model(sd)
    m.eval()
    with torch.no_grad():
        ev = m(torch.from_numpy(x).cuda()).cpu()
    idx = g["sample_idx"]
    ref = g["sample_logp_eval"]
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(ev.numpy().reshape(-1)[idx] - ref).max())
    assert err <= 1e-4 * scale, "eval log-prob samples: %.3e (scale %.3e)" % (err, scale)   # measured 1.3e-6; north_star bar 1e-3
    am = ev.argmax(1).numpy().astype(np.uint8)
    nlow = len(g["low_margin_idx"])
    if nlow == 0:
        assert hashlib.sha256(am.tobytes()).hexdigest() == str(g["argmax_sha256_eval"]), "class map differs from the reference's"
    assert np.abs(np.bincount(am.reshape(-1), minlength=3) - g["class_counts_eval"]).sum() <= 2 * nlow
    # train step
    m.train()
    out = m.forward(torch.from_numpy(x).cuda())
    loss = PixelWiseNLLLoss().forward(out, torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda())
    loss.backward()
    torch.cuda.synchronize()
    ref_t = g["sample_logp_train"]
    err_t = float(np.abs(out.detach().cpu().numpy().reshape(-1)[idx] - ref_t).max())
    assert err_t <= 1e-4 * max(1.0, float(np.abs(ref_t).max())), "train log-prob samples: %.3e" % err_t   # measured 2e-6
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    # gradients: norms of all 165 tensors and the sampled entries.  The reference numbers are its fp32 CPU run, which
    # is itself a few 1e-3 away from exact arithmetic on the ReLU-mask-sensitive tensors (see _grad_verdict), so the
    # bar is 5e-3 on each norm and 2e-2 on the samples relative to the tensor's largest sampled entry.
    params = dict(m.named_parameters())
    bad = []
    worst_norm = 0.0
    for name, nref in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        gv = params[name].grad.detach().cpu().double().reshape(-1)
        if name in ("conv1.bias", "conv10.bias"):        # analytically zero (a BatchNorm follows)
            assert gv.abs().max().item() <= 1e-4
            continue
        nrm = float(torch.sqrt((gv * gv).sum()))
        worst_norm = max(worst_norm, abs(nrm - float(nref)) / float(nref))
        if abs(nrm - float(nref)) > 5e-3 * float(nref):          # measured worst 2.5e-4
            bad.append("%s norm %.6e vs %.6e" % (name, nrm, float(nref)))
        rs = np.random.RandomState(7)
        sidx = np.sort(rs.choice(gv.numel(), size=min(16, gv.numel()), replace=False))
        sref = g["gs__" + name].astype(np.float64)
        sgot = gv.numpy()[sidx]
        tol_abs = 2e-2 * max(float(np.abs(sref).max()), float(nref) / np.sqrt(gv.numel()))
        if np.abs(sgot - sref).max() > tol_abs:
            bad.append("%s samples max err %.3e (tol %.3e)" % (name, np.abs(sgot - sref).max(), tol_abs))
    print("full-size config 1: eval sample err %.2e, train sample err %.2e, loss rel %.2e, worst grad-norm rel %.2e"
          % (err / scale, err_t / max(1.0, float(np.abs(ref_t).max())), abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), worst_norm))
    assert not bad, "; ".join(bad[:6])


def test_gradients_dense_input_tight():
    """Same check on a dense noise image with the reference's init (gamma=1, beta=0).  Even here the
    fp32 CPU oracle is ~1e-2 from its fp64 evaluation on some tensors (ReLU mask flips through 52
    BatchNorm layers), so each tensor is judged against 10x its own fp32-oracle floor + 2e-4."""
    torch.manual_seed(5)
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 7)
    for k in sd:
        if k.endswith("weight") and sd[k].dim() == 1:
            sd[k] = torch.ones_like(sd[k])
        elif k.endswith("bias") and (".bn" in k or k.startswith("bn")):
            sd[k] = torch.zeros_like(sd[k])
    B, H, W = 2, 64, 96
    xt = torch.randn(B, 1, H, W) * 30
    lt = torch.randint(0, 3, (B, H, W))
    wt = torch.rand(B, H, W) + 0.5
    m = _model(sd)
    m.train()
    loss = PixelWiseNLLLoss()(m(xt.cuda()), lt.cuda(), wt.cuda())
    loss.backward()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    l64, g64, _, _ = O.train_step_grads(O.uresnet_forward, sd64, xt.double(), lt, wt.double())
    _, g32, _, _ = O.train_step_grads(O.uresnet_forward, sd, xt, lt, wt)
    assert abs(loss.item() - float(l64)) <= 1e-5 * abs(float(l64))
    params = dict(m.named_parameters())
    rows, fails = [], []
    for n in g64:
        gv = params[n].grad.cpu().double()
        if n in ("conv1.bias", "conv10.bias"):
            assert gv.abs().max().item() <= 1e-5
            continue
        rows.append(_grad_row(n, gv, g32[n].double(), g64[n]))
        fails += _grad_verdict(rows[-1])
    print("dense: worst grad (max-abs rel, l2 rel, min cos):", max(r[1] for r in rows), max(r[3] for r in rows), min(r[5] for r in rows),
          " fp32 oracle floors:", max(r[2] for r in rows), max(r[4] for r in rows))
    assert not fails, "; ".join(fails[:8])


def test_training_trajectory_matches_oracle():
    """Five Adam steps from the same weights on the same batch: the loss curves of the HIP path and
    of the CPU oracle agree to 5e-3 relative (whole train step incl. BN running stats and optimizer)."""
    from collections import OrderedDict
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
    x, lab, wgt = synthetic.make_batch(2, 64, 64, 1000)
    xt, lt, wt = torch.from_numpy(x), torch.from_numpy(lab), torch.from_numpy(wgt)
    m = _model(sd)
    m.train()
    crit = PixelWiseNLLLoss()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    xd, ld, wd = xt.cuda(), lt.cuda(), wt.cuda()
    hip = []
    for _ in range(5):
        loss = crit(m(xd), ld, wd)
        opt.zero_grad()
        loss.backward()
        opt.step()
        hip.append(loss.item())
    p = OrderedDict((k, (v.clone().requires_grad_(True) if O.is_param_key(k) else v.clone())) for k, v in sd.items())
    oopt = torch.optim.Adam([v for k, v in p.items() if O.is_param_key(k)], lr=1e-3, weight_decay=1e-4)
    ref = []
    for _ in range(5):
        ns = {}
        loss = O.pixelwise_nll(O.uresnet_forward(p, xt, True, ns), lt, wt)
        oopt.zero_grad()
        loss.backward()
        oopt.step()
        p.update(ns)
        ref.append(loss.item())
    print("loss trajectories hip", hip, "oracle", ref)
    assert ref[-1] < ref[0]
    for a, b in zip(hip, ref):
        assert abs(a - b) <= 5e-3 * abs(b), "loss trajectory diverges: %s vs %s" % (hip, ref)
    after = m.state_dict()
    for k in ("bn1.running_mean", "enc_layer3.res1.bn2.running_var", "bn10.running_var"):
        assert _rel(after[k].cpu(), p[k].detach()) <= 5e-2, k   # trajectories of two fp32 implementations drift apart step by step


def test_four_classes_and_metrics(golden_dir):
    g = np.load(os.path.join(golden_dir, "uresnet_ip16_nc4_1x1x64x96.npz"))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.uresnet_schema(4, C, 16, 16), wseed)
    x, lab, _ = synthetic.make_batch(B, H, W, seed0)
    m = _model(sd, num_classes=4)
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(x).cuda())
    ref = torch.from_numpy(g["logp_eval"])
    assert _rel(out.cpu(), ref) <= 1e-3
    lt = torch.from_numpy(lab)
    assert torch.equal(metrics.confusion_matrix(out, lt.cuda()).cpu(), O.confusion_matrix(out.cpu(), lt))


def test_bf16_tracks_fp32():
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
    x, lab, wgt = synthetic.make_batch(2, 128, 128, 1000)
    xt = torch.from_numpy(x).cuda()
    m = _model(sd)
    m.eval()
    with torch.no_grad():
        ref = m(xt)
        m.compute_dtype = torch.bfloat16
        out = m(xt)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            m.compute_dtype = None
            out2 = m(xt)
    assert torch.equal(out, out2), "autocast and compute_dtype select the same path"
    cm = torch.zeros(3, 3, dtype=torch.int64)
    a, b = ref.argmax(1).cpu().reshape(-1), out.argmax(1).cpu().reshape(-1)
    cm = torch.bincount(a * 3 + b, minlength=9).reshape(3, 3)
    iou = O.iou_from_confusion(cm)
    agree = float((a == b).float().mean())
    print("bf16 vs fp32: max abs logp diff %.3e, pixel agreement %.5f, IoU %s" % ((out - ref).abs().max().item(), agree, iou))
    assert agree >= 0.98
    # bf16 train step runs and produces finite gradients of the right scale
    m.train()
    m.compute_dtype = torch.bfloat16
    crit = PixelWiseNLLLoss()
    loss = crit(m(xt), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda())
    loss.backward()
    m2 = _model(sd)
    m2.train()
    loss2 = crit(m2(xt), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda())
    loss2.backward()
    assert abs(loss.item() - loss2.item()) <= 3e-2 * abs(loss2.item())
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.isfinite(p.grad).all(), n
        cos = torch.nn.functional.cosine_similarity(p.grad.reshape(1, -1).double(), q.grad.reshape(1, -1).double()).item()
        assert cos >= 0.7 or q.grad.abs().max() < 1e-6, "bf16 gradient of %s diverges from fp32 (cos %.3f)" % (n, cos)


def test_bench_config_full_size_properties():
    """BASELINE.json configs[1] at its real size (ub_uresnet 3-class bf16, batch 16, 512x512), where the oracle is too
    slow: size-independent properties instead.  (1) eval mode is per-image: image i of the batch of 16 gives bitwise
    the log-probabilities it gives alone.  (2) per-class IoU of the bf16 class map against the fp32 path (itself pinned
    to the reference at this size by test_full_size_config1_matches_reference_summary).  (3) the loss is linear in the
    pixel weights.  (4) a bf16 train step is deterministic: two runs give bitwise identical gradients (fixed-order
    slab sums, no float atomics) -- also across the second weight-gradient stream."""
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
    x, lab, wgt = synthetic.make_batch(16, 512, 512, 1000)
    xt, lt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda()
    m = _model(sd)
    m.eval()
    with torch.no_grad():
        m.compute_dtype = torch.bfloat16
        out = m(xt)
        one = m(xt[5:6])
    assert torch.equal(out[5:6], one), "eval-mode output of an image depends on its batch neighbours"
    # (2) class maps, bf16 vs fp32, in TRAIN mode (batch statistics: the benchmark's mode; with these untrained seeded
    # weights the eval-mode logits reach 1e5 and say nothing about precision).  Untrained weights leave many near-ties,
    # so IoU is asserted where the fp32 top-2 margin exceeds 0.2 nat (75 % of pixels) and reported for all pixels.
    m.train()
    with torch.no_grad():
        m.compute_dtype = None
        ref_t = m(xt)
        m.compute_dtype = torch.bfloat16
        out_t = m(xt)
    a, b = ref_t.argmax(1).reshape(-1), out_t.argmax(1).reshape(-1)
    iou = O.iou_from_confusion(torch.bincount(a * 3 + b, minlength=9).reshape(3, 3).cpu())
    top2 = torch.topk(ref_t, 2, dim=1)[0]
    safe = ((top2[:, 0] - top2[:, 1]) > 0.2).reshape(-1)
    iou_safe = O.iou_from_confusion(torch.bincount(a[safe] * 3 + b[safe], minlength=9).reshape(3, 3).cpu())
    dmean = float((out_t - ref_t).abs().mean())
    print("bf16 vs fp32 at 16x512x512 (train-mode forward): mean |dlogp| %.4f; per-class IoU %s; margin>0.2 (%.1f %% of pixels): IoU %s"
          % (dmean, [round(float(v), 4) for v in iou], 100.0 * float(safe.float().mean()), [round(float(v), 5) for v in iou_safe]))
    assert dmean <= 0.05 and min(float(v) for v in iou) >= 0.95
    assert min(float(v) for v in iou_safe) >= 0.999
    m.eval()
    crit = PixelWiseNLLLoss()
    w2 = torch.rand_like(wt)
    l1, l2, l12 = crit(out, lt, wt).item(), crit(out, lt, w2).item(), crit(out, lt, wt + w2).item()
    assert abs(l12 - (l1 + l2)) <= 1e-5 * abs(l12)
    grads = []
    for _ in range(2):
        mm = _model(sd)
        mm.train()
        mm.compute_dtype = torch.bfloat16
        crit(mm(xt), lt, wt).backward()
        torch.cuda.synchronize()
        grads.append([p.grad.clone() for p in mm.parameters()])
    for (n, _), g1, g2 in zip(mm.named_parameters(), grads[0], grads[1]):
        assert torch.isfinite(g1).all(), n
        assert torch.equal(g1, g2), "gradient of %s is not reproducible run to run" % n


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_two_stream_backward_equals_single_stream(dt, monkeypatch):
    """The weight gradients run on a second HIP stream.  A FIRST backward of a fresh model (the slab workspace grows
    and is replaced while earlier side-stream launches are still in flight) must give bitwise the gradients of the
    single-stream schedule -- regression test for a recycled-workspace race that the full-size fixture test caught."""
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
    x, lab, wgt = synthetic.make_batch(2, 512, 512, 1000)
    xt, lt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda()
    crit = PixelWiseNLLLoss()
    res = {}
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("UBR_WGRAD_STREAM", mode)
        mm = _model(sd)
        mm.train()
        mm.compute_dtype = dt
        crit(mm(xt), lt, wt).backward()
        torch.cuda.synchronize()
        g = [p.grad.clone() for p in mm.parameters()]
        if mode in res:
            for (n, _), a, b in zip(mm.named_parameters(), res[mode], g):
                assert torch.equal(a, b), "two-stream gradient of %s differs run to run" % n
        res[mode] = g
    for (n, _), a, b in zip(mm.named_parameters(), res["0"], res["1"]):
        assert torch.equal(a, b), "gradient of %s: two-stream schedule differs from the single-stream one" % n


def test_errors_are_exceptions():
    m = UResNet(num_classes=3, input_channels=1, inplanes=16).cuda()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 48, 64, device="cuda"))       # not a multiple of 32
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64, device="cuda"))       # wrong channel count
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 64, 64))                      # CPU tensor: no fallback


def test_weights_updated_by_fused_optimizer_are_used():
    """torch's fused Adam updates parameters WITHOUT bumping their version counters; the executor must not
    serve stale packed weight images (regression test: it repacks every pass)."""
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
    x, lab, wgt = synthetic.make_batch(1, 64, 64, 1000)
    xd, ld, wd = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda()
    m = _model(sd)
    m.train()
    try:
        opt = torch.optim.Adam(m.parameters(), lr=1e-2, fused=True)
    except Exception:
        pytest.skip("fused Adam unavailable")
    crit = PixelWiseNLLLoss()
    loss = crit(m(xd), ld, wd)
    opt.zero_grad()
    loss.backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        after = m(xd)
    fresh = _model({k: v.detach().cpu() for k, v in m.state_dict().items()})
    fresh.eval()
    with torch.no_grad():
        want = fresh(xd)
    assert torch.equal(after, want), "forward after a fused optimizer step used stale weights"
    assert (after - want).abs().max().item() == 0.0


def test_inplanes32_variant(golden_dir):
    """UResNet(inplanes=32) -- the value training/train_ubresnet2018_wlarcv2.py:88 passes -- against the reference fixture
    (tests/golden/uresnet_ip32_1x1x64x64.npz: the reference's own eval output, train output, loss and gradient norms) and,
    tensor by tensor, against the fp64 oracle through the gradient rule of this file."""
    g = np.load(os.path.join(golden_dir, "uresnet_ip32_1x1x64x64.npz"))
    B, C, H, W, seed0, wseed = [int(v) for v in g["meta"]]
    sd = O.seeded_state_dict(O.uresnet_schema(3, C, 32, 16), wseed)
    x, lab, wgt = synthetic.make_batch(B, H, W, seed0)
    xt, lt, wt = torch.from_numpy(x), torch.from_numpy(lab), torch.from_numpy(wgt)
    m = UResNet(num_classes=3, input_channels=1, inplanes=32)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        ev = m(xt.cuda()).cpu()
    assert _rel(ev, torch.from_numpy(g["logp_eval"])) <= 1e-3
    m.train()
    out = m(xt.cuda())
    loss = PixelWiseNLLLoss()(out, lt.cuda(), wt.cuda())
    loss.backward()
    ref = torch.from_numpy(g["logp_train"])
    worst = ((out.detach().cpu() - ref).abs() - 1e-3 * ref.abs()).max().item()
    assert worst <= 1e-4, "ip32 train log-probabilities: worst |a-b| - 1e-3|b| = %.3e" % worst
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    assert sum(p.numel() for p in m.parameters()) == 72340003
    _, g32, _, _ = O.train_step_grads(O.uresnet_forward, sd, xt, lt, wt)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, g64, _, _ = O.train_step_grads(O.uresnet_forward, sd64, xt.double(), lt, wt.double())
    params = dict(m.named_parameters())
    rows, fails = [], []
    for n in g64:
        gv = params[n].grad.detach().cpu().double()
        if n in ("conv1.bias", "conv10.bias"):
            assert gv.abs().max().item() <= 1e-4, n
            continue
        rows.append(_grad_row(n, gv, g32[n].double(), g64[n]))
        fails += _grad_verdict(rows[-1])
    print("ip32 worst grad (max-abs rel, l2 rel, min cos):", max(r[1] for r in rows), max(r[3] for r in rows), min(r[5] for r in rows))
    assert not fails, "; ".join(fails[:8])
    for n, ref_norm in zip([str(v) for v in g["grad_names"]], g["grad_norms"]):
        if n in ("conv1.bias", "conv10.bias"):
            continue
        norm = float(params[n].grad.double().norm())
        assert abs(norm - ref_norm) <= 5e-2 * ref_norm + 1e-6, "grad norm %s: %g vs reference %g" % (n, norm, ref_norm)


def test_device_stager_feeds_training():
    from ubresnet_amd.synthetic import SyntheticLArCVDataset, DeviceStager
    ld = SyntheticLArCVDataset(height=64, width=64, tag="train", nentries=8)
    ld.start(2)
    st = DeviceStager(ld, 2, 64, 64, tag="train")
    a1, l1, w1 = st.next()
    a2, l2, w2 = st.next()
    want, wl, ww = synthetic.make_batch(4, 64, 64, 1000)
    assert a1.is_cuda and l1.dtype == torch.int64 and tuple(a1.shape) == (2, 1, 64, 64)
    assert torch.equal(a1.cpu(), torch.from_numpy(want[:2])) and torch.equal(a2.cpu(), torch.from_numpy(want[2:4]))
    assert torch.equal(l2.cpu(), torch.from_numpy(wl[2:4])) and torch.equal(w1.cpu(), torch.from_numpy(ww[:2]))
