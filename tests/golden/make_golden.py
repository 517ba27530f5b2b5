#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE's own model code.

Runs only in the build container (needs /root/reference).  Recipe = SURVEY.md section 8c:
the reference's .py text is read, lib2to3 ``fix_print`` is applied IN MEMORY (the files
are Python 2), and the result is exec'd into fresh modules; torchvision/ROOT/larcv/commands
are pre-seeded as empty stubs (the model code never uses them).  Nothing from the reference
is written anywhere; only inputs/outputs (data) are stored, as small .npz files next to
this script.  Version shims (torch 0.4 -> 2.x): ``crit.size_average = True``;
for the backward fixtures each BasicBlock.relu2 is replaced by nn.Threshold(0,0) so the
reference's in-place residual add (models/common_layers.py:52,54) is differentiable.

Usage:  python tests/golden/make_golden.py            # every fixture
        python tests/golden/make_golden.py --only norm   # only the normalised-eval (deployment) fixtures
        python tests/golden/make_golden.py --only aspp   # only the round-3 ASPP_ResNet / inplanes=32 fixtures
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import uresnet_oracle as O          # noqa: E402  (seeded params + schema only)
from ubresnet_amd import synthetic              # noqa: E402  (seeded inputs)


def _load_py2(path, name):
    from lib2to3 import refactor
    src = open(path).read()
    rt = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
    src3 = str(rt.refactor_string(src + "\n", path))
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(src3, path, "exec"), mod.__dict__)
    return mod


def import_reference():
    sys.path.insert(0, REF + "/models")
    for n in ["torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models",
              "commands", "ROOT"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    larcv = types.ModuleType("larcv")
    larcv.larcv = larcv
    sys.modules["larcv"] = larcv
    ub = _load_py2(REF + "/models/ub_uresnet.py", "ub_uresnet")
    aspp = _load_py2(REF + "/models/ASPP_ResNet.py", "ASPP_ResNet")
    pl = _load_py2(REF + "/training/pixelwise_nllloss.py", "pixelwise_nllloss")
    import common_layers
    return ub, aspp, pl, common_layers


def make_differentiable(model):
    for m in model.modules():
        if hasattr(m, "relu2") and hasattr(m, "bn2"):
            m.relu2 = nn.Threshold(0.0, 0.0)
    return model


def sample_indices(n, k, seed):
    rs = np.random.RandomState(seed)
    return np.sort(rs.choice(n, size=min(k, n), replace=False)).astype(np.int64)


def grad_summary(model, prefix=""):
    names, norms, samples = [], [], {}
    for k, p in model.named_parameters():
        g = p.grad.detach().reshape(-1).numpy()
        names.append(k)
        norms.append(np.float64(np.sqrt((g.astype(np.float64) ** 2).sum())))
        idx = sample_indices(g.shape[0], 16, 7)
        samples[k] = g[idx].copy()
    return names, np.array(norms), samples


def run_train_step(model, crit, x, lab, wgt):
    model.train()
    model.zero_grad()
    out = model.forward(torch.from_numpy(x))
    loss = crit.forward(out, torch.from_numpy(lab), torch.from_numpy(wgt))
    loss.backward()
    return out.detach().numpy(), float(loss.item())


def calibrate_running_stats(model, batches):
    """Deployment-like BatchNorm statistics for seeded (untrained) weights: the REFERENCE model itself is run in train
    mode over calibration batches with cumulative averaging (momentum=None, statistics reset first), so its running
    statistics become the mean of the batch statistics it saw and eval-mode activations are O(1) -- what a trained
    checkpoint looks like to the inference path (and what keeps fp16 inside its range)."""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.reset_running_stats()
            m.momentum = None
    model.train()
    with torch.no_grad():
        for xb in batches:
            model.forward(torch.from_numpy(xb))
    model.eval()
    return model


def norm_fixtures(ub):
    """eval-mode fixtures with normalising running statistics (BASELINE configs[4] parity: fp32 and fp16 forward of
    deploy/run_ubresnet_wholeview.py-shaped tiles, UResNet(ip16, 1 plane, 4 classes) of deploy/ubresnet_funcs.py:43)"""
    sd4 = O.seeded_state_dict(O.uresnet_schema(4, 1, 16, 16), 43)
    bn_keys = [k for k in sd4 if k.endswith("running_mean") or k.endswith("running_var")]
    for tag, (H, W, seed_x, seed_cal) in {"1x1x64x96": (64, 96, 1200, 1400), "1x1x512x832": (512, 832, 1500, 1600)}.items():
        m = ub.UResNet(num_classes=4, input_channels=1, inplanes=16)
        m.load_state_dict(sd4)
        cal = [synthetic.make_batch(2, H, W, seed_cal + 10 * i)[0] for i in range(2)]
        calibrate_running_stats(m, cal)
        after = m.state_dict()
        x = synthetic.make_batch(1, H, W, seed_x)[0]
        with torch.no_grad():
            out = m.forward(torch.from_numpy(x))
        stats = np.concatenate([after[k].numpy().reshape(-1) for k in bn_keys]).astype(np.float32)
        am = out.max(1)[1].numpy().astype(np.uint8)
        top2 = torch.topk(out, 2, dim=1)[0]
        margin = (top2[:, 0] - top2[:, 1]).numpy().reshape(-1)
        common = dict(bn_keys=np.array(bn_keys), bn_stats=stats, meta=np.array([1, 1, H, W, seed_x, 43]),
                      absmax=np.float32(out.abs().max().item()))
        if H * W <= 64 * 96:
            np.savez_compressed(os.path.join(HERE, "uresnet_ip16_nc4_norm_%s.npz" % tag), logp_eval=out.numpy().astype(np.float32), **common)
        else:
            idx = sample_indices(out.numel(), 4096, 13)
            np.savez_compressed(
                os.path.join(HERE, "uresnet_ip16_nc4_norm_%s_summary.npz" % tag),
                sample_idx=idx, sample_logp_eval=out.numpy().reshape(-1)[idx],
                argmax=am, argmax_sha256=np.array(hashlib.sha256(am.tobytes()).hexdigest()),
                class_counts=np.bincount(am.reshape(-1), minlength=4),
                safe_0p02=np.packbits(margin > 0.02), safe_0p2=np.packbits(margin > 0.2),
                margin_hist=np.histogram(margin, bins=[0, 1e-4, 1e-3, 1e-2, 1e-1, 1, 10, 1e9])[0], **common)
        print("norm fixture", tag, "absmax logp %.3f" % out.abs().max().item(), "class counts", np.bincount(am.reshape(-1), minlength=4))


def aspp_ip32_fixtures(ub, aspp, crit):
    """Round 3: (a) BASELINE configs[3] at its real size -- the reference ASPP_ResNet (models/ASPP_ResNet.py:291,416-523) on a
    1 x 3 x 512 x 832 crop, eval forward and one train step, stored as a summary (sampled log-probabilities, class maps' hashes
    and counts, loss, every gradient tensor's norm and 16 samples); (b) a normalised-eval ASPP fixture (running statistics
    calibrated by the reference in train mode, then its eval output on a 1 x 3 x 64 x 96 crop, full tensor); (c) the
    inplanes=32 U-ResNet of training/train_ubresnet2018_wlarcv2.py:88 -- one train step at 1 x 1 x 64 x 64."""
    sda = O.seeded_state_dict(O.aspp_resnet_schema(3, 3, 16), 44)
    # (a)
    x, lab, wgt = synthetic.make_batch(1, 512, 832, 1700, planes=3)
    m = aspp.ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False)
    m.load_state_dict(sda)
    m.eval()
    with torch.no_grad():
        out = m.forward(torch.from_numpy(x))
    am = out.max(1)[1].numpy().astype(np.uint8)
    top2 = torch.topk(out, 2, dim=1)[0]
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    idx = sample_indices(out.numel(), 4096, 17)
    m = make_differentiable(aspp.ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False))
    m.load_state_dict(sda)
    out_tr, loss = run_train_step(m, crit, x, lab, wgt)
    names, norms, samples = grad_summary(m)
    am_tr = out_tr.argmax(1).astype(np.uint8)
    t2 = np.sort(out_tr, axis=1)
    margin_tr = t2[:, -1] - t2[:, -2]
    np.savez_compressed(
        os.path.join(HERE, "aspp_ip16_1x3x512x832_summary.npz"),
        argmax_sha256_eval=np.array(hashlib.sha256(am.tobytes()).hexdigest()), class_counts_eval=np.bincount(am.reshape(-1), minlength=3),
        low_margin_idx_eval=np.nonzero(margin.reshape(-1) < 1e-3 * np.maximum(1.0, np.abs(out.numpy()).max(1).reshape(-1)))[0].astype(np.int64),
        sample_idx=idx, sample_logp_eval=out.numpy().reshape(-1)[idx], sample_logp_train=out_tr.reshape(-1)[idx],
        absmax_eval=np.float32(out.abs().max().item()), absmax_train=np.float32(np.abs(out_tr).max()),
        argmax_sha256_train=np.array(hashlib.sha256(am_tr.tobytes()).hexdigest()), class_counts_train=np.bincount(am_tr.reshape(-1), minlength=3),
        argmax_train=np.packbits(am_tr.reshape(-1) == 0), argmax_train1=np.packbits(am_tr.reshape(-1) == 1),
        low_margin_idx_train=np.nonzero(margin_tr.reshape(-1) < 1e-3)[0].astype(np.int64),
        loss=np.float64(loss), grad_names=np.array(names), grad_norms=norms, **{"gs__" + k: v for k, v in samples.items()},
        meta=np.array([1, 3, 512, 832, 1700, 44]))
    print("aspp 512x832 summary: loss", loss, "eval counts", np.bincount(am.reshape(-1), minlength=3), "absmax eval %.3g train %.3g" % (out.abs().max().item(), np.abs(out_tr).max()))
    # (b)
    bn_keys = [k for k in sda if k.endswith("running_mean") or k.endswith("running_var")]
    m = aspp.ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False)
    m.load_state_dict(sda)
    cal = [synthetic.make_batch(2, 64, 96, 1800 + 10 * i, planes=3)[0] for i in range(2)]
    calibrate_running_stats(m, cal)
    after = m.state_dict()
    x = synthetic.make_batch(1, 64, 96, 1300, planes=3)[0]
    with torch.no_grad():
        out = m.forward(torch.from_numpy(x))
    stats = np.concatenate([after[k].numpy().reshape(-1) for k in bn_keys]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "aspp_ip16_norm_1x3x64x96.npz"), logp_eval=out.numpy().astype(np.float32), bn_keys=np.array(bn_keys),
                        bn_stats=stats, absmax=np.float32(out.abs().max().item()), meta=np.array([1, 3, 64, 96, 1300, 44]))
    print("aspp norm fixture: absmax logp %.3f" % out.abs().max().item())
    # (c)
    sd32 = O.seeded_state_dict(O.uresnet_schema(3, 1, 32, 16), 45)
    x, lab, wgt = synthetic.make_batch(1, 64, 64, 1900)
    m = ub.UResNet(num_classes=3, input_channels=1, inplanes=32)
    m.load_state_dict(sd32)
    m.eval()
    with torch.no_grad():
        out_eval = m.forward(torch.from_numpy(x)).numpy()
    m = make_differentiable(ub.UResNet(num_classes=3, input_channels=1, inplanes=32))
    m.load_state_dict(sd32)
    out_train, loss = run_train_step(m, crit, x, lab, wgt)
    names, norms, samples = grad_summary(m)
    np.savez_compressed(os.path.join(HERE, "uresnet_ip32_1x1x64x64.npz"), logp_eval=out_eval.astype(np.float32), logp_train=out_train.astype(np.float32),
                        loss=np.float64(loss), grad_names=np.array(names), grad_norms=norms, **{"gs__" + k: v for k, v in samples.items()},
                        meta=np.array([1, 1, 64, 64, 1900, 45]))
    print("uresnet ip32 1x1x64x64: loss", loss)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ub, aspp, pl, cl = import_reference()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "norm":
        norm_fixtures(ub)
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "aspp":
        crit = pl.PixelWiseNLLLoss()
        crit.size_average = True   # shim 1
        aspp_ip32_fixtures(ub, aspp, crit)
        return

    # ---------------- schema check: our key order/shape == reference state_dict -------------
    for ctor, schema in (
        (lambda: ub.UResNet(num_classes=3, input_channels=1, inplanes=16), O.uresnet_schema(3, 1, 16, 16)),
        (lambda: ub.UResNet(num_classes=4, input_channels=1, inplanes=16), O.uresnet_schema(4, 1, 16, 16)),
        (lambda: ub.UResNet(num_classes=3, input_channels=1, inplanes=32), O.uresnet_schema(3, 1, 32, 16)),
        (lambda: aspp.ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False), O.aspp_resnet_schema(3, 3, 16)),
    ):
        ref_sd = ctor().state_dict()
        assert list(ref_sd.keys()) == list(schema.keys()), "state_dict key order mismatch"
        for k, v in ref_sd.items():
            assert tuple(v.shape) == tuple(schema[k]), (k, v.shape, schema[k])
    keys_u = list(O.uresnet_schema(3, 1, 16, 16).keys())
    keys_a = list(O.aspp_resnet_schema(3, 3, 16).keys())
    with open(os.path.join(HERE, "state_dict_keys_uresnet_ip16.txt"), "w") as f:
        f.write("\n".join("%s %s" % (k, "x".join(map(str, s)) or "scalar") for k, s in O.uresnet_schema(3, 1, 16, 16).items()) + "\n")
    with open(os.path.join(HERE, "state_dict_keys_aspp_ip16.txt"), "w") as f:
        f.write("\n".join("%s %s" % (k, "x".join(map(str, s)) or "scalar") for k, s in O.aspp_resnet_schema(3, 3, 16).items()) + "\n")

    crit = pl.PixelWiseNLLLoss()
    crit.size_average = True   # shim 1

    # ---------------- UResNet ip16, forward eval + train, small shapes ----------------
    sd = O.seeded_state_dict(O.uresnet_schema(3, 1, 16, 16), 42)
    for tag, (B, H, W, seed0) in {"2x1x64x64": (2, 64, 64, 1000), "1x1x96x128": (1, 96, 128, 1100)}.items():
        x, lab, wgt = synthetic.make_batch(B, H, W, seed0)
        m = ub.UResNet(num_classes=3, input_channels=1, inplanes=16)
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            out_eval = m.forward(torch.from_numpy(x)).numpy()
        m = make_differentiable(ub.UResNet(num_classes=3, input_channels=1, inplanes=16))
        m.load_state_dict(sd)
        out_train, loss = run_train_step(m, crit, x, lab, wgt)
        names, norms, samples = grad_summary(m)
        after = m.state_dict()
        np.savez_compressed(
            os.path.join(HERE, "uresnet_ip16_%s.npz" % tag),
            logp_eval=out_eval.astype(np.float32), logp_train=out_train.astype(np.float32),
            loss=np.float64(loss), grad_names=np.array(names), grad_norms=norms,
            **{"gs__" + k: v for k, v in samples.items()},
            bn1_running_mean=after["bn1.running_mean"].numpy(), bn1_running_var=after["bn1.running_var"].numpy(),
            bn10_running_mean=after["bn10.running_mean"].numpy(), bn10_running_var=after["bn10.running_var"].numpy(),
            nbt=after["bn1.num_batches_tracked"].numpy(),
            meta=np.array([B, 1, H, W, seed0, 42]))
        print("uresnet", tag, "loss", loss)

    # ---------------- UResNet num_classes=4 (deploy/ubresnet_funcs.py:43), eval only ---------
    sd4 = O.seeded_state_dict(O.uresnet_schema(4, 1, 16, 16), 43)
    x, lab, wgt = synthetic.make_batch(1, 64, 96, 1200)
    m = ub.UResNet(num_classes=4, input_channels=1, inplanes=16)
    m.load_state_dict(sd4)
    m.eval()
    with torch.no_grad():
        out = m.forward(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "uresnet_ip16_nc4_1x1x64x96.npz"), logp_eval=out, meta=np.array([1, 1, 64, 96, 1200, 43]))

    # ---------------- ASPP_ResNet ip16 at 1x3x64x96 ----------------
    sda = O.seeded_state_dict(O.aspp_resnet_schema(3, 3, 16), 44)
    x, lab, wgt = synthetic.make_batch(1, 64, 96, 1300, planes=3)
    m = aspp.ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False)
    m.load_state_dict(sda)
    m.eval()
    with torch.no_grad():
        out_eval = m.forward(torch.from_numpy(x)).numpy()
    m = make_differentiable(aspp.ASPP_ResNet(num_classes=3, in_channels=3, inplanes=16, showsizes=False))
    m.load_state_dict(sda)
    out_train, loss = run_train_step(m, crit, x, lab, wgt)
    names, norms, samples = grad_summary(m)
    np.savez_compressed(os.path.join(HERE, "aspp_ip16_1x3x64x96.npz"),
                        logp_eval=out_eval, logp_train=out_train, loss=np.float64(loss),
                        grad_names=np.array(names), grad_norms=norms,
                        **{"gs__" + k: v for k, v in samples.items()},
                        meta=np.array([1, 3, 64, 96, 1300, 44]))
    print("aspp loss", loss)

    # ---------------- per-block goldens (common_layers imports natively) ----------------
    rs = np.random.RandomState(5)
    blk = {}
    for name, (cin, cout, stride) in {"id": (8, 8, 1), "proj": (8, 16, 1), "down": (8, 16, 2)}.items():
        b = cl.BasicBlock(cin, cout, stride)
        bsd = O.seeded_state_dict(O._block_keys("b", cin, cout, stride), 50 + stride + cout)
        b.load_state_dict({k[2:]: v for k, v in bsd.items()})
        xin = rs.standard_normal((2, cin, 12, 20)).astype(np.float32)
        b.eval()     # eval first: the train pass below updates the running stats in place
        y_ev = b(torch.from_numpy(xin)).detach().numpy()
        b.train()
        y_tr = b(torch.from_numpy(xin)).detach().numpy()
        blk["block_%s_x" % name] = xin
        blk["block_%s_train" % name] = y_tr
        blk["block_%s_eval" % name] = y_ev
    ctl = cl.ConvTransposeLayer(16, 8, 8)
    csd = O.seeded_state_dict(O.OrderedDict([("d.deconv.weight", (16, 8, 4, 4))] + list(O._double_keys("d.res", 16, 8, 1).items())), 60)
    ctl.load_state_dict({k[2:]: v for k, v in csd.items()})
    xin = rs.standard_normal((2, 16, 6, 10)).astype(np.float32)
    skip = rs.standard_normal((2, 8, 12, 20)).astype(np.float32)
    ctl.eval()
    blk["ctl_x"], blk["ctl_skip"] = xin, skip
    blk["ctl_eval"] = ctl(torch.from_numpy(xin), torch.from_numpy(skip)).detach().numpy()
    ctl.train()
    blk["ctl_train"] = ctl(torch.from_numpy(xin), torch.from_numpy(skip)).detach().numpy()
    # loss: weighted NLL incl. ignore_index
    lp = torch.log_softmax(torch.from_numpy(rs.standard_normal((2, 3, 9, 11)).astype(np.float32)), 1)
    tg = torch.from_numpy(rs.randint(0, 3, (2, 9, 11)).astype(np.int64))
    tg[0, 0, :3] = -100
    pw = torch.from_numpy(rs.uniform(0.5, 10, (2, 9, 11)).astype(np.float32))
    blk["loss_logp"], blk["loss_target"], blk["loss_pw"] = lp.numpy(), tg.numpy(), pw.numpy()
    blk["loss_value"] = np.float64(crit.forward(lp, tg, pw).item())
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **blk)

    # ---------------- 512x512 summary (BASELINE config shapes): hashes + samples ----------
    x, lab, wgt = synthetic.make_batch(2, 512, 512, 1000)
    m = ub.UResNet(num_classes=3, input_channels=1, inplanes=16)
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        out = m.forward(torch.from_numpy(x))
    am = out.max(1)[1].numpy().astype(np.uint8)
    top2 = torch.topk(out, 2, dim=1)[0]
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    idx = sample_indices(out.numel(), 4096, 11)
    m = make_differentiable(ub.UResNet(num_classes=3, input_channels=1, inplanes=16))
    m.load_state_dict(sd)
    out_tr, loss = run_train_step(m, crit, x, lab, wgt)
    names, norms, samples = grad_summary(m)
    am_tr = out_tr.argmax(1).astype(np.uint8)
    np.savez_compressed(
        os.path.join(HERE, "uresnet_ip16_2x1x512x512_summary.npz"),
        argmax_sha256_eval=np.array(hashlib.sha256(am.tobytes()).hexdigest()),
        class_counts_eval=np.bincount(am.reshape(-1), minlength=3),
        margin_hist=np.histogram(margin, bins=[0, 1e-4, 1e-3, 1e-2, 1e-1, 1, 10, 1e9])[0],
        # pixels whose top-2 margin is below 1e-3 (argmax may legitimately flip there)
        low_margin_idx=np.nonzero(margin.reshape(-1) < 1e-3)[0].astype(np.int64),
        sample_idx=idx, sample_logp_eval=out.numpy().reshape(-1)[idx],
        sample_logp_train=out_tr.reshape(-1)[idx],
        argmax_sha256_train=np.array(hashlib.sha256(am_tr.tobytes()).hexdigest()),
        class_counts_train=np.bincount(am_tr.reshape(-1), minlength=3),
        loss=np.float64(loss), grad_names=np.array(names), grad_norms=norms,
        **{"gs__" + k: v for k, v in samples.items()},
        meta=np.array([2, 1, 512, 512, 1000, 42]))
    print("512 summary loss", loss, "counts", np.bincount(am.reshape(-1), minlength=3))
    norm_fixtures(ub)
    aspp_ip32_fixtures(ub, aspp, crit)


if __name__ == "__main__":
    main()
