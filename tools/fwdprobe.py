#!/usr/bin/env python3
"""Per-launch breakdown of the TRAIN-MODE FORWARD pass alone (no second stream beside it): which layers are far from their own
roofline when nothing contends with them.  usage: python tools/fwdprobe.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubresnet_amd import ops, plan, synthetic
from ubresnet_amd.models.ub_uresnet import UResNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(1)
m = UResNet(num_classes=3, input_channels=1, inplanes=16).cuda().train()
m.compute_dtype = torch.bfloat16
x = torch.from_numpy(synthetic.make_batch(B, 512, 512, 1000)[0]).cuda()
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m(x)
    e1.record()
    torch.cuda.synchronize()
    print("train-mode forward (no_grad, replayed tape): %.3f ms" % (e0.elapsed_time(e1) / 10))
    prof = ops.LaunchProfiler()
    ops._prof = prof
    plan.TIMED = prof.timed
    m(x)
    ops._prof = None
    plan.TIMED = None
    torch.cuda.synchronize()
rows = sorted(prof.summary(by="shape").items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print("sum of launch durations %.3f ms" % (1e3 * tot))
for (nm, sg), (c, t, b, fl) in rows[:40]:
    hbm_us = b / 6.0e12 * 1e6 / max(c, 1)
    mfma_us = fl / 2.5e15 * 1e6 / max(c, 1)
    print("%-14s %-58s n=%2d %7.1f us each | bound %5.1f us (hbm@6TB/s %5.1f, mfma %5.1f) x%.1f" % (nm, sg, c, 1e6 * t / c, max(hbm_us, mfma_us), hbm_us, mfma_us, (1e6 * t / c) / max(hbm_us, mfma_us, 1e-3)))
