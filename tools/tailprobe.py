#!/usr/bin/env python3
"""What the tail of a train step costs: the bench step timed with and without opt.step() (the Adam kernel alone moves ~0.5 GB,
~0.1 ms), and with a device synchronisation between backward and optimizer (exposes the host-side enqueue latency).
usage: python tools/tailprobe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubresnet_amd import synthetic
from ubresnet_amd.models.ub_uresnet import UResNet
from ubresnet_amd.optim import FlatAdam
from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
torch.manual_seed(1234)
model = UResNet(num_classes=3, input_channels=1, inplanes=16).to(dev)
model.compute_dtype = torch.bfloat16
model.train()
crit = PixelWiseNLLLoss()
opt = FlatAdam(model, lr=1e-5, weight_decay=1e-4)
x, lab, wgt = synthetic.make_batch(16, 512, 512, seed0=1000, planes=1)
x, lab, wgt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(wgt).to(dev)


def step(mode):
    out = model.forward(x)
    loss = crit.forward(out, lab, wgt)
    opt.zero_grad()
    loss.backward()
    if mode == "sync":
        torch.cuda.synchronize()
    if mode != "noopt":
        opt.step()
    return loss


def timed(mode):
    for _ in range(5):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step(mode)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


for mode in ("full", "noopt", "full", "noopt", "sync"):
    print("%-6s %.3f ms/step" % (mode, timed(mode)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    opt.step()
e1.record()
torch.cuda.synchronize()
print("opt.step() alone, back to back: %.3f ms" % (e0.elapsed_time(e1) / 20))
