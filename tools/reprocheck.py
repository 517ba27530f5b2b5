#!/usr/bin/env python3
"""Run-to-run reproducibility of one bf16 train step at the bench size: two fresh models, same weights and batch, gradients
compared bitwise.  Prints the parameters whose gradients differ.  usage: python tools/reprocheck.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubresnet_amd import synthetic
from ubresnet_amd.models.ub_uresnet import UResNet
from ubresnet_amd.training.pixelwise_nllloss import PixelWiseNLLLoss

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.manual_seed(3)
ref = UResNet(num_classes=3, input_channels=1, inplanes=16).cuda()
sd = {k: v.clone() for k, v in ref.state_dict().items()}
x, lab, wgt = synthetic.make_batch(16, 512, 512, seed0=1000, planes=1)
x, lab, wgt = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(wgt).cuda()
crit = PixelWiseNLLLoss()


def grads():
    m = UResNet(num_classes=3, input_channels=1, inplanes=16).cuda()
    m.load_state_dict(sd)
    m.train()
    m.compute_dtype = torch.bfloat16
    out = []
    for _ in range(2):          # first pass records the launch tapes, second replays them
        m.zero_grad()
        crit(m(x), lab, wgt).backward()
        torch.cuda.synchronize()
        out.append({n: p.grad.clone() for n, p in m.named_parameters()})
    return out


base = grads()
bad_total = 0
for r in range(R):
    g = grads()
    for which in (0, 1):
        bad = [n for n in base[0] if not torch.equal(base[which][n], g[which][n])]
        bad_total += len(bad)
        print("repeat %d %s pass: %d of %d gradients differ%s" % (r, "recorded" if which == 0 else "replayed", len(bad), len(base[0]), (": " + ", ".join(bad[:6])) if bad else ""))
same = [n for n in base[0] if not torch.equal(base[0][n], base[1][n])]
print("recorded vs replayed pass of one model (running statistics do not enter the gradients): %d differ" % len(same))
sys.exit(1 if bad_total else 0)
