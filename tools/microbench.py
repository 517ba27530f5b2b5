#!/usr/bin/env python3
"""Time single C-ABI launches at benchmark shapes (used under rocprofv3 for counters).
usage: python tools/microbench.py conv16|conv32|conv64|conv7|wgrad16|wgrad32|tail [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubresnet_amd import ops
from ubresnet_amd.ops import Affine

what = sys.argv[1] if len(sys.argv) > 1 else "conv16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.bfloat16
dev = "cuda"
N = 16


def mk(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(dt)


def aff(C):
    return Affine(torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev))


cfgs = {"conv16": (512, 16, 16, 3), "conv32": (256, 32, 32, 3), "conv64": (128, 64, 64, 3), "conv7": (512, 16, 16, 7),
        "conv128": (64, 128, 128, 3), "conv256": (32, 256, 256, 3), "conv512": (16, 512, 512, 3),
        "wgrad16": (512, 16, 16, 3), "wgrad32": (256, 32, 32, 3), "wgrad64": (128, 64, 64, 3), "wgrad256": (32, 256, 256, 3)}
if what.startswith("conv"):
    HW, Cin, Cout, k = cfgs[what]
    x, y = mk(N, HW, HW, Cin), torch.empty((N, HW, HW, Cout), dtype=dt, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wp = ops.pack_weights(w, dt, Cout, Cin, Cin * k * k, k * k, k * k)
    st = torch.zeros(32 * 2 * Cout, dtype=torch.float64, device=dev)
    xf = aff(Cin) if "noxf" not in sys.argv else None
    stats = st if "nostats" not in sys.argv else None
    hint = max([int(a[4:]) for a in sys.argv if a.startswith("hint")] + [0])
    ctaps = ops.conv_taps(k, 1, k // 2)
    fn = lambda: ops.conv(x, wp, y, ctaps, Cout, xf=xf, stats=stats, tile_hint=hint)
    nbytes = x.numel() * 2 + y.numel() * 2
    flops = 2.0 * N * HW * HW * Cin * Cout * k * k
elif what.startswith("wgrad"):
    HW, Cin, Cout, k = cfgs[what]
    x, g = mk(N, HW, HW, Cin), mk(N, HW, HW, Cout)
    dW = torch.empty(Cout, Cin, k, k, device=dev)
    ws = ops.WgradWorkspace()
    xfw = aff(Cin)
    taps = ops.conv_taps(k, 1, k // 2)
    fn = lambda: ops.wgrad(x, g, taps, dW, Cin * k * k, k * k, Cout, Cin, ws, xf=xfw)
    nbytes = x.numel() * 2 + g.numel() * 2
    flops = 2.0 * N * HW * HW * Cin * Cout * k * k
else:
    raise SystemExit("unknown " + what)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("%s: %.3f ms  %.0f GB/s  %.1f TFLOP/s" % (what, ms, nbytes / ms / 1e6, flops / ms / 1e9))
