#!/usr/bin/env python3
"""Block-tail / BatchNorm backward passes at the network's layer shapes, isolated, with the launch geometry swept through the
diagnostic build's ubr_tune_set (hipcc ... -DUBR_TUNE, loaded through UBR_LIB=).  Without a tune build only the default runs.
usage: python tools/elembench.py [iters]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ubresnet_amd import _lib as L, ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt, dev, N = torch.bfloat16, "cuda", 16
lib = L.lib()
tune = getattr(lib, "ubr_tune_set", None)
if tune is not None:
    tune.argtypes = [ctypes.c_char_p, ctypes.c_int]
    tune.restype = None


def mk(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(dt)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def cases(C):
    HW = 512 * 16 // C
    f32 = lambda: torch.rand(C, device=dev) + 0.5
    g, c, cb, o = mk(N, HW, HW, C), mk(N, HW, HW, C), mk(N, HW, HW, C), mk(N, HW, HW, C)
    sc, sh, mu, istd, k1, k2 = f32(), f32(), f32(), f32(), f32(), f32()
    red, redb = torch.zeros(32 * 2 * C, dtype=torch.float64, device=dev), torch.zeros(32 * 2 * C, dtype=torch.float64, device=dev)
    gc, gs = torch.empty_like(c), torch.empty_like(c)
    mask = torch.randint(0, 255, (N * HW * HW * C // 8,), dtype=torch.uint8, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    cnt = N * HW * HW
    u = g.numel() * 2
    return {
        "bnred": (lambda: ops.bn_bwd_reduce(g, None, c, sc, sh, mu, istd, True, red), 2 * u),
        "bnapp": (lambda: ops.bn_bwd_apply(g, None, c, sc, sh, mu, istd, True, k1, k2, gc), 3 * u),
        "bnappfin": (lambda: ops.bn_bwd_apply_fin(g, None, c, sc, sh, mu, istd, True, red, cnt, dg, db, gc), 3 * u),
        "tailred": (lambda: ops.block_tail_bwd_reduce(g, None, o, c, sc, sh, mu, istd, None, None, None, red, None, relu_mask=mask), 2 * u + u // 16),
        "tailredbyp": (lambda: ops.block_tail_bwd_reduce(g, None, o, c, sc, sh, mu, istd, cb, mu, istd, red, redb, relu_mask=mask), 3 * u + u // 16),
        "tailapp": (lambda: ops.block_tail_bwd_apply(g, None, o, c, sc, sh, mu, istd, k1, k2, None, None, None, None, None, None, gc, gs, relu_mask=mask), 4 * u + u // 16),
        "tailappfin_nosc": (lambda: ops.block_tail_bwd_apply_fin(g, None, mask, c, sc, sh, mu, istd, red, dg, db, None, None, None, None, None, None, None, cnt, gc, None), 3 * u + u // 16),
        "tailappbyp": (lambda: ops.block_tail_bwd_apply(g, None, o, c, sc, sh, mu, istd, k1, k2, cb, sc, mu, istd, k1, k2, gc, gs, relu_mask=mask), 5 * u + u // 16),
        "tailfwd": (lambda: ops.block_tail_fwd(c, mu, sc, sh, cb, None, None, None, gc, relu_mask=mask), 3 * u + u // 16),
        "tailfwdbyp": (lambda: ops.block_tail_fwd(c, mu, sc, sh, cb, mu, sc, sh, gc, relu_mask=mask), 3 * u + u // 16),
    }


def sweep(tag, settings):
    for key, val in settings.items():
        if tune is not None:
            tune(key.encode(), val)
    print("== %s %s" % (tag, settings if tune is not None else "(library defaults: no tune build)"), flush=True)
    for C in (16, 32, 64, 128, 512):
        cs = cases(C)
        line = "C=%-4d" % C
        for name, (fn, nbytes) in cs.items():
            us = timeit(fn)
            line += " %s %.1fus %.2fTB/s |" % (name, us, nbytes / us / 1e6)
        print(line, flush=True)
        del cs
        torch.cuda.empty_cache()


base = {"red_blocks": 0, "red_iters": 0, "flush": 1, "app_blocks": 0, "slots": 0}
sweep("default", base)
if tune is not None:
    for spec in sys.argv[2:]:          # e.g. red_blocks=256 app_blocks=1024,red_iters=16
        kv = dict(base)
        for item in spec.split(","):
            key, val = item.split("=")
            kv[key] = int(val)
        sweep(spec, kv)
