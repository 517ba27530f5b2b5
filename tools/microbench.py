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
        "conv1x1": (512, 16, 32, 1), "conv1x1b": (512, 32, 16, 1), "conv128": (64, 128, 128, 3), "conv256": (32, 256, 256, 3), "conv512": (16, 512, 512, 3),
        "wgrad16": (512, 16, 16, 3), "wgrad32": (256, 32, 32, 3), "wgrad64": (128, 64, 64, 3), "wgrad128": (64, 128, 128, 3), "wgrad256": (32, 256, 256, 3), "wgrad512": (16, 512, 512, 3), "wgradstem": (512, 16, 16, 7)}
cfgs.update({"conv32s2": (256, 32, 64, 3), "conv64s2": (128, 64, 128, 3), "conv128s2": (64, 128, 256, 3), "conv256s2": (32, 256, 512, 3),
             "conv32s2k1": (256, 32, 64, 1), "conv64s2k1": (128, 64, 128, 1)})
if what.startswith("conv"):
    HW, Cin, Cout, k = cfgs[what]
    S = 2 if "s2" in what else 1
    x, y = mk(N, HW, HW, Cin), torch.empty((N, HW // S, HW // S, Cout), dtype=dt, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wp = ops.pack_weights(w, dt, Cout, Cin, Cin * k * k, k * k, k * k)
    st = torch.zeros(32 * 2 * Cout, dtype=torch.float64, device=dev)
    xf = aff(Cin) if "noxf" not in sys.argv else None
    stats = st if "nostats" not in sys.argv else None
    hint = max([int(a[4:]) for a in sys.argv if a.startswith("hint")] + [0])
    ctaps = ops.conv_taps(k, 1, k // 2)
    stamps = torch.zeros(16 * 8192, dtype=torch.int64, device=dev)
    os.environ["UBR_CONV_STAMP_PTR"] = str(stamps.data_ptr())     # read once by a -DUBR_CONV_STAMPS build
    import atexit
    def _dump():
        st = stamps.cpu().view(-1, 16)
        if ops.last_conv_kernel().startswith("conv_pc_kernel"):
            st = st[st[:, 3] > 0].float()
            if len(st):
                m = st.median(0).values
                print("pc stamps (median cycles over %d workgroups, %d cin blocks each): consumer wave 0: MFMA loops %d, epilogues %d, barrier waits %d, total %d | "
                      "producer wave 4: transform + LDS stores %d, load issue %d, barrier waits %d, waiting for loads %d, total %d"
                      % (len(st), m[4], m[0], m[1], m[2], m[3], m[8], m[9], m[10], m[11], m[12]))
            return
        st = st[st[:, 4] > 0]
        if len(st):
            m = st.float().median(0).values
            print("   prologue %d, main loop %d, epilogue %d cycles (medians)" % ((st[:, 5] - st[:, 7]).float().median(), (st[:, 8] - st[:, 5]).float().median(), (st[:, 9] - st[:, 8]).float().median()))
            if (st[:, 15] > 0).any():
                print("   prologue of wave 0: entry -> halo loads issued %d, tap / weight-source tables built %d, first barrier %d, weight offsets + block-0 weight loads issued (+ constants to LDS) %d"
                      % ((st[:, 10] - st[:, 7]).float().median(), (st[:, 14] - st[:, 10]).float().median(), (st[:, 15] - st[:, 14]).float().median(), (st[:, 5] - st[:, 15]).float().median()))
            print("   epilogue of wave 0: values + stores issued %d, statistics (wave sums, LDS, atomics) %d, waiting for the other waves %d"
                  % ((st[:, 12] - st[:, 11]).float().median(), (st[:, 13] - st[:, 12]).float().median(), (st[:, 9] - st[:, 13]).float().median()))
            print("stamps (median cycles of wave 0 per workgroup, %d WGs): barriers %d, load wait %d, transform+LDS store %d, load issue %d, mfma loop %d, thin-kernel epilogue %d, main loop total %d"
                  % (len(st), m[3], m[6], m[0], m[1], m[2], m[10], m[4]))
    atexit.register(_dump)
    ad = torch.zeros_like(y) if "addend" in sys.argv else None
    slots = max([int(a[5:]) for a in sys.argv if a.startswith("slots")] + [0])
    fn = lambda: ops.conv(x, wp, y, ctaps, Cout, S=S, xf=xf, stats=stats, tile_hint=hint, addend=ad, stats_slots=slots)
    nbytes = x.numel() * 2 + y.numel() * 2
    flops = 2.0 * N * (HW // S) * (HW // S) * Cin * Cout * k * k
elif what.startswith("wgrad"):
    HW, Cin, Cout, k = cfgs[what]
    x, g = mk(N, HW, HW, Cin), mk(N, HW, HW, Cout)
    dW = torch.empty(Cout, Cin, k, k, device=dev)
    ws = ops.WgradWorkspace()
    xfw = aff(Cin) if "noxf" not in sys.argv else None
    taps = ops.conv_taps(k, 1, k // 2)
    if what == "wgradstem":       # the column-expanded stem: 7 vertical taps over 16 channels at full resolution
        taps = [(dy, 0, dy + 3) for dy in range(-3, 4)]
    if "kernelonly" in sys.argv:      # time ubr_wgrad alone (no slab reduction)
        import ctypes as C
        from ubresnet_amd import _lib as L
        d = L.WgradDesc(); d.dtype = L.dtype_id(dt); d.N, d.H, d.W, d.Cin = N, HW, HW, Cin
        d.x = ops._tv(x); d.xf = ops._xf(xfw) if xfw is not None else d.xf; d.GH, d.GW, d.Cout = HW, HW, Cout; d.g = ops._tv(g); d.ntaps = len(taps)
        for i, t in enumerate(taps): d.dy[i], d.dx[i] = t[0], t[1]
        d.S, d.iy0, d.ix0 = 1, 0, 0
        ns, nb = C.c_int32(0), C.c_int64(0)
        L.check(L.lib().ubr_wgrad_plan(C.byref(d), C.byref(ns), C.byref(nb)), "plan")
        slabs = torch.empty(nb.value // 4 + 16, device=dev); d.slabs = slabs.data_ptr(); d.nsplit = ns.value
        print("nsplit", ns.value, "slab MB", nb.value / 1e6)
        stamps = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
        os.environ["UBR_WGRAD_STAMP_PTR"] = str(stamps.data_ptr())     # read once by a -DUBR_WGRAD_STAMPS build
        fn = lambda: L.check(L.lib().ubr_wgrad(C.byref(d), L.stream_ptr()), "wgrad")
        import atexit
        def ops_last_wgrad_pc():
            return Cout % 64 == 0 and Cin % 64 == 0 and k == 3 and os.environ.get("UBR_WGRAD_PC", "1") != "0"
        def _dump():
            st = stamps.cpu().view(-1, 8)
            if ops_last_wgrad_pc():
                st = stamps.cpu().view(-1, 16)
                st = st[st[:, 4] > 0].float()
                if len(st):
                    m = st.median(0).values
                    print("pc stamps (median cycles per workgroup over %d WGs): consumer wave 0: barrier waits %d, compute %d, epilogue %d, total %d | producer wave 4: transform (+ load wait) %d, "
                          "load issue %d, LDS writes %d, barrier waits %d, total %d" % (len(st), m[6], m[2], m[3], m[4], m[8], m[9], m[10], m[11], m[12]))
                return
            st = st[st[:, 4] > 0]
            if len(st):
                m = st.float().median(0).values
                print("stamps (median cycles per workgroup over %d WGs): transform+LDS store %d, barriers %d, load wait %d, load issue %d, compute %d, epilogue %d, total %d"
                      % (len(st), m[0], m[6], m[7], m[1], m[2], m[3], m[4]))
        atexit.register(_dump)
    else:
        fn = lambda: ops.wgrad(x, g, taps, dW, Cin * k * k, k * k, Cout, Cin, ws, xf=xfw)
    nbytes = x.numel() * 2 + g.numel() * 2
    flops = 2.0 * N * HW * HW * Cin * Cout * k * k
elif what.startswith(("bnred", "bnapp", "tailred", "tailapp")):
    # BatchNorm / block-tail backward passes at a network level: bnred16 = reduce pass, 16 channels at 512x512 (32 -> 256x256 ...)
    C = int(what.lstrip("bnredaptil"))
    HW = 512 * 16 // C
    f32 = lambda: torch.rand(C, device=dev) + 0.5
    g, c, o = mk(N, HW, HW, C), mk(N, HW, HW, C), mk(N, HW, HW, C)
    sc, sh, mu, istd, k1, k2 = f32(), f32(), f32(), f32(), f32(), f32()
    red = torch.zeros(64 * 2 * C, dtype=torch.float64, device=dev)
    gc, gs = torch.empty_like(c), torch.empty_like(c)
    if what.startswith("bnred"):
        fn = lambda: ops.bn_bwd_reduce(g, None, c, sc, sh, mu, istd, True, red)
        nbytes = 2 * g.numel() * 2
    elif what.startswith("bnapp"):
        fn = lambda: ops.bn_bwd_apply(g, None, c, sc, sh, mu, istd, True, k1, k2, gc)
        nbytes = 3 * g.numel() * 2
    elif what.startswith("tailred"):
        fn = lambda: ops.block_tail_bwd_reduce(g, None, o, c, sc, sh, mu, istd, None, None, None, red, None)
        nbytes = 3 * g.numel() * 2
    else:
        fn = lambda: ops.block_tail_bwd_apply(g, None, o, c, sc, sh, mu, istd, k1, k2, None, None, None, None, None, None, gc, gs)
        nbytes = 5 * g.numel() * 2
    flops = 0.0
else:
    raise SystemExit("unknown " + what)
for _ in range(3):
    fn()
torch.cuda.synchronize()
if what.startswith("conv"):
    print("   kernel:", ops.last_conv_kernel())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("%s: %.3f ms  %.0f GB/s  %.1f TFLOP/s" % (what, ms, nbytes / ms / 1e6, flops / ms / 1e9))
