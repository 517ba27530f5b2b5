"""Operator layer: torch-tensor wrappers over the C ABI (include/ubresnet_hip.h).

Activations are NHWC torch tensors (shape [N,H,W,C], stride(3)==1); channel slices of a concat
buffer and stride-2 phase views are passed as ordinary torch views -- the kernels take explicit
strides.  Nothing here computes on the host or falls back to torch ops.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

NEG_BIG = -3.0e38

# Optional per-launch timing (bench.py's kernel breakdown): when set, every wrapped op records a
# torch.cuda.Event pair on the current stream (the stream the kernels are launched on).
_prof = None
_rec_sink = None      # ubresnet_amd.plan.Recording while a launch tape records: operator calls tag their launches


class LaunchProfiler:
    """collects (op, kernel symbol, shape signature, algorithmic bytes and flops, start/end events) per launch"""

    def __init__(self):
        self.records = []
        self.timed = []          # (op, kernel, shape, bytes, flops, seconds) from timed tape replays (ubresnet_amd.plan.TIMED)

    def summary(self, by="shape"):
        """aggregate -> {key: [launches, seconds, bytes, flops]}; by = 'shape' (op, shape) or 'kernel' (symbol)"""
        torch.cuda.synchronize()
        agg = {}
        for name, kern, sig, nbytes, flops, e0, e1 in self.records:
            k = (name, sig) if by == "shape" else kern
            a = agg.setdefault(k, [0, 0.0, 0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += nbytes
            a[3] += flops
        for name, kern, sig, nbytes, flops, sec in self.timed:
            k = (name, sig) if by == "shape" else kern
            a = agg.setdefault(k, [0, 0.0, 0, 0.0])
            a[0] += 1
            a[1] += sec
            a[2] += nbytes
            a[3] += flops
        return agg


_DT_NAME = {torch.float32: "float", torch.bfloat16: "bf16_t", torch.float16: "f16_t"}


def _act_bytes(args):
    n = 0
    for a in args:
        if isinstance(a, torch.Tensor) and a.dim() == 4:
            n += a.numel() * a.element_size()
    return n


def _timed(name):
    def deco(fn):
        def wrapper(*args, **kwargs):
            sink = _rec_sink
            if _prof is None and sink is None:
                return fn(*args, **kwargs)
            acts = [a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor) and a.dim() == 4]
            sig = " ".join("x".join(map(str, a.shape)) for a in acts[:3])
            taps = kwargs.get("taps", args[3] if name in ("conv",) and len(args) > 3 else (args[2] if name == "wgrad" and len(args) > 2 else None))
            ntaps = len(taps) if isinstance(taps, (list, tuple)) else 0
            if ntaps:
                sig += " taps%d" % ntaps
            if "S" in kwargs:
                sig += " S%d" % kwargs["S"]
            lab = sink.label_begin() if sink is not None else -1
            if _prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            r = fn(*args, **kwargs)
            if _prof is not None:
                e1.record()
            kern, flops = name, 0.0
            if name == "conv":
                x, wp, y = args[0], args[1], args[2]
                cout = args[4]
                npix = y.shape[0] * (y.shape[2] * y.shape[3] if y.shape[1] == cout and y.dtype == torch.float32 and kwargs.get("logsoftmax") else y.shape[1] * y.shape[2])
                flops = 2.0 * npix * x.shape[3] * cout * ntaps
                buf = C.create_string_buffer(160)
                L.lib().ubr_conv_last_kernel(buf, 160)
                kern = buf.value.decode()
            if sink is not None:
                sink.label_end(lab, (name, kern, sig, _act_bytes(acts), flops))
            if _prof is not None:
                _prof.records.append((name, kern, sig, _act_bytes(acts), flops, e0, e1))
            return r
        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        return wrapper
    return deco


def last_conv_kernel() -> str:
    """kernel symbol (as rocprofv3 prints it) of this thread's last ubr_conv launch"""
    buf = C.create_string_buffer(160)
    L.lib().ubr_conv_last_kernel(buf, 160)
    return buf.value.decode()


def _tv(t: Optional[torch.Tensor]) -> L.Tensor:
    if t is None:
        return L.Tensor(None, 0, 0, 0)
    assert t.dim() == 4 and t.stride(3) == 1, "NHWC view with contiguous channels expected"
    return L.Tensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def _ps(t: torch.Tensor) -> int:
    """pixel stride of a pixel-dense NHWC view (channel slices allowed)"""
    n, h, w, c = t.shape
    ps = t.stride(2)
    if not (t.stride(3) == 1 and t.stride(1) == w * ps and (n == 1 or t.stride(0) == h * w * ps)):
        raise RuntimeError("ubresnet_amd: expected a pixel-dense NHWC view, got shape %s strides %s" % (tuple(t.shape), t.stride()))
    return ps


class Affine:
    """per-channel v -> max((v - sub)*scale + shift, lo) folded into a consumer's operand load"""
    __slots__ = ("sub", "scale", "shift", "lo")

    def __init__(self, sub, scale, shift, lo):
        self.sub, self.scale, self.shift, self.lo = sub, scale, shift, lo

    def c(self) -> L.ChanAffine:
        return L.ChanAffine(self.sub.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(), self.lo.data_ptr())


def _xf(a: Optional[Affine]) -> L.ChanAffine:
    return a.c() if a is not None else L.ChanAffine(None, None, None, None)


# ------------------------------------------------------------------------------------------
# tap geometry
# ------------------------------------------------------------------------------------------
def conv_taps(k: int, dil: int, pad: int) -> List[Tuple[int, int, int]]:
    """nn.Conv2d forward taps: (dy, dx, weight tap index)"""
    return [(ky * dil - pad, kx * dil - pad, ky * k + kx) for ky in range(k) for kx in range(k)]


def conv_dgrad_taps_s1(k: int, dil: int, pad: int):
    """data gradient of a stride-1 conv: gx[i] = sum_k gy[i + pad - k*dil] * W[k]"""
    return [(pad - ky * dil, pad - kx * dil, ky * k + kx) for ky in range(k) for kx in range(k)]


def transposed_phase_taps(k: int, dil: int, pad: int, s: int, ry: int, rx: int):
    """taps of output phase (ry, rx) of a stride-s transposed conv (= dgrad of a stride-s conv, or
    ConvTranspose2d forward): o = i*s - pad + k*dil  =>  i = (a*s + r + pad - k*dil)/s"""
    out = []
    for ky in range(k):
        if (ry + pad - ky * dil) % s:
            continue
        for kx in range(k):
            if (rx + pad - kx * dil) % s:
                continue
            out.append(((ry + pad - ky * dil) // s, (rx + pad - kx * dil) // s, ky * k + kx))
    return out


# ------------------------------------------------------------------------------------------
# conv / pack / wgrad
# ------------------------------------------------------------------------------------------
@_timed("pack_weights")
def pack_weights(src: torch.Tensor, dtype: torch.dtype, M: int, K: int, sm: int, sk: int, ntaps: int,
                 tapidx: Optional[Sequence[int]] = None, Kpad: Optional[int] = None, src_offset: int = 0) -> torch.Tensor:
    """-> packed image [ntaps][Kpad/CPU][Mpad][CPU] (see ubr_pack_weights); src_offset in elements"""
    L.require_cuda(src, "weights")
    assert src.dtype == torch.float32 and src.is_contiguous()
    cpu = L.chans_per_unit(dtype)
    Mpad = (M + 15) // 16 * 16
    if Kpad is None:
        Kpad = (K + cpu - 1) // cpu * cpu
    dst = torch.empty((ntaps, Kpad // cpu, Mpad, cpu), dtype=dtype, device=src.device)
    idx = (C.c_int32 * ntaps)(*(tapidx if tapidx is not None else range(ntaps)))
    L.check(L.lib().ubr_pack_weights(L.dtype_id(dtype), src.data_ptr() + 4 * src_offset, dst.data_ptr(), M, Mpad, K, Kpad, sm, sk,
                                     ntaps, idx, L.stream_ptr()), "pack_weights")
    return dst


_TAP_CACHE = {}


def _tap_arrays(taps):
    """ctypes images of a tap list, built once per distinct list (the tap lists are a handful of module constants)"""
    key = tuple(taps)
    r = _TAP_CACHE.get(key)
    if r is None:
        n = len(key)
        r = ((C.c_int8 * n)(*[t[0] for t in key]), (C.c_int8 * n)(*[t[1] for t in key]), (C.c_uint8 * n)(*[t[2] for t in key]),
             max(t[2] for t in key))
        _TAP_CACHE[key] = r
    return r


@_timed("conv")
def conv(x: torch.Tensor, wp: torch.Tensor, y: torch.Tensor, taps, Cout: int, S: int = 1, iy0: int = 0, ix0: int = 0,
         xf: Optional[Affine] = None, bias: Optional[torch.Tensor] = None, addend: Optional[torch.Tensor] = None,
         stats: Optional[torch.Tensor] = None, logsoftmax: bool = False, tile_hint: int = 0, in_hw=None, act: int = 0,
         addend_mask: Optional[torch.Tensor] = None, bnb=None, stats_slots: int = 0):
    """One ubr_conv launch.  x: NHWC input view; y: NHWC output-grid view (or, with logsoftmax, the
    contiguous fp32 NCHW result); taps: [(dy,dx,packed tap index)].
    addend_mask: the ReLU bit mask of a block tail gating `addend` (out = conv + addend * bit);
    bnb = (c, mean, scale, shift, invstd): `stats` receives the BatchNorm-backward sums of a = relu(bn(c)) for g = this output."""
    d = L.ConvDesc()
    d.dtype = L.dtype_id(x.dtype)
    N, H, W, Cin = x.shape
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.x = _tv(x)
    d.xf = _xf(xf)
    d.w = wp.data_ptr()
    d.Cout, d.Cout_pad = Cout, wp.shape[2]
    if wp.shape[1] * wp.shape[3] != Cin:
        raise RuntimeError("conv: packed weights expect %d input channels, tensor has %d" % (wp.shape[1] * wp.shape[3], Cin))
    d.ntaps = len(taps)
    if not 1 <= len(taps) <= L.MAX_TAPS:
        raise RuntimeError("conv: %d taps unsupported" % len(taps))
    dy_a, dx_a, wt_a, wt_max = _tap_arrays(taps)
    C.memmove(d.dy, dy_a, len(taps)); C.memmove(d.dx, dx_a, len(taps)); C.memmove(d.wt, wt_a, len(taps))
    if wt_max >= wp.shape[0]:
        raise RuntimeError("conv: tap index %d outside packed image" % wt_max)
    d.S, d.iy0, d.ix0 = S, iy0, ix0
    if logsoftmax:
        assert y.dtype == torch.float32 and y.is_contiguous() and y.shape[1] == Cout
        d.OH, d.OW = y.shape[2], y.shape[3]
        d.y = L.Tensor(y.data_ptr(), 0, 0, 0)
        d.epilogue = 1
        if y.shape[0] != N:
            raise RuntimeError("conv: batch mismatch")
    else:
        assert y.dtype == x.dtype
        d.OH, d.OW = y.shape[1], y.shape[2]
        d.y = _tv(y)
        d.epilogue = 0
        if y.shape[0] != N or y.shape[3] != Cout:
            raise RuntimeError("conv: output view %s does not match N=%d Cout=%d" % (tuple(y.shape), N, Cout))
    if addend is not None:
        assert addend.shape == y.shape and addend.dtype == y.dtype
        d.addend = _tv(addend)
    d.bias = L.ptr(bias)
    d.stats = L.ptr(stats)
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.numel() >= 2 * Cout * L.STAT_SLOTS
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= Cout
    if xf is not None:
        assert xf.scale.numel() >= Cin
    d.tile_hint = tile_hint
    d.act = act
    d.stats_slots = stats_slots
    if addend_mask is not None:
        assert addend is not None and addend_mask.dtype == torch.uint8 and addend_mask.is_contiguous()
        assert addend_mask.numel() >= y.shape[0] * y.shape[1] * y.shape[2] * (Cout // L.chans_per_unit(x.dtype))
        d.addend_mask = addend_mask.data_ptr()
    if bnb is not None:
        cten, mean, scale, shift, invstd = bnb
        assert stats is not None and cten.shape == y.shape and cten.dtype == y.dtype
        d.bnb_c = _tv(cten)
        d.bnb_mean, d.bnb_scale, d.bnb_shift, d.bnb_invstd = mean.data_ptr(), scale.data_ptr(), shift.data_ptr(), invstd.data_ptr()
    L.check(L.lib().ubr_conv(C.byref(d), L.stream_ptr()), "conv")


@_timed("conv")
def conv_phases(x: torch.Tensor, wp: torch.Tensor, y0: torch.Tensor, taps, Cout: int, phases=None, y_full: torch.Tensor = None,
                addend_full: Optional[torch.Tensor] = None, xf: Optional[Affine] = None, bias: Optional[torch.Tensor] = None):
    """The output phases of a stride-2 transposed conv (or of the data gradient of a stride-2 conv) in ONE launch.
    y_full: the whole NHWC output [N, 2*OH, 2*OW, Cout]; y0 = y_full[:, 0::2, 0::2, :] (passed for the launch labels);
    phases: [(ry, rx, taps_of_phase)], taps = their concatenation; addend_full: optional tensor of y_full's shape."""
    d = L.ConvDesc()
    d.dtype = L.dtype_id(x.dtype)
    N, H, W, Cin = x.shape
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.x = _tv(x)
    d.xf = _xf(xf)
    d.w = wp.data_ptr()
    d.Cout, d.Cout_pad = Cout, wp.shape[2]
    if wp.shape[1] * wp.shape[3] != Cin:
        raise RuntimeError("conv: packed weights expect %d input channels, tensor has %d" % (wp.shape[1] * wp.shape[3], Cin))
    if not 1 <= len(taps) <= L.MAX_TAPS or not 2 <= len(phases) <= 4:
        raise RuntimeError("conv_phases: %d taps / %d phases unsupported" % (len(taps), len(phases)))
    dy_a, dx_a, wt_a, wt_max = _tap_arrays(taps)
    C.memmove(d.dy, dy_a, len(taps)); C.memmove(d.dx, dx_a, len(taps)); C.memmove(d.wt, wt_a, len(taps))
    if wt_max >= wp.shape[0]:
        raise RuntimeError("conv: tap index %d outside packed image" % wt_max)
    d.ntaps = len(taps)
    d.S, d.iy0, d.ix0 = 1, 0, 0
    assert y_full.dtype == x.dtype and y_full.shape[0] == N and y_full.shape[3] == Cout and y_full.shape[1] % 2 == 0 and y_full.shape[2] % 2 == 0
    d.OH, d.OW = y_full.shape[1] // 2, y_full.shape[2] // 2
    sn, sy, sx = y_full.stride(0), y_full.stride(1), y_full.stride(2)
    d.y = L.Tensor(y_full.data_ptr(), sn, 2 * sy, 2 * sx)
    if addend_full is not None:
        assert addend_full.shape == y_full.shape and addend_full.dtype == y_full.dtype and addend_full.stride(3) == 1
        an, ay, ax = addend_full.stride(0), addend_full.stride(1), addend_full.stride(2)
        d.addend = L.Tensor(addend_full.data_ptr(), an, 2 * ay, 2 * ax)
    d.bias = L.ptr(bias)
    d.nphase = len(phases)
    t0 = 0
    for i, (ry, rx, tp) in enumerate(phases):
        d.phase_tap0[i], d.phase_ntaps[i] = t0, len(tp)
        t0 += len(tp)
        d.phase_yoff[i] = ry * sy + rx * sx
        d.phase_aoff[i] = (ry * ay + rx * ax) if addend_full is not None else 0
    assert t0 == len(taps)
    L.check(L.lib().ubr_conv(C.byref(d), L.stream_ptr()), "conv (phased)")


class WgradWorkspace:
    """grow-only fp32 slab workspace shared by all weight-gradient launches of a backward pass"""

    def __init__(self):
        self.buf = None
        self.pin = False      # launch plans bake the address into their tapes: an outgrown buffer must stay allocated
        self.old = []

    def get(self, nbytes: int, device) -> torch.Tensor:
        n = (nbytes + 3) // 4
        if self.buf is None or self.buf.numel() < n or self.buf.device != device:
            if self.pin and self.buf is not None:
                self.old.append(self.buf)
            self.buf = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=device)
        return self.buf

    # slabs that must outlive the launch (their sums are deferred to a batched reduction): one bump arena PER STREAM, reset by
    # release().  Everything that touches a stream's arena is ordered by that stream; a shared arena would let a launch on one
    # stream overwrite slabs a batched sum on the other stream is still reading.
    def hold(self, nbytes: int, device, stream=None) -> torch.Tensor:
        n = ((nbytes + 3) // 4 + 63) // 64 * 64
        key = 0 if stream is None else stream.cuda_stream
        ar = self.__dict__.setdefault("_arenas", {})
        a = ar.get(key)
        if a is None or a[0].device != device or a[1] + n > a[0].numel():
            if a is not None:
                self.__dict__.setdefault("_arena_old", []).append(a[0])      # pending items still point into it
            a = ar[key] = [torch.empty(max(2 * n, 64 << 20), dtype=torch.float32, device=device), 0]
        t = a[0][a[1]:a[1] + n]
        a[1] += n
        return t

    def release(self, stream=None):
        a = self.__dict__.get("_arenas", {}).get(0 if stream is None else stream.cuda_stream)
        if a is not None:
            a[1] = 0
        if not self.pin:
            self.__dict__.get("_arena_old", []).clear()


class ReduceBatch:
    """Weight-gradient slab sums deferred to ONE launch per group (ubr_wgrad_reduce_batched): `wgrad(..., defer=batch)` launches
    the MFMA kernel only; `flush()` sums every pending item's slabs -- the same additions per element as ubr_wgrad_reduce."""

    def __init__(self, ws: "WgradWorkspace"):
        self.ws = ws
        self.items = []          # (WgradReduceItem fields, slab bytes)
        self.stream = None

    def add(self, item, nbytes, stream):
        assert not self.items or stream is self.stream, "a batch belongs to one stream"
        self.stream = stream
        self.items.append((item, nbytes))

    def flush(self):
        if not self.items:
            return
        lib = L.lib()
        stream = self.stream
        st = L.stream_ptr() if stream is None else stream.cuda_stream
        sink = _rec_sink
        for lo in range(0, len(self.items), L.REDUCE_BATCH):
            chunk = self.items[lo:lo + L.REDUCE_BATCH]
            arr = (L.WgradReduceItem * len(chunk))(*[c[0] for c in chunk])
            if _prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            lab = sink.label_begin() if sink is not None else -1
            L.check(lib.ubr_wgrad_reduce_batched(arr, len(chunk), st), "wgrad_reduce_batched")
            meta = ("wgrad_reduce", "wgrad_reduce_batched_kernel", "%d weight gradients" % len(chunk), sum(c[1] for c in chunk), 0.0)
            if sink is not None:
                sink.label_end(lab, meta)
            if _prof is not None:
                e1.record(stream)
                _prof.records.append(meta + (e0, e1))
        self.items = []
        self.ws.release(stream)


_TAP_CACHE_W = {}


def _tap_arrays_w(taps):
    key = tuple(taps)
    r = _TAP_CACHE_W.get(key)
    if r is None:
        n = len(key)
        r = ((C.c_int8 * n)(*[t[0] for t in key]), (C.c_int8 * n)(*[t[1] for t in key]), (C.c_int32 * n)(*[t[2] for t in key]), n)
        _TAP_CACHE_W[key] = r
    return r


def wgrad(x: torch.Tensor, g: torch.Tensor, taps, dst: torch.Tensor, sm: int, sk: int, Cout_valid: int, Cin_valid: int,
          ws: WgradWorkspace, S: int = 1, iy0: int = 0, ix0: int = 0, xf: Optional[Affine] = None, accumulate: bool = False,
          dst_offset: int = 0, stream=None, exclusive: bool = False, defer: Optional["ReduceBatch"] = None):
    """dst[co*sm + ci*sk + tapidx] (+)= sum_pixels g[p][co] * xform(x)[p*S + tap][ci]
    taps: [(dy, dx, tapidx into the PyTorch weight layout)]; stream: torch.cuda.Stream to launch on (default: current)"""
    if defer is not None and (accumulate or (defer.items and defer.stream is not stream)):
        defer.flush()            # an accumulating sum must see the pending ones; a batch belongs to one stream
        if accumulate:
            defer = None
    d = L.WgradDesc()
    d.dtype = L.dtype_id(x.dtype)
    N, H, W, Cin = x.shape
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.x = _tv(x)
    d.xf = _xf(xf)
    assert g.dtype == x.dtype and g.shape[0] == N
    d.GH, d.GW, d.Cout = g.shape[1], g.shape[2], g.shape[3]
    d.g = _tv(g)
    d.ntaps = len(taps)
    dy_a, dx_a, _, _ = _tap_arrays_w(taps)
    C.memmove(d.dy, dy_a, len(taps)); C.memmove(d.dx, dx_a, len(taps))
    d.S, d.iy0, d.ix0 = S, iy0, ix0
    d.exclusive = 1 if exclusive else 0        # nothing runs beside this launch: fill the GPU
    nsplit, nbytes = C.c_int32(0), C.c_int64(0)
    lib = L.lib()
    L.check(lib.ubr_wgrad_plan(C.byref(d), C.byref(nsplit), C.byref(nbytes)), "wgrad_plan")
    slabs = ws.get(nbytes.value, x.device) if defer is None else ws.hold(nbytes.value, x.device, stream)
    if stream is not None:
        # the workspace is grow-only: when a later call replaces it, the caching allocator must not recycle the old block
        # for the compute stream while this (side-stream) launch still reads it
        slabs.record_stream(stream)
    d.slabs = slabs.data_ptr()
    d.nsplit = nsplit.value
    st = L.stream_ptr() if stream is None else stream.cuda_stream
    sink = _rec_sink
    if _prof is not None:
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(stream)
    lab0 = sink.label_begin() if sink is not None else -1
    L.check(lib.ubr_wgrad(C.byref(d), st), "wgrad")
    if _prof is not None:
        e1.record(stream)
    idx = _tap_arrays_w(taps)[2]
    assert dst.dtype == torch.float32
    lab1 = -1
    if defer is not None:
        it = L.WgradReduceItem()
        it.slabs, it.dst = slabs.data_ptr(), dst.data_ptr() + 4 * dst_offset
        it.nsplit, it.ntaps, it.Cout_pad, it.Cin, it.Cout_valid, it.Cin_valid = nsplit.value, len(taps), d.Cout, Cin, Cout_valid, Cin_valid
        it.accumulate, it.sm, it.sk = 1 if accumulate else 0, sm, sk
        for i in range(len(taps)):
            it.tapidx[i] = idx[i]
        defer.add(it, nsplit.value * len(taps) * d.Cout * Cin * 4, stream)
    else:
        lab1 = sink.label_begin() if sink is not None else -1
        L.check(lib.ubr_wgrad_reduce(slabs.data_ptr(), nsplit.value, len(taps), d.Cout, Cin, Cout_valid, Cin_valid,
                                     dst.data_ptr() + 4 * dst_offset, sm, sk, idx, 1 if accumulate else 0, st), "wgrad_reduce")
    if _prof is not None or sink is not None:
        if _prof is not None and defer is None:
            e2.record(stream)
        a, b, c, dd, bx = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        lib.ubr_wgrad_last_config(C.byref(a), C.byref(b), C.byref(c), C.byref(dd), C.byref(bx))
        kern = "wgrad_kernel<%s, %d, %d, %d, %s, %s, %s>" % (_DT_NAME[x.dtype], a.value, b.value, c.value, "true" if dd.value else "false",
                                                             "true" if bx.value else "false", "true" if lib.ubr_wgrad_last_pc() else "false")
        sig = "%s %s taps%d S%d" % ("x".join(map(str, x.shape)), "x".join(map(str, g.shape)), len(taps), S)
        nbytes = (x.numel() + g.numel()) * x.element_size()
        flops = 2.0 * g.shape[0] * g.shape[1] * g.shape[2] * g.shape[3] * Cin * len(taps)
        red = ("wgrad_reduce", "wgrad_reduce_kernel", sig, nsplit.value * len(taps) * d.Cout * Cin * 4, 0.0)
        if sink is not None:
            sink.labels[lab0] = ("wgrad", kern, sig, nbytes, flops)
            if lab1 >= 0:
                sink.label_end(lab1, red)
            else:
                sink.tape.set_label(-1)
        if _prof is not None:
            _prof.records.append(("wgrad", kern, sig, nbytes, flops, e0, e1))
            if defer is None:
                _prof.records.append(red + (e1, e2))


# ------------------------------------------------------------------------------------------
# stem
# ------------------------------------------------------------------------------------------
@_timed("stem_forward")
def stem_forward(x_nchw, weight, bias, y, stats):
    L.require_cuda(x_nchw, "input image")
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    N, Cin, H, W = x_nchw.shape
    Cout = weight.shape[0]
    assert weight.is_contiguous() and tuple(weight.shape) == (Cout, Cin, 7, 7)
    L.check(L.lib().ubr_stem_forward(L.dtype_id(y.dtype), x_nchw.data_ptr(), N, Cin, H, W, weight.data_ptr(), L.ptr(bias), Cout,
                                     _tv(y), L.ptr(stats), L.stream_ptr()), "stem_forward")


@_timed("stem_expand")
def stem_expand(x_nchw, out):
    """NCHW fp32 image -> NHWC [N,H,W,16*Cin]: channel 16*ci+kx = plane ci shifted by kx-3 columns"""
    L.require_cuda(x_nchw, "input image")
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    N, Cin, H, W = x_nchw.shape
    assert tuple(out.shape) == (N, H, W, 16 * Cin)
    L.check(L.lib().ubr_stem_expand(L.dtype_id(out.dtype), x_nchw.data_ptr(), N, Cin, H, W, out.data_ptr(), _ps(out),
                                    L.stream_ptr()), "stem_expand")


@_timed("stem_wgrad")
def stem_wgrad(x_nchw, g, dweight, dbias, ws: WgradWorkspace, accumulate=False):
    N, Cin, H, W = x_nchw.shape
    Cout = g.shape[3]
    lib = L.lib()
    nbytes = lib.ubr_stem_wgrad_workspace(N, Cin, H, W, Cout)
    part = ws.get(nbytes, g.device)
    L.check(lib.ubr_stem_wgrad(L.dtype_id(g.dtype), x_nchw.data_ptr(), N, Cin, H, W, _tv(g), Cout, part.data_ptr(), nbytes,
                               dweight.data_ptr(), L.ptr(dbias), 1 if accumulate else 0, L.stream_ptr()), "stem_wgrad")


# ------------------------------------------------------------------------------------------
# batch norm
# ------------------------------------------------------------------------------------------
def bn_finalize(stats, count, gamma, beta, rmean, rvar, nbt, momentum, eps, scale, shift, mean, invstd):
    Cn = gamma.numel()
    L.check(L.lib().ubr_bn_finalize(stats.data_ptr(), float(count), gamma.data_ptr(), beta.data_ptr(), L.ptr(rmean), L.ptr(rvar),
                                    L.ptr(nbt), float(momentum), float(eps), Cn, scale.data_ptr(), shift.data_ptr(),
                                    mean.data_ptr(), invstd.data_ptr(), L.stream_ptr()), "bn_finalize")


def bn_eval_affine(gamma, beta, rmean, rvar, eps, scale, shift, mean, invstd):
    L.check(L.lib().ubr_bn_eval_affine(gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), float(eps),
                                       gamma.numel(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                       L.stream_ptr()), "bn_eval_affine")


def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


@_timed("bn_bwd_reduce")
def bn_bwd_reduce(ga, ga2, c, scale, shift, mean, invstd, relu, red):
    L.check(L.lib().ubr_bn_bwd_reduce(L.dtype_id(c.dtype), _npix(c), c.shape[3], ga.data_ptr(), _ps(ga),
                                      L.ptr(ga2), _ps(ga2) if ga2 is not None else 0, c.data_ptr(), _ps(c),
                                      scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1 if relu else 0,
                                      red.data_ptr(), L.stream_ptr()), "bn_bwd_reduce")


def bn_bwd_finalize(red, count, Cn, dgamma, dbeta, accumulate, k1, k2):
    L.check(L.lib().ubr_bn_bwd_finalize(red.data_ptr(), float(count), None, None, Cn, L.ptr(dgamma), L.ptr(dbeta),
                                        1 if accumulate else 0, k1.data_ptr(), k2.data_ptr(), L.stream_ptr()), "bn_bwd_finalize")


@_timed("bn_bwd_apply")
def bn_bwd_apply(ga, ga2, c, scale, shift, mean, invstd, relu, k1, k2, gc):
    L.check(L.lib().ubr_bn_bwd_apply(L.dtype_id(c.dtype), _npix(c), c.shape[3], ga.data_ptr(), _ps(ga),
                                     L.ptr(ga2), _ps(ga2) if ga2 is not None else 0, c.data_ptr(), _ps(c),
                                     scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1 if relu else 0,
                                     k1.data_ptr(), k2.data_ptr(), gc.data_ptr(), _ps(gc), L.stream_ptr()), "bn_bwd_apply")


@_timed("bn_bwd_apply")
def bn_bwd_apply_fin(ga, ga2, c, scale, shift, mean, invstd, relu, red, count, dgamma, dbeta, gc):
    """apply pass with the finalize fused: k1 / k2 are formed from `red` inside the kernel, workgroup 0 writes dgamma / dbeta"""
    L.check(L.lib().ubr_bn_bwd_apply_fin(L.dtype_id(c.dtype), _npix(c), c.shape[3], ga.data_ptr(), _ps(ga),
                                         L.ptr(ga2), _ps(ga2) if ga2 is not None else 0, c.data_ptr(), _ps(c),
                                         scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1 if relu else 0,
                                         red.data_ptr(), float(count), L.ptr(dgamma), L.ptr(dbeta), gc.data_ptr(), _ps(gc), L.stream_ptr()),
            "bn_bwd_apply_fin")


# ------------------------------------------------------------------------------------------
# BasicBlock tail
# ------------------------------------------------------------------------------------------
@_timed("block_tail_fwd")
def block_tail_fwd(c2, mean2, scale2, shift2, sc, mean_b, scale_b, shift_b, out, relu_mask=None):
    """relu_mask: uint8 [npix * C / channels-per-unit], written with one bit per channel (stored output > 0) for the backward"""
    a = (L.dtype_id(c2.dtype), _npix(c2), c2.shape[3], c2.data_ptr(), _ps(c2), mean2.data_ptr(),
         scale2.data_ptr(), shift2.data_ptr(), sc.data_ptr(), _ps(sc), L.ptr(mean_b), L.ptr(scale_b),
         L.ptr(shift_b), out.data_ptr(), _ps(out))
    if relu_mask is None:
        L.check(L.lib().ubr_block_tail_fwd(*a, L.stream_ptr()), "block_tail_fwd")
    else:
        L.check(L.lib().ubr_block_tail_fwd_masked(*a, relu_mask.data_ptr(), L.stream_ptr()), "block_tail_fwd_masked")


def bn_fwd_fin(stats, bn, scale, shift, mean, invstd) -> "L.BnFwdFin":
    """descriptor of a train-mode BatchNorm site whose finalize is fused into its consumer (ubr_bn_fwd_fin)"""
    f = L.BnFwdFin()
    track = bn.track_running_stats and bn.running_mean is not None
    f.stats, f.gamma, f.beta = stats.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr()
    f.running_mean = bn.running_mean.data_ptr() if track else None
    f.running_var = bn.running_var.data_ptr() if track else None
    f.num_batches_tracked = bn.num_batches_tracked.data_ptr() if track else None
    f.momentum = -1.0 if bn.momentum is None else float(bn.momentum)
    f.eps = float(bn.eps)
    f.scale, f.shift, f.mean, f.invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    return f


@_timed("block_tail_fwd")
def block_tail_fwd_fin(c2, fin2, sc, fin_b, count, out, relu_mask=None):
    """block tail forward with the BatchNorm finalize(s) of bn2 (and bnpass) fused: fin2 / fin_b from bn_fwd_fin()"""
    L.check(L.lib().ubr_block_tail_fwd_fin(L.dtype_id(c2.dtype), _npix(c2), c2.shape[3], c2.data_ptr(), _ps(c2), C.byref(fin2),
                                           sc.data_ptr(), _ps(sc), C.byref(fin_b) if fin_b is not None else None, float(count),
                                           out.data_ptr(), _ps(out), L.ptr(relu_mask), L.stream_ptr()), "block_tail_fwd_fin")


@_timed("block_tail_bwd_reduce")
def block_tail_bwd_reduce(go, go2, out, c2, scale2, shift2, mean2, invstd2, cb, mean_b, invstd_b, red2, red_b, relu_mask=None):
    """relu_mask (from block_tail_fwd): read instead of `out`"""
    head = (L.dtype_id(c2.dtype), _npix(c2), c2.shape[3], go.data_ptr(), _ps(go), L.ptr(go2), _ps(go2) if go2 is not None else 0)
    rest = (c2.data_ptr(), _ps(c2), scale2.data_ptr(), shift2.data_ptr(), mean2.data_ptr(), invstd2.data_ptr(),
            L.ptr(cb), _ps(cb) if cb is not None else 0, L.ptr(mean_b), L.ptr(invstd_b), red2.data_ptr(), L.ptr(red_b), L.stream_ptr())
    if relu_mask is None:
        L.check(L.lib().ubr_block_tail_bwd_reduce(*head, out.data_ptr(), _ps(out), *rest), "block_tail_bwd_reduce")
    else:
        L.check(L.lib().ubr_block_tail_bwd_reduce_masked(*head, relu_mask.data_ptr(), *rest), "block_tail_bwd_reduce_masked")


@_timed("block_tail_bwd_apply")
def block_tail_bwd_apply(go, go2, out, c2, scale2, shift2, mean2, invstd2, k1_2, k2_2,
                         cb, scale_b, mean_b, invstd_b, k1_b, k2_b, g_c2, g_sc, relu_mask=None):
    head = (L.dtype_id(c2.dtype), _npix(c2), c2.shape[3], go.data_ptr(), _ps(go), L.ptr(go2), _ps(go2) if go2 is not None else 0)
    rest = (c2.data_ptr(), _ps(c2), scale2.data_ptr(), shift2.data_ptr(), mean2.data_ptr(), invstd2.data_ptr(),
            k1_2.data_ptr(), k2_2.data_ptr(),
            L.ptr(cb), _ps(cb) if cb is not None else 0, L.ptr(scale_b), L.ptr(mean_b), L.ptr(invstd_b), L.ptr(k1_b), L.ptr(k2_b),
            g_c2.data_ptr(), _ps(g_c2), g_sc.data_ptr(), _ps(g_sc), L.stream_ptr())
    if relu_mask is None:
        L.check(L.lib().ubr_block_tail_bwd_apply(*head, out.data_ptr(), _ps(out), *rest), "block_tail_bwd_apply")
    else:
        L.check(L.lib().ubr_block_tail_bwd_apply_masked(*head, relu_mask.data_ptr(), *rest), "block_tail_bwd_apply_masked")


@_timed("block_tail_bwd_apply")
def block_tail_bwd_apply_fin(go, go2, relu_mask, c2, scale2, shift2, mean2, invstd2, red2, dgamma2, dbeta2,
                             cb, scale_b, mean_b, invstd_b, red_b, dgamma_b, dbeta_b, count, g_c2, g_sc):
    """masked apply pass with both finalizes fused; g_sc may be None on an identity block (skip gradient re-formed by the consumer)"""
    L.check(L.lib().ubr_block_tail_bwd_apply_fin(
        L.dtype_id(c2.dtype), _npix(c2), c2.shape[3], go.data_ptr(), _ps(go), L.ptr(go2), _ps(go2) if go2 is not None else 0,
        relu_mask.data_ptr(), c2.data_ptr(), _ps(c2), scale2.data_ptr(), shift2.data_ptr(), mean2.data_ptr(), invstd2.data_ptr(),
        red2.data_ptr(), L.ptr(dgamma2), L.ptr(dbeta2),
        L.ptr(cb), _ps(cb) if cb is not None else 0, L.ptr(scale_b), L.ptr(mean_b), L.ptr(invstd_b), L.ptr(red_b), L.ptr(dgamma_b), L.ptr(dbeta_b),
        float(count), g_c2.data_ptr(), _ps(g_c2), L.ptr(g_sc), _ps(g_sc) if g_sc is not None else 0, L.stream_ptr()), "block_tail_bwd_apply_fin")


# ------------------------------------------------------------------------------------------
# max pool
# ------------------------------------------------------------------------------------------
@_timed("maxpool_fwd")
def maxpool_fwd(x, xf, pooled, xcopy, stride, argmax=None):
    """argmax: optional uint8 [N,OH,OW,C] receiving each window's arg-max tap (for maxpool_bwd)"""
    N, H, W, Cn = x.shape
    if argmax is not None:
        assert argmax.dtype == torch.uint8 and argmax.is_contiguous() and tuple(argmax.shape) == tuple(pooled.shape)
    L.check(L.lib().ubr_maxpool_fwd(L.dtype_id(x.dtype), N, H, W, Cn, stride, x.data_ptr(), _ps(x), _xf(xf),
                                    pooled.data_ptr(), _ps(pooled), L.ptr(xcopy), _ps(xcopy) if xcopy is not None else 0,
                                    L.ptr(argmax), L.stream_ptr()), "maxpool_fwd")


@_timed("maxpool_bwd")
def maxpool_bwd(x, xf, g_pooled, g_extra, gx, stride, argmax=None):
    N, H, W, Cn = x.shape
    if argmax is not None:
        assert argmax.dtype == torch.uint8 and argmax.is_contiguous() and tuple(argmax.shape) == tuple(g_pooled.shape)
    L.check(L.lib().ubr_maxpool_bwd(L.dtype_id(x.dtype), N, H, W, Cn, stride, x.data_ptr(), _ps(x), _xf(xf),
                                    g_pooled.data_ptr(), _ps(g_pooled), L.ptr(g_extra), _ps(g_extra) if g_extra is not None else 0,
                                    gx.data_ptr(), _ps(gx), L.ptr(argmax), L.stream_ptr()), "maxpool_bwd")


# ------------------------------------------------------------------------------------------
# head / loss / misc
# ------------------------------------------------------------------------------------------
@_timed("logsoftmax_bwd")
def logsoftmax_bwd(g_logp, logp, g_logits):
    N, Cn, H, W = logp.shape
    assert g_logp.is_contiguous() and logp.is_contiguous() and g_logp.dtype == torch.float32
    L.check(L.lib().ubr_logsoftmax_bwd(L.dtype_id(g_logits.dtype), N, Cn, H, W, g_logp.data_ptr(), logp.data_ptr(),
                                       g_logits.data_ptr(), _ps(g_logits), L.stream_ptr()), "logsoftmax_bwd")


def pixelwise_nll_fwd(predict, target, pixelweights, classw, ignore_index, acc, bad=None):
    N, Cn, H, W = predict.shape
    L.check(L.lib().ubr_pixelwise_nll_fwd(predict.data_ptr(), target.data_ptr(), pixelweights.data_ptr(), L.ptr(classw),
                                          N, Cn, H, W, int(ignore_index), acc.data_ptr(), L.ptr(bad), L.stream_ptr()), "pixelwise_nll_fwd")


def pixelwise_nll_bwd(g_loss, target, pixelweights, classw, ignore_index, shape, g_predict):
    N, Cn, H, W = shape
    L.check(L.lib().ubr_pixelwise_nll_bwd(g_loss.data_ptr(), target.data_ptr(), pixelweights.data_ptr(), L.ptr(classw),
                                          N, Cn, H, W, int(ignore_index), g_predict.data_ptr(), L.stream_ptr()), "pixelwise_nll_bwd")


def confusion(logp, target, cm):
    N, Cn, H, W = logp.shape
    L.check(L.lib().ubr_confusion(logp.data_ptr(), target.data_ptr(), N, Cn, H, W, cm.data_ptr(), L.stream_ptr()), "confusion")


@_timed("channel_sum")
def channel_sum(g, red):
    L.check(L.lib().ubr_channel_sum(L.dtype_id(g.dtype), _npix(g), g.shape[3], g.data_ptr(), _ps(g), red.data_ptr(),
                                    L.stream_ptr()), "channel_sum")


def stat_buffer(n: int, device) -> torch.Tensor:
    """zeroed fp64 accumulator [UBR_STAT_SLOTS][n] (see include/ubresnet_hip.h)"""
    t = torch.empty(L.STAT_SLOTS * n, dtype=torch.float64, device=device)
    zero_(t)
    return t


def cast_f64_to_f32(src, dst, n, scale=1.0, accumulate=False, stride=None, slots=L.STAT_SLOTS):
    """dst[:n] (+)= scale * sum over slots of src[slot*stride : slot*stride+n]"""
    L.check(L.lib().ubr_cast_f64_to_f32(src.data_ptr(), n if stride is None else stride, slots, dst.data_ptr(), n, float(scale),
                                        1 if accumulate else 0, L.stream_ptr()), "cast_f64_to_f32")


def zero_(t: torch.Tensor):
    assert t.is_contiguous()
    L.check(L.lib().ubr_zero(t.data_ptr(), t.numel() * t.element_size(), L.stream_ptr()), "zero")
