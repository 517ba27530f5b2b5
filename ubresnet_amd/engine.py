"""Graph executor for the U-ResNet family: drives the HIP kernels for forward and backward.

The reference expresses the network as torch.nn layer calls (models/ub_uresnet.py:88-147,
models/common_layers.py:39-58,127-132) and leaves scheduling to autograd.  Here the module tree
only OWNS parameters; this executor runs the fused schedule explicitly:

  * NHWC activations; skip connections are written straight into the channel slice of the
    decoder's concat buffer (torch.cat of models/common_layers.py:130 never materialises).
  * train-mode BatchNorm: the producing conv accumulates sum/sum-of-squares in its epilogue,
    a tiny finalize kernel makes (scale, shift), and the CONSUMER applies scale/shift/ReLU
    while loading its operand -- conv -> BN -> ReLU costs one write and one read.
  * backward is scheduled by hand in reverse order; parameter gradients land in one flat fp32
    buffer laid out in completion order so data-parallel all-reduce can start per stage.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from . import _lib as L
from . import ops
from .ops import Affine

T3 = ops.conv_taps(3, 1, 1)
T1 = ops.conv_taps(1, 1, 0)
T7 = ops.conv_taps(7, 1, 3)
DG3 = ops.conv_dgrad_taps_s1(3, 1, 1)
DG1 = ops.conv_dgrad_taps_s1(1, 1, 0)
DG7 = ops.conv_dgrad_taps_s1(7, 1, 3)


import os as _os
# UBR_INFER_FOLD=0: eval forward on the training schedule (BatchNorm applied on load, separate block tails) -- for A/B tests
_INFER_FOLD = _os.environ.get("UBR_INFER_FOLD", "1") != "0"
_RELU_MASK = _os.environ.get("UBR_RELU_MASK", "1") != "0"      # block tails keep their final ReLU's mask as bits for the backward
# BatchNorm-backward finalize fused into the apply pass (every workgroup re-sums the reduce pass's 8 stripes) up to this many
# channels.  Measured: isolated, the fused apply costs +0.3 ... +2 us over the plain one at every width (512 channels: 8.7 vs
# 8.4 us) against ~5 us of finalize launch; in the train step 64 vs all widths is within noise -- all widths, for the launches.
_FIN_MAX_C = int(_os.environ.get("UBR_FIN_MAX_C", "1024"))
# BatchNorm-backward reduce passes folded into the epilogue of the data-gradient conv that produces their gradient operand
# (ubr_conv_desc.bnb_c), and identity-block skip gradients re-formed from the ReLU bit mask in the consuming conv's epilogue
# (ubr_conv_desc.addend_mask) instead of being written by the tail's backward
_BNB_FUSE = _os.environ.get("UBR_BNB_FUSE", "0") != "0"
_MASK_ADDEND = _os.environ.get("UBR_MASK_ADDEND", "1") != "0"
# every k-th weight gradient of a backward pass stays on the compute stream (0 = all on the side stream): static balancing of
# the two streams of the backward pass
_WG_MAIN_EVERY = int(_os.environ.get("UBR_WGRAD_MAIN_EVERY", "0"))
# UBR_WGRAD_ORDER=after: a block's weight gradients are issued right AFTER the data-gradient conv that shares their gradient
# operand (single-stream schedule: the operand is then still in L2 / MALL); default: before it (side stream: earliest start)
_WG_AFTER = _os.environ.get("UBR_WGRAD_ORDER", "before") == "after"
# train-mode forward: the finalize launches of a block's bn2 / bnpass are fused into the block tail kernel
_DEFER_REDUCE = _os.environ.get("UBR_DEFER_REDUCE", "1") != "0"     # weight-gradient slab sums: one launch per backward stage
_TAIL_FIN = _os.environ.get("UBR_TAIL_FIN", "1") != "0"
# the four output phases of a transposed conv / of a stride-2 conv's data gradient in ONE launch when the layer has at least this
# many output channels (conv_igemm_kernel: a phase per blockIdx.z; conv_thin_kernel, Cin <= 32 without an addend: a phase loop over
# one staged halo)
_PHASE_MIN_C = int(_os.environ.get("UBR_PHASE_MIN_C", "16"))


def _phased(k, pad):
    """[(ry, rx, taps)] of the non-empty output phases of a stride-2 transposed conv, and their concatenation"""
    ph = [(ry, rx, ops.transposed_phase_taps(k, 1, pad, 2, ry, rx)) for ry in range(2) for rx in range(2)]
    ph = [p for p in ph if p[2]]
    return ph, [t for p in ph for t in p[2]]


_PH4 = _phased(4, 1)
_PH3 = _phased(3, 1)


def _phase(t, ry, rx):
    """stride-2 phase view of an NHWC tensor"""
    return t[:, ry::2, rx::2, :]


class BNSite:
    """Per-BatchNorm2d runtime vectors (views into the per-pass workspaces)."""
    __slots__ = ("mod", "C", "stats", "scale", "shift", "mean", "invstd", "red", "k1", "k2")

    def __init__(self, mod):
        self.mod = mod
        self.C = mod.num_features


class Saved:
    """What one forward pass keeps for backward."""
    pass


class Engine:
    def __init__(self, model, kind: str):
        self.model = model
        self.kind = kind  # "uresnet" | "aspp"
        self._plans: Dict[tuple, dict] = {}
        self._images: Dict[tuple, torch.Tensor] = {}
        self._const: Dict[tuple, torch.Tensor] = {}
        self.wws = ops.WgradWorkspace()
        self.side = None
        self._side_on = False
        self._wg_count = 0
        self._evs, self._ev_next, self._main = [], 0, None
        self._red_buf, self._red_off, self._red_elems = None, 0, 0
        self._bwd_packed, self._pack_evs = None, None
        self._rec = None                      # plan being recorded (ubresnet_amd/plan.py)
        self._planned: Dict[tuple, object] = {}
        self.wws.pin = True                   # tapes bake workspace addresses: outgrown slabs stay allocated
        self.bn_sites: List[BNSite] = []
        self._bn_of: Dict[int, BNSite] = {}
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                s = BNSite(m)
                self.bn_sites.append(s)
                self._bn_of[id(m)] = s
        # gradient layout: parameters in the order their gradients complete in backward
        self.grad_order = list(model._grad_completion_order())
        self.grad_offsets = {}
        off = 0
        for name, p in self.grad_order:
            self.grad_offsets[name] = off
            off += (p.numel() + 3) // 4 * 4
        self.grad_numel = off

    # ------------------------------------------------------------------ helpers
    def bn(self, mod) -> BNSite:
        return self._bn_of[id(mod)]

    def const(self, device, value: float, n: int) -> torch.Tensor:
        key = (device, value)
        t = self._const.get(key)
        if t is None or t.numel() < n:
            t = torch.full((max(n, 2048),), value, dtype=torch.float32, device=device)
            self._const[key] = t
        return t

    def _new(self, shape, dtype=None, device=None):
        """every per-pass buffer: while a launch plan is being recorded the tensor is pinned to the plan (its device
        address is baked into the tape, and the caching allocator must not hand the block to anything else)"""
        t = torch.empty(shape, dtype=dtype, device=device)
        if self._rec is not None:
            self._rec.keep.append(t)
        return t

    @staticmethod
    def _fresh(shape, dtype=None, device=None):
        """tensors handed to the caller (the log-probabilities): new memory on every pass, planned or not"""
        return torch.empty(shape, dtype=dtype, device=device)

    def _untaped(self, fn):
        """ops whose operands are not owned by the plan (the caller's image, the loss gradient, the fresh output):
        run now, stay off the tape; a replayed pass calls them from Python around the replay"""
        if self._rec is None:
            return fn()
        self._rec.tape.pause()
        try:
            return fn()
        finally:
            self._rec.tape.resume()

    def _fork(self, a, b):
        if self._rec is not None and self._rec.nstreams > 1:
            self._rec.tape.fork(a, b)

    def relu_affine(self, site: BNSite) -> Affine:
        return Affine(site.mean, site.scale, site.shift, self.const(site.scale.device, 0.0, site.C))

    # ------------------------------------------------------------------ weight images
    # Every pass repacks ALL weight images with one batched launch per direction (forward images at the start
    # of forward, data-gradient images at the start of backward).  Nothing is cached across passes: fused
    # optimizers (torch._fused_adam_) update parameters without bumping their version counters, so a cache keyed
    # on `_version` silently trains on stale weights.
    def _plan_items(self):
        """[(group, key, param, src_offset, M, Kvalid, Kpad_or_None, sm, sk, ntaps, tap_stride)]"""
        m = self.model
        items = []
        conv1 = getattr(m, "conv1", None)
        conv11 = getattr(m, "conv11", None)
        for mod in m.modules():
            if isinstance(mod, torch.nn.ConvTranspose2d):
                w = mod.weight
                d0, d1, kh, kw = w.shape
                kk = kh * kw
                items.append(("fwd", (id(w), "tfwd"), w, 0, d1, d0, None, kk, d1 * kk, kk, 1))
                items.append(("bwd", (id(w), "tdgrad"), w, 0, d0, d1, None, d1 * kk, kk, kk, 1))
            elif isinstance(mod, torch.nn.Conv2d):
                w = mod.weight
                d0, d1, kh, kw = w.shape
                kk = kh * kw
                if mod is conv1 and self.kind != "custom":
                    for ci in range(d1):     # stem: packed[ky][kx (7 of 16)][co] = w[co][ci][ky][kx]
                        items.append(("fwd", (id(w), "stem%d" % ci), w, ci * 49, d0, 7, 16, d1 * 49, 1, 7, 7))
                    continue
                items.append(("fwd", (id(w), "fwd"), w, 0, d0, d1, None, d1 * kk, kk, kk, 1))
                if mod is conv11 and self.kind != "custom":   # K (= num_classes) zero-padded to the 16 channels of g_logits
                    items.append(("bwd", (id(w), "dgrad"), w, 0, d1, d0, 16, kk, d1 * kk, kk, 1))
                else:
                    items.append(("bwd", (id(w), "dgrad"), w, 0, d1, d0, None, kk, d1 * kk, kk, 1))
        if self.kind == "custom":             # test harnesses wrap single blocks: a 7x7 stem conv gets stem images too
            for mod in m.modules():
                if isinstance(mod, torch.nn.Conv2d) and mod.kernel_size == (7, 7) and mod.in_channels <= 4:
                    w = mod.weight
                    for ci in range(w.shape[1]):
                        items.append(("fwd", (id(w), "stem%d" % ci), w, ci * 49, w.shape[0], 7, 16, w.shape[1] * 49, 1, 7, 7))
        return items

    def _pack_plan(self, dt, device):
        import struct
        key = (dt, device)
        plan = self._plans.get(key)
        ptrs = tuple(p.data_ptr() for _, p in self.grad_order)
        if plan is not None and plan["ptrs"] == ptrs:
            return plan
        cpu = L.chans_per_unit(dt)
        images, tables = {}, {"fwd": b"", "bwd": b""}
        counts = {"fwd": 0, "bwd": 0}
        for group, k, w, soff, M, Kv, Kpad, sm, sk, ntaps, tstride in self._plan_items():
            if not w.is_contiguous() or w.dtype != torch.float32 or w.device != device:
                raise RuntimeError("ubresnet_amd: parameters must be contiguous float32 on %s" % device)
            Mpad = (M + 15) // 16 * 16
            Kp = Kpad if Kpad is not None else (Kv + cpu - 1) // cpu * cpu
            dst = self._new((ntaps, Kp // cpu, Mpad, cpu), dtype=dt, device=device)
            images[k] = dst
            tables[group] += struct.pack("<QQqqqQiiiiii", w.data_ptr() + 4 * soff, dst.data_ptr(), sm, sk, tstride, 0, M, Mpad, Kv, Kp // cpu, ntaps, 0)
            counts[group] += 1
        plan = {"ptrs": ptrs, "images": images, "counts": counts}
        for g in ("fwd", "bwd"):
            plan[g] = torch.frombuffer(bytearray(tables[g]), dtype=torch.uint8).to(device) if counts[g] else None
        self._plans[key] = plan
        return plan

    def pack_all(self, dt, device, group, stream=None):
        plan = self._pack_plan(dt, device)
        if plan["counts"][group]:
            st = L.stream_ptr() if stream is None else stream.cuda_stream
            L.check(L.lib().ubr_pack_weights_batched(L.dtype_id(dt), plan[group].data_ptr(), plan["counts"][group], st),
                    "pack_weights_batched")
        self._images = plan["images"]

    def _pack_bwd_early(self, dt, dev):
        """Training forward: the backward-orientation weight images are not needed before backward starts, so their
        repack runs on the side stream under the forward pass instead of at the head of the backward chain."""
        import os
        self._bwd_packed = None
        if dev.type != "cuda" or os.environ.get("UBR_WGRAD_STREAM", "1") == "0":
            return
        self._ensure_side(dev)
        if self._pack_evs is None:
            self._pack_evs = (torch.cuda.Event(), torch.cuda.Event())
        e0, e1 = self._pack_evs
        e0.record(torch.cuda.current_stream(dev))      # weights are final and every earlier reader of the images is queued
        self.side.wait_event(e0)
        self._fork(0, 1)
        self.pack_all(dt, dev, "bwd", stream=self.side)
        e1.record(self.side)
        self._bwd_packed = (dt, dev)

    def _pack_bwd(self, dt, dev):
        if self._bwd_packed == (dt, dev):
            torch.cuda.current_stream(dev).wait_event(self._pack_evs[1])
            self._fork(1, 0)
            self._bwd_packed = None
            self._images = self._pack_plan(dt, dev)["images"]
        else:
            self.pack_all(dt, dev, "bwd")

    def packed(self, param: torch.Tensor, dtype, orient: str) -> torch.Tensor:
        """packed image of a weight for this pass (written by pack_all)"""
        return self._images[(id(param), orient)]

    def _packed_dgrad_padded(self, w_param, dt):
        return self._images[(id(w_param), "dgrad")]

    def _alloc_pass_workspaces(self, sv: Saved, device, training: bool):
        nf = sum(4 * s.C for s in self.bn_sites)
        sv.fws = self._new(nf, dtype=torch.float32, device=device)
        off = 0
        for s in self.bn_sites:
            s.scale = sv.fws[off:off + s.C]; off += s.C
            s.shift = sv.fws[off:off + s.C]; off += s.C
            s.mean = sv.fws[off:off + s.C]; off += s.C
            s.invstd = sv.fws[off:off + s.C]; off += s.C
        sv.sites = [(s, s.scale, s.shift, s.mean, s.invstd) for s in self.bn_sites]
        if training:
            NS = L.STAT_SLOTS
            nd = sum(2 * s.C for s in self.bn_sites) * NS
            sv.dws = self._new(nd, dtype=torch.float64, device=device)
            ops.zero_(sv.dws)
            off = 0
            for s in self.bn_sites:
                s.stats = sv.dws[off:off + 2 * s.C * NS]; off += 2 * s.C * NS
        else:
            for s in self.bn_sites:
                s.stats = None

    def _rebind(self, sv: Saved):
        """point the BN sites at the vectors of the pass that is being back-propagated"""
        for s, scale, shift, mean, invstd in sv.sites:
            s.scale, s.shift, s.mean, s.invstd = scale, shift, mean, invstd

    def _finish_bn(self, site: BNSite, count: int, training: bool):
        m = site.mod
        if training:
            mom = -1.0 if m.momentum is None else m.momentum     # None: cumulative moving average (factor 1 / num_batches_tracked)
            track = m.track_running_stats and m.running_mean is not None
            ops.bn_finalize(site.stats, count, m.weight, m.bias, m.running_mean if track else None,
                            m.running_var if track else None, m.num_batches_tracked if track else None,
                            mom, m.eps, site.scale, site.shift, site.mean, site.invstd)
        else:
            ops.bn_eval_affine(m.weight, m.bias, m.running_mean, m.running_var, m.eps, site.scale, site.shift, site.mean, site.invstd)

    # ------------------------------------------------------------------ BasicBlock
    def block_fwd(self, blk, x, out, training, dt, xf_in=None):
        """BasicBlock.forward (models/common_layers.py:39-58); x, out: NHWC views.
        xf_in: per-channel affine of a virtual input (only for blocks with a bypass conv)."""
        if xf_in is not None and blk.bypass is None:
            raise RuntimeError("identity-shortcut block cannot take a virtual (affine) input")
        N, H, W, Cin = x.shape
        S = blk.stride
        OH, OW = out.shape[1], out.shape[2]
        Cout = out.shape[3]
        dev = x.device
        bn1, bn2 = self.bn(blk.bn1), self.bn(blk.bn2)
        cnt = N * OH * OW
        c1 = self._new((N, OH, OW, Cout), dtype=dt, device=dev)
        ops.conv(x, self.packed(blk.conv1.weight, dt, "fwd"), c1, T3, Cout, S=S, xf=xf_in, stats=bn1.stats)
        self._finish_bn(bn1, cnt, training)
        c2 = self._new((N, OH, OW, Cout), dtype=dt, device=dev)
        fuse = _TAIL_FIN and training and 24 * Cout <= 65536
        slots = L.RED_SLOTS if fuse else 0
        ops.conv(c1, self.packed(blk.conv2.weight, dt, "fwd"), c2, T3, Cout, xf=self.relu_affine(bn1), stats=bn2.stats, stats_slots=slots)
        if not fuse:
            self._finish_bn(bn2, cnt, training)
        cb = None
        if blk.bypass is not None:
            bnb = self.bn(blk.bnpass)
            cb = self._new((N, OH, OW, Cout), dtype=dt, device=dev)
            ops.conv(x, self.packed(blk.bypass.weight, dt, "fwd"), cb, T1, Cout, S=S, xf=xf_in, stats=bnb.stats, stats_slots=slots)
            if not fuse:
                self._finish_bn(bnb, cnt, training)
        # the final ReLU's mask as one byte per 16-byte channel unit: the backward's two passes read it instead of `out`
        mask = None
        if self._save and _RELU_MASK:
            mask = self._new((cnt * (Cout // L.chans_per_unit(dt)),), dtype=torch.uint8, device=dev)
        if fuse:
            f2 = ops.bn_fwd_fin(bn2.stats, blk.bn2, bn2.scale, bn2.shift, bn2.mean, bn2.invstd)
            fb = ops.bn_fwd_fin(bnb.stats, blk.bnpass, bnb.scale, bnb.shift, bnb.mean, bnb.invstd) if blk.bypass is not None else None
            ops.block_tail_fwd_fin(c2, f2, cb if blk.bypass is not None else x, fb, cnt, out, relu_mask=mask)
        elif blk.bypass is not None:
            ops.block_tail_fwd(c2, bn2.mean, bn2.scale, bn2.shift, cb, bnb.mean, bnb.scale, bnb.shift, out, relu_mask=mask)
        else:
            ops.block_tail_fwd(c2, bn2.mean, bn2.scale, bn2.shift, x, None, None, None, out, relu_mask=mask)
        if not self._save:
            return None
        rec = Saved()
        rec.blk, rec.x, rec.c1, rec.c2, rec.cb, rec.out, rec.xf_in = blk, x, c1, c2, cb, out, xf_in
        rec.mask = mask
        return rec

    # ------------------------------------------------------------------ backward reduction arena
    def _red_begin(self, dev):
        """one zeroed fp64 arena per backward pass for every striped reduction buffer (one memset instead of ~45)"""
        if self._red_elems == 0:
            NS = L.STAT_SLOTS
            tot = 0
            for s_ in self.bn_sites:
                tot += 2 * s_.C * NS
            self._red_elems = 2 * tot + 64 * NS * 8      # BN sites (block tails take 2 sites' worth) + head/stem/bias sums
        self._red_buf = self._new(self._red_elems, dtype=torch.float64, device=dev)
        ops.zero_(self._red_buf)
        self._red_off = 0

    def _red(self, n, dev):
        k = L.STAT_SLOTS * n
        if self._red_buf is None or self._red_off + k > self._red_buf.numel() or self._red_buf.device != dev:
            t = ops.stat_buffer(n, dev)             # outside a backward pass, or arena exhausted
            if self._rec is not None:
                self._rec.keep.append(t)
            return t
        t = self._red_buf[self._red_off:self._red_off + k]
        self._red_off += k
        return t

    # ------------------------------------------------------------------ weight-gradient side stream
    def _side_begin(self, dev):
        """Weight gradients depend on nothing downstream, so they run on a second HIP stream next to the
        dgrad / BatchNorm-backward chain: the low-resolution layers launch too few workgroups to fill 256 CUs
        on their own.  UBR_WGRAD_STREAM=0 serialises everything on one stream (clean per-kernel timings); the
        launch profiler otherwise times kernels under the same two-stream contention as the real step (and as
        rocprofv3 sees them)."""
        import os
        self._side_on = dev.type == "cuda" and os.environ.get("UBR_WGRAD_STREAM", "1") != "0"
        self._wg_count = 0
        if self._side_on:
            self._ensure_side(dev)
        if self._side_on:
            self._main = torch.cuda.current_stream(dev)
            self._ev_next = 0

    def _ensure_side(self, dev):
        if self.side is not None:
            return
        import os
        # HIP maps normal-priority streams round-robin onto a few hardware queues; once RCCL has created its own
        # streams the side stream can land on the compute stream's queue and the two serialise (measured under
        # torchrun: 17.7 instead of 15.1 ms/step).  High-priority streams use separate queues, so in a
        # process-group job the side stream is created with high priority (costs 0.2 ms/step standalone).
        import torch.distributed as _dist
        in_job = _dist.is_available() and _dist.is_initialized()
        prio = int(os.environ.get("UBR_SIDE_PRIORITY", "-1" if in_job else "0"))
        self.side = torch.cuda.Stream(device=dev, priority=prio)

    def _wg_flush(self):
        """one launch for the slab sums of every weight gradient issued since the last flush (ops.ReduceBatch)"""
        b = self.__dict__.get("_red_batch")
        if b is not None:
            b.flush()

    def _side_end(self, dev):
        self._wg_flush()
        if self._side_on:
            torch.cuda.current_stream(dev).wait_stream(self.side)
            self._fork(1, 0)
            self._side_on = False

    def _wg(self, x, g, *args, **kw):
        self._wg_count += 1
        if _DEFER_REDUCE and "defer" not in kw:
            b = self.__dict__.get("_red_batch")
            if b is None:
                b = self.__dict__["_red_batch"] = ops.ReduceBatch(self.wws)
            kw["defer"] = b
        if not self._side_on or (_WG_MAIN_EVERY > 0 and self._wg_count % _WG_MAIN_EVERY == 0):
            return ops.wgrad(x, g, *args, **kw)
        # side stream waits for everything queued on the compute stream so far (x and g are produced there); the
        # launch goes straight to the side stream's handle -- no stream-context switch, one pooled event per call
        ev = self._event()
        ev.record(self._main)
        self.side.wait_event(ev)
        self._fork(0, 1)
        ops.wgrad(x, g, *args, stream=self.side, **kw)
        # the caching allocator must not hand these blocks to later main-stream kernels while the side stream reads them
        x.record_stream(self.side)
        g.record_stream(self.side)

    def _event(self):
        i = self._ev_next
        if i == len(self._evs):
            self._evs.append(torch.cuda.Event())
        self._ev_next = i + 1
        return self._evs[i]

    def _bn_bwd(self, site: BNSite, ga, ga2, c, relu, G, cnt, red=None):
        """backward through a = relu(bn(c)) (or bn only): returns g_c; writes dgamma/dbeta.
        red: the reduce pass's sums, when the conv that produced `ga` already accumulated them in its epilogue"""
        if red is None:
            red = self._red(2 * site.C, c.device)
            ops.bn_bwd_reduce(ga, ga2, c, site.scale, site.shift, site.mean, site.invstd, relu, red)
        gc = self._new(c.shape, dtype=c.dtype, device=c.device)
        if site.C <= _FIN_MAX_C:
            ops.bn_bwd_apply_fin(ga, ga2, c, site.scale, site.shift, site.mean, site.invstd, relu, red, cnt,
                                 G(site.mod.weight), G(site.mod.bias), gc)
            return gc
        k = self._new(2 * site.C, dtype=torch.float32, device=c.device)
        k1, k2 = k[:site.C], k[site.C:]
        ops.bn_bwd_finalize(red, cnt, site.C, G(site.mod.weight), G(site.mod.bias), False, k1, k2)
        ops.bn_bwd_apply(ga, ga2, c, site.scale, site.shift, site.mean, site.invstd, relu, k1, k2, gc)
        return gc

    def _conv_dgrad(self, conv_mod, g, gx, S, addend=None, k=3, addend_mask=None, bnb=None, stats=None):
        """data gradient of Conv2d(k, stride S, pad k//2): g (conv output grad) -> gx (input grad view)"""
        dt = g.dtype
        wp = self.packed(conv_mod.weight, dt, "dgrad")
        Cin = gx.shape[3]
        pad = k // 2
        if S == 1:
            taps = DG3 if k == 3 else (DG1 if k == 1 else DG7)
            ops.conv(g, wp, gx, taps, Cin, addend=addend, addend_mask=addend_mask, bnb=bnb, stats=stats)
        else:
            if k == 3 and Cin >= _PHASE_MIN_C and len(_PH3[0]) == 4 and addend_mask is None and bnb is None:
                ops.conv_phases(g, wp, _phase(gx, 0, 0), _PH3[1], Cin, phases=_PH3[0], y_full=gx, addend_full=addend)
                return
            for ry in range(2):
                for rx in range(2):
                    taps = ops.transposed_phase_taps(k, 1, pad, 2, ry, rx)
                    if not taps:
                        continue   # (1x1 stride-2: only phase (0,0) receives gradient; addend already in place)
                    yv = _phase(gx, ry, rx)
                    av = _phase(addend, ry, rx) if addend is not None else None
                    ops.conv(g, wp, yv, taps, Cin, addend=av)

    def block_bwd(self, rec, go, go2, G, need_gx=True):
        """backward of BasicBlock; go (+go2): gradient wrt the block output. Returns g_x (or None)."""
        blk, x, c1, c2, cb, out = rec.blk, rec.x, rec.c1, rec.c2, rec.cb, rec.out
        dt, dev = c2.dtype, c2.device
        N, OH, OW, Cout = c2.shape
        cnt = N * OH * OW
        S = blk.stride
        bn1, bn2 = self.bn(blk.bn1), self.bn(blk.bn2)
        byp = cb is not None
        bnb = self.bn(blk.bnpass) if byp else None
        NS = L.STAT_SLOTS
        red = self._red((4 if byp else 2) * Cout, dev)
        red2 = red[:2 * Cout * NS]
        redb = red[2 * Cout * NS:] if byp else None
        mask = getattr(rec, "mask", None)
        ops.block_tail_bwd_reduce(go, go2, out, c2, bn2.scale, bn2.shift, bn2.mean, bn2.invstd,
                                  cb, bnb.mean if byp else None, bnb.invstd if byp else None, red2, redb, relu_mask=mask)
        g_c2 = self._new(c2.shape, dtype=dt, device=dev)
        # identity block with one gradient operand: the skip gradient go*[out>0] is not written; conv1's data-gradient epilogue
        # re-forms it from go and the bit mask
        lazy_sc = _MASK_ADDEND and mask is not None and not byp and go2 is None and need_gx and Cout <= _FIN_MAX_C
        g_sc = None if lazy_sc else self._new(c2.shape, dtype=dt, device=dev)
        if mask is not None and Cout <= _FIN_MAX_C:
            ops.block_tail_bwd_apply_fin(go, go2, mask, c2, bn2.scale, bn2.shift, bn2.mean, bn2.invstd, red2, G(blk.bn2.weight), G(blk.bn2.bias),
                                         cb, bnb.scale if byp else None, bnb.mean if byp else None, bnb.invstd if byp else None,
                                         redb, G(blk.bnpass.weight) if byp else None, G(blk.bnpass.bias) if byp else None, cnt, g_c2, g_sc)
        else:
            k = self._new(4 * Cout, dtype=torch.float32, device=dev)
            ops.bn_bwd_finalize(red2, cnt, Cout, G(blk.bn2.weight), G(blk.bn2.bias), False, k[:Cout], k[Cout:2 * Cout])
            if byp:
                ops.bn_bwd_finalize(redb, cnt, Cout, G(blk.bnpass.weight), G(blk.bnpass.bias), False, k[2 * Cout:3 * Cout], k[3 * Cout:])
            ops.block_tail_bwd_apply(go, go2, out, c2, bn2.scale, bn2.shift, bn2.mean, bn2.invstd, k[:Cout], k[Cout:2 * Cout],
                                     cb, bnb.scale if byp else None, bnb.mean if byp else None, bnb.invstd if byp else None,
                                     k[2 * Cout:3 * Cout] if byp else None, k[3 * Cout:] if byp else None, g_c2, g_sc, relu_mask=mask)
        # conv2: weight grad (input = relu(bn1(c1)) re-formed on load) and data grad
        kk = 9
        if not _WG_AFTER:
            self._wg(c1, g_c2, T3, G(blk.conv2.weight), Cout * kk, kk, Cout, Cout, self.wws, xf=self.relu_affine(bn1))
        g_a1 = self._new(c1.shape, dtype=dt, device=dev)
        if _BNB_FUSE:
            # the reduce pass of bn1's backward rides in the epilogue of the conv that produces its gradient operand
            red1 = self._red(2 * Cout, dev)
            self._conv_dgrad(blk.conv2, g_c2, g_a1, 1, bnb=(c1, bn1.mean, bn1.scale, bn1.shift, bn1.invstd), stats=red1)
            if _WG_AFTER:
                self._wg(c1, g_c2, T3, G(blk.conv2.weight), Cout * kk, kk, Cout, Cout, self.wws, xf=self.relu_affine(bn1))
            del g_c2
            g_c1 = self._bn_bwd(bn1, g_a1, None, c1, True, G, cnt, red=red1)
        else:
            self._conv_dgrad(blk.conv2, g_c2, g_a1, 1)
            if _WG_AFTER:
                self._wg(c1, g_c2, T3, G(blk.conv2.weight), Cout * kk, kk, Cout, Cout, self.wws, xf=self.relu_affine(bn1))
            del g_c2
            g_c1 = self._bn_bwd(bn1, g_a1, None, c1, True, G, cnt)
        del g_a1
        Cin = x.shape[3]

        def wg_conv1():
            self._wg(x, g_c1, T3, G(blk.conv1.weight), Cin * kk, kk, Cout, Cin, self.wws, S=S, xf=rec.xf_in)
            if byp:
                self._wg(x, g_sc, T1, G(blk.bypass.weight), Cin, 1, Cout, Cin, self.wws, S=S, xf=rec.xf_in)
        if not _WG_AFTER or not need_gx:
            wg_conv1()
        if not need_gx:
            return None
        gx = self._new(x.shape, dtype=dt, device=dev)
        if byp:
            self._conv_dgrad(blk.conv1, g_c1, gx, S)
            self._conv_dgrad(blk.bypass, g_sc, gx, S, addend=gx, k=1)
        elif lazy_sc:
            self._conv_dgrad(blk.conv1, g_c1, gx, S, addend=go, addend_mask=mask)
        else:
            self._conv_dgrad(blk.conv1, g_c1, gx, S, addend=g_sc)
        if _WG_AFTER:
            wg_conv1()
        return gx

    # ------------------------------------------------------------------ DoubleResNet
    def double_fwd(self, dbl, x, out, training, dt, xf_in=None):
        N, H, W, _ = x.shape
        S = dbl.res1.stride
        mid = self._new((N, out.shape[1], out.shape[2], out.shape[3]), dtype=dt, device=x.device)
        r1 = self.block_fwd(dbl.res1, x, mid, training, dt, xf_in)
        r2 = self.block_fwd(dbl.res2, mid, out, training, dt)
        return (r1, r2) if self._save else None

    def double_bwd(self, recs, go, go2, G, need_gx=True):
        r1, r2 = recs
        g_mid = self.block_bwd(r2, go, go2, G)
        return self.block_bwd(r1, g_mid, None, G, need_gx)

    # ------------------------------------------------------------------ ConvTransposeLayer
    def deconv_fwd(self, dl, x, cat, Cd, dt, xf_x=None):
        """ConvTranspose2d(k4,s2,p1) of x into channels [0,Cd) of the concat buffer (4 output phases)."""
        wp = self.packed(dl.deconv.weight, dt, "tfwd")
        up = cat[..., :Cd]
        if Cd >= _PHASE_MIN_C and len(_PH4[0]) == 4:
            ops.conv_phases(x, wp, _phase(up, 0, 0), _PH4[1], Cd, phases=_PH4[0], y_full=up, xf=xf_x)
            return
        for ry in range(2):
            for rx in range(2):
                ops.conv(x, wp, _phase(up, ry, rx), ops.transposed_phase_taps(4, 1, 1, 2, ry, rx), Cd, xf=xf_x)

    def declayer_fwd(self, dl, x, cat, Cd, out, training, dt, xf_x=None, xf_cat=None):
        """ConvTransposeLayer.forward (models/common_layers.py:127-132); the skip half of `cat` is already filled.
        xf_x / xf_cat: affines of a virtual deconv input / virtual skip channels (ASPP_ResNet)."""
        self.deconv_fwd(dl, x, cat, Cd, dt, xf_x)
        recs = self.double_fwd(dl.res, cat, out, training, dt, xf_cat)
        if not self._save:
            return None
        rec = Saved()
        rec.dl, rec.x, rec.cat, rec.Cd, rec.recs, rec.xf_x = dl, x, cat, Cd, recs, xf_x
        return rec

    def declayer_bwd(self, rec, go, G, xf_x: Optional[Affine] = None):
        """returns (g_x, g_cat); g_cat[..., Cd:] is the gradient of the skip tensor (w.r.t. the transformed
        values when the layer was given affines)"""
        dl, x, cat, Cd = rec.dl, rec.x, rec.cat, rec.Cd
        xf_x = rec.xf_x if xf_x is None else xf_x
        g_cat = self.double_bwd(rec.recs, go, None, G)
        g_up = g_cat[..., :Cd]
        Cin = x.shape[3]
        dW = G(dl.deconv.weight)
        for ry in range(2):
            for rx in range(2):
                taps = ops.transposed_phase_taps(4, 1, 1, 2, ry, rx)
                self._wg(x, _phase(g_up, ry, rx), taps, dW, 16, Cd * 16, Cd, Cin, self.wws, xf=xf_x)
        # data gradient of the transposed conv = ordinary stride-2 conv over g_up
        gx = self._new(x.shape, dtype=x.dtype, device=x.device)
        wp = self.packed(dl.deconv.weight, x.dtype, "tdgrad")
        ops.conv(g_up, wp, gx, ops.conv_taps(4, 1, 1), Cin, S=2)
        return gx, g_cat

    # ------------------------------------------------------------------ stem
    STEM_TAPS = [(ky - 3, 0, ky) for ky in range(7)]

    def stem_fwd(self, conv1, x, c0, stats, dt):
        """conv1 7x7 on the caller's NCHW image (models/ub_uresnet.py:41,94) on the matrix cores: the image is
        expanded to 16 channels per plane (column shifts -3..3), then each plane is a 7-tap vertical conv."""
        N, Cin, H, W = x.shape
        Cout = conv1.out_channels
        x16 = self._new((N, H, W, 16 * Cin), dtype=dt, device=x.device)
        self._untaped(lambda: ops.stem_expand(x, x16))
        if self._rec is not None:
            self._rec.pre = lambda xx: ops.stem_expand(xx, x16)
        w = conv1.weight
        for ci in range(Cin):
            last = ci == Cin - 1
            ops.conv(x16[..., 16 * ci:16 * ci + 16], self.packed(w, dt, "stem%d" % ci), c0, self.STEM_TAPS, Cout,
                     bias=conv1.bias if ci == 0 else None, addend=c0 if ci > 0 else None, stats=stats if last else None)
        return x16

    def stem_bwd(self, conv1, x16, g_c0, G):
        Cout = conv1.out_channels
        Cin = x16.shape[3] // 16
        dW = G(conv1.weight)
        taps = [(ky - 3, 0, 7 * ky) for ky in range(7)]
        for ci in range(Cin):
            # the last launch(es) of a backward pass: the compute stream has nothing left to run beside them
            self._wg(x16[..., 16 * ci:16 * ci + 16], g_c0, taps, dW, Cin * 49, 1, Cout, 7, self.wws, dst_offset=ci * 49, exclusive=True)
        red = self._red(Cout, g_c0.device)
        ops.channel_sum(g_c0, red)
        ops.cast_f64_to_f32(red, G(conv1.bias), Cout)


    # ------------------------------------------------------------------ head (shared by both networks)
    def head_fwd(self, m, d1o, training, dt):
        N, H, W, _ = d1o.shape
        bn10 = self.bn(m.bn10)
        nk = m.conv10.out_channels
        c10 = self._new((N, H, W, nk), dtype=dt, device=d1o.device)
        ops.conv(d1o, self.packed(m.conv10.weight, dt, "fwd"), c10, T7, nk, bias=m.conv10.bias, stats=bn10.stats)
        self._finish_bn(bn10, N * H * W, training)
        ncls = m.conv11.out_channels
        out = self._final_logsoftmax(m, c10, self.packed(m.conv11.weight, dt, "fwd"), self.relu_affine(bn10), (N, ncls, H, W))
        return c10, out

    def head_bwd(self, m, sv, g_logp, G):
        dt, dev = sv.dt, sv.x.device
        N, ncls, H, W = sv.out.shape
        ip = m.conv10.in_channels
        g_l = self._new((N, H, W, 16), dtype=dt, device=dev)
        self._logsoftmax_bwd(g_logp, sv.out, g_l)
        bn10 = self.bn(m.bn10)
        nk = m.conv10.out_channels
        self._wg(sv.c10, g_l, T7, G(m.conv11.weight), nk * 49, 49, ncls, nk, self.wws, xf=self.relu_affine(bn10))
        NS = L.STAT_SLOTS
        red = self._red(16 + nk, dev)
        ops.channel_sum(g_l, red[:16 * NS])
        ops.cast_f64_to_f32(red[:16 * NS], G(m.conv11.bias), ncls, stride=16)
        g_a10 = self._new((N, H, W, nk), dtype=dt, device=dev)
        red10 = self._red(2 * nk, dev) if _BNB_FUSE else None
        ops.conv(g_l, self._packed_dgrad_padded(m.conv11.weight, dt), g_a10, DG7, nk,
                 bnb=(sv.c10, bn10.mean, bn10.scale, bn10.shift, bn10.invstd) if _BNB_FUSE else None, stats=red10)
        del g_l
        g_c10 = self._bn_bwd(bn10, g_a10, None, sv.c10, True, G, N * H * W, red=red10)
        del g_a10
        self._wg(sv.d1o, g_c10, T7, G(m.conv10.weight), ip * 49, 49, nk, ip, self.wws)
        ops.channel_sum(g_c10, red[16 * NS:])
        ops.cast_f64_to_f32(red[16 * NS:], G(m.conv10.bias), nk)
        g = self._new(sv.d1o.shape, dtype=dt, device=dev)
        self._conv_dgrad(m.conv10, g_c10, g, 1, k=7)
        return g

    def _final_logsoftmax(self, m, c10, wp, xf, shape):
        """conv11 + bias + LogSoftmax (models/ub_uresnet.py:64,143) into a NEW fp32 NCHW tensor: the one forward op that is
        never taped, so callers get fresh memory from every pass"""
        def head():
            out = self._fresh(shape, dtype=torch.float32, device=c10.device)
            ops.conv(c10, wp, out, T7, shape[1], xf=xf, bias=m.conv11.bias, logsoftmax=True)
            return out
        if self._rec is not None:
            self._rec.post = head
        return self._untaped(head)

    def _logsoftmax_bwd(self, g_logp, out, g_l):
        """first op of backward; reads the loss gradient and the log-probabilities of THIS pass (caller-owned): never taped"""
        self._untaped(lambda: ops.logsoftmax_bwd(g_logp, out, g_l))
        if self._rec is not None:
            self._rec.pre = lambda g, o: ops.logsoftmax_bwd(g, o, g_l)

    def _check_input(self, x, cin):
        L.require_cuda(x, "input")
        if x.dtype != torch.float32:
            raise RuntimeError("ubresnet_amd: input must be float32 NCHW (got %s)" % x.dtype)
        if x.dim() != 4 or x.shape[1] != cin:
            raise RuntimeError("ubresnet_amd: expected input [B,%d,H,W], got %s" % (cin, tuple(x.shape)))
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise RuntimeError("ubresnet_amd: H and W must be multiples of 32 (ConvTranspose2d output_size contract of the "
                               "reference, models/common_layers.py:128); got %dx%d" % (x.shape[2], x.shape[3]))
        return x if x.is_contiguous() else x.contiguous()

    def _grad_views(self, dev):
        flat = self._new(self.grad_numel, dtype=torch.float32, device=dev)
        views = {}
        for name, p in self.grad_order:
            o = self.grad_offsets[name]
            n = p.numel()
            views[id(p)] = flat[o:o + n].view(p.shape)
            if n % 4:       # padding up to the next 16-byte boundary (conv11.bias of a 3-class net): defined, for flat optimizers
                flat[o + n:o + (n + 3) // 4 * 4].zero_()
        return flat, views

    def _stage_notifier(self, flat, grad_ready):
        done = [0]
        ids = [id(p) for _, p in self.grad_order]

        def stage_done(last_param):
            self._wg_flush()         # the stage's weight gradients are final only after their (batched) slab sums
            i = ids.index(id(last_param))
            hi = self.grad_offsets[self.grad_order[i][0]] + (self.grad_order[i][1].numel() + 3) // 4 * 4
            if self._rec is not None and hi > done[0]:
                # a replayed backward hands flat[done:hi] to the data-parallel reducer through these tape events
                m0 = self._rec.tape.mark(0)
                m1 = self._rec.tape.mark(1) if (self._side_on and self._rec.nstreams > 1) else None
                self._rec.stages.append((done[0], hi, m0, m1))
            if grad_ready is None:
                if hi > done[0]:
                    done[0] = hi
                return
            if hi > done[0]:
                if self._side_on:
                    # flat[done:hi] is final once BOTH streams reach this point: the consumer is told to wait for the
                    # side stream's event as well (neither producer stream stalls for the exchange)
                    ev = self._event()
                    ev.record(self.side)
                    grad_ready(flat, done[0], hi, wait_events=(ev,))
                else:
                    grad_ready(flat, done[0], hi)
                done[0] = hi
        return stage_done

    # ------------------------------------------------------------------ ASPP_ResNet (models/ASPP_ResNet.py:416-523)
    def _affine_arena(self, sv, device, sizes, relu_ranges):
        """per-pass [4][T] float arena of per-channel affines: identity (sub 0, scale 1, shift 0, lo -inf) except
        lo = 0 on `relu_ranges` (where BatchNorm+ReLU sites will be bound).  One template copy per pass."""
        T = sum(sizes)
        key = (device, tuple(sizes), tuple(relu_ranges))
        tmpl = self._const.get(key)
        if tmpl is None:
            tmpl = torch.zeros((4, T), dtype=torch.float32, device=device)
            tmpl[1].fill_(1.0)
            tmpl[3].fill_(ops.NEG_BIG)
            for lo, hi in relu_ranges:
                tmpl[3, lo:hi].zero_()
            self._const[key] = tmpl
        arena = self._new(tmpl.shape, dtype=tmpl.dtype, device=tmpl.device)
        arena.copy_(tmpl)
        sv.arena = arena
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += n
        return arena, offs

    def _bind_site(self, site, arena, off):
        """BatchNorm+ReLU site whose vectors live at arena[:, off:off+C]"""
        Cn = site.C
        site.mean, site.scale, site.shift = arena[0, off:off + Cn], arena[1, off:off + Cn], arena[2, off:off + Cn]

    def _arena_affine(self, arena, off, n):
        return Affine(arena[0, off:off + n], arena[1, off:off + n], arena[2, off:off + n], arena[3, off:off + n])

    def aspp_level_fwd(self, layer, post, e, cpost, arena, off_acat, training, dt):
        """ASPP.forward + ASPP_post.forward (models/ASPP_ResNet.py:227-263,280-286).  e: encoder output view
        [N,h,w,C]; cpost: destination view of the RAW 1x1 output (its BN+ReLU is folded into consumers)."""
        N, h, w, Cn = e.shape
        acat = self._new((N, h, w, 64 + Cn), dtype=dt, device=e.device)
        cnt = N * h * w
        for b, (conv, bn, k, dil) in enumerate(layer.branches()):
            site = self.bn(bn)
            ops.conv(e, self.packed(conv.weight, dt, "fwd"), acat[..., 16 * b:16 * b + 16], ops.conv_taps(k, dil, dil * (k // 2)),
                     16, bias=conv.bias, stats=site.stats)
            self._finish_bn(site, cnt, training)
        ops.maxpool_fwd(e, None, acat[..., 64:], None, 1)
        psite = self.bn(post.ASPP_bn)
        xf = self._arena_affine(arena, off_acat, 64 + Cn)
        ops.conv(acat, self.packed(post.ASPP_conv.weight, dt, "fwd"), cpost, T1, Cn, xf=xf, bias=post.ASPP_conv.bias, stats=psite.stats)
        self._finish_bn(psite, cnt, training)
        if not self._save:
            return None
        rec = Saved()
        rec.layer, rec.post, rec.e, rec.cpost, rec.acat, rec.xf = layer, post, e, cpost, acat, xf
        return rec

    def aspp_level_bwd(self, rec, g_post, g_base, G):
        """g_post: gradient w.r.t. relu(bn(cpost)) (a slice of the consumer's input gradient); g_base: gradient
        already owed to e through the direct skip (accumulated into the result).  Returns the ASPP's total g_e."""
        layer, post, e, cpost, acat = rec.layer, rec.post, rec.e, rec.cpost, rec.acat
        N, h, w, Cn = e.shape
        dt, dev, cnt = e.dtype, e.device, N * h * w
        psite = self.bn(post.ASPP_bn)
        g_cpost = self._bn_bwd(psite, g_post, None, cpost, True, G, cnt)
        self._wg(acat, g_cpost, T1, G(post.ASPP_conv.weight), 64 + Cn, 1, Cn, 64 + Cn, self.wws, xf=rec.xf)
        red = self._red(Cn, dev)
        ops.channel_sum(g_cpost, red)
        ops.cast_f64_to_f32(red, G(post.ASPP_conv.bias), Cn)
        g_acat = self._new(acat.shape, dtype=dt, device=dev)
        ops.conv(g_cpost, self.packed(post.ASPP_conv.weight, dt, "dgrad"), g_acat, DG1, 64 + Cn)
        del g_cpost
        g_e = self._new(e.shape, dtype=dt, device=dev)
        ops.maxpool_bwd(e, None, g_acat[..., 64:], g_base, g_e, 1)
        for b, (conv, bn, k, dil) in enumerate(layer.branches()):
            site = self.bn(bn)
            g_cb = self._bn_bwd(site, g_acat[..., 16 * b:16 * b + 16], None, acat[..., 16 * b:16 * b + 16], True, G, cnt)
            kk = k * k
            taps = ops.conv_taps(k, dil, dil * (k // 2))
            self._wg(e, g_cb, taps, G(conv.weight), Cn * kk, kk, 16, Cn, self.wws)
            redb = self._red(16, dev)
            ops.channel_sum(g_cb, redb)
            ops.cast_f64_to_f32(redb, G(conv.bias), 16)
            ops.conv(g_cb, self.packed(conv.weight, dt, "dgrad"), g_e, ops.conv_dgrad_taps_s1(k, dil, dil * (k // 2)), Cn, addend=g_e)
        return g_e

    def aspp_forward(self, x, training, dt, save):
        m = self.model
        self._save = save
        if save and not training:
            raise RuntimeError("ubresnet_amd: gradients through eval-mode BatchNorm are not supported; "
                               "call model.train() or wrap inference in torch.no_grad()")
        x = self._check_input(x, m.conv1.in_channels)
        N, Cin, H, W = x.shape
        dev, ip = x.device, m.inplanes
        sv = Saved()
        self._alloc_pass_workspaces(sv, dev, training)
        self.pack_all(dt, dev, "fwd")
        if save:
            self._pack_bwd_early(dt, dev)
        E = lambda *shape: self._new(shape, dtype=dt, device=dev)
        C3, C4, C5 = 8 * ip, 16 * ip, 32 * ip
        # affine arena: [acat3 | acat4 | acat5 | cat4 (up,post3,e3) | cat5 (up,post4,e4) | skip5 (post5,e5)]
        sizes = [64 + C3, 64 + C4, 64 + C5, C3 + 2 * C3, C4 + 2 * C4, 2 * C5]
        o, offs0 = 0, []
        for n in sizes:
            offs0.append(o)
            o += n
        relu_ranges = [(offs0[0], offs0[0] + 64), (offs0[1], offs0[1] + 64), (offs0[2], offs0[2] + 64),
                       (offs0[3] + C3, offs0[3] + 2 * C3), (offs0[4] + C4, offs0[4] + 2 * C4), (offs0[5], offs0[5] + C5)]
        arena, offs = self._affine_arena(sv, dev, sizes, relu_ranges)
        for lvl, o_acat in ((3, offs[0]), (4, offs[1]), (5, offs[2])):
            layer = getattr(m, "ASPP_layer_enc%d" % lvl)
            for b, (_, bn, _, _) in enumerate(layer.branches()):
                self._bind_site(self.bn(bn), arena, o_acat + 16 * b)
        self._bind_site(self.bn(m.ASPP_combine_enc3.ASPP_bn), arena, offs[3] + C3)
        self._bind_site(self.bn(m.ASPP_combine_enc4.ASPP_bn), arena, offs[4] + C4)
        self._bind_site(self.bn(m.ASPP_combine_enc5.ASPP_bn), arena, offs[5])
        sv.sites = [(s, s.scale, s.shift, s.mean, s.invstd) for s in self.bn_sites]

        bn1 = self.bn(m.bn1)
        c0 = E(N, H, W, ip)
        x16 = self.stem_fwd(m.conv1, x, c0, bn1.stats, dt)
        self._finish_bn(bn1, N * H * W, training)
        cat1 = E(N, H, W, 2 * ip)
        p0 = E(N, H // 2, W // 2, ip)
        amax = self._new((N, H // 2, W // 2, ip), dtype=torch.uint8, device=dev) if save else None   # window arg-max for the backward
        ops.maxpool_fwd(c0, self.relu_affine(bn1), p0, cat1[..., ip:], 2, argmax=amax)
        cat2 = E(N, H // 2, W // 2, 4 * ip)
        cat3 = E(N, H // 4, W // 4, 8 * ip)
        cat4 = E(N, H // 8, W // 8, 3 * C3)
        cat5 = E(N, H // 16, W // 16, 3 * C4)
        skip5 = E(N, H // 32, W // 32, 2 * C5)
        e1, e2 = cat2[..., 2 * ip:], cat3[..., 4 * ip:]
        e3, e4, e5 = cat4[..., 2 * C3:], cat5[..., 2 * C4:], skip5[..., C5:]
        r1 = self.double_fwd(m.enc_layer1, p0, e1, training, dt)
        r2 = self.double_fwd(m.enc_layer2, e1, e2, training, dt)
        r3 = self.double_fwd(m.enc_layer3, e2, e3, training, dt)
        r4 = self.double_fwd(m.enc_layer4, e3, e4, training, dt)
        r5 = self.double_fwd(m.enc_layer5, e4, e5, training, dt)
        a3 = self.aspp_level_fwd(m.ASPP_layer_enc3, m.ASPP_combine_enc3, e3, cat4[..., C3:2 * C3], arena, offs[0], training, dt)
        a4 = self.aspp_level_fwd(m.ASPP_layer_enc4, m.ASPP_combine_enc4, e4, cat5[..., C4:2 * C4], arena, offs[1], training, dt)
        a5 = self.aspp_level_fwd(m.ASPP_layer_enc5, m.ASPP_combine_enc5, e5, skip5[..., :C5], arena, offs[2], training, dt)

        d5o = E(N, H // 16, W // 16, 32 * ip)
        d5 = self.declayer_fwd(m.dec_layer5, skip5, cat5, C4, d5o, training, dt,
                               xf_x=self._arena_affine(arena, offs[5], 2 * C5), xf_cat=self._arena_affine(arena, offs[4], 3 * C4))
        d4o = E(N, H // 8, W // 8, 16 * ip)
        d4 = self.declayer_fwd(m.dec_layer4, d5o, cat4, C3, d4o, training, dt, xf_cat=self._arena_affine(arena, offs[3], 3 * C3))
        d3o = E(N, H // 4, W // 4, 4 * ip)
        d3 = self.declayer_fwd(m.dec_layer3, d4o, cat3, 4 * ip, d3o, training, dt)
        d2o = E(N, H // 2, W // 2, 2 * ip)
        d2 = self.declayer_fwd(m.dec_layer2, d3o, cat2, 2 * ip, d2o, training, dt)
        d1o = E(N, H, W, ip)
        d1 = self.declayer_fwd(m.dec_layer1, d2o, cat1, ip, d1o, training, dt)
        c10, out = self.head_fwd(m, d1o, training, dt)
        if not save:
            return out, None
        sv.x, sv.x16, sv.c0, sv.amax = x, x16, c0, amax
        sv.enc, sv.aspp, sv.dec = (r1, r2, r3, r4, r5), (a3, a4, a5), (d1, d2, d3, d4, d5)
        sv.d1o, sv.c10, sv.out, sv.dt = d1o, c10, out.detach(), dt     # (an alias without grad_fn: autograd attaches the node to `out` itself, and node -> ctx -> sv -> out would be a cycle)
        return out, sv

    def aspp_backward(self, sv, g_logp, grad_ready=None):
        m = self.model
        dt, dev = sv.dt, sv.x.device
        self._rebind(sv)
        self._pack_bwd(dt, dev)
        flat, views = self._grad_views(dev)
        G = lambda p: views[id(p)]
        stage_done = self._stage_notifier(flat, grad_ready)
        self._side_begin(dev)
        self._red_begin(dev)
        N, ncls, H, W = sv.out.shape
        ip = m.inplanes
        C3, C4, C5 = 8 * ip, 16 * ip, 32 * ip
        if not g_logp.is_contiguous():
            g_logp = g_logp.contiguous()
        g = self.head_bwd(m, sv, g_logp, G); stage_done(m.bn10.bias)
        d1, d2, d3, d4, d5 = sv.dec
        g, gc1 = self.declayer_bwd(d1, g, G); stage_done(m.dec_layer1.deconv.weight)
        g, gc2 = self.declayer_bwd(d2, g, G); stage_done(m.dec_layer2.deconv.weight)
        g, gc3 = self.declayer_bwd(d3, g, G); stage_done(m.dec_layer3.deconv.weight)
        g, gc4 = self.declayer_bwd(d4, g, G); stage_done(m.dec_layer4.deconv.weight)
        gs5, gc5 = self.declayer_bwd(d5, g, G); stage_done(m.dec_layer5.deconv.weight)
        r1, r2, r3, r4, r5 = sv.enc
        a3, a4, a5 = sv.aspp
        g_e5 = self.aspp_level_bwd(a5, gs5[..., :C5], gs5[..., C5:], G)
        g = self.double_bwd(r5, g_e5, None, G); stage_done(m.enc_layer5.res1.conv1.weight)
        g_e4 = self.aspp_level_bwd(a4, gc5[..., C4:2 * C4], gc5[..., 2 * C4:], G)
        g = self.double_bwd(r4, g, g_e4, G); stage_done(m.enc_layer4.res1.conv1.weight)
        g_e3 = self.aspp_level_bwd(a3, gc4[..., C3:2 * C3], gc4[..., 2 * C3:], G)
        g = self.double_bwd(r3, g, g_e3, G); stage_done(m.enc_layer3.res1.conv1.weight)
        g = self.double_bwd(r2, g, gc3[..., 4 * ip:], G); stage_done(m.enc_layer2.res1.conv1.weight)
        g = self.double_bwd(r1, g, gc2[..., 2 * ip:], G); stage_done(m.enc_layer1.res1.conv1.weight)
        bn1 = self.bn(m.bn1)
        g_x0 = self._new(sv.c0.shape, dtype=dt, device=dev)
        ops.maxpool_bwd(sv.c0, self.relu_affine(bn1), g, gc1[..., ip:], g_x0, 2, argmax=sv.amax)
        g_c0 = self._bn_bwd(bn1, g_x0, None, sv.c0, True, G, N * H * W)
        self.stem_bwd(m.conv1, sv.x16, g_c0, G)
        stage_done(self.grad_order[-1][1])
        self._side_end(dev)
        self._red_buf = None
        return flat, views

    # ------------------------------------------------------------------ UResNet
    def uresnet_forward(self, x: torch.Tensor, training: bool, dt: torch.dtype, save: bool):
        """training: BatchNorm uses batch statistics (and updates running stats); save: keep activations for backward"""
        m = self.model
        self._save = save
        if save and not training:
            raise RuntimeError("ubresnet_amd: gradients through eval-mode BatchNorm are not supported; "
                               "call model.train() or wrap inference in torch.no_grad()")
        L.require_cuda(x, "input")
        if x.dtype != torch.float32:
            raise RuntimeError("ubresnet_amd: input must be float32 NCHW (got %s)" % x.dtype)
        if x.dim() != 4 or x.shape[1] != m.conv1.in_channels:
            raise RuntimeError("ubresnet_amd: expected input [B,%d,H,W], got %s" % (m.conv1.in_channels, tuple(x.shape)))
        N, Cin, H, W = x.shape
        if H % 32 or W % 32:
            raise RuntimeError("ubresnet_amd: H and W must be multiples of 32 (ConvTranspose2d output_size contract of the "
                               "reference, models/common_layers.py:128); got %dx%d" % (H, W))
        if not x.is_contiguous():
            x = x.contiguous()
        dev = x.device
        ip = m.inplanes
        sv = Saved()
        self._alloc_pass_workspaces(sv, dev, training)
        self.pack_all(dt, dev, "fwd")
        if save:
            self._pack_bwd_early(dt, dev)
        E = lambda *shape: self._new(shape, dtype=dt, device=dev)

        # stem: conv1 -> (bn1 + relu folded into consumers) -> pool ; x0 goes into dec1's concat buffer
        bn1 = self.bn(m.bn1)
        c0 = E(N, H, W, ip)
        x16 = self.stem_fwd(m.conv1, x, c0, bn1.stats, dt)
        self._finish_bn(bn1, N * H * W, training)
        cat1 = E(N, H, W, 2 * ip)
        p0 = E(N, H // 2, W // 2, ip)
        amax = self._new((N, H // 2, W // 2, ip), dtype=torch.uint8, device=dev) if save else None   # window arg-max for the backward
        ops.maxpool_fwd(c0, self.relu_affine(bn1), p0, cat1[..., ip:], 2, argmax=amax)

        # encoder: each level's output is written into the skip half of the matching concat buffer
        cat2 = E(N, H // 2, W // 2, 4 * ip)
        cat3 = E(N, H // 4, W // 4, 8 * ip)
        cat4 = E(N, H // 8, W // 8, 16 * ip)
        cat5 = E(N, H // 16, W // 16, 32 * ip)
        x5 = E(N, H // 32, W // 32, 32 * ip)
        x1, x2, x3, x4 = cat2[..., 2 * ip:], cat3[..., 4 * ip:], cat4[..., 8 * ip:], cat5[..., 16 * ip:]
        e1 = self.double_fwd(m.enc_layer1, p0, x1, training, dt)
        e2 = self.double_fwd(m.enc_layer2, x1, x2, training, dt)
        e3 = self.double_fwd(m.enc_layer3, x2, x3, training, dt)
        e4 = self.double_fwd(m.enc_layer4, x3, x4, training, dt)
        e5 = self.double_fwd(m.enc_layer5, x4, x5, training, dt)

        d5o = E(N, H // 16, W // 16, 16 * ip)
        d5 = self.declayer_fwd(m.dec_layer5, x5, cat5, 16 * ip, d5o, training, dt)
        d4o = E(N, H // 8, W // 8, 8 * ip)
        d4 = self.declayer_fwd(m.dec_layer4, d5o, cat4, 8 * ip, d4o, training, dt)
        d3o = E(N, H // 4, W // 4, 4 * ip)
        d3 = self.declayer_fwd(m.dec_layer3, d4o, cat3, 4 * ip, d3o, training, dt)
        d2o = E(N, H // 2, W // 2, 2 * ip)
        d2 = self.declayer_fwd(m.dec_layer2, d3o, cat2, 2 * ip, d2o, training, dt)
        d1o = E(N, H, W, ip)
        d1 = self.declayer_fwd(m.dec_layer1, d2o, cat1, ip, d1o, training, dt)

        # head: conv10 + bias -> bn10 -> relu -> conv11 + bias -> log-softmax (fused epilogue, NCHW fp32)
        bn10 = self.bn(m.bn10)
        nk = m.conv10.out_channels
        c10 = E(N, H, W, nk)
        ops.conv(d1o, self.packed(m.conv10.weight, dt, "fwd"), c10, T7, nk, bias=m.conv10.bias, stats=bn10.stats)
        self._finish_bn(bn10, N * H * W, training)
        ncls = m.conv11.out_channels
        out = self._final_logsoftmax(m, c10, self.packed(m.conv11.weight, dt, "fwd"), self.relu_affine(bn10), (N, ncls, H, W))
        if not save:
            return out, None
        sv.x, sv.x16, sv.c0, sv.cat1, sv.p0, sv.amax = x, x16, c0, cat1, p0, amax
        sv.enc = (e1, e2, e3, e4, e5)
        sv.dec = (d1, d2, d3, d4, d5)
        sv.cats = (cat1, cat2, cat3, cat4, cat5)
        sv.d1o, sv.c10, sv.out, sv.dt = d1o, c10, out.detach(), dt     # (an alias without grad_fn: autograd attaches the node to `out` itself, and node -> ctx -> sv -> out would be a cycle)
        return out, sv

    def uresnet_backward(self, sv: Saved, g_logp: torch.Tensor, grad_ready: Optional[Callable] = None):
        """-> flat fp32 gradient buffer (layout self.grad_offsets)."""
        m = self.model
        dt = sv.dt
        dev = sv.x.device
        self._rebind(sv)
        self._pack_bwd(dt, dev)
        flat, views = self._grad_views(dev)
        G = lambda p: views[id(p)]
        stage_done = self._stage_notifier(flat, grad_ready)
        self._side_begin(dev)
        self._red_begin(dev)

        N, ncls, H, W = sv.out.shape
        ip = m.inplanes
        if not g_logp.is_contiguous():
            g_logp = g_logp.contiguous()
        # ---- head ----
        g_l = self._new((N, H, W, 16), dtype=dt, device=dev)
        self._logsoftmax_bwd(g_logp, sv.out, g_l)
        bn10 = self.bn(m.bn10)
        nk = m.conv10.out_channels
        self._wg(sv.c10, g_l, T7, G(m.conv11.weight), nk * 49, 49, ncls, nk, self.wws, xf=self.relu_affine(bn10))
        NS = L.STAT_SLOTS
        red = self._red(16 + nk, dev)
        ops.channel_sum(g_l, red[:16 * NS])
        ops.cast_f64_to_f32(red[:16 * NS], G(m.conv11.bias), ncls, stride=16)
        g_a10 = self._new((N, H, W, nk), dtype=dt, device=dev)
        # data gradient of conv11: K = the 16 (zero-padded) logit channels
        wp = self._packed_dgrad_padded(m.conv11.weight, dt)
        red10 = self._red(2 * nk, dev) if _BNB_FUSE else None
        ops.conv(g_l, wp, g_a10, DG7, nk, bnb=(sv.c10, bn10.mean, bn10.scale, bn10.shift, bn10.invstd) if _BNB_FUSE else None, stats=red10)
        del g_l
        g_c10 = self._bn_bwd(bn10, g_a10, None, sv.c10, True, G, N * H * W, red=red10)
        del g_a10
        self._wg(sv.d1o, g_c10, T7, G(m.conv10.weight), ip * 49, 49, nk, ip, self.wws)
        ops.channel_sum(g_c10, red[16 * NS:])
        ops.cast_f64_to_f32(red[16 * NS:], G(m.conv10.bias), nk)
        g = self._new(sv.d1o.shape, dtype=dt, device=dev)
        self._conv_dgrad(m.conv10, g_c10, g, 1, k=7)
        del g_c10
        stage_done(m.bn10.bias)
        # ---- decoder ----
        d1, d2, d3, d4, d5 = sv.dec
        g, gc1 = self.declayer_bwd(d1, g, G); stage_done(m.dec_layer1.deconv.weight)
        g, gc2 = self.declayer_bwd(d2, g, G); stage_done(m.dec_layer2.deconv.weight)
        g, gc3 = self.declayer_bwd(d3, g, G); stage_done(m.dec_layer3.deconv.weight)
        g, gc4 = self.declayer_bwd(d4, g, G); stage_done(m.dec_layer4.deconv.weight)
        g, gc5 = self.declayer_bwd(d5, g, G); stage_done(m.dec_layer5.deconv.weight)
        # ---- encoder (skip gradients arrive through the concat buffers' second halves) ----
        e1, e2, e3, e4, e5 = sv.enc
        g = self.double_bwd(e5, g, None, G); stage_done(m.enc_layer5.res1.conv1.weight)
        g = self.double_bwd(e4, g, gc5[..., 16 * ip:], G); stage_done(m.enc_layer4.res1.conv1.weight)
        g = self.double_bwd(e3, g, gc4[..., 8 * ip:], G); stage_done(m.enc_layer3.res1.conv1.weight)
        g = self.double_bwd(e2, g, gc3[..., 4 * ip:], G); stage_done(m.enc_layer2.res1.conv1.weight)
        g = self.double_bwd(e1, g, gc2[..., 2 * ip:], G); stage_done(m.enc_layer1.res1.conv1.weight)
        # ---- stem ----
        bn1 = self.bn(m.bn1)
        g_x0 = self._new(sv.c0.shape, dtype=dt, device=dev)
        ops.maxpool_bwd(sv.c0, self.relu_affine(bn1), g, gc1[..., ip:], g_x0, 2, argmax=sv.amax)
        g_c0 = self._bn_bwd(bn1, g_x0, None, sv.c0, True, G, N * H * W)
        self.stem_bwd(m.conv1, sv.x16, g_c0, G)
        stage_done(self.grad_order[-1][1])
        self._side_end(dev)
        self._red_buf = None
        return flat, views

    # ------------------------------------------------------------------ inference schedule (UResNet, eval mode)
    # SURVEY.md section 8d, k = 1: eval-mode BatchNorm is a fixed per-channel affine, so it is folded into the packed
    # weights (scale) and the conv bias; ReLU and the residual add run in the conv epilogue (ubr_conv_desc.act).  Every
    # tensor is written once, activated, and read by its consumers with no transform; block tails, BatchNorm finalize
    # launches and the raw conv2 / bypass outputs of the training schedule disappear (201 M instead of 250 M elements
    # per 512x512 image).  Reference call sites: deploy/run_ubresnet_precropped.py:88-89,147 (model.eval(); forward).
    def _infer_plan(self, dt, device):
        import struct
        m = self.model
        key = ("inf", dt, device)
        ptrs = tuple(p.data_ptr() for _, p in self.grad_order) + tuple(b.data_ptr() for b in m.buffers())
        plan = self._plans.get(key)
        if plan is not None and plan["ptrs"] == ptrs:
            return plan
        cpu = L.chans_per_unit(dt)
        pairs = [(m.conv1, m.bn1), (m.conv10, m.bn10)]
        for mod in m.modules():
            if hasattr(mod, "bn2") and hasattr(mod, "conv2"):          # BasicBlock
                pairs += [(mod.conv1, mod.bn1), (mod.conv2, mod.bn2)]
                if mod.bypass is not None:
                    pairs.append((mod.bypass, mod.bnpass))
        total = sum(bn.num_features for _, bn in pairs)
        vec = self._new(2 * total, dtype=torch.float32, device=device)
        scale_of, bias_of, fold_tbl, off = {}, {}, b"", 0
        for conv, bn in pairs:
            Cn = bn.num_features
            sc, bi = vec[off:off + Cn], vec[total + off:total + off + Cn]
            off += Cn
            scale_of[id(conv.weight)], bias_of[id(bn)] = sc, bi
            for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var):
                if t is None or t.dtype != torch.float32 or t.device != device:
                    raise RuntimeError("ubresnet_amd: inference needs affine BatchNorm2d with float32 running statistics on %s" % device)
            fold_tbl += struct.pack("<QQQQQQQif", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                    conv.bias.data_ptr() if conv.bias is not None else 0, sc.data_ptr(), bi.data_ptr(), Cn, float(bn.eps))
        images, pack_tbl, n = {}, b"", 0
        for group, k, w, soff, M, Kv, Kpad, sm, sk, ntaps, tstride in self._plan_items():
            if group != "fwd":
                continue
            if not w.is_contiguous() or w.dtype != torch.float32 or w.device != device:
                raise RuntimeError("ubresnet_amd: parameters must be contiguous float32 on %s" % device)
            Mpad = (M + 15) // 16 * 16
            Kp = Kpad if Kpad is not None else (Kv + cpu - 1) // cpu * cpu
            dst = self._new((ntaps, Kp // cpu, Mpad, cpu), dtype=dt, device=device)
            images[k] = dst
            sc = scale_of.get(id(w))
            pack_tbl += struct.pack("<QQqqqQiiiiii", w.data_ptr() + 4 * soff, dst.data_ptr(), sm, sk, tstride,
                                    sc.data_ptr() if sc is not None else 0, M, Mpad, Kv, Kp // cpu, ntaps, 0)
            n += 1
        dev_tbl = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(device)
        plan = {"ptrs": ptrs, "images": images, "bias": bias_of, "vec": vec, "fold": dev_tbl(fold_tbl), "nfold": len(pairs),
                "pack": dev_tbl(pack_tbl), "npack": n}
        self._plans[key] = plan
        return plan

    def _block_infer(self, blk, x, out, img, fb, dt):
        """BasicBlock.forward (models/common_layers.py:39-58) with folded BatchNorms: 2 launches (3 with a bypass conv)"""
        N, OH, OW, Cout = out.shape
        S = blk.stride
        c1 = self._new((N, OH, OW, Cout), dtype=dt, device=x.device)
        ops.conv(x, img[(id(blk.conv1.weight), "fwd")], c1, T3, Cout, S=S, bias=fb[id(blk.bn1)], act=1)
        sc = x
        if blk.bypass is not None:
            sc = self._new((N, OH, OW, Cout), dtype=dt, device=x.device)
            ops.conv(x, img[(id(blk.bypass.weight), "fwd")], sc, T1, Cout, S=S, bias=fb[id(blk.bnpass)])
        ops.conv(c1, img[(id(blk.conv2.weight), "fwd")], out, T3, Cout, bias=fb[id(blk.bn2)], addend=sc, act=3)

    def _double_infer(self, dbl, x, out, img, fb, dt):
        mid = self._new(out.shape, dtype=dt, device=x.device)
        self._block_infer(dbl.res1, x, mid, img, fb, dt)
        self._block_infer(dbl.res2, mid, out, img, fb, dt)

    def _declayer_infer(self, dl, x, cat, Cd, out, img, fb, dt):
        wp = img[(id(dl.deconv.weight), "tfwd")]
        up = cat[..., :Cd]
        if Cd >= _PHASE_MIN_C:
            ops.conv_phases(x, wp, _phase(up, 0, 0), _PH4[1], Cd, phases=_PH4[0], y_full=up)
        else:
            for ry in range(2):
                for rx in range(2):
                    ops.conv(x, wp, _phase(up, ry, rx), ops.transposed_phase_taps(4, 1, 1, 2, ry, rx), Cd)
        self._double_infer(dl.res, cat, out, img, fb, dt)

    def uresnet_infer(self, x, dt):
        """eval-mode UResNet.forward (models/ub_uresnet.py:88-147), nothing saved"""
        m = self.model
        x = self._check_input(x, m.conv1.in_channels)
        N, Cin, H, W = x.shape
        dev, ip = x.device, m.inplanes
        plan = self._infer_plan(dt, dev)
        st = L.stream_ptr()
        L.check(L.lib().ubr_bn_fold_batched(plan["fold"].data_ptr(), plan["nfold"], st), "bn_fold_batched")
        L.check(L.lib().ubr_pack_weights_batched(L.dtype_id(dt), plan["pack"].data_ptr(), plan["npack"], st), "pack_weights_batched")
        img, fb = plan["images"], plan["bias"]
        E = lambda *shape: self._new(shape, dtype=dt, device=dev)
        # stem: conv1 (+bn1 folded, ReLU in the epilogue) writes x0 straight into dec1's concat buffer; pool reads it
        cat1 = E(N, H, W, 2 * ip)
        x0 = cat1[..., ip:]
        x16 = E(N, H, W, 16 * Cin)
        self._untaped(lambda: ops.stem_expand(x, x16))
        if self._rec is not None:
            self._rec.pre = lambda xx: ops.stem_expand(xx, x16)
        for ci in range(Cin):
            ops.conv(x16[..., 16 * ci:16 * ci + 16], img[(id(m.conv1.weight), "stem%d" % ci)], x0, self.STEM_TAPS, ip,
                     bias=fb[id(m.bn1)] if ci == 0 else None, addend=x0 if ci > 0 else None, act=2 if ci == Cin - 1 else 0)
        p0 = E(N, H // 2, W // 2, ip)
        ops.maxpool_fwd(x0, None, p0, None, 2)
        cat2 = E(N, H // 2, W // 2, 4 * ip)
        cat3 = E(N, H // 4, W // 4, 8 * ip)
        cat4 = E(N, H // 8, W // 8, 16 * ip)
        cat5 = E(N, H // 16, W // 16, 32 * ip)
        x5 = E(N, H // 32, W // 32, 32 * ip)
        x1, x2, x3, x4 = cat2[..., 2 * ip:], cat3[..., 4 * ip:], cat4[..., 8 * ip:], cat5[..., 16 * ip:]
        self._double_infer(m.enc_layer1, p0, x1, img, fb, dt)
        self._double_infer(m.enc_layer2, x1, x2, img, fb, dt)
        self._double_infer(m.enc_layer3, x2, x3, img, fb, dt)
        self._double_infer(m.enc_layer4, x3, x4, img, fb, dt)
        self._double_infer(m.enc_layer5, x4, x5, img, fb, dt)
        d5o = E(N, H // 16, W // 16, 16 * ip)
        self._declayer_infer(m.dec_layer5, x5, cat5, 16 * ip, d5o, img, fb, dt)
        d4o = E(N, H // 8, W // 8, 8 * ip)
        self._declayer_infer(m.dec_layer4, d5o, cat4, 8 * ip, d4o, img, fb, dt)
        d3o = E(N, H // 4, W // 4, 4 * ip)
        self._declayer_infer(m.dec_layer3, d4o, cat3, 4 * ip, d3o, img, fb, dt)
        d2o = E(N, H // 2, W // 2, 2 * ip)
        self._declayer_infer(m.dec_layer2, d3o, cat2, 2 * ip, d2o, img, fb, dt)
        d1o = E(N, H, W, ip)
        self._declayer_infer(m.dec_layer1, d2o, cat1, ip, d1o, img, fb, dt)
        nk = m.conv10.out_channels
        c10 = E(N, H, W, nk)
        ops.conv(d1o, img[(id(m.conv10.weight), "fwd")], c10, T7, nk, bias=fb[id(m.bn10)], act=1)
        ncls = m.conv11.out_channels
        return self._final_logsoftmax(m, c10, img[(id(m.conv11.weight), "fwd")], None, (N, ncls, H, W))

    # ------------------------------------------------------------------ dispatch
    def forward(self, x, training, dt, save):
        from . import plan
        return plan.forward(self, x, training, dt, save)

    def backward(self, sv, g_out, grad_ready=None, allow_plan=True):
        from . import plan
        return plan.backward(self, sv, g_out, grad_ready, allow_plan)

    def forward_eager(self, x, training, dt, save):
        if self.kind == "uresnet":
            if not training and not save and _INFER_FOLD:
                return self.uresnet_infer(x, dt), None
            return self.uresnet_forward(x, training, dt, save)
        if self.kind == "aspp":
            return self.aspp_forward(x, training, dt, save)
        raise RuntimeError("ubresnet_amd: unknown network kind %r" % self.kind)

    def backward_eager(self, sv, g_out, grad_ready=None):
        if self.kind == "uresnet":
            return self.uresnet_backward(sv, g_out, grad_ready)
        if self.kind == "aspp":
            return self.aspp_backward(sv, g_out, grad_ready)
        raise RuntimeError("ubresnet_amd: unknown network kind %r" % self.kind)
