"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference's multi-GPU mode is single-process nn.DataParallel
(training/train_ubresnet2018_wlarcv2.py:99,103): per step it broadcasts all parameters,
scatters the batch, gathers the outputs and reduce-adds 165 gradient tensors onto device 0,
with BatchNorm statistics local to each replica.  Here parameters stay replicated, each rank
trains on its own shard of the global batch (local BatchNorm statistics, the same semantics),
and the ONLY exchange is an averaged all-reduce of the flat gradient buffer.  The graph
executor reports contiguous gradient ranges as backward completes them (decoder/head first),
so buckets are reduced on a side stream while the encoder's backward is still running.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradAllReducer:
    """Replicas start identical: construction broadcasts rank 0's parameters and buffers (nn.DataParallel re-broadcasts
    device 0's every forward, training/train_ubresnet2018_wlarcv2.py:99; here once is enough because every rank applies the
    same averaged gradient).  BatchNorm running statistics are NOT exchanged afterwards: each rank tracks its own shard
    (DataParallel semantics: only replica 0's persist) and a checkpoint written by rank 0 holds rank 0's, exactly what the
    reference saves; `average_bn_stats()` makes them rank-independent before a checkpoint when that is wanted.

    Gradient accumulation (several backward passes per optimizer step, or zero_grad(set_to_none=False)): the first pass
    of a step is exchanged bucket by bucket during backward; a pass that finds existing .grad tensors accumulates locally
    and `finish()` all-reduces the accumulated .grad tensors (see autograd_fn._NetFn.backward).

    Bucket size: gradients complete head -> decoder -> encoder -> stem, and the large tensors (enc/dec level 5: 9.4 MB per
    3x3x512x512 convolution) finish in the middle of backward, so every bucket but the last overlaps the remaining
    backward.  The LAST bucket is the exposed one: it can only start when the stem's gradient exists.  With 8 MB buckets
    the 256-channel encoder level (9.4 MB) leaves on its own while levels 3..1 are still running and the tail is the
    ~3 MB of the shallow levels; with 16 MB they waited together for the end of backward (a 13 MB tail)."""

    def __init__(self, model, bucket_bytes: int = 8 << 20, group=None, broadcast_init: bool = True):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        import os
        # UBR_FORCE_REDUCER=1 exercises the bucketed async all-reduce even with one rank (single-GPU rehearsal)
        self.force = os.environ.get("UBR_FORCE_REDUCER", "0") == "1" and dist.is_initialized()
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.stream = None
        self.pending_lo = None
        self.pending_hi = None
        self.flat = None
        self.works = []
        self.wait_events = []
        self.use_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self._accum = False
        self._need_current = True
        model._grad_ready_hook = self._hook
        model._grad_accum_pending = self._begin_accumulate
        if broadcast_init and self.world > 1 and hasattr(model, "parameters"):
            self.broadcast_state()

    def _tensors(self, what):
        return [t for t in (self.model.parameters() if what == "params" else self.model.buffers())]

    def broadcast_state(self, src: int = 0):
        """rank `src`'s parameters and buffers to every rank (one coalesced broadcast per dtype)"""
        from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
        with torch.no_grad():
            by_dtype = {}
            for t in self._tensors("params") + self._tensors("buffers"):
                by_dtype.setdefault(t.dtype, []).append(t.data)
            for ts in by_dtype.values():
                flat = _flatten_dense_tensors(ts)
                dist.broadcast(flat, src, group=self.group)
                for t, v in zip(ts, _unflatten_dense_tensors(flat, ts)):
                    t.copy_(v)

    def average_bn_stats(self):
        """mean over ranks of every floating-point buffer (BatchNorm running_mean / running_var): optional, for
        rank-independent checkpoints; integer buffers (num_batches_tracked) are equal on every rank already"""
        if self.world == 1:
            return
        from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
        with torch.no_grad():
            ts = [b.data for b in self._tensors("buffers") if b.dtype.is_floating_point]
            if not ts:
                return
            flat = _flatten_dense_tensors(ts)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            for t, v in zip(ts, _unflatten_dense_tensors(flat, ts)):
                t.copy_(v)

    def _drain(self):
        """compute stream waits for every bucket in flight; the flat buffer is then final (averaged)"""
        if self.flat is None:
            return
        if self.pending_hi is not None and self.pending_hi > self.pending_lo:
            self._launch(self.pending_lo, self.pending_hi)
            self.pending_lo = self.pending_hi
        for w in self.works:
            w.wait()
        self.works = []
        if self.flat.is_cuda and self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        if not self.use_avg:
            self.flat.div_(self.world)
        self.flat = None

    # called by _NetFn.backward before a pass that will ADD into existing .grad tensors
    def _begin_accumulate(self):
        if self.world == 1 and not self.force:
            return
        self._drain()          # an earlier pass's buckets must land before .grad (their alias) is added to
        self._accum = True

    # called by the executor on the compute stream: flat[lo:hi] is final once the compute stream AND every event in
    # wait_events (work the executor queued on its other streams) have been reached
    def _hook(self, flat: torch.Tensor, lo: int, hi: int, wait_events=(), ordered=True):
        """ordered=False (replayed launch plans): the whole backward is already queued when the ranges are reported, so
        the exchange must wait ONLY for `wait_events` (tape marks inside the replayed streams), never for the compute
        stream's current position -- that would serialise the all-reduce behind the entire backward pass."""
        if self.world == 1 and not self.force:
            return
        if self.flat is not flat or lo == 0:
            self.flat, self.pending_lo, self.pending_hi = flat, lo, lo
            self.wait_events = []
            self._need_current = False
        self._need_current = self._need_current or ordered
        self.wait_events.extend(wait_events)
        self.pending_hi = hi
        last = hi >= flat.numel()
        if self.pending_hi - self.pending_lo >= self.bucket_elems or last:
            self._launch(self.pending_lo, self.pending_hi)
            self.pending_lo = self.pending_hi

    def _launch(self, lo: int, hi: int):
        if hi <= lo:
            return
        chunk = self.flat[lo:hi]
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        if chunk.is_cuda:
            if self.stream is None:
                # high priority: its own hardware queue (a normal-priority stream can alias the compute stream's queue
                # and the all-reduce would then serialise with the backward kernels instead of overlapping them)
                self.stream = torch.cuda.Stream(device=chunk.device, priority=-1)
            if self._need_current:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(chunk.device))
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                for e in self.wait_events:
                    if hasattr(e, "wait_on"):
                        e.wait_on(self.stream)          # ubresnet_amd.plan.TapeMark
                    else:
                        self.stream.wait_event(e)
                self.wait_events = []
                self.works.append(dist.all_reduce(chunk, op=op, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(chunk, op=op, group=self.group, async_op=True))

    def finish(self):
        """make the compute stream wait for every outstanding bucket (call before optimizer.step)"""
        if self.world == 1 and not self.force:
            return
        self._drain()
        if self._accum:
            # accumulated passes were not exchanged during backward: reduce the .grad tensors themselves
            from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
            self._accum = False
            gs = [p.grad for p in self._tensors("params") if p.grad is not None]
            if gs:
                flat = _flatten_dense_tensors(gs)
                dist.all_reduce(flat, op=dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM, group=self.group)
                if not self.use_avg:
                    flat.div_(self.world)
                for g, v in zip(gs, _unflatten_dense_tensors(flat, gs)):
                    g.copy_(v)


def shard_range(global_batch: int, rank: int, world: int):
    """rank r takes images [r*b, (r+1)*b) of the global batch (equal shards => mean of rank means
    equals the global-mean loss PixelWiseNLLLoss computes, training/pixelwise_nllloss.py:59)"""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    b = global_batch // world
    return rank * b, (rank + 1) * b
