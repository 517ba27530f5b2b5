"""Synthetic LArTPC wire-plane crops behind the ``larcvdataset`` batch interface.

The reference reads MicroBooNE crops through ``larcvdataset.LArCVDataset`` (an absent
git submodule; call sites training/train_ubresnet2018_wlarcv2.py:164-171,205,597) and
``LArCV1Dataset.getbatch`` (training/larcv1_interface.py:36-66).  Neither ROOT nor LArCV
exists here, so this module produces seeded synthetic crops with the same array contract:

    loader = SyntheticLArCVDataset(height, width, ...); loader.start(batchsize)
    data = loader[0]      # {"source_train": flat float32, "label_train": flat float32,
                          #  "weight_train": flat float32}   (flat, length B*H*W)
    loader.stop()

so ``prep_data`` (training/train_ubresnet2018_wlarcv2.py:576-615) runs unchanged.

Crop model: background 0; 2-8 straight "tracks" (width 1-3 px,
ADC ~ U(20,120) + noise) -> label 1; 1-4 cone-shaped "showers" of scattered hits
(ADC ~ Exp(40)) -> label 2; about 1-3 % non-zero pixels; weight 1 on background and a
class-balancing constant on labelled pixels.  Image i of a stream uses
``numpy.random.RandomState(seed0 + i)`` so any consumer can regenerate it bit for bit.
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_crop", "make_batch", "SyntheticLArCVDataset"]


def make_crop(height: int, width: int, seed: int, labelled_weight: float = 10.0):
    """One crop -> (adc float32 [H,W], label int64 [H,W], weight float32 [H,W])."""
    rs = np.random.RandomState(seed)
    adc = np.zeros((height, width), dtype=np.float32)
    lab = np.zeros((height, width), dtype=np.int64)
    scale = (height * width) / float(512 * 512)

    # tracks: straight segments, width 1-3 px
    for _ in range(rs.randint(2, 9)):
        y0, y1 = rs.uniform(0, height, 2)
        x0, x1 = rs.uniform(0, width, 2)
        n = int(max(abs(y1 - y0), abs(x1 - x0))) + 1
        ts = np.linspace(0.0, 1.0, n)
        ys = y0 + (y1 - y0) * ts
        xs = x0 + (x1 - x0) * ts
        wpx = rs.randint(1, 4)
        base = rs.uniform(20.0, 120.0)
        for dw in range(wpx):
            yy = np.clip(np.round(ys).astype(np.int64), 0, height - 1)
            xx = np.clip(np.round(xs).astype(np.int64) + dw, 0, width - 1)
            vals = base + rs.normal(0.0, 5.0, n)
            adc[yy, xx] = np.maximum(vals, 1.0).astype(np.float32)
            lab[yy, xx] = 1

    # showers: cone-shaped scatter of hits
    for _ in range(rs.randint(1, 5)):
        cy, cx = rs.uniform(0, height), rs.uniform(0, width)
        ang = rs.uniform(0, 2 * np.pi)
        nhits = int(rs.randint(2000, 5001) * scale) + 8
        r = rs.exponential(60.0 * np.sqrt(scale) + 4.0, nhits)
        spread = rs.normal(0.0, 0.18, nhits)
        yy = np.round(cy + r * np.sin(ang + spread)).astype(np.int64)
        xx = np.round(cx + r * np.cos(ang + spread)).astype(np.int64)
        ok = (yy >= 0) & (yy < height) & (xx >= 0) & (xx < width)
        yy, xx = yy[ok], xx[ok]
        vals = rs.exponential(40.0, yy.shape[0]).astype(np.float32) + 1.0
        adc[yy, xx] = vals
        lab[yy, xx] = 2

    wgt = np.where(lab > 0, np.float32(labelled_weight), np.float32(1.0)).astype(np.float32)
    return adc, lab, wgt


def make_batch(batchsize: int, height: int, width: int, seed0: int = 1000, planes: int = 1):
    """-> (adc [B,planes,H,W] float32, label [B,H,W] int64, weight [B,H,W] float32).

    With planes>1 (ASPP_ResNet's three stacked wire planes) plane p of image i is crop
    ``seed0 + i*planes + p`` and the label/weight come from the LAST plane (the collection
    plane convention of the reference cfgs, training/ubresnet_train.cfg:14-27).
    """
    adc = np.zeros((batchsize, planes, height, width), dtype=np.float32)
    lab = np.zeros((batchsize, height, width), dtype=np.int64)
    wgt = np.zeros((batchsize, height, width), dtype=np.float32)
    for i in range(batchsize):
        for p in range(planes):
            a, l, w = make_crop(height, width, seed0 + i * planes + p)
            adc[i, p] = a
        lab[i], wgt[i] = l, w
    return adc, lab, wgt


class SyntheticLArCVDataset(object):
    """Drop-in for ``larcvdataset.LArCVDataset`` (absent submodule) over synthetic crops.

    Same surface the train script touches: ``start(batchsize)``, ``stop()``, ``len()``,
    ``loader[0]`` -> dict of flat float32 arrays named ``source_<tag>``, ``label_<tag>``,
    ``weight_<tag>`` (names from training/ubresnet_train.cfg:9-27; the label is float32 on
    the wire and cast by the caller, train_ubresnet2018_wlarcv2.py:601).
    """

    def __init__(self, height=512, width=512, tag="train", nentries=1000, seed0=1000,
                 planes=1, cache=8):
        self.height, self.width, self.tag = int(height), int(width), tag
        self.nentries, self.seed0, self.planes = int(nentries), int(seed0), int(planes)
        self.batchsize = None
        self._cursor = 0
        self._cache = {}
        self._cache_max = int(cache)

    def start(self, batchsize):
        self.batchsize = int(batchsize)
        self._cursor = 0

    def stop(self):
        self.batchsize = None

    def __len__(self):
        return self.nentries

    def _entry(self, i):
        i = i % self.nentries
        if i not in self._cache:
            if len(self._cache) >= self._cache_max:
                self._cache.pop(next(iter(self._cache)))
            adc, lab, wgt = make_batch(1, self.height, self.width, self.seed0 + i * self.planes, self.planes)
            self._cache[i] = (adc[0], lab[0], wgt[0])
        return self._cache[i]

    def __getitem__(self, idx):
        if self.batchsize is None:
            raise RuntimeError("SyntheticLArCVDataset: call start(batchsize) first")
        b = self.batchsize
        adc = np.empty((b, self.planes, self.height, self.width), dtype=np.float32)
        lab = np.empty((b, self.height, self.width), dtype=np.float32)
        wgt = np.empty((b, self.height, self.width), dtype=np.float32)
        for j in range(b):
            a, l, w = self._entry(self._cursor + j)
            adc[j], lab[j], wgt[j] = a, l.astype(np.float32), w
        self._cursor += b
        return {"source_%s" % self.tag: adc.reshape(-1),
                "label_%s" % self.tag: lab.reshape(-1),
                "weight_%s" % self.tag: wgt.reshape(-1)}


class DeviceStager(object):
    """Double-buffered host->device staging of loader batches.

    Wraps any object with the larcvdataset interface (``loader[0] -> {name: flat float32 ndarray}``): while the
    training loop consumes batch i on the compute stream, batch i+1 is copied from pinned host memory on a side
    stream.  ``next()`` returns ``(adc [B,planes,H,W] float32, label [B,H,W] int64, weight [B,H,W] float32)``
    device tensors, i.e. what ``prep_data`` (training/train_ubresnet2018_wlarcv2.py:576-615) builds, already on
    the GPU (the reference discards its ``.to(device)`` results, :611-613).
    """

    def __init__(self, loader, batchsize, height, width, planes=1, tag="train", device="cuda"):
        import torch
        self.torch = torch
        self.loader, self.tag = loader, tag
        self.shape = (int(batchsize), int(planes), int(height), int(width))
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        b, p, h, w = self.shape
        self._pin = [(torch.empty((b, p, h, w), dtype=torch.float32).pin_memory(),
                      torch.empty((b, h, w), dtype=torch.int64).pin_memory(),
                      torch.empty((b, h, w), dtype=torch.float32).pin_memory()) for _ in range(2)]
        self._slot = 0
        self._inflight = None
        self._copied = [None, None]          # per pinned slot: event after its last host->device copy
        self._prefetch()

    def _prefetch(self):
        torch = self.torch
        b, p, h, w = self.shape
        data = self.loader[0]
        if self._copied[self._slot] is not None:
            self._copied[self._slot].synchronize()      # the async copy that last read this pinned slot must be done before the host refills it
        src, lab, wgt = self._pin[self._slot]
        src.copy_(torch.from_numpy(data["source_%s" % self.tag].reshape((b, p, h, w))))
        lab.copy_(torch.from_numpy(data["label_%s" % self.tag].reshape((b, h, w))))        # float32 -> int64
        key = "weight_%s" % self.tag
        if key in data:
            wgt.copy_(torch.from_numpy(data[key].reshape((b, h, w))))
        else:
            wgt.fill_(1.0)
        with torch.cuda.stream(self.stream):
            dev = tuple(t.to(self.device, non_blocking=True) for t in (src, lab, wgt))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._inflight = (dev, ev)
        self._copied[self._slot] = ev
        self._slot ^= 1

    def next(self):
        dev, ev = self._inflight
        self.torch.cuda.current_stream(self.device).wait_event(ev)
        for t in dev:
            t.record_stream(self.torch.cuda.current_stream(self.device))
        self._prefetch()
        return dev
