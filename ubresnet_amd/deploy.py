"""Forward-only inference helpers: checkpoint loading and whole-view tiled segmentation.

Mirrors the inference side of the reference:
  * ``load_cosmic_retrain_model`` (deploy/ubresnet_funcs.py:41-68): build ``UResNet(inplanes=16,
    input_channels=1, num_classes=4)``, ``torch.load`` the checkpoint with a ``map_location``,
    strip the ``module.`` prefix a DataParallel checkpoint carries, ``load_state_dict``.
  * the pre-cropped loop (deploy/run_ubresnet_precropped.py:115-182): ``model.eval()`` forward per
    batch -> ``segment_crops``.
  * the whole-view loop (deploy/run_ubresnet_wholeview.py:191-277, a larflow script in the
    reference; only its shape is reusable): slice (bs,1,512,832) crops out of [3,1,rows,cols]
    plane images, run the model, stitch -> ``WholeViewSegmenter``.  The reference obtains crop boxes
    from larcv's UBSplitDetector (absent C++); here the tiling is regular with overlap
    (rows {0,496}, cols {0,656,1312,1968,2624} for a 1008 x 3456 view).
    Crop and stitch are HIP kernels; the per-batch forward is captured once in a hipGraph and
    replayed (fixed shapes), tiles are independent so multi-GPU inference is replicas only.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L


def load_model(checkpointfile: Optional[str], device, num_classes: int = 4, inplanes: int = 16, input_channels: int = 1,
               map_location=None, state_dict=None):
    """UResNet for deployment (deploy/ubresnet_funcs.py:41-68).  `checkpointfile` is the reference's
    ``{iter, epoch, state_dict, best_prec1, optimizer}`` tar; tensors only are read (weights_only)."""
    from .models.ub_uresnet import UResNet
    model = UResNet(inplanes=inplanes, input_channels=input_channels, num_classes=num_classes, showsizes=False)
    if state_dict is None and checkpointfile is not None:
        ckpt = torch.load(checkpointfile, map_location=map_location or "cpu", weights_only=True)
        state_dict = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    if state_dict is not None:
        clean = {}
        for k, v in state_dict.items():
            clean[k[len("module."):] if k.startswith("module.") else k] = v
        model.load_state_dict(clean)
    model = model.to(device=torch.device(device))
    model.eval()
    return model


def save_checkpoint(state: dict, is_best: bool, p: int, filename: str = "checkpoint.pth.tar"):
    """save_checkpoint (training/train_ubresnet2018_wlarcv2.py:474-479): same files, same dict layout"""
    import shutil
    if p > 0:
        filename = "checkpoint.%dth.tar" % p
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, "model_best.tar")
    return filename


@torch.no_grad()
def segment_crops(model, adc: torch.Tensor, batch: int = 4) -> torch.Tensor:
    """eval forward over pre-cropped images [n,C,H,W] in batches (deploy/run_ubresnet_precropped.py:115-182)"""
    model.eval()
    outs = []
    for i in range(0, adc.shape[0], batch):
        outs.append(model(adc[i:i + batch]))
    return torch.cat(outs, 0)


def regular_tiling(rows: int, cols: int, th: int = 512, tw: int = 832) -> Tuple[List[int], List[int]]:
    """row/col origins of the fewest tiles covering rows x cols with the overlap spread evenly"""
    def origins(n, t):
        if n <= t:
            return [0]
        k = -(-n // t)                       # number of tiles
        step = (n - t) / float(k - 1)
        return sorted(set(int(round(i * step)) for i in range(k)))
    return origins(rows, th), origins(cols, tw)


def _keep_windows(origins: Sequence[int], t: int, n: int):
    """split the overlaps in the middle: tile i keeps [lo_i, hi_i) in view coordinates"""
    out = []
    for i, o in enumerate(origins):
        lo = 0 if i == 0 else (origins[i - 1] + t + o) // 2
        hi = min(n, o + t) if i == len(origins) - 1 else (o + t + origins[i + 1]) // 2
        out.append((lo, hi))
    return out


class WholeViewSegmenter:
    """Tiled whole-view inference: crop -> model (hipGraph replay) -> stitch.

        seg = WholeViewSegmenter(model, rows=1008, cols=3456, planes=3, tile=(512, 832), batch=10,
                                 dtype=torch.float16)
        scores = seg(view)          # view [planes,1,rows,cols] float32 on the GPU -> [planes,C,rows,cols]
    """

    def __init__(self, model, rows: int, cols: int, planes: int = 3, tile=(512, 832), batch: int = 10,
                 dtype: torch.dtype = torch.float16, use_graph: bool = True):
        self.model, self.rows, self.cols, self.planes = model, rows, cols, planes
        self.th, self.tw = tile
        if self.th % 32 or self.tw % 32:
            raise ValueError("tile size must be a multiple of 32")
        self.batch, self.dtype, self.use_graph = batch, dtype, use_graph
        ro, co = regular_tiling(rows, cols, self.th, self.tw)
        rk, ck = _keep_windows(ro, self.th, rows), _keep_windows(co, self.tw, cols)
        self.tiles = []          # (plane, r0, c0, kr0, kr1, kc0, kc1)
        for p in range(planes):
            for (r0, (rl, rh)) in zip(ro, rk):
                for (c0, (cl, ch)) in zip(co, ck):
                    self.tiles.append((p, r0, c0, rl - r0, rh - r0, cl - c0, ch - c0))
        if batch > L.MAX_TAPS:
            raise ValueError("batch must be <= 64 tiles")
        self.nclass = model.conv11.out_channels
        self._graph = None
        self._static_in = None
        self._static_out = None
        self._captured_sig = None

    @property
    def tiles_per_event(self):
        return len(self.tiles)

    def _desc(self, tiles):
        flat = [v for t in tiles for v in t]
        return (C.c_int32 * len(flat))(*flat)

    def _forward_batch(self, x):
        old = getattr(self.model, "compute_dtype", None)
        self.model.compute_dtype = self.dtype
        try:
            return self.model(x)
        finally:
            self.model.compute_dtype = old

    def _signature(self):
        return tuple(t.data_ptr() for t in self.model.parameters()) + tuple(t.data_ptr() for t in self.model.buffers())

    def _ensure_graph(self, device):
        # the captured graph bakes in the device addresses of parameters, buffers and packed weight images: when the model's
        # storage was replaced since the capture (model.to(), a flat optimizer adopting the parameters,
        # load_state_dict(assign=True)) the replay would read freed memory -- capture again
        sig = self._signature()
        if self._static_in is not None and sig != self._captured_sig:
            self._graph = self._static_in = self._static_out = None
        if self._static_in is not None:
            return
        self._captured_sig = sig
        self._static_in = torch.zeros((self.batch, 1, self.th, self.tw), dtype=torch.float32, device=device)
        self.model.eval()
        with torch.no_grad():
            self._forward_batch(self._static_in)            # warm-up: packs weights, raises LDS limits, fills allocator
            torch.cuda.synchronize()
            if self.use_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._static_out = self._forward_batch(self._static_in)
                self._graph = g

    @torch.no_grad()
    def __call__(self, view: torch.Tensor) -> torch.Tensor:
        L.require_cuda(view, "view")
        if view.dtype != torch.float32 or tuple(view.shape) != (self.planes, 1, self.rows, self.cols):
            raise RuntimeError("WholeViewSegmenter: expected float32 [%d,1,%d,%d], got %s %s"
                               % (self.planes, self.rows, self.cols, view.dtype, tuple(view.shape)))
        view = view.contiguous()
        self._ensure_graph(view.device)
        out = torch.empty((self.planes, self.nclass, self.rows, self.cols), dtype=torch.float32, device=view.device)
        lib = L.lib()
        for i in range(0, len(self.tiles), self.batch):
            chunk = self.tiles[i:i + self.batch]
            n = len(chunk)
            desc = self._desc(chunk)
            st = L.stream_ptr()
            L.check(lib.ubr_crop_tiles(view.data_ptr(), self.planes, self.rows, self.cols, desc, n, self.th, self.tw,
                                       self._static_in.data_ptr(), st), "crop_tiles")
            if self._graph is not None:
                self._graph.replay()
                scores = self._static_out
            else:
                scores = self._forward_batch(self._static_in)
            L.check(lib.ubr_stitch_tiles(scores.data_ptr(), self.nclass, self.th, self.tw, desc, n, out.data_ptr(),
                                         self.planes, self.rows, self.cols, L.stream_ptr()), "stitch_tiles")
        return out
