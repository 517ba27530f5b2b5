#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[4]): whole-view tiled inference, 3456x1008 views cut into
30 tiles of 512x832 per event, fp16 forward-only, hipGraph-captured, 1 GPU.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=30)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    from ubresnet_amd import deploy, synthetic
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
    torch.manual_seed(7)
    m = deploy.load_model(None, "cuda:0", num_classes=4)
    rows, cols = 1008, 3456
    adc = np.zeros((3, 1, rows, cols), np.float32)
    for p in range(3):                                   # synthetic view: the crop generator at full-view size
        adc[p, 0] = synthetic.make_crop(rows, cols, 5000 + p)[0]
    view = torch.from_numpy(adc).cuda()
    seg = deploy.WholeViewSegmenter(m, rows, cols, planes=3, tile=(512, 832), batch=a.batch, dtype=dt, use_graph=not a.no_graph)
    for _ in range(a.warmup):
        seg(view)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.events):
        out = seg(view)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ntiles = seg.tiles_per_event
    gb_tile = 0.654          # algorithmic bytes per 512x832 tile, 2-byte activations
    print(json.dumps({"metric": "tiles/sec, whole-view 3456x1008 tiled inference (512x832 tiles, forward only)",
                      "value": ntiles * a.events / el, "unit": "tiles/sec", "events_per_sec": a.events / el,
                      "ms_per_event": 1e3 * el / a.events, "tiles_per_event": ntiles, "dtype": a.dtype,
                      "hipgraph": not a.no_graph, "n_gpus": 1,
                      "step_model": {"algorithmic_GB_per_tile": gb_tile,
                                     "achieved_GBs": gb_tile * ntiles * a.events / el * (2 if a.dtype == "f32" else 1),
                                     "frac_of_hbm_peak": gb_tile * ntiles * a.events / el / 8000.0}}))


if __name__ == "__main__":
    main()
