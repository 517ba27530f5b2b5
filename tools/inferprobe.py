#!/usr/bin/env python3
"""The whole-view inference leg of bench.py alone (BASELINE configs[4]); run under rocprofv3 for its per-kernel statistics:
   rocprofv3 --kernel-trace --stats -d out -- python3 tools/inferprobe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

torch.cuda.set_device(0)
r = bench.infer_leg(events=int(sys.argv[1]) if len(sys.argv) > 1 else 6)
print(json.dumps({k: r[k] for k in ("metric", "value", "unit", "ms_per_event", "kernel_time_ms_per_event", "roofline_dominant_kernel")}))
