#!/usr/bin/env python3
"""Print the headline numbers of bench.py's JSON line read from stdin (A/B runs on a GPU box)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        r = d.get("roofline") or {}
        print("%.3f ms/step  %.1f %s | dominant %s %.1f us frac %.3f" % (d["ms_per_step"], d["value"], d["unit"], r.get("kernel"), r.get("avg_launch_us") or 0.0, r.get("frac") or 0.0))
