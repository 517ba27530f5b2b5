"""HBM traffic per kernel launch from two rocprofv3 --pmc passes -> profiles/pmc_traffic.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT/write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown
    python tools/pmc_traffic.py OUT/fetch OUT/write profiles/pmc_traffic.json

Counters are collected in separate passes (one counter each) with no trace domain, as the MI355X guide prescribes;
FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Kernels are keyed by the symbol bench.py uses
(template name without namespace and argument list).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def symbol(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                       # cut the argument list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def collect(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        k = symbol(row["Kernel_Name"])
        tot[k] += float(row["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt)):
        n = max(fc.get(k, 0), wc.get(k, 0))
        f = ft.get(k, 0.0) / max(fc.get(k, 1), 1)
        w = wt.get(k, 0.0) / max(wc.get(k, 1), 1)
        kernels[k] = {"launches_profiled": n, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
                      "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    steps = 4          # bench.py --steps 3 --warmup 1
    per_step = sum(v["launches_profiled"] / steps * v["hbm_bytes_per_launch"] for v in kernels.values())
    from ubresnet_amd.build import source_hash
    doc = {"dtype": "bf16", "batch": 16, "steps_profiled": steps, "hbm_bytes_per_step": per_step,
           "csrc_sha256": source_hash(),      # bench.py reports these figures only while the kernel sources still hash to this
           "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)",
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    step_bytes = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in kernels.values())
    print("wrote %s: %d kernels, %.2f GB over the profiled launches" % (out, len(kernels), step_bytes / 1e9))


if __name__ == "__main__":
    main()
