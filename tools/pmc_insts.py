#!/usr/bin/env python3
"""Per-kernel instruction totals from one rocprofv3 --pmc pass (counter_collection.csv) -> table sorted by total.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d OUT -o i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-infer
    python tools/pmc_insts.py OUT [steps_profiled]
"""
import collections
import csv
import glob
import sys

from pmc_traffic import symbol

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
seen = set()
for row in csv.DictReader(open(files[0])):
    k = symbol(row["Kernel_Name"])
    tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key)
        cnt[k] += 1
names = sorted({c for v in tot.values() for c in v})
print("%-64s %6s " % ("kernel", "n/step") + " ".join("%14s" % (n + "/step") for n in names))
for k, v in sorted(tot.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-64s %6.1f " % (k[:64], cnt[k] / steps) + " ".join("%14.3e" % (v.get(n, 0.0) / steps) for n in names))
print("%-64s %6s " % ("TOTAL", "") + " ".join("%14.3e" % (sum(v.get(n, 0.0) for v in tot.values()) / steps) for n in names))
