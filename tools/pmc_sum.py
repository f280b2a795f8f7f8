#!/usr/bin/env python3
"""Per-kernel SUMS of rocprofv3 --pmc counters over all dispatches: python3 tools/pmc_sum.py DIR [name-filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void\s+", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name).replace("acx::", "")
        if pat in name:
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name][r["Counter_Name"]] += 1
for k in sorted(acc):
    c = acc[k]
    n = max(calls[k].values())
    print(f"{k}  dispatches={n}")
    print("   " + " ".join(f"{a}={c[a]:.4g}" for a in sorted(c)))
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        print(f"   FETCH_SIZE {c.get('FETCH_SIZE', 0) / 1e6:.3f} GB (x2 for wide coalesced streams on gfx950), WRITE_SIZE {c.get('WRITE_SIZE', 0) / 1e6:.3f} GB")
