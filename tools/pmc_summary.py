#!/usr/bin/env python3
"""Average per-dispatch PMC values for kernels whose name contains a pattern, from rocprofv3 --pmc CSV output."""
import csv
import glob
import sys
from collections import defaultdict

path, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v):4d} mean={sum(v) / len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
