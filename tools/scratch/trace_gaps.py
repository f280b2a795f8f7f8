#!/usr/bin/env python3
"""reads rocprofv3 --kernel-trace / --memory-copy-trace csv files of a run of repeated greedy sweeps and prints, per sweep, the device-side
timeline: first operation, the two k_greedy_sched launches, and the idle gaps longer than 1 ms"""
import csv, glob, sys
d = sys.argv[1]
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ops.sort()
t0 = ops[0][0]
last_end = ops[0][0]
for s, e, name in ops:
    if s - last_end > 1e6 or "greedy_sched" in name:
        print(f"{(s - t0) / 1e6:10.2f} ms  gap before {(s - last_end) / 1e6:7.2f} ms  dur {(e - s) / 1e6:8.2f} ms  {name}")
    last_end = max(last_end, e)
