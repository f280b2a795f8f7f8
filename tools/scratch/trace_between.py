#!/usr/bin/env python3
"""rocprofv3 csv traces of repeated greedy sweeps: every device operation between the end of one sweep's last k_greedy_sched and the start of the next sweep's first"""
import csv, glob, sys
d = sys.argv[1]
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), "-"))
ops.sort()
t0 = ops[0][0]
big = [i for i, o in enumerate(ops) if "greedy_sched" in o[2]]
ends = sorted(set(big))
# sweeps: pairs of big kernels
for a in range(1, len(big) - 1, 2):
    last_end = max(ops[big[a - 1]][1], ops[big[a]][1])
    nxt = big[a + 1]
    print(f"--- sweep ends at {(last_end - t0) / 1e6:.2f} ms; next sweep's first big kernel starts at {(ops[nxt][0] - t0) / 1e6:.2f} ms")
    for s, e, name, q in ops:
        if s >= last_end - 2e6 and s <= ops[nxt][0] and "greedy_sched" not in name:
            print(f"   {(s - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e3:8.1f} us  q {q}  {name}")
