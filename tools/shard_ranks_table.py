#!/usr/bin/env python3
"""Table of tools/profile_shard_ranks.sh: per kernel of the sharded engine, the device time of ALL ranks of one search at every world
size (rocprofv3 --kernel-trace --stats of thread ranks on one GPU, kernels serialised), and the per-rank share.
    python3 tools/shard_ranks_table.py DIR SEARCHES W [W ...]"""
import csv
import glob
import os
import re
import sys

root, reps, worlds = sys.argv[1], int(sys.argv[2]), [int(w) for w in sys.argv[3:]]
ROWS = [("k_shard_expand", "k_shard_expand"), ("k_shard_insert", "k_shard_insert"), ("k_shard_commit (records)", "k_shard_commit<"), ("k_shard_commit_born", "k_shard_commit_born"),
        ("k_shard_pack", "k_shard_pack"), ("k_shard_scan", "k_shard_scan"), ("k_shard_decide + k_shard_prep", ("k_shard_decide", "k_shard_prep")),
        ("table fills", "fillBuffer")]
COMM = "ThreadComm collectives (torch copy / cat / reduce kernels: RCCL on real ranks)"
table, calls, other = {}, {}, {}
for w in worlds:
    files = glob.glob(os.path.join(root, f"w{w}", "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        continue
    per = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            per[row["Name"]] = per.get(row["Name"], (0, 0))
            per[row["Name"]] = (per[row["Name"]][0] + float(row["TotalDurationNs"]), per[row["Name"]][1] + int(row["Calls"]))
    used = set()
    for label, pats in ROWS:
        pats = (pats,) if isinstance(pats, str) else pats
        t = c = 0
        for name, (ns, n) in per.items():
            short = re.sub(r"^void acx::", "", name)
            if any((short.startswith(p) if p.startswith("k_") else p in name) for p in pats) and name not in used:
                if label == "k_shard_commit (records)" and "commit_born" in name:
                    continue
                used.add(name)
                t += ns
                c += n
        table[(label, w)] = t / reps / 1e6
        calls[(label, w)] = c // reps
    other[w] = sum(ns for name, (ns, n) in per.items() if name not in used) / reps / 1e6
print(f"# Device work of ALL ranks of one sharded bfs (AK(3), L = 25, 2^{os.environ.get('CHUNK_LOG2', '21')} parents per chunk) at world W, measured on ONE GPU: W thread ranks")
print("# share the device (tests/shard_helpers.py: ThreadComm; the all-to-all is a device copy per pair), kernels serialised (AMD_SERIALIZE_KERNEL=3)")
print("# so that a kernel's duration is its exclusive time, `rocprofv3 --kernel-trace --stats`, totals / searches.  tools/profile_shard_ranks.sh.")
print("# Not a scaling measurement (a box has one GPU): it says what a rank of a real W-GPU run has to do = total / W (+ the exchange over xGMI).")
print("#")
print("# ms per search, all ranks together (calls per search)")
print(f"{'kernel':34s}" + "".join(f"{'W = ' + str(w):>18s}" for w in worlds))
tot = {w: 0.0 for w in worlds}
tot_nc = {w: 0.0 for w in worlds}
for label, _ in ROWS:
    print(f"{label:34s}" + "".join(f"{table.get((label, w), 0):11.2f} ({calls.get((label, w), 0):4d})" for w in worlds))
    for w in worlds:
        tot[w] += table.get((label, w), 0)
print(f"{'engine total':34s}" + "".join(f"{tot[w]:11.2f}       " for w in worlds))
print(f"{'engine per rank (total / W)':34s}" + "".join(f"{tot[w] / w:11.2f}       " for w in worlds))
print(f"{'(everything else on the device)':34s}" + "".join(f"{other.get(w, 0):11.2f}       " for w in worlds) + "  " + COMM)
