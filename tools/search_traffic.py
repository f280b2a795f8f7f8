#!/usr/bin/env python3
"""profiles/<tag>_*_pmc_summary.txt (tools/pmc_sum.py) -> profiles/search_traffic.json: HBM bytes of each search workload by the
counters, per search / per sweep.  fetch = max(FETCH_SIZE, TCC_MISS_sum x 64 B) per kernel (FETCH_SIZE counts 128-B read requests
at 64 B on gfx950; random 32-B probes show up in the miss count), + WRITE_SIZE; table fills (fillBufferAligned) listed apart.
    python3 tools/search_traffic.py [tag[,older tag ...]]       (children counts: the bench's own, see KEYS; a workload without a summary of
the first tag takes the next one's -- round 6 re-profiled the kernels it changed: `r6,r5`)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = (sys.argv[1] if len(sys.argv) > 1 else "r6,r5").split(",")
# key in search_traffic.json -> (summary file, kernel-name filter, searches the profiled command ran, children per search / sweep, what it was)
KEYS = {
    "bfs_ak3_1e8": ("bfs_1e8", ("k_bfs_", "k_decide_tab"), 1, 332140812, "tools/bfs_only.py 1e8 (a 2e4-node warm-up + ONE 1e8-node search)"),
    "bfs_sharded_ak3_1e8": ("shard_1e8", ("k_shard_",), 4, 332140812, "tools/shard_bench.py 1e8 21 (four searches at world 1: sums / 4)"),
    "greedy_ak3_1e7": ("greedy_1e7", ("k_greedy_", "k_gm_"), 2, 14174304, "tools/greedy_only.py 1e7 2 (two searches: sums / 2)"),
    "bfs_ms_sweep_1e6": ("ms_sweep_bfs", ("k_bfs_", "k_decide_tab", "k_paths"), 1, 5359092624, "ONE sweep (tools/ms_sweep.py bfs 1e6 16 1 together)"),
    "greedy_ms_sweep_1e6": ("ms_sweep_greedy", ("k_greedy_",), 1, 1073045016, "ONE sweep (tools/ms_sweep.py greedy 1e6 16 0 together)"),
}
out = {}
for key, (name, pats, runs, children, what) in KEYS.items():
    tag = next((t for t in tags if os.path.exists(os.path.join(ROOT, "profiles", f"{t}_{name}_pmc_summary.txt"))), None)
    if tag is None:
        continue
    path = os.path.join(ROOT, "profiles", f"{tag}_{name}_pmc_summary.txt")
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
        elif cur and "=" in line and not line.strip().startswith("FETCH_SIZE "):
            for k, v in re.findall(r"(\w+)=([0-9.e+\-]+)", line):
                kernels[cur][k] = float(v)
    fetch_raw = fetch_miss = write = fills = atomics = valu = 0.0
    parts = []
    for kn, c in sorted(kernels.items()):
        if "fillBuffer" in kn:
            fills += c.get("WRITE_SIZE", 0) * 1024
            continue
        if not any(p in kn for p in pats):
            continue
        fr, fm, wr = c.get("FETCH_SIZE", 0) * 1024, c.get("TCC_MISS_sum", 0) * 64, c.get("WRITE_SIZE", 0) * 1024
        fetch_raw += fr
        fetch_miss += max(fr, fm)
        write += wr
        atomics += c.get("TCC_EA0_ATOMIC_sum", 0)
        valu += c.get("SQ_INSTS_VALU", 0)
        parts.append(f"{kn.split('<')[0]}{'<u128>' if '__int128' in kn else ''} {max(fr, fm) / runs / 1e9:.2f} + {wr / runs / 1e9:.2f} GB")
    out[key] = {"children": children, "hbm_bytes": round((fetch_miss + write) / runs), "fetch_bytes_raw_FETCH_SIZE": round(fetch_raw / runs),
                "fetch_bytes_max_FETCH_TCC_MISS_x64": round(fetch_miss / runs), "write_bytes_WRITE_SIZE": round(write / runs),
                "table_fill_bytes_not_included": round(fills / runs), "memory_side_atomics": round(atomics / runs), "valu_wave_instructions": round(valu / runs),
                "source": f"profiles/{tag}_{name}_pmc_summary.txt: rocprofv3 --pmc passes (tools/pmc_passes.sh, one counter group per run) over {what}; per kernel fetch + write: "
                          + "; ".join(parts) + "; not measured in the bench run itself"}
json.dump(out, open(os.path.join(ROOT, "profiles", "search_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k}: {v['hbm_bytes'] / 1e9:.2f} GB (+ {v['table_fill_bytes_not_included'] / 1e9:.2f} GB of fills), {v['memory_side_atomics']:.3g} atomics, {v['valu_wave_instructions']:.3g} VALU wave-instructions")
