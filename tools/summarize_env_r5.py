#!/usr/bin/env python3
"""gpurun_out/r5env/* (tools/profile_env_r5.sh) -> one JSON: at 65 536 envs the bench line's own HIP-event period, the rocprofv3
kernel-trace durations of the same command (the dispatches of the timed graph replays), the PMC traffic per launch, the stamps."""
import csv
import glob
import json
import sys

O = sys.argv[1]
N = 65536
e = {}


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


try:
    p = line(f"{O}/plain.json")
    e["hip_event"] = {"envs": N, "steps": p["steps"], "launches_timed": p["roofline"]["launches_timed"], "hip_event_us_per_launch": p["roofline"]["avg_launch_us"],
                      "wall_us_per_step": p["ms_per_step"] * 1e3, "rollout_rows": p["roofline"].get("rollout_rows"), "working_set_bytes": p["roofline"]["working_set_bytes"],
                      "hbm_bytes_beyond_mall": p["roofline"]["hbm_bytes_beyond_mall"], "frac_of_8TBps": p["roofline"]["frac"], "value": p["value"]}
    e["hip_event_under_kernel_trace"] = line(f"{O}/kt.json")["roofline"]["avg_launch_us"]
except Exception as ex:  # noqa: BLE001
    e["error_plain"] = str(ex)
durs, starts = [], []
for f in glob.glob(f"{O}/kt/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_env_step" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    starts = [int(r["Start_Timestamp"]) / 1e3 for r in rows]
if durs:
    n_timed = e.get("hip_event", {}).get("launches_timed", len(durs))
    steady = sorted(durs[-n_timed:])
    gaps = sorted(b - a for a, b in zip(starts[-n_timed:], starts[-n_timed + 1:]))
    e["rocprof_kernel_trace"] = {"kernel": "k_env_step<u64, int8, L = 25>", "calls": len(steady), "calls_all": len(durs), "avg_us": sum(steady) / len(steady),
                                 "median_us": steady[len(steady) // 2], "min_us": steady[0], "max_us": steady[-1],
                                 "period_us_under_rocprof": gaps[len(gaps) // 2] if gaps else None,
                                 "note": "the dispatches of the timed region (the last `launches_timed` of the run: replayed graph nodes, back to back), begin-to-end per dispatch; "
                                         "period = median distance between consecutive dispatch starts in the same trace"}
for name in ("fetch", "write"):
    vals = []
    for f in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_env_step" in r["Kernel_Name"]]
    if vals:
        vals = vals[-400:]
        e[name.upper() + "_SIZE_KB_per_launch"] = sum(vals) / len(vals)
if "FETCH_SIZE_KB_per_launch" in e and "WRITE_SIZE_KB_per_launch" in e:
    e["traffic_bytes_per_launch"] = (2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"]) * 1024
    e["traffic_formula"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950 counts 128-B read requests at 64 B; Infinity-Cache hits are included in both counters)"
algo = 107 * N
e["algorithmic_bytes_per_launch"] = algo
if "rocprof_kernel_trace" in e:
    e["frac_of_8TBps_by_rocprof_avg"] = algo / (e["rocprof_kernel_trace"]["avg_us"] * 1e-6) / 8e12
out = {str(N): e}
try:
    out["stamps_65536"] = json.load(open(f"{O}/stamps_65536.json"))
except Exception as ex:  # noqa: BLE001
    out["stamps_error"] = str(ex)
out["command"] = "python3 bench.py --steps 20 --warmup 5 --no-search --no-extras --no-cpu-baseline (plain, under rocprofv3 --kernel-trace --stats, under --pmc FETCH_SIZE, under --pmc WRITE_SIZE)"
print(json.dumps(out, indent=1))
