#!/usr/bin/env python3
"""gpurun_out/r4env/* -> one JSON: per batch size the HIP-event time (median block), the rocprofv3 kernel-trace durations of the same
command -- all dispatches AND the steady ones (the last 5 x 40, behind the 0.3 s warm-up) -- and the PMC traffic per launch."""
import csv
import glob
import json
import sys

O = sys.argv[1]
out = {}
for N in (1048576, 4194304):
    e = {}
    try:
        e["hip_event"] = json.loads(open(f"{O}/plain_{N}.json").read().strip().splitlines()[-1])
        e["hip_event_under_kernel_trace"] = json.loads(open(f"{O}/kt_{N}.json").read().strip().splitlines()[-1])
    except Exception as ex:  # noqa: BLE001
        e["error_plain"] = str(ex)
    durs = []
    for f in glob.glob(f"{O}/kt_{N}/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_env_step" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    if durs:
        steady = sorted(durs[-200:])
        e["rocprof_kernel_trace"] = {"calls": len(durs), "avg_us_all": sum(durs) / len(durs), "steady_calls": len(steady), "avg_us": sum(steady) / len(steady),
                                     "median_us": steady[len(steady) // 2], "min_us": steady[0], "max_us": steady[-1],
                                     "note": "steady = the last 200 dispatches (the five timed blocks of 40), begin-to-end per dispatch"}
    for name in ("fetch", "write"):
        vals = []
        for f in glob.glob(f"{O}/{name}_{N}/**/*counter_collection.csv", recursive=True):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_env_step" in r["Kernel_Name"]]
        if vals:
            vals = vals[-40:]
            e[name.upper() + "_SIZE_KB_per_launch"] = sum(vals) / len(vals)
    if "FETCH_SIZE_KB_per_launch" in e and "WRITE_SIZE_KB_per_launch" in e:
        e["traffic_bytes_per_launch"] = (2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"]) * 1024
        e["traffic_formula"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950 counts 128-B read requests at 64 B; Infinity-Cache hits are included in both counters)"
    algo = 107 * N
    e["algorithmic_bytes_per_launch"] = algo
    if "rocprof_kernel_trace" in e:
        e["frac_of_8TBps_by_rocprof_avg"] = algo / (e["rocprof_kernel_trace"]["avg_us"] * 1e-6) / 8e12
    if "hip_event" in e:
        e["frac_of_8TBps_by_hip_events"] = e["hip_event"]["frac_of_8TBps"]
    out[str(N)] = e
print(json.dumps(out, indent=1))
