#!/usr/bin/env python3
"""gpurun_out/r3env/* -> one JSON: per batch size the HIP-event time, the rocprofv3 kernel-trace average of the same command and
the PMC traffic (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md section HBM), all per launch of k_env_step."""
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
out = {}
for N in (65536, 1048576, 4194304):
    e = {}
    try:
        e["hip_event"] = json.loads(open(f"{O}/plain_{N}.json").read().strip().splitlines()[-1])
        e["hip_event_under_kernel_trace"] = json.loads(open(f"{O}/kt_{N}.json").read().strip().splitlines()[-1])["hip_event_us_per_launch"]
    except Exception as ex:  # noqa: BLE001
        e["error_plain"] = str(ex)
    for f in glob.glob(f"{O}/kt_{N}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_env_step" in r["Name"]:
                e["rocprof_kernel_trace"] = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                             "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
    for name in ("fetch", "write"):
        vals = []
        for f in glob.glob(f"{O}/{name}_{N}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_env_step" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        if vals:
            vals = vals[10:] or vals  # (the first launches are the warm-up)
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
for name in ("stamps_65536", "stamps_1048576", "plain_f32_1048576"):
    p = f"{O}/{name}.json"
    if os.path.exists(p) and os.path.getsize(p):
        try:
            out[name] = json.loads(open(p).read())
        except Exception as ex:  # noqa: BLE001
            out[name] = {"error": str(ex)}
print(json.dumps(out, indent=1))
