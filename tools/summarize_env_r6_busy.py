#!/usr/bin/env python3
"""gpurun_out/r6busy/* (tools/profile_env_r6_busy.sh) -> the `sq_busy` entry of profiles/r6_env_step_roofline.json["65536"]:
SQ_BUSY_CYCLES per k_env_step launch at 65 536 envs, turned into microseconds with the unit calibrated at 4 Mi envs (where the kernel
trace and the HIP events of profiles/r6_env_step_roofline.json agree on the launch's duration)."""
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(tag, last):
    out = {}
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_env_step" in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v[-last:]) / len(v[-last:]) for k, v in out.items()}


small, big, waves = counters("busy_65536", 400), counters("busy_4194304", 40), counters("waves_65536", 400)
ref = json.load(open(os.path.join(ROOT, "profiles", "r6_env_step_roofline.json")))
us_big = ref["4194304"]["rocprof_kernel_trace"]["avg_us"]
out = {"SQ_BUSY_CYCLES_per_launch_65536": small.get("SQ_BUSY_CYCLES"), "SQ_BUSY_CYCLES_per_launch_4Mi": big.get("SQ_BUSY_CYCLES"),
       "GRBM_GUI_ACTIVE_per_launch_65536": small.get("GRBM_GUI_ACTIVE"), "GRBM_GUI_ACTIVE_per_launch_4Mi": big.get("GRBM_GUI_ACTIVE"),
       "traced_us_per_launch_4Mi": us_big, "other_counters_65536": waves}
if small.get("SQ_BUSY_CYCLES") and big.get("SQ_BUSY_CYCLES"):
    per_us = big["SQ_BUSY_CYCLES"] / us_big  # counter units per microsecond of a launch that keeps every SQ busy
    out["busy_us_per_launch_65536"] = small["SQ_BUSY_CYCLES"] / per_us
    out["frac_of_8TBps_by_busy_time"] = 107 * 65536 / (out["busy_us_per_launch_65536"] * 1e-6) / 8e12
    out["note"] = ("SQ_BUSY_CYCLES per launch / (SQ_BUSY_CYCLES per microsecond of the 4 Mi-env launch): the time some wave of the launch is on the chip, seen by the "
                   "profiler; to set beside the in-kernel stamps' `active_us` (first wave's begin to last wave's end)")
print(json.dumps(out, indent=1))
