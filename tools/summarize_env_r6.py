#!/usr/bin/env python3
"""gpurun_out/r6env/* (tools/profile_env_r6.sh) -> one JSON (profiles/r6_env_step_roofline.json).
Per size: HIP-event period, rocprofv3 kernel-trace durations of the same command, PMC traffic per launch (2 x FETCH_SIZE + WRITE_SIZE:
gfx950 tallies 128-B read requests at 64 B), GRBM_GUI_ACTIVE per launch (sum over the 8 XCDs -> busy time at the nominal 2.4 GHz).
At 65 536 envs the plain / traced command is bench.py itself; the in-kernel wave stamps are appended."""
import csv
import glob
import json
import sys

O = sys.argv[1]
CLOCK_GHZ = 2.4  # MI355X peak engine clock; DVFS runs a loaded chip lower, which makes the busy time below a LOWER bound
L = 25


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def trace(pattern, last):
    durs, starts = [], []
    for f in glob.glob(pattern, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_env_step" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        starts = [int(r["Start_Timestamp"]) / 1e3 for r in rows]
    if not durs:
        return None
    steady = sorted(durs[-last:])
    gaps = sorted(b - a for a, b in zip(starts[-last:], starts[-last + 1:]))
    return {"calls": len(steady), "calls_all": len(durs), "avg_us": sum(steady) / len(steady), "median_us": steady[len(steady) // 2], "min_us": steady[0],
            "max_us": steady[-1], "period_us_under_rocprof": gaps[len(gaps) // 2] if gaps else None}


def counter(pattern, last):
    vals = []
    for f in glob.glob(pattern, recursive=True):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_env_step" in r["Kernel_Name"]]
    vals = vals[-last:]
    return sum(vals) / len(vals) if vals else None


def counters(e, tag, last):
    f, w, g = (counter(f"{O}/pmc_{c}_{tag}/**/*counter_collection.csv", last) for c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"))
    if f is not None and w is not None:
        e["FETCH_SIZE_KB_per_launch"], e["WRITE_SIZE_KB_per_launch"] = f, w
        e["traffic_bytes_per_launch"] = (2 * f + w) * 1024
        e["traffic_formula"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950 counts 128-B read requests at 64 B; Infinity-Cache hits are included in both counters)"
    if g is not None:
        e["GRBM_GUI_ACTIVE_per_launch"] = g
        e["grbm_busy_us_per_launch_at_2p4GHz"] = g / 8 / (CLOCK_GHZ * 1e3)
        e["grbm_note"] = ("rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; / 8 / 2.4 GHz = busy time per launch if the clock were at its peak (a lower bound of the "
                          "busy time: the loaded chip clocks lower; the guide: the quotient reads high on dispatches shorter than ~0.3 ms)")


out = {}
e = {}
N = 65536
algo = (4 * L + 7) * N
try:
    p = line(f"{O}/plain.json")
    e["hip_event"] = {"envs": N, "steps": p["steps"], "launches_timed": p["roofline"]["launches_timed"], "hip_event_us_per_launch": p["roofline"]["avg_launch_us"],
                      "wall_us_per_step": p["ms_per_step"] * 1e3, "rollout_rows": p["roofline"].get("rollout_rows"), "working_set_bytes": p["roofline"]["working_set_bytes"],
                      "hbm_bytes_beyond_mall": p["roofline"]["hbm_bytes_beyond_mall"], "frac_of_8TBps": p["roofline"]["frac"], "value": p["value"]}
    e["hip_event_under_kernel_trace"] = line(f"{O}/kt.json")["roofline"]["avg_launch_us"]
except Exception as ex:  # noqa: BLE001
    e["error_plain"] = str(ex)
kt = trace(f"{O}/kt/**/*kernel_trace.csv", e.get("hip_event", {}).get("launches_timed", 17340))
if kt:
    kt["kernel"] = "k_env_step<u64, int8, L = 25>"
    kt["note"] = ("the dispatches of the timed region (the last `launches_timed` of the run: replayed graph nodes, back to back), begin-to-end per dispatch; "
                  "period = median distance between consecutive dispatch starts in the same trace")
    e["rocprof_kernel_trace"] = kt
    e["frac_of_8TBps_by_rocprof_avg"] = algo / (kt["avg_us"] * 1e-6) / 8e12
counters(e, "65536", 400)
try:
    e["pmc_pass_hip_event_us_per_launch"] = {c: line(f"{O}/pmc_{c}_65536.json")["hip_event_us_per_launch"] for c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE")}
except Exception as ex:  # noqa: BLE001
    e["error_pmc_lines"] = str(ex)
e["algorithmic_bytes_per_launch"] = algo
out[str(N)] = e
for N, dt in ((1 << 20, "int8"), (1 << 22, "int8"), (1 << 17, "float32")):
    e = {}
    tag = f"{N}_{dt}"
    algo = (4 * L + 7 if dt == "int8" else 12 * L + 10) * N
    try:
        e["hip_event"] = line(f"{O}/plain_{tag}.json")
        e["hip_event_under_kernel_trace"] = line(f"{O}/kt_{tag}.json")
    except Exception as ex:  # noqa: BLE001
        e["error_plain"] = str(ex)
    kt = trace(f"{O}/kt_{tag}/**/*kernel_trace.csv", 200)
    if kt:
        kt["note"] = "steady = the last 200 dispatches (the five timed blocks of 40 behind the 0.3 s warm-up), begin-to-end per dispatch"
        e["rocprof_kernel_trace"] = kt
        e["frac_of_8TBps_by_rocprof_avg"] = algo / (kt["avg_us"] * 1e-6) / 8e12
    counters(e, tag, 40)
    e["algorithmic_bytes_per_launch"] = algo
    if "hip_event" in e:
        e["frac_of_8TBps_by_hip_events"] = e["hip_event"]["frac_of_8TBps"]
    out[tag if dt != "int8" else str(N)] = e
try:
    out["stamps_65536"] = json.load(open(f"{O}/stamps_65536.json"))
except Exception as ex:  # noqa: BLE001
    out["stamps_error"] = str(ex)
out["command"] = ("65536: python3 bench.py --steps 20 --warmup 5 --no-search --no-extras --no-cpu-baseline (plain, under rocprofv3 --kernel-trace --stats); counters: rocprofv3 --pmc "
                  "FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE -- python3 tools/env_roofline.py 65536 400 int8 128 1.  Other sizes: tools/env_roofline.py N 40 dtype (tools/profile_env_r6.sh)")
print(json.dumps(out, indent=1))
