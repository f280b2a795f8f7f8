#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name (and grid size), count / median / min / max
duration in us, in order of first appearance; consecutive runs of the same kernel are reported separately
when --runs is given (useful for microbenchmarks that launch the same kernel in different configurations)."""
import csv
import glob
import statistics as S
import sys


def main():
    path = sys.argv[1]
    runs = "--runs" in sys.argv
    f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0] if not path.endswith(".csv") else path
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups = []
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][-60:]
        key = (name, r.get("Grid_Size", ""))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if runs:
            if groups and groups[-1][0] == key:
                groups[-1][1].append(d)
            else:
                groups.append((key, [d]))
        else:
            for g in groups:
                if g[0] == key:
                    g[1].append(d)
                    break
            else:
                groups.append((key, [d]))
    for (name, grid), d in groups:
        if len(d) < 3 and runs:
            continue
        print(f"{name:60s} grid={grid:>9s} n={len(d):5d} med={S.median(d):8.2f}us min={min(d):8.2f} max={max(d):8.2f}")


if __name__ == "__main__":
    main()
