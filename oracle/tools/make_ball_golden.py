#!/usr/bin/env python3
"""Golden vectors for the neighbourhood-size workload: sizes printed by the REFERENCE's own program
(oracle/_ref/ball_ref, built by oracle/Makefile from /root/reference/barcode_analysis/5_steps_neibourhoods) for the
five inputs of its README.txt:24-44 and a sample of Miller-Schupp presentations, both move sets.
Runs in the build container only; writes tests/golden/ball_sizes.json (data only)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ball_ref")


def ref_sizes(rows, radius, classic):
    text = "\n".join(str(list(map(int, r))) for r in rows) + "\n"
    out = subprocess.run([REF, str(radius), str(int(classic))], input=text, capture_output=True, text=True, check=True).stdout.split()
    return [int(v) for v in out]


def main():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "ms_pool.json")))
    pool = []
    for n in range(1, 8):
        for w in range(1, 8):
            pool += g["by_n"][str(n)][str(w)]
    readme = [[1, 0, 0, 0, 2, 0, 0, 0], [1, 2, 1, 0, 2, 0, 0, 0], [1, 2, 1, -2, 0, 0, 2, 1, -2, 0, 0, 0], [1, 2], [2, 1]]
    sample = [0, 7, 169, 170, 340, 500, 681, 777, 1019, 1100, 1189]
    cases = []
    for classic in (False, True):
        for radius in (5,):
            for row, size in zip(readme, ref_sizes(readme, radius, classic)):
                cases.append({"tag": "readme", "presentation": row, "radius": radius, "classic": classic, "size": size})
        for radius in (2, 4, 5):
            rows = [pool[k] for k in sample]
            for k, row, size in zip(sample, rows, ref_sizes(rows, radius, classic)):
                cases.append({"tag": f"ms_{k}", "presentation": row, "radius": radius, "classic": classic, "size": size})
    json.dump({"note": "sizes printed by the reference's neibourhoods.cpp (README.txt:38-44 lists the five prime-move answers)", "cases": cases},
              open(os.path.join(ROOT, "tests", "golden", "ball_sizes.json"), "w"))
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
