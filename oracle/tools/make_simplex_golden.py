#!/usr/bin/env python3
"""Golden vectors for the simplex-data workload: the four files written by the REFERENCE's own programs
(oracle/_ref/simplex_{prime,classic}, built by oracle/Makefile from /root/reference/barcode_analysis/simplex_data_generation)
for n = 4, 6, 8 as arrays, and for n = 10, 12 as SHA-256 digests of the files.  Build container only;
writes tests/golden/simplex_data.npz and tests/golden/simplex_digests.json (data only)."""
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FILES = ("zero_simplices", "zero_filtrations", "one_simplices", "one_filtrations")


def run_reference(n, classic):
    d = tempfile.mkdtemp()
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", f"simplex_{'classic' if classic else 'prime'}"), str(n)], cwd=d, check=True, capture_output=True)
    return {f: open(os.path.join(d, f"{f}_{n}")).read() for f in FILES}


def main():
    arrays, digests = {}, {}
    for classic in (False, True):
        tag = "classic" if classic else "prime"
        for n in (4, 6, 8):
            t = run_reference(n, classic)
            arrays[f"{tag}_{n}_node_size"] = np.array(json.loads(t["zero_filtrations"])["0-filt"][:-1], np.uint8)
            arrays[f"{tag}_{n}_edges"] = np.array(json.loads(t["one_simplices"])["1-simplices"][:-1], np.uint32).reshape(-1, 2)
            arrays[f"{tag}_{n}_edge_filt"] = np.array(json.loads(t["one_filtrations"])["1-filt"][:-1], np.uint8)
            assert json.loads(t["zero_simplices"])["0-simplices"][:-1] == [[k] for k in range(len(arrays[f"{tag}_{n}_node_size"]))]
        for n in (8, 10, 12):
            t = run_reference(n, classic)
            digests[f"{tag}_{n}"] = {f: hashlib.sha256(t[f].encode()).hexdigest() for f in FILES}
            digests[f"{tag}_{n}"]["vertices"] = len(json.loads(t["zero_filtrations"])["0-filt"]) - 1
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "simplex_data.npz"), **arrays)
    json.dump({"note": "sha256 of the files written by the reference's ac_bfs.cpp (prime / classic) for n = 8, 10, 12", "digests": digests},
              open(os.path.join(ROOT, "tests", "golden", "simplex_digests.json"), "w"), indent=1)
    print({k: v["vertices"] for k, v in digests.items()})


if __name__ == "__main__":
    main()
