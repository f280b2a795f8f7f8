"""`neighbourhood_sizes`: for every presentation the number of distinct presentations within `radius` AC moves, as
computed by the reference's `neibourhood` (barcode_analysis/5_steps_neibourhoods/neibourhoods.cpp:18-54): presentations
are SORTED pairs of freely reduced relators of unbounded length (no cyclic reduction, no length cap), the moves are the
14 "classic" or the 12 "prime" moves of AC_UTILS_no_hash.cpp:144-211.  One GPU workgroup per presentation
(csrc/acx_ball.hip, C ABI `acx_ball_sizes`).

    python -m ac_solver.barcode.neighbourhoods presentations.txt sizes.txt --radius 5 --moves prime
"""
import argparse
import ctypes as C
from ast import literal_eval

import numpy as np

from ac_solver import _acx


def neighbourhood_sizes(presentations, radius=5, classic=False, return_max_length=False):
    """presentations: sequence of equal-length integer rows [r1 | r2] (zero padded, like the lines of the reference's
    input files) or one such row.  -> list of ball sizes (and, optionally, the longest relator met in each ball)."""
    rows = np.asarray(presentations)
    single = rows.ndim == 1
    rows = np.ascontiguousarray(np.atleast_2d(rows), dtype=np.int8)
    if rows.shape[1] % 2:
        raise ValueError("a presentation row holds two relators of equal width")
    n, width = rows.shape
    sizes = np.zeros(n, np.int64)
    maxlen = np.zeros(n, np.int32)
    rc = _acx.lib.acx_ball_sizes(_acx.ptr(rows, C.c_int8), n, width // 2, int(radius), int(bool(classic)), _acx.ptr(sizes, C.c_int64),
                                 _acx.ptr(maxlen, C.c_int32))
    if rc == _acx.E_ROWERR:
        raise ValueError(_acx.last_error())
    _acx.check(rc, "acx_ball_sizes")
    out = sizes.tolist()
    if return_max_length:
        return (out[0], int(maxlen[0])) if single else (out, maxlen.tolist())
    return out[0] if single else out


def neighbourhood_sizes_of_file(source, out, radius=5, classic=False):
    """The reference's read_do_and_write (neibourhoods.cpp:58-100): one presentation (a Python-style list) per input
    line, one size per output line.  Rows of different widths are handled width by width."""
    rows = [literal_eval(line.strip()) for line in open(source) if line.strip()]
    sizes = [0] * len(rows)
    by_width = {}
    for k, r in enumerate(rows):
        by_width.setdefault(len(r), []).append(k)
    for idx in by_width.values():
        for k, s in zip(idx, neighbourhood_sizes([rows[k] for k in idx], radius, classic)):
            sizes[k] = s
    with open(out, "w") as f:
        for s in sizes:
            f.write(f"{s}\n")
    return sizes


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="sizes of radius-r neighbourhoods of presentations in the AC graph")
    ap.add_argument("source")
    ap.add_argument("out")
    ap.add_argument("--radius", type=int, default=5)
    ap.add_argument("--moves", choices=["prime", "classic"], default="prime")
    a = ap.parse_args()
    neighbourhood_sizes_of_file(a.source, a.out, a.radius, a.moves == "classic")
