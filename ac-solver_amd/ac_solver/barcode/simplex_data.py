"""Simplex data of the AC graph of presentations of total length <= n -- the reference's
barcode_analysis/simplex_data_generation/{prime,classic}_moves/ac_bfs.cpp on the GPU (C ABI `acx_simplex_graph`).

    python -m ac_solver.barcode.simplex_data 14 --moves prime --out-dir .

writes the reference's four files: zero_simplices_n, zero_filtrations_n, one_simplices_n, one_filtrations_n.
"""
import argparse
import ctypes as C
import os

import numpy as np

from ac_solver import _acx


def simplex_graph(n, classic=False, cap_nodes=None, cap_edges=None):
    """-> (node_size uint8 [V], edges uint32 [E, 2], edge_filtration uint8 [E]) in the reference's order."""
    # the graph grows by about 9.3x (vertices) per two units of n: 100, 884, 9172, 84996 for n = 4, 6, 8, 10
    guess = int(120 * 3.1 ** max(n - 4, 0))
    cap_nodes = int(cap_nodes or min(max(guess, 4096), 1 << 29))
    cap_edges = int(cap_edges or 3 * cap_nodes)
    for _ in range(16):
        sizes = np.empty(cap_nodes, np.uint8)
        edges = np.empty((cap_edges, 2), np.uint32)
        filt = np.empty(cap_edges, np.uint8)
        nv, ne = C.c_int64(0), C.c_int64(0)
        rc = _acx.lib.acx_simplex_graph(int(n), int(bool(classic)), cap_nodes, cap_edges, C.byref(nv), _acx.ptr(sizes, C.c_uint8), C.byref(ne),
                                        _acx.ptr(edges, C.c_uint32), _acx.ptr(filt, C.c_uint8))
        if rc == _acx.E_CAPACITY:  # grow and repeat
            cap_nodes = max(4 * cap_nodes, int(1.5 * nv.value))  # (the search stops at the first overflow: the counts are lower bounds)
            cap_edges = max(4 * cap_edges, int(1.5 * ne.value), 3 * cap_nodes)
            continue
        _acx.check(rc, "acx_simplex_graph")
        return sizes[:nv.value].copy(), edges[:ne.value].copy(), filt[:ne.value].copy()
    raise MemoryError("the graph did not fit after sixteen capacity increases")


def _join(values):
    return ",".join(map(str, values))


def write_simplex_files(n, classic=False, out_dir="."):
    """The four files of ac_bfs.cpp:23-37 / :88-91, byte for byte (lists end with the reference's terminators)."""
    sizes, edges, filt = simplex_graph(n, classic)
    os.makedirs(out_dir, exist_ok=True)
    sep = "," if len(sizes) else ""
    with open(os.path.join(out_dir, f"zero_simplices_{n}"), "w") as f:
        f.write('{"0-simplices":[' + "".join(f"[{k}]," for k in range(len(sizes))) + "[]]}")
    with open(os.path.join(out_dir, f"zero_filtrations_{n}"), "w") as f:
        f.write('{"0-filt":[' + _join(sizes.tolist()) + sep + "-5]}")
    with open(os.path.join(out_dir, f"one_simplices_{n}"), "w") as f:
        f.write('{"1-simplices":[' + "".join(f"[{a},{b}]," for a, b in edges.tolist()) + "[]]}")
    with open(os.path.join(out_dir, f"one_filtrations_{n}"), "w") as f:
        f.write('{"1-filt":[' + _join(filt.tolist()) + ("," if len(filt) else "") + "-5]}")
    return len(sizes), len(edges)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="simplex data of the AC graph of presentations of total length <= n")
    ap.add_argument("n", type=int)
    ap.add_argument("--moves", choices=["prime", "classic"], default="prime")
    ap.add_argument("--out-dir", default=".")
    a = ap.parse_args()
    print(a.n)
    v, e = write_simplex_files(a.n, a.moves == "classic", a.out_dir)
    print(f"{v} vertices, {e} edges")
