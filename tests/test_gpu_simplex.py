"""GPU parity of acx_simplex_graph with the reference's simplex-data programs: arrays for n <= 8, SHA-256 of the four
written files for n = 8, 10, 12 (tests/golden/simplex_digests.json), and the C oracle at n = 9, 11."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def simplex():
    from ac_solver import _acx
    from ac_solver import barcode

    _acx.require_device()
    return barcode


@pytest.mark.parametrize("tag,classic", [("prime", False), ("classic", True)])
def test_small_graphs_equal_reference_arrays(simplex, golden_npz, tag, classic):
    g = golden_npz("simplex_data.npz")
    for n in (4, 6, 8):
        sizes, edges, filt = simplex.simplex_graph(n, classic)
        assert np.array_equal(sizes, g[f"{tag}_{n}_node_size"]), n
        assert np.array_equal(edges, g[f"{tag}_{n}_edges"]) and np.array_equal(filt, g[f"{tag}_{n}_edge_filt"]), n


@pytest.mark.parametrize("tag,classic", [("prime", False), ("classic", True)])
def test_written_files_have_the_reference_digests(simplex, golden_json, tmp_path, tag, classic):
    digests = golden_json("simplex_digests.json")["digests"]
    for n in (8, 10, 12):
        v, e = simplex.write_simplex_files(n, classic, str(tmp_path))
        want = digests[f"{tag}_{n}"]
        assert v == want["vertices"]
        for f in ("zero_simplices", "zero_filtrations", "one_simplices", "one_filtrations"):
            got = hashlib.sha256(open(os.path.join(tmp_path, f"{f}_{n}"), "rb").read()).hexdigest()
            assert got == want[f], (n, f)


def test_odd_sizes_against_oracle_and_capacity_growth(simplex):
    from oracle import ac_oracle as O

    for n, classic in ((9, False), (11, True)):
        sizes, edges, filt = simplex.simplex_graph(n, classic, cap_nodes=1000, cap_edges=100)  # forces the grow-and-repeat path
        ws, we, wf = O.simplex_graph(n, classic)
        assert np.array_equal(sizes, ws) and np.array_equal(edges, we) and np.array_equal(filt, wf)
