"""GPU parity of the AC moves: the HIP kernels, reached through the C ABI / the drop-in Python
surface, against the golden vectors of the reference and against the oracle.  Bit exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = int(os.environ.get("ACX_FUZZ_SEED", "0"))  # soak runs (tools/fuzz_soak.sh): other seeds, other random cases

EXC = {1: AssertionError, 2: IndexError, 3: ValueError}


@pytest.fixture(scope="module")
def acx():
    from ac_solver import _acx

    _acx.require_device()
    return _acx


def packable(st, L):
    ok = (np.abs(st) <= 2).all(1)
    for h in (0, 1):
        half = st[:, h * L:(h + 1) * L]
        nz = half != 0
        ok &= (nz == (np.arange(L)[None, :] < nz.sum(1)[:, None])).all(1)
    return ok


# ---- the reference's own tests/test_ac_env.py tables, through the drop-in functions -------------
def test_simplify_relator_reference_table(acx, golden_json):
    from ac_solver.envs.utils import simplify_relator

    for r in golden_json("unit_tables.json")["simplify_relator"]:
        out, n = simplify_relator(np.array(r["relator"]), r["L"], cyclical=r["cyclical"], padded=r["padded"])
        assert np.array_equal(out, np.array(r["out"])) and n == r["length"], r


def test_simplify_relator_fuzz(acx, golden_json):
    from ac_solver.envs.utils import simplify_relator

    for r in golden_json("simplify_fuzz.json"):
        if r["err"]:
            with pytest.raises(EXC[r["err"]]):
                simplify_relator(np.array(r["relator"]), r["L"], cyclical=r["cyclical"], padded=r["padded"])
        else:
            out, n = simplify_relator(np.array(r["relator"]), r["L"], cyclical=r["cyclical"], padded=r["padded"])
            assert out.tolist() == r["out"] and n == r["length"], r


def test_simplify_presentation_table(acx, golden_json):
    from ac_solver.envs.utils import simplify_presentation

    for r in golden_json("unit_tables.json")["simplify_presentation"]:
        out, lens = simplify_presentation(np.array(r["presentation"]), r["L"], r["lengths"])
        assert out.tolist() == r["out"] and lens == r["out_lengths"]


@pytest.mark.parametrize("name", ["concatenate_relators", "conjugate"])
def test_move_tables(acx, golden_json, name):
    from ac_solver.envs import ac_moves

    fn = getattr(ac_moves, name)
    for r in golden_json("unit_tables.json")[name]:
        out, lens = fn(np.array(r["presentation"]), r["L"], r["i"], r["j"], r["sign"], list(r["lengths"]))
        assert out.tolist() == r["out"] and lens == r["out_lengths"], r


def test_raw_moves_fuzz(acx, golden_json):
    from ac_solver.envs.ac_moves import concatenate_relators, conjugate

    rows = golden_json("moves_raw_fuzz.json")
    for r in rows[::3]:
        fn = concatenate_relators if r["fn"] == "cat" else conjugate
        if r["err"]:
            with pytest.raises(EXC[r["err"]]):
                fn(np.array(r["p"]), r["L"], r["i"], r["j"], r["sign"], list(r["lengths"]))
        else:
            out, lens = fn(np.array(r["p"]), r["L"], r["i"], r["j"], r["sign"], list(r["lengths"]))
            assert out.tolist() == r["out"] and lens == r["out_lengths"], r


def test_acmove_table_and_dtype(acx, golden_json):
    from ac_solver.envs.ac_moves import ACMove

    for r in golden_json("unit_tables.json")["ACMove"]:
        p = np.array(r["presentation"])
        out, lens = ACMove(r["move"], p, r["L"], [4, 4], cyclical=r["cyclical"])
        assert out.tolist() == r["out"] and lens == r["out_lengths"]
        assert out.dtype == p.dtype and out is not p
    with pytest.raises(AssertionError):
        ACMove(12, np.array([1, 0, 2, 0]), 2, [1, 1])
    with pytest.raises(AssertionError):  # r1 = r0: move 1 cancels r0 completely
        ACMove(1, np.array([1, 2, 0, 1, 2, 0]), 3, [2, 2])


def test_stable_ak3_notebook_sequence(acx, golden_json):
    from ac_solver.envs.ac_moves import ACMove
    from ac_solver.envs.utils import convert_relators_to_presentation

    g = golden_json("stable_ak3.json")
    state = convert_relators_to_presentation(g["relator1"], g["relator2"], g["L"])
    lens = [13, 12]
    for m in g["sequence_one_based"]:
        state, lens = ACMove(m - 1, state, g["L"], lens, cyclical=False)
    assert state.tolist() == g["end_state"] and lens == g["end_lengths"]


# ---- batched kernels against the reference fuzz vectors ---------------------------------------------
@pytest.mark.parametrize("L", [2, 3, 4, 5, 7, 12, 25, 36])
def test_byte_kernel_fuzz(acx, golden_npz, L):
    z = golden_npz("acmove_fuzz.npz")
    st, mv, cy = z[f"L{L}_state"], z[f"L{L}_move"], z[f"L{L}_cyclical"]
    for c in (0, 1):
        m = cy == c
        out, lens, err, _ = acx.move_rows(st[m], mv[m], L, acx.F_BYTES | (acx.F_CYCLICAL if c else 0))
        assert np.array_equal(err, z[f"L{L}_err"][m])
        assert np.array_equal(out, z[f"L{L}_out"][m])
        assert np.array_equal(lens, z[f"L{L}_lens"][m])


@pytest.mark.parametrize("L", [2, 3, 4, 5, 7, 12, 25, 36])
def test_packed_kernel_fuzz(acx, golden_npz, L):
    z = golden_npz("acmove_fuzz.npz")
    st, mv, cy = z[f"L{L}_state"], z[f"L{L}_move"], z[f"L{L}_cyclical"]
    ok = packable(st, L)
    for c in (0, 1):
        m = cy == c
        out, lens, err, _ = acx.move_rows(st[m], mv[m], L, acx.F_CYCLICAL if c else 0)
        okm = ok[m]
        assert (err[~okm] == acx.ERR_UNPACKABLE).all() and np.array_equal(out[~okm], st[m][~okm])
        assert np.array_equal(err[okm], z[f"L{L}_err"][m][okm])
        assert np.array_equal(out[okm], z[f"L{L}_out"][m][okm])
        assert np.array_equal(lens[okm], z[f"L{L}_lens"][m][okm])


def _random_states(rng, n, L):
    st = np.zeros((n, 2 * L), np.int8)
    for r in range(n):
        for h in (0, 1):
            ln = int(rng.integers(1, L + 1)) if r % 5 else L
            w = rng.choice([1, -1, 2, -2], size=ln)
            if r % 3 == 0:
                red = []
                for c in w:
                    if red and red[-1] == -c:
                        red.pop()
                    else:
                        red.append(c)
                w = np.array(red or [1])
            st[r, h * L:h * L + len(w)] = w
        if r % 11 == 0:
            w = st[r, :L][st[r, :L] != 0]
            st[r, L:] = 0
            st[r, L:L + len(w)] = -w[::-1] if r % 2 else w
    return st


@pytest.mark.parametrize("L", [1, 2, 6, 25, 31, 32, 33, 40, 63, 64])
def test_kernels_vs_oracle_random(acx, L):
    """word-width edges: L = 32 fills a u64 lane word, L = 33..64 use the 128-bit instantiation"""
    from oracle import ac_oracle as O

    rng = np.random.default_rng(100 + L + 1000 * SEED)
    n = 5003  # ragged: not a multiple of the 64-row wave tile
    st = _random_states(rng, n, L)
    mv = rng.integers(0, 12, size=n).astype(np.uint8)
    for c in (0, 1):
        want = O.move_batch(st, mv, L, cyclical=bool(c))
        for flags in (0, acx.F_BYTES):
            got = acx.move_rows(st, mv, L, flags | (acx.F_CYCLICAL if c else 0))
            for g, w in zip(got[:3], want):
                assert np.array_equal(g, w)


def test_empty_batch_and_bad_args(acx):
    out, lens, err, _ = acx.move_rows(np.zeros((0, 50), np.int8), np.zeros(0, np.uint8), 25, acx.F_CYCLICAL)
    assert out.shape == (0, 50)
    with pytest.raises(acx.AcxError):
        acx.move_rows(np.zeros((1, 2 * 70), np.int8), [0], 70, 0)  # packed path is L <= 64


def test_large_batch_packed_equals_bytes_and_oracle(acx):
    """BASELINE config 2 size: 65 536 rows at L = 25 through both kernels and the oracle."""
    from oracle import ac_oracle as O

    rng = np.random.default_rng(2)
    L, n = 25, 65536
    st = _random_states(rng, n, L)
    mv = rng.integers(0, 12, size=n).astype(np.uint8)
    want = O.move_batch(st, mv, L, cyclical=True)
    a = acx.move_rows(st, mv, L, acx.F_CYCLICAL)
    b = acx.move_rows(st, mv, L, acx.F_CYCLICAL | acx.F_BYTES)
    for x, y, w in zip(a[:3], b[:3], want):
        assert np.array_equal(x, w) and np.array_equal(y, w)


def test_inverse_moves_round_trip_at_full_size(acx):
    """Size-independent property at 2 Mi rows per move pair (L = 25, no oracle in the loop): on freely reduced relators every AC
    move that is carried out is undone by its inverse move -- r_i r_j then r_i r_j^-1 (0 <-> 2, 1 <-> 3), conjugation by g then by
    g^-1 (4 <-> 8, 5 <-> 9, 6 <-> 10, 7 <-> 11) -- and a move that does not fit max_relator_length leaves the row as it was."""
    L, n = 25, 1 << 21
    rng = np.random.default_rng(11)
    letters = np.array([1, -1, 2, -2], np.int8)
    st = np.zeros((n, 2 * L), np.int8)
    for h in (0, 1):
        code = np.empty((n, L), np.int64)
        code[:, 0] = rng.integers(0, 4, n)
        step = rng.integers(0, 3, (n, L))
        for k in range(1, L):
            code[:, k] = ((code[:, k - 1] ^ 1) + 1 + step[:, k]) % 4  # any letter but the inverse of the previous one
        ln = rng.integers(1, L + 1, n)
        half = letters[code]
        half[np.arange(L)[None, :] >= ln[:, None]] = 0
        st[:, h * L:(h + 1) * L] = half
    total = applied = 0
    for a, b in ((0, 2), (2, 0), (1, 3), (3, 1), (4, 8), (8, 4), (5, 9), (9, 5), (6, 10), (10, 6), (7, 11), (11, 7)):
        s1, len1, err1, _ = acx.move_rows(st, np.full(n, a, np.uint8), L, 0)
        moved = (s1 != st).any(1)
        # (a product that cancels completely leaves an empty relator: the reference's ACMove raises there, err1 != 0)
        alive = (len1 > 0).all(1) & (err1 == 0)
        assert float(alive.mean()) > 0.99
        s2, _, err2, _ = acx.move_rows(np.where(alive[:, None], s1, st), np.full(n, b, np.uint8), L, 0)
        ok = moved & alive
        assert not err2[ok].any() and np.array_equal(s2[ok], st[ok]), (a, b)
        total += n
        applied += int(ok.sum())
    assert applied > 0.5 * total


def test_miller_schupp_generator(acx, golden_json):
    from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

    g = golden_json("ms_pool.json")
    for key, want in g["small"].items():
        n, mw = (int(v) for v in key.split(","))
        got = generate_miller_schupp_presentations(n, mw)
        assert {str(k): v for k, v in got.items()} == want
    total = 0
    for n in range(1, 8):
        got = generate_miller_schupp_presentations(n, 7)
        assert {str(k): v for k, v in got.items()} == g["by_n"][str(n)]
        assert [len(got[k]) for k in range(1, 8)] == [2, 2, 2, 6, 18, 42, 98]  # 170 per n (reference test_miller_schupp.py:42-56)
        total += sum(len(v) for v in got.values())
    assert total == 1190


@pytest.mark.parametrize("L", [65, 100, 128])
def test_byte_kernel_wide_rows(acx, L):
    """the byte-exact kernel covers max_relator_length up to 128 (the packed one stops at 64)"""
    from oracle import ac_oracle as O

    rng = np.random.default_rng(L + 1000 * SEED)
    st = _random_states(rng, 700, L)
    mv = rng.integers(0, 12, size=len(st)).astype(np.uint8)
    for c in (0, 1):
        want = O.move_batch(st, mv, L, cyclical=bool(c))
        got = acx.move_rows(st, mv, L, acx.F_BYTES | (acx.F_CYCLICAL if c else 0))
        for g, w in zip(got[:3], want):
            assert np.array_equal(g, w)


def test_device_pointer_entry_with_torch_tensors(acx):
    """acx_move_batch_device on torch-owned buffers: every action dtype, an unaligned base pointer (byte-wise tile path)
    and a non-default stream"""
    import torch

    from oracle import ac_oracle as O

    rng = np.random.default_rng(11)
    L, n = 25, 3001
    st = _random_states(rng, n, L)
    mv = rng.integers(0, 12, size=n)
    want = O.move_batch(st, mv.astype(np.uint8), L, cyclical=True)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for adt, code in ((torch.uint8, acx.U8), (torch.int32, acx.I32), (torch.int64, acx.I64), (torch.int8, acx.I8)):
            for shift in (0, 1):
                buf_in = torch.zeros(n * 2 * L + 16, dtype=torch.int8, device="cuda")
                buf_out = torch.zeros(n * 2 * L + 16, dtype=torch.int8, device="cuda")
                d_in = buf_in[shift:shift + n * 2 * L]
                d_in.copy_(torch.as_tensor(st.reshape(-1)))
                d_out = buf_out[shift:shift + n * 2 * L]
                d_act = torch.as_tensor(mv).to(device="cuda", dtype=adt)
                d_len = torch.empty((n, 2), dtype=torch.int32, device="cuda")
                d_err = torch.empty(n, dtype=torch.uint8, device="cuda")
                acx.check(acx.lib.acx_move_batch_device(d_in.data_ptr(), d_act.data_ptr(), code, n, L, acx.F_CYCLICAL, d_out.data_ptr(), d_len.data_ptr(),
                                                        d_err.data_ptr(), None, stream.cuda_stream))
                stream.synchronize()
                assert np.array_equal(d_out.cpu().numpy().reshape(n, 2 * L), want[0])
                assert np.array_equal(d_len.cpu().numpy(), want[1]) and np.array_equal(d_err.cpu().numpy(), want[2])
