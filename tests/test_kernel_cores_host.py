"""CPU check of the kernels' lane code: ac-solver_amd/csrc/acx_word.h (packed path) and acx_bytes.h
(byte-exact path) compiled for the host by tests/hostshim, against the golden fixtures and the
oracle.  This is not the product path (which only exists as HIP kernels); it lets the non-GPU suite
catch arithmetic regressions before a GPU run.  The GPU parity tests live in test_gpu_*.py."""
import ctypes as C

import numpy as np
import pytest

from oracle import ac_oracle as O
from tests.hostshim import build as shimbuild

F_CYC, F_NOSIMP, F_NOMOVE = 1, 2, 4


@pytest.fixture(scope="module")
def shim():
    return C.CDLL(shimbuild.build())


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def packed(shim, st, mv, L, cyc):
    st = np.ascontiguousarray(st, np.int8); mv = np.ascontiguousarray(mv, np.uint8)
    n = len(st)
    out = np.empty_like(st); lens = np.empty((n, 2), np.int32); err = np.empty(n, np.uint8)
    shim.shim_move_packed(_p(st, C.c_int8), _p(mv, C.c_uint8), C.c_int64(n), L, int(cyc), int(L > 32), _p(out, C.c_int8), _p(lens, C.c_int32), _p(err, C.c_uint8))
    return out, lens, err


def bytes_(shim, st, mv, L, flags):
    st = np.ascontiguousarray(st, np.int8); mv = np.ascontiguousarray(mv, np.uint8)
    n = len(st)
    out = np.empty_like(st); lens = np.empty((n, 2), np.int32); err = np.empty(n, np.uint8); fit = np.empty(n, np.int32)
    shim.shim_move_bytes(_p(st, C.c_int8), _p(mv, C.c_uint8), C.c_int64(n), L, flags, _p(out, C.c_int8), _p(lens, C.c_int32), _p(err, C.c_uint8), _p(fit, C.c_int32))
    return out, lens, err, fit


def packable(st, L):
    """rows the packed path accepts: letters in {0,+-1,+-2}, zeros only as right padding"""
    ok = (np.abs(st) <= 2).all(1)
    for h in (0, 1):
        half = st[:, h * L:(h + 1) * L]
        nz = (half != 0)
        n = nz.sum(1)
        ok &= (nz == (np.arange(L)[None, :] < n[:, None])).all(1)
    return ok


@pytest.mark.parametrize("L", [2, 3, 4, 5, 7, 12, 25, 36])
def test_bytes_core_matches_reference_fuzz(shim, golden_npz, L):
    z = golden_npz("acmove_fuzz.npz")
    st, mv, cy = z[f"L{L}_state"], z[f"L{L}_move"], z[f"L{L}_cyclical"]
    for c in (0, 1):
        m = cy == c
        out, lens, err, _ = bytes_(shim, st[m], mv[m], L, F_CYC if c else 0)
        assert np.array_equal(err, z[f"L{L}_err"][m])
        assert np.array_equal(out, z[f"L{L}_out"][m])
        assert np.array_equal(lens, z[f"L{L}_lens"][m])


@pytest.mark.parametrize("L", [2, 3, 4, 5, 7, 12, 25, 36])
def test_packed_core_matches_reference_fuzz(shim, golden_npz, L):
    z = golden_npz("acmove_fuzz.npz")
    st, mv, cy = z[f"L{L}_state"], z[f"L{L}_move"], z[f"L{L}_cyclical"]
    ok = packable(st, L)
    assert ok.sum() > 0.5 * len(st)
    for c in (0, 1):
        m = cy == c
        out, lens, err = packed(shim, st[m], mv[m], L, c)
        okm = ok[m]
        assert (err[~okm] == 250).all() and np.array_equal(out[~okm], st[m][~okm])
        assert np.array_equal(err[okm], z[f"L{L}_err"][m][okm])
        assert np.array_equal(out[okm], z[f"L{L}_out"][m][okm])
        assert np.array_equal(lens[okm], z[f"L{L}_lens"][m][okm])


@pytest.mark.parametrize("L", [1, 2, 6, 25, 31, 32, 33, 40, 63, 64])
def test_packed_core_vs_oracle_random(shim, L):
    """dense random sweep incl. the word-width edges (L = 32 fills a u64, L = 64 a u128)"""
    rng = np.random.default_rng(100 + L)
    n = 4000
    st = np.zeros((n, 2 * L), np.int8)
    for r in range(n):
        for h in (0, 1):
            ln = int(rng.integers(1, L + 1)) if r % 5 else L
            w = rng.choice([1, -1, 2, -2], size=ln)
            if r % 3 == 0:  # freely reduce most rows so that long words survive
                red = []
                for c in w:
                    if red and red[-1] == -c: red.pop()
                    else: red.append(c)
                w = np.array(red or [1])
            st[r, h * L:h * L + len(w)] = w
        if r % 11 == 0:  # r1 = r0^-1 or r0: full cancellation -> AssertionError
            w = st[r, :L][st[r, :L] != 0]
            st[r, L:] = 0
            st[r, L:L + len(w)] = -w[::-1] if r % 2 else w
    mv = rng.integers(0, 12, size=n).astype(np.uint8)
    for c in (0, 1):
        want = O.move_batch(st, mv, L, cyclical=bool(c))
        got = packed(shim, st, mv, L, c)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        gotb = bytes_(shim, st, mv, L, F_CYC if c else 0)
        for g, w in zip(gotb[:3], want):
            assert np.array_equal(g, w)


def test_bytes_core_raw_moves(shim, golden_json):
    """concatenate_relators / conjugate without the simplify step (ACX_F_NO_SIMPLIFY)"""
    cat = {(1, 0, 1): 0, (0, 1, -1): 1, (1, 0, -1): 2, (0, 1, 1): 3}
    conj = {(1, 1, -1): 4, (0, 2, -1): 5, (1, 2, -1): 6, (0, 1, 1): 7, (1, 1, 1): 8, (0, 2, 1): 9, (1, 2, 1): 10, (0, 1, -1): 11}
    for r in golden_json("moves_raw_fuzz.json"):
        mid = (cat if r["fn"] == "cat" else conj)[(r["i"], r["j"], r["sign"])]
        out, lens, err, fit = bytes_(shim, np.array([r["p"]], np.int8), [mid], r["L"], F_NOSIMP)
        assert err[0] == r["err"], r
        if not r["err"]:
            assert out[0].tolist() == r["out"], r
            want = list(r["lengths"])
            if fit[0] >= 0:
                want[r["i"]] = int(fit[0])
            assert want == r["out_lengths"], r


def test_simplify_rows_core(shim, golden_json):
    rows = golden_json("simplify_fuzz.json") + [dict(r, err=0) for r in golden_json("unit_tables.json")["simplify_relator"]]
    for r in rows:
        rel = np.array(r["relator"], np.int8)
        w = len(rel)
        if w == 0:
            continue
        out = np.empty(w, np.int8); lens = np.empty(2, np.int32); err = np.empty(1, np.uint8)
        shim.shim_simplify_rows(_p(rel, C.c_int8), C.c_int64(1), w, int(r["cyclical"]), _p(out, C.c_int8), _p(lens, C.c_int32), _p(err, C.c_uint8))
        n, nz = int(lens[0]), int(lens[1])
        alen = w - (nz - n)  # array length after the reference's np.delete calls
        if err[0]:
            assert r["err"] == 1, r
            continue
        if r["padded"] and r["L"] - alen < 0:
            assert r["err"] == 3, r  # np.pad ValueError
            continue
        if r["L"] < n:
            assert r["err"] == 1, r
            continue
        assert r["err"] == 0, r
        got = out[:n].tolist() + [0] * ((r["L"] if r["padded"] else alen) - n)
        assert got == r["out"] and n == r["length"], r


def test_packed_comparator_is_signed_tuple_order(shim):
    rng = np.random.default_rng(5)
    for L in (3, 25, 36):
        rows = []
        for _ in range(300):
            p = np.zeros(2 * L, np.int8)
            for h in (0, 1):
                ln = int(rng.integers(1, min(L, 6) + 1))
                p[h * L:h * L + ln] = rng.choice([1, -1, 2, -2], size=ln)
            rows.append(p)
        for a in rows[:60]:
            for b in rows:
                want = (tuple(a.tolist()) > tuple(b.tolist())) - (tuple(a.tolist()) < tuple(b.tolist()))
                got = shim.shim_compare(_p(a, C.c_int8), _p(b, C.c_int8), L, int(L > 32))
                assert got == want, (a, b)


@pytest.mark.parametrize("L", [2, 3, 7, 25, 31, 32, 36, 64])
def test_reduced_fast_path_matches_oracle(shim, L):
    """apply_move_reduced (the env kernel's steady-state path) == ACMove(cyclical=True) on cyclically reduced states"""
    rng = np.random.default_rng(900 + L)
    n = 6000
    st = np.zeros((n, 2 * L), np.int8)
    mv = rng.integers(0, 12, size=n).astype(np.uint8)
    for r in range(n):
        for h in (0, 1):
            hi = L if r % 4 else min(L, 4)
            w = list(rng.choice([1, -1, 2, -2], size=int(rng.integers(1, hi + 1))))
            st[r, h * L:h * L + len(w)] = w
        if r % 13 == 0:  # r1 = r0^+-1: concatenation cancels completely -> AssertionError
            w = st[r, :L][st[r, :L] != 0]
            st[r, L:] = 0
            st[r, L:L + len(w)] = -w[::-1] if r % 2 else w
    # normalise with the oracle so that the inputs are in the steady-state form, then step twice
    cur, lens0, err0 = O.move_batch(st, np.full(n, 4, np.uint8), L, cyclical=True)
    keep = err0 == 0
    cur, mv = cur[keep], mv[keep]
    for _ in range(2):
        out = np.empty_like(cur); lens = np.empty((len(cur), 2), np.int32); err = np.empty(len(cur), np.uint8)
        shim.shim_move_reduced(_p(cur, C.c_int8), _p(mv, C.c_uint8), C.c_int64(len(cur)), L, int(L > 32), _p(out, C.c_int8), _p(lens, C.c_int32), _p(err, C.c_uint8))
        want, wl, we = O.move_batch(cur, mv, L, cyclical=True)
        empty = (cur[:, :L] == 0).all(1) | (cur[:, L:] == 0).all(1)
        assert np.array_equal(err == 251, empty)  # only states with an emptied relator are outside the fast path
        assert (~empty).mean() > 0.5
        m = err != 251
        assert np.array_equal(err[m], we[m]) and np.array_equal(out[m], want[m]) and np.array_equal(lens[m], wl[m])
        cur = np.ascontiguousarray(want)
        mv = np.roll(mv, 1)


@pytest.mark.parametrize("cyc", [False, True])
@pytest.mark.parametrize("L", [2, 3, 7, 25, 29, 36, 61])
def test_normal_form_search_path_matches_oracle(shim, L, cyc):
    """apply_move_nf (what the search kernels run on every node but an unreduced root) == ACMove on states in the normal
    form ACMove itself leaves behind, for both `cyclical` values; states outside the normal form are refused (251)."""
    rng = np.random.default_rng(1700 + 2 * L + int(cyc))
    n = 6000
    st = np.zeros((n, 2 * L), np.int8)
    mv = rng.integers(0, 12, size=n).astype(np.uint8)
    for r in range(n):
        for h in (0, 1):
            hi = L if r % 4 else min(L, 4)
            w = list(rng.choice([1, -1, 2, -2], size=int(rng.integers(1, hi + 1))))
            st[r, h * L:h * L + len(w)] = w
        if r % 11 == 0:  # r1 = r0^+-1: a concatenation cancels completely -> AssertionError in the reference
            w = st[r, :L][st[r, :L] != 0]
            st[r, L:] = 0
            st[r, L:L + len(w)] = -w[::-1] if r % 2 else w
    raw_err = np.empty(n, np.uint8)
    out = np.empty_like(st); lens = np.empty((n, 2), np.int32)
    shim.shim_move_nf(_p(st, C.c_int8), _p(mv, C.c_uint8), C.c_int64(n), L, int(cyc), int(L > 32), _p(out, C.c_int8), _p(lens, C.c_int32), _p(raw_err, C.c_uint8))
    assert (raw_err == 251).mean() > 0.2  # random words are mostly NOT reduced: refused
    cur, _, err0 = O.move_batch(st, np.full(n, 4, np.uint8), L, cyclical=cyc)  # one ACMove puts both relators into normal form
    # (an unreduced input can leave an EMPTY relator without an error, utils.py:261-278: such states are outside the normal form)
    keep = (err0 == 0) & (cur[:, 0] != 0) & (cur[:, L] != 0)
    cur, mv = np.ascontiguousarray(cur[keep]), mv[keep]
    for _ in range(3):
        out = np.empty_like(cur); lens = np.empty((len(cur), 2), np.int32); err = np.empty(len(cur), np.uint8)
        shim.shim_move_nf(_p(cur, C.c_int8), _p(mv, C.c_uint8), C.c_int64(len(cur)), L, int(cyc), int(L > 32), _p(out, C.c_int8), _p(lens, C.c_int32), _p(err, C.c_uint8))
        want, wl, we = O.move_batch(cur, mv, L, cyclical=cyc)
        assert (err != 251).all()
        assert np.array_equal(err, we) and np.array_equal(out, want) and np.array_equal(lens, wl)
        assert (we != 0).any() or L > 7  # the AssertionError rows are exercised at the small widths
        ok = we == 0
        cur = np.ascontiguousarray(want[ok])
        mv = np.roll(mv[ok], 1)


# ---- keys of max_relator_length 62 .. 64 (csrc/acx_keys.h: keyops<u128x>) ----------------------------------------------------
_CODE = {-2: 0, -1: 1, 1: 2, 2: 3}


def _long_key_python(word):
    """the 128-bit key of a freely reduced word of 1 .. 64 letters, restated: the plain first letter, code[k] ^ code[k - 1] for the
    others, the field 3 as a terminator behind the last letter (none at 64 letters)"""
    n = len(word)
    codes = [_CODE[int(a)] for a in word]
    fields = [codes[0]] + [codes[k] ^ codes[k - 1] for k in range(1, n)] + ([3] if n < 64 else [])
    return sum(f << (2 * k) for k, f in enumerate(fields))


def _reduced_word(rng, n):
    w = []
    while len(w) < n:
        c = int(rng.choice([-2, -1, 1, 2]))
        if not w or w[-1] != -c:
            w.append(c)
    return w


def test_long_keys_name_reduced_words_of_up_to_64_letters(shim):
    """keyops<u128x>::make / split on the CPU: the key is the restated one, decodes to the word it was made of, and distinct words get
    distinct keys -- lengths 1 .. 64, prefixes of one another included (a word and the same word with letters appended differ only in
    where the terminator sits)"""
    rng = np.random.default_rng(64)
    seen = {}
    for trial in range(3000):
        n = int(rng.integers(1, 65)) if trial % 7 else 64 - trial % 3
        w = _reduced_word(rng, n)
        for m in {n, max(1, n - 1), max(1, n // 2)}:  # the word and two of its prefixes
            word = np.array(w[:m], np.int8)
            key2 = np.zeros(2, np.uint64)
            back_n = C.c_int32(-1)
            back = np.zeros(64, np.int8)
            shim.shim_long_key(_p(word, C.c_int8), m, _p(key2, C.c_uint64), C.byref(back_n), _p(back, C.c_int8))
            key = int(key2[0]) | (int(key2[1]) << 64)
            assert key == _long_key_python(word), (m, word.tolist())
            assert back_n.value == m and back[:m].tolist() == word.tolist() and not back[m:].any(), (m, word.tolist(), back.tolist())
            assert seen.setdefault(key, tuple(word.tolist())) == tuple(word.tolist())


@pytest.mark.parametrize("L", [62, 63, 64])
def test_moves_on_the_long_word_type_match_the_oracle(shim, L):
    """ACMove through Pres<u128x> (the word functions take the type unchanged, range-checked shifts: a shift by 64 letters is the word's
    width) against the oracle's ACMove (ac_moves.py:159-231), relators up to the full 64 letters"""
    rng = np.random.default_rng(L)
    n = 1500
    st = np.zeros((n, 2 * L), np.int8)
    for i in range(n):
        for h in (0, 1):
            m = int(rng.integers(1, L + 1)) if i % 5 else L - int(rng.integers(0, 3))
            st[i, h * L:h * L + m] = _reduced_word(rng, m)
    mv = rng.integers(0, 12, n).astype(np.uint8)
    for cyc in (False, True):
        out = np.empty_like(st)
        lens = np.empty((n, 2), np.int32)
        err = np.empty(n, np.uint8)
        shim.shim_move_long(_p(st, C.c_int8), _p(mv, C.c_uint8), C.c_int64(n), L, int(cyc), _p(out, C.c_int8), _p(lens, C.c_int32), _p(err, C.c_uint8))
        want, wlens, werr = O.move_batch(st, mv, L, cyclical=cyc)
        assert np.array_equal(err, werr)
        ok = werr == 0
        assert np.array_equal(out[ok], want[ok]) and np.array_equal(lens[ok], wlens[ok])
