"""Helpers on balanced two-relator presentations -- drop-in for ac_solver/envs/utils.py.

A presentation is a 1-D integer array of even length 2L: two relators over the letters
{+1 (x), -1 (x^-1), +2 (y), -2 (y^-1)}, each left aligned and right padded with zeros
(reference: ac_solver/envs/utils.py:4-8).

Format checks and converters are host-side bookkeeping; the reductions (`simplify_relator`,
`simplify_presentation`) run in the byte-exact HIP kernel of libacx (csrc/acx_bytes.h), which
reproduces the reference's behaviour on arbitrary int8 input, including its exceptions.
"""
import numpy as np

from ac_solver import _acx

_RAISES = {_acx.ERR_ASSERT: AssertionError, _acx.ERR_INDEX: IndexError, _acx.ERR_VALUE: ValueError}


def _halves(array):
    L = len(array) // 2
    return array[:L], array[L:]


def is_array_valid_presentation(array):
    """True iff `array` has even length and each half is a non-empty word followed only by zeros.
    Reference: utils.py:13-54 (e.g. [1,2,0,0,-2,-1,0,0] is valid, [1,0,2,0,-2,-1,0,0] is not)."""
    assert isinstance(array, (list, np.ndarray)), f"array must be a list or a numpy array, got {type(array)}"
    arr = np.asarray(array)
    if arr.ndim != 1 or len(arr) % 2:
        return False
    for half in _halves(arr):
        n = int(np.count_nonzero(half))
        if n == 0 or np.any(half[n:] != 0):  # a zero inside the word pushes a letter past position n
            return False
    return True


def are_rows_valid_presentations(rows):
    """`is_array_valid_presentation` for every row of a 2-D array at once (ACVecEnv checks 10^5..10^6 initial states): -> bool
    array.  Same predicate: each half of a row has at least one letter and zeros only behind its last letter."""
    rows = np.asarray(rows)
    assert rows.ndim == 2 and rows.shape[1] % 2 == 0
    L = rows.shape[1] // 2
    ok = np.ones(rows.shape[0], dtype=bool)
    pos = np.arange(L)[None, :]
    for half in (rows[:, :L], rows[:, L:]):
        nz = half != 0
        n = nz.sum(axis=1)
        ok &= (n > 0) & ~(nz & (pos >= n[:, None])).any(axis=1)
    return ok


def is_presentation_trivial(presentation):
    """True iff the presentation is one of the eight length-2 trivial ones <x^+-1, y^+-1> / <y^+-1, x^+-1>.
    Reference: utils.py:57-87."""
    if not is_array_valid_presentation(presentation):
        return False
    arr = np.asarray(presentation)
    r0, r1 = _halves(arr)
    if np.count_nonzero(r0) != 1 or np.count_nonzero(r1) != 1:
        return False
    return sorted((abs(int(r0[0])), abs(int(r1[0])))) == [1, 2]


def generate_trivial_states(max_relator_length):
    """The 8 trivial states as an (8, 2L) array, ordered generator-first then the two signs
    (reference: utils.py:91-114)."""
    L = max_relator_length
    states = np.zeros((8, 2 * L), dtype=np.int64)
    row = 0
    for gen in (1, 2):
        for s0 in (-1, 1):
            for s1 in (-1, 1):
                states[row, 0] = s0 * gen
                states[row, L] = s1 * (3 - gen)
                row += 1
    return states


def convert_relators_to_presentation(relator1, relator2, max_relator_length):
    """Two zero-free lists -> int8 presentation of width 2*max_relator_length (reference: utils.py:117-145)."""
    assert 0 not in relator1 and 0 not in relator2, "relator1 and relator2 must not be padded with zeros."
    assert max_relator_length >= max(len(relator1), len(relator2)), \
        "max_relator_length must be greater than or equal to the lengths of relator1 and rel2."
    assert isinstance(relator1, list) and isinstance(relator2, list), \
        f"got types {type(relator1)} for relator1 and {type(relator2)} for relator2"
    out = np.zeros(2 * max_relator_length, dtype=np.int8)
    out[: len(relator1)] = relator1
    out[max_relator_length: max_relator_length + len(relator2)] = relator2
    return out


def change_max_relator_length_of_presentation(presentation, new_max_length):
    """Re-embed a presentation at another relator width.  Like the reference (utils.py:148-172) this
    needs a Python list: the words are handed to convert_relators_to_presentation, which insists on lists."""
    old = len(presentation) // 2
    n0 = int(np.count_nonzero(presentation[:old]))
    n1 = int(np.count_nonzero(presentation[old:]))
    return convert_relators_to_presentation(presentation[:n0], presentation[old: old + n1], new_max_length)


def simplify_relator(relator, max_relator_length, cyclical=False, padded=True):
    """Free (and optionally cyclic) reduction of one word.  Reference: utils.py:175-240.

    Returns (array, length).  With padded=True the array has max_relator_length entries; otherwise it is
    what the reference's np.delete calls leave: the reduced word plus whatever zero padding came in.
    """
    assert isinstance(relator, np.ndarray), "expect relator to be a numpy array"
    width = len(relator)
    if width == 0:
        n = nz = 0
        word = relator[:0]
    else:
        out, lens, err = _acx.simplify_rows(relator.reshape(1, width), cyclical)
        if err[0]:
            raise _RAISES[int(err[0])]("expect all zeros to be at the right end")
        n, nz = int(lens[0, 0]), int(lens[0, 1])
        word = out[0, :n].astype(relator.dtype)
    kept = width - (nz - n)  # array length after the cancelled letters were deleted
    if padded:
        if max_relator_length - kept < 0:
            raise ValueError("index can't contain negative values")  # what np.pad raises in the reference
        total = max_relator_length
    else:
        total = kept
    assert max_relator_length >= n, "Increase max length! Length of simplified word is bigger than maximum allowed length."
    result = np.zeros(total, dtype=relator.dtype)
    result[:n] = word
    return result, n


def simplify_presentation(presentation, max_relator_length, lengths_of_words, cyclical=True):
    """Reduce both relators of a presentation.  Reference: utils.py:243-280 (the incoming lengths are
    ignored there too: they are recomputed)."""
    arr = np.array(presentation)
    out, lens, err, _ = _acx.move_rows(arr.reshape(1, -1), None, max_relator_length,
                                       _acx.F_BYTES | _acx.F_NO_MOVE | (_acx.F_CYCLICAL if cyclical else 0))
    if err[0]:
        raise _RAISES[int(err[0])](f"{arr} is not a valid presentation. Expect all zeros to be padded to the right.")
    return out[0].astype(arr.dtype), [int(lens[0, 0]), int(lens[0, 1])]
