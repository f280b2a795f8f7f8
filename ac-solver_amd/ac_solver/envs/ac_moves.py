"""The twelve Andrews-Curtis moves -- drop-in for ac_solver/envs/ac_moves.py.

Every function hands its presentation to the byte-exact HIP kernel of libacx (one row, flag
ACX_F_BYTES) and reshapes the answer into the reference's return convention.  Move table
(reference: ac_moves.py:165-179):

    0  r1 <- r1 r0        1  r0 <- r0 r1^-1     2  r1 <- r1 r0^-1     3  r0 <- r0 r1
    4  r1 <- x^-1 r1 x    5  r0 <- y^-1 r0 y    6  r1 <- y^-1 r1 y    7  r0 <- x r0 x^-1
    8  r1 <- x r1 x^-1    9  r0 <- y r0 y^-1   10  r1 <- y r1 y^-1   11  r0 <- x^-1 r0 x
"""
import numpy as np

from ac_solver import _acx

_RAISES = {_acx.ERR_ASSERT: AssertionError, _acx.ERR_INDEX: IndexError, _acx.ERR_VALUE: ValueError}

# (i, j, sign) -> move id, read off the table above
_CONCAT_ID = {(1, 0, 1): 0, (0, 1, -1): 1, (1, 0, -1): 2, (0, 1, 1): 3}
_CONJ_ID = {(1, 1, -1): 4, (0, 2, -1): 5, (1, 2, -1): 6, (0, 1, 1): 7, (1, 1, 1): 8, (0, 2, 1): 9, (1, 2, 1): 10, (0, 1, -1): 11}


def _one_row(presentation, move_id, max_relator_length, flags):
    arr = np.asarray(presentation)
    out, lens, err, fit = _acx.move_rows(arr.reshape(1, -1), [move_id], max_relator_length, _acx.F_BYTES | flags)
    if err[0]:
        raise _RAISES[int(err[0])](f"move {move_id} on {arr}: the reference raises here "
                                   "(empty relator / zeros not padded to the right)")
    return out[0].astype(arr.dtype), [int(lens[0, 0]), int(lens[0, 1])], int(fit[0])


def concatenate_relators(presentation, max_relator_length, i, j, sign, lengths):
    """r_i <- r_i r_j^sign with free cancellation at the junction; unchanged when the product is longer
    than max_relator_length.  Like the reference (ac_moves.py:4-76) the caller's `lengths` list is
    updated in place and returned."""
    assert i in [0, 1] and j in [0, 1] and i == 1 - j, f"expect i and j to be 0 or 1 and i != j; got i = {i}, j = {j}"
    assert sign in [1, -1], f"expect sign to be +1 or -1, received {sign}"
    out, _, fit = _one_row(presentation, _CONCAT_ID[(i, j, sign)], max_relator_length, _acx.F_NO_SIMPLIFY)
    if fit >= 0:
        lengths[i] = fit
    return out, lengths


def conjugate(presentation, max_relator_length, i, j, sign, lengths):
    """r_i <- g r_i g^-1 for g = x_j^sign (j in {1, 2}), cancelling at most one letter at each end;
    unchanged when the result is longer than max_relator_length.  Reference: ac_moves.py:79-156
    (returns a fresh lengths list when the move applies)."""
    assert i in [0, 1] and j in [1, 2], f"expect i to be 0 and 1 and j to be 1 or 2; got i = {i}, j = {j}"
    assert sign in [1, -1], f"expect sign to be +1 or -1, received {sign}"
    out, _, fit = _one_row(presentation, _CONJ_ID[(i, j, sign)], max_relator_length, _acx.F_NO_SIMPLIFY)
    if fit >= 0:
        lengths = list(lengths)
        lengths[i] = fit
    return out, lengths


def ACMove(move_id, presentation, max_relator_length, lengths, cyclical=True):
    """Apply AC move `move_id` (0..11) and reduce both relators freely (and cyclically when `cyclical`).
    Returns (new_presentation, [len0, len1]).  Reference: ac_moves.py:159-231; `lengths` is accepted for
    signature compatibility and, as in the reference, has no influence on the result."""
    assert move_id in range(0, 12), f"Expect n to be in range 0-11 (both inclusive); got {move_id}"
    flags = _acx.F_CYCLICAL if cyclical else 0
    arr = np.asarray(presentation)
    if max_relator_length <= 64 and arr.size == 2 * max_relator_length:
        # a two-generator presentation goes through the packed kernel (the one the environments use): a third of the byte-exact
        # kernel's time per call; rows it cannot pack (other letters, zeros that are not right padding) report 250 and take the
        # byte-exact route below, which reproduces the reference on ANY int8 array
        out, len0, len1, err = _acx.move_one_row(arr.reshape(-1), int(move_id), max_relator_length, flags)
        if err == 0:
            return out.astype(arr.dtype), [len0, len1]
        if err != _acx.ERR_UNPACKABLE:
            raise _RAISES[err](f"move {move_id} on {arr}: the reference raises here (empty relator / zeros not padded to the right)")
    out, lens, _ = _one_row(presentation, int(move_id), max_relator_length, flags)
    return out, lens
