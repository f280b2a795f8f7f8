// acx_bytes.h -- byte-exact AC moves on raw int8 rows (any letters, any zero pattern).
//
// This is the general path behind the Python functional surface (ACMove, concatenate_relators,
// conjugate, simplify_presentation, simplify_relator): it reproduces what the reference's NumPy code
// does to an ARBITRARY int8 array, including rows that are not valid presentations (interior zeros,
// empty relators), letters other than +-1/+-2 (the reference's own tests use up to 6) and the
// AssertionError / IndexError outcomes.  The packed path (acx_word.h) is the fast path for valid
// two-generator states; both are checked against the same oracle fixtures.
//
// One row per lane, working storage in lane-private arrays.  __host__ __device__ for the same reason
// as acx_word.h.
#pragma once
#include <stdint.h>

#include "acx_word.h"

namespace acx {

enum : int {
    ACX_F_CYCLICAL = 1,     // cyclically reduce after the move (ACMove(cyclical=True))
    ACX_F_NO_SIMPLIFY = 2,  // stop after concatenate_relators / conjugate (ac_moves.py:4-156)
    ACX_F_NO_MOVE = 4,      // simplify_presentation only (utils.py:243-280)
    ACX_F_BYTES = 8         // select the byte-exact kernel instead of the packed one
};

constexpr int kMaxBytesL = 128;  // widest relator the byte path handles

ACX_HD int8_t neg8(int8_t a) { return (int8_t)(-a); }  // int8 wrap-around like NumPy's

// nonzeros of r[0..L) -> w; returns the count; *extent = 1 + index of the last nonzero (0 if none)
ACX_HD int take_nonzero(const int8_t* r, int L, int8_t* w, int* extent) {
    int c = 0, e = 0;
    for (int k = 0; k < L; k++)
        if (r[k] != 0) {
            w[c++] = r[k];
            e = k + 1;
        }
    *extent = e;
    return c;
}

// free + cyclic reduction of the compact word w[0..n) in place (utils.py:208-229); returns new length
ACX_HD int reduce_word(int8_t* w, int n, bool cyclical) {
    int on = 0;
    for (int k = 0; k < n; k++) {  // stack pass == the reference's delete-and-step-back normal form
        if (on > 0 && w[on - 1] == neg8(w[k])) on--;
        else w[on++] = w[k];
    }
    if (cyclical && on > 0) {
        int p = 0;
        while (2 * p + 1 < on && w[p] == neg8(w[on - 1 - p])) p++;
        if (p) {
            for (int k = 0; k < on - 2 * p; k++) w[k] = w[k + p];
            on -= 2 * p;
        }
    }
    return on;
}

// Decode move id (ac_moves.py:192-206) -> is_conj, i, (j, sign) or g
ACX_HD void decode_move(int a, int& is_conj, int& i, int& negate, int& g) {
    const int m = a + 1;
    i = m & 1;
    if (a < 4) {
        is_conj = 0;
        negate = ((m - i) >> 1) & 1;
        g = 0;
    } else {
        is_conj = 1;
        const int jp = ((m - i) >> 1) & 1;
        const int sp = ((m - i - 2 * jp) >> 2) & 1;
        negate = 0;
        g = sp ? -(jp + 1) : (jp + 1);
    }
}

// One ACMove on a raw row.  in/out: 2L int8 (may alias is NOT allowed); lens: 2 ints.
// `fit` (nullable) receives the new length of r_i when the move was applied, else -1 -- what the
// Python wrappers need to rebuild the `lengths` list the raw move functions return.
// Scratch: w1, w2 of L int8 each.
ACX_HD int move_bytes(const int8_t* in, int L, int a, int flags, int8_t* out, int* lens, int* fit, int8_t* w1, int8_t* w2) {
    for (int k = 0; k < 2 * L; k++) out[k] = in[k];  // presentation.copy()
    if (fit) *fit = -1;
    if (!(flags & ACX_F_NO_MOVE)) {
        int is_conj, i, negate, g, ext;
        decode_move(a, is_conj, i, negate, g);
        int8_t* ri = out + i * L;
        if (!is_conj) {  // ac_moves.py:36-74
            const int8_t* rj = in + (1 - i) * L;
            const int n1 = take_nonzero(in + i * L, L, w1, &ext);
            int n2 = take_nonzero(rj, L, w2, &ext);
            if (negate) {  // reversed and negated (filtering zeros commutes with the reversal)
                for (int k = 0; k < n2 / 2; k++) {
                    int8_t t = w2[k];
                    w2[k] = w2[n2 - 1 - k];
                    w2[n2 - 1 - k] = t;
                }
                for (int k = 0; k < n2; k++) w2[k] = neg8(w2[k]);
            }
            int acc = 0;
            const int m = n1 < n2 ? n1 : n2;
            while (acc < m && w1[n1 - 1 - acc] == neg8(w2[acc])) acc++;
            const int nn = n1 + n2 - 2 * acc;
            if (nn <= L) {
                for (int k = 0; k < n1 - acc; k++) ri[k] = w1[k];
                for (int k = acc; k < n2; k++) ri[n1 - 2 * acc + k] = w2[k];
                for (int k = nn; k < L; k++) ri[k] = 0;
                if (fit) *fit = nn;
            }
        } else {  // ac_moves.py:108-154
            const int n = take_nonzero(in + i * L, L, w1, &ext);
            if (n == 0) return ACX_ERR_INDEX;  // relator_nonzero[0]
            const int sc = w1[0] == neg8((int8_t)g);
            const int ec = w1[n - 1] == (int8_t)g;
            const int nn = n + 2 - 2 * (sc + ec);
            if (nn <= L) {  // only the slots named by the reference are written; a dirty tail stays
                for (int k = sc; k < n - ec; k++) ri[1 - 2 * sc + k] = w1[k];
                if (!sc) ri[0] = (int8_t)g;
                if (!ec) ri[n + 1 - 2 * sc] = neg8((int8_t)g);
                if (sc && ec)
                    for (int k = i * L + nn; k < i * L + nn + 2 && k < 2 * L; k++) out[k] = 0;
                if (fit) *fit = nn;
            }
        }
    }
    if (flags & ACX_F_NO_SIMPLIFY) {
        int e;
        lens[0] = take_nonzero(out, L, w1, &e);
        lens[1] = take_nonzero(out + L, L, w1, &e);
        return ACX_ERR_NONE;
    }
    // simplify_presentation: validity assert (utils.py:261-263) then reduce each half
    int n[2], ext[2];
    n[0] = take_nonzero(out, L, w1, &ext[0]);
    n[1] = take_nonzero(out + L, L, w2, &ext[1]);
    if (n[0] == 0 || n[1] == 0 || ext[0] != n[0] || ext[1] != n[1]) return ACX_ERR_ASSERT;
    for (int h = 0; h < 2; h++) {
        int8_t* w = h ? w2 : w1;
        const int nn = reduce_word(w, n[h], (flags & ACX_F_CYCLICAL) != 0);
        for (int k = 0; k < L; k++) out[h * L + k] = k < nn ? w[k] : (int8_t)0;
        lens[h] = nn;
    }
    return ACX_ERR_NONE;
}

// simplify_relator core on a row of `width` int8 (utils.py:198-229): out = reduced word left-aligned
// and zero padded to `width`; *nz_in = letters on entry, returns the reduced length or -ACX_ERR_ASSERT
ACX_HD int simplify_row(const int8_t* in, int width, bool cyclical, int8_t* out, int* nz_in) {
    int ext;
    const int n = take_nonzero(in, width, out, &ext);
    *nz_in = n;
    if (ext != n) return -ACX_ERR_ASSERT;  // zeros must sit at the right end
    const int nn = reduce_word(out, n, cyclical);
    for (int k = nn; k < width; k++) out[k] = 0;
    return nn;
}

}  // namespace acx
