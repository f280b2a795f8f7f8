// acx_word.h -- packed-word arithmetic for two-generator Andrews-Curtis presentations.
//
// A relator over {x, x^-1, y, y^-1} is held as (w, n): letter k occupies bits [2k, 2k+1] of w,
// n is the length, bits above 2n are zero.  The 2-bit code is ORDER PRESERVING on the int8 letters
// of the reference (-2 -> 0, -1 -> 1, +1 -> 2, +2 -> 3) so that inverse(letter) = code ^ 3 and
// tuple comparison of int8 states can be done on codes.  W = uint64_t carries L <= 32 letters,
// W = unsigned __int128 carries L <= 64.
//
// With this layout the reference's word operations become a handful of shifts and masks:
//   inverse word        reverse the 2-bit groups, complement              (ac_moves.py:41-48)
//   junction cancel     ctz(inv(w1) ^ w2) / 2                             (ac_moves.py:56-60)
//   concatenation       w1 | (w2 >> 2acc) << 2(n1-acc)                    (ac_moves.py:62-74)
//   conjugation         compare first / last code with g                  (ac_moves.py:119-154)
//   "is freely reduced" any 2-bit field of w ^ (w >> 2) equal to 3        (utils.py:208-217)
//   cyclic reduction    ctz(w ^ inv(w)) / 2 letters off both ends         (utils.py:220-229)
//
// Everything here is __host__ __device__ so that tests/hostshim can run the very same code on the
// CPU against the oracle (test infrastructure); the product only ever runs it inside HIP kernels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define ACX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define ACX_HD inline
#endif

namespace acx {

typedef unsigned __int128 u128;

enum : int {
    ACX_ERR_NONE = 0,
    ACX_ERR_ASSERT = 1,      // the reference raises AssertionError (invalid presentation after the move)
    ACX_ERR_INDEX = 2,       // the reference raises IndexError (conjugating an empty relator)
    ACX_ERR_VALUE = 3,       // the reference raises ValueError (np.pad with a negative width)
    ACX_ERR_UNPACKABLE = 250 // row is not a valid {0,+-1,+-2} presentation: use the byte path
};

ACX_HD int code_of_letter(int a) { return a < 0 ? a + 2 : a + 1; }   // a in {-2,-1,1,2}
ACX_HD int letter_of_code(int c) { return c < 2 ? c - 2 : c - 1; }

// ---- width-specific primitives -------------------------------------------------------------
template <typename W> struct wtraits;

template <> struct wtraits<uint64_t> {
    static constexpr int kBits = 64;
    static constexpr int kMaxLetters = 32;
    static ACX_HD int ctz(uint64_t w) { return __builtin_ctzll(w); }  // w != 0
    static ACX_HD uint64_t rev2(uint64_t w) {                         // reverse the 32 two-bit groups
        uint64_t r = __builtin_bitreverse64(w);
        return ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    }
    static ACX_HD uint64_t lo_ones() { return 0x5555555555555555ull; }
};

template <> struct wtraits<u128> {
    static constexpr int kBits = 128;
    static constexpr int kMaxLetters = 64;
    static ACX_HD int ctz(u128 w) {
        uint64_t lo = (uint64_t)w;
        return lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll((uint64_t)(w >> 64));
    }
    static ACX_HD u128 rev2(u128 w) {
        return ((u128)wtraits<uint64_t>::rev2((uint64_t)w) << 64) | wtraits<uint64_t>::rev2((uint64_t)(w >> 64));
    }
    static ACX_HD u128 lo_ones() { return ((u128)0x5555555555555555ull << 64) | 0x5555555555555555ull; }
};

// Shifts by a letter count.  SAFE = true tolerates counts that reach the full width (needed only when
// max_relator_length equals the word capacity, 32 or 64); SAFE = false is the plain hardware shift.
template <typename W, bool SAFE = true> ACX_HD W shl(W w, int letters) {
    if (SAFE) return 2 * letters >= wtraits<W>::kBits ? (W)0 : (W)(w << (2 * letters));
    return (W)(w << (2 * letters));
}
template <typename W, bool SAFE = true> ACX_HD W shr(W w, int letters) {
    if (SAFE) return 2 * letters >= wtraits<W>::kBits ? (W)0 : (W)(w >> (2 * letters));
    return (W)(w >> (2 * letters));
}
template <typename W, bool SAFE = true> ACX_HD W mask(int letters) { return (W)(shl<W, SAFE>((W)1, letters) - 1); }  // letters == capacity -> all ones
template <typename W, bool SAFE = true> ACX_HD int get(W w, int k) { return (int)((uint32_t)shr<W, SAFE>(w, k) & 3u); }

// inverse word: reversed and letter-wise inverted (code ^ 3)
template <typename W, bool SAFE = true> ACX_HD W inv(W w, int n) {
    // n == 0: the word is 0 and so is its reversal; the shift count is clamped to stay defined
    const int sh = wtraits<W>::kBits - 2 * n;
    W r = wtraits<W>::rev2(w) >> (sh >= wtraits<W>::kBits ? wtraits<W>::kBits - 2 : sh);
    return n == 0 ? (W)0 : (W)(r ^ mask<W, SAFE>(n));
}

// number of leading letters on which a and b agree, capped at `cap`
template <typename W> ACX_HD int common_prefix(W a, W b, int cap) {
    W t = a ^ b;
    int cp = t ? (wtraits<W>::ctz(t) >> 1) : wtraits<W>::kMaxLetters;
    return cp < cap ? cp : cap;
}

// true when some adjacent pair of the n-letter word is mutually inverse
template <typename W, bool SAFE = true> ACX_HD bool has_inverse_pair(W w, int n) {
    W t = w ^ (w >> 2);  // field k = code[k] ^ code[k+1]; inverse pair <=> field == 3
    return n >= 2 && (t & (t >> 1) & wtraits<W>::lo_ones() & mask<W, SAFE>(n - 1)) != 0;
}

// free reduction (utils.py:208-217): stack pass; the freely reduced form is unique, so this equals
// the reference's delete-and-step-back loop.  States produced by ACMove are already reduced, so the
// loop only runs for unreduced initial states.
template <typename W, bool SAFE = true> ACX_HD void free_reduce(W& w, int& n) {
    if (!has_inverse_pair<W, SAFE>(w, n)) return;
    W o = 0;
    int on = 0;
    for (int k = 0; k < n; k++) {
        int c = get<W, SAFE>(w, k);
        if (on > 0 && get<W, SAFE>(o, on - 1) == (c ^ 3)) {
            on--;
            o &= mask<W, SAFE>(on);
        } else {
            o |= shl<W, SAFE>((W)c, on);
            on++;
        }
    }
    w = o;
    n = on;
}

// cyclic reduction of a freely reduced word (utils.py:220-229), branch free: p letters leave each end,
// p = common prefix of w and w^-1, which for a reduced word stops before the middle
template <typename W, bool SAFE = true> ACX_HD void cyclic_reduce(W& w, int& n) {
    const W t = w ^ inv<W, SAFE>(w, n);  // non-zero for a reduced non-empty word
    int p = t ? (wtraits<W>::ctz(t) >> 1) : 0;
    p = 2 * p < n ? p : 0;  // unreachable for reduced words; keeps the shifts defined
    w = shr<W, SAFE>(w, p) & mask<W, SAFE>(n - 2 * p);
    n -= 2 * p;
}

// A packed presentation.  Plain scalar fields (no arrays): runtime-indexed members would be spilled
// to scratch / LDS by hipcc.
template <typename W> struct Pres {
    W w0, w1;
    int n0, n1;
};

// r1 r2^{+-1} with junction cancellation (ac_moves.py:4-76): (w1,n1) is r_i, (w2,n2) is r_j.
// Returns false when the product does not fit (outputs untouched).
template <typename W, bool SAFE = true> ACX_HD bool concat_words(W w1, int n1, W w2, int n2, bool negate, int L, W& out, int& nout) {
    const W w2i = inv<W, SAFE>(w2, n2);
    w2 = negate ? w2i : w2;
    const int acc = common_prefix<W>(inv<W, SAFE>(w1, n1), w2, n1 < n2 ? n1 : n2);
    const int nn = n1 + n2 - 2 * acc;
    const bool fits = nn <= L;
    const W r = (w1 & mask<W, SAFE>(n1 - acc)) | shl<W, SAFE>(shr<W, SAFE>(w2, acc), n1 - acc);
    out = fits ? r : out;
    nout = fits ? nn : nout;
    return fits;
}

// g r g^-1 with end cancellation (ac_moves.py:79-156); gc is the code of g; n >= 1
template <typename W, bool SAFE = true> ACX_HD bool conjugate_word(W w, int n, int gc, int L, W& out, int& nout) {
    const int sc = get<W, SAFE>(w, 0) == (gc ^ 3);
    const int ec = get<W, SAFE>(w, n - 1) == gc;
    const int nn = n + 2 - 2 * (sc + ec);
    const bool fits = nn <= L;
    const int nb = n - sc - ec;  // letters of r that survive
    W r = shr<W, SAFE>(w, sc) & mask<W, SAFE>(nb);
    r = sc ? r : (W)((r << 2) | (W)gc);
    r |= ec ? (W)0 : shl<W, SAFE>((W)(gc ^ 3), fits ? nn - 1 : 0);
    out = fits ? r : out;
    nout = fits ? nn : nout;
    return fits;
}

// ACMove (ac_moves.py:159-231) on a packed presentation (zero-padded words over {+-1,+-2}; a relator
// may be empty, as the reference's own outputs can be when the input was not freely reduced).
// Returns ACX_ERR_NONE, ACX_ERR_INDEX (conjugating an empty relator, ac_moves.py:119) or
// ACX_ERR_ASSERT (a relator is empty after the move: the validity assert of simplify_presentation,
// utils.py:261-263); on error the state is left unchanged.
template <typename W, bool SAFE = true> ACX_HD int apply_move(Pres<W>& s, int a, int L, bool cyclical) {
    const int m = a + 1;
    const bool i1 = (m & 1) != 0;  // ac_moves.py:192-206: odd ids touch r_1
    const int i = i1 ? 1 : 0;
    W wi = i1 ? s.w1 : s.w0, wj = i1 ? s.w0 : s.w1;
    int ni = i1 ? s.n1 : s.n0, nj = i1 ? s.n0 : s.n1;
    if (a < 4) {
        concat_words<W, SAFE>(wi, ni, wj, nj, (((m - i) >> 1) & 1) != 0, L, wi, ni);
    } else {
        if (ni == 0) return ACX_ERR_INDEX;
        const int jp = ((m - i) >> 1) & 1;
        const int sp = ((m - i - 2 * jp) >> 2) & 1;
        conjugate_word<W, SAFE>(wi, ni, sp ? 1 - jp : 2 + jp, L, wi, ni);  // g = -(jp+1) -> code 1-jp ; +(jp+1) -> 2+jp
    }
    if (ni == 0 || nj == 0) return ACX_ERR_ASSERT;
    // simplify_presentation, utils.py:267-278 (both relators, also the untouched one)
    free_reduce<W, SAFE>(wi, ni);
    free_reduce<W, SAFE>(wj, nj);
    if (cyclical) {
        cyclic_reduce<W, SAFE>(wi, ni);
        cyclic_reduce<W, SAFE>(wj, nj);
    }
    s.w0 = i1 ? wj : wi;
    s.w1 = i1 ? wi : wj;
    s.n0 = i1 ? nj : ni;
    s.n1 = i1 ? ni : nj;
    return ACX_ERR_NONE;
}

// true when the word is freely AND cyclically reduced (the normal form every ACMove(cyclical=True) leaves)
template <typename W, bool SAFE = true> ACX_HD bool is_cyc_reduced(W w, int n) {
    return !has_inverse_pair<W, SAFE>(w, n) && (n < 2 || get<W, SAFE>(w, 0) != (get<W, SAFE>(w, n - 1) ^ 3));
}

// ACMove(cyclical=True) when BOTH relators are already freely and cyclically reduced and non-empty -- the
// steady state of ACEnv (every state ACEnv.step produces is in this form).  Equivalent to apply_move(..., true):
//   * the untouched relator is a fixed point of simplify_relator;
//   * a concatenation of reduced words is freely reduced once the junction is cancelled, so only its
//     cyclic reduction remains;
//   * a conjugation g r g^-1 of a cyclically reduced r either cancels at exactly one end (a rotation of r by
//     one letter, still reduced) or at none (then the cyclic reduction strips g and g^-1 again, or the result
//     does not fit: unchanged either way); cancelling at both ends would contradict cyclic reducedness.
template <typename W, bool SAFE = true> ACX_HD int apply_move_reduced(Pres<W>& s, int a, int L) {
    const int m = a + 1;
    const bool i1 = (m & 1) != 0;
    const int i = i1 ? 1 : 0;
    W wi = i1 ? s.w1 : s.w0;
    const W wj = i1 ? s.w0 : s.w1;
    int ni = i1 ? s.n1 : s.n0;
    const int nj = i1 ? s.n0 : s.n1;
    if (a < 4) {
        concat_words<W, SAFE>(wi, ni, wj, nj, (((m - i) >> 1) & 1) != 0, L, wi, ni);
        if (ni == 0) return ACX_ERR_ASSERT;
        cyclic_reduce<W, SAFE>(wi, ni);
    } else {
        const int jp = ((m - i) >> 1) & 1;
        const int sp = ((m - i - 2 * jp) >> 2) & 1;
        const int gc = sp ? 1 - jp : 2 + jp;
        const bool sc = get<W, SAFE>(wi, 0) == (gc ^ 3);
        const bool ec = get<W, SAFE>(wi, ni - 1) == gc;
        const W left = (wi >> 2) | shl<W, SAFE>((W)(gc ^ 3), ni - 1);   // r[1:] + [g^-1]
        const W right = ((wi << 2) | (W)gc) & mask<W, SAFE>(ni);        // [g] + r[:-1]
        wi = sc == ec ? wi : (sc ? left : right);
    }
    s.w0 = i1 ? wj : wi;
    s.w1 = i1 ? wi : wj;
    s.n0 = i1 ? nj : ni;
    s.n1 = i1 ? ni : nj;
    return ACX_ERR_NONE;
}

// true when both relators are non-empty and in the normal form every ACMove leaves behind: freely reduced, and
// cyclically reduced when `cyclical`
template <typename W, bool SAFE = true> ACX_HD bool is_normal_form(const Pres<W>& s, bool cyclical) {
    if (s.n0 == 0 || s.n1 == 0) return false;
    if (cyclical) return is_cyc_reduced<W, SAFE>(s.w0, s.n0) && is_cyc_reduced<W, SAFE>(s.w1, s.n1);
    return !has_inverse_pair<W, SAFE>(s.w0, s.n0) && !has_inverse_pair<W, SAFE>(s.w1, s.n1);
}

// ACMove on a presentation in normal form (is_normal_form) -- every node of a search except possibly its root.
// Equivalent to apply_move: the untouched relator is a fixed point of simplify_relator; a concatenation of freely
// reduced words is freely reduced once the junction is cancelled (concat_words); g r g^-1 with the end cancellation of
// conjugate_word is freely reduced when r is (the two ends cannot both cancel against a reduced r of length <= 2); with
// `cyclical` see apply_move_reduced.
template <typename W, bool SAFE = true> ACX_HD int apply_move_nf(Pres<W>& s, int a, int L, bool cyclical) {
    if (cyclical) return apply_move_reduced<W, SAFE>(s, a, L);
    const int m = a + 1;
    const bool i1 = (m & 1) != 0;
    const int i = i1 ? 1 : 0;
    W wi = i1 ? s.w1 : s.w0;
    const W wj = i1 ? s.w0 : s.w1;
    int ni = i1 ? s.n1 : s.n0;
    const int nj = i1 ? s.n0 : s.n1;
    if (a < 4) {
        concat_words<W, SAFE>(wi, ni, wj, nj, (((m - i) >> 1) & 1) != 0, L, wi, ni);
        if (ni == 0) return ACX_ERR_ASSERT;
    } else {
        const int jp = ((m - i) >> 1) & 1;
        const int sp = ((m - i - 2 * jp) >> 2) & 1;
        conjugate_word<W, SAFE>(wi, ni, sp ? 1 - jp : 2 + jp, L, wi, ni);
    }
    s.w0 = i1 ? wj : wi;
    s.w1 = i1 ? wi : wj;
    s.n0 = i1 ? nj : ni;
    s.n1 = i1 ? ni : nj;
    return ACX_ERR_NONE;
}

// 4 int8 letters (one per byte, little endian) from 8 code bits; bytes at positions >= valid read 0.
// One v_perm_b32 does the code -> letter lookup: selector bytes 0..3 pick a byte of 0x0201FFFE
// (= letters -2,-1,+1,+2), selector 0x0c yields 0x00 (padding).
ACX_HD uint32_t letters4(uint32_t c8, int valid) {
    uint32_t x = c8 & 0xffu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    const int v = valid < 0 ? 0 : (valid > 4 ? 4 : valid);
    const uint32_t sel = x | (uint32_t)(0x0c0c0c0cull << (8 * v));
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, 0x0201FFFEu, sel);
#else
    uint32_t out = 0;
    for (int b = 0; b < 4; b++) {
        const uint32_t q = (sel >> (8 * b)) & 0xffu;
        const uint32_t byte = q >= 12 ? (q == 12 ? 0x00u : 0xffu) : ((0x0201FFFEu >> (8 * (q & 3))) & 0xffu);
        out |= byte << (8 * b);
    }
    return out;
#endif
}

// dword j (letters 4j .. 4j+3) of the zero-padded int8 image of a relator
template <typename W> ACX_HD uint32_t relator_dword(W w, int n, int j) {
    return 8 * j >= wtraits<W>::kBits ? 0u : letters4((uint32_t)(w >> (8 * j)), n - 4 * j);
}

// dword j (bytes 4j .. 4j+3) of the 2L-byte observation row: r0 zero-padded to L, then r1 zero-padded to L.
// Bytes past the row read 0.  With a compile-time L every branch folds.
template <typename W> ACX_HD uint32_t row_dword(W w0, int n0, W w1, int n1, int L, int j) {
    const int b = 4 * j;
    if (b + 4 <= L) return relator_dword<W>(w0, n0, j);
    if (b >= L) {
        const int off = b - L, q = off >> 2, sh = off & 3;
        const uint32_t lo = relator_dword<W>(w1, n1, q);
        if (sh == 0) return lo;
        const uint32_t hi = relator_dword<W>(w1, n1, q + 1);
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh));
    }
    return relator_dword<W>(w0, n0, j) | (relator_dword<W>(w1, n1, 0) << (8 * (L - b)));  // the row's middle: r0 ends, r1 starts
}

// ---- int8 row <-> packed ------------------------------------------------------------------------
// Pack one relator from L int8 letters.  Returns false if the row is not a right-padded word over
// {+-1, +-2} (interior zero or foreign letter).
template <typename W> ACX_HD bool pack_relator(const int8_t* r, int L, W& w, int& n) {
    W o = 0;
    int len = 0;
    bool ok = true, ended = false;
    for (int k = 0; k < L; k++) {
        int a = r[k];
        if (a == 0) {
            ended = true;
        } else {
            ok = ok && !ended && a >= -2 && a <= 2;
            o |= shl<W>((W)(code_of_letter(a) & 3), len);
            len++;
        }
    }
    w = o;
    n = len;
    return ok;
}

template <typename W> ACX_HD void unpack_relator(W w, int n, int L, int8_t* r) {
    for (int k = 0; k < L; k++) r[k] = k < n ? (int8_t)letter_of_code(get<W>(w, k)) : (int8_t)0;
}

// 8 letters starting at letter `k0` (< capacity) as 8 int8 in one u64 (little endian), zero beyond n.
template <typename W> ACX_HD uint64_t unpack8(W w, int n, int k0) {
    uint64_t x = (uint64_t)(w >> (2 * k0)) & 0xFFFFull;             // 8 codes
    x = (x | (x << 24)) & 0x000000FF000000FFull;
    x = (x | (x << 12)) & 0x000F000F000F000Full;
    x = (x | (x << 6)) & 0x0303030303030303ull;                     // one code per byte
    uint64_t hi = (x >> 1) & 0x0101010101010101ull;                 // code >= 2
    x = ((x + hi + 0x7E7E7E7E7E7E7E7Eull) ^ 0x8080808080808080ull); // code -> letter: {0,1,2,3} -> {-2,-1,1,2}
    int cnt = n - k0;
    cnt = cnt < 0 ? 0 : (cnt > 8 ? 8 : cnt);
    uint64_t m = cnt >= 8 ? ~0ull : ((1ull << (8 * cnt)) - 1);
    return x & m;
}

// 8 consecutive entries of the 2L-entry int8 row (r0 padded to L, then r1 padded to L) starting at
// position p; entries past the row read as 0.  The branches depend only on p and L (wave uniform).
template <typename W> ACX_HD uint64_t row8(W w0, int n0, W w1, int n1, int L, int p) {
    if (p >= L) return p - L < L ? unpack8<W>(w1, n1, p - L) : 0ull;
    uint64_t x = unpack8<W>(w0, n0, p);
    const int cnt = L - p;
    if (cnt < 8) x |= unpack8<W>(w1, n1, 0) << (8 * cnt);
    return x;
}

// Signed-tuple order of two packed presentations as the reference's Python tuples compare them
// (greedy.py:104-113: zeros pad each half, -2 < -1 < 0 < 1 < 2).  Returns <0, 0, >0.
template <typename W> ACX_HD int compare_relator(W a, int na, W b, int nb) {
    W t = a ^ b;
    int m = na < nb ? na : nb;
    int cp = t ? (wtraits<W>::ctz(t) >> 1) : wtraits<W>::kMaxLetters;
    if (cp < m) return get<W>(a, cp) < get<W>(b, cp) ? -1 : 1;
    if (na == nb) return 0;
    // the shorter word is a prefix of the longer: compare its padding 0 with the longer word's next letter
    if (na < nb) return get<W>(b, na) >= 2 ? -1 : 1;  // 0 < positive letter
    return get<W>(a, nb) >= 2 ? 1 : -1;
}

template <typename W> ACX_HD int compare_pres(const Pres<W>& a, const Pres<W>& b) {
    int c = compare_relator<W>(a.w0, a.n0, b.w0, b.n0);
    return c ? c : compare_relator<W>(a.w1, a.n1, b.w1, b.n1);
}

}  // namespace acx
