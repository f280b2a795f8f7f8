// acx_keys.h -- search keys: one machine word per relator that names (word, length).
//
// W = uint64_t (max_relator_length <= 29) and W = unsigned __int128 (<= 61): key = word | length << (bits - 6): the length sits in the
// six bits the word cannot reach.
//
// W = u128x (round 6: max_relator_length 62 .. 64; the reference takes any length, breadth_first.py:42-45, and its own Miller-Schupp
// generator reaches 64 at n = 14, search/miller_schupp/miller_schupp.py:43).  A 64-letter word fills all 128 bits and all four 2-bit
// codes are letters, so there is no room for a length -- for ARBITRARY words.  The states of a search are FREELY REDUCED words (ACMove
// simplifies both relators, ac_moves.py:224-229; a root that is not reduced is refused for these lengths), and a reduced word never holds
// a letter next to its inverse: field k of  e = w ^ (w << 2)  is code[k] ^ code[k - 1], which is 3 exactly for an inverse pair.  So
//     key(w, n) = (w' ^ (w' << 2)) restricted to fields 0 .. n,   w' = w with the INVERSE of its last letter appended at field n (n < 64)
// is the plain first letter, n - 1 fields in {0, 1, 2}, and the field 3 as a terminator behind the last letter (none when n = 64):
// injective on reduced words of 1 .. 64 letters, 128 bits.  Decoding: n = the lowest field >= 1 that holds 3 (64 if there is none),
// w = the prefix-xor of the fields (six shift-xor steps) below field n.  u128x is unsigned __int128 under another name -- it converts
// both ways, so every word function (acx_word.h) and every kernel template takes it unchanged; only keyops differs.
//
// __host__ __device__ throughout: tests/hostshim runs keyops on the CPU against a Python restatement.
#pragma once
#include "acx_word.h"

namespace acx {

struct u128x {
    u128 v;
    u128x() = default;
    ACX_HD constexpr u128x(u128 x) : v(x) {}
    ACX_HD constexpr operator u128() const { return v; }
    ACX_HD u128x& operator&=(u128 x) { v &= x; return *this; }
    ACX_HD u128x& operator|=(u128 x) { v |= x; return *this; }
    ACX_HD u128x& operator^=(u128 x) { v ^= x; return *this; }
    ACX_HD u128x& operator<<=(int s) { v <<= s; return *this; }
    ACX_HD u128x& operator>>=(int s) { v >>= s; return *this; }
};
template <> struct wtraits<u128x> : wtraits<u128> {};

template <typename W> struct is_long_key { static constexpr bool value = false; };
template <> struct is_long_key<u128x> { static constexpr bool value = true; };

template <typename W> struct keyops {
    static constexpr int kShift = wtraits<W>::kBits - 6;
    static constexpr int kMaxL = kShift / 2;  // 29 / 61
    static ACX_HD W make(W w, int n) { return w | ((W)n << kShift); }
    static ACX_HD int len(W k) { return (int)(uint32_t)(k >> kShift); }
    static ACX_HD W word(W k) { return k & (((W)1 << kShift) - 1); }
    static ACX_HD void split(W k, W& w, int& n) {
        w = word(k);
        n = len(k);
    }
};

template <> struct keyops<u128x> {
    static constexpr int kMaxL = 64;
    static ACX_HD u128x make(u128x w, int n) {  // w freely reduced, 1 <= n <= 64 (0: the empty word, never a stored state)
        u128 x = w;
        if (n <= 0) return u128x((u128)0);
        if (n < 64) x |= (u128)((((uint32_t)(x >> (2 * (n - 1)))) & 3u) ^ 3u) << (2 * n);
        u128 e = x ^ (x << 2);
        if (n < 63) e &= (((u128)1 << (2 * (n + 1))) - 1);
        return u128x(e);
    }
    static ACX_HD void split(u128x k, u128x& w, int& n) {
        const u128 e = k;
        if (e == 0) {
            w = u128x((u128)0);
            n = 0;
            return;
        }
        u128 t = e & (e >> 1) & wtraits<u128>::lo_ones();  // bit 2 f set: field f holds 3
        t &= ~(u128)1;                                       // (field 0 is the plain first letter)
        n = t ? (wtraits<u128>::ctz(t) >> 1) : 64;
        u128 x = e;
        x ^= x << 2;
        x ^= x << 4;
        x ^= x << 8;
        x ^= x << 16;
        x ^= x << 32;
        x ^= x << 64;
        w = u128x(n < 64 ? (x & ((((u128)1) << (2 * n)) - 1)) : x);
    }
    static ACX_HD int len(u128x k) {
        u128x w;
        int n;
        split(k, w, n);
        return n;
    }
    static ACX_HD u128x word(u128x k) {
        u128x w;
        int n;
        split(k, w, n);
        return w;
    }
};

}  // namespace acx
