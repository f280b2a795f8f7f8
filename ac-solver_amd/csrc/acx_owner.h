// acx_owner.h -- which rank owns a state of the sharded BFS (acx_shard.hip; ac_solver/search/sharded.py:owner_of is the same
// arithmetic in Python).  Round 5.
//
// Every AC move rewrites ONE relator (ac_moves.py:192-229): r_i <- r_i r_j^{+-1} for the action ids 0..3, r_i <- g r_i g^-1
// for 4..11, and the simplification behind it (utils.py:267-278) leaves a relator that is already in normal form alone.  Write a
// freely reduced relator as r = u c u^-1 with c cyclically reduced.  A conjugation by a generator does not change the CONJUGACY
// CLASS of r_i -- c read as a cyclic word -- and it adds or removes a letter at the FRONT of u (or rotates c when u is empty).  So
//
//     owner(state) = scale(mix(class_hash(r_0) + class_hash(r_1) + K0 inner(r_0) + K1 inner(r_1)), world)
//
// with class_hash a function of the cyclic word only and inner(r) the LAST letter of u (the one next to the core; "none" when u is
// empty) sends most of the eight conjugation children of a node to the node's own rank -- inner(r) only changes while |u| <= 1 --;
// the four concatenation children can leave, and at L = 25 most of those do not fit and are dropped as unchanged.
//
// Measured on the reference's BFS order (tools/owner_balance.cpp; AK(3) at L = 25, 3e7 nodes / Miller-Schupp n = 7 at L = 36, 1e7
// nodes; 8 ranks):                                  children that cross the exchange   nodes per rank   parents per chunk and rank
//                                                   (of those k_shard_expand routes)     max / mean           max / mean
//     hash of the whole key (rounds 1-4)                    87.5 %                       1.00                 1.00
//     hash of r_0 alone                                     43 %                         1.16 / 1.16          --
//     class hashes alone                                    5.5 % / 2.7 %                1.10 / 1.33          1.16 / 1.42
//     class hashes + inner letters (this file)              27 %  / 23 %                 1.01 / 1.07          1.04 / 1.08
//     ... + the two innermost letters of u                  43 %  / 38 %                 1.01 / 1.02          1.02 / 1.04
// The class hashes alone keep whole orbits {(u_0 c_0 u_0^-1, u_1 c_1 u_1^-1)} on one rank: too lumpy, the slowest rank of a chunk
// sets its time.  One inner letter per relator splits every orbit ~16 ways for a quarter of the children on the wire.
//
// Correctness never depends on WHICH function this is, only on it being a function of the key (equal keys meet in one table): the
// result of a sharded search is the same for every partition, which the 1 / 2 / 3 / 4 / 8-rank tests check.  What the engine
// additionally uses is the invariance (a child made by a conjugation inherits its parent's class hashes without recomputing them;
// only the inner letter of the relator it rewrote is looked up again); acx_shard_check_owners recomputes every node's owner from
// its key and the GPU tests require zero mismatches.
//
// class_hash: the multiset of the cyclic word's bigrams (letter, cyclic successor), twelve popcounts (an inverse pair cannot be
// adjacent in a cyclically reduced word), combined with odd 32-bit constants; the length enters as well.  Rotation invariant by
// construction; ~160 vector instructions for 64-bit words, against ~250 for the smallest rotation and with the same balance.
#pragma once
#include "acx_word.h"

namespace acx {

ACX_HD int popc_w(uint64_t w) { return __builtin_popcountll(w); }
ACX_HD int popc_w(u128 w) { return __builtin_popcountll((uint64_t)w) + __builtin_popcountll((uint64_t)(w >> 64)); }

// the word with every 2-bit field of `lo`'s positions holding the code c (c = 0 .. 3, a compile-time constant after unrolling)
template <typename W> ACX_HD W spread_code(W lo, int c) { return (W)(((c & 1) ? lo : (W)0) | ((c & 2) ? (W)(lo << 1) : (W)0)); }

template <typename W, bool SAFE = true> ACX_HD uint32_t class_hash(W w, int n) {
    cyclic_reduce<W, SAFE>(w, n);
    if (n <= 0) return 0u;
    const W m = mask<W, SAFE>(n), lo = (W)(wtraits<W>::lo_ones() & m);
    const W y = (W)((w >> 2) | shl<W, SAFE>((W)(w & 3), n - 1));  // field k = the cyclic successor of letter k
    W I[4], J[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const W t = w ^ spread_code<W>(lo, c), u = y ^ spread_code<W>(lo, c);
        I[c] = (W)(~(t | (t >> 1)) & lo);  // bit 2k set: letter k has code c
        J[c] = (W)(~(u | (u >> 1)) & lo);
    }
    constexpr uint32_t K[16] = {0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u, 0x9E3779B9u,
                                0x7F4A7C15u, 0x94D049BBu, 0xBF58476Du, 0x1CE4E5B9u, 0x2545F491u, 0x4F6CDD1Du, 0x6C62272Fu, 0x07BB0143u};
    uint32_t h = (uint32_t)n * 0x9E3779B1u;
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            if ((p ^ q) != 3) h += (uint32_t)popc_w((W)(I[p] & J[q])) * K[4 * p + q];
    return h;
}

// letters that the cyclic reduction strips from each end of (w, n) = |u| for r = u c u^-1 (acx_word.h: cyclic_reduce)
template <typename W, bool SAFE = true> ACX_HD int conj_prefix(W w, int n) {
    const W t = w ^ inv<W, SAFE>(w, n);
    const int p = t ? (wtraits<W>::ctz(t) >> 1) : 0;
    return 2 * p < n ? p : 0;
}
// 0 when u is empty, else 1 + the code of u's last letter (the letter in front of the core)
template <typename W, bool SAFE = true> ACX_HD uint32_t inner_letter(W w, int n) {
    const int p = conj_prefix<W, SAFE>(w, n);
    return p ? 1u + (uint32_t)get<W, SAFE>(w, p - 1) : 0u;
}

// one multiply-xorshift round per 64-bit word: the engine's key hash (table bucket, fold slot, fingerprint) and the finaliser of the
// owner function
ACX_HD uint64_t shard_mix(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
// the 32 bits from bit 20 of a hash scaled to [0, world): no division
ACX_HD uint32_t owner_of_hash(uint64_t h, uint32_t world) { return (uint32_t)(((uint64_t)(uint32_t)(h >> 20) * world) >> 32); }
// what names the owner: the two class hashes and the two inner letters
ACX_HD uint32_t owner_sum(uint32_t c0, uint32_t c1, uint32_t in0, uint32_t in1) { return c0 + c1 + in0 * 0x9E3779B1u + in1 * 0x85EBCA77u; }
ACX_HD uint32_t owner_of_sum(uint32_t sum, uint32_t world) { return owner_of_hash(shard_mix(0, (uint64_t)sum), world); }

}  // namespace acx
