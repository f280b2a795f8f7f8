// owner_balance.cpp -- which function of a state should name the rank that owns it in the sharded BFS?  (CPU experiment, round 5.)
//   g++ -O2 -std=c++17 -I ac-solver_amd/csrc -o /tmp/owner_balance tools/owner_balance.cpp && /tmp/owner_balance [budget] [cyclical] [chunk parents] [L] [r0 letters, comma separated] [r1 letters]
// Runs the reference's BFS (breadth_first.py:55-97 order) on AK(3) at L = 25 (or the given presentation) with the kernels' own packed-word code (acx_word.h
// compiles for the host) and, for every candidate owner function and world size, counts
//   * the nodes every rank would own (max / mean: the imbalance of the expansion, the table and the node arena),
//   * of the children that are neither unchanged nor the undo of their parent's move (what k_shard_expand routes): how many have
//     an owner other than their parent's (they cross the exchange as records; the others are BORN on their owner),
//   * the records every rank would RECEIVE per level (max / mean: the imbalance of k_shard_insert).
// Every AC move rewrites ONE relator (ac_moves.py:192-229: even action ids r_1, odd ids r_0), so an owner that is a function of
// r_0 alone keeps the six children of the even actions at home; one that is a function of the CONJUGACY CLASS of r_0 (the
// cyclically reduced word up to rotation) keeps the four conjugations of r_0 at home as well.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <vector>
#include "acx_owner.h"

using namespace acx;
typedef u128 WW;  // 128-bit words: max_relator_length up to 61
constexpr int kLenShift = 122;
struct Key {
    WW a, b;
    bool operator==(const Key& o) const { return a == o.a && b == o.b; }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t x = ((uint64_t)k.a ^ (uint64_t)(k.a >> 64) * 0x9e3779b97f4a7c15ull) * 0x9e3779b97f4a7c15ull ^ ((uint64_t)k.b + (uint64_t)(k.b >> 64) * 0x7f4a7c15ull) * 0xd6e8feb86659fd93ull;
        x ^= x >> 29;
        return (size_t)(x * 0xbf58476d1ce4e5b9ull);
    }
};
static inline uint64_t mix(uint64_t h, uint64_t w) { return shard_mix(h, w); }
static inline uint32_t scale(uint64_t h, uint32_t world) { return (uint32_t)(((uint64_t)(uint32_t)(h >> 20) * world) >> 32); }

// smallest rotation of the cyclic reduction of (w, n): the canonical name of the conjugacy class
static uint64_t conj_class(WW w, int n) {
    cyclic_reduce<WW, true>(w, n);
    if (n == 0) return 0;
    const WW m = mask<WW, true>(n);
    WW best = w;
    for (int i = 1; i < n; i++) {
        const WW r = ((w >> (2 * i)) | (w << (2 * (n - i)))) & m;
        best = std::min(best, r);
    }
    return mix(mix((uint64_t)n, (uint64_t)best), (uint64_t)(best >> 64));
}
// a cheap rotation invariant: length, the four letter counts and the number of positions at which the cyclic word agrees with
// itself shifted by 1..4 letters
static int popc(WW x) { return __builtin_popcountll((uint64_t)x) + __builtin_popcountll((uint64_t)(x >> 64)); }
static uint64_t conj_cheap(WW w, int n) {
    cyclic_reduce<WW, true>(w, n);
    if (n == 0) return 0;
    const WW m = mask<WW, true>(n), lo = wtraits<WW>::lo_ones() & m;
    uint64_t h = (uint64_t)n;
    for (int c = 0; c < 4; c++) {
        const WW t = w ^ (lo * (WW)c);
        h = h * 64 + (uint64_t)popc(~(t | (t >> 1)) & lo);
    }
    for (int s = 1; s <= 4 && s < n; s++) {
        const WW r = ((w >> (2 * s)) | (w << (2 * (n - s)))) & m, t = w ^ r;
        h = h * 64 + (uint64_t)popc(~(t | (t >> 1)) & lo);
    }
    return h;
}

// the cyclic bigram counts of the cyclic reduction: the engine's own function (acx_owner.h: class_hash)
static uint64_t conj_bigram(WW w, int n) { return class_hash<WW, true>(w, n); }

// the k letters of the conjugator next to the cyclically reduced core (r = u c u^-1: the LAST k letters of u): conjugating r by a
// generator adds or removes a letter at the FRONT of u, so this only changes while |u| <= k
static uint64_t inner_letters(WW w, int n, int k) {
    WW c = w;
    int m = n;
    cyclic_reduce<WW, true>(c, m);
    const int p = (n - m) / 2;
    uint64_t f = 1;
    for (int i = 0; i < k && i < p; i++) f = f * 5 + 1 + (uint64_t)get<WW, true>(w, p - 1 - i);
    return f;
}

enum { S_WHOLE = 0, S_R0, S_CLASS0, S_CHEAP0, S_CLASS01, S_CLASS0_R1LEN, S_BIGRAM01, S_CHEAP01, S_BIGRAM01_IN1, S_BIGRAM01_IN2, S_BIGRAM01_IN3, S_BIGRAM01_IN1_R0, S_BIGRAM01_IN1_R1, S_N };
static const char* kNames[S_N] = {"hash(r0, r1)           [round 4]", "hash(r0)", "hash(class(r0))", "hash(cheap invariant(r0))",
                                  "hash(class(r0)) + hash(class(r1))", "hash(class(r0), |r1| >> 2)",
                                  "hash(bigrams(r0)) + hash(bigrams(r1))", "hash(cheap(r0)) + hash(cheap(r1))",
                                  "bigrams + 1 innermost conjugator letter per relator", "bigrams + 2 innermost conjugator letters", "bigrams + 3 innermost conjugator letters",
                                  "bigrams + the innermost conjugator letter of r0 only", "bigrams + the innermost conjugator letter of r1 only"};
static uint64_t owner_hash(int scheme, const Key& k) {
    const WW w0 = k.a & (((WW)1 << kLenShift) - 1), w1 = k.b & (((WW)1 << kLenShift) - 1);
    const int n0 = (int)(k.a >> kLenShift), n1 = (int)(k.b >> kLenShift);
    switch (scheme) {
        case S_WHOLE: return mix(mix(mix(mix(0, (uint64_t)k.a), (uint64_t)(k.a >> 64)), (uint64_t)k.b), (uint64_t)(k.b >> 64));
        case S_R0: return mix(mix(0, (uint64_t)k.a), (uint64_t)(k.a >> 64));
        case S_CLASS0: return mix(0, conj_class(w0, n0));
        case S_CHEAP0: return mix(0, conj_cheap(w0, n0));
        case S_CLASS01: return mix(0, conj_class(w0, n0)) + mix(1, conj_class(w1, n1));
        case S_BIGRAM01: return mix(0, (uint32_t)(conj_bigram(w0, n0) + conj_bigram(w1, n1)));
        case S_BIGRAM01_IN1:
        case S_BIGRAM01_IN2:
        case S_BIGRAM01_IN3: {
            const int k = scheme - S_BIGRAM01_IN1 + 1;
            return mix(0, (uint32_t)(conj_bigram(w0, n0) + conj_bigram(w1, n1) + inner_letters(w0, n0, k) * 0x9E3779B1u + inner_letters(w1, n1, k) * 0x85EBCA77u));
        }
        case S_BIGRAM01_IN1_R0: return mix(0, (uint32_t)(conj_bigram(w0, n0) + conj_bigram(w1, n1) + inner_letters(w0, n0, 1) * 0x9E3779B1u));
        case S_BIGRAM01_IN1_R1: return mix(0, (uint32_t)(conj_bigram(w0, n0) + conj_bigram(w1, n1) + inner_letters(w1, n1, 1) * 0x85EBCA77u));
        case S_CHEAP01: return mix(0, conj_cheap(w0, n0)) + mix(1, conj_cheap(w1, n1));
        default: return mix(mix(0, conj_class(w0, n0)), (uint64_t)(n1 >> 2));
    }
}

int main(int argc, char** argv) {
    const long long budget = argc > 1 ? atoll(argv[1]) : 3000000;
    const bool cyc = argc > 2 && atoi(argv[2]) != 0;
    const uint32_t chunk = argc > 3 ? (uint32_t)atoll(argv[3]) : 1u << 21;
    const int L = argc > 4 ? atoi(argv[4]) : 25;
    const int worlds[3] = {2, 4, 8};
    std::vector<Key> nodes;
    std::vector<uint8_t> act;
    std::vector<uint32_t> level_end;
    std::unordered_map<Key, uint32_t, KeyHash> seen;
    seen.reserve((size_t)budget * 2);
    auto mk = [](const Pres<WW>& s) { return Key{s.w0 | ((WW)s.n0 << kLenShift), s.w1 | ((WW)s.n1 << kLenShift)}; };
    Pres<WW> root;
    int8_t r0[64] = {1, 1, 1, -2, -2, -2, -2}, r1[64] = {1, 2, 1, -2, -1, -2};
    for (int h = 0; h < 2 && argc > 5 + h; h++) {  // relators as comma-separated letters
        int8_t* r = h ? r1 : r0;
        memset(r, 0, 64);
        int k = 0;
        for (char* t = strtok(argv[5 + h], ","); t && k < 64; t = strtok(nullptr, ",")) r[k++] = (int8_t)atoi(t);
    }
    pack_relator<WW>(r0, L, root.w0, root.n0);
    pack_relator<WW>(r1, L, root.w1, root.n1);
    nodes.push_back(mk(root));
    act.push_back(0xff);
    seen[mk(root)] = 0;
    // per scheme, world: routed children, remote ones; per level and rank: records received
    static unsigned long long routed[S_N], remote[S_N][3];
    std::vector<std::vector<unsigned long long>> recv_level;  // [level][(scheme * 3 + wi) * 8 + rank]
    uint32_t head = 0;
    bool done = false;
    auto inverse_action = [](uint32_t a) { return a < 4 ? a ^ 2u : (a < 8 ? a + 4u : a - 4u); };
    while (!done && head < nodes.size()) {
        const uint32_t lvl_hi = (uint32_t)nodes.size();
        recv_level.emplace_back((size_t)S_N * 3 * 8, 0ull);
        auto& rl = recv_level.back();
        for (uint32_t p = head; p < lvl_hi && !done; p++) {
            const Key pk = nodes[p];
            uint64_t ph[S_N];
            for (int s = 0; s < S_N; s++) ph[s] = owner_hash(s, pk);
            for (int a = 0; a < 12; a++) {
                Pres<WW> s{pk.a & (((WW)1 << kLenShift) - 1), pk.b & (((WW)1 << kLenShift) - 1), (int)(pk.a >> kLenShift), (int)(pk.b >> kLenShift)};
                apply_move<WW, true>(s, a, L, cyc);
                const Key ck = mk(s);
                const bool unchanged = ck == pk, undo = !cyc && act[p] < 12 && (uint32_t)a == inverse_action(act[p]);
                if (!unchanged && !undo) {
                    for (int sc = 0; sc < S_N; sc++) {
                        const uint64_t ch = owner_hash(sc, ck);
                        routed[sc]++;
                        for (int wi = 0; wi < 3; wi++) {
                            const uint32_t po = scale(ph[sc], worlds[wi]), co = scale(ch, worlds[wi]);
                            if (po != co) {
                                remote[sc][wi]++;
                                rl[((size_t)sc * 3 + wi) * 8 + co]++;
                            }
                        }
                    }
                }
                if (seen.find(ck) == seen.end()) {
                    seen[ck] = (uint32_t)nodes.size();
                    nodes.push_back(ck);
                    act.push_back((uint8_t)a);
                }
            }
            if ((long long)nodes.size() >= budget) done = true;
        }
        head = lvl_hi;
        level_end.push_back(lvl_hi);
    }
    printf("L = %d, cyclical = %d: %zu nodes, %zu levels expanded (the last one possibly cut by the budget)\n", L, (int)cyc, nodes.size(), level_end.size());
    for (int sc = 0; sc < S_N; sc++) {
        printf("\n== owner = %s\n", kNames[sc]);
        for (int wi = 0; wi < 3; wi++) {
            const int W = worlds[wi];
            std::vector<unsigned long long> own(W, 0), own_last(W, 0);
            const uint32_t last_lo = level_end.size() >= 1 ? level_end.back() : 0;  // nodes born in the last expanded level
            for (size_t i = 0; i < nodes.size(); i++) {
                const uint32_t o = scale(owner_hash(sc, nodes[i]), W);
                own[o]++;
                if (i >= last_lo) own_last[o]++;
            }
            auto imb = [&](const std::vector<unsigned long long>& v) {
                unsigned long long mx = 0, sum = 0;
                for (auto x : v) mx = std::max(mx, x), sum += x;
                return sum ? (double)mx * v.size() / (double)sum : 0.0;
            };
            std::vector<unsigned long long> rv(W, 0);
            const auto& rl = recv_level.back();
            for (int r = 0; r < W; r++) rv[r] = rl[((size_t)sc * 3 + wi) * 8 + r];
            // chunks of consecutive frontier positions of the deepest complete level: the parents a rank expands per chunk
            double worst = 0, wsum = 0, wn = 0;
            if (level_end.size() >= 2) {
                const uint32_t lo = level_end[level_end.size() - 2], hi = level_end.back();
                for (uint32_t c = lo; c < hi; c += chunk) {
                    std::vector<unsigned long long> cnt(W, 0);
                    const uint32_t e = std::min<uint32_t>(hi, c + chunk);
                    for (uint32_t i = c; i < e; i++) cnt[scale(owner_hash(sc, nodes[i]), W)]++;
                    const double q = imb(cnt);
                    worst = std::max(worst, q);
                    wsum += q * (e - c);
                    wn += e - c;
                }
            }
            printf("  W = %d: remote %.3f of the routed children (ideal whole-key hash: %.3f); nodes max/mean %.3f (deepest level %.3f); records received in the deepest level max/mean %.3f; "
                   "local parents per chunk of %u, max/mean: %.3f on average, %.3f worst\n",
                   W, (double)remote[sc][wi] / (double)routed[sc], (double)(W - 1) / W, imb(own), imb(own_last), imb(rv), chunk, wn ? wsum / wn : 0.0, worst);
        }
    }
    return 0;
}
