// dup_structure.cpp -- where do the duplicate children of a BFS over the AC graph come from?  (CPU experiment that
// guided the visited-table design: which probes can be answered without touching the big table.)
//   g++ -O2 -std=c++17 -I ac-solver_amd/csrc -o /tmp/dups tools/dup_structure.cpp && /tmp/dups [budget] [batch_parents] [tile_parents]
// Runs the reference's BFS (breadth_first.py:55-97 order) on AK(3) at L = 25 with the kernels' own packed-word code
// (acx_word.h compiles for the host) and classifies every generated child.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unordered_map>
#include <vector>
#include "acx_word.h"

using namespace acx;
struct Key {
    uint64_t a, b;
    bool operator==(const Key& o) const { return a == o.a && b == o.b; }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t x = k.a * 0x9e3779b97f4a7c15ull ^ (k.b + 0x7f4a7c15ull) * 0xd6e8feb86659fd93ull;
        x ^= x >> 29;
        return (size_t)(x * 0xbf58476d1ce4e5b9ull);
    }
};
struct Info {
    uint32_t id;
    uint32_t batch;
    uint32_t ppos;  // position of the discoverer's parent inside its batch
};

int main(int argc, char** argv) {
    const long long budget = argc > 1 ? atoll(argv[1]) : 3000000;
    const uint32_t bmax = argc > 2 ? (uint32_t)atoll(argv[2]) : 1u << 20;
    const uint32_t tile = argc > 3 ? (uint32_t)atoll(argv[3]) : 85;  // parents per 1024-lane tile
    const int L = 25;
    std::vector<Key> nodes;
    std::vector<uint32_t> parent;
    std::unordered_map<Key, Info, KeyHash> seen;
    seen.reserve((size_t)budget * 2);
    auto mk = [](const Pres<uint64_t>& s) { return Key{s.w0 | ((uint64_t)s.n0 << 58), s.w1 | ((uint64_t)s.n1 << 58)}; };
    Pres<uint64_t> root;
    const int8_t r0[25] = {1, 1, 1, -2, -2, -2, -2}, r1[25] = {1, 2, 1, -2, -1, -2};
    pack_relator<uint64_t>(r0, L, root.w0, root.n0);
    pack_relator<uint64_t>(r1, L, root.w1, root.n1);
    nodes.push_back(mk(root));
    parent.push_back(0xffffffffu);
    seen[mk(root)] = Info{0, 0, 0};
    unsigned long long c_noop = 0, c_back = 0, c_old = 0, c_new = 0, c_inb = 0, c_total = 0, c_intile = 0, c_insame = 0, c_in1k = 0, c_in16k = 0;
    unsigned long long c_old_prevlevel = 0;
    uint32_t head = 0, batch = 0;
    bool done = false;
    while (!done && head < nodes.size()) {
        batch++;
        const uint32_t np = (uint32_t)std::min<size_t>(nodes.size() - head, bmax);
        const size_t level_nodes = nodes.size();
        for (uint32_t p = 0; p < np && !done; p++) {
            const Key pk = nodes[head + p];
            for (int a = 0; a < 12; a++) {
                Pres<uint64_t> s{pk.a & ((1ull << 58) - 1), pk.b & ((1ull << 58) - 1), (int)(pk.a >> 58), (int)(pk.b >> 58)};
                apply_move<uint64_t, true>(s, a, L, false);
                const Key ck = mk(s);
                c_total++;
                if (ck == pk) {
                    c_noop++;
                    continue;
                }
                const uint32_t gp = parent[head + p];
                if (gp != 0xffffffffu && ck == nodes[gp]) {
                    c_back++;
                    continue;
                }
                auto it = seen.find(ck);
                if (it == seen.end()) {
                    seen.emplace(ck, Info{(uint32_t)nodes.size(), batch, p});
                    nodes.push_back(ck);
                    parent.push_back(head + p);
                    c_new++;
                } else if (it->second.batch == batch) {
                    c_inb++;
                    const uint32_t d = p - it->second.ppos;
                    if (d == 0) c_insame++;
                    if (p / tile == it->second.ppos / tile) c_intile++;
                    if (p / 1024 == it->second.ppos / 1024) c_in1k++;
                    if (p / 16384 == it->second.ppos / 16384) c_in16k++;
                } else {
                    c_old++;
                    if (it->second.id >= head) c_old_prevlevel++;
                }
            }
            if ((long long)nodes.size() >= budget) done = true;
        }
        (void)level_nodes;
        head += np;
    }
    printf("budget %lld, batch %u parents, tile %u parents: children %llu\n", budget, bmax, tile, c_total);
    auto pc = [&](const char* n, unsigned long long v) { printf("  %-44s %12llu  %5.1f %%\n", n, v, 100.0 * v / c_total); };
    pc("no-op (child == parent)", c_noop);
    pc("undoes the parent's move (== grandparent)", c_back);
    pc("new state (first discoverer)", c_new);
    pc("duplicate of a state of an earlier batch", c_old);
    pc("  ... of which in the frontier itself (id >= head)", c_old_prevlevel);
    pc("duplicate of a state first seen in THIS batch", c_inb);
    pc("  ... same parent", c_insame);
    pc("  ... discoverer in the same tile of parents", c_intile);
    pc("  ... discoverer in the same 1024 parents", c_in1k);
    pc("  ... discoverer in the same 16384 parents", c_in16k);
    return 0;
}
