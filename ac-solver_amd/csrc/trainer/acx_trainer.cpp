// acx_trainer.cpp -- host utilities of the PPO trainer (libacx_trainer.so, include/acx_trainer.h).  No device work, no HIP: plain C++.
//
// Round 6: these two restatements of third-party generators (NumPy's legacy MT19937 shuffle, CPython's random.Random) lived in
// libacx.so's public header until round 5; the trainer is outside the accelerated hot path, so they are a helper
// library of their own that only ac_solver/agents/ loads.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../../include/acx_trainer.h"

namespace {
thread_local char g_err[256];
int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
}  // namespace

// ---- NumPy's legacy shuffle, off the interpreter lock (acxt_np_shuffle_epochs) ----------------------------------------------------
// The PPO update of the reference shuffles np.arange(batch_size) with the global generator it seeded at the top of the update
// (agents/training.py:121, 273-275).  np.random.shuffle holds the GIL (95 ms for 4 Mi indices, the whole time), so it cannot run beside
// the rollout's Python loop; this restatement of NumPy's published algorithm can (ctypes releases the GIL).  NumPy is a third-party
// dependency of the reference (numpy/random: legacy RandomState): seeding with an integer = MT19937 init_genrand(seed & 2^32 - 1);
// shuffle of a 1-D array = for i = n - 1 down to 1: j = random_interval(i); swap(x[i], x[j]); random_interval(max) for max < 2^32 =
// 32-bit draws masked with the smallest 2^k - 1 >= max, redrawn until <= max.  tests/test_agents_cpu.py pins it against numpy itself.
namespace {
struct Mt19937 {
    uint32_t key[624];
    int pos;
    explicit Mt19937(uint32_t seed) {
        for (int i = 0; i < 624; i++) {
            key[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void refill() {
        constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;
        int i = 0;
        for (; i < 624 - 397; i++) {
            const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
        }
        for (; i < 623; i++) {
            const uint32_t y = (key[i] & kUpper) | (key[i + 1] & kLower);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
        }
        const uint32_t y = (key[623] & kUpper) | (key[0] & kLower);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
        pos = 0;
    }
    uint32_t next() {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t interval(uint32_t max) {  // uniform on [0, max]
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        uint32_t v;
        while ((v = next() & mask) > max) {
        }
        return v;
    }
};
}  // namespace

extern "C" {

int acxt_np_shuffle_epochs(uint32_t seed, int64_t n, int epochs, int64_t* out) {
    if (n < 1 || n > 0xffffffffll || epochs < 1 || !out) return fail(ACXT_E_INVAL, "acxt_np_shuffle_epochs: bad argument (1 <= n < 2^32)");
    Mt19937 rng(seed);
    // the swaps run on 32-bit indices (half the footprint of the random accesses: 16 MB at 4 Mi indices, which a host's last-level
    // cache holds) and every epoch's result is widened into its row
    std::vector<uint32_t> idx((size_t)n);
    for (int64_t i = 0; i < n; i++) idx[(size_t)i] = (uint32_t)i;
    for (int e = 0; e < epochs; e++) {  // the reference shuffles the SAME array again, epoch after epoch
        for (int64_t i = n - 1; i >= 1; i--) {
            const uint32_t j = rng.interval((uint32_t)i);
            const uint32_t t = idx[(size_t)i];
            idx[(size_t)i] = idx[j];
            idx[j] = t;
        }
        int64_t* row = out + (int64_t)e * n;
        for (int64_t i = 0; i < n; i++) row[i] = (int64_t)idx[(size_t)i];
    }
    return ACXT_OK;
}

// CPython's random.Random restated for the curriculum draws of the PPO driver (agents/training.py: choose_next_state, reference
// training.py:199-221): the generator is the same MT19937; random() = (a >> 5, b >> 6) of two outputs as a 53-bit fraction;
// uniform(0, 1) = 0 + (1 - 0) * random(); choice(seq) = seq[_randbelow(len(seq))], _randbelow(n) = getrandbits(n.bit_length())
// redrawn until < n, getrandbits(k <= 32) = output >> (32 - k)  (Lib/random.py, Modules/_randommodule.c of CPython 3.10).
int acxt_py_curriculum_draws(uint32_t* mt_state, int32_t* mt_pos, int64_t n, int64_t n_solved, int64_t n_unsolved, double repeat_solved_prob,
                            uint8_t* which, int64_t* index) {
    if (!mt_state || !mt_pos || n < 0 || (n && (!which || !index)) || n_solved < 0 || n_unsolved < 0 || (n && n_solved + n_unsolved == 0) ||
        n_solved > 0xffffffffll || n_unsolved > 0xffffffffll || *mt_pos < 0 || *mt_pos > 624)
        return fail(ACXT_E_INVAL, "acxt_py_curriculum_draws: bad argument");
    Mt19937 rng(0u);
    memcpy(rng.key, mt_state, sizeof(rng.key));
    rng.pos = *mt_pos;
    auto randbelow = [&](uint64_t m) {
        int k = 0;
        while ((m >> k) != 0) k++;  // m.bit_length()
        uint32_t r;
        do r = rng.next() >> (32 - k);
        while (r >= m);
        return (int64_t)r;
    };
    for (int64_t i = 0; i < n; i++) {
        bool unsolved = n_solved == 0;
        if (!unsolved && n_unsolved > 0) {  // (`unsolved and uniform(0, 1) > p`: no draw when nothing is unsolved)
            const uint32_t a = rng.next() >> 5, b = rng.next() >> 6;
            unsolved = ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0) > repeat_solved_prob;
        }
        which[i] = unsolved ? 0 : 1;
        index[i] = randbelow((uint64_t)(unsolved ? n_unsolved : n_solved));
    }
    memcpy(mt_state, rng.key, sizeof(rng.key));
    *mt_pos = rng.pos;
    return ACXT_OK;
}

const char* acxt_last_error(void) { return g_err; }

}  // extern "C"
