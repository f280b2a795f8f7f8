// acx_policy.hip -- rollout-time inference of the PPO agent's two tanh MLPs (actor + critic, ac_solver/agents/ppo_agent.py:
// in -> 256 -> 256 -> {n_actions, 1}) fused with the action sampling, on the matrix cores.
//
// Why it is here: a PPO rollout step is  policy(obs) -> action -> ACEnv.step.  With the env kernel at ~9 us per 131 072
// environments, torch's eager fp32 MLP (0.7 ms per step, 0.4 ms under bf16 autocast: some twenty small launches) was 99 % of a
// rollout step (BASELINE config 5).  This is the one GEMM-shaped piece of the hot path's caller, so it gets an MFMA kernel.
//
// Formulation (one wave = 32 environments, one workgroup = 8 waves = 256 environments):
//   every layer is computed TRANSPOSED,  H^T[out, env] = W[out, in] . X^T[in, env] + b,  with v_mfma_f32_32x32x16_bf16:
//   A = a 32 x 16 tile of the nn.Linear weight (a 1 KB per-lane fragment packed on the host), B = 16 inputs x 32 environments,
//   C = 32 outputs x 32 environments with the environment on the lane.  Output block by output block (32 hidden units = one
//   16-register accumulator): its bias step and k-steps, then tanh -> bf16 in registers -- which IS the B operand of two
//   k-steps of the next layer (see "K order" below): activations never touch LDS.
//   Weights: the whole fragment sequence of both networks (2 x 185 KB at 50 inputs) streams once per workgroup through a ring
//   of six 17 KB LDS slots, filled by global_load_lds_dwordx4 five chunks ahead of the MFMAs that read them (counted vmcnt,
//   raw s_barrier per chunk), shared by the eight waves.
//   The heads leave the 12 logits / the value of an environment in two lanes (l and l + 32); each lane draws among the actions it holds
//   (an exponential race on a counter-based hash of (seed, env, action): an exact sample of Categorical(softmax(logits))), the two
//   combine their soft-max sums and winners with a few shuffles, and lane l < 32 stores action, log-probability and value.
//   A workgroup walks tiles of 256 environments (one workgroup per compute unit); the weight stream carries on across tiles.
// Arithmetic: bf16 inputs, weights and activations, f32 accumulation (the reference's policy is f32 torch: the test compares
// with a torch f32 forward at bf16 tolerance and with a bf16-rounded emulation tightly).  Only inference: the PPO update
// keeps torch autograd on the f32 master weights; FusedPolicy.refresh() re-packs them after every optimizer step.
#include <type_traits>

#include "acx_common.h"

namespace acx {
namespace policy {

typedef __attribute__((ext_vector_type(8))) short frag_ab;   // 8 bf16
typedef __attribute__((ext_vector_type(16))) float frag_cd;  // 32 x 32 f32 tile: 16 per lane


__device__ __forceinline__ unsigned short bf16_of(float x) {  // round to nearest even
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float tanh_scaled(float y) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1) for y = x * 2 / ln 2 (the factor is folded into the layer's weights and bias on the host):
    // v_exp_f32 (2^y), v_add, v_rcp_f32, v_fma; overflow gives +inf -> 1, underflow 0 -> -1
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y) + 1.0f), 1.0f);
}
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    return x;
}

// ---- packed network (ac_solver/agents/fused_policy.py::pack_network) ----------------------------------------------------------
// A sequence of 1 KB FRAGMENTS (64 lanes x 8 bf16, the A operand of one MFMA), grouped by 32-row output block:
//   layer 1: 8 blocks x (1 bias fragment + KS1 k-steps), layer 2: 8 x (1 + 16), head: 1 x (1 + 16).
// The bias fragment carries b[row] as bf16 hi + lo in k = 0, 1; its B operand is the constant (1, 1, 0, ...), so an output
// block starts as  acc = mfma(bias fragment, ones, 0)  -- no bias loads, no accumulator initialisation.
// K order of layer 2 / head: the hidden value that the PREVIOUS layer's accumulator holds in register v of lane (env, h) of
// output block ob is row 32 ob + 8 (v >> 2) + 4 h + (v & 3).  A k-step may contract ANY 16 hidden units as long as A and B
// agree, so k-step 2 ob + half takes exactly the eight values v = 8 half .. 8 half + 7 that each lane already owns:
// tanh -> bf16 -> the next layer's B operand never leaves the lane's registers (no activation image, no LDS round trip).
constexpr int kWaves = 8;          // waves per workgroup: 256 environments share every staged fragment
constexpr int kSlotFrags = 17;     // fragments per ring slot (one output block of a 256-wide layer)
constexpr int kSlotBytes = kSlotFrags * 1024;
#ifndef ACX_POLICY_RING
#define ACX_POLICY_RING 6
#endif
constexpr int kRing = ACX_POLICY_RING;  // slots
constexpr int kAhead = kRing - 1;  // chunks in flight ahead of the one being consumed
#ifndef ACX_POLICY_FRAG_AHEAD
#define ACX_POLICY_FRAG_AHEAD 5
#endif
constexpr int kFragAhead = ACX_POLICY_FRAG_AHEAD;  // A fragments read from LDS this many MFMAs ahead of their use

template <int KS1> struct plan {
    static constexpr int F1 = KS1 + 1;             // fragments per layer-1 output block
    static constexpr int OPC = kSlotFrags / F1;    // layer-1 output blocks per chunk
    static constexpr int C1 = (8 + OPC - 1) / OPC; // layer-1 chunks
    static constexpr int CN = C1 + 8 + 1;          // chunks per network
    static constexpr int C = 2 * CN;               // actor, then critic
    static constexpr int first(int cc) { return cc < C1 ? cc * OPC * F1 : (cc < C1 + 8 ? 8 * F1 + (cc - C1) * 17 : 8 * F1 + 8 * 17); }
    static constexpr int obs_in(int cc) { return (cc + 1) * OPC < 8 ? OPC : 8 - cc * OPC; }
    static constexpr int count(int cc) { return cc < C1 ? obs_in(cc) * F1 : 17; }
    // The weight stream is endless: behind a tile's last chunk comes chunk 0 of the workgroup's next tile (the same weights again; behind
    // the last tile the loads are issued all the same and nobody reads them).  Every wave issues three loads per chunk (fragments w and
    // w + 8, an eighth of fragment 16).  Top of loop iteration c: chunks .. c + kAhead - 1 have been issued (chunk c + kAhead only goes out
    // AFTER this wait and its barrier, into the slot chunk c - 1 leaves).  The running block reads the first fragments of chunk c + 1
    // before it ends, so chunk c + 1 must have landed: at most the loads of the kAhead - 2 chunks issued behind it may stay in flight.
    // (A tile boundary waits for vmcnt(0): the epilogue's stores and the next tile's observation loads are not part of this count.)
    static constexpr int pending_at_top = 3 * (kAhead - 2);
    static constexpr int pending_first = 3 * (kAhead - 1);  // before the first tile: chunk 0 has landed
    static_assert(kAhead >= 2, "the ring holds the running chunk, the next one and at least one in flight");
};

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// one lane-linear 1 KB fragment global -> LDS without a register stop (global_load_lds_dwordx4; M0 = LDS destination of the wave).
// `mask`: the lanes that take part (EXEC for this one instruction) -- a masked-off load is issued all the same, so every wave
// issues the same number of loads per chunk and the counted s_waitcnt below holds for all of them, without a branch.
__device__ __forceinline__ void glds16(const frag_ab* __restrict__ src_lane, uint32_t lds_dst, unsigned long long mask) {
    unsigned keep;
    unsigned long long keep_exec;
    asm volatile(
        "s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0\n\ts_mov_b64 exec, %1"
        : "=&s"(keep), "=&s"(keep_exec)
        : "v"(src_lane), "s"(lds_dst), "s"(mask)
        : "memory");
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: round to nearest even, two at a time
    f32x2 v;
    v[0] = a;
    v[1] = b;
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
template <int KS1>
__global__ void __launch_bounds__(64 * kWaves, 2) k_policy_sample(const void* __restrict__ obs_any, int obs_i8, int64_t n_env, int in_dim, const frag_ab* __restrict__ actor,
                                                                 const frag_ab* __restrict__ critic, int n_actions, unsigned long long seed,
                                                                 int64_t* __restrict__ action, float* __restrict__ logprob, float* __restrict__ value) {
    using P = plan<KS1>;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];  // the fragment ring
    ACX_VGPR_PAD("v255");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, r = lane & 31u, h = lane >> 5;
#ifdef ACX_POLICY_STAMP  // (the 100 MHz counter all compute units share: where a workgroup's life sits inside the launch)
#define ACX_POLICY_WALL(k) do { if (wave == 0) logprob[n_env + 1024 + (size_t)blockIdx.x * 4 + (k)] = (float)(uint32_t)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFull); } while (0)
#else
#define ACX_POLICY_WALL(k) do { } while (0)
#endif
    ACX_POLICY_WALL(0);
    // A workgroup walks the tiles of 256 environments blockIdx.x, blockIdx.x + gridDim.x, ... (the launch has one workgroup per compute
    // unit): the weight stream runs on across the tile boundary, and the next tile's observations are loaded behind the sampling of the
    // running one -- a second round of workgroups paid the wave launches, the ring's first fill and the observation loads again (5 us
    // before its first MFMA; round 6 measured it with the chip's 100 MHz counter, tools/policy_stamps.py).
    const int64_t tiles = (n_env + 32 * kWaves - 1) / (32 * kWaves);
    int64_t env = ((int64_t)blockIdx.x * kWaves + wave) * 32 + r;
    const uint32_t ring = (uint32_t)(uintptr_t)s_mem;  // LDS byte address (low half of the flat address)
    // The observations of a tile are one contiguous block of the observation matrix (256 rows): it is loaded as such -- every lane four
    // consecutive elements per step, a wave 1 KB (f32) or 256 B (int8 rows, what acx_env_step writes with ACX_I8) per instruction --
    // rounded to bf16 and kept behind the ring as a plain copy [256][in_dim]; the B operand of layer 1 (the lane's environment, inputs
    // 16 ks + 8 h .. + 7, zero beyond in_dim) is read from there behind the barrier of the tile's first chunk.  Rounds 2-6 had every lane
    // load its own 8 KS1 elements: 32-40 instructions of 64 different cache lines each, 3.3 us per tile in the address path (measured with
    // the chip's 100 MHz counter, tools/policy_stamps.py).  Load and conversion are two steps: the loads of the next tile stay in flight
    // across the sampling epilogue.
    constexpr int NI = 2 * KS1;  // steps: 256 rows x in_dim <= 256 x 16 KS1 elements = 2048 per step x 2 KS1
    unsigned short* s_obs = (unsigned short*)(s_mem + kRing * kSlotBytes);
    constexpr uint32_t kObsZero = 256u * 16u * (uint32_t)KS1;  // a bf16 zero behind the copy: what the inputs beyond in_dim read
    if (tid == 0) *(uint32_t*)(s_obs + kObsZero) = 0;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4a;
    uint32_t raw[4 * NI];
    auto load_obs = [&](int64_t tile_, bool live) {
        int width = in_dim;
        asm volatile("" : "+s"(width));  // (made here, not once for all tiles: what hangs on it would be hoisted out of the tile loop)
        const int64_t g0 = tile_ * (32 * kWaves) * width, total = n_env * (int64_t)width;
        const bool vec = obs_i8 ? ((uintptr_t)obs_any & 3u) == 0 : ((uintptr_t)obs_any & 15u) == 0;  // (uniform)
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const uint32_t idx = 4u * tid + 2048u * (uint32_t)i;
            const int64_t g = g0 + idx;
#pragma unroll
            for (int e = 0; e < 4; e++) raw[4 * i + e] = 0;
            if (!live || idx >= (uint32_t)(32 * kWaves * width)) continue;
            if (obs_i8) {  // four int8 elements in raw[4 i]
                const int8_t* p8 = (const int8_t*)obs_any + g;
                if (vec && g + 3 < total) raw[4 * i] = *(const uint32_t*)p8;
                else
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (g + e < total) raw[4 * i] |= (uint32_t)(uint8_t)p8[e] << (8 * e);
            } else {
                const float* pf = (const float*)obs_any + g;
                if (vec && g + 3 < total) {
                    const u32x4a v = *(const u32x4a*)pf;
#pragma unroll
                    for (int e = 0; e < 4; e++) raw[4 * i + e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (g + e < total) raw[4 * i + e] = __float_as_uint(pf[e]);
                }
            }
        }
    };
    auto stage_obs = [&]() {  // raw -> bf16 -> the copy in LDS (rows beyond n_env: zeros)
        int width = in_dim;
        asm volatile("" : "+s"(width));
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const uint32_t idx = 4u * tid + 2048u * (uint32_t)i;
            if (idx >= (uint32_t)(32 * kWaves * width)) continue;
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; e++) f[e] = obs_i8 ? (float)(int32_t)(int8_t)(raw[4 * i] >> (8 * e)) : __uint_as_float(raw[4 * i + e]);
            uint2 pk;
            pk.x = pack_bf16(f[0], f[1]);
            pk.y = pack_bf16(f[2], f[3]);
            *(uint2*)(s_obs + idx) = pk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the hand-written barriers of this kernel wait for nothing themselves)
    };
    frag_ab x[KS1];
    auto read_obs = [&]() {  // behind a barrier that follows stage_obs
        int width = in_dim;
        asm volatile("" : "+s"(width));
        const uint32_t row = (wave * 32u + r) * (uint32_t)width;
        if ((width & 1) == 0) {  // (uniform; 2 x max_relator_length is even) two inputs per 32-bit read: a pair lies inside the row or beyond it
            const uint32_t* s_obs2 = (const uint32_t*)s_obs;
#pragma unroll
            for (int ks = 0; ks < KS1; ks++) {
                u32x4a v;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t k = 16u * (uint32_t)ks + 8u * h + 2u * (uint32_t)q;
                    v[q] = s_obs2[k < (uint32_t)width ? (row + k) >> 1 : kObsZero >> 1];
                }
                x[ks] = __builtin_bit_cast(frag_ab, v);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS1; ks++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t k = 16u * (uint32_t)ks + 8u * h + (uint32_t)j;
                    x[ks][j] = (short)s_obs[k < (uint32_t)width ? row + k : kObsZero];
                }
        }
    };
    load_obs(blockIdx.x, true);
    stage_obs();
    frag_ab ones;
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = (h == 0 && j < 2) ? (short)0x3F80 : (short)0;

    // Every second workgroup of an XCD (workgroups go to the eight XCDs in turn) runs the critic first: the compute units of an XCD stay
    // in step with each other, and all of them pulling the same 17 KB of weights through the same L2 channels at the same moment is what
    // the streaming costs -- with two orders, two address streams are in flight at any time.
    const bool critic_first = ((blockIdx.x >> 3) & 1u) != 0;  // (uniform)
    const frag_ab* net0 = critic_first ? critic : actor;
    const frag_ab* net1 = critic_first ? actor : critic;
    const frag_ab* src_w[2] = {net0 + wave * 64 + lane, net1 + wave * 64 + lane};  // fragment `wave` of a chunk, this lane's 16 bytes
    const frag_ab* src_l[2] = {net0 + lane, net1 + lane};                            // fragment 0 of a chunk, this lane's 16 bytes
    const uint32_t ring_w = __builtin_amdgcn_readfirstlane(ring + wave * 1024u);
    auto uniform64 = [](unsigned long long v) {  // (the EXEC masks must live in scalar registers)
        return ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)v);
    };
    const unsigned long long all = uniform64(~0ull), none = uniform64(0ull);
    const unsigned long long eighth = uniform64(0xFFull << (8u * wave));
    // Ring slot of the running tile's chunk 0 (uniform).  A tile's chunks take consecutive slots and the next tile carries on behind them,
    // so chunk c of the stream (c >= P::C: chunk c - P::C of the NEXT tile, the same weights again) sits in slot (slot0 + c) % kRing.
    uint32_t slot0 = 0;
    auto slot_bytes = [&](int c) -> uint32_t {  // (c is a compile-time constant at every call: one scalar add and compare)
        uint32_t s0 = slot0;
        asm volatile("" : "+s"(s0));  // (recomputed at every use: hoisted to the top of a tile the slot addresses of all chunks would spill)
        const uint32_t sl = s0 + (uint32_t)(c % kRing);
        return __builtin_amdgcn_readfirstlane((sl >= (uint32_t)kRing ? sl - (uint32_t)kRing : sl) * (uint32_t)kSlotBytes);
    };
    auto issue = [&](auto ic) {  // chunk c of the stream -> its ring slot
        constexpr int c = decltype(ic)::value;
        constexpr int cc = (c % P::C) % P::CN, cnt = P::count(cc), net = (c % P::C) < P::CN ? 0 : 1;
        constexpr size_t first = (size_t)P::first(cc) * 64;
        const uint32_t sb = slot_bytes(c);
        // (the source addresses are made HERE, from pointers the optimiser cannot see through: inside the tile loop they are loop
        // invariants, and hoisted out of it -- fifty 64-bit values -- they would live in scratch memory)
        const frag_ab* pw = src_w[net];
        const frag_ab* pl = src_l[net];
        asm volatile("" : "+v"(pw), "+v"(pl));
        glds16(pw + first, ring_w + sb, all);
        glds16(pw + first + 8 * 64, ring_w + sb + 8 * 1024, cnt >= 16 ? all : uniform64(wave + 8 < (uint32_t)cnt ? ~0ull : 0ull));
        glds16(pl + first + 16 * 64, __builtin_amdgcn_readfirstlane(ring + sb + 16 * 1024), cnt == 17 ? eighth : none);
    };
    static_for<0, kAhead>(issue);

#ifdef ACX_POLICY_STAMP
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    ACX_POLICY_WALL(1);
#endif
    frag_ab hb1[16], hb2[16];  // tanh outputs of layers 1 and 2 as B operands (k-step 2 ob + half <- accumulator registers 8 half ..)
    frag_cd pend;              // accumulator of the previous output block: its tanh is spread among this block's MFMAs
    frag_cd head[2];
    const frag_cd zero = {0};
    // One output block: NM MFMAs (bias step, then the k-steps with B = bfrag(ks)) on the fragments S[0 .. NM), and -- in the gaps the
    // matrix pipe leaves in the wave's issue slots -- the tanh of the PREVIOUS block's sixteen accumulator registers, a few after each
    // MFMA.  __builtin_amdgcn_sched_barrier(0) pins that interleaving (left alone the scheduler runs all MFMAs, then all tanh:
    // the wave then alternates between a phase that idles the matrix pipe and one that idles the vector pipe); the A fragments
    // are read from LDS kFragAhead MFMAs ahead.
    frag_ab apre[kFragAhead];  // the first fragments of the NEXT block, read at the end of the running one (no LDS latency bubble at a block's start)
    auto block = [&](auto nm, const frag_ab* __restrict__ S, auto nm_next, const frag_ab* __restrict__ Snext, auto bfrag, bool have_pend, frag_ab& out_lo,
                     frag_ab& out_hi) -> frag_cd {
        constexpr int NM = decltype(nm)::value, NMN = decltype(nm_next)::value;
        frag_ab a[NM];
        u32x4 lo, hi;
#pragma unroll
        for (int i = 0; i < kFragAhead && i < NM; i++) a[i] = apre[i];
        frag_cd acc;
        constexpr int U = (NM + 1) / 2;                    // units of two MFMAs
        constexpr int UT = (NM & 1) && U > 1 ? U - 1 : U;  // the units that carry tanh work (an odd block's last MFMA may consume out_hi)
        static_for<0, U>([&](auto uu) {
            constexpr int u = decltype(uu)::value;
            if constexpr (u == U - 1) {  // the next block's first fragments: on their way before the block's last MFMAs, not behind them
#pragma unroll
                for (int i = 0; i < kFragAhead && i < NMN; i++) apre[i] = Snext[i * 64];
            }
            static_for<2 * u, (2 * u + 2 < NM ? 2 * u + 2 : NM)>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
#ifdef ACX_POLICY_NO_LDS_READ  // timing experiment only (wrong numbers): every MFMA reuses the block's first fragments -- what the LDS reads cost
                if constexpr (i + kFragAhead < NM) a[i + kFragAhead] = a[i % kFragAhead];
#else
                if constexpr (i + kFragAhead < NM) a[i + kFragAhead] = S[(i + kFragAhead) * 64];
#endif
                if constexpr (i == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], ones, zero, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bfrag(std::integral_constant<int, i - 1>{}), acc, 0, 0, 0);
            });
            if (have_pend && u < UT) {
                // two values at a time, their instruction chains interleaved: a transcendental's result is not consumed by the next
                // instruction (that costs an s_nop each time), and the pair shares one v_cvt_pk_bf16_f32
#pragma unroll
                for (int q = 8 * u / UT; q < 8 * (u + 1) / UT; q++) {
#ifdef ACX_POLICY_NO_TANH  // timing experiment only (wrong numbers): what the kernel costs without the tanh's vector instructions
                    const uint32_t pk0 = pack_bf16(pend[2 * q], pend[2 * q + 1]);
                    if (q < 4) lo[q] = pk0;
                    else hi[q - 4] = pk0;
                    if (q == 3) out_lo = __builtin_bit_cast(frag_ab, lo);
                    if (q == 7) out_hi = __builtin_bit_cast(frag_ab, hi);
                    continue;
#endif
                    const float ea = __builtin_amdgcn_exp2f(pend[2 * q]), eb = __builtin_amdgcn_exp2f(pend[2 * q + 1]);
                    const float da = ea + 1.0f, db = eb + 1.0f;
                    const float ra = __builtin_amdgcn_rcpf(da), rb = __builtin_amdgcn_rcpf(db);
                    const uint32_t pk = pack_bf16(__builtin_fmaf(-2.0f, ra, 1.0f), __builtin_fmaf(-2.0f, rb, 1.0f));
                    if (q < 4) lo[q] = pk;
                    else hi[q - 4] = pk;
                    if (q == 3) out_lo = __builtin_bit_cast(frag_ab, lo);
                    if (q == 7) out_hi = __builtin_bit_cast(frag_ab, hi);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        return acc;
    };
    frag_ab unused_lo, unused_hi;
    // chunk 0 has landed (for everyone: barrier) -> its first fragments
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(P::pending_first) : "memory");
#pragma unroll
    for (int i = 0; i < kFragAhead && i < P::F1; i++) apre[i] = ((const frag_ab*)s_mem + lane)[i * 64];
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    static_for<0, P::C>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        constexpr int net = c / P::CN, cc = c % P::CN;
        constexpr int ccn = (c + 1) % P::CN, nm_of_next_chunk = ccn < P::C1 ? P::F1 : 17;  // (behind the tile's last chunk: the next tile's first block)
        // chunk c + 1 has landed for this wave's own loads ... and, behind the barrier, for everyone's (the running block reads its
        // first fragments before it ends); chunk c - 1 is consumed, so its slot takes chunk c + kAhead
#ifdef ACX_POLICY_NO_BARRIER  // timing experiment only (races on the ring): what the per-chunk workgroup barrier costs
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::pending_at_top) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(P::pending_at_top) : "memory");
#endif
#ifdef ACX_POLICY_STAMP  // diagnostic build: shader-clock stamp per chunk into the (over-allocated) logprob buffer
        logprob[n_env + (blockIdx.x & 1) * 512 + wave * 64 + (tile >= (int64_t)gridDim.x ? 30 : 0) + c] = (float)(long long)(__builtin_amdgcn_s_memtime() - t_start);  // (no branch: all lanes store the same word)
#endif
#ifndef ACX_POLICY_NO_STREAM  // (timing experiment only when defined: the weights are not streamed beyond the ring's first fill)
        issue(std::integral_constant<int, c + kAhead>{});
#endif
        if constexpr (c == 0) read_obs();  // (the tile's observations: staged before this chunk's barrier)
        const frag_ab* S = (const frag_ab*)(s_mem + slot_bytes(c)) + lane;
        const frag_ab* Sn = (const frag_ab*)(s_mem + slot_bytes(c + 1)) + lane;  // first block of the next chunk
        if constexpr (cc < P::C1) {  // ---- layer 1: a few output blocks per chunk
            static_for<0, P::obs_in(cc)>([&](auto kk) {
                constexpr int k = decltype(kk)::value, ob = cc * P::OPC + k;
                constexpr int prev = ob > 0 ? ob - 1 : 0;
                constexpr bool last = k + 1 == P::obs_in(cc);
                const frag_cd acc = block(std::integral_constant<int, P::F1>{}, S + (k * P::F1) * 64, std::integral_constant<int, last ? nm_of_next_chunk : P::F1>{},
                                          last ? Sn : S + ((k + 1) * P::F1) * 64, [&](auto ks) { return x[decltype(ks)::value]; }, ob > 0,
                                          ob > 0 ? hb1[2 * prev] : unused_lo, ob > 0 ? hb1[2 * prev + 1] : unused_hi);
                pend = acc;
            });
        } else if constexpr (cc < P::C1 + 8) {  // ---- layer 2: one output block per chunk; block 0 also finishes layer 1's last tanh
            constexpr int ob = cc - P::C1;
            constexpr int prev = ob > 0 ? ob - 1 : 0;
            const frag_cd acc = block(std::integral_constant<int, 17>{}, S, std::integral_constant<int, nm_of_next_chunk>{}, Sn,
                                      [&](auto ks) { return hb1[decltype(ks)::value]; }, true, ob > 0 ? hb2[2 * prev] : hb1[14],
                                      ob > 0 ? hb2[2 * prev + 1] : hb1[15]);
            pend = acc;
        } else {  // ---- head (with the tanh of layer 2's last block)
            head[net] = block(std::integral_constant<int, 17>{}, S, std::integral_constant<int, nm_of_next_chunk>{}, Sn,
                              [&](auto ks) { return hb2[decltype(ks)::value]; }, true, hb2[14], hb2[15]);
        }
    });
    float logit[8];  // (the actor's head: the network that ran first or second)
#pragma unroll
    for (int k = 0; k < 8; k++) logit[k] = critic_first ? head[1][k] : head[0][k];
    const float val0 = critic_first ? head[0][0] : head[1][0];
    ACX_POLICY_WALL(2);
    // the next tile's observations start their trip now and are looked at behind the sampling (no tile left: no load goes out)
    const int64_t env_next = env + (int64_t)gridDim.x * (32 * kWaves);
    load_obs(tile + gridDim.x, tile + gridDim.x < tiles);
    __builtin_amdgcn_sched_barrier(0);
    {
        // Head output a of an environment sits in lane half (a >> 2) & 1, register (a & 3) + 4 (a >> 3): a lane draws for the eight actions
        // its own registers hold -- no shuffle per logit (rounds 2-6 gave half h the actions 8 h .. 8 h + 7 and fetched half of them from
        // the partner lane) -- and the halves combine the soft-max's maximum and sum and, at the end, their winners.
        // The draw is an exponential race: argmin over a of E_a / w_a with E_a = -ln u_a i.i.d. Exp(1) and w_a = exp(logit_a - max) is an
        // exact sample of Categorical(softmax(logits)) (the same law as the Gumbel-max of rounds 2-6 without its two logarithms per action:
        // w_a is the soft-max's own term).  Transcendentals as the bare instructions (v_exp_f32 / v_log_f32 are base 2; every argument
        // here is a normal number, so none of the denormal handling of expf / logf is needed).
        constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
        bool valid[8];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            valid[k] = (k & 3) + 4 * (int)h + 8 * (k >> 2) < n_actions;
            mx = valid[k] ? fmaxf(mx, logit[k]) : mx;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float w[8], sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            w[k] = valid[k] ? __builtin_amdgcn_exp2f((logit[k] - mx) * kLog2e) : 0.0f;
            sum += w[k];
        }
        sum += __shfl_xor(sum, 32);
        const float ln_sum = __builtin_amdgcn_logf(sum) * kLn2;
        // counter-based uniforms: one hash of (seed, environment) -- two rounds of the murmur3 finaliser, keyed by both seed halves -- and
        // one more round per action (32-bit multiplies are quarter rate: rounds 2-6 spent four per action)
        const uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
        const uint32_t base = fmix32(fmix32((uint32_t)env ^ s_lo) + s_hi + (uint32_t)(env >> 32));
        int best = 0;
        float best_t = 3.0e38f, best_lp = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int a = (k & 3) + 4 * (int)h + 8 * (k >> 2);
            const uint32_t bits = fmix32(base + (uint32_t)(a + 1) * 0x9E3779B9u);
            // 23 random bits: (k + 0.5) * 2^-23 is exact in f32 for every k < 2^23, so u stays strictly inside (0, 1)
            const float u = ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f);
            const float t = -__builtin_amdgcn_logf(u) * __builtin_amdgcn_rcpf(w[k]);  // (w = 0: an action of probability < 1e-38, or none: infinite)
            if (valid[k] && t < best_t) {  // (the lane's own actions in rising order: a tie keeps the lower one)
                best_t = t;
                best = a;
                best_lp = (logit[k] - mx) - ln_sum;
            }
        }
        const float o_t = __shfl_xor(best_t, 32), o_lp = __shfl_xor(best_lp, 32);
        const int o_best = __shfl_xor(best, 32);
        if (o_t < best_t || (o_t == best_t && o_best < best)) {  // (ties: the lower action, as a scan in action order would)
            best = o_best;
            best_lp = o_lp;
        }
        if (h == 0 && env < n_env) {
            action[env] = best;
            logprob[env] = best_lp;
            value[env] = val0;
        }
#ifdef ACX_POLICY_STAMP
        ACX_POLICY_WALL(3);
        logprob[n_env + (blockIdx.x & 1) * 512 + wave * 64 + (tile >= (int64_t)gridDim.x ? 63 : 29)] = (float)(long long)(__builtin_amdgcn_s_memtime() - t_start);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    env = env_next;
    stage_obs();  // (every wave read the running tile's copy behind the barrier of chunk 0, twenty barriers ago)
    slot0 = __builtin_amdgcn_readfirstlane((slot0 + (uint32_t)(P::C % kRing)) % (uint32_t)kRing);
    // everything this wave has in flight -- the stores above and the ring's next chunks, issued up to kAhead ago -- is through before the
    // counted waits of the next tile start from zero
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }  // tiles
}

}  // namespace policy
}  // namespace acx

using namespace acx;

extern "C" int acx_policy_sample(const void* d_obs, int obs_dtype, int64_t n_env, int in_dim, const void* d_actor, const void* d_critic, int n_actions, uint64_t seed,
                                 int64_t* d_action, float* d_logprob, float* d_value, void* stream) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (!d_obs || !d_actor || !d_critic || !d_action || !d_logprob || !d_value || n_env < 0) return fail(ACX_E_INVAL, "acx_policy_sample: bad argument");
    if (obs_dtype != ACX_F32 && obs_dtype != ACX_I8) return fail(ACX_E_INVAL, "acx_policy_sample: observations are ACX_F32 or ACX_I8");
    // 80 inputs = max_relator_length 40 (the reference trains at 36); wider first layers would not leave the register margin of ACX_VGPR_PAD
    if (in_dim < 1 || in_dim > 80 || n_actions < 1 || n_actions > 16) return fail(ACX_E_INVAL, "acx_policy_sample handles 1..80 inputs and 1..16 actions");
    if (n_env == 0) return ACX_OK;
    const int ks1 = (in_dim + 15) / 16;
    const policy::frag_ab* a = (const policy::frag_ab*)d_actor;
    const policy::frag_ab* c = (const policy::frag_ab*)d_critic;
    const int64_t tiles = (n_env + 32 * policy::kWaves - 1) / (32 * policy::kWaves);
    static int cus = 0;  // (a workgroup fills a compute unit: 255 registers per lane, 102 KB of LDS)
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus = n > 0 ? n : 256;
    }
    const dim3 grid((unsigned)(tiles < cus ? tiles : cus)), block(64 * policy::kWaves);
    const size_t lds = (size_t)policy::kRing * policy::kSlotBytes + (size_t)256 * 16 * ks1 * 2 + 16;  // the fragment ring, the tile's observations as bf16
    hipStream_t st = (hipStream_t)stream;
#define ACX_POLICY_LAUNCH(KS)                                                                                                                        \
    ACX_HIP_TRY(hipFuncSetAttribute((const void*)policy::k_policy_sample<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
    hipLaunchKernelGGL((policy::k_policy_sample<KS>), grid, block, lds, st, d_obs, obs_dtype == ACX_I8 ? 1 : 0, n_env, in_dim, a, c, n_actions, (unsigned long long)seed, d_action, \
                       d_logprob, d_value)
    switch (ks1) {
        case 1: ACX_POLICY_LAUNCH(1); break;
        case 2: ACX_POLICY_LAUNCH(2); break;
        case 3: ACX_POLICY_LAUNCH(3); break;
        case 4: ACX_POLICY_LAUNCH(4); break;
        default: ACX_POLICY_LAUNCH(5); break;
    }
#undef ACX_POLICY_LAUNCH
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

/* bytes of one packed network for acx_policy_sample: the fragments of the three layers (bias fragments included) */
extern "C" int64_t acx_policy_packed_bytes(int in_dim) {
    const int64_t ks1 = (in_dim + 15) / 16;
    return (8 * (ks1 + 1) + 8 * 17 + 17) * 1024;
}
