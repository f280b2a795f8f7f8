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
//   The heads leave the 12 logits / the value of an environment in two lanes (l and l + 32); one shuffle gathers them and
//   lane l < 32 finishes: log-softmax, a Gumbel-max draw (counter-based hash of (seed, env, action): an exact sample of
//   Categorical(softmax(logits))), log-probability, value.
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
    // vmcnt that retires chunk c: every wave issues three loads per chunk (fragments w and w + 8, an eighth of fragment 16)
    static constexpr int pending_after(int c) {
        int n = 0;
        for (int k = c + 1; k < C && k < c + kAhead; k++) n += 3;  // (c >= C - 1: nothing)
        return n;
    }
    // Top of loop iteration c: chunks 0 .. min(c + kAhead, C) - 1 have been issued (chunk c + kAhead only goes out AFTER this wait
    // and its barrier, into the slot chunk c - 1 leaves).  The running block reads the first fragments of chunk c + 1 before it
    // ends, so chunk c + 1 must have landed: at most the loads of the chunks issued behind it may stay in flight.
    static constexpr int issued_at_top(int c) { return c + kAhead < C ? c + kAhead : C; }
    static constexpr int pending_at_top(int c) { return issued_at_top(c) > c + 2 ? 3 * (issued_at_top(c) - (c + 2)) : 0; }
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
    const int64_t env = ((int64_t)blockIdx.x * kWaves + wave) * 32 + r;
    const uint32_t ring = (uint32_t)(uintptr_t)s_mem;  // LDS byte address (low half of the flat address)
    // B operand of layer 1: the lane's environment, inputs 16 ks + 8 h .. + 7 (zero beyond in_dim / n_env)
    frag_ab x[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = 16 * ks + 8 * (int)h + j;
            float o = 0.0f;  // (int8 observations -- what acx_env_step writes with ACX_I8 -- are a quarter of the bytes of the f32 rows)
            if (env < n_env && k < in_dim) o = obs_i8 ? (float)((const int8_t*)obs_any)[env * in_dim + k] : ((const float*)obs_any)[env * in_dim + k];
            x[ks][j] = (short)bf16_of(o);
        }
    frag_ab ones;
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = (h == 0 && j < 2) ? (short)0x3F80 : (short)0;

    const frag_ab* src_w[2] = {actor + wave * 64 + lane, critic + wave * 64 + lane};  // fragment `wave` of a chunk, this lane's 16 bytes
    const frag_ab* src_l[2] = {actor + lane, critic + lane};                              // fragment 0 of a chunk, this lane's 16 bytes
    const uint32_t ring_w = __builtin_amdgcn_readfirstlane(ring + wave * 1024u);
    auto uniform64 = [](unsigned long long v) {  // (the EXEC masks must live in scalar registers)
        return ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)v);
    };
    const unsigned long long all = uniform64(~0ull), none = uniform64(0ull);
    const unsigned long long eighth = uniform64(0xFFull << (8u * wave));
    auto issue = [&](auto ic) {  // chunk c of the stream -> ring slot c % kRing
        constexpr int c = decltype(ic)::value;
        constexpr int cc = c % P::CN, cnt = P::count(cc), slot = c % kRing, net = c < P::CN ? 0 : 1;
        constexpr size_t first = (size_t)P::first(cc) * 64;
        glds16(src_w[net] + first, ring_w + slot * kSlotBytes, all);
        glds16(src_w[net] + first + 8 * 64, ring_w + slot * kSlotBytes + 8 * 1024, cnt >= 16 ? all : uniform64(wave + 8 < (uint32_t)cnt ? ~0ull : 0ull));
        glds16(src_l[net] + first + 16 * 64, __builtin_amdgcn_readfirstlane(ring + slot * kSlotBytes + 16 * 1024), cnt == 17 ? eighth : none);
    };
    static_for<0, kAhead>(issue);

#ifdef ACX_POLICY_STAMP
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
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
            if constexpr (u == U - 1) {
#pragma unroll
                for (int i = 0; i < kFragAhead && i < NMN; i++) apre[i] = Snext[i * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        return acc;
    };
    frag_ab unused_lo, unused_hi;
    // chunk 0 has landed (for everyone: barrier) -> its first fragments
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(P::pending_after(0)) : "memory");
#pragma unroll
    for (int i = 0; i < kFragAhead && i < P::F1; i++) apre[i] = ((const frag_ab*)s_mem + lane)[i * 64];
    static_for<0, P::C>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        constexpr int net = c / P::CN, cc = c % P::CN, slot = c % kRing;
        constexpr int ccn = (c + 1) % P::CN, nm_of_next_chunk = c + 1 >= P::C ? 0 : (ccn < P::C1 ? P::F1 : 17);
        // chunk c + 1 has landed for this wave's own loads ... and, behind the barrier, for everyone's (the running block reads its
        // first fragments before it ends); chunk c - 1 is consumed, so its slot takes chunk c + kAhead
        static_assert(P::issued_at_top(c) - P::pending_at_top(c) / 3 >= (c + 2 < P::C ? c + 2 : P::C), "chunk c + 1 must have landed behind this wait");
#ifdef ACX_POLICY_NO_BARRIER  // timing experiment only (races on the ring): what the per-chunk workgroup barrier costs
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::pending_at_top(c)) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(P::pending_at_top(c)) : "memory");
#endif
#ifdef ACX_POLICY_STAMP  // diagnostic build: shader-clock stamp per chunk into the (over-allocated) logprob buffer
        logprob[n_env + (blockIdx.x & 1) * 512 + wave * 64 + c] = (float)(long long)(__builtin_amdgcn_s_memtime() - t_start);  // (no branch: all lanes store the same word)
#endif
#ifndef ACX_POLICY_NO_STREAM  // (timing experiment only when defined: the weights are not streamed beyond the ring's first fill)
        if constexpr (c + kAhead < P::C) issue(std::integral_constant<int, c + kAhead>{});
#endif
        const frag_ab* S = (const frag_ab*)(s_mem + slot * kSlotBytes) + lane;
        const frag_ab* Sn = (const frag_ab*)(s_mem + ((c + 1) % kRing) * kSlotBytes) + lane;  // first block of the next chunk
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
    const frag_cd& logit = head[0];
    const frag_cd& val = head[1];
    {
        // Head output a of an environment sits in lane half (a >> 2) & 1, register (a & 3) + 4 (a >> 3).  Lane half h draws for the eight
        // actions 8 h .. 8 h + 7: four of them are its own registers, four its partner's (lane ^ 32) -- one shuffle each; the soft-max's
        // maximum and sum are taken over the lane's eight and combined with the partner's (round 6: the sixteen outputs used to be gathered
        // into an array that both halves indexed by h, which the compiler kept in scratch memory).
        float mine[8];  // logit of action 8 h + k
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float other_lo = __shfl_xor(logit[k], 32), other_hi = __shfl_xor(logit[4 + k], 32);
            mine[k] = h ? other_hi : logit[k];          // action 8 h + k lives in half 0: registers k (h = 0: my own) and 4 + k (h = 1: the partner's)
            mine[4 + k] = h ? logit[4 + k] : other_lo;  // action 8 h + 4 + k lives in half 1
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 8; k++) mx = 8 * (int)h + k < n_actions ? fmaxf(mx, mine[k]) : mx;
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) sum += 8 * (int)h + k < n_actions ? __expf(mine[k] - mx) : 0.0f;
        sum += __shfl_xor(sum, 32);
        const float lse = mx + __logf(sum);
        int best = 0;
        float best_score = -3.0e38f, best_lp = 0.0f;
        // Gumbel-max with a counter-based 32-bit hash (two rounds of the murmur3 finaliser over (env, action), keyed by both seed
        // halves): cheap next to the 64-bit multiplies of mix64, which were a tenth of the kernel
        const uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int a = 8 * (int)h + k;
            const float lga = mine[k];
            const uint32_t bits = fmix32(fmix32(((uint32_t)env * 16u + (uint32_t)a) ^ s_lo) + s_hi + (uint32_t)(env >> 28));
            // 23 random bits: (k + 0.5) * 2^-23 is exact in f32 for every k < 2^23, so u stays strictly inside (0, 1)
            // (with 24 bits k + 0.5 rounds to 2^24 for the largest k: u = 1, an infinite Gumbel score, once per 2^24 draws)
            const float u = ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f);
            const float lp = lga - lse;
            const float score = lp - __logf(-__logf(u));
            if (a < n_actions && score > best_score) {
                best_score = score;
                best = a;
                best_lp = lp;
            }
        }
        const float o_score = __shfl_xor(best_score, 32), o_lp = __shfl_xor(best_lp, 32);
        const int o_best = __shfl_xor(best, 32);
        if (o_score > best_score || (o_score == best_score && o_best < best)) {  // (ties: the lower action, as a scan in action order would)
            best = o_best;
            best_lp = o_lp;
        }
        if (h == 0 && env < n_env) {
            action[env] = best;
            logprob[env] = best_lp;
            value[env] = val[0];
        }
#ifdef ACX_POLICY_STAMP
        logprob[n_env + (blockIdx.x & 1) * 512 + wave * 64 + 63] = (float)(long long)(__builtin_amdgcn_s_memtime() - t_start);
#endif
    }
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
    const dim3 grid((unsigned)((n_env + 32 * policy::kWaves - 1) / (32 * policy::kWaves))), block(64 * policy::kWaves);
    const size_t lds = (size_t)policy::kRing * policy::kSlotBytes;
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
