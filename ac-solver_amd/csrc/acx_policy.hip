// acx_policy.hip -- rollout-time inference of the PPO agent's two tanh MLPs (actor + critic, ac_solver/agents/ppo_agent.py:
// in -> 256 -> 256 -> {n_actions, 1}) fused with the action sampling, on the matrix cores.
//
// Why it is here: a PPO rollout step is  policy(obs) -> action -> ACEnv.step.  With the env kernel at ~9 us per 131 072
// environments, torch's eager fp32 MLP (0.7 ms per step, 0.4 ms under bf16 autocast: some twenty small launches) was 99 % of a
// rollout step (BASELINE config 5).  This is the one GEMM-shaped piece of the hot path's caller, so it gets an MFMA kernel.
//
// Formulation (one wave = 32 environments, one workgroup = 8 waves = 256 environments):
//   every layer is computed TRANSPOSED,  H^T[out, env] = W[out, in] . X^T[in, env] + b,  with v_mfma_f32_32x32x16_bf16:
//   A = a 32 x 16 tile of the nn.Linear weight (pre-packed on the host into per-lane fragments, 16 B per lane; fetched once
//   per workgroup and k-step from L2 -- the whole network is 350 KB -- into a double-buffered LDS stage that all eight
//   waves read), B = 16 inputs x 32 environments, C = 32 outputs x 32 environments
//   with the environment on the lane.  The 256 outputs of a hidden layer are eight such tiles = 128 accumulator registers;
//   bias is the accumulator's initial value, tanh is applied in registers, and the activations go through a bf16
//   [env][hidden] image in LDS (row pitch 528 B: conflict-free 16-byte fragment reads) to become the next layer's B operand.
//   The heads leave the 12 logits / the value of an environment in two lanes (l and l + 32); one shuffle gathers them and
//   lane l < 32 finishes: log-softmax, a Gumbel-max draw (counter-based hash of (seed, env, action): an exact sample of
//   Categorical(softmax(logits))), log-probability, value.
// Arithmetic: bf16 inputs, weights and activations, f32 accumulation (the reference's policy is f32 torch: the test compares
// with a torch f32 forward at bf16 tolerance and with a bf16-rounded emulation tightly).  Only inference: the PPO update
// keeps torch autograd on the f32 master weights; FusedPolicy.refresh() re-packs them after every optimizer step.
#include "acx_common.h"

namespace acx {
namespace policy {

typedef __attribute__((ext_vector_type(8))) short frag_ab;   // 8 bf16
typedef __attribute__((ext_vector_type(16))) float frag_cd;  // 32 x 32 f32 tile: 16 per lane

constexpr int kHidden = 256;
constexpr int kRowPitch = kHidden * 2 + 16;  // bytes per environment in the LDS activation image

__device__ __forceinline__ unsigned short bf16_of(float x) {  // round to nearest even
    unsigned int u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float tanh_fast(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1) with v_exp_f32 (2^y) and v_rcp_f32; overflow gives +inf -> 1, underflow 0 -> -1
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2 / ln 2
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    return x;
}

// packed network: [layer 1: 8 out-blocks x KS1 k-steps][layer 2: 8 x 16][head: 1 x 16] fragments of 64 lanes x 8 bf16,
// then the f32 biases (256 + 256 + 32)
struct Net {
    const frag_ab* w1;
    const frag_ab* w2;
    const frag_ab* w3;
    const float* b1;
    const float* b2;
    const float* b3;
};

constexpr int kWaves = 8;                                  // waves per workgroup: 256 environments share every staged weight tile
constexpr int kActBytes = kWaves * 32 * kRowPitch;         // LDS: the activation images ...
constexpr int kStageFrags = 8 * 64;                        // ... and two buffers of the 8 weight fragments (8 KB) of one k-step

// one hidden layer: acc[ob] += W[32 ob .. +32][16 ks .. +16] . B(ks) over `nks` k-steps.  The k-step's eight weight fragments
// are fetched ONCE per workgroup (thread t brings fragment t of 512), parked in LDS and read from there by all eight waves;
// the fetch of k-step ks + 1 is in flight while the MFMAs of k-step ks run (two buffers, one barrier per k-step).  Reading the
// fragments per wave straight from L2 needs 128 B/clk per CU at full MFMA rate -- twice what a CU's vector memory path
// delivers -- and left the matrix pipe at 12 % (232 -> 158 us per 131 072 environments with register prefetch alone).
template <typename BFRAG>
__device__ __forceinline__ void layer(frag_cd (&acc)[8], const frag_ab* __restrict__ w, int nks, BFRAG bfrag, frag_ab* __restrict__ stage, uint32_t tid, uint32_t lane) {
    const uint32_t ob_t = tid >> 6, lane_t = tid & 63u;
    frag_ab nxt = w[(ob_t * nks) * 64 + lane_t];
    stage[tid] = nxt;
    __syncthreads();
    for (int ks = 0; ks < nks; ks++) {
        const frag_ab* cur = stage + (ks & 1) * kStageFrags;
        if (ks + 1 < nks) nxt = w[(ob_t * nks + ks + 1) * 64 + lane_t];
        const frag_ab b = bfrag(ks);
#pragma unroll
        for (int ob = 0; ob < 8; ob++) acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[ob * 64 + lane], b, acc[ob], 0, 0, 0);
        if (ks + 1 < nks) stage[((ks + 1) & 1) * kStageFrags + tid] = nxt;
        __syncthreads();
    }
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: round to nearest even, two at a time
    f32x2 v;
    v[0] = a;
    v[1] = b;
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ void store_tanh(const frag_cd (&acc)[8], uint8_t* __restrict__ row, uint32_t h) {
    // tanh -> bf16 -> LDS image [env r][hidden]: the lane owns hidden rows (v&3) + 8 (v>>2) + 4h of each 32-block
#pragma unroll
    for (int ob = 0; ob < 8; ob++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            *(uint2*)(row + 2 * (32 * ob + 8 * g + 4 * h)) = make_uint2(pack_bf16(tanh_fast(acc[ob][4 * g]), tanh_fast(acc[ob][4 * g + 1])),
                                                                        pack_bf16(tanh_fast(acc[ob][4 * g + 2]), tanh_fast(acc[ob][4 * g + 3])));
}

// accumulators start from the bias: the lane's rows (v&3) + 8 (v>>2) + 4h of block ob are four runs of four consecutive floats
__device__ __forceinline__ void load_bias(frag_cd (&acc)[8], const float* __restrict__ b, uint32_t h) {
#pragma unroll
    for (int ob = 0; ob < 8; ob++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 v = *(const float4*)(b + 32 * ob + 8 * g + 4 * h);
            acc[ob][4 * g] = v.x;
            acc[ob][4 * g + 1] = v.y;
            acc[ob][4 * g + 2] = v.z;
            acc[ob][4 * g + 3] = v.w;
        }
}

// one network on the wave's 32 environments; returns the head tile (outputs 0..31 x 32 environments)
template <int KS1>
__device__ __forceinline__ frag_cd forward(const Net& n, const frag_ab (&x)[KS1], uint8_t* __restrict__ act, frag_ab* __restrict__ stage, uint32_t tid, uint32_t lane) {
    const uint32_t r = lane & 31u, h = lane >> 5;
    uint8_t* row = act + r * kRowPitch;
    frag_cd acc[8];
    // ---- layer 1 -----------------------------------------------------------------------------------------------------------
    load_bias(acc, n.b1, h);
    layer(acc, n.w1, KS1, [&](int ks) {
        frag_ab b = x[0];
#pragma unroll
        for (int k = 1; k < KS1; k++) b = ks == k ? x[k] : b;  // register array, no dynamic indexing
        return b;
    }, stage, tid, lane);
    store_tanh(acc, row, h);
    // ---- layer 2 (a wave reads back only its own 32 rows; the barriers of layer() order the image anyway) -----------------------
    load_bias(acc, n.b2, h);
    layer(acc, n.w2, 16, [&](int ks) { return *(const frag_ab*)(row + 2 * (16 * ks + 8 * h)); }, stage, tid, lane);
    store_tanh(acc, row, h);
    // ---- head: one output block, its 16 fragments straight from L2 -------------------------------------------------------------
    frag_cd out;
#pragma unroll
    for (int v = 0; v < 16; v++) out[v] = n.b3[(v & 3) + 8 * (v >> 2) + 4 * h];
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's rows of the image are complete
#pragma unroll 8
    for (int ks = 0; ks < 16; ks++) {
        const frag_ab b = *(const frag_ab*)(row + 2 * (16 * ks + 8 * h));
        out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(n.w3[ks * 64 + lane], b, out, 0, 0, 0);
    }
    return out;
}

template <int KS1>
__global__ void __launch_bounds__(64 * kWaves, 2) k_policy_sample(const float* __restrict__ obs, int64_t n_env, int in_dim, Net actor, Net critic, int n_actions,
                                                                 unsigned long long seed, int64_t* __restrict__ action, float* __restrict__ logprob,
                                                                 float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];  // activation images, then the two weight stages
    ACX_VGPR_PAD("v255");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, r = lane & 31u, h = lane >> 5;
    const int64_t env = ((int64_t)blockIdx.x * kWaves + wave) * 32 + r;
    uint8_t* act = s_mem + wave * 32 * kRowPitch;
    frag_ab* stage = (frag_ab*)(s_mem + kActBytes);
    // B operand of layer 1: the lane's environment, inputs 16 ks + 8 h .. + 7 (zero beyond in_dim / n_env)
    frag_ab x[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = 16 * ks + 8 * (int)h + j;
            x[ks][j] = (env < n_env && k < in_dim) ? (short)bf16_of(obs[env * in_dim + k]) : (short)0;
        }
    const frag_cd logit = forward<KS1>(actor, x, act, stage, tid, lane);
    const frag_cd val = forward<KS1>(critic, x, act, stage, tid, lane);
    // outputs 0..3 and 8..11 sit in lane half 0 (registers 0..3, 4..7), outputs 4..7 and 12..15 in half 1: gather into half 0
    float lg[16];
#pragma unroll
    for (int v = 0; v < 8; v++) {
        const float other = __shfl_xor(logit[v], 32);
        lg[(v & 3) + 8 * (v >> 2)] = logit[v];
        lg[(v & 3) + 8 * (v >> 2) + 4] = other;
    }
    if (h == 0 && env < n_env) {
        float mx = -3.0e38f;
        for (int a = 0; a < n_actions; a++) mx = fmaxf(mx, lg[a]);
        float sum = 0.0f;
        for (int a = 0; a < n_actions; a++) sum += __expf(lg[a] - mx);
        const float lse = mx + __logf(sum);
        int best = 0;
        float best_score = -3.0e38f, best_lp = 0.0f;
        for (int a = 0; a < n_actions; a++) {
            const uint64_t bits = mix64(seed ^ mix64((uint64_t)env * 16u + (uint64_t)a + 0x9e3779b97f4a7c15ull));
            const float u = ((float)(bits >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
            const float lp = lg[a] - lse;
            const float score = lp - __logf(-__logf(u));
            if (score > best_score) {
                best_score = score;
                best = a;
                best_lp = lp;
            }
        }
        action[env] = best;
        logprob[env] = best_lp;
        value[env] = val[0];
    }
}

}  // namespace policy
}  // namespace acx

using namespace acx;

extern "C" int acx_policy_sample(const float* d_obs, int64_t n_env, int in_dim, const void* d_actor, const void* d_critic, int n_actions, uint64_t seed,
                                 int64_t* d_action, float* d_logprob, float* d_value, void* stream) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (!d_obs || !d_actor || !d_critic || !d_action || !d_logprob || !d_value || n_env < 0) return fail(ACX_E_INVAL, "acx_policy_sample: bad argument");
    // 80 inputs = max_relator_length 40 (the reference trains at 36); wider first layers would not leave the register margin of ACX_VGPR_PAD
    if (in_dim < 1 || in_dim > 80 || n_actions < 1 || n_actions > 16) return fail(ACX_E_INVAL, "acx_policy_sample handles 1..80 inputs and 1..16 actions");
    if (n_env == 0) return ACX_OK;
    const int ks1 = (in_dim + 15) / 16;
    auto net = [&](const void* p) {
        policy::Net n;
        const policy::frag_ab* f = (const policy::frag_ab*)p;
        n.w1 = f;
        n.w2 = n.w1 + 8 * ks1 * 64;
        n.w3 = n.w2 + 8 * 16 * 64;
        n.b1 = (const float*)(n.w3 + 16 * 64);
        n.b2 = n.b1 + 256;
        n.b3 = n.b2 + 256;
        return n;
    };
    const policy::Net a = net(d_actor), c = net(d_critic);
    const dim3 grid((unsigned)((n_env + 32 * policy::kWaves - 1) / (32 * policy::kWaves))), block(64 * policy::kWaves);
    const size_t lds = policy::kActBytes + 2 * policy::kStageFrags * sizeof(policy::frag_ab);
    hipStream_t st = (hipStream_t)stream;
#define ACX_POLICY_LAUNCH(KS)                                                                                                                        \
    ACX_HIP_TRY(hipFuncSetAttribute((const void*)policy::k_policy_sample<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
    hipLaunchKernelGGL((policy::k_policy_sample<KS>), grid, block, lds, st, d_obs, n_env, in_dim, a, c, n_actions, (unsigned long long)seed, d_action, \
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

/* bytes of one packed network for acx_policy_sample: fragments of the three layers + the f32 biases */
extern "C" int64_t acx_policy_packed_bytes(int in_dim) {
    const int64_t ks1 = (in_dim + 15) / 16;
    return (8 * ks1 + 8 * 16 + 16) * 64 * 16 + (256 + 256 + 32) * 4;
}
