// acx_step.hip -- AC moves and the vectorised ACEnv on gfx950: kernels + their C-ABI entry points.
//
// Kernels (one presentation per lane, 64-lane waves, 256-thread workgroups):
//   k_move_bytes     byte-exact ACMove on raw int8 rows (any letters / invalid rows)   ac_moves.py:159
//   k_simplify_rows  simplify_relator on raw rows                                      utils.py:175
//   k_move_packed    ACMove on packed words; rows staged through LDS for 16-B coalesced global I/O
//   k_env_step       ACEnv.step for n resident envs; obs (int8 / f32) staged through LDS     ac_env.py:95
//   k_env_rollout    T fused ACEnv.step from an action tape, state resident in registers
//   k_env_load / k_env_observe / k_env_gather   reset / observation plumbing               ac_env.py:115
//
// Roofline: HBM (integer byte shuffling, no MFMA).  Algorithmic bytes per env step: 4L + 7
// (state in + out, action, f32 reward, done, truncated) -- see DESIGN.md.
#include <string.h>

#include <new>
#include <vector>

#include "acx_bytes.h"
#include "acx_common.h"
#include "acx_word.h"

#ifndef ACX_STEP_WAVES
#define ACX_STEP_WAVES 8  // min waves per SIMD the env step kernel is compiled for (register budget)
#endif

namespace acx {

// ---------------------------------------------------------------- host helpers (shared) -------
char* last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

bool have_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        fail(ACX_E_NODEVICE, "no HIP device visible: libacx has no CPU fallback");
        return false;
    }
    return true;
}

int Scratch::ensure(size_t bytes) {
    if (bytes <= cap) return ACX_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
    if (hipMalloc(&p, want) != hipSuccess) return fail(ACX_E_NOMEM, "hipMalloc(%zu) failed", want);
    cap = want;
    return ACX_OK;
}
Scratch::~Scratch() {}  // freed with the context at process exit
Scratch& scratch(int slot) {
    static thread_local Scratch s[4];
    return s[slot & 3];
}

// ---------------------------------------------------------------- device helpers --------------
__device__ __forceinline__ int load_action(const void* p, int dtype, int64_t k) {  // (a non-temporal load here measured 0.1 us slower per step)
    switch (dtype) {
        case ACX_U8: return ((const uint8_t*)p)[k];
        case ACX_I32: return ((const int32_t*)p)[k];
        case ACX_I64: return (int)((const int64_t*)p)[k];
        default: return ((const int8_t*)p)[k];
    }
}

// Per-wave cooperative copy of `nbytes` contiguous bytes, 16 B per lane per trip when both sides
// are 16-B aligned (`vec`), byte-wise for the ragged tail / unaligned callers.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef ACX_STEP_NT_FLAGS
#define ACX_STEP_NT_FLAGS 1  // 1: reward / done / truncated leave with non-temporal stores
#endif
template <bool NT, typename T> __device__ __forceinline__ void env_store_out(T* p, T v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
#ifndef ACX_OBS_NT
#define ACX_OBS_NT 1  // 1: observation tiles leave with non-temporal (streaming) stores: nothing re-reads them on the GPU side of a step
#endif
template <bool NT = false> __device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, int nbytes, int lane, bool vec) {
    int o = lane * 16;
    if (vec) {
        for (; o + 16 <= nbytes; o += 64 * 16) {
            if (NT) __builtin_nontemporal_store(*(const u32x4*)(src + o), (u32x4*)(dst + o));
            else *(uint4*)(dst + o) = *(const uint4*)(src + o);
        }
        if (o < nbytes)
            for (int b = o; b < nbytes && b < o + 16; b++) dst[b] = src[b];
    } else {
        for (int b = lane; b < nbytes; b += 64) dst[b] = src[b];
    }
}

__device__ __forceinline__ float clip_reward(float r, float lo, float hi) {
    return lo < hi ? fminf(fmaxf(r, lo), hi) : r;
}

// ---------------------------------------------------------------- byte-exact kernels ----------
template <int MAXL>
__global__ void __launch_bounds__(64) k_move_bytes(const int8_t* __restrict__ in, const void* __restrict__ act, int adt, int64_t n,
                                                   int L, int flags, int8_t* __restrict__ out, int32_t* __restrict__ len,
                                                   uint8_t* __restrict__ err, int32_t* __restrict__ fit) {
    if (MAXL <= 32) ACX_VGPR_PAD("v127");  // <128> already needs all 256 registers (byte-exact path: no packed-word code)
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int8_t row[2 * MAXL], res[2 * MAXL], w1[MAXL], w2[MAXL];
    for (int k = 0; k < 2 * L; k++) row[k] = in[r * 2 * L + k];
    const int a = (flags & ACX_F_NO_MOVE) ? 0 : load_action(act, adt, r);
    int lens[2] = {0, 0}, f = -1, e;
    if (a < 0 || a >= 12) e = ACX_ERR_ASSERT;  // ac_moves.py:188-190
    else e = move_bytes(row, L, a, flags, res, lens, &f, w1, w2);
    if (e != ACX_ERR_NONE) {  // the reference raised: pass the row through
        int ext;
        for (int k = 0; k < 2 * L; k++) res[k] = row[k];
        lens[0] = take_nonzero(row, L, w1, &ext);
        lens[1] = take_nonzero(row + L, L, w1, &ext);
        f = -1;
    }
    for (int k = 0; k < 2 * L; k++) out[r * 2 * L + k] = res[k];
    len[2 * r] = lens[0];
    len[2 * r + 1] = lens[1];
    err[r] = (uint8_t)e;
    if (fit) fit[r] = f;
}

template <int MAXW>
__global__ void __launch_bounds__(64) k_simplify_rows(const int8_t* __restrict__ in, int64_t n, int width, int cyclical,
                                                      int8_t* __restrict__ out, int32_t* __restrict__ len, uint8_t* __restrict__ err) {
    ACX_VGPR_PAD("v23");
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int8_t row[MAXW], res[MAXW];
    for (int k = 0; k < width; k++) row[k] = in[r * width + k];
    int nz;
    const int nn = simplify_row(row, width, cyclical != 0, res, &nz);
    for (int k = 0; k < width; k++) out[r * width + k] = nn < 0 ? row[k] : res[k];
    len[2 * r] = nn < 0 ? nz : nn;
    len[2 * r + 1] = nz;
    err[r] = nn < 0 ? (uint8_t)(-nn) : 0;
}

// ---------------------------------------------------------------- path replay ------------------
// One lane per search path: the moves of the path applied to its presentation one after the other (what the reference's search
// scripts do with the path they return, breadth_first.py:113-126 / greedy.py:130-143), the total length after every move.  A move
// on which the reference's ACMove raises ends the path: its entry and every later one read -1, err[i] says why.
constexpr int kReplayBadAction = 251;
template <typename W>
__global__ void __launch_bounds__(64) k_replay_paths(const int8_t* __restrict__ rows, int64_t n, int L, int cyclical, const int32_t* __restrict__ actions,
                                                     const int64_t* __restrict__ offsets, int32_t* __restrict__ tlen, uint8_t* __restrict__ err,
                                                     int8_t* __restrict__ final_rows) {
    ACX_VGPR_PAD_W(W, "v47", "v79");
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pres<W> s;
    bool ok = pack_relator<W>(rows + i * 2 * L, L, s.w0, s.n0);
    ok = pack_relator<W>(rows + i * 2 * L + L, L, s.w1, s.n1) && ok;
    int e = ok ? ACX_ERR_NONE : ACX_ERR_UNPACKABLE;
    for (int64_t k = offsets[i]; k < offsets[i + 1]; k++) {
        const int a = actions[k];
        if (!e) e = (a < 0 || a > 11) ? kReplayBadAction : apply_move<W, true>(s, a, L, cyclical != 0);
        tlen[k] = e ? -1 : s.n0 + s.n1;
    }
    err[i] = (uint8_t)e;
    if (final_rows && ok) {
        unpack_relator<W>(s.w0, s.n0, L, final_rows + i * 2 * L);
        unpack_relator<W>(s.w1, s.n1, L, final_rows + i * 2 * L + L);
    } else if (final_rows) {  // a row that does not unpack (err ACX_ERR_UNPACKABLE) ends as zeros, not as whatever the scratch held
        for (int k = 0; k < 2 * L; k++) final_rows[i * 2 * L + k] = 0;
    }
}

// ---------------------------------------------------------------- packed stateless kernel -----
// One wave = 64 rows = one contiguous 128*L byte tile, moved global <-> LDS with 16-B lanes; each
// lane then packs / unpacks its own row inside LDS.
template <typename W>
__global__ void __launch_bounds__(256) k_move_packed(const int8_t* __restrict__ in, const void* __restrict__ act, int adt, int64_t n,
                                                     int L, int cyclical, int8_t* __restrict__ out, int32_t* __restrict__ len,
                                                     uint8_t* __restrict__ err, int vec) {
    ACX_VGPR_PAD("v63");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int RB = 2 * L;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int rows = (int)(n - row0 < 64 ? (n - row0 < 0 ? 0 : n - row0) : 64);
    uint8_t* tile = lds + wave * 64 * RB;
    if (rows > 0) wave_copy(tile, (const uint8_t*)in + row0 * RB, rows * RB, lane, vec != 0);
    __syncthreads();
    if (lane < rows) {
        const int64_t r = row0 + lane;
        int8_t* my = (int8_t*)tile + lane * RB;
        Pres<W> s;
        bool ok = pack_relator<W>(my, L, s.w0, s.n0);
        ok = pack_relator<W>(my + L, L, s.w1, s.n1) && ok;
        const int a = load_action(act, adt, r);
        int e;
        if (!ok) {
            e = ACX_ERR_UNPACKABLE;
            s.n0 = s.n1 = 0;
            for (int k = 0; k < L; k++) {
                s.n0 += my[k] != 0;
                s.n1 += my[L + k] != 0;
            }
        } else if (a < 0 || a >= 12) {
            e = ACX_ERR_ASSERT;
        } else {
            e = apply_move<W, true>(s, a, L, cyclical != 0);
            if (e == ACX_ERR_NONE) {
                unpack_relator<W>(s.w0, s.n0, L, my);
                unpack_relator<W>(s.w1, s.n1, L, my + L);
            }
        }
        len[2 * r] = s.n0;
        len[2 * r + 1] = s.n1;
        err[r] = (uint8_t)e;
    }
    __syncthreads();
    if (rows > 0) wave_copy((uint8_t*)out + row0 * RB, tile, rows * RB, lane, vec != 0);
}

// ---------------------------------------------------------------- vectorised env --------------
// Resident state, structure of arrays (8-byte lanes -> 512-B coalesced wave accesses):
//   w0[n], w1[n]   packed relators
//   meta[n]        n0 | n1 << 8 | sticky_err << 16 | normal_form_flag << 24 | count_steps << 32
template <typename W> struct EnvDev {
    W* w0;
    W* w1;
    uint64_t* meta;
    W* iw0;  // ACEnvConfig.initial_state, packed
    W* iw1;
    uint64_t* imeta;   // n0 | n1 << 8 of the initial state
    uint8_t* hist;     // [H, n] action history ring (row = count_steps % H), or NULL
    int32_t* last_len; // [n] length of the episode that last finished (for final_info["actions"])
    int64_t n;
    int32_t L, H;
    int64_t horizon;
    float max_reward;
    // supermoves (opt-in, acx_env_set_supermoves): action 12 + s runs the base moves
    // sm_moves[sm_off[s] .. sm_off[s + 1]) as ONE environment step; null when the env has none
    const uint8_t* sm_moves;
    const int32_t* sm_off;
    int32_t n_super;
};

template <typename W> struct EnvLane {
    Pres<W> s;
    int32_t cnt;
    uint32_t err;
    uint32_t red;  // both relators non-empty, freely and cyclically reduced: the steady state of ACEnv
};

template <typename W> __device__ __forceinline__ void env_unpack_meta(uint64_t m, EnvLane<W>& v) {
    v.s.n0 = (int)(m & 0xff);
    v.s.n1 = (int)((m >> 8) & 0xff);
    v.err = (uint32_t)((m >> 16) & 0xff);
    v.red = (uint32_t)((m >> 24) & 1);
    v.cnt = (int32_t)(m >> 32);
}
template <typename W> __device__ __forceinline__ uint64_t env_pack_meta(const EnvLane<W>& v) {
    return (uint64_t)v.s.n0 | ((uint64_t)v.s.n1 << 8) | ((uint64_t)v.err << 16) | ((uint64_t)v.red << 24) | ((uint64_t)(uint32_t)v.cnt << 32);
}
template <typename W> __device__ __forceinline__ void env_load(const EnvDev<W>& e, int64_t i, EnvLane<W>& v) {
    v.s.w0 = e.w0[i];
    v.s.w1 = e.w1[i];
    env_unpack_meta<W>(e.meta[i], v);
}
template <typename W> __device__ __forceinline__ void env_store(const EnvDev<W>& e, int64_t i, const EnvLane<W>& v) {
    e.w0[i] = v.s.w0;
    e.w1[i] = v.s.w1;
    e.meta[i] = env_pack_meta<W>(v);
}

// One env transition (ac_env.py:95-113) incl. the optional gymnasium-style autoreset.
// `fin` receives the terminal state when the env finished and was reset.
// SUPER: the env may have supermoves (a separate instantiation: the plain kernels carry none of this).
template <typename W, bool SAFE, bool SUPER = false>
__device__ __forceinline__ void env_transition(const EnvDev<W>& e, int64_t i, EnvLane<W>& v, int a, bool autoreset, float clip_lo,
                                               float clip_hi, float& reward, int& done, int& trunc, bool& was_reset, Pres<W>& fin) {
    if (e.hist) e.hist[(int64_t)(v.cnt % e.H) * e.n + i] = (uint8_t)a;  // self.actions += [action], :96
    int er;  // ACMove(action, state, L, lengths) with cyclical=True, :97
    if (SUPER && a >= 12 && a < 12 + e.n_super) {
        // a supermove: its base moves one after the other, all or nothing -- if one of them raises in the reference's ACMove
        // the state is what it was before the step (as when step() raises before assigning)
        const Pres<W> s0 = v.s;
        const uint32_t red0 = v.red;
        er = ACX_ERR_NONE;
        for (int k = e.sm_off[a - 12]; k < e.sm_off[a - 11] && !er; k++) {
            const int b = e.sm_moves[k];
            er = v.red ? apply_move_reduced<W, SAFE>(v.s, b, e.L) : apply_move<W, SAFE>(v.s, b, e.L, true);
            if (!er) v.red = (uint32_t)(v.s.n0 > 0 && v.s.n1 > 0);
        }
        if (er) {
            v.s = s0;
            v.red = red0;
        }
    } else if (a < 0 || a >= 12) er = ACX_ERR_ASSERT;
    else if (v.red) er = apply_move_reduced<W, SAFE>(v.s, a, e.L);  // steady state: every state a step produces is in normal form
    else er = apply_move<W, SAFE>(v.s, a, e.L, true);
    v.err = er ? (uint32_t)er : v.err;
    v.red = er ? v.red : (uint32_t)(v.s.n0 > 0 && v.s.n1 > 0);
    const int tot = v.s.n0 + v.s.n1;
    done = tot == 2;                                                            // :101
    reward = clip_reward(done ? e.max_reward : -(float)tot, clip_lo, clip_hi);  // :102 (+ TransformReward clip)
    v.cnt += er ? 0 : 1;  // :104 (when the reference's ACMove raises, step() aborts before the counter moves)
    trunc = v.cnt >= e.horizon;                                                 // :105
    was_reset = autoreset && (done || trunc);
    if (was_reset) {  // SyncVectorEnv autoreset: ACEnv.reset() -> initial_state, :115-131
        fin = v.s;
        e.last_len[i] = v.cnt;
        v.s.w0 = e.iw0[i];
        v.s.w1 = e.iw1[i];
        const uint64_t m = e.imeta[i];
        v.s.n0 = (int)(m & 0xff);
        v.s.n1 = (int)((m >> 8) & 0xff);
        v.red = (uint32_t)((m >> 24) & 1);
        v.cnt = 0;
    }
}

// The lane's observation row (2L entries) into its LDS slot, four letters per v_perm_b32 (acx_word.h
// letters4).  LC > 0 is a compile-time max_relator_length (loops unroll, the row layout folds to constants).
// `part` of `parts`: only that contiguous share of the row's dwords is written (k_env_step_team splits a row over the four
// waves of a workgroup; part / parts are wave uniform).  parts == 1: the whole row.
template <typename W, int LC> __device__ __forceinline__ void write_obs_row(int8_t* row, const Pres<W>& s, int Lrt, int part = 0, int parts = 1) {
    uint16_t* r16 = (uint16_t*)row;  // rows are 2L bytes apart: 2-byte aligned
    auto put = [&](int L, int j) {
        const uint32_t d = row_dword<W>(s.w0, s.n0, s.w1, s.n1, L, j);
        r16[2 * j] = (uint16_t)d;
        if (4 * j + 2 < 2 * L) r16[2 * j + 1] = (uint16_t)(d >> 16);
    };
    if constexpr (LC > 0) {
        constexpr int ND = (2 * LC + 3) / 4;
        const int per = (ND + parts - 1) / parts;
#pragma unroll
        for (int j = 0; j < ND; j++)
            if (parts == 1 || (j >= part * per && j < (part + 1) * per)) put(LC, j);
    } else {
        const int nd = (2 * Lrt + 3) / 4, per = (nd + parts - 1) / parts;
        for (int j = part * per; j < nd && j < (part + 1) * per; j++) put(Lrt, j);
    }
}
template <typename W, int LC> __device__ __forceinline__ void write_obs_row(float* row, const Pres<W>& s, int Lrt, int part = 0, int parts = 1) {
    float2* r2 = (float2*)row;  // rows are 8L bytes apart: 8-byte aligned
    auto put = [&](int L, int j) {
        const uint32_t d = row_dword<W>(s.w0, s.n0, s.w1, s.n1, L, j);
        r2[2 * j] = make_float2((float)(int8_t)d, (float)(int8_t)(d >> 8));
        if (4 * j + 2 < 2 * L) r2[2 * j + 1] = make_float2((float)(int8_t)(d >> 16), (float)(int8_t)(d >> 24));
    };
    if constexpr (LC > 0) {
        constexpr int ND = (2 * LC + 3) / 4;
        const int per = (ND + parts - 1) / parts;
#pragma unroll
        for (int j = 0; j < ND; j++)
            if (parts == 1 || (j >= part * per && j < (part + 1) * per)) put(LC, j);
    } else {
        const int nd = (2 * Lrt + 3) / 4, per = (nd + parts - 1) / parts;
        for (int j = part * per; j < nd && j < (part + 1) * per; j++) put(Lrt, j);
    }
}

// LDS hand-off inside ONE wave: the wave's own ds_writes are ordered before its later ds_reads by the LDS
// queue, so no workgroup barrier (and none of the vmcnt(0) drain a __syncthreads() drags in) is needed --
// only a compiler-level fence so that the accesses are not reordered.
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef ACX_STEP_STAMP
// DIAGNOSTIC build only (tools/step_stamps.py, -DACX_STEP_STAMP through tools/build_variant.sh; in the shipped library no stamp
// executes): every wave of k_env_step logs when it started and when its last store was issued, on the 100 MHz constant
// clock (s_memrealtime).  The log is read by nothing else in the kernel and no output is computed from it.
// No shared cursor (1024 waves adding to ONE word take ~11 us): wave w of a launch owns the slots w, w + waves, w + 2 waves ...
// and keeps its own launch count in g_stamp_seq[w]; the bookkeeping runs BEHIND the end stamp.
constexpr unsigned int kStampCap = 1u << 21, kStampWaves = 1u << 16;
__device__ unsigned long long g_stamp_log[2 * kStampCap];
__device__ unsigned int g_stamp_seq[kStampWaves];
#endif

template <typename W, bool SAFE, typename OBS, int LC, bool SUPER = false>
__global__ void __launch_bounds__(256, SUPER ? 5 : ACX_STEP_WAVES) k_env_step(W* __restrict__ sw0, W* __restrict__ sw1, uint64_t* __restrict__ smeta,
                                                  const void* __restrict__ act, int64_t n_envs, int adt, EnvDev<W> e, OBS* __restrict__ obs,
                                                  float* __restrict__ rew, float clip_lo, float clip_hi, uint8_t* __restrict__ done,
                                                  uint8_t* __restrict__ trunc, OBS* __restrict__ final_obs, int autoreset, int vec) {
    if (SUPER) {
        ACX_VGPR_PAD("v111");
    } else {
        ACX_VGPR_PAD("v71");
    }
    // The first six arguments (state arrays, actions, n, action dtype = 11 dwords) are what the first memory
    // accesses need: with -mllvm -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave, so the
    // state loads issue without waiting for a kernarg fetch (this kernel is latency bound at 65 536 envs).
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
#ifdef ACX_STEP_STAMP
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = LC > 0 ? LC : e.L;
    const int RB = 2 * L * (int)sizeof(OBS);
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int rows = (int)(n_envs - row0 < 64 ? (n_envs - row0 < 0 ? 0 : n_envs - row0) : 64);
    uint8_t* tile = lds + wave * 64 * RB;  // private to this wave
    OBS* my = (OBS*)(tile + lane * RB);
    EnvLane<W> v;
    Pres<W> fin;
    bool was_reset = false;
    if (lane < rows) {
        const int64_t i = row0 + lane;
        v.s.w0 = sw0[i];
        v.s.w1 = sw1[i];
        const uint64_t m = smeta[i];
        const int a = load_action(act, adt, i);
        env_unpack_meta<W>(m, v);
        float r;
        int d, t;
        env_transition<W, SAFE, SUPER>(e, i, v, a, autoreset != 0, clip_lo, clip_hi, r, d, t, was_reset, fin);
        sw0[i] = v.s.w0;
        sw1[i] = v.s.w1;
        smeta[i] = env_pack_meta<W>(v);
        // reward and flags leave with non-temporal (streaming) stores like the observation tile: nothing on the GPU side of a step
        // re-reads them, and as ordinary stores they cost the 65 536-env step 0.3 us (3.86 -> 3.55 us per launch)
        if (rew) env_store_out<ACX_STEP_NT_FLAGS != 0>(rew + i, r);
        if (done) env_store_out<ACX_STEP_NT_FLAGS != 0>(done + i, (uint8_t)d);
        if (trunc) env_store_out<ACX_STEP_NT_FLAGS != 0>(trunc + i, (uint8_t)t);
        if (obs) write_obs_row<W, LC>(my, v.s, L);
    }
    if (obs) {
        wave_lds_handoff();
        if (rows > 0) wave_copy<ACX_OBS_NT != 0>((uint8_t*)obs + row0 * RB, tile, rows * RB, lane, vec != 0);
    }
    if (final_obs) {  // terminal observation of envs that were just reset, current observation otherwise
        wave_lds_handoff();
        if (lane < rows) write_obs_row<W, LC>(my, was_reset ? fin : v.s, L);
        wave_lds_handoff();
        if (rows > 0) wave_copy<ACX_OBS_NT != 0>((uint8_t*)final_obs + row0 * RB, tile, rows * RB, lane, vec != 0);
    }
#ifdef ACX_STEP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's stores have left (the real kernel simply ends here)
    const unsigned long long stamp_end = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        const unsigned int w = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
        if (w < kStampWaves) {
            const unsigned int seq = g_stamp_seq[w];
            const unsigned long long k = (unsigned long long)seq * nw + w;
            if (k < kStampCap) {
                g_stamp_log[2 * k] = stamp_begin;
                g_stamp_log[2 * k + 1] = stamp_end;
            }
            g_stamp_seq[w] = seq + 1;
        }
    }
#endif
}

// Small batches (one wave per SIMD: the per-wave instruction chain is what the kernel time consists of): a TEAM of four
// waves serves 64 envs.  Wave 0 does what only one wave can do -- load the state, evaluate the move, store state / reward /
// flags -- and publishes the new packed state in LDS; then all four waves unpack a quarter of every observation row into
// the LDS tile and stream a quarter of the tile out.  No work is duplicated; the serial chain of wave 0 loses three
// quarters of the unpack and of the copy-out.
template <typename W, bool SAFE, typename OBS, int LC>
__global__ void __launch_bounds__(256, 4) k_env_step_team(W* __restrict__ sw0, W* __restrict__ sw1, uint64_t* __restrict__ smeta,
                                                         const void* __restrict__ act, int64_t n_envs, int adt, EnvDev<W> e, OBS* __restrict__ obs,
                                                         float* __restrict__ rew, float clip_lo, float clip_hi, uint8_t* __restrict__ done,
                                                         uint8_t* __restrict__ trunc, int autoreset, int vec) {
    ACX_VGPR_PAD("v63");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = LC > 0 ? LC : e.L;
    const int RB = 2 * L * (int)sizeof(OBS);
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)(n_envs - row0 < 64 ? n_envs - row0 : 64);
    uint8_t* tile = lds;                       // 64 * RB bytes
    W* xs0 = (W*)(lds + ((64 * RB + 15) / 16) * 16);  // new packed state of the 64 envs
    W* xs1 = xs0 + 64;
    uint32_t* xsn = (uint32_t*)(xs1 + 64);
    if (wave == 0 && lane < rows) {
        const int64_t i = row0 + lane;
        EnvLane<W> v;
        Pres<W> fin;
        bool was_reset;
        v.s.w0 = sw0[i];
        v.s.w1 = sw1[i];
        const uint64_t m = smeta[i];
        const int a = load_action(act, adt, i);
        env_unpack_meta<W>(m, v);
        float r;
        int d, t;
        env_transition<W, SAFE>(e, i, v, a, autoreset != 0, clip_lo, clip_hi, r, d, t, was_reset, fin);
        xs0[lane] = v.s.w0;
        xs1[lane] = v.s.w1;
        xsn[lane] = (uint32_t)v.s.n0 | ((uint32_t)v.s.n1 << 8);
        sw0[i] = v.s.w0;
        sw1[i] = v.s.w1;
        smeta[i] = env_pack_meta<W>(v);
        if (rew) env_store_out<ACX_STEP_NT_FLAGS != 0>(rew + i, r);
        if (done) env_store_out<ACX_STEP_NT_FLAGS != 0>(done + i, (uint8_t)d);
        if (trunc) env_store_out<ACX_STEP_NT_FLAGS != 0>(trunc + i, (uint8_t)t);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: wave 0's global stores stay in flight
    if (lane < rows) {
        Pres<W> s;
        s.w0 = xs0[lane];
        s.w1 = xs1[lane];
        const uint32_t nn = xsn[lane];
        s.n0 = (int)(nn & 0xff);
        s.n1 = (int)(nn >> 8);
        write_obs_row<W, LC>((OBS*)(tile + lane * RB), s, L, wave, 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int nbytes = rows * RB;
    uint8_t* dst = (uint8_t*)obs + row0 * RB;
    if (vec) {
        int o = (int)threadIdx.x * 16;
        for (; o + 16 <= nbytes; o += 256 * 16) {
            if (ACX_OBS_NT) __builtin_nontemporal_store(*(const u32x4*)(tile + o), (u32x4*)(dst + o));
            else *(uint4*)(dst + o) = *(const uint4*)(tile + o);
        }
        if (o < nbytes)
            for (int b = o; b < nbytes && b < o + 16; b++) dst[b] = tile[b];
    } else {
        for (int b = (int)threadIdx.x; b < nbytes; b += 256) dst[b] = tile[b];
    }
}

template <typename W, bool SAFE, bool SUPER = false>
__global__ void __launch_bounds__(256) k_env_rollout(EnvDev<W> e, const uint8_t* __restrict__ tape, int64_t T, float* __restrict__ rew,
                                                     float clip_lo, float clip_hi, uint8_t* __restrict__ done,
                                                     uint8_t* __restrict__ trunc, int autoreset) {
    if (SUPER) {
        ACX_VGPR_PAD_W(W, "v79", "v103");
    } else {
        ACX_VGPR_PAD_W(W, "v63", "v79");
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    EnvLane<W> v;
    Pres<W> fin;
    env_load<W>(e, i, v);
    for (int64_t t = 0; t < T; t++) {
        float r;
        int d, tr;
        bool was_reset;
        env_transition<W, SAFE, SUPER>(e, i, v, tape[t * e.n + i], autoreset != 0, clip_lo, clip_hi, r, d, tr, was_reset, fin);
        if (rew) env_store_out<ACX_STEP_NT_FLAGS != 0>(rew + t * e.n + i, r);
        if (done) env_store_out<ACX_STEP_NT_FLAGS != 0>(done + t * e.n + i, (uint8_t)d);
        if (trunc) env_store_out<ACX_STEP_NT_FLAGS != 0>(trunc + t * e.n + i, (uint8_t)tr);
    }
    env_store<W>(e, i, v);
}

template <typename W, typename OBS>
__global__ void __launch_bounds__(256) k_env_observe(EnvDev<W> e, OBS* __restrict__ obs, int vec) {
    ACX_VGPR_PAD("v63");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int RB = 2 * e.L * (int)sizeof(OBS);
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int rows = (int)(e.n - row0 < 64 ? (e.n - row0 < 0 ? 0 : e.n - row0) : 64);
    uint8_t* tile = lds + wave * 64 * RB;
    if (lane < rows) {
        EnvLane<W> v;
        env_load<W>(e, row0 + lane, v);
        write_obs_row<W, 0>((OBS*)(tile + lane * RB), v.s, e.L);
    }
    wave_lds_handoff();
    if (rows > 0) wave_copy<ACX_OBS_NT != 0>((uint8_t*)obs + row0 * RB, tile, rows * RB, lane, vec != 0);
}

// rows [m, 2L] int8 (device staging) -> packed state of envs idx[k] (or env k when idx == NULL).
// to_initial: also overwrite the stored initial state.  rows == NULL: reset to the initial state.
template <typename W>
__global__ void k_env_load(EnvDev<W> e, const int8_t* __restrict__ rows, const int64_t* __restrict__ idx, int64_t m, int to_initial,
                           uint8_t* __restrict__ rowerr) {
    ACX_VGPR_PAD_W(W, "v31", "v39");
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int64_t i = idx ? idx[k] : k;
    EnvLane<W> v;
    if (rows) {
        const int8_t* r = rows + k * 2 * e.L;
        bool ok = pack_relator<W>(r, e.L, v.s.w0, v.s.n0);
        ok = pack_relator<W>(r + e.L, e.L, v.s.w1, v.s.n1) && ok;
        if (to_initial) ok = ok && v.s.n0 > 0 && v.s.n1 > 0;  // ACEnvConfig validates; reset(options=) does not
        rowerr[k] = ok ? 0 : (uint8_t)ACX_ERR_UNPACKABLE;
        if (!ok) return;
        v.red = (uint32_t)(v.s.n0 > 0 && v.s.n1 > 0 && is_cyc_reduced<W, true>(v.s.w0, v.s.n0) && is_cyc_reduced<W, true>(v.s.w1, v.s.n1));
        if (to_initial) {
            e.iw0[i] = v.s.w0;
            e.iw1[i] = v.s.w1;
            e.imeta[i] = (uint64_t)v.s.n0 | ((uint64_t)v.s.n1 << 8) | ((uint64_t)v.red << 24);
        }
    } else {
        v.s.w0 = e.iw0[i];
        v.s.w1 = e.iw1[i];
        const uint64_t mm = e.imeta[i];
        v.s.n0 = (int)(mm & 0xff);
        v.s.n1 = (int)((mm >> 8) & 0xff);
        v.red = (uint32_t)((mm >> 24) & 1);
        rowerr[k] = 0;
    }
    v.cnt = 0;
    v.err = 0;
    env_store<W>(e, i, v);
    e.last_len[i] = 0;
}

template <typename W>
__global__ void k_env_gather(EnvDev<W> e, const int64_t* __restrict__ idx, int64_t m, int8_t* __restrict__ rows, int32_t* __restrict__ len,
                             int32_t* __restrict__ cnt, uint8_t* __restrict__ err, int clear_err) {
    ACX_VGPR_PAD_W(W, "v47", "v63");
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int64_t i = idx ? idx[k] : k;
    EnvLane<W> v;
    env_load<W>(e, i, v);
    if (rows) {
        unpack_relator<W>(v.s.w0, v.s.n0, e.L, rows + k * 2 * e.L);
        unpack_relator<W>(v.s.w1, v.s.n1, e.L, rows + k * 2 * e.L + e.L);
    }
    if (len) {
        len[2 * k] = v.s.n0;
        len[2 * k + 1] = v.s.n1;
    }
    if (cnt) cnt[k] = v.cnt;
    if (err) err[k] = (uint8_t)v.err;
    if (clear_err && v.err) {
        v.err = 0;
        env_store<W>(e, i, v);
    }
}

}  // namespace acx

// =================================================================== C ABI ======================
using namespace acx;

// host-buffer entry points with at most this many rows / environments hand the pinned staging block itself to the kernels (zero copy)
constexpr int64_t kDirectRows = 64;

// pinned host staging of the host-buffer step, one grow-only buffer per host thread
static uint8_t* step_staging(size_t bytes) {
    static thread_local uint8_t* buf = nullptr;
    static thread_local size_t cap = 0;
    if (bytes <= cap) return buf;
    if (buf) (void)hipHostFree(buf);
    buf = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    // (portable + mapped: the kernels of a small call read and write this block themselves, on whichever device the thread has current)
    if (hipHostMalloc((void**)&buf, want, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return nullptr;
    cap = want;
    return buf;
}


struct acx_env {
    int64_t n;
    int L, H, flags, device;
    int64_t horizon;
    bool wide;     // W = u128
    bool safe;     // L equals the word capacity: shifts may reach the full width
    void* arena;   // one allocation holding all device arrays
    void* super_buf = nullptr;  // supermove tables (acx_env_set_supermoves)
    int n_super = 0;
    EnvDev<uint64_t> d64;
    EnvDev<u128> d128;
};

extern "C" {

int acx_version(void) { return ACX_VERSION; }
const char* acx_last_error(void) { return last_error_buf(); }
int acx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int acx_move_batch_device(const int8_t* d_in, const void* d_action, int action_dtype, int64_t n, int L, int flags, int8_t* d_out,
                          int32_t* d_len, uint8_t* d_err, int32_t* d_fit, void* stream) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n < 0 || L < 1 || !d_in || !d_out || !d_len || !d_err) return fail(ACX_E_INVAL, "acx_move_batch_device: bad argument");
    if (!d_action && !(flags & ACX_F_NO_MOVE)) return fail(ACX_E_INVAL, "acx_move_batch_device: actions required");
    if (action_dtype < ACX_U8 || action_dtype > ACX_I8) return fail(ACX_E_INVAL, "acx_move_batch_device: bad action dtype");
    if (n == 0) return ACX_OK;
    hipStream_t st = (hipStream_t)stream;
    if (flags & ACX_F_BYTES) {
        if (L > kMaxBytesL) return fail(ACX_E_INVAL, "byte path handles L <= %d, got %d", kMaxBytesL, L);
        const unsigned grid = (unsigned)ceil_div<int64_t>(n, 64);
        if (L <= 32)
            hipLaunchKernelGGL(k_move_bytes<32>, dim3(grid), dim3(64), 0, st, d_in, d_action, action_dtype, n, L, flags, d_out, d_len, d_err, d_fit);
        else
            hipLaunchKernelGGL(k_move_bytes<kMaxBytesL>, dim3(grid), dim3(64), 0, st, d_in, d_action, action_dtype, n, L, flags, d_out, d_len, d_err, d_fit);
    } else {
        if (flags & (ACX_F_NO_SIMPLIFY | ACX_F_NO_MOVE)) return fail(ACX_E_INVAL, "ACX_F_NO_SIMPLIFY / ACX_F_NO_MOVE need ACX_F_BYTES");
        if (L > 64) return fail(ACX_E_INVAL, "packed path handles L <= 64, got %d (use ACX_F_BYTES)", L);
        const unsigned grid = (unsigned)ceil_div<int64_t>(n, 256);
        const size_t lds = (size_t)4 * 64 * 2 * L;
        const int vec = aligned16(d_in) && aligned16(d_out);
        if (L <= 32)
            hipLaunchKernelGGL(k_move_packed<uint64_t>, dim3(grid), dim3(256), lds, st, d_in, d_action, action_dtype, n, L, flags & ACX_F_CYCLICAL, d_out, d_len, d_err, vec);
        else
            hipLaunchKernelGGL(k_move_packed<u128>, dim3(grid), dim3(256), lds, st, d_in, d_action, action_dtype, n, L, flags & ACX_F_CYCLICAL, d_out, d_len, d_err, vec);
    }
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_move_batch(const int8_t* h_in, const uint8_t* h_action, int64_t n, int L, int flags, int8_t* h_out, int32_t* h_len,
                   uint8_t* h_err, int32_t* h_fit) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n < 0 || L < 1 || !h_in || !h_out || !h_len || !h_err) return fail(ACX_E_INVAL, "acx_move_batch: bad argument");
    if (n == 0) return ACX_OK;
    const size_t row = (size_t)2 * L, a16 = 256;
    auto up = [&](size_t b) { return (b + a16 - 1) / a16 * a16; };
    const size_t o_in = 0, o_act = o_in + up(n * row), o_out = o_act + up(n), o_len = o_out + up(n * row), o_err = o_len + up(n * 8),
                 o_fit = o_err + up(n), total = o_fit + up(n * 4);
    Scratch& s = scratch(0);
    int rc = s.ensure(total);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)s.p;
    uint8_t* h = step_staging(total);  // pinned mirror of the block: one upload, one read-back, one synchronisation
    if (!h) return fail(ACX_E_NOMEM, "acx_move_batch: hipHostMalloc(%zu) failed", total);
    memcpy(h + o_in, h_in, (size_t)n * row);
    if (h_action) memcpy(h + o_act, h_action, (size_t)n);
    // A handful of rows (the reference's single-call surface: ACMove on ONE presentation): the kernel reads and writes the pinned block itself
    // (hipHostMalloc memory is mapped into the device's address space) -- one launch and one synchronisation, no copy engine round trips.
    const bool direct = n <= kDirectRows;
    if (direct) b = h;
    else ACX_HIP_TRY(hipMemcpyAsync(b + o_in, h + o_in, h_action ? o_act + (size_t)n : (size_t)n * row, hipMemcpyHostToDevice, nullptr));
    rc = acx_move_batch_device((const int8_t*)(b + o_in), h_action ? b + o_act : nullptr, ACX_U8, n, L, flags, (int8_t*)(b + o_out),
                               (int32_t*)(b + o_len), b + o_err, h_fit ? (int32_t*)(b + o_fit) : nullptr, nullptr);
    if (rc) return rc;
    if (!direct) ACX_HIP_TRY(hipMemcpyAsync(h + o_out, b + o_out, (h_fit ? total : o_fit) - o_out, hipMemcpyDeviceToHost, nullptr));
    ACX_HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(h_out, h + o_out, (size_t)n * row);
    memcpy(h_len, h + o_len, (size_t)n * 8);
    memcpy(h_err, h + o_err, (size_t)n);
    if (h_fit) memcpy(h_fit, h + o_fit, (size_t)n * 4);
    return ACX_OK;
}

int acx_simplify_relators(const int8_t* h_in, int64_t n, int width, int cyclical, int8_t* h_out, int32_t* h_len, uint8_t* h_err) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n < 0 || width < 0 || width > 256 || !h_out || !h_len || !h_err) return fail(ACX_E_INVAL, "acx_simplify_relators: bad argument (width <= 256)");
    if (n == 0) return ACX_OK;
    if (width == 0) {  // np.array([]) -> ([], 0)
        for (int64_t k = 0; k < n; k++) h_len[2 * k] = h_len[2 * k + 1] = 0, h_err[k] = 0;
        return ACX_OK;
    }
    const size_t a16 = 256;
    auto up = [&](size_t b) { return (b + a16 - 1) / a16 * a16; };
    const size_t o_in = 0, o_out = up(n * width), o_len = o_out + up(n * width), o_err = o_len + up(n * 8), total = o_err + up(n);
    Scratch& s = scratch(0);
    int rc = s.ensure(total);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)s.p;
    uint8_t* h = step_staging(total);
    if (!h) return fail(ACX_E_NOMEM, "acx_simplify_relators: hipHostMalloc(%zu) failed", total);
    memcpy(h + o_in, h_in, (size_t)n * width);
    ACX_HIP_TRY(hipMemcpyAsync(b + o_in, h + o_in, (size_t)n * width, hipMemcpyHostToDevice, nullptr));
    const unsigned grid = (unsigned)ceil_div<int64_t>(n, 64);
    if (width <= 64)
        hipLaunchKernelGGL(k_simplify_rows<64>, dim3(grid), dim3(64), 0, nullptr, (const int8_t*)(b + o_in), n, width, cyclical, (int8_t*)(b + o_out), (int32_t*)(b + o_len), b + o_err);
    else
        hipLaunchKernelGGL(k_simplify_rows<256>, dim3(grid), dim3(64), 0, nullptr, (const int8_t*)(b + o_in), n, width, cyclical, (int8_t*)(b + o_out), (int32_t*)(b + o_len), b + o_err);
    ACX_HIP_TRY(hipGetLastError());
    ACX_HIP_TRY(hipMemcpyAsync(h + o_out, b + o_out, total - o_out, hipMemcpyDeviceToHost, nullptr));
    ACX_HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(h_out, h + o_out, (size_t)n * width);
    memcpy(h_len, h + o_len, (size_t)n * 8);
    memcpy(h_err, h + o_err, (size_t)n);
    return ACX_OK;
}

int acx_replay_paths(const int8_t* h_presentations, int64_t n, int L, int cyclical, const int32_t* h_actions, const int64_t* h_offsets,
                     int32_t* h_total_len, uint8_t* h_err, int8_t* h_final) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n < 0 || L < 1 || L > 64 || !h_presentations || !h_offsets || !h_err) return fail(ACX_E_INVAL, "acx_replay_paths: bad argument (1 <= L <= 64)");
    if (n == 0) return ACX_OK;
    const int64_t m = h_offsets[n];
    if (h_offsets[0] != 0 || m < 0 || (m > 0 && (!h_actions || !h_total_len))) return fail(ACX_E_INVAL, "acx_replay_paths: bad offsets");
    for (int64_t i = 0; i < n; i++)
        if (h_offsets[i + 1] < h_offsets[i]) return fail(ACX_E_INVAL, "acx_replay_paths: offsets must not decrease");
    const size_t row = (size_t)2 * L, a16 = 256;
    auto up = [&](size_t b) { return (b + a16 - 1) / a16 * a16; };
    const size_t o_in = 0, o_off = up(n * row), o_act = o_off + up((n + 1) * 8), o_len = o_act + up(m * 4), o_err = o_len + up(m * 4), o_fin = o_err + up(n),
                 total = o_fin + up(n * row);
    Scratch& s = scratch(0);
    int rc = s.ensure(total);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)s.p;
    uint8_t* h = step_staging(total);  // pinned mirror: one upload, ONE launch for all paths, one read-back, one synchronisation
    if (!h) return fail(ACX_E_NOMEM, "acx_replay_paths: hipHostMalloc(%zu) failed", total);
    memcpy(h + o_in, h_presentations, (size_t)n * row);
    memcpy(h + o_off, h_offsets, (size_t)(n + 1) * 8);
    if (m) memcpy(h + o_act, h_actions, (size_t)m * 4);
    ACX_HIP_TRY(hipMemcpyAsync(b, h, o_len, hipMemcpyHostToDevice, nullptr));
    const unsigned grid = (unsigned)ceil_div<int64_t>(n, 64);
    if (L <= 32)
        hipLaunchKernelGGL(k_replay_paths<uint64_t>, dim3(grid), dim3(64), 0, nullptr, (const int8_t*)(b + o_in), n, L, cyclical, (const int32_t*)(b + o_act),
                           (const int64_t*)(b + o_off), (int32_t*)(b + o_len), b + o_err, h_final ? (int8_t*)(b + o_fin) : nullptr);
    else
        hipLaunchKernelGGL(k_replay_paths<u128>, dim3(grid), dim3(64), 0, nullptr, (const int8_t*)(b + o_in), n, L, cyclical, (const int32_t*)(b + o_act),
                           (const int64_t*)(b + o_off), (int32_t*)(b + o_len), b + o_err, h_final ? (int8_t*)(b + o_fin) : nullptr);
    ACX_HIP_TRY(hipGetLastError());
    ACX_HIP_TRY(hipMemcpyAsync(h + o_len, b + o_len, (h_final ? total : o_fin) - o_len, hipMemcpyDeviceToHost, nullptr));
    ACX_HIP_TRY(hipStreamSynchronize(nullptr));
    if (m) memcpy(h_total_len, h + o_len, (size_t)m * 4);
    memcpy(h_err, h + o_err, (size_t)n);
    if (h_final) memcpy(h_final, h + o_fin, (size_t)n * row);
    return ACX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- env ---
template <typename W> static void carve(EnvDev<W>& d, uint8_t* base, int64_t n, int L, int H, int64_t horizon, bool hist, size_t* total) {
    size_t o = 0;
    auto take = [&](size_t bytes) {
        uint8_t* p = base ? base + o : nullptr;
        o += (bytes + 255) / 256 * 256;
        return p;
    };
    d.w0 = (W*)take(n * sizeof(W));
    d.w1 = (W*)take(n * sizeof(W));
    d.iw0 = (W*)take(n * sizeof(W));
    d.iw1 = (W*)take(n * sizeof(W));
    d.meta = (uint64_t*)take(n * 8);
    d.imeta = (uint64_t*)take(n * 8);
    d.last_len = (int32_t*)take(n * 4);
    d.hist = hist ? (uint8_t*)take((size_t)H * n) : nullptr;
    d.n = n;
    d.L = L;
    d.H = H;
    d.horizon = horizon;
    d.max_reward = (float)(horizon * L * 2);
    d.sm_moves = nullptr;
    d.sm_off = nullptr;
    d.n_super = 0;
    *total = o;
}

// launch helper: picks the (word width, safe-shift) instantiation of a kernel template
#define ACX_ENV_DISPATCH(e, ...)                                          \
    do {                                                                  \
        if ((e)->wide) {                                                  \
            typedef u128 W;                                               \
            auto& dev = (e)->d128;                                        \
            if ((e)->safe) { constexpr bool SAFE = true; (void)SAFE; __VA_ARGS__; }   \
            else { constexpr bool SAFE = false; (void)SAFE; __VA_ARGS__; }            \
        } else {                                                          \
            typedef uint64_t W;                                           \
            auto& dev = (e)->d64;                                         \
            if ((e)->safe) { constexpr bool SAFE = true; (void)SAFE; __VA_ARGS__; }   \
            else { constexpr bool SAFE = false; (void)SAFE; __VA_ARGS__; }            \
        }                                                                 \
    } while (0)

extern "C" {

acx_env* acx_env_create(int64_t n, int L, int64_t horizon, int flags) {
    if (!have_device()) return nullptr;
    if (n < 1 || L < 1 || L > 64 || horizon < 1) {
        fail(ACX_E_INVAL, "acx_env_create: need n >= 1, 1 <= L <= 64, horizon >= 1");
        return nullptr;
    }
    acx_env* e = new (std::nothrow) acx_env();
    if (!e) return nullptr;
    e->n = n;
    e->L = L;
    e->horizon = horizon;
    e->flags = flags;
    e->wide = L > 32;
    e->safe = (L == 32 || L == 64);
    e->H = (int)(horizon < (1 << 20) ? horizon : (1 << 20));
    (void)hipGetDevice(&e->device);
    size_t total = 0;
    const bool hist = (flags & ACX_ENV_RECORD_ACTIONS) != 0;
    if (e->wide) carve<u128>(e->d128, nullptr, n, L, e->H, horizon, hist, &total);
    else carve<uint64_t>(e->d64, nullptr, n, L, e->H, horizon, hist, &total);
    if (hipMalloc(&e->arena, total) != hipSuccess) {
        fail(ACX_E_NOMEM, "acx_env_create: hipMalloc(%zu) failed", total);
        delete e;
        return nullptr;
    }
    (void)hipMemset(e->arena, 0, total);
    (void)hipDeviceSynchronize();  // the fill runs on the null stream; later calls come in on caller streams that may not wait for it
    if (e->wide) carve<u128>(e->d128, (uint8_t*)e->arena, n, L, e->H, horizon, hist, &total);
    else carve<uint64_t>(e->d64, (uint8_t*)e->arena, n, L, e->H, horizon, hist, &total);
    // default ACEnvConfig.initial_state is the trivial presentation <x, y> (ac_env.py:16-18)
    std::vector<int8_t> init((size_t)n * 2 * L, 0);
    for (int64_t i = 0; i < n; i++) {
        init[i * 2 * L] = 1;
        init[i * 2 * L + L] = 2;
    }
    if (acx_env_set_initial(e, init.data(), nullptr, n, nullptr) != ACX_OK) {
        acx_env_destroy(e);
        return nullptr;
    }
    return e;
}

void acx_env_destroy(acx_env* e) {
    if (!e) return;
    if (e->arena) (void)hipFree(e->arena);
    if (e->super_buf) (void)hipFree(e->super_buf);
    delete e;
}

int acx_env_set_supermoves(acx_env* e, const uint8_t* h_moves, const int32_t* h_offsets, int n_super, void* stream) {
    if (!e || n_super < 0 || n_super > 52 || (n_super > 0 && (!h_moves || !h_offsets))) return fail(ACX_E_INVAL, "acx_env_set_supermoves: 0..52 supermoves");
    if (n_super && h_offsets[0] != 0) return fail(ACX_E_INVAL, "acx_env_set_supermoves: offsets start at 0");
    for (int s = 0; s < n_super; s++) {
        if (h_offsets[s + 1] <= h_offsets[s] || h_offsets[s + 1] - h_offsets[s] > 64) return fail(ACX_E_INVAL, "acx_env_set_supermoves: a supermove has 1..64 moves");
        for (int k = h_offsets[s]; k < h_offsets[s + 1]; k++)
            if (h_moves[k] >= 12) return fail(ACX_E_INVAL, "acx_env_set_supermoves: base moves are 0..11");
    }
    hipStream_t st = (hipStream_t)stream;
    ACX_HIP_TRY(hipStreamSynchronize(st));  // steps that still read the old tables
    if (e->super_buf) (void)hipFree(e->super_buf);
    e->super_buf = nullptr;
    e->n_super = n_super;
    const uint8_t* d_moves = nullptr;
    const int32_t* d_off = nullptr;
    if (n_super) {
        const size_t nm = (size_t)h_offsets[n_super], off_bytes = (size_t)(n_super + 1) * 4;
        if (hipMalloc(&e->super_buf, off_bytes + nm) != hipSuccess) return fail(ACX_E_NOMEM, "acx_env_set_supermoves: hipMalloc failed");
        ACX_HIP_TRY(hipMemcpyAsync(e->super_buf, h_offsets, off_bytes, hipMemcpyHostToDevice, st));
        ACX_HIP_TRY(hipMemcpyAsync((uint8_t*)e->super_buf + off_bytes, h_moves, nm, hipMemcpyHostToDevice, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        d_off = (const int32_t*)e->super_buf;
        d_moves = (const uint8_t*)e->super_buf + off_bytes;
    }
    e->d64.sm_moves = e->d128.sm_moves = d_moves;
    e->d64.sm_off = e->d128.sm_off = d_off;
    e->d64.n_super = e->d128.n_super = n_super;
    return ACX_OK;
}

int64_t acx_env_max_reward(const acx_env* e) { return e ? e->horizon * e->L * 2 : 0; }

// Host-buffer entries of the environment: everything they do (uploads, the kernel, the read-back) is queued on the CALLER's
// stream and the call returns after that stream has drained, so a reset between two steps is ordered with the steps of the
// same stream also when that stream is a non-blocking side stream (torch side streams, graph capture streams are).
static int env_load_rows(acx_env* e, const int8_t* h_states, const int64_t* h_idx, int64_t m, int to_initial, hipStream_t st) {
    if (!e || m < 0 || (h_idx == nullptr && m != e->n && m != 0)) return fail(ACX_E_INVAL, "env reset: idx == NULL needs n_idx == n");
    if (m == 0) return ACX_OK;
    if (h_idx)
        for (int64_t k = 0; k < m; k++)
            if (h_idx[k] < 0 || h_idx[k] >= e->n) return fail(ACX_E_INVAL, "env reset: index %lld out of range", (long long)h_idx[k]);
    const size_t row = (size_t)2 * e->L;
    auto up = [&](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_rows = 0, o_idx = up(m * row), o_err = o_idx + up(m * 8), total = o_err + up(m);
    Scratch& s = scratch(1);
    int rc = s.ensure(total);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)s.p;
    if (h_states) ACX_HIP_TRY(hipMemcpyAsync(b + o_rows, h_states, m * row, hipMemcpyHostToDevice, st));
    if (h_idx) ACX_HIP_TRY(hipMemcpyAsync(b + o_idx, h_idx, m * 8, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)ceil_div<int64_t>(m, 256);
    const int8_t* rows = h_states ? (const int8_t*)(b + o_rows) : nullptr;
    const int64_t* idx = h_idx ? (const int64_t*)(b + o_idx) : nullptr;
    ACX_ENV_DISPATCH(e, hipLaunchKernelGGL(k_env_load<W>, dim3(grid), dim3(256), 0, st, dev, rows, idx, m, to_initial, b + o_err));
    ACX_HIP_TRY(hipGetLastError());
    std::vector<uint8_t> err((size_t)m);
    ACX_HIP_TRY(hipMemcpyAsync(err.data(), b + o_err, m, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    for (int64_t k = 0; k < m; k++)
        if (err[k]) return fail(ACX_E_ROWERR, "row %lld is not a valid presentation over {+-1,+-2} (ACEnvConfig raises ValueError)", (long long)k);
    return ACX_OK;
}

int acx_env_set_initial(acx_env* e, const int8_t* h_states, const int64_t* h_idx, int64_t n_idx, void* stream) {
    if (!h_states) return fail(ACX_E_INVAL, "acx_env_set_initial: states required");
    return env_load_rows(e, h_states, h_idx, n_idx, 1, (hipStream_t)stream);
}

int acx_env_reset(acx_env* e, const int8_t* h_states, const int64_t* h_idx, int64_t n_idx, void* stream) {
    return env_load_rows(e, h_states, h_idx, n_idx, 0, (hipStream_t)stream);
}

int acx_env_reset_device(acx_env* e, const int8_t* d_states, const int64_t* d_idx, int64_t n_idx, uint8_t* d_rowerr, void* stream) {
    if (!e || n_idx < 0 || !d_idx || !d_rowerr) return fail(ACX_E_INVAL, "acx_env_reset_device: env, indices and the row-error bytes are required");
    if (n_idx == 0) return ACX_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div<int64_t>(n_idx, 256);
    ACX_ENV_DISPATCH(e, hipLaunchKernelGGL(k_env_load<W>, dim3(grid), dim3(256), 0, st, dev, d_states, d_idx, n_idx, 0, d_rowerr));
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_env_step(acx_env* e, const void* d_actions, int action_dtype, void* d_obs, int obs_dtype, float* d_reward, float clip_lo,
                 float clip_hi, uint8_t* d_done, uint8_t* d_trunc, void* d_final_obs, int autoreset, void* stream) {
    if (!e || !d_actions) return fail(ACX_E_INVAL, "acx_env_step: env and actions required");
    if (action_dtype < ACX_U8 || action_dtype > ACX_I8) return fail(ACX_E_INVAL, "acx_env_step: bad action dtype");
    if (obs_dtype != ACX_I8 && obs_dtype != ACX_F32) return fail(ACX_E_INVAL, "acx_env_step: obs dtype must be ACX_I8 or ACX_F32");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div<int64_t>(e->n, 256);
    const bool f32 = obs_dtype == ACX_F32;
    const size_t lds = (d_obs || d_final_obs) ? (size_t)4 * 64 * 2 * e->L * (f32 ? 4 : 1) : 0;
    const int vec = aligned16(d_obs) && aligned16(d_final_obs);
#define ACX_STEP(OBS, LC)                                                                                                                    \
    ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_step<W, SAFE, OBS, LC>), dim3(grid), dim3(256), lds, st, dev.w0, dev.w1, dev.meta, d_actions, dev.n, action_dtype, dev, (OBS*)d_obs, \
                                           d_reward, clip_lo, clip_hi, d_done, d_trunc, (OBS*)d_final_obs, autoreset, vec))
    // small batches (fewer than one wave of envs per SIMD of the chip: 256 CUs x 4 SIMDs x 64 lanes = 65 536): a team of four waves per
    // 64 envs (k_env_step_team), when an observation but no terminal observation is written.  Measured: 2.66 vs 3.05 us at 16 384 envs,
    // no difference at 65 536 (every SIMD then carries a full chain either way), slower above.
    constexpr int64_t kTeamMaxEnvs = 32768;
    const bool team = d_obs && !d_final_obs && e->n <= kTeamMaxEnvs;
    const unsigned tgrid = (unsigned)ceil_div<int64_t>(e->n, 64);
    const size_t tlds = ((size_t)64 * 2 * e->L * (f32 ? 4 : 1) + 15) / 16 * 16 + 64 * (2 * (e->wide ? 16 : 8) + 4);
#define ACX_STEP_TEAM(OBS, LC)                                                                                                               \
    ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_step_team<W, SAFE, OBS, LC>), dim3(tgrid), dim3(256), tlds, st, dev.w0, dev.w1, dev.meta, d_actions, dev.n, action_dtype, dev, (OBS*)d_obs, \
                                           d_reward, clip_lo, clip_hi, d_done, d_trunc, autoreset, vec))
    if (e->n_super) {  // an env with supermoves: the general kernel with the macro loop (no unrolled / team variants)
        if (f32)
            ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_step<W, true, float, 0, true>), dim3(grid), dim3(256), lds, st, dev.w0, dev.w1, dev.meta, d_actions, dev.n,
                                                   action_dtype, dev, (float*)d_obs, d_reward, clip_lo, clip_hi, d_done, d_trunc, (float*)d_final_obs, autoreset, vec));
        else
            ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_step<W, true, int8_t, 0, true>), dim3(grid), dim3(256), lds, st, dev.w0, dev.w1, dev.meta, d_actions, dev.n,
                                                   action_dtype, dev, (int8_t*)d_obs, d_reward, clip_lo, clip_hi, d_done, d_trunc, (int8_t*)d_final_obs, autoreset, vec));
    } else if (team) {
        if (e->L == 25) {
            if (f32) ACX_STEP_TEAM(float, 25);
            else ACX_STEP_TEAM(int8_t, 25);
        } else if (e->L == 36) {
            if (f32) ACX_STEP_TEAM(float, 36);
            else ACX_STEP_TEAM(int8_t, 36);
        } else {
            if (f32) ACX_STEP_TEAM(float, 0);
            else ACX_STEP_TEAM(int8_t, 0);
        }
    } else if (e->L == 25) {  // BASELINE max_relator_length: fully unrolled observation writer
        if (f32) ACX_STEP(float, 25);
        else ACX_STEP(int8_t, 25);
    } else if (e->L == 36) {  // the reference's PPO configuration (agents/environment.py:87)
        if (f32) ACX_STEP(float, 36);
        else ACX_STEP(int8_t, 36);
    } else {
        if (f32) ACX_STEP(float, 0);
        else ACX_STEP(int8_t, 0);
    }
#undef ACX_STEP
#undef ACX_STEP_TEAM
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_env_step_host(acx_env* e, const int64_t* h_actions, int8_t* h_obs, float* h_reward, uint8_t* h_done, uint8_t* h_trunc,
                      int8_t* h_final_obs, int autoreset, uint8_t* h_err, void* stream) {
    if (!e || !h_actions) return fail(ACX_E_INVAL, "acx_env_step_host: env and actions required");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = e->n;
    const size_t row = (size_t)2 * e->L;
    auto up = [&](size_t b) { return (b + 255) / 256 * 256; };
    // one device block and its pinned mirror: [actions | obs | final obs | reward | done | truncated | err] -- ONE upload, the
    // kernel(s), ONE read-back of everything behind the actions, ONE synchronisation (round 2: an upload, a device-wide
    // synchronisation, five read-backs and a second call for the error bytes)
    const size_t o_act = 0, o_obs = up(n * 8), o_fin = o_obs + up(n * row), o_rew = o_fin + up(n * row), o_done = o_rew + up(n * 4),
                 o_trunc = o_done + up(n), o_err = o_trunc + up(n), total = o_err + up(n);
    Scratch& s = scratch(2);
    int rc = s.ensure(total);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)s.p;
    uint8_t* h = step_staging(total);
    if (!h) return fail(ACX_E_NOMEM, "acx_env_step_host: hipHostMalloc(%zu) failed", total);
    memcpy(h + o_act, h_actions, (size_t)n * 8);
    const bool direct = n <= kDirectRows;  // (a few environments -- ACEnv.step on ONE: the kernels work on the pinned block, as acx_move_batch)
    if (direct) b = h;
    else ACX_HIP_TRY(hipMemcpyAsync(b + o_act, h + o_act, (size_t)n * 8, hipMemcpyHostToDevice, st));
    rc = acx_env_step(e, b + o_act, ACX_I64, b + o_obs, ACX_I8, (float*)(b + o_rew), 0.f, 0.f, b + o_done, b + o_trunc,
                      h_final_obs ? b + o_fin : nullptr, autoreset, st);
    if (rc) return rc;
    if (h_err) {  // the sticky error bytes of this step, cleared on the way (acx_env_get_errors with clear = 1)
        const unsigned grid = (unsigned)ceil_div<int64_t>(n, 256);
        ACX_ENV_DISPATCH(e, hipLaunchKernelGGL(k_env_gather<W>, dim3(grid), dim3(256), 0, st, dev, (const int64_t*)nullptr, n, (int8_t*)nullptr,
                                                            (int32_t*)nullptr, (int32_t*)nullptr, b + o_err, 1));
        ACX_HIP_TRY(hipGetLastError());
    }
    const size_t first = h_obs || h_final_obs ? o_obs : o_rew;
    if (!direct) ACX_HIP_TRY(hipMemcpyAsync(h + first, b + first, total - first, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    if (h_obs) memcpy(h_obs, h + o_obs, (size_t)n * row);
    if (h_final_obs) memcpy(h_final_obs, h + o_fin, (size_t)n * row);
    if (h_reward) memcpy(h_reward, h + o_rew, (size_t)n * 4);
    if (h_done) memcpy(h_done, h + o_done, (size_t)n);
    if (h_trunc) memcpy(h_trunc, h + o_trunc, (size_t)n);
    if (h_err) memcpy(h_err, h + o_err, (size_t)n);
    return ACX_OK;
}

#ifdef ACX_STEP_STAMP
// diagnostic build only: the wave stamps of k_env_step since the last reset, as (begin, end) pairs of 10 ns ticks
// h_out: [launches][waves] (begin, end) pairs; *n_launches = launches of `waves` waves logged since the last reset
int acx_debug_stamps(unsigned long long* h_out, int64_t cap_pairs, int64_t waves, int64_t* n_launches, int reset) {
    unsigned int seq = 0;
    ACX_HIP_TRY(hipDeviceSynchronize());
    ACX_HIP_TRY(hipMemcpyFromSymbol(&seq, HIP_SYMBOL(g_stamp_seq), 4));  // wave 0's count
    int64_t n = seq;
    if (waves > 0 && n * waves > (int64_t)kStampCap) n = kStampCap / waves;
    *n_launches = n;
    const int64_t m = n * waves < cap_pairs ? n * waves : cap_pairs;
    if (h_out && m > 0) ACX_HIP_TRY(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_stamp_log), (size_t)m * 16));
    if (reset) {
        static unsigned int zeros[kStampWaves];
        ACX_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_seq), zeros, sizeof(zeros)));
    }
    return ACX_OK;
}
#endif

int acx_env_rollout(acx_env* e, const uint8_t* d_tape, int64_t T, float* d_reward, float clip_lo, float clip_hi, uint8_t* d_done,
                    uint8_t* d_trunc, int autoreset, void* stream) {
    if (!e || !d_tape || T < 0) return fail(ACX_E_INVAL, "acx_env_rollout: bad argument");
    if (T == 0) return ACX_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div<int64_t>(e->n, 256);
    if (e->n_super)
        ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_rollout<W, true, true>), dim3(grid), dim3(256), 0, st, dev, d_tape, T, d_reward, clip_lo, clip_hi, d_done, d_trunc,
                                                autoreset));
    else
        ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_rollout<W, SAFE>), dim3(grid), dim3(256), 0, st, dev, d_tape, T, d_reward, clip_lo, clip_hi, d_done, d_trunc,
                                                autoreset));
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_env_observe(acx_env* e, void* d_obs, int obs_dtype, void* stream) {
    if (!e || !d_obs) return fail(ACX_E_INVAL, "acx_env_observe: bad argument");
    if (obs_dtype != ACX_I8 && obs_dtype != ACX_F32) return fail(ACX_E_INVAL, "acx_env_observe: obs dtype must be ACX_I8 or ACX_F32");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div<int64_t>(e->n, 256);
    const bool f32 = obs_dtype == ACX_F32;
    const size_t lds = (size_t)4 * 64 * 2 * e->L * (f32 ? 4 : 1);
    const int vec = aligned16(d_obs);
    if (f32) ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_observe<W, float>), dim3(grid), dim3(256), lds, st, dev, (float*)d_obs, vec));
    else ACX_ENV_DISPATCH(e, hipLaunchKernelGGL((k_env_observe<W, int8_t>), dim3(grid), dim3(256), lds, st, dev, (int8_t*)d_obs, vec));
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_env_get(acx_env* e, const int64_t* h_idx, int64_t m, int8_t* h_state, int32_t* h_len, int32_t* h_count, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!e || m < 0 || (!h_idx && m != e->n)) return fail(ACX_E_INVAL, "acx_env_get: bad argument");
    if (m == 0) return ACX_OK;
    if (h_idx)
        for (int64_t k = 0; k < m; k++)
            if (h_idx[k] < 0 || h_idx[k] >= e->n) return fail(ACX_E_INVAL, "acx_env_get: index out of range");
    const size_t row = (size_t)2 * e->L;
    auto up = [&](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_rows = 0, o_idx = up(m * row), o_len = o_idx + up(m * 8), o_cnt = o_len + up(m * 8), total = o_cnt + up(m * 4);
    Scratch& s = scratch(1);
    int rc = s.ensure(total);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)s.p;
    if (h_idx) ACX_HIP_TRY(hipMemcpyAsync(b + o_idx, h_idx, m * 8, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)ceil_div<int64_t>(m, 256);
    const int64_t* idx = h_idx ? (const int64_t*)(b + o_idx) : nullptr;
    ACX_ENV_DISPATCH(e, hipLaunchKernelGGL(k_env_gather<W>, dim3(grid), dim3(256), 0, st, dev, idx, m, (int8_t*)(b + o_rows), (int32_t*)(b + o_len),
                                                        (int32_t*)(b + o_cnt), (uint8_t*)nullptr, 0));
    ACX_HIP_TRY(hipGetLastError());
    if (h_state) ACX_HIP_TRY(hipMemcpyAsync(h_state, b + o_rows, m * row, hipMemcpyDeviceToHost, st));
    if (h_len) ACX_HIP_TRY(hipMemcpyAsync(h_len, b + o_len, m * 8, hipMemcpyDeviceToHost, st));
    if (h_count) ACX_HIP_TRY(hipMemcpyAsync(h_count, b + o_cnt, m * 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    return ACX_OK;
}

int acx_env_get_actions(acx_env* e, int64_t i, int which, int32_t* h_out, int64_t cap, int64_t* n_out, void* stream) {
    if (!e || i < 0 || i >= e->n || !n_out) return fail(ACX_E_INVAL, "acx_env_get_actions: bad argument");
    if (!(e->flags & ACX_ENV_RECORD_ACTIONS)) return fail(ACX_E_INVAL, "env was created without ACX_ENV_RECORD_ACTIONS");
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* hist = e->wide ? e->d128.hist : e->d64.hist;
    int32_t cnt = 0;
    if (which) {
        ACX_HIP_TRY(hipMemcpyAsync(&cnt, (e->wide ? e->d128.last_len : e->d64.last_len) + i, 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
    } else {
        uint64_t m = 0;
        ACX_HIP_TRY(hipMemcpyAsync(&m, (e->wide ? e->d128.meta : e->d64.meta) + i, 8, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        cnt = (int32_t)(m >> 32);
    }
    *n_out = cnt;
    if (cnt > e->H) return fail(ACX_E_CAPACITY, "episode has %d steps but the history ring keeps %d", cnt, e->H);
    if (cnt > cap) return fail(ACX_E_CAPACITY, "actions buffer too small: need %d", cnt);
    if (cnt == 0) return ACX_OK;
    std::vector<uint8_t> col((size_t)cnt);
    // strided column read: row t of the ring, column i
    ACX_HIP_TRY(hipMemcpy2DAsync(col.data(), 1, hist + i, (size_t)e->n, 1, (size_t)cnt, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    for (int32_t t = 0; t < cnt; t++) h_out[t] = col[t];
    return ACX_OK;
}

int acx_env_get_errors(acx_env* e, uint8_t* h_err, int clear, void* stream) {
    if (!e || !h_err) return fail(ACX_E_INVAL, "acx_env_get_errors: bad argument");
    hipStream_t st = (hipStream_t)stream;
    Scratch& s = scratch(1);
    int rc = s.ensure((size_t)e->n);
    if (rc) return rc;
    const unsigned grid = (unsigned)ceil_div<int64_t>(e->n, 256);
    ACX_ENV_DISPATCH(e, hipLaunchKernelGGL(k_env_gather<W>, dim3(grid), dim3(256), 0, st, dev, (const int64_t*)nullptr, e->n, (int8_t*)nullptr,
                                                        (int32_t*)nullptr, (int32_t*)nullptr, (uint8_t*)s.p, clear));
    ACX_HIP_TRY(hipGetLastError());
    ACX_HIP_TRY(hipMemcpyAsync(h_err, s.p, e->n, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    return ACX_OK;
}

}  // extern "C"
