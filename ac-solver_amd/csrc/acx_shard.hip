// acx_shard.hip -- per-GPU engine of the sharded BFS frontier (multi-GPU form of bfs, breadth_first.py:15-97) and its
// C ABI (acx_shard_*).  Orchestration across ranks: ac-solver_amd/ac_solver/search/sharded.py.
//
// Round 3.  States are partitioned by hash(key) mod world.  A level is processed in chunks of consecutive global frontier
// positions [c0, c1); per chunk a rank runs, WITHOUT any host read-back in between,
//
//   k_shard_prep     frontier slice of the chunk, headers of the send regions
//   k_shard_expand   12 children per local frontier node; a child equal to its parent, a child that undoes the move that made
//                    its parent (normal-form searches with cyclical = False: that child IS the grandparent) and every
//                    duplicate inside the workgroup's 1024 children (LDS fold, smallest tag stays) are never sent; the rest
//                    goes straight into the send region of the rank that owns its key
//   (all-to-all of fixed-size regions: a header with the record count and the sender's success / error / failure words in
//    front of every region, so the exchange needs no count round trip and carries the chunk's scalars as well)
//   k_shard_insert   exact dedup of the received records in the rank's stamp table, minimum tag wins
//   k_shard_pack     one 12-bit child mask per parent: the new states this rank owns
//   (all-reduce (sum == or) of the masks: 4 bytes per PARENT)
//   k_shard_scan / k_shard_decide   prefix counts of the masks, the budget / success / error decision of the reference
//                    (breadth_first.py:84-95) -- on the device, identical on every rank, written to the control block
//   k_shard_commit   the new states below the cutoff become nodes, numbered through the masks (no sort)
//
// The receive area of a chunk is a slice of the RECORD LOG, which is never recycled: a stamp-table slot names its state by the
// word offset of the record that claimed it (fingerprint(28) | offset(36)), so a slot is written exactly once per state --
// round 2 rewrote every winner's slot with its node id at commit time, a random 8-byte store per new state (4.5 of the
// engine's 22 ms of kernel time at 1e8 nodes).  What the log costs is memory, which an MI355X has: 24 (u64 keys) or 40
// (u128) bytes per received record.
//
// Round 4: BORN stamps.  A child whose key this rank owns and whose parent is a local node never leaves the expansion kernel: it
// claims its table slot right there, with a stamp that names it as "child `action` of local node `parent`" (the fused search's
// stamp, acx_bfs.h) instead of the offset of a record -- no record is written, none is read back by the dedup, none by the commit:
// k_shard_commit_born rebuilds the winners' keys from their parents.  At world 1 that is every child (the record log shrinks to
// the regions' headers and k_shard_insert has nothing to do); at world N it is 1/N of them.  A slot's type bit tells the two
// kinds of stamp apart, and both kinds of occupant are compared by their tags, so records and born children of one chunk fold
// to the first discoverer as before.  The expansion of chunk k + 1 may run (side stream) beside the dedup of chunk k: its born
// children carry larger tags than anything in chunk k, so a record of chunk k that meets one pushes it out (and flags it in the
// flag set of chunk k + 1: the byte flags are double-buffered by chunk parity), and a born child that meets a record is "seen".
// The orchestrator keeps the expansion at most ONE chunk ahead of the dedup (an event, no host wait).
//
// The host never waits inside a level: it enqueues chunk after chunk and reads a snapshot of the control block two chunks
// late (acx_shard_ctl_snapshot / acx_shard_ctl_wait); once the status word leaves 0 every later kernel returns at once.
#include <mutex>
#include <vector>

#include "acx_bfs.h"  // search_move: the shorter move code for searches whose root is in normal form
#include "acx_owner.h"  // class_hash, inner_letter, owner_of_sum: which rank owns a state

#ifndef ACX_SHARD_SUBREGIONS
#define ACX_SHARD_SUBREGIONS 16
#endif

namespace acx {

constexpr int kShardHdr = 4;   // int64 words in front of every region: [0] records written (> capacity: overflow), [1] smallest tag of a
                               // length-2 child the sender generated, [2] smallest (tag << 8 | code) of a move the reference raises on,
                               // [3] the sender's sticky failure code (0 = healthy)
constexpr int kShardSub = ACX_SHARD_SUBREGIONS;  // sub-regions per destination: a region's cursor is ONE word and one word takes ~90 returning
                               // atomics per microsecond; the workgroups of a launch reserve in sub-region blockIdx % kShardSub
#ifndef ACX_SHARD_EXPAND_PARENTS
#define ACX_SHARD_EXPAND_PARENTS 128  // parents per workgroup of k_shard_expand: 64 (4 waves) or 128 (8 waves)
#endif
// a workgroup of k_shard_expand: kExpandParents parents x 12 actions; four waves x three actions per group of 64 parents
constexpr int kExpandParents = ACX_SHARD_EXPAND_PARENTS, kExpandItems = 3, kExpandThreads = kExpandParents * 4, kExpandTile = kExpandParents * 12;
#ifndef ACX_SHARD_FOLD_SLOTS
#define ACX_SHARD_FOLD_SLOTS (ACX_SHARD_EXPAND_PARENTS <= 64 ? 2048 : 4096)
#endif
constexpr int kFoldSlots = ACX_SHARD_FOLD_SLOTS;  // LDS fold table of a tile (a power of two >= 2 x the tile)
constexpr int kTileBits = kExpandParents <= 64 ? 10 : 11;        // bits of a tile slot / of a tag inside the tile
static_assert((kExpandParents == 64 || kExpandParents == 128) && kExpandTile <= (1 << kTileBits), "tile slots and tags fit kTileBits");
constexpr int kScanTile = 4096;                    // parents per workgroup of k_shard_scan
constexpr unsigned long long kShardInf = 1ull << 62;

template <typename W> struct recio;
template <> struct recio<uint64_t> {
    static constexpr int KW = 2, RW = 3;
    static ACX_HD void put(int64_t* r, uint64_t k0, uint64_t k1) { r[0] = (int64_t)k0; r[1] = (int64_t)k1; }
    static ACX_HD void get(const int64_t* r, uint64_t& k0, uint64_t& k1) { k0 = (uint64_t)r[0]; k1 = (uint64_t)r[1]; }
};
template <> struct recio<u128> {
    static constexpr int KW = 4, RW = 5;
    static ACX_HD void put(int64_t* r, u128 k0, u128 k1) {
        r[0] = (int64_t)(uint64_t)k0; r[1] = (int64_t)(uint64_t)(k0 >> 64);
        r[2] = (int64_t)(uint64_t)k1; r[3] = (int64_t)(uint64_t)(k1 >> 64);
    }
    static ACX_HD void get(const int64_t* r, u128& k0, u128& k1) {
        k0 = ((u128)(uint64_t)r[1] << 64) | (uint64_t)r[0];
        k1 = ((u128)(uint64_t)r[3] << 64) | (uint64_t)r[2];
    }
};

template <> struct recio<u128x> {  // max_relator_length 62 .. 64 (acx_keys.h): the same four words
    static constexpr int KW = 4, RW = 5;
    static ACX_HD void put(int64_t* r, u128x k0, u128x k1) { recio<u128>::put(r, (u128)k0, (u128)k1); }
    static ACX_HD void get(const int64_t* r, u128x& k0, u128x& k1) {
        u128 a, b;
        recio<u128>::get(r, a, b);
        k0 = u128x(a);
        k1 = u128x(b);
    }
};

// The key hash of the sharded engine: ONE value per key serves the LDS fold slot of the expansion, the stamp-table bucket and the
// 27-bit fingerprint.  k_shard_expand is bound by vector issue, and the hash of the fused search (hash_key: five 64-bit
// multiplies, i.e. fifteen quarter-rate 32-bit multiplies per child) was ~30 % of its vector cycles; this one is one
// multiply-xorshift round per key word (acx_owner.h: shard_mix).  Bits: table bucket = low bits, fold slot = bits 40..,
// fingerprint = bits 37..63.  The OWNER of a key is a different function (acx_owner.h): it only looks at the conjugacy classes of
// the two relators, so that most children are owned by the rank that makes them.
ACX_HD uint64_t shard_hash(uint64_t k0, uint64_t k1) { return shard_mix(shard_mix(0, k0), k1); }
ACX_HD uint64_t shard_hash(u128 k0, u128 k1) {
    uint64_t h = shard_mix(shard_mix(0, (uint64_t)k0), (uint64_t)(k0 >> 64));
    return shard_mix(shard_mix(h, (uint64_t)k1), (uint64_t)(k1 >> 64));
}
// what names the owner of a state (acx_owner.h): the class hashes of its two relators and their inner letters.  Every node of a
// multi-rank engine carries them (ShardDev::cls, ::inn): a child inherits what its move does not change.
struct OwnerParts {
    uint32_t c0, c1, in0, in1;
    ACX_HD uint32_t sum() const { return owner_sum(c0, c1, in0, in1); }
};
template <typename W> ACX_HD OwnerParts owner_parts_of_key(W k0, W k1) {
    OwnerParts o;
    o.c0 = class_hash<W, kSearchSafeOf<W>>(keyops<W>::word(k0), keyops<W>::len(k0));
    o.c1 = class_hash<W, kSearchSafeOf<W>>(keyops<W>::word(k1), keyops<W>::len(k1));
    o.in0 = inner_letter<W, kSearchSafeOf<W>>(keyops<W>::word(k0), keyops<W>::len(k0));
    o.in1 = inner_letter<W, kSearchSafeOf<W>>(keyops<W>::word(k1), keyops<W>::len(k1));
    return o;
}
template <typename W> ACX_HD uint32_t owner_of_key(W k0, W k1, uint32_t world) { return owner_of_sum(owner_parts_of_key<W>(k0, k1).sum(), world); }
// the parts of the child that move `a` made of a node with parts `o`, in a normal-form search: the move rewrote ONE relator (even
// action ids r_1, ac_moves.py:192-206), a conjugation (a >= 4) kept its class, and only its inner letter has to be looked up again
template <typename W> ACX_HD OwnerParts owner_parts_of_child(OwnerParts o, uint32_t a, const Pres<W>& s) {
    if (a & 1u) {
        if (a < 4u) o.c0 = class_hash<W, kSearchSafeOf<W>>(s.w0, s.n0);
        o.in0 = inner_letter<W, kSearchSafeOf<W>>(s.w0, s.n0);
    } else {
        if (a < 4u) o.c1 = class_hash<W, kSearchSafeOf<W>>(s.w1, s.n1);
        o.in1 = inner_letter<W, kSearchSafeOf<W>>(s.w1, s.n1);
    }
    return o;
}
template <typename W> ACX_HD OwnerParts owner_parts_of_pres(const Pres<W>& s) {
    return owner_parts_of_key<W>(keyops<W>::make(s.w0, s.n0), keyops<W>::make(s.w1, s.n1));
}

// ---- control block (device, int64 words; include/acx.h: ACX_SHARD_CTL_*) ---------------------------------------------------
enum : int {
    C_STATUS = 0,        // 0 running, 1 solved, 2 budget reached, 3 a move raised (AssertionError), 4 a rank failed
    C_NODES_GLOBAL = 1,  // len(tree_nodes) over all ranks
    C_NEXT_COUNT = 2,    // new states of the running level so far = size of the next level
    C_EXPANDED = 3,      // parents expanded
    C_SOLVED_TAG = 4,    // 12 * global position + action of the child that ended the search
    C_NODES = 5,         // local nodes
    C_LVL_LO = 6,        // the local nodes [lvl_lo, lvl_hi) are this rank's slice of the running level, ascending in gpos
    C_LVL_HI = 7,
    C_FAIL_LOCAL = 8,    // sticky: this rank's failure code (travels in the headers of its next chunk)
    C_MIN_LEN = 9,       // smallest total length this rank generated
    C_FAIL_SEEN = 10,    // failure code received (status 4)
    C_CHUNKS = 11,       // chunks decided
    C_LEVEL_FILL = 12,   // fullest region this rank received in the running level, in 1/256 of the even share (chunks with an even share >= 2048 records)
    C_WORDS = 16
};
enum : int { FAIL_REGION = 1, FAIL_NODES = 2, FAIL_TABLE = 3, FAIL_HOST = 4 };

// decision of the running chunk, written by k_shard_decide for k_shard_commit
struct ChunkDec {
    uint32_t commit;      // 1: turn the winners below cutoff into nodes
    uint32_t cutoff;      // relative tag: 12 * (p_end + 1 - c0)
    uint32_t node_base;   // first local id of this chunk's nodes
    uint32_t gpos_base;   // global position (inside the next level) of the chunk's first new state
};

template <typename W> struct ShardDev {
    W* k0;
    W* k1;
    int64_t* pref;       // parent_ref = rank << 40 | local id of the parent; -1 for the root
    uint32_t* gpos;      // global FIFO position inside the node's level
    uint8_t* act;
    uint8_t* tlen;
    uint2* cls;          // world > 1: class hashes of the node's two relators (acx_owner.h) -- a child made by a conjugation inherits them
    uint8_t* inn;        // world > 1: inner letters of the two relators, in0 | in1 << 4
    unsigned long long* stab;
    uint32_t stmask;
    uint32_t epoch;      // of this engine's stamps (1 .. 254)
    int64_t* log;        // record log = receive areas of all chunks
    uint32_t* tk;        // one word per PARENT of a chunk, bit a: child (parent, a) took a slot / was pushed out again (all zero between chunks).
    uint32_t* rp;        // Round 6: bits instead of one byte per tag (k_shard_pack read 24 bytes per parent on every rank, now 8).  TWO sets,
    size_t flag_stride;  // `flag_stride` words apart, indexed by the chunk's parity (ChunkGeo::par): the expansion of chunk k + 1 claims beside the dedup of chunk k
    uint32_t* lmp;       // [chunk parents] bits 0..11 = new states of this rank among the parent's children, bits 12..27 = their exclusive count over the
                         // earlier parents of the parent's kScanTile tile (k_shard_pack)
    uint2* pm;           // .x = lmp, .y = the same from the all-reduced masks (k_shard_scan writes both halves: whole, coalesced entries).  ONE
                         // 8-byte entry: a commit kernel gathers a parent's numbering from one 32-byte sector, not from two or four
    int32_t* gmask;      // the same for all ranks after the caller's all-reduce, TWO parents per word (parent p: bits 16 (p & 1) .. + 11 of word
                         // p >> 1): every (parent, action) child has one owner, so the sum of the ranks' words is their union and no field carries
    uint32_t* lblk;      // per tile: total, turned into the exclusive prefix over the tiles by k_shard_decide
    uint32_t* gblk;
    unsigned long long* ctl;
    uint32_t* bounds;    // [2][2] frontier slice (local node ids [lo, hi)) of the chunk of either parity
    ChunkDec* dec;
    uint32_t cap_nodes;
    int32_t L, cyclical;
    uint32_t world, rank;
};

struct ChunkGeo {
    int64_t c0;            // first global position of the chunk
    int64_t log_off;       // word offset of the chunk's receive area in the log
    uint32_t n_par;        // global parents in the chunk
    uint32_t subcap;       // records a sub-region can take
    uint32_t region_words; // kShardHdr + subcap * RW
    uint32_t even;         // records of the even share per sub-region (12 n_par / (world^2 * sub-regions)); 0 at world 1
    uint32_t par;          // parity of the chunk inside its level: which set of byte flags and which bounds pair it uses
};

__device__ __forceinline__ uint32_t gmask_of(const int32_t* __restrict__ gmask, uint32_t p) { return ((uint32_t)gmask[p >> 1] >> (16u * (p & 1u))) & 0xFFFu; }

// the one place that fixes the geometry of a chunk's regions: every rank (and the NumPy test engine, through
// acx_shard_layout) computes the same numbers from (parents of the chunk, world)
// Region capacity at world > 1: `fill_q8` / 256 x the even share of ALL children + two workgroups' worth (never more than the
// hard bound).  The default (fill_q8 <= 0) is 1.25 x: safe whatever the presentation does, but the records that are really
// sent (the unchanged children, the undo children and the in-tile duplicates stay home) fill ~38 % of that, and the all-to-all
// moves the whole region.  The orchestrator therefore passes the fullest region of the previous level x 1.25 + 12 / 256 of the even share
// (ctl[C_LEVEL_FILL]); an overflow fails the search (FAIL_REGION) and the orchestrator reruns it with the default.
constexpr int kShardFillDefault = 320;
constexpr int kShardFillHard = 1 << 20;  // fill_q8 at or above this: the hard bound itself (the orchestrator's last resort after two overflows)
static inline void shard_layout(int64_t n_par, int world, int RW, int fill_q8, int64_t* subcap, int64_t* region_words, int64_t* even_out = nullptr) {
    const int64_t n_blocks = (12 * n_par + kExpandTile - 1) / kExpandTile;
    const int64_t hard = (n_blocks + kShardSub - 1) / kShardSub * kExpandTile;  // every workgroup that reserves in a sub-region sends it all it has
    int64_t cap = hard, even = 0;
    if (world == 1) cap = 0;  // every child is born where it is owned: a region is its header
    if (world > 1 && fill_q8 < kShardFillHard) {
        if (fill_q8 <= 0 || fill_q8 > kShardFillDefault) fill_q8 = kShardFillDefault;
        even = (12 * n_par + (int64_t)world * world * kShardSub - 1) / ((int64_t)world * world * kShardSub);
        cap = std::min<int64_t>(hard, (even * fill_q8 + 255) / 256 + 2 * kExpandTile);
    }
    *subcap = cap;
    *region_words = kShardHdr + cap * RW;
    if (even_out) *even_out = even;
}

// first node of [lo, hi) whose gpos is >= c
__device__ __forceinline__ uint32_t lower_gpos(const uint32_t* __restrict__ gpos, uint32_t lo, uint32_t hi, uint32_t c) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (gpos[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Start of a chunk: the level switch (first chunk of a level), the frontier slice [bounds[0], bounds[1]) of the local nodes
// with gpos in [c0, c0 + n_par) -- found ONCE per chunk, not per workgroup of the expansion -- and the headers of the send regions.
template <typename W>
__global__ void __launch_bounds__(256) k_shard_prep(ShardDev<W> d, ChunkGeo g, int level_first, int64_t* __restrict__ send) {
    __shared__ uint32_t s_lo[2], s_hi[2];
    ACX_VGPR_PAD("v31");
    if (d.ctl[C_STATUS] != 0) return;
    if (threadIdx.x == 0) {
        if (level_first) {
            d.ctl[C_LVL_LO] = d.ctl[C_LVL_HI];
            d.ctl[C_LVL_HI] = d.ctl[C_NODES];
            d.ctl[C_NEXT_COUNT] = 0;
            d.ctl[C_LEVEL_FILL] = 0;
        }
        s_lo[0] = s_lo[1] = (uint32_t)d.ctl[C_LVL_LO];
        s_hi[0] = s_hi[1] = (uint32_t)d.ctl[C_LVL_HI];
    }
    __syncthreads();
    // both lower bounds by a 128-ary search (threads 0..127 the first, 128..255 the second): every round one load per thread
    // instead of the ~25 dependent loads per bound of a one-lane binary search (16 us per chunk in round 3's first profile)
    const uint32_t which = threadIdx.x >> 7, t = threadIdx.x & 127u;
    const uint32_t target = which ? (uint32_t)(g.c0 + g.n_par) : (uint32_t)g.c0;
    for (;;) {
        const uint32_t lo = s_lo[which], hi = s_hi[which], n = hi - lo;
        const bool finished = s_hi[0] == s_lo[0] && s_hi[1] == s_lo[1];  // both ranges, read by everybody BEFORE anybody narrows them
        __syncthreads();
        if (finished) break;
        if (n) {
            // probe t looks at position lo + t * step: the answer lies behind the last probe whose gpos is < target
            const uint32_t step = (n + 127u) / 128u, pos = lo + t * step;
            const bool below = pos < hi && d.gpos[pos] < target;
            const bool next_below = pos + step < hi && t + 1 < 128u && d.gpos[pos + step] < target;
            if (t == 0 && !below) s_hi[which] = lo;                       // the very first element is >= target: the bound is lo
            if (below && !next_below) {                                    // the last probe below the target
                s_lo[which] = pos + 1;
                s_hi[which] = min(hi, pos + step);
                if (step == 1) s_hi[which] = pos + 1;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d.bounds[2 * g.par] = s_lo[0];
        d.bounds[2 * g.par + 1] = s_lo[1];
    }
    const unsigned long long fail = d.ctl[C_FAIL_LOCAL];
    for (uint32_t r = threadIdx.x; r < d.world * kShardSub; r += blockDim.x) {
        int64_t* h = send + (int64_t)r * g.region_words;
        h[0] = 0;
        h[1] = (int64_t)kShardInf;
        h[2] = (int64_t)kShardInf;
        h[3] = (int64_t)fail;
    }
}

// ---- stamps -----------------------------------------------------------------------------------------------------------------
// 8-byte slots: epoch(8) | fingerprint(19, the top bits of the key's hash) | type(1) | payload(36); another epoch = free:
//   type 0 (REC)   payload = word offset, in this rank's record log, of the record that claimed the slot
//   type 1 (BORN)  payload = local id of the parent node << 4 | action: the state is that child of that node; its key is rebuilt
//                  from the parent's key in the node arena (written by the commit of an earlier level) and one move
// A slot only ever moves free -> first claimer of key K -> a claimer of K with a smaller (chunk, tag), so whatever a (possibly
// stale) plain load shows, an occupant that beats me stays beaten, and every change goes through a device-scope CAS.
constexpr unsigned long long kStampOff = (1ull << 36) - 1;   // payload
constexpr unsigned long long kStampBorn = 1ull << 36;        // type bit
// the top 27 bits of a stamp: the engine's EPOCH (8 bits; ShardDev::epoch, 1 .. 254) over a 19-bit fingerprint of the key's hash.  A
// slot whose epoch field differs is FREE: what an earlier search left in the table's memory, or the 0xFF.. of a fresh fill (the table
// comes from the pool through StampBuf, acx_frontier.h, and is not refilled from search to search); a free slot is claimed by a CAS
// on the value that was seen there.
__device__ __forceinline__ unsigned long long stamp_top(uint64_t hk, uint32_t epoch) { return ((unsigned long long)epoch << 56) | ((hk >> 45) << 37); }
__device__ __forceinline__ bool stamp_free(unsigned long long st, uint32_t epoch) { return (uint32_t)(st >> 56) != epoch; }
__device__ __forceinline__ bool stamp_fp_eq(unsigned long long a, unsigned long long b) { return (a >> 37) == (b >> 37); }  // epoch and fingerprint
__device__ __forceinline__ uint32_t born_parent(unsigned long long st) { return (uint32_t)((st & kStampOff) >> 4); }
__device__ __forceinline__ uint32_t born_action(unsigned long long st) { return (uint32_t)st & 15u; }
template <typename W, int MODE> __device__ __forceinline__ void stamp_key(const ShardDev<W>& d, unsigned long long st, W& q0, W& q1) {
    if (st & kStampBorn) {
        const uint32_t hp = born_parent(st);
        Pres<W> s;
        key_to_pres<W>(d.k0[hp], d.k1[hp], s);
        (void)search_move<W, MODE>(s, (int)born_action(st), d.L, d.cyclical != 0);
        q0 = keyops<W>::make(s.w0, s.n0);
        q1 = keyops<W>::make(s.w1, s.n1);
    } else {
        recio<W>::get(d.log + (int64_t)(st & kStampOff), q0, q1);
    }
}

// the action of item `it` (0 .. 2) of wave `w` (0 .. 3) of a group of four in k_shard_expand: one of the four concatenations
// (it == 0: actions 0 .. 3) and two of the eight conjugations per wave
__device__ __forceinline__ uint32_t expand_action(uint32_t w, int it) { return w + 4u * (uint32_t)it; }

// action that undoes action a on a presentation in normal form, cyclical = False (ac_moves.py:165-179: 0 <-> 2, 1 <-> 3 are
// r_i <- r_i r_j^{+-1}; 4 <-> 8, 5 <-> 9, 6 <-> 10, 7 <-> 11 conjugate by a generator and its inverse)
__device__ __forceinline__ uint32_t inverse_action(uint32_t a) { return a < 4 ? a ^ 2u : (a < 8 ? a + 4u : a - 4u); }

// Children of the local frontier nodes with gpos in [c0, c1), each written straight into the send region of the rank that
// owns its key.  Never sent: a child equal to its parent (over-long product, ac_moves.py:64, :126: visited by construction);
// with MODE == kMoveNf the child of action inverse(act[parent]) (it is the parent's own tree parent: g^-1 (g r g^-1) g = r and
// (r_i r_j) r_j^-1 = r_i as reduced words, and the result fits because it did before -- checked against the oracle on every
// node of the CPU suite's searches, tests/test_sharded_cpu.py); every duplicate among the workgroup's children except the one
// with the smallest tag (LDS table).  The success and error words are taken BEFORE any of that, as the reference tests a child
// before it looks it up (breadth_first.py:84).
//
// A workgroup serves 128 parents x 12 actions (8 waves; 64 x 12 with 4): wave w of a group of four, item it computes action
// w + 4 it for the group's 64 parents (round 5; 3 w + it before: every wave now has ONE concatenation, it = 0, and two
// conjugations -- the concatenation children are the ones whose owner has to be computed), ONE ACTION PER WAVE-INSTRUCTION.  The kernel is bound by vector issue (6 waves per SIMD, 27 % of the wave cycles issuing:
// profiles/r3_shard_1e8_pmc_summary.txt), and with a lane per (parent, action) in tag order -- the first version, as the fused
// k_bfs_expand_insert has it -- the twelve actions of a parent sit in adjacent lanes, so every wave ran the concatenation AND
// the conjugation path of the move for every child.  With the action uniform across the wave only the taken path issues.
// SOLO (world 1 with born stamps): every child is owned here and born here -- no owner masks, no region reservation, no record;
// the tile's took-flags leave through LDS as coalesced dwords (the chunk's local parents are ALL its parents, so the tile's tags
// are 1536 consecutive bytes), as in the fused search.
template <typename W, int MODE, bool SOLO>
__global__ void __launch_bounds__(kExpandThreads) k_shard_expand(ShardDev<W> d, ChunkGeo g, int64_t* __restrict__ send) {
    __shared__ uint32_t s_tk[kExpandParents];  // bit a of word l: child a of the tile's parent l took a slot
    __shared__ W s_k0[kExpandTile];
    __shared__ W s_k1[kExpandTile];
    __shared__ uint32_t s_slot[kFoldSlots];
    __shared__ uint32_t s_cnt[64];
    __shared__ uint32_t s_base[64];
    extern __shared__ uint32_t s_bits[];  // [world][64]: bits 0..11 the surviving actions of lane l's parent that go to owner o; later | their prefix << 16
    ACX_VGPR_PAD_W(W, "v55", "v111");  // the code needs 46-78 / 98-106 registers (tools/kernel_resources.py); the 128-bit general SOLO build had exactly
                                       // 104 with a v_lshrrev_b64 amount in v103: the build's shift check refused it
    if (d.ctl[C_STATUS] != 0) return;
    const uint32_t tid = threadIdx.x, l = (tid & 63u) + 64u * (tid >> 8), w = (tid >> 6) & 3u;  // parent slot in the workgroup; wave inside its group of four
    const uint32_t s_lo = d.bounds[2 * g.par], s_hi = d.bounds[2 * g.par + 1];
    const uint32_t np = s_hi - s_lo;
    if (blockIdx.x * kExpandParents >= np) return;
    if (tid < 64) s_cnt[tid] = 0;
    for (uint32_t i = tid; i < (uint32_t)kFoldSlots; i += kExpandThreads) s_slot[i] = kEmpty;
    if (!SOLO)
        for (uint32_t i = tid; i < d.world * (uint32_t)kExpandParents; i += kExpandThreads) s_bits[i] = 0;
    if (tid < (uint32_t)kExpandParents) s_tk[tid] = 0;
    // this lane's parent (the same for its three actions)
    const uint32_t p = blockIdx.x * kExpandParents + l, id = s_lo + p;
    const bool live = p < np;
    W pk0 = 0, pk1 = 0;
    uint32_t gp = 0, pa = 0xffu;
    OwnerParts po = {0u, 0u, 0u, 0u};  // class hashes and inner letters of the parent's relators (acx_owner.h)
    if (live) {
        pk0 = d.k0[id];
        pk1 = d.k1[id];
        gp = d.gpos[id];
        pa = d.act[id];  // 0xff for the root
        if (!SOLO) {
            const uint2 c = d.cls[id];
            const uint32_t in = d.inn[id];
            po = OwnerParts{c.x, c.y, in & 15u, in >> 4};
        }
    }
    W c0[kExpandItems], c1[kExpandItems];
    bool send_it[kExpandItems];
    uint32_t tl_min = 0xFFFFFFFFu;
    const uint32_t hsub = blockIdx.x % kShardSub;
#pragma unroll
    for (int it = 0; it < kExpandItems; it++) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)expand_action(w, it));  // uniform across the wave
        __builtin_assume(it == 0 ? a < 4u : a >= 4u);  // the item fixes the kind of move: no branch between concatenation and conjugation
        const uint32_t j = a * (uint32_t)kExpandParents + l;  // the child's slot in the tile (action major: conflict-free LDS rows)
        send_it[it] = false;
        c0[it] = c1[it] = 0;
        if (live) {
            Pres<W> s;
            key_to_pres<W>(pk0, pk1, s);
            const int e = search_move<W, MODE>(s, (int)a, d.L, d.cyclical != 0);
            const unsigned long long tag = 12ull * gp + a;  // the reference's generation order inside the level
            if (e)  // first erroring move: into the header of EVERY region of this workgroup's sub-region (rare)
                for (uint32_t o = 0; o < d.world; o++)
                    atomicMin((unsigned long long*)(send + (int64_t)(o * kShardSub + hsub) * g.region_words + 2), (tag << 8) | (unsigned long long)e);
            c0[it] = keyops<W>::make(s.w0, s.n0);
            c1[it] = keyops<W>::make(s.w1, s.n1);
            const uint32_t tl = (uint32_t)(s.n0 + s.n1);
            tl_min = min(tl_min, tl);
            if (tl == 2)
                for (uint32_t o = 0; o < d.world; o++)
                    atomicMin((unsigned long long*)(send + (int64_t)(o * kShardSub + hsub) * g.region_words + 1), tag);
            send_it[it] = !(c0[it] == pk0 && c1[it] == pk1);
            if (MODE == kMoveNf && pa < 12u && a == inverse_action(pa)) send_it[it] = false;
        }
        s_k0[j] = c0[it];
        s_k1[j] = c1[it];
    }
    {  // smallest total length: wave minimum, one atomic per wave that lowers it
        for (int o = 32; o > 0; o >>= 1) tl_min = min(tl_min, (uint32_t)__shfl_xor((int)tl_min, o));
        if ((tid & 63u) == 0 && (unsigned long long)tl_min < *(volatile unsigned long long*)(d.ctl + C_MIN_LEN)) atomicMin(d.ctl + C_MIN_LEN, (unsigned long long)tl_min);
    }
    __syncthreads();
    // ---- duplicates inside the tile: of equal keys the smallest tag stays.  A table entry is (tag inside the tile) << 10 | slot j:
    // ordered by the tag (12 * lane + action), so atomicMin keeps the first discoverer -------------------------------------
    uint32_t ls[kExpandItems], me[kExpandItems];
    uint64_t hk[kExpandItems];
#pragma unroll
    for (int it = 0; it < kExpandItems; it++) {
        const uint32_t a = expand_action(w, it), j = a * (uint32_t)kExpandParents + l;
        me[it] = ((12u * l + a) << kTileBits) | j;
        ls[it] = 0;
        hk[it] = 0;
        if (!send_it[it]) continue;
        hk[it] = shard_hash(c0[it], c1[it]);
        uint32_t q = (uint32_t)(hk[it] >> 40) & (kFoldSlots - 1);
        for (;;) {
            uint32_t v = s_slot[q];
            if (v == kEmpty) {
                v = atomicCAS(&s_slot[q], kEmpty, me[it]);
                if (v == kEmpty) break;
            }
            if (s_k0[v & ((1u << kTileBits) - 1u)] == c0[it] && s_k1[v & ((1u << kTileBits) - 1u)] == c1[it]) {  // any holder of this slot has my key
                if (v > me[it]) atomicMin(&s_slot[q], me[it]);
                break;
            }
            q = (q + 1) & (kFoldSlots - 1);
        }
        ls[it] = q;
    }
    __syncthreads();
    // ---- route the survivors.  Inside the workgroup's share of a destination's sub-region the records stand in TAG order (parent,
    // then action), although the lanes hold them action-major: the receiver's commit numbers the nodes in tag order, and its
    // stores coalesce only when consecutive records are consecutive tags (with the arrival order of an LDS counter the commit
    // took 2.7 instead of 2.0 ms per 1e8-node search).  Per owner: a 12-bit survivor mask per parent, a wave scan over the 64
    // parents, position = survivors of earlier parents + earlier actions of the own parent ------------------------------------
    uint32_t owner[kExpandItems], pos[kExpandItems];
    bool born[kExpandItems];
#pragma unroll
    for (int it = 0; it < kExpandItems; it++) {
        owner[it] = 0xFFFFFFFFu;
        pos[it] = 0;
        born[it] = false;
        if (SOLO) {
            born[it] = send_it[it] && s_slot[ls[it]] == me[it];
        } else if (send_it[it] && s_slot[ls[it]] == me[it]) {
            // what names the child's owner -- computed for the survivors of the tile's fold only (round 6; before: for every child that
            // differed from its parent).  In a normal-form search a move leaves the other relator alone and a conjugation (a >= 4) keeps
            // the class of the one it rewrites: only a concatenation (it == 0) has a class hash to compute, the others look up one inner
            // letter.  A search whose root is not in normal form (its first level simplifies BOTH relators) computes everything.
            const uint32_t a = expand_action(w, it);
            __builtin_assume(it == 0 ? a < 4u : a >= 4u);
            Pres<W> s;
            key_to_pres<W>(c0[it], c1[it], s);
            const uint32_t csum = (MODE == kMoveGeneral ? owner_parts_of_pres<W>(s) : owner_parts_of_child<W>(po, a, s)).sum();
            owner[it] = owner_of_sum(csum, d.world);
            if (owner[it] == d.rank) {  // stays home: claims its slot below, no record
                born[it] = true;
                owner[it] = 0xFFFFFFFFu;
            } else {
                atomicOr(&s_bits[owner[it] * (uint32_t)kExpandParents + l], 1u << expand_action(w, it));
            }
        }
    }
    if (!SOLO) {
    __syncthreads();
    for (uint32_t o = tid >> 6; o < d.world; o += kExpandThreads / 64) {  // every wave scans some of the owners: survivors per parent, prefix over the parents
        uint32_t run = 0;
        for (uint32_t g0 = 0; g0 < (uint32_t)kExpandParents; g0 += 64) {
            const uint32_t at = o * (uint32_t)kExpandParents + g0 + (tid & 63u);
            const uint32_t bits = s_bits[at], c = (uint32_t)__popc(bits);
            uint32_t incl = c;
            for (int sft = 1; sft < 64; sft <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, sft);
                if ((tid & 63u) >= (uint32_t)sft) incl += v;
            }
            s_bits[at] = bits | ((run + incl - c) << 16);
            run += (uint32_t)__shfl((int)incl, 63);
        }
        if ((tid & 63u) == 0) s_cnt[o] = run;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kExpandItems; it++)
        if (owner[it] != 0xFFFFFFFFu) {
            const uint32_t e = s_bits[owner[it] * (uint32_t)kExpandParents + l], a = expand_action(w, it);
            pos[it] = (e >> 16) + (uint32_t)__popc(e & ((1u << a) - 1u));
        }
    if (tid < d.world && s_cnt[tid])
        s_base[tid] = (uint32_t)atomicAdd((unsigned long long*)(send + (int64_t)(tid * kShardSub + hsub) * g.region_words), (unsigned long long)s_cnt[tid]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kExpandItems; it++) {
        if (owner[it] == 0xFFFFFFFFu) continue;
        const uint32_t at = s_base[owner[it]] + pos[it];
        if (at >= g.subcap) continue;  // overflow: the count in the header says so, the receiver reports it
        const uint32_t a = expand_action(w, it);
        int64_t* r = send + (int64_t)(owner[it] * kShardSub + hsub) * g.region_words + kShardHdr + (int64_t)at * recio<W>::RW;
        recio<W>::put(r, c0[it], c1[it]);
        // record word: parent's local id | (tag relative to the chunk) << 32
        r[recio<W>::KW] = (int64_t)(((unsigned long long)(uint32_t)(12ull * gp + a - 12ull * (unsigned long long)g.c0) << 32) | id);
    }
    }  // !SOLO
    // ---- the children this rank owns itself claim their slots here (BORN stamps): the probe loop of the fused search
    // (acx_bfs.h: k_bfs_expand_insert), a bucket of four slots = one 32-byte sector per step.  Occupants: a record (a chunk that
    // was dedup'ed before this kernel started, or -- on the main stream, beside this kernel -- the chunk before mine: smaller tags
    // either way) or a born child; a born child of MY chunk (its parent is one of this chunk's local parents: id >= s_lo) folds
    // with me by (parent, action), which IS the tag order; every other occupant with my key has been seen before me.
    uint32_t* __restrict__ tk = d.tk + g.par * d.flag_stride;
    uint32_t* __restrict__ rp = d.rp + g.par * d.flag_stride;
#pragma unroll
    for (int it = 0; it < kExpandItems; it++) {
        if (!born[it]) continue;
        const uint32_t a = expand_action(w, it);
        const unsigned long long mine = stamp_top(hk[it], d.epoch) | kStampBorn | ((unsigned long long)id << 4) | a;
        uint32_t base = (uint32_t)hk[it] & d.stmask & ~3u, probes = 0;
        bool open = true, took = false;
        while (open) {
            const ulonglong2 lo = *(const ulonglong2*)(d.stab + base), hi = *(const ulonglong2*)(d.stab + base + 2);
            const unsigned long long v0 = lo.x, v1 = lo.y, v2 = hi.x, v3 = hi.y;
            auto hot = [&](unsigned long long v) { return stamp_free(v, d.epoch) || stamp_fp_eq(v, mine); };
            uint32_t cand = (hot(v0) ? 1u : 0u) | (hot(v1) ? 2u : 0u) | (hot(v2) ? 4u : 0u) | (hot(v3) ? 8u : 0u);
            while (cand && open) {
                const uint32_t jj = (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1;
                unsigned long long st = jj == 0 ? v0 : (jj == 1 ? v1 : (jj == 2 ? v2 : v3));
                unsigned long long* slot = d.stab + base + jj;
                for (;;) {  // until this slot is decided for me (once it holds my key it only changes among holders of MY key)
                    if (stamp_free(st, d.epoch)) {  // (whoever changes a free slot first writes a stamp of THIS epoch: a failed CAS returns one)
                        const unsigned long long old = atomicCAS(slot, st, mine);
                        if (old == st) {
                            took = true;
                            open = false;
                            break;
                        }
                        st = old;
                    }
                    if (!stamp_fp_eq(st, mine)) break;  // another key's fingerprint: next slot
                    W q0, q1;
                    stamp_key<W, MODE>(d, st, q0, q1);
                    if (q0 != c0[it] || q1 != c1[it]) break;  // same fingerprint, other key: next slot
                    open = false;
                    if (!(st & kStampBorn) || born_parent(st) < s_lo || st < mine) break;  // seen before me, or a smaller tag of my chunk holds it
                    const unsigned long long old = atomicCAS(slot, st, mine);  // push the larger tag of my chunk out
                    if (old == st) {
                        took = true;
                        atomicOr(rp + (d.gpos[born_parent(st)] - (uint32_t)g.c0), 1u << born_action(st));  // no longer the first discoverer
                        break;
                    }
                    st = old;  // somebody else replaced it meanwhile: look again
                    open = true;
                }
            }
            base = (base + 4) & d.stmask;
            if (open && ++probes > d.stmask / 4) {
                atomicMax(d.ctl + C_FAIL_LOCAL, (unsigned long long)FAIL_TABLE);
                open = false;
            }
        }
        if (took) atomicOr(&s_tk[l], 1u << a);
    }
    // the took-bits of the tile's parents: ONE plain store per parent -- a chunk's word of a LOCAL parent is written by nobody else (a
    // record this rank receives is the child of another rank's parent).  SOLO: the chunk's local parents are ALL its parents, the tile's
    // words are 128 consecutive ones (coalesced); else the parent's position in the chunk is its gpos.
    __syncthreads();
    if (w == 0 && live && s_tk[l]) tk[SOLO ? p : gp - (uint32_t)g.c0] = s_tk[l];
}

// ---- dedup of the received records ------------------------------------------------------------------------------------------
// Stamp table: 8-byte slots probed four at a time (one 32-byte sector), all ones = free, else fingerprint(28) | offset(36):
// the word offset in the log of the record that claimed the slot.  An offset at or behind the running chunk's receive area is a
// record of this chunk: among equal keys the smaller tag takes the slot with a CAS (retried when another record got there
// first); an older offset is a state that was seen before.  Two byte flags per tag of the chunk, both zero on entry:
// btook[tag] is set by a record that takes a slot, brepl[tag] by the record that pushes it out again -- "took and was not
// replaced" does not depend on the order in which the two stores land.
template <typename W, int MODE> __device__ __forceinline__ void shard_insert_tile(const ShardDev<W>& d, const ChunkGeo& g, uint32_t r, uint32_t bx) {
    const int64_t roff = g.log_off + (int64_t)r * g.region_words;
    const unsigned long long written = (unsigned long long)d.log[roff];
    if (written > g.subcap && bx == 0 && threadIdx.x == 0) atomicMax(d.ctl + C_FAIL_LOCAL, (unsigned long long)FAIL_REGION);
    const uint32_t cnt = written > g.subcap ? g.subcap : (uint32_t)written;
    const uint32_t i = bx * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int64_t off = roff + kShardHdr + (int64_t)i * recio<W>::RW;
    const int64_t* rec = d.log + off;
    W c0, c1;
    recio<W>::get(rec, c0, c1);
    const uint32_t tag = (uint32_t)((unsigned long long)rec[recio<W>::KW] >> 32);
    const uint64_t hk = shard_hash(c0, c1);
    const unsigned long long me = stamp_top(hk, d.epoch) | (unsigned long long)off;
    uint32_t* __restrict__ tk = d.tk + g.par * d.flag_stride;
    uint32_t* __restrict__ rp = d.rp + g.par * d.flag_stride;
    uint32_t* __restrict__ rp_next = d.rp + (g.par ^ 1u) * d.flag_stride;
    const uint32_t b0 = d.bounds[2 * g.par], b1 = d.bounds[2 * g.par + 1];  // this chunk's local parents
    uint32_t base = (uint32_t)hk & d.stmask & ~3u, probes = 0, took = 0;
    bool open = true;
    while (open) {
        const ulonglong2 lo = *(const ulonglong2*)(d.stab + base), hi = *(const ulonglong2*)(d.stab + base + 2);
        const unsigned long long v0 = lo.x, v1 = lo.y, v2 = hi.x, v3 = hi.y;
        auto hot = [&](unsigned long long v) { return stamp_free(v, d.epoch) || stamp_fp_eq(v, me); };
        uint32_t cand = (hot(v0) ? 1u : 0u) | (hot(v1) ? 2u : 0u) | (hot(v2) ? 4u : 0u) | (hot(v3) ? 8u : 0u);
        while (cand && open) {
            const uint32_t j = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1;
            unsigned long long st = j == 0 ? v0 : (j == 1 ? v1 : (j == 2 ? v2 : v3));
            unsigned long long* slot = d.stab + base + j;
            for (;;) {  // until this slot is decided for me (once it holds my key it only changes among holders of MY key)
                if (stamp_free(st, d.epoch)) {
                    const unsigned long long old = atomicCAS(slot, st, me);
                    if (old == st) {
                        took = 1;
                        open = false;
                        break;
                    }
                    st = old;
                }
                if (!stamp_fp_eq(st, me)) break;  // another key's fingerprint: next slot
                W q0, q1;
                stamp_key<W, MODE>(d, st, q0, q1);
                if (q0 != c0 || q1 != c1) break;  // same fingerprint, other key: next slot
                open = false;
                uint32_t* flag;  // where the holder is flagged when I push it out: word of its parent, bit of its action
                uint32_t fbit;
                if (st & kStampBorn) {
                    const uint32_t hp = born_parent(st);
                    if (hp < b0) break;  // a child of an earlier chunk or level: seen before
                    const uint32_t ogp = d.gpos[hp];
                    if (hp < b1) {  // a born child of THIS chunk
                        const uint32_t otag = 12u * (ogp - (uint32_t)g.c0) + born_action(st);
                        if (otag < tag) break;
                        flag = rp + (ogp - (uint32_t)g.c0);
                    } else {  // of the NEXT chunk (its expansion runs beside me on the side stream): every tag of mine is smaller
                        flag = rp_next + (ogp - (uint32_t)g.c0 - g.n_par);
                    }
                    fbit = 1u << born_action(st);
                } else {
                    const int64_t qoff = (int64_t)(st & kStampOff);
                    if (qoff < g.log_off) break;  // a record of an earlier chunk: seen before
                    const uint32_t qtag = (uint32_t)((unsigned long long)d.log[qoff + recio<W>::KW] >> 32);
                    if (qtag < tag) break;  // a record of this chunk with a smaller tag holds it
                    flag = rp + qtag / 12u;
                    fbit = 1u << (qtag % 12u);
                }
                const unsigned long long old = atomicCAS(slot, st, me);  // push the larger tag out
                if (old == st) {
                    took = 1;
                    atomicOr(flag, fbit);  // no longer the first discoverer
                    break;
                }
                st = old;  // somebody else replaced it meanwhile: look again
                open = true;
            }
        }
        base = (base + 4) & d.stmask;
        if (open && ++probes > d.stmask / 4) {
            atomicMax(d.ctl + C_FAIL_LOCAL, (unsigned long long)FAIL_TABLE);
            open = false;
        }
    }
    if (took) atomicOr(tk + tag / 12u, 1u << (tag % 12u));
}

// grid (x, regions): x = tiles per region, or FEWER -- then a workgroup walks the tiles of its region x, x + gridDim.x, ...: a
// bounded number of resident workgroups, so that an expansion on the side stream finds room beside them (ShardEngine::insert_wgs).
template <typename W, int MODE>
__global__ void __launch_bounds__(256) k_shard_insert(ShardDev<W> d, ChunkGeo g, uint32_t tiles) {
    ACX_VGPR_PAD_W(W, "v63", "v95");
    if (d.ctl[C_STATUS] != 0) return;
    for (uint32_t bx = blockIdx.x; bx < tiles; bx += gridDim.x) shard_insert_tile<W, MODE>(d, g, blockIdx.y, bx);
}

// One word per parent of the chunk: bits 0..11 = the children (parent, a) that are new states of this rank (took a slot and were not
// pushed out again), bits 12..27 = how many such children the earlier parents of the same kScanTile tile have; the tile's total goes
// to lblk.  The flag words it read are zeroed again for the next chunk of the same parity (no memset launches).  Two parents share a
// word of gmask (the all-reduce's buffer).  Round 6: the flags are one word per parent (8 bytes read per parent instead of 24) and the
// local prefix is computed here (k_shard_scan only has the all-reduced masks left): a workgroup = one tile, four parents per lane.
template <typename W> __global__ void __launch_bounds__(1024) k_shard_pack(ShardDev<W> d, uint32_t n_par, uint32_t par) {
    __shared__ uint32_t s_w[16];
    ACX_VGPR_PAD("v39");
    if (d.ctl[C_STATUS] != 0) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t p0 = blockIdx.x * kScanTile + tid * 4;  // (the arrays are padded by a tile: words behind the chunk's last parent are zero and stay zero)
    uint4* t = (uint4*)(d.tk + par * d.flag_stride + p0);
    uint4* r = (uint4*)(d.rp + par * d.flag_stride + p0);
    const uint4 tv = *t, rv = *r;
    if (tv.x | tv.y | tv.z | tv.w) *t = make_uint4(0, 0, 0, 0);
    if (rv.x | rv.y | rv.z | rv.w) *r = make_uint4(0, 0, 0, 0);
    uint32_t m[4] = {tv.x & ~rv.x & 0xFFFu, tv.y & ~rv.y & 0xFFFu, tv.z & ~rv.z & 0xFFFu, tv.w & ~rv.w & 0xFFFu};
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (p0 + k >= n_par) m[k] = 0;
    const uint32_t sum = (uint32_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]));
    uint32_t incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = incl - sum, total = 0;
#pragma unroll
    for (uint32_t w2 = 0; w2 < 16; w2++) {
        if (w2 < wave) before += s_w[w2];
        total += s_w[w2];
    }
    if (p0 < n_par) {
        uint4 o;
        o.x = m[0] | (before << 12);
        before += (uint32_t)__popc(m[0]);
        o.y = m[1] | (before << 12);
        before += (uint32_t)__popc(m[1]);
        o.z = m[2] | (before << 12);
        before += (uint32_t)__popc(m[2]);
        o.w = m[3] | (before << 12);
        *(uint4*)(d.lmp + p0) = o;  // (padded to whole quads)
        *(int2*)(d.gmask + (p0 >> 1)) = make_int2((int)(m[0] | (m[1] << 16)), (int)(m[2] | (m[3] << 16)));
    }
    if (tid == 0) d.lblk[blockIdx.x] = total;
}

// exclusive popcount prefixes of the all-reduced masks inside tiles of kScanTile parents (pm[].y: mask | prefix << 12) + the tiles' totals
template <typename W> __global__ void __launch_bounds__(1024) k_shard_scan(ShardDev<W> d, uint32_t n_par) {
    __shared__ uint32_t s_g[16];
    ACX_VGPR_PAD("v39");
    if (d.ctl[C_STATUS] != 0) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t p0 = blockIdx.x * kScanTile + tid * 4;
    uint32_t gm[4] = {0u, 0u, 0u, 0u};
    if (p0 < n_par) {  // (gmask is padded to whole quads; the masks behind the chunk's last parent are dropped here)
        const int2 b = *(const int2*)(d.gmask + (p0 >> 1));  // four parents = two words
        gm[0] = (uint32_t)b.x & 0xFFFu, gm[1] = ((uint32_t)b.x >> 16) & 0xFFFu, gm[2] = (uint32_t)b.y & 0xFFFu, gm[3] = ((uint32_t)b.y >> 16) & 0xFFFu;
#pragma unroll
        for (int k = 1; k < 4; k++)
            if (p0 + k >= n_par) gm[k] = 0;
    }
    const uint32_t gsum = (uint32_t)(__popc(gm[0]) + __popc(gm[1]) + __popc(gm[2]) + __popc(gm[3]));
    uint32_t gi = gsum;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t b = (uint32_t)__shfl_up((int)gi, o);
        if (lane >= (uint32_t)o) gi += b;
    }
    if (lane == 63) s_g[wave] = gi;
    __syncthreads();
    uint32_t ge = gi - gsum, gt = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        if (w < wave) ge += s_g[w];
        gt += s_g[w];
    }
    if (p0 < n_par) {
        uint4 o;
        o.x = gm[0] | (ge << 12);
        ge += (uint32_t)__popc(gm[0]);
        o.y = gm[1] | (ge << 12);
        ge += (uint32_t)__popc(gm[1]);
        o.z = gm[2] | (ge << 12);
        ge += (uint32_t)__popc(gm[2]);
        o.w = gm[3] | (ge << 12);
        const uint4 l = *(const uint4*)(d.lmp + p0);
        uint4* e = (uint4*)(d.pm + p0);
        e[0] = make_uint4(l.x, o.x, l.y, o.y);
        e[1] = make_uint4(l.z, o.z, l.w, o.w);
    }
    if (tid == 0) d.gblk[blockIdx.x] = gt;
}

// The reference's decisions for the chunk, from the all-reduced masks and the received headers -- the same numbers on every rank:
//   budget   "first parent after which len(tree_nodes) >= max_nodes" (breadth_first.py:91-95): the first parent whose inclusive
//            count of new states reaches what is left of the budget;
//   success  the smallest tag of a length-2 child, if its parent is expanded at all (:84-85);
//   error    a move on which the reference's ACMove raises, if the reference gets that far.
// One workgroup: prefix over the tiles' totals (turned into exclusive prefixes in place), then lane 0 decides.
template <typename W> __global__ void __launch_bounds__(1024) k_shard_decide(ShardDev<W> d, ChunkGeo g, int64_t max_nodes) {
    __shared__ uint32_t s_l[1024], s_g[1024];
    __shared__ unsigned long long s_solved, s_err, s_fail, s_fill;
    __shared__ uint32_t s_tile, s_pb;
    ACX_VGPR_PAD("v39");
    if (d.ctl[C_STATUS] != 0) {  // the search has ended: this chunk commits nothing
        if (threadIdx.x == 0) d.dec->commit = 0;
        return;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t nt = (g.n_par + kScanTile - 1) / kScanTile, per = (nt + 1023) / 1024;
    if (tid == 0) s_solved = kShardInf, s_err = kShardInf, s_fail = 0, s_fill = 0, s_tile = 0xFFFFFFFFu, s_pb = 0xFFFFFFFFu;
    uint32_t lsum = 0, gsum = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t t = tid * per + k;
        if (t < nt) lsum += d.lblk[t], gsum += d.gblk[t];
    }
    s_l[tid] = lsum;
    s_g[tid] = gsum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele over the 1024 partial sums
        const uint32_t a = tid >= o ? s_l[tid - o] : 0u, b = tid >= o ? s_g[tid - o] : 0u;
        __syncthreads();
        s_l[tid] += a;
        s_g[tid] += b;
        __syncthreads();
    }
    const uint32_t l_total = s_l[1023], g_total = s_g[1023];
    uint32_t le = s_l[tid] - lsum, ge = s_g[tid] - gsum;
    const unsigned long long nodes_global = d.ctl[C_NODES_GLOBAL];
    const unsigned long long need = (unsigned long long)max_nodes > nodes_global ? (unsigned long long)max_nodes - nodes_global : 0ull;
    for (uint32_t k = 0; k < per; k++) {  // exclusive prefixes of the tiles in place; the tile in which the budget is reached
        const uint32_t t = tid * per + k;
        if (t < nt) {
            const uint32_t lt = d.lblk[t], gt = d.gblk[t];
            d.lblk[t] = le;
            d.gblk[t] = ge;
            if (need >= 1 && (unsigned long long)ge < need && (unsigned long long)ge + gt >= need) atomicMin(&s_tile, t);
            le += lt;
            ge += gt;
        }
    }
    // the headers of everything this rank received: success / error words and failure codes of all senders
    unsigned long long sv = kShardInf, ev = kShardInf, fv = 0, cv = 0;
    for (uint32_t r = tid; r < d.world * kShardSub; r += 1024) {
        const int64_t* h = d.log + g.log_off + (int64_t)r * g.region_words;
        cv = max(cv, (unsigned long long)h[0]);
        sv = min(sv, (unsigned long long)h[1]);
        ev = min(ev, (unsigned long long)h[2]);
        fv = max(fv, (unsigned long long)h[3]);
    }
    if (sv < kShardInf) atomicMin(&s_solved, sv);
    if (ev < kShardInf) atomicMin(&s_err, ev);
    if (fv) atomicMax(&s_fail, fv);
    if (cv) atomicMax(&s_fill, cv);
    __syncthreads();
    const bool over = nodes_global + g_total >= (unsigned long long)max_nodes && need >= 1;
    if (over) {  // the parent inside tile s_tile whose inclusive count reaches the budget
        const uint32_t t = s_tile;
        for (uint32_t k = tid; k < (uint32_t)kScanTile; k += 1024) {
            const uint32_t p = t * kScanTile + k;
            if (p < g.n_par) {
                const uint32_t gm = d.pm[p].y;
                const unsigned long long ex = (unsigned long long)d.gblk[t] + (gm >> 12), in = ex + (unsigned long long)__popc(gm & 0xFFFu);
                if (ex < need && in >= need) s_pb = p;
            }
        }
    }
    __syncthreads();
    if (tid != 0) return;
    if (g.even >= 2048u) {  // (a small chunk says little about the fill of a large one)
        const unsigned long long q8 = (s_fill * 256ull + g.even - 1) / g.even;
        if (q8 > d.ctl[C_LEVEL_FILL]) d.ctl[C_LEVEL_FILL] = q8;
    }
    ChunkDec dec = {0, 0, 0, 0};
    if (s_fail) {  // some rank failed in an earlier chunk: everybody stops here (same chunk on every rank: the headers are the same)
        d.ctl[C_STATUS] = 4;
        d.ctl[C_FAIL_SEEN] = s_fail;
        *d.dec = dec;
        return;
    }
    uint32_t p_end = g.n_par - 1;
    bool budget_hit = false;
    unsigned long long commit_global = g_total, commit_local = l_total;
    if (need < 1) {  // only the very first parent can see this (budget <= 1)
        p_end = 0;
        budget_hit = true;
        commit_global = (unsigned long long)__popc(d.pm[0].y & 0xFFFu);
        commit_local = (unsigned long long)__popc(d.pm[0].x & 0xFFFu);
    } else if (over) {
        p_end = s_pb;
        budget_hit = true;
        const uint32_t t = p_end / kScanTile;
        const uint32_t gm = d.pm[p_end].y, lm = d.pm[p_end].x;
        commit_global = (unsigned long long)d.gblk[t] + (gm >> 12) + (unsigned long long)__popc(gm & 0xFFFu);
        commit_local = (unsigned long long)d.lblk[t] + (lm >> 12) + (unsigned long long)__popc(lm & 0xFFFu);
    }
    const unsigned long long end_pos = (unsigned long long)g.c0 + p_end;
    const unsigned long long stag = s_solved, eword = s_err;
    const bool is_solved = stag < kShardInf && stag / 12ull <= end_pos;
    d.ctl[C_CHUNKS] += 1;
    if (eword < kShardInf && (eword >> 8) / 12ull <= end_pos && !(is_solved && stag < (eword >> 8))) {
        d.ctl[C_STATUS] = 3;  // the reference executes this move before it stops: its ACMove raises
        *d.dec = dec;
        return;
    }
    if (is_solved) {
        const uint32_t q = (uint32_t)(stag / 12ull - (unsigned long long)g.c0), a = (uint32_t)(stag % 12ull);
        const uint32_t gq = d.pm[q].y;
        const unsigned long long before = (unsigned long long)d.gblk[q / kScanTile] + (gq >> 12) + (unsigned long long)__popc(gq & ((1u << a) - 1u));
        d.ctl[C_EXPANDED] += (unsigned long long)q + 1;
        d.ctl[C_NODES_GLOBAL] = nodes_global + before;  // new states with a smaller tag
        d.ctl[C_SOLVED_TAG] = stag;
        d.ctl[C_STATUS] = 1;
        *d.dec = dec;
        return;
    }
    const unsigned long long nodes = d.ctl[C_NODES];
    if (nodes + commit_local > (unsigned long long)d.cap_nodes) {
        // refused BEFORE anything is written; the other ranks learn it from the headers of the next chunk (and the closing
        // all-reduce of the orchestrator), so that everybody stops at the same chunk
        atomicMax(d.ctl + C_FAIL_LOCAL, (unsigned long long)FAIL_NODES);
    } else {
        dec.commit = 1;
        dec.cutoff = 12u * (p_end + 1);
        dec.node_base = (uint32_t)nodes;
        dec.gpos_base = (uint32_t)d.ctl[C_NEXT_COUNT];
        d.ctl[C_NODES] = nodes + commit_local;
    }
    d.ctl[C_NEXT_COUNT] += commit_global;
    d.ctl[C_NODES_GLOBAL] = nodes_global + commit_global;
    d.ctl[C_EXPANDED] += (unsigned long long)p_end + 1;
    if (budget_hit) d.ctl[C_STATUS] = 2;
    *d.dec = dec;
}

// Winners with a tag below the cutoff become local nodes: id = base + (winners of this rank with a smaller tag), which the
// per-parent masks give without a sort; gpos likewise from the all-reduced masks.  The stamp table is not touched.
template <typename W> __global__ void __launch_bounds__(256) k_shard_commit(ShardDev<W> d, ChunkGeo g) {
    ACX_VGPR_PAD_W(W, "v39", "v47");
    const ChunkDec dec = *d.dec;
    if (!dec.commit) return;
    const uint32_t r = blockIdx.y;
    const int64_t roff = g.log_off + (int64_t)r * g.region_words;
    const unsigned long long written = (unsigned long long)d.log[roff];
    const uint32_t cnt = written > g.subcap ? g.subcap : (uint32_t)written;
    // gridDim.x workgroups walk the region's records (a region is rarely full: a workgroup per 256 record SLOTS spent two
    // thirds of its launches on workgroups that found nothing to do).  (Four records per lane and step, to overlap their
    // chains of dependent loads, changed nothing: 139 us per 2^21-parent chunk either way.)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const int64_t* rec = d.log + roff + kShardHdr + (int64_t)i * recio<W>::RW;
        const unsigned long long x = (unsigned long long)rec[recio<W>::KW];
        const uint32_t tag = (uint32_t)(x >> 32);
        if (tag >= dec.cutoff) continue;
        const uint32_t par = tag / 12u, a = tag - 12u * par;
        const uint2 pmv = d.pm[par];
        const uint32_t lm = pmv.x;
        if (!((lm >> a) & 1u)) continue;
        const uint32_t below = (1u << a) - 1u, tile = par / kScanTile, gm = pmv.y;
        const uint32_t id = dec.node_base + d.lblk[tile] + (lm >> 12) + (uint32_t)__popc(lm & below);
        if (id >= d.cap_nodes) continue;  // k_shard_decide has refused such a commit already: never reached
        W k0, k1;
        recio<W>::get(rec, k0, k1);
        d.k0[id] = k0;
        d.k1[id] = k1;
        d.act[id] = (uint8_t)a;
        d.tlen[id] = (uint8_t)(keyops<W>::len(k0) + keyops<W>::len(k1));
        d.gpos[id] = dec.gpos_base + d.gblk[tile] + (gm >> 12) + (uint32_t)__popc(gm & below);
        d.pref[id] = (int64_t)(((unsigned long long)(r / kShardSub) << 40) | (x & 0xFFFFFFFFull));
        const OwnerParts o = owner_parts_of_key<W>(k0, k1);  // (a record only exists at world > 1, and most children never become one: acx_owner.h)
        d.cls[id] = make_uint2(o.c0, o.c1);
        d.inn[id] = (uint8_t)(o.in0 | (o.in1 << 4));
    }
}

// The born winners of a chunk become nodes: for a LOCAL parent every set bit of its mask is a child that was born here (a child of
// a local parent that another rank owns is flagged on that rank; a record this rank received has a parent on another rank), so the
// kernel walks the chunk's local parents, not records.  Numbering as in k_shard_commit (one sequence over both kinds of winner, in
// tag order); the keys are rebuilt from the parents (one move per new node, as k_bfs_compact does).  A workgroup takes 256
// consecutive local parents, lists their winners below the cutoff in LDS in tag order and hands them out one per lane, so that
// the node stores of a wave are consecutive ids (all of them at world 1, where every winner is born).
constexpr int kBornParents = 256;
template <typename W, int MODE> __global__ void __launch_bounds__(kBornParents) k_shard_commit_born(ShardDev<W> d, ChunkGeo g) {
    __shared__ W s_pk0[kBornParents];
    __shared__ W s_pk1[kBornParents];
    __shared__ uint32_t s_q[kBornParents], s_lm[kBornParents], s_idb[kBornParents], s_gpb[kBornParents], s_gm[kBornParents];
    __shared__ uint16_t s_list[kBornParents * 12];
    __shared__ uint32_t s_wsum[kBornParents / 64];
    __shared__ uint2 s_cls[kBornParents];
    __shared__ uint8_t s_inn[kBornParents];
    ACX_VGPR_PAD_W(W, "v47", "v63");
    const ChunkDec dec = *d.dec;
    if (!dec.commit) return;
    const uint32_t b0 = d.bounds[2 * g.par], b1 = d.bounds[2 * g.par + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t first = b0 + blockIdx.x * (uint32_t)kBornParents;
    if (first >= b1) return;
    const uint32_t i = first + tid;
    uint32_t lm = 0;
    if (i < b1) {
        const uint32_t q = d.gpos[i] - (uint32_t)g.c0, tile = q / kScanTile;
        const uint2 pmv = d.pm[q];
        const uint32_t lw = pmv.x, gw = pmv.y, full = lw & 0xFFFu;
        const uint32_t t0 = 12u * q;  // bits a with t0 + a < cutoff
        lm = t0 + 12u <= dec.cutoff ? full : (t0 >= dec.cutoff ? 0u : full & ((1u << (dec.cutoff - t0)) - 1u));
        s_q[tid] = q;
        s_lm[tid] = full;
        s_gm[tid] = gw & 0xFFFu;
        s_idb[tid] = dec.node_base + d.lblk[tile] + (lw >> 12);
        s_gpb[tid] = dec.gpos_base + d.gblk[tile] + (gw >> 12);
        s_pk0[tid] = d.k0[i];
        s_pk1[tid] = d.k1[i];
        if (d.cls) {
            s_cls[tid] = d.cls[i];
            s_inn[tid] = d.inn[i];
        }
    }
    const uint32_t cnt = (uint32_t)__popc(lm);
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t off = incl - cnt, total = 0;
    for (uint32_t w2 = 0; w2 < (uint32_t)kBornParents / 64; w2++) {
        if (w2 < wave) off += s_wsum[w2];
        total += s_wsum[w2];
    }
    for (uint32_t f = lm; f; f &= f - 1) s_list[off++] = (uint16_t)((tid << 4) | (uint32_t)__builtin_ctz(f));
    __syncthreads();
    for (uint32_t j = tid; j < total; j += (uint32_t)kBornParents) {
        const uint32_t e = s_list[j], pl = e >> 4, a = e & 15u, below = (1u << a) - 1u;
        const uint32_t id = s_idb[pl] + (uint32_t)__popc(s_lm[pl] & below);
        if (id >= d.cap_nodes) continue;  // k_shard_decide has refused such a commit already: never reached
        Pres<W> s;
        key_to_pres<W>(s_pk0[pl], s_pk1[pl], s);
        (void)search_move<W, MODE>(s, (int)a, d.L, d.cyclical != 0);
        d.k0[id] = keyops<W>::make(s.w0, s.n0);
        d.k1[id] = keyops<W>::make(s.w1, s.n1);
        d.act[id] = (uint8_t)a;
        d.tlen[id] = (uint8_t)(s.n0 + s.n1);
        d.gpos[id] = s_gpb[pl] + (uint32_t)__popc(s_gm[pl] & below);
        d.pref[id] = (int64_t)(((unsigned long long)d.rank << 40) | (unsigned long long)(first + pl));
        if (d.cls) {  // world > 1: what names the node's owner -- its parent's parts, except for what the move rewrote (as k_shard_expand)
            const OwnerParts o = MODE == kMoveGeneral ? owner_parts_of_pres<W>(s) : owner_parts_of_child<W>(OwnerParts{s_cls[pl].x, s_cls[pl].y, s_inn[pl] & 15u, (uint32_t)s_inn[pl] >> 4}, a, s);
            d.cls[id] = make_uint2(o.c0, o.c1);
            d.inn[id] = (uint8_t)(o.in0 | (o.in1 << 4));
        }
    }
}

// root: local node 0 of its owner (global position 0 of level 0); its record sits at the start of the log
template <typename W> __global__ void k_shard_seed(ShardDev<W> d, W k0, W k1) {
    ACX_VGPR_PAD("v23");
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.act[0] = 0xff;
    d.tlen[0] = (uint8_t)(keyops<W>::len(k0) + keyops<W>::len(k1));
    d.gpos[0] = 0;
    d.pref[0] = -1;
    if (d.cls) {
        const OwnerParts o = owner_parts_of_key<W>(k0, k1);
        d.cls[0] = make_uint2(o.c0, o.c1);
        d.inn[0] = (uint8_t)(o.in0 | (o.in1 << 4));
    }
    recio<W>::put(d.log, k0, k1);
    d.log[recio<W>::KW] = 0;
    const uint64_t hk = shard_hash(k0, k1);
    d.stab[(uint32_t)hk & d.stmask & ~3u] = stamp_top(hk, d.epoch);  // a record stamp with offset 0: first slot of its bucket
    d.ctl[C_NODES] = 1;
}

// test hook (acx_shard_check_owners): every local node must live on the rank that owns its key, and the class hashes it carries
// (inherited along conjugations) must be the ones its key gives
template <typename W> __global__ void __launch_bounds__(256) k_shard_check_owners(ShardDev<W> d, uint32_t own_from, unsigned long long* __restrict__ bad) {
    ACX_VGPR_PAD_W(W, "v63", "v95");
    const uint32_t n = (uint32_t)d.ctl[C_NODES];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const OwnerParts o = owner_parts_of_key<W>(d.k0[i], d.k1[i]);
        // (the nodes of the replicated levels, below own_from, live on EVERY rank: only what they carry is checked)
        const bool ok = (i < own_from || owner_of_sum(o.sum(), d.world) == d.rank) && (!d.cls || (d.cls[i].x == o.c0 && d.cls[i].y == o.c1 && d.inn[i] == (uint8_t)(o.in0 | (o.in1 << 4))));
        if (!ok) atomicAdd(bad, 1ull);
    }
}


// ---- replicated small levels -> owner-partitioned frontier (round 6) -----------------------------------------------------------
// While the levels are small every rank processes the WHOLE frontier with the world-1 kernels (no exchange, no collective: a level
// of a few thousand parents costs a rank a few microseconds of expansion against >= 100 us of collectives): all ranks hold the same
// nodes, in the same order, with every state stamped in their own table.  At the first large level the frontier is PARTITIONED: of
// the newest level's nodes [base, base + n) a rank keeps a COPY of those it owns (acx_owner.h: from the class hashes and inner letters
// every node carries), appended to its arena in the same (gpos) order -- its slice of the running level from then on.  The copy's
// parent reference is the original's (a replicated node, named with this rank's own number: every rank holds it at the same id), so a
// path walks from the sharded part into the replicated prefix without another exchange.  The table needs nothing: every state of the
// replicated levels already has its stamp on every rank.
constexpr int kPartTile = 1024;
template <typename W> __device__ __forceinline__ bool part_owned(const ShardDev<W>& d, uint32_t i) {
    const uint2 c = d.cls[i];
    const uint32_t in = d.inn[i];
    return owner_of_sum(owner_sum(c.x, c.y, in & 15u, in >> 4), d.world) == d.rank;
}
template <typename W> __global__ void __launch_bounds__(kPartTile) k_shard_part_count(ShardDev<W> d, uint32_t base, uint32_t n, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_c;
    ACX_VGPR_PAD("v31");
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * kPartTile + threadIdx.x;
    const bool own = i < n && part_owned<W>(d, base + i);
    const unsigned long long b = __ballot(own);
    if ((threadIdx.x & 63u) == 0 && b) atomicAdd(&s_c, (uint32_t)__popcll(b));
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_c;
}
// one workgroup: the tiles' counts -> exclusive prefixes in place; the level switch that the next chunk's k_shard_prep performs then
// finds the copies [old nodes, old nodes + total) as the new level
template <typename W> __global__ void __launch_bounds__(1024) k_shard_part_scan(ShardDev<W> d, uint32_t tiles, uint32_t* __restrict__ cnt, uint32_t nodes_old) {
    __shared__ uint32_t s_p[1024];
    ACX_VGPR_PAD("v31");
    const uint32_t tid = threadIdx.x, per = (tiles + 1023u) / 1024u;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t t = tid * per + k;
        if (t < tiles) sum += cnt[t];
    }
    s_p[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t a = tid >= o ? s_p[tid - o] : 0u;
        __syncthreads();
        s_p[tid] += a;
        __syncthreads();
    }
    uint32_t ex = s_p[tid] - sum;
    const uint32_t total = s_p[1023];
    const bool fits = (unsigned long long)nodes_old + total <= (unsigned long long)d.cap_nodes;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t t = tid * per + k;
        if (t < tiles) {
            const uint32_t c = cnt[t];
            cnt[t] = fits ? ex : 0xFFFFFFFFu;  // (no room: nothing is copied, the search fails with FAIL_NODES on every path that looks)
            ex += c;
        }
    }
    if (tid == 0) {
        d.ctl[C_LVL_HI] = nodes_old;
        if (fits) d.ctl[C_NODES] = (unsigned long long)nodes_old + total;
        else atomicMax(d.ctl + C_FAIL_LOCAL, (unsigned long long)FAIL_NODES);
    }
}
template <typename W> __global__ void __launch_bounds__(kPartTile) k_shard_part_write(ShardDev<W> d, uint32_t base, uint32_t n, const uint32_t* __restrict__ cnt, uint32_t nodes_old) {
    __shared__ uint32_t s_w[kPartTile / 64];
    ACX_VGPR_PAD_W(W, "v31", "v39");
    const uint32_t off = cnt[blockIdx.x];
    if (off == 0xFFFFFFFFu) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i = blockIdx.x * kPartTile + tid;
    const bool own = i < n && part_owned<W>(d, base + i);
    const unsigned long long b = __ballot(own);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    for (uint32_t w2 = 0; w2 < wave; w2++) before += s_w[w2];
    if (!own) return;
    const uint32_t src = base + i, dst = nodes_old + off + before;
    d.k0[dst] = d.k0[src];
    d.k1[dst] = d.k1[src];
    d.pref[dst] = d.pref[src];
    d.gpos[dst] = d.gpos[src];
    d.act[dst] = d.act[src];
    d.tlen[dst] = d.tlen[src];
    d.cls[dst] = d.cls[src];
    d.inn[dst] = d.inn[src];
}

// The path of local node `id` towards the root for as long as the parents are local (one lane: a chain of dependent loads, run once per
// search): out[0] = the first parent reference that is NOT local (-1: the root was reached), out[1] = n, then n pairs (action, total
// length), node `id` first; the root's action is -1.
template <typename W> __global__ void k_shard_walk(ShardDev<W> d, uint32_t id, uint32_t cap, int64_t* __restrict__ out) {
    ACX_VGPR_PAD("v23");
    uint32_t cur = id, n = 0;
    int64_t next = -1;
    for (;;) {
        const int64_t pr = d.pref[cur];
        out[2 + 2 * n] = pr < 0 ? -1 : (int64_t)d.act[cur];
        out[3 + 2 * n] = (int64_t)d.tlen[cur];
        n++;
        next = pr;
        if (pr < 0 || (uint32_t)((unsigned long long)pr >> 40) != d.rank || n == cap) break;
        cur = (uint32_t)((unsigned long long)pr & 0xFFFFFFFFull);
    }
    out[0] = next;
    out[1] = (int64_t)n;
}

template <typename W> __global__ void k_shard_find(ShardDev<W> d, uint32_t gpos, int64_t* __restrict__ out) {
    ACX_VGPR_PAD("v23");
    const uint32_t lo = (uint32_t)d.ctl[C_LVL_LO], hi = (uint32_t)d.ctl[C_LVL_HI];
    const uint32_t k = lower_gpos(d.gpos, lo, hi, gpos);
    *out = (k < hi && d.gpos[k] == gpos) ? (int64_t)k : -1;
}

// the engine's failure code, set from the host (an exception on the orchestrator's side of this rank)
__global__ void k_shard_ctl_init(unsigned long long* ctl) {
    ACX_VGPR_PAD("v15");
    ctl[C_MIN_LEN] = kShardInf;
    ctl[C_NODES_GLOBAL] = 1;
}

__global__ void k_shard_fail(unsigned long long* ctl, unsigned long long code) {
    ACX_VGPR_PAD("v15");
    atomicMax(ctl + C_FAIL_LOCAL, code);
}

constexpr int kCtlSlots = 4;

// The pinned snapshot slots + their events of an engine.  hipHostMalloc / hipHostFree synchronise with the device and cost
// ~0.15 ms each, which a 13 ms search notices: released sets are kept (per device) and handed to the next engine.
struct CtlHost {
    unsigned long long* pinned = nullptr;  // kCtlSlots x C_WORDS
    hipEvent_t ev[kCtlSlots] = {};
    hipEvent_t ready = nullptr;  // behind the fills of the engine's set-up (null stream)
    int dev = -1;
};
static std::mutex g_ctl_mutex;
static std::vector<CtlHost> g_ctl_free;
static int ctl_host_take(CtlHost& c) {
    int dev = 0;
    ACX_HIP_TRY(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(g_ctl_mutex);
        for (size_t i = 0; i < g_ctl_free.size(); i++)
            if (g_ctl_free[i].dev == dev) {
                c = g_ctl_free[i];
                g_ctl_free.erase(g_ctl_free.begin() + (long)i);
                return ACX_OK;
            }
    }
    c.dev = dev;
    ACX_HIP_TRY(hipHostMalloc((void**)&c.pinned, (size_t)kCtlSlots * C_WORDS * 8, hipHostMallocDefault));
    for (auto& e : c.ev) ACX_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ACX_HIP_TRY(hipEventCreateWithFlags(&c.ready, hipEventDisableTiming));
    return ACX_OK;
}
static void ctl_host_give(CtlHost& c) {
    if (!c.pinned) return;
    std::lock_guard<std::mutex> lock(g_ctl_mutex);
    g_ctl_free.push_back(c);
    c = CtlHost();
}

template <typename W> struct ShardEngine {
    ShardDev<W> d;
    StampBuf tab_buf;  // the visited table: from the pool with a new epoch, no refill (acx_frontier.h)
    DevBuf nodes_buf, chunk_buf, scal_buf;
    int64_t* d_find = nullptr;
    int64_t* d_send = nullptr;  // the caller's send buffer (null when world == 1: the chunk is expanded straight into the log)
    uint64_t cap_nodes = 0, n_slots = 0, chunk_parents = 0, log_words = 0, send_words = 0;
    int64_t log_off = 0;        // next free word of the log
    // chunks between acx_shard_chunk_expand and acx_shard_chunk_commit, oldest first: the orchestrator expands chunk k + 1 (on a side
    // stream) before it inserts and commits chunk k
    static constexpr int kGeoRing = 4;
    ChunkGeo geos[kGeoRing] = {};
    int geo_head = 0, geo_count = 0;   // ring of open chunks
    uint32_t chunk_seq = 0;            // chunks expanded so far: its low bit is the chunk's parity (flag set, bounds pair)
    int geo_inserted = 0;              // how many of them have been through acx_shard_chunk_insert
    uint64_t nodes_host = 0, lvl_lo_host = 0, lvl_hi_host = 0;  // what the last control-block snapshot said
    int rank = 0, world = 1;
    // k_shard_insert and k_shard_commit run as a bounded number of workgroups that walk their region's records (0 = a
    // workgroup per 256 record slots).  Regions are sized for the worst case and are usually less than half full, so two
    // thirds of the per-slot workgroups found nothing to do, and the dedup gains nothing from more than ~4 workgroups per
    // compute unit in flight: the memory-side atomic units are saturated by then and a deeper queue is only more latency.
    // Measured at 1e8 nodes (round 4, a sweep over the workgroup counts): insert 545 -> 528 us per 2^21-parent chunk, commit 158 -> ~125.
    unsigned insert_wgs = 0, commit_wgs = 0;
    int move_mode = kMoveGeneral;  // acx_bfs.h: set from the root (acx_shard_root_record, which every rank calls)
    CtlHost host;  // pinned snapshot slots + events
    // Round 6: while `replicated` is set the engine processes whole levels with the world-1 kernels (every rank the same nodes, no
    // exchange); acx_shard_partition ends that phase.  The nodes below own_from are the replicated ones.
    bool replicated = false;
    uint64_t own_from = 0;
    uint32_t* part_cnt = nullptr;  // per tile of kPartTile nodes: owned nodes (k_shard_part_*)
    int64_t* d_walk = nullptr;     // k_shard_walk's output
    static constexpr int kWalkCap = 1024;
    int world_eff() const { return replicated ? 1 : world; }
    ShardDev<W> dev() const {
        ShardDev<W> x = d;
        if (replicated) x.world = 1;  // (the rank stays: parent references name this rank)
        return x;
    }

    ~ShardEngine() {
        for (auto& e : host.ev)
            if (e) (void)hipEventSynchronize(e);  // (a snapshot still in flight would land in the next owner's slots)
        if (host.ready) (void)hipEventSynchronize(host.ready);
        ctl_host_give(host);
    }

    // The fills of the set-up (2 GB of table for a 1e8-node search: 0.35 ms) run on the null stream and the host does not wait for
    // them; a caller's other stream waits for them once, in front of its first engine call.
    std::vector<hipStream_t> ready_streams;
    int await_ready(hipStream_t st) {
        if (!st) return ACX_OK;  // the null stream itself
        for (hipStream_t s : ready_streams)
            if (s == st) return ACX_OK;
        ACX_HIP_TRY(hipStreamWaitEvent(st, host.ready, 0));
        ready_streams.push_back(st);
        return ACX_OK;
    }

    int init(int L, int cyclical, int64_t node_cap, int64_t chunk_parents_, int rank_, int world_) {
        memset(&d, 0, sizeof(d));
        d.L = L;
        d.cyclical = cyclical;
        d.world = (uint32_t)world_;
        d.rank = (uint32_t)rank_;
        rank = rank_;
        world = world_;
        {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            insert_wgs = 4u * (unsigned)cus;
            commit_wgs = 8u * (unsigned)cus;
        }
        cap_nodes = (uint64_t)node_cap + 64;
        chunk_parents = (uint64_t)std::max<int64_t>(chunk_parents_, 1);
        int64_t subcap, region_words;
        shard_layout((int64_t)chunk_parents, world, recio<W>::RW, 0, &subcap, &region_words);
        // what a chunk can put into the table beyond the nodes it commits: the records it receives + the children born here, of
        // the chunk being dedup'ed and of the one whose expansion runs ahead of it
        // (a rank's share of a chunk's parents: the owner function keeps families of states together, so a chunk of consecutive frontier
        // positions is shared out less evenly than the nodes as a whole -- 1.1-1.4 x the even share at 8 ranks, tools/owner_balance.cpp)
        const uint64_t local_parents = world == 1 ? chunk_parents : std::min<uint64_t>(chunk_parents, 2 * chunk_parents / (uint64_t)world);
        const uint64_t chunk_records = (uint64_t)subcap * kShardSub * (uint64_t)world + 2 * (12 * local_parents + kExpandTile);
        n_slots = 1024;
        while (n_slots < 2 * (cap_nodes + chunk_records)) n_slots <<= 1;
        if (n_slots > (1ull << 31) || cap_nodes > (1ull << 31)) return fail(ACX_E_INVAL, "acx_shard: capacity too large for 32-bit node ids");
        size_t o = 0;
        auto take = [&](uint8_t* base, size_t bytes) {
            uint8_t* p = base ? base + o : nullptr;
            o += (bytes + 255) / 256 * 256;
            return p;
        };
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)nodes_buf.p;
            o = 0;
            d.k0 = (W*)take(b, cap_nodes * sizeof(W));
            d.k1 = (W*)take(b, cap_nodes * sizeof(W));
            d.pref = (int64_t*)take(b, cap_nodes * 8);
            d.gpos = (uint32_t*)take(b, cap_nodes * 4);
            d.act = (uint8_t*)take(b, cap_nodes);
            d.tlen = (uint8_t*)take(b, cap_nodes);
            d.cls = world > 1 ? (uint2*)take(b, cap_nodes * 8) : nullptr;
            d.inn = world > 1 ? (uint8_t*)take(b, cap_nodes) : nullptr;
            if (pass == 0 && nodes_buf.alloc(o)) return ACX_E_NOMEM;
        }
        const uint64_t n_tiles = (chunk_parents + kScanTile - 1) / kScanTile + 1;
        size_t flag_bytes = 0;
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)chunk_buf.p;
            o = 0;
            const size_t one_set = (chunk_parents + kScanTile + 63) / 64 * 64;  // words (+ a tile of k_shard_pack)
            d.tk = (uint32_t*)take(b, 2 * one_set * 4);  // two sets, by chunk parity
            d.rp = (uint32_t*)take(b, 2 * one_set * 4);
            d.flag_stride = one_set;
            flag_bytes = o;
            d.lmp = (uint32_t*)take(b, 4 * chunk_parents + 64);
            d.pm = (uint2*)take(b, 8 * chunk_parents + 64);
            d.lblk = (uint32_t*)take(b, 4 * n_tiles);
            d.gblk = (uint32_t*)take(b, 4 * n_tiles);
            part_cnt = (uint32_t*)take(b, 4 * (cap_nodes / kPartTile + 2));
            d_walk = (int64_t*)take(b, 8 * (2 + 2 * kWalkCap));
            if (pass == 0 && chunk_buf.alloc(o)) return ACX_E_NOMEM;
        }
        if (int rc = tab_buf.alloc(n_slots * 8, nullptr)) return rc;  // (a fill, when there is one, is queued on the null stream like the others)
        d.stab = (unsigned long long*)tab_buf.p;
        d.epoch = tab_buf.epoch;
        d.stmask = (uint32_t)(n_slots - 1);
        if (scal_buf.alloc(1024)) return ACX_E_NOMEM;
        uint8_t* sc = (uint8_t*)scal_buf.p;
        d.ctl = (unsigned long long*)sc;              // C_WORDS x 8 = 128 bytes
        d.bounds = (uint32_t*)(sc + 256);
        d.dec = (ChunkDec*)(sc + 320);
        d_find = (int64_t*)(sc + 384);
        d.cap_nodes = (uint32_t)cap_nodes;
        if (int rc = ctl_host_take(host)) return rc;
        ACX_HIP_TRY(hipMemsetAsync(chunk_buf.p, 0, flag_bytes, nullptr));
        ACX_HIP_TRY(hipMemsetAsync(scal_buf.p, 0, 1024, nullptr));
        hipLaunchKernelGGL(k_shard_ctl_init, dim3(1), dim3(1), 0, nullptr, d.ctl);  // smallest length = none yet; len(tree_nodes) counts the root, on every rank
        ACX_HIP_TRY(hipGetLastError());
        ACX_HIP_TRY(hipEventRecord(host.ready, nullptr));  // (every entry point: await_ready)
        return ACX_OK;
    }
};

struct ShardAny {
    int width;  // 0: uint64_t keys (max_relator_length <= 29), 1: unsigned __int128 (<= 61), 2: u128x (<= 64, acx_keys.h)
    ShardEngine<uint64_t>* e64 = nullptr;
    ShardEngine<u128>* e128 = nullptr;
    ShardEngine<u128x>* e128x = nullptr;
};

#define ACX_SHARD_DISPATCH(h, ...)                   \
    do {                                             \
        if ((h)->width == 2) {                       \
            typedef u128x W;                         \
            auto& E = *(h)->e128x;                   \
            (void)sizeof(W);                         \
            __VA_ARGS__;                             \
        } else if ((h)->width == 1) {                \
            typedef u128 W;                          \
            auto& E = *(h)->e128;                    \
            (void)sizeof(W);                         \
            __VA_ARGS__;                             \
        } else {                                     \
            typedef uint64_t W;                      \
            auto& E = *(h)->e64;                     \
            (void)sizeof(W);                         \
            __VA_ARGS__;                             \
        }                                            \
    } while (0)

template <typename W> static int shard_root(ShardEngine<W>& E, const int8_t* pres, int64_t* rec) {
    Pres<W> root;
    bool ok = pack_relator<W>(pres, E.d.L, root.w0, root.n0);
    ok = pack_relator<W>(pres + E.d.L, E.d.L, root.w1, root.n1) && ok;
    if (!ok) return fail(ACX_E_ROWERR, "acx_shard: the presentation is not a zero-padded word pair over {+-1,+-2}");
    if (is_long_key<W>::value && (has_inverse_pair<W, true>(root.w0, root.n0) || has_inverse_pair<W, true>(root.w1, root.n1)))
        return fail(ACX_E_INVAL, "acx_shard: at max_relator_length %d (> 61) the presentation must be freely reduced (acx_keys.h)", (int)E.d.L);
    // a root in normal form keeps the whole search in normal form (acx_bfs.h)
    E.move_mode = !is_normal_form<W>(root, E.d.cyclical != 0) ? kMoveGeneral : (E.d.cyclical ? kMoveNfCyclical : kMoveNf);
    recio<W>::put(rec, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1));
    rec[recio<W>::KW] = 0;
    rec[recio<W>::KW + 1] = -1;
    return ACX_OK;
}

template <typename W> static int shard_attach(ShardEngine<W>& E, int64_t* log, int64_t log_words, int64_t* send, int64_t send_words, int32_t* gmask) {
    if (!log || !gmask || log_words < 64 || (E.world > 1 && !send)) return fail(ACX_E_INVAL, "acx_shard_attach: bad argument");
    if ((uint64_t)log_words > (1ull << 36)) return fail(ACX_E_INVAL, "acx_shard_attach: the log is limited to 2^36 words");
    if (!E.d.log) E.log_off = 8;  // first attach; words 0 .. RW - 1: the root's record.  (A later attach hands over a LARGER log with the old content in front.)
    else if ((uint64_t)log_words < E.log_words) return fail(ACX_E_INVAL, "acx_shard_attach: the log can only grow");
    E.d.log = log;
    E.log_words = (uint64_t)log_words;
    E.d_send = E.world > 1 ? send : nullptr;
    E.send_words = (uint64_t)send_words;
    E.d.gmask = gmask;
    return ACX_OK;
}

template <typename W> static int shard_seed(ShardEngine<W>& E, const int64_t* rec, hipStream_t st) {
    if (int rc = E.await_ready(st)) return rc;
    if (!E.d.log) return fail(ACX_E_INVAL, "acx_shard_seed: call acx_shard_attach first");
    if (E.nodes_host) return fail(ACX_E_INVAL, "acx_shard_seed: the engine already holds nodes");
    if (rec) {
        W k0, k1;
        recio<W>::get(rec, k0, k1);
        hipLaunchKernelGGL(k_shard_seed<W>, dim3(1), dim3(1), 0, st, E.d, k0, k1);
        ACX_HIP_TRY(hipGetLastError());
        E.nodes_host = 1;
    }
    return ACX_OK;
}

template <typename W>
static int shard_chunk_expand(ShardEngine<W>& E, int64_t c0, int64_t c1, int level_first, int fill_q8, int64_t* recv_off, int64_t* words, hipStream_t st) {
    if (int rc = E.await_ready(st)) return rc;
    if (!E.d.log) return fail(ACX_E_INVAL, "acx_shard_chunk_expand: call acx_shard_attach first");
    const int64_t n_par = c1 - c0;
    if (n_par < 1 || (uint64_t)n_par > E.chunk_parents) return fail(ACX_E_CAPACITY, "acx_shard_chunk_expand: a chunk of %lld parents exceeds the engine's %llu", (long long)n_par, (unsigned long long)E.chunk_parents);
    if (c1 > (int64_t)0xFFFFFFFFll) return fail(ACX_E_INVAL, "acx_shard_chunk_expand: a level is limited to 2^32 - 1 positions");
    int64_t subcap, region_words, even;
    shard_layout(n_par, E.world_eff(), recio<W>::RW, fill_q8, &subcap, &region_words, &even);
    const int64_t total = region_words * kShardSub * E.world_eff();
    if (E.geo_count == ShardEngine<W>::kGeoRing) return fail(ACX_E_INVAL, "acx_shard_chunk_expand: too many chunks in flight (commit the oldest first)");
    ChunkGeo geo{};
    geo.c0 = c0;
    geo.n_par = (uint32_t)n_par;
    geo.subcap = (uint32_t)subcap;
    geo.region_words = (uint32_t)region_words;
    geo.even = (uint32_t)even;
    geo.par = E.chunk_seq++ & 1u;
    geo.log_off = E.log_off;
    int64_t* send = (E.d_send && !E.replicated) ? E.d_send : E.d.log + E.log_off;  // (replicated levels: as world 1, the regions are their headers, in the log)
    if ((uint64_t)(E.log_off + total) > E.log_words || (E.d_send && !E.replicated && (uint64_t)total > E.send_words))
        return fail(ACX_E_CAPACITY, "acx_shard_chunk_expand: the record log (%llu words, %lld used) cannot take a chunk of %lld words: attach a larger one",
                    (unsigned long long)E.log_words, (long long)E.log_off, (long long)total);
    if (level_first) {  // the nodes committed since the previous switch are this rank's slice of the new level (host mirror; the device switches in k_shard_prep)
        E.lvl_lo_host = E.lvl_hi_host;
        E.lvl_hi_host = E.nodes_host;
    }
    *recv_off = E.log_off;
    *words = total;
    E.log_off += total;
    E.geos[(E.geo_head + E.geo_count) % ShardEngine<W>::kGeoRing] = geo;
    E.geo_count++;
    hipLaunchKernelGGL(k_shard_prep<W>, dim3(1), dim3(256), 0, st, E.dev(), geo, level_first, send);
    const int64_t np_max = std::min<int64_t>(n_par, (int64_t)(E.lvl_hi_host - E.lvl_lo_host));
    if (np_max > 0) {
        const dim3 grid((unsigned)((np_max + kExpandParents - 1) / kExpandParents));
        const size_t lds = (size_t)E.world_eff() * kExpandParents * 4;  // s_bits
        if (E.world_eff() == 1) {
            if (E.move_mode == kMoveNf) hipLaunchKernelGGL((k_shard_expand<W, kMoveNf, true>), grid, dim3(kExpandThreads), lds, st, E.dev(), geo, send);
            else if (E.move_mode == kMoveNfCyclical) hipLaunchKernelGGL((k_shard_expand<W, kMoveNfCyclical, true>), grid, dim3(kExpandThreads), lds, st, E.dev(), geo, send);
            else hipLaunchKernelGGL((k_shard_expand<W, kMoveGeneral, true>), grid, dim3(kExpandThreads), lds, st, E.dev(), geo, send);
        } else if (E.move_mode == kMoveNf) hipLaunchKernelGGL((k_shard_expand<W, kMoveNf, false>), grid, dim3(kExpandThreads), lds, st, E.dev(), geo, send);
        else if (E.move_mode == kMoveNfCyclical) hipLaunchKernelGGL((k_shard_expand<W, kMoveNfCyclical, false>), grid, dim3(kExpandThreads), lds, st, E.dev(), geo, send);
        else hipLaunchKernelGGL((k_shard_expand<W, kMoveGeneral, false>), grid, dim3(kExpandThreads), lds, st, E.dev(), geo, send);
    }
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

// `dedup` false (acx_shard_chunk_insert_dead): the masks only -- the chunk's records are dropped, the chunk stays in the ring
template <typename W> static int shard_chunk_insert(ShardEngine<W>& E, hipStream_t st, bool dedup = true) {
    if (int rc = E.await_ready(st)) return rc;
    if (E.geo_inserted >= E.geo_count) return fail(ACX_E_INVAL, "acx_shard_chunk_insert: no expanded chunk is waiting");
    const ChunkGeo& geo = E.geos[(E.geo_head + E.geo_inserted) % ShardEngine<W>::kGeoRing];
    E.geo_inserted++;
    const unsigned tiles = (unsigned)((geo.subcap + 255) / 256), regions = (unsigned)(kShardSub * E.world_eff());
    unsigned gx = tiles;
    if (E.insert_wgs) gx = std::min(tiles, std::max(1u, E.insert_wgs / regions));
    if (dedup && tiles) {  // (no tiles: world 1 with born stamps -- no record ever arrives)
        if (E.move_mode == kMoveNf) hipLaunchKernelGGL((k_shard_insert<W, kMoveNf>), dim3(gx, regions), dim3(256), 0, st, E.dev(), geo, tiles);
        else if (E.move_mode == kMoveNfCyclical) hipLaunchKernelGGL((k_shard_insert<W, kMoveNfCyclical>), dim3(gx, regions), dim3(256), 0, st, E.dev(), geo, tiles);
        else hipLaunchKernelGGL((k_shard_insert<W, kMoveGeneral>), dim3(gx, regions), dim3(256), 0, st, E.dev(), geo, tiles);
    }
    hipLaunchKernelGGL(k_shard_pack<W>, dim3((geo.n_par + kScanTile - 1) / kScanTile), dim3(1024), 0, st, E.dev(), geo.n_par, geo.par);
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W> static int shard_chunk_commit(ShardEngine<W>& E, int64_t max_nodes, hipStream_t st) {
    if (int rc = E.await_ready(st)) return rc;
    if (E.geo_inserted < 1) return fail(ACX_E_INVAL, "acx_shard_chunk_commit: no inserted chunk is waiting");
    const ChunkGeo geo = E.geos[E.geo_head];
    E.geo_head = (E.geo_head + 1) % ShardEngine<W>::kGeoRing;
    E.geo_count--;
    E.geo_inserted--;
    hipLaunchKernelGGL(k_shard_scan<W>, dim3((geo.n_par + kScanTile - 1) / kScanTile), dim3(1024), 0, st, E.dev(), geo.n_par);
    hipLaunchKernelGGL(k_shard_decide<W>, dim3(1), dim3(1024), 0, st, E.dev(), geo, max_nodes);
    const unsigned tiles = (unsigned)((geo.subcap + 255) / 256), regions = (unsigned)(kShardSub * E.world_eff());
    unsigned gx = tiles;
    if (E.commit_wgs) gx = std::min(tiles, std::max(1u, E.commit_wgs / regions));
    if (tiles) hipLaunchKernelGGL(k_shard_commit<W>, dim3(gx, regions), dim3(256), 0, st, E.dev(), geo);
    const int64_t np_max = std::min<int64_t>((int64_t)geo.n_par, (int64_t)(E.lvl_hi_host - E.lvl_lo_host));  // this rank's share of the chunk's parents, at most
    if (np_max > 0) {
        const dim3 grid((unsigned)((np_max + kBornParents - 1) / kBornParents));
        if (E.move_mode == kMoveNf) hipLaunchKernelGGL((k_shard_commit_born<W, kMoveNf>), grid, dim3(kBornParents), 0, st, E.dev(), geo);
        else if (E.move_mode == kMoveNfCyclical) hipLaunchKernelGGL((k_shard_commit_born<W, kMoveNfCyclical>), grid, dim3(kBornParents), 0, st, E.dev(), geo);
        else hipLaunchKernelGGL((k_shard_commit_born<W, kMoveGeneral>), grid, dim3(kBornParents), 0, st, E.dev(), geo);
    }
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W> static int shard_ctl_snapshot(ShardEngine<W>& E, int slot, hipStream_t st) {
    if (int rc = E.await_ready(st)) return rc;
    ACX_HIP_TRY(hipMemcpyAsync(E.host.pinned + (size_t)slot * C_WORDS, E.d.ctl, C_WORDS * 8, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipEventRecord(E.host.ev[slot], st));
    return ACX_OK;
}

template <typename W> static int shard_ctl_wait(ShardEngine<W>& E, int slot, int64_t* out) {
    ACX_HIP_TRY(hipEventSynchronize(E.host.ev[slot]));
    const unsigned long long* p = E.host.pinned + (size_t)slot * C_WORDS;
    for (int k = 0; k < C_WORDS; k++) out[k] = (int64_t)p[k];
    E.nodes_host = p[C_NODES];  // (the level bounds of the host mirror follow at the next level switch)
    return ACX_OK;
}

template <typename W> static int shard_find(ShardEngine<W>& E, int64_t gpos, int64_t* id, hipStream_t st) {
    if (int rc = E.await_ready(st)) return rc;
    *id = -1;
    if (gpos < 0 || gpos > (int64_t)0xFFFFFFFFll) return ACX_OK;
    hipLaunchKernelGGL(k_shard_find<W>, dim3(1), dim3(1), 0, st, E.d, (uint32_t)gpos, E.d_find);
    ACX_HIP_TRY(hipMemcpyAsync(id, E.d_find, 8, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    return ACX_OK;
}


// end of the replicated phase: this rank's share of the newest level becomes its frontier slice (k_shard_part_*).  Called between two
// levels, by every rank, after the level's last control block has been read (acx_shard_ctl_wait): the host mirror then knows the
// nodes [lvl_hi_host, nodes_host) of the newest level.  Synchronises once (the number of copies comes back).
template <typename W> static int shard_partition(ShardEngine<W>& E, hipStream_t st) {
    if (int rc = E.await_ready(st)) return rc;
    if (!E.replicated) return fail(ACX_E_INVAL, "acx_shard_partition: the engine is not in its replicated phase");
    if (E.geo_count) return fail(ACX_E_INVAL, "acx_shard_partition: chunks are in flight (commit them first)");
    E.replicated = false;
    const uint64_t nodes_old = E.nodes_host, base = E.lvl_hi_host, n = nodes_old - base;
    E.own_from = nodes_old;
    E.lvl_hi_host = nodes_old;  // the next chunk's level switch: [nodes_old, nodes_old + copies)
    if (E.world == 1) return ACX_OK;  // (a one-rank engine owns everything: the level is its slice as it stands)
    if (n) {
        const unsigned tiles = (unsigned)((n + kPartTile - 1) / kPartTile);
        hipLaunchKernelGGL(k_shard_part_count<W>, dim3(tiles), dim3(kPartTile), 0, st, E.d, (uint32_t)base, (uint32_t)n, E.part_cnt);
        hipLaunchKernelGGL(k_shard_part_scan<W>, dim3(1), dim3(1024), 0, st, E.d, tiles, E.part_cnt, (uint32_t)nodes_old);
        hipLaunchKernelGGL(k_shard_part_write<W>, dim3(tiles), dim3(kPartTile), 0, st, E.d, (uint32_t)base, (uint32_t)n, E.part_cnt, (uint32_t)nodes_old);
        ACX_HIP_TRY(hipGetLastError());
    }
    unsigned long long now = 0;
    ACX_HIP_TRY(hipMemcpyAsync(&now, E.d.ctl + C_NODES, 8, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    E.nodes_host = now;
    return ACX_OK;
}

template <typename W> static int shard_walk(ShardEngine<W>& E, int64_t id, int64_t cap, int64_t* out, hipStream_t st) {
    if (int rc = E.await_ready(st)) return rc;
    if (id < 0 || (uint64_t)id >= E.cap_nodes || cap < 1) return fail(ACX_E_INVAL, "acx_shard_walk: bad argument");
    const uint32_t c = (uint32_t)std::min<int64_t>(cap, ShardEngine<W>::kWalkCap);
    hipLaunchKernelGGL(k_shard_walk<W>, dim3(1), dim3(1), 0, st, E.d, (uint32_t)id, c, E.d_walk);
    ACX_HIP_TRY(hipGetLastError());
    ACX_HIP_TRY(hipMemcpyAsync(out, E.d_walk, 16, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    if (out[1] > 0) ACX_HIP_TRY(hipMemcpy(out + 2, E.d_walk + 2, (size_t)out[1] * 16, hipMemcpyDeviceToHost));
    return ACX_OK;
}

template <typename W> static int shard_node_info(ShardEngine<W>& E, int64_t id, int64_t* info) {
    ACX_HIP_TRY(hipDeviceSynchronize());
    unsigned long long n = 0;
    ACX_HIP_TRY(hipMemcpy(&n, E.d.ctl + C_NODES, 8, hipMemcpyDeviceToHost));
    if (id < 0 || (uint64_t)id >= n) return fail(ACX_E_INVAL, "acx_shard_node_info: id out of range");
    uint8_t a = 0, l = 0;
    int64_t pr = 0;
    ACX_HIP_TRY(hipMemcpy(&a, E.d.act + id, 1, hipMemcpyDeviceToHost));
    ACX_HIP_TRY(hipMemcpy(&l, E.d.tlen + id, 1, hipMemcpyDeviceToHost));
    ACX_HIP_TRY(hipMemcpy(&pr, E.d.pref + id, 8, hipMemcpyDeviceToHost));
    info[0] = pr < 0 ? -1 : (int64_t)a;
    info[1] = l;
    info[2] = pr;
    return ACX_OK;
}

}  // namespace acx

using namespace acx;

// ------------------------------------------------------------------ sharded frontier: C ABI ----
struct acx_shard {
    acx::ShardAny any;
};

extern "C" {

int acx_shard_key_words(int L) { return L <= 29 ? 2 : 4; }  // (62 .. 64: four words as well, in the encoding of acx_keys.h)

int acx_shard_layout(int64_t n_parents, int world, int key_words, int fill_q8, int64_t* subregions, int64_t* subcap, int64_t* region_words) {
    if (n_parents < 1 || world < 1 || (key_words != 2 && key_words != 4) || !subregions || !subcap || !region_words) return fail(ACX_E_INVAL, "acx_shard_layout: bad argument");
    *subregions = kShardSub;
    shard_layout(n_parents, world, key_words + 1, fill_q8, subcap, region_words);
    return ACX_OK;
}

acx_shard* acx_shard_create(int L, int cyclical, int64_t node_cap, int64_t chunk_parents, int rank, int world) {
    if (!have_device()) return nullptr;
    if (L < 1 || L > 64 || node_cap < 1 || chunk_parents < 1 || chunk_parents > (1ll << 27) || world < 1 || world > 64 || rank < 0 || rank >= world) {
        fail(ACX_E_INVAL, "acx_shard_create: bad argument (1 <= L <= 64, world <= 64, chunk_parents <= 2^27)");
        return nullptr;
    }
    acx_shard* h = new (std::nothrow) acx_shard();
    if (!h) return nullptr;
    h->any.width = L <= 29 ? 0 : (L <= 61 ? 1 : 2);
    int rc;
    if (h->any.width == 2) {
        h->any.e128x = new ShardEngine<u128x>();
        rc = h->any.e128x->init(L, cyclical, node_cap, chunk_parents, rank, world);
    } else if (h->any.width == 1) {
        h->any.e128 = new ShardEngine<u128>();
        rc = h->any.e128->init(L, cyclical, node_cap, chunk_parents, rank, world);
    } else {
        h->any.e64 = new ShardEngine<uint64_t>();
        rc = h->any.e64->init(L, cyclical, node_cap, chunk_parents, rank, world);
    }
    if (rc != ACX_OK) {
        delete h->any.e64;
        delete h->any.e128;
        delete h->any.e128x;
        delete h;
        return nullptr;
    }
    return h;
}

void acx_shard_destroy(acx_shard* h) {
    if (!h) return;
    delete h->any.e64;
    delete h->any.e128;
    delete h->any.e128x;
    delete h;
}

int acx_shard_attach(acx_shard* h, int64_t* d_log, int64_t log_words, int64_t* d_send, int64_t send_words, int32_t* d_gmask) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_attach: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_attach<W>(E, d_log, log_words, d_send, send_words, d_gmask));
}

int acx_shard_root_record(acx_shard* h, const int8_t* h_presentation, int64_t* h_record) {
    if (!h || !h_presentation || !h_record) return fail(ACX_E_INVAL, "acx_shard_root_record: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_root<W>(E, h_presentation, h_record));
}

int acx_shard_seed(acx_shard* h, const int64_t* h_record, void* stream) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_seed: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_seed<W>(E, h_record, (hipStream_t)stream));
}

int acx_shard_chunk_expand(acx_shard* h, int64_t c0, int64_t c1, int level_first, int fill_q8, int64_t* recv_off, int64_t* words, void* stream) {
    if (!h || c0 < 0 || c1 <= c0 || !recv_off || !words) return fail(ACX_E_INVAL, "acx_shard_chunk_expand: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_chunk_expand<W>(E, c0, c1, level_first, fill_q8, recv_off, words, (hipStream_t)stream));
}

int acx_shard_chunk_insert(acx_shard* h, void* stream) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_chunk_insert: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_chunk_insert<W>(E, (hipStream_t)stream));
}

int acx_shard_chunk_insert_dead(acx_shard* h, void* stream) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_chunk_insert_dead: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_chunk_insert<W>(E, (hipStream_t)stream, false));
}

int acx_shard_chunk_commit(acx_shard* h, int64_t max_nodes, void* stream) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_chunk_commit: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_chunk_commit<W>(E, max_nodes, (hipStream_t)stream));
}

int acx_shard_ctl_snapshot(acx_shard* h, int slot, void* stream) {
    if (!h || slot < 0 || slot >= kCtlSlots) return fail(ACX_E_INVAL, "acx_shard_ctl_snapshot: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_ctl_snapshot<W>(E, slot, (hipStream_t)stream));
}

int acx_shard_ctl_wait(acx_shard* h, int slot, int64_t* h_ctl) {
    if (!h || slot < 0 || slot >= kCtlSlots || !h_ctl) return fail(ACX_E_INVAL, "acx_shard_ctl_wait: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_ctl_wait<W>(E, slot, h_ctl));
}

int acx_shard_fail(acx_shard* h, void* stream) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_fail: bad argument");
    ACX_SHARD_DISPATCH(&h->any, {
        if (int rc = E.await_ready((hipStream_t)stream)) return rc;
        hipLaunchKernelGGL(k_shard_fail, dim3(1), dim3(1), 0, (hipStream_t)stream, E.d.ctl, (unsigned long long)FAIL_HOST);
        ACX_HIP_TRY(hipGetLastError());
    });
    return ACX_OK;
}

int acx_shard_owner(int L, const int64_t* h_key_words, int world) {
    if (L < 1 || L > 64 || !h_key_words || world < 1) return fail(ACX_E_INVAL, "acx_shard_owner: bad argument");
    if (L > 61) {
        u128x k0, k1;
        recio<u128x>::get(h_key_words, k0, k1);
        return (int)owner_of_key<u128x>(k0, k1, (uint32_t)world);
    }
    if (L <= 29) {
        uint64_t k0, k1;
        recio<uint64_t>::get(h_key_words, k0, k1);
        return (int)owner_of_key<uint64_t>(k0, k1, (uint32_t)world);
    }
    u128 k0, k1;
    recio<u128>::get(h_key_words, k0, k1);
    return (int)owner_of_key<u128>(k0, k1, (uint32_t)world);
}

int acx_shard_check_owners(acx_shard* h, int64_t* n_bad, void* stream) {
    if (!h || !n_bad) return fail(ACX_E_INVAL, "acx_shard_check_owners: bad argument");
    ACX_SHARD_DISPATCH(&h->any, {
        hipStream_t st = (hipStream_t)stream;
        if (int rc = E.await_ready(st)) return rc;
        ACX_HIP_TRY(hipMemsetAsync(E.d_find, 0, 8, st));
        hipLaunchKernelGGL(k_shard_check_owners<W>, dim3(1024), dim3(256), 0, st, E.d, E.replicated ? 0xFFFFFFFFu : (uint32_t)E.own_from, (unsigned long long*)E.d_find);  // (still replicated: every node lives on every rank)
        ACX_HIP_TRY(hipGetLastError());
        ACX_HIP_TRY(hipMemcpyAsync(n_bad, E.d_find, 8, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
    });
    return ACX_OK;
}


int acx_shard_set_replicated(acx_shard* h, int on) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_set_replicated: bad argument");
    ACX_SHARD_DISPATCH(&h->any, {
        if (on && (E.nodes_host || E.chunk_seq)) return fail(ACX_E_INVAL, "acx_shard_set_replicated: only before the root is seeded");
        if (!on && E.replicated) return fail(ACX_E_INVAL, "acx_shard_set_replicated: a replicated phase ends with acx_shard_partition");
        E.replicated = on != 0;
    });
    return ACX_OK;
}

int acx_shard_partition(acx_shard* h, void* stream) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_partition: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_partition<W>(E, (hipStream_t)stream));
}

int acx_shard_walk(acx_shard* h, int64_t id, int64_t cap, int64_t* h_out, void* stream) {
    if (!h || !h_out) return fail(ACX_E_INVAL, "acx_shard_walk: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_walk<W>(E, id, cap, h_out, (hipStream_t)stream));
}

int acx_shard_find(acx_shard* h, int64_t gpos, int64_t* id, void* stream) {
    if (!h || !id) return fail(ACX_E_INVAL, "acx_shard_find: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_find<W>(E, gpos, id, (hipStream_t)stream));
}

int acx_shard_node_info(acx_shard* h, int64_t id, int64_t* h_info3) {
    if (!h || !h_info3) return fail(ACX_E_INVAL, "acx_shard_node_info: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_node_info<W>(E, id, h_info3));
}

}  // extern "C"
