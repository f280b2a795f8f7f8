/*
 * acx.h -- C ABI of libacx.so: the MI355X (gfx950) Andrews-Curtis environment step and search
 * frontier.  This is the drop-in boundary for the hot path of shehper/AC-Solver.
 *
 * The reference is pure Python and has no FFI; its boundary is the Python surface
 *     ac_solver/envs/ac_moves.py:159   ACMove            (+ :4 concatenate_relators, :79 conjugate)
 *     ac_solver/envs/utils.py:175,243  simplify_relator / simplify_presentation
 *     ac_solver/envs/ac_env.py:95,115  ACEnv.step / ACEnv.reset
 *     ac_solver/search/breadth_first.py:15  bfs
 *     ac_solver/search/greedy.py:15         greedy_search
 * Each entry point below names the reference function it replaces.  The ctypes binding a maintainer
 * would add is ac-solver_amd/ac_solver/_acx.py.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer
 *   - every function returns 0 on success, a negative ACX_E_* code on failure, and
 *     acx_last_error() then returns a thread-local message
 *   - pointers named d_* are device pointers on the current device (torch `tensor.data_ptr()`),
 *     h_* are host pointers; `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream) or NULL
 *   - a presentation is 2L int8: two relators, letters in {+-1, +-2}, left aligned, zero padded
 *     (ac_solver/envs/utils.py:4-8); rows of a batch are contiguous: [n, 2L]
 *   - per-row error bytes: 0 ok, 1 the reference raises AssertionError, 2 IndexError, 3 ValueError,
 *     250 row is not a packable presentation (use ACX_F_BYTES)
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with ACX_E_NODEVICE
 */
#ifndef ACX_H
#define ACX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACX_VERSION 600  /* history: INTEGRATION.md, "ACX_VERSION history" */

/* return codes */
#define ACX_OK 0
#define ACX_E_INVAL (-1)    /* bad argument */
#define ACX_E_NODEVICE (-2) /* no HIP device / HIP runtime error */
#define ACX_E_NOMEM (-3)
#define ACX_E_ROWERR (-4)   /* some row hit a reference exception: inspect the err bytes */
#define ACX_E_CAPACITY (-5) /* a caller-provided output buffer is too small */

/* move flags */
#define ACX_F_CYCLICAL 1    /* ACMove(cyclical=True) */
#define ACX_F_NO_SIMPLIFY 2 /* concatenate_relators / conjugate only (ACX_F_BYTES only) */
#define ACX_F_NO_MOVE 4     /* simplify_presentation only (ACX_F_BYTES only) */
#define ACX_F_BYTES 8       /* byte-exact kernel: any int8 letters, invalid rows, L <= 128 */

/* dtypes of action / observation buffers */
#define ACX_U8 0
#define ACX_I32 1
#define ACX_I64 2
#define ACX_I8 3
#define ACX_F32 4

int acx_version(void);
const char *acx_last_error(void);
int acx_device_count(void); /* 0 when no GPU is visible; never fails */

/* ---- stateless batched moves --------------------------------------------------------------
 * Replaces ACMove (ac_moves.py:159-231) applied row-wise: out[k] = ACMove(action[k], in[k], L,
 * cyclical).  d_len [n,2] int32 receives the relator lengths, d_err [n] the per-row error byte
 * (rows in error are copied through unchanged).  d_fit (nullable, ACX_F_BYTES only) [n] int32
 * receives the new length of the touched relator or -1 when the move did not fit.
 * Without ACX_F_BYTES the packed kernel runs (L <= 64, valid {+-1,+-2} rows). */
int acx_move_batch_device(const int8_t *d_in, const void *d_action, int action_dtype, int64_t n, int L, int flags,
                          int8_t *d_out, int32_t *d_len, uint8_t *d_err, int32_t *d_fit, void *stream);
/* host-buffer convenience around the same kernels (staging through a cached device buffer) */
int acx_move_batch(const int8_t *h_in, const uint8_t *h_action, int64_t n, int L, int flags, int8_t *h_out,
                   int32_t *h_len, uint8_t *h_err, int32_t *h_fit);
/* Replaces simplify_relator (utils.py:175-240) on n rows of `width` int8: h_out [n,width] gets the
 * reduced word left aligned, h_len [n,2] = (reduced length, letters on entry), h_err as above. */
int acx_simplify_relators(const int8_t *h_in, int64_t n, int width, int cyclical, int8_t *h_out, int32_t *h_len,
                          uint8_t *h_err);

/* ---- path replay -------------------------------------------------------------------
 * What the reference's search scripts do with a path they found (breadth_first.py:113-126, greedy.py:130-143): apply its moves
 * to the presentation one after the other -- for n paths in ONE launch, a lane per path.  Path i has the actions
 * h_actions[h_offsets[i] .. h_offsets[i + 1]) (0..11, WITHOUT the root entry of a search path; h_offsets[0] = 0, n + 1 entries);
 * h_total_len (same layout) receives the total length after every move; a move on which the reference's ACMove raises ends its
 * path: h_err[i] = that error code (251: an action outside 0..11; 250: the row is not a presentation over {+-1,+-2}), its entry
 * and the later ones read -1.  h_final (nullable) [n, 2L]: the presentations the paths end at.  L <= 64. */
int acx_replay_paths(const int8_t *h_presentations, int64_t n, int L, int cyclical, const int32_t *h_actions,
                     const int64_t *h_offsets, int32_t *h_total_len, uint8_t *h_err, int8_t *h_final);

/* ---- vectorised environment ---------------------------------------------------------------
 * n independent ACEnv instances (ac_env.py:56-134) resident on one device in packed form.
 * A handle is bound to the device current at creation and is not thread safe. */
typedef struct acx_env acx_env;

#define ACX_ENV_RECORD_ACTIONS 1 /* keep the per-episode action history (info["actions"], ac_env.py:96,112) */

acx_env *acx_env_create(int64_t n, int L, int64_t horizon, int flags);
void acx_env_destroy(acx_env *env);
/* The entries below that take HOST buffers and a `stream` queue their uploads, their kernel and their read-back on that
 * stream and return once it has drained: they are ordered with the steps the caller queued on the same stream (NULL =
 * the null stream), also when that stream is a non-blocking side stream. */
/* ACEnvConfig.initial_state for the envs idx[0..n_idx) (NULL = all, rows in env order); also resets them.
 * Rows must be valid presentations (ACEnvConfig.__post_init__, ac_env.py:22-35): else ACX_E_ROWERR. */
int acx_env_set_initial(acx_env *env, const int8_t *h_states, const int64_t *h_idx, int64_t n_idx, void *stream);
/* ACEnv.reset (ac_env.py:115-131): h_states NULL -> back to the initial state, else
 * options={"starting_state": row}.  Zeroes count_steps and the action history of those envs. */
int acx_env_reset(acx_env *env, const int8_t *h_states, const int64_t *h_idx, int64_t n_idx, void *stream);
/* The same from DEVICE buffers (the PPO driver restarts the finished environments of a rollout step from rows it gathers on the
 * device): d_states [n_idx, 2L] int8 or NULL (-> initial state), d_idx [n_idx] (required; every index in [0, n) -- not checked),
 * d_rowerr [n_idx] receives 0 or ACX_ERR_UNPACKABLE per row (an env whose row is flagged keeps its state).  Only queues the kernel:
 * no copy, no synchronisation; ordered with the steps on `stream`. */
int acx_env_reset_device(acx_env *env, const int8_t *d_states, const int64_t *d_idx, int64_t n_idx, uint8_t *d_rowerr,
                         void *stream);
/* Supermoves.  The reference declares `use_supermoves` (envs/ac_env.py:20, agents/args.py:107-114) and raises
 * NotImplementedError when it is set (ac_env.py:62-65); there is no reference behaviour to reproduce.  Opt-in here: action
 * 12 + s runs the base moves h_moves[h_offsets[s] .. h_offsets[s + 1]) (each 0..11) as ONE environment step -- state = the
 * composition of ACMove, reward / done / truncated from the final state, one count_steps, one history entry; if one of the
 * base moves raises in the reference's ACMove the whole step does (state unchanged).  n_super = 0 removes them.
 * At most 52 supermoves of at most 64 moves. */
int acx_env_set_supermoves(acx_env *env, const uint8_t *h_moves, const int32_t *h_offsets, int n_super, void *stream);
/* ACEnv.step for all n envs (ac_env.py:95-113).  d_obs [n,2L] (ACX_I8 or ACX_F32) receives the state
 * after the step -- after the autoreset when `autoreset` and the env finished, in which case
 * d_final_obs (nullable, same dtype) holds the terminal observation.  d_reward [n] f32 =
 * max_reward*done - total_length*(1-done), clipped to [clip_lo, clip_hi] when clip_lo < clip_hi.
 * d_done / d_trunc [n] u8.  Any pointer may be NULL to skip that output. */
int acx_env_step(acx_env *env, const void *d_actions, int action_dtype, void *d_obs, int obs_dtype, float *d_reward,
                 float clip_lo, float clip_hi, uint8_t *d_done, uint8_t *d_trunc, void *d_final_obs, int autoreset,
                 void *stream);
/* Same step with host buffers (single-env Python surface): h_actions [n] int64, h_obs / h_final_obs
 * [n,2L] int8, h_reward [n] f32 (unclipped), h_done / h_trunc [n] u8, h_err [n] (nullable) = the per-env error bytes of
 * this step (what acx_env_get_errors(clear = 1) would return next).  One upload, the kernel, one read-back and one
 * synchronisation of `stream`. */
int acx_env_step_host(acx_env *env, const int64_t *h_actions, int8_t *h_obs, float *h_reward, uint8_t *h_done,
                      uint8_t *h_trunc, int8_t *h_final_obs, int autoreset, uint8_t *h_err, void *stream);
/* T fused steps from an action tape d_tape [T,n] u8 with the state resident in registers;
 * d_reward / d_done / d_trunc are [T,n] (nullable).  Same semantics as T calls of acx_env_step. */
int acx_env_rollout(acx_env *env, const uint8_t *d_tape, int64_t T, float *d_reward, float clip_lo, float clip_hi,
                    uint8_t *d_done, uint8_t *d_trunc, int autoreset, void *stream);
/* current observation of all envs, [n,2L] ACX_I8 / ACX_F32, device buffer */
int acx_env_observe(acx_env *env, void *d_obs, int obs_dtype, void *stream);
/* host copies for the single-env Python surface: state [n_idx,2L] i8, lengths [n_idx,2] i32,
 * count_steps [n_idx] i32 */
int acx_env_get(acx_env *env, const int64_t *h_idx, int64_t n_idx, int8_t *h_state, int32_t *h_len, int32_t *h_count,
                void *stream);
/* info["actions"] of env i (needs ACX_ENV_RECORD_ACTIONS): which = 0 the actions since its last reset,
 * which = 1 the actions of the episode that the last step's autoreset just ended (final_info). */
int acx_env_get_actions(acx_env *env, int64_t i, int which, int32_t *h_out, int64_t cap, int64_t *n_out, void *stream);
/* sticky per-env error bytes (a move hit the reference's AssertionError / IndexError: that env's state and
 * step counter were left untouched, as when the reference's step() raises); h_err [n]; clear != 0 zeroes them */
int acx_env_get_errors(acx_env *env, uint8_t *h_err, int clear, void *stream);
int64_t acx_env_max_reward(const acx_env *env); /* horizon * L * 2, ac_env.py:80 */

/* ---- search ---------------------------------------------------------------------------------
 * Replaces bfs (breadth_first.py:15-97) and greedy_search (greedy.py:15-121): same visiting order,
 * same (solved, path) result.  path_* receive (action, total_length) pairs starting with (-1, len0);
 * *path_n == 0 encodes the reference's `None`. */
typedef struct {
    int64_t nodes;     /* len(tree_nodes) at exit */
    int64_t expanded;  /* parents expanded */
    int64_t children;  /* ACMove evaluations */
    int64_t levels;    /* BFS levels / greedy batches processed */
    int32_t min_len;   /* smallest total length generated (in whole batches: an unsolved search may count children of its last batch that the reference would not have generated any more) */
    double seconds;    /* device time of the search loop */
} acx_search_stats;

#define ACX_SEARCH_BFS 0
#define ACX_SEARCH_GREEDY 1

/* max_relator_length L <= 64 (the reference takes any, breadth_first.py:42-45; its Miller-Schupp generator reaches 64 at n = 14): one
 * 128-bit key word per relator.  L <= 29: 64-bit keys; <= 61: word | length in one unsigned __int128; 62 .. 64: a key that names FREELY
 * REDUCED words (csrc/acx_keys.h) -- every state ACMove produces is one, the presentation handed in has to be (ACX_E_INVAL otherwise);
 * acx_search_groups takes a batch of such presentations through acx_search_many (shared bfs launches / greedy jobs on workgroup slots, as for the
 * other widths) instead of the one-launch-per-key-width scheduler. */
int acx_search(int kind, const int8_t *h_presentation, int L, int64_t max_nodes, int cyclical, int32_t *solved,
               int32_t *path_action, int32_t *path_len, int64_t path_cap, int64_t *path_n, acx_search_stats *stats);

/* verbose=True of bfs / greedy_search (breadth_first.py:79-82, greedy.py:85-89: "New minimal length found: l" whenever a child
 * is shorter than everything generated before it): with the switch on (it is per calling thread: searches running on other
 * threads are not affected), acx_search records those lengths, in the order the
 * reference prints them and up to the child that ends the search; acx_search_last_minima returns the calling thread's last
 * sequence (*n = its length, the first min(*n, cap) entries are written).  greedy_search then runs batch by batch on the
 * host-driven path (same result, slower). */
int acx_search_minima_enable(int on);
int acx_search_last_minima(int32_t *lengths, int64_t cap, int64_t *n);

/* Test hook (repeat-determinism tests): with the switch on, every acx_search of the process ends with one extra pass that
 * folds (id, packed key, parent, action) of ALL nodes of the search into a 64-bit sum; acx_search_last_digest returns the
 * one of the calling thread's last search.  Two runs of the same search must give the same digest, node for node. */
int acx_search_digest_enable(int on);
int acx_search_last_digest(uint64_t *digest);

/* acx_search keeps the device blocks of a finished search for the next search of the same host thread (hipMalloc /
 * hipFree are slow and hipFree synchronises the device); this returns the cached blocks -- and the idle pinned host buffers the
 * sweeps read their results into -- to the driver */
int acx_release_cached_memory(void);

/* n independent searches of the same kind / budget (the batch driver trivialize_miller_schupp_through_search,
 * miller_schupp.py:95-177, runs them one after another).  Both kinds run as GROUPS of searches, each search with its own visited
 * table and node arena.  bfs: the searches of a group share the launches of the fused single search (acx_bfs_many.h: a tile of one
 * search's batch per workgroup, a batch of every running search per round of four launches).  greedy_search: ONE launch of up to
 * two persistent workgroups per compute unit (512 on an MI355X; ACX_OPT_GREEDY_SLOTS), each with the memory of one search, which take the searches from a counter one
 * after the other (k_greedy_sched, acx_greedy.h).  A greedy search that outgrows a capacity of its workgroup is rerun alone through
 * acx_search.  `n_threads` only matters on the fallback path (n == 1, the option ACX_OPT_GREEDY_HOST, `verbose` minima or the digest
 * hook on): there that many host threads run one acx_search each, every search on its own HIP stream; 1..64, clamped.
 * Row k of every output belongs to presentation k ([n, path_cap] for the paths); rc_out[k] is that search's return code
 * (ACX_E_CAPACITY when its path needs more than path_cap entries: path_n[k] then holds the required size).  Results are
 * identical to n calls of acx_search. */
/* Several batches of searches in one call: batch g has n[g] presentations of max_relator_length L[g] (h_presentations[g]:
 * [n[g], 2 L[g]]) -- the Miller-Schupp presentations of each n have their own max_relator_length, and the reference's driver walks
 * through all of them (miller_schupp.py:140-158).  Outputs as acx_search_many's, the batches one behind the other in batch order
 * (sum of n[g] rows).  greedy_search: ALL searches are jobs of one launch per key width (k_greedy_sched, acx_greedy.h): a fixed set of
 * workgroups, each with the memory of one search, takes them from a counter one after the other -- no launch waits for its slowest
 * search, no batch for another.  bfs: the batches one after the other through acx_search_many (a batch fills the GPU by itself). */
int acx_search_groups(int kind, int n_groups, const int8_t *const *h_presentations, const int64_t *n, const int32_t *L,
                      int64_t max_nodes, int cyclical, int32_t *solved, int32_t *path_action, int32_t *path_len,
                      int64_t path_cap, int64_t *path_n, acx_search_stats *stats, int32_t *rc_out);
int acx_search_many(int kind, const int8_t *h_presentations, int64_t n, int L, int64_t max_nodes, int cyclical,
                    int n_threads, int32_t *solved, int32_t *path_action, int32_t *path_len, int64_t path_cap,
                    int64_t *path_n, acx_search_stats *stats, int32_t *rc_out);

/* ---- options ---------------------------------------------------------------------------------------------
 * Tuning and test knobs of the search entry points, set through the ABI (process wide); nothing on a call path reads the
 * environment.  value < 0 restores the built-in default; acx_get_option returns -1 for "default".
 * The only environment variable libacx looks at is ACX_DEBUG (diagnostics on stderr), once, when it is loaded. */
#define ACX_OPT_BFS_NO_RUNAHEAD 0  /* 1: acx_search(bfs) reads every batch's decision back (the path small frontiers and verbose searches take) */
#define ACX_OPT_GREEDY_HOST 1      /* 1: greedy searches run batch per launch from the host (the path of verbose searches and of capacity fallbacks) */
#define ACX_OPT_GREEDY_HAND_MIN 2  /* a bucket of at least this many queued parents of a single greedy search goes to the whole-GPU kernels (default 512; 0: never) */
#define ACX_OPT_MEGA_RANK_MAX 3    /* handed-off buckets up to this size are ordered by the counting sort (default 16384, at least 256) */
#define ACX_OPT_BFS_MANY_BMAX 4    /* parents per search and round of acx_search_many(bfs) (default 32768, 128 .. 2^22) */
#define ACX_OPT_GREEDY_SLOTS 5     /* persistent workgroups of acx_search_many / acx_search_groups (greedy) (default 512 = two per compute unit; slots of the call = min(this, jobs, what the memory holds)) */
#define ACX_OPT_GENERAL_MOVE 6     /* 1: the general move code also for roots in normal form (tests: both codes must build the same arena) */
#define ACX_OPT_GREEDY_SCRATCH 7   /* acx_search_many / acx_search_groups, greedy: entries of a slot's own sort scratch (default 65536; a bucket that needs more borrows a shared full-size region; tests: small) */
#define ACX_OPT_GREEDY_KEEP_ORDER 8 /* 1: acx_search_many / acx_search_groups (greedy) take the jobs in the caller's order (default: smaller max_relator_length first, longer relators first) */
#define ACX_OPT_COUNT 9
int acx_set_option(int option, int64_t value);
int64_t acx_get_option(int option);

/* ---- sharded BFS frontier (one engine per GPU) ---------------------------------------------------
 * The multi-GPU form of bfs (breadth_first.py:15-97): states are partitioned over the ranks by acx_shard_owner -- a function of
 * the conjugacy classes of the two relators, which the eight conjugation moves of ac_moves.py:192-229 leave alone, so that a rank
 * owns most children of its own nodes (ac-solver_amd/csrc/acx_owner.h).  A level is processed
 * in chunks of consecutive global frontier positions [c0, c1); per chunk every rank calls
 *     acx_shard_chunk_expand  -> all-to-all of the send buffer into the receive area  (RCCL; nothing at world 1)
 *     acx_shard_chunk_insert  -> all-reduce (sum) of the child masks                   (RCCL; nothing at world 1)
 *     acx_shard_chunk_commit
 * and nothing is read back in between: the budget / success / error decisions of the reference (breadth_first.py:84-95) are
 * taken on the device, identically on every rank, and land in the CONTROL BLOCK, which the host reads a few chunks late
 * (acx_shard_ctl_snapshot / acx_shard_ctl_wait).  Once its status word leaves 0 every later call is a no-op on the device.
 * Orchestration: ac-solver_amd/ac_solver/search/sharded.py.
 *
 * Buffers (device memory of the CALLER, e.g. torch tensors, handed over once with acx_shard_attach):
 *   log    int64[log_words]: the record log.  The receive area of a chunk is the slice [recv_off, recv_off + words) that
 *          acx_shard_chunk_expand returns; it is never recycled (a visited-table slot names its state by the offset of the
 *          record that claimed it).  At world 1 the chunk is expanded straight into it.
 *   send   int64[send_words] (world > 1): the chunk's send buffer, `words` long, same layout.
 *   gmask  int32[(chunk_parents + 3) / 4 * 2] (whole quads of parents; the all-reduce covers the first (n + 1) / 2 words of a chunk of n parents): one 12-bit mask per parent of the chunk -- bit a set when child (parent, a) is a new
 *          state owned by this rank --, two parents per word (parent p: bits 16 (p & 1) .. + 11 of word p >> 1); the caller
 *          all-reduces (sum == or: a child has one owner, no field carries) its first (c1 - c0 + 1) / 2 words between insert and
 *          commit.
 * Layout of a chunk's `words`: world * S regions (region d * S + s = sub-region s for / from rank d; S, subcap, region_words from
 * acx_shard_layout), each = 4 header words [records written, smallest tag of a length-2 child, smallest (tag << 8 | code) of a
 * raising move, the sender's failure code] + subcap records of key_words + 1 int64: the packed key, then
 * parent's local id | (12 * (global parent position - c0) + action) << 32.  An all-to-all with equal splits of
 * S * region_words int64 per rank delivers it. */
typedef struct acx_shard acx_shard;
int acx_shard_key_words(int L); /* 2 for L <= 29, 4 for L <= 64 (62 .. 64: the reduced-word key of csrc/acx_keys.h) */
/* geometry of a chunk of n_parents global parents (pure host arithmetic, no device needed).  fill_q8: capacity of a region at
 * world > 1 in 1/256 of the even share of ALL children (+ two workgroups' worth); <= 0 or 321 .. 2^20 - 1: the default 320 =
 * 1.25 x; >= 2^20: the hard bound (every workgroup sends a region all it has: safe for any input, world^2 x the even share).
 * Only the children whose owner is another rank are sent (a few per cent, acx_owner.h), and the all-to-all moves whole regions:
 * the orchestrator passes 1.25 x the fullest region of the previous level + 12 / 256 (control word ACX_SHARD_CTL_LEVEL_FILL, max over the
 * ranks) and reruns the search with the default, then with the hard bound, if a region ever overflows (failure code 1). */
int acx_shard_layout(int64_t n_parents, int world, int key_words, int fill_q8, int64_t *subregions, int64_t *subcap, int64_t *region_words);
/* node_cap: local nodes; chunk_parents: the largest chunk (global parents) */
acx_shard *acx_shard_create(int L, int cyclical, int64_t node_cap, int64_t chunk_parents, int rank, int world);
void acx_shard_destroy(acx_shard *h);
int acx_shard_attach(acx_shard *h, int64_t *d_log, int64_t log_words, int64_t *d_send, int64_t send_words, int32_t *d_gmask);
/* the root as a record (tag 0, parent_ref -1), host buffer of key_words + 2 int64; every rank calls it (it also selects
 * the move code of the search from the root's form) */
int acx_shard_root_record(acx_shard *h, const int8_t *h_presentation, int64_t *h_record);
/* rank that owns the state with these key words (the first key_words int64 of a record): pure host arithmetic, no device needed.
 * Negative: an error code. */
int acx_shard_owner(int L, const int64_t *h_key_words, int world);
/* the owner of the root passes its record (local node 0, global position 0), every other rank NULL */
int acx_shard_seed(acx_shard *h, const int64_t *h_record, void *stream);
/* children of the local nodes of the running level whose global positions lie in [c0, c1), routed to their owners (children
 * equal to their parent, children that undo their parent's move in a normal-form search with cyclical = 0, and duplicates
 * inside a workgroup are never sent).  level_first != 0 on the first chunk of a level: the nodes committed during the previous
 * level become the running one.  fill_q8: as in acx_shard_layout (the same value on every rank). */
int acx_shard_chunk_expand(acx_shard *h, int64_t c0, int64_t c1, int level_first, int fill_q8, int64_t *recv_off, int64_t *words, void *stream);
/* exact dedup of what the receive area holds against the visited table and among itself (minimum tag wins) -> gmask */
int acx_shard_chunk_insert(acx_shard *h, void *stream);
/* the failure path of a rank whose dedup call could not run: the oldest expanded chunk's masks WITHOUT the dedup (its records are
 * dropped), so that the chunk stays in the engine's ring and acx_shard_chunk_commit still takes the global decisions from the
 * all-reduced masks; the caller has marked the rank failed (acx_shard_fail), every rank stops at the next chunk's headers */
int acx_shard_chunk_insert_dead(acx_shard *h, void *stream);
/* decisions + the new states below the cutoff become local nodes, in tag order (max_nodes = max_nodes_to_explore) */
int acx_shard_chunk_commit(acx_shard *h, int64_t max_nodes, void *stream);
/* control block: ACX_SHARD_CTL_WORDS int64 */
#define ACX_SHARD_CTL_WORDS 16
#define ACX_SHARD_CTL_STATUS 0       /* 0 running, 1 solved, 2 budget reached, 3 a move raised (AssertionError), 4 a rank failed */
#define ACX_SHARD_CTL_NODES_GLOBAL 1 /* len(tree_nodes) */
#define ACX_SHARD_CTL_NEXT_COUNT 2   /* new states of the running level so far */
#define ACX_SHARD_CTL_EXPANDED 3     /* parents expanded */
#define ACX_SHARD_CTL_SOLVED_TAG 4   /* 12 * global position + action of the child that ended the search */
#define ACX_SHARD_CTL_NODES 5        /* local nodes */
#define ACX_SHARD_CTL_FAIL_LOCAL 8   /* this rank's failure code: 1 region / log overflow, 2 node capacity, 3 table full, 4 host */
#define ACX_SHARD_CTL_MIN_LEN 9      /* smallest total length this rank generated */
#define ACX_SHARD_CTL_FAIL_SEEN 10   /* failure code received from some rank (status 4) */
#define ACX_SHARD_CTL_LEVEL_FILL 12  /* fullest region this rank received in the running level, in 1/256 of the even share (0: no chunk large enough to tell) */
/* queue a copy of the control block into pinned slot `slot` (0..3) behind the work queued so far / wait for it */
int acx_shard_ctl_snapshot(acx_shard *h, int slot, void *stream);
int acx_shard_ctl_wait(acx_shard *h, int slot, int64_t *h_ctl);
/* mark this rank as failed (an exception on the caller's side): travels to every rank with the next chunk's headers */
int acx_shard_fail(acx_shard *h, void *stream);
/* Replicated small levels (round 6).  acx_shard_set_replicated(h, 1), before the root is seeded and on EVERY rank (each of which then
 * seeds the root): the levels that follow are processed WHOLE by every rank with the world-1 kernels -- acx_shard_layout with world 1,
 * no all-to-all, no mask all-reduce; every rank commits the same nodes in the same order, each parent reference naming the rank itself.
 * acx_shard_partition ends the phase between two levels (after the level's last acx_shard_ctl_wait, no chunk in flight): of the newest
 * level a rank keeps copies of the nodes it owns (acx_shard_owner) as its slice of the next level; from then on the chunks are exchanged
 * as described above.  A search that ends inside the replicated phase has needed no collective at all.  Reference semantics unchanged
 * (breadth_first.py:61-95): the partition only changes WHO expands a frontier node. */
int acx_shard_set_replicated(acx_shard *h, int on);
int acx_shard_partition(acx_shard *h, void *stream);
/* the path of local node `id` towards the root while the parents are local: h_out[0] = the first parent reference that is not local
 * (rank << 40 | id on that rank; -1: the root was reached), h_out[1] = n <= cap (<= 1024), then n pairs (action, total length), node
 * `id` first, the root's action -1.  h_out: 2 + 2 * cap words.  (One launch + one copy per SEGMENT of a path instead of
 * acx_shard_node_info's synchronisation per node.) */
int acx_shard_walk(acx_shard *h, int64_t id, int64_t cap, int64_t *h_out, void *stream);
/* local id of the node of the running level at global position gpos, -1 when another rank owns it */
int acx_shard_find(acx_shard *h, int64_t gpos, int64_t *id, void *stream);
/* h_info3 = (action, total_length, parent_ref = rank << 40 | local id) of a local node; the root has action -1, parent_ref -1 */
int acx_shard_node_info(acx_shard *h, int64_t id, int64_t *h_info3);
/* test hook: *n_bad = local nodes that do not live on the rank acx_shard_owner names for their key, or whose stored class hashes
 * (inherited from the parent along conjugations) differ from the ones their key gives.  0 on a healthy engine. */
int acx_shard_check_owners(acx_shard *h, int64_t *n_bad, void *stream);

/* ---- PPO rollout: fused policy inference ---------------------------------------------------------------
 * The agent of ac_solver/agents/ppo_agent.py:11-109 -- actor and critic, each in_dim -> 256 -> 256 -> {n_actions, 1} with
 * tanh -- evaluated on n_env observations and sampled, in ONE kernel on the matrix cores (bf16 operands, f32 accumulation):
 * d_action[e] ~ Categorical(softmax(actor(obs[e]))) (an exponential race on a counter-based hash of (seed, e, action): pass a
 * fresh seed per call), d_logprob[e] its log-probability, d_value[e] = critic(obs[e]).  d_obs [n_env, in_dim] row-major, obs_dtype
 * ACX_F32 or ACX_I8 (what acx_env_step writes with either; the int8 rows are a quarter of the bytes).  d_actor / d_critic: the networks packed by
 * ac_solver/agents/fused_policy.py:pack_network (acx_policy_packed_bytes(in_dim) bytes each).  in_dim <= 80,
 * n_actions <= 16, hidden width 256.  Inference only; the PPO update runs in torch on the f32 master weights. */
int acx_policy_sample(const void *d_obs, int obs_dtype, int64_t n_env, int in_dim, const void *d_actor, const void *d_critic,
                      int n_actions, uint64_t seed, int64_t *d_action, float *d_logprob, float *d_value, void *stream);
int64_t acx_policy_packed_bytes(int in_dim);

/* ---- neighbourhood sizes ------------------------------------------------------------
 * Replaces `neibourhood` of the reference's C++ side program (barcode_analysis/5_steps_neibourhoods/neibourhoods.cpp:18-54,
 * moves and word arithmetic AC_UTILS_no_hash.cpp:83-211): number of distinct SORTED pairs of freely reduced relators
 * (no length cap, free reduction only) within `radius` moves of each presentation, under the 14 "classic" (classic != 0)
 * or the 12 "prime" moves.  h_presentations [n, 2L] int8 (zeros are dropped wherever they stand, as the reference's reader
 * does); h_sizes [n]; h_max_len [n] (nullable): the longest relator met.  One workgroup per presentation; relators of up to
 * 192 letters (ACX_E_CAPACITY beyond). */
int acx_ball_sizes(const int8_t *h_presentations, int64_t n, int L, int radius, int classic, int64_t *h_sizes,
                   int32_t *h_max_len);

/* Replaces the reference's simplex-data programs barcode_analysis/simplex_data_generation/{prime,classic}_moves/ac_bfs.cpp:12-98:
 * breadth-first search from <a, b> over sorted pairs of freely reduced relators of total length <= n; vertex k is the
 * k-th presentation the reference names, h_node_size[k] its total length ("0-filt"); h_edges [n_edges, 2] / h_edge_filt are
 * its "1-simplices" / "1-filt" lists in the order it writes them (an edge per (vertex, move) whose child has a larger name,
 * repeats included; filtration = the larger of the two sizes).  ACX_E_CAPACITY when cap_nodes / cap_edges are too small
 * (*n_nodes / *n_edges then hold the counts reached). */
int acx_simplex_graph(int n, int classic, int64_t cap_nodes, int64_t cap_edges, int64_t *n_nodes, uint8_t *h_node_size,
                      int64_t *n_edges, uint32_t *h_edges, uint8_t *h_edge_filt);

/* ---- a whole sharded bfs in ONE call (round 6) -----------------------------------------------------------------------------
 * bfs(presentation, max_nodes_to_explore, verbose, cyclically_reduce_after_moves) of breadth_first.py:15-97 over the ranks of a
 * communicator: what ac_solver/search/sharded.py drives chunk by chunk through acx_shard_* (five or six calls and two collectives per
 * chunk, from Python), as one C call per rank -- the same chunk loop, two-stream pipeline, lagged control block, replicated small
 * levels, adaptive region capacity and failure protocol, on the same engine (csrc/acx_shard_run.hip).  Every rank calls it with the
 * same arguments and gets the same (solved, path); identical to acx_search(ACX_SEARCH_BFS, ...) on one GPU for every world size.
 *
 * acx_comm: the two collectives the search needs, as plain function pointers (the library links no collective library); both are
 * enqueued on `stream` and return 0 on success.  all_to_all: equal splits of int64 words -- rank d receives words [d k, (d + 1) k) of
 * every rank's d_send, k = words / world, rank r's block at [r k, (r + 1) k) of d_recv.  all_reduce: in place, dtype ACX_I32 / ACX_I64,
 * op ACX_RED_SUM / ACX_RED_MAX.  acx_comm_rccl fills one in for an ncclComm_t of the caller (RCCL over xGMI): librccl is resolved at run
 * time -- the copy the process already runs (torch's), else librccl.so -- e.g. torch.distributed's own communicator,
 * dist.distributed_c10d._get_default_group()._get_backend(torch.device("cuda"))._comm_ptr(), or one made with acx_rccl_comm_create from
 * an id that rank 0 got from acx_rccl_unique_id and broadcast (128 bytes). */
#define ACX_RED_SUM 0
#define ACX_RED_MAX 1
typedef struct acx_comm {
    int32_t rank, world;
    void *ctx;
    int (*all_to_all)(void *ctx, const int64_t *d_send, int64_t *d_recv, int64_t words, void *stream);
    int (*all_reduce)(void *ctx, void *d_buf, int64_t n, int dtype, int op, void *stream);
} acx_comm;
int acx_rccl_available(void); /* 1 when librccl could be resolved */
int acx_comm_rccl(void *nccl_comm, acx_comm *out);
int acx_rccl_unique_id(void *id128);
int acx_rccl_comm_create(const void *id128, int rank, int world, void **nccl_comm); /* ncclCommInitRank on the current device */
int acx_rccl_comm_destroy(void *nccl_comm);

typedef struct acx_shard_opts { /* all zero = the defaults of sharded.py:bfs_sharded */
    int64_t batch_parents;      /* global parents per chunk (default 2^21, 2^23 from 8 ranks on) */
    int64_t replicate_below;    /* levels of fewer parents are processed whole by every rank, no collectives (default 2^18; < 0: every level exchanged) */
    int32_t region_fill;        /* capacity of the exchanged regions in 1/256 of the even share (default: adaptive; ACX 320 = 1.25 x) */
    int32_t overlap;            /* 0: expansion + all-to-all of chunk k + 1 on a side stream beside the dedup of chunk k when world > 1; 1: one stream; 2: side stream always */
    const acx_comm *mask_comm;  /* a second communicator for the per-chunk mask all-reduce (runs beside the all-to-all of the next chunk); NULL: the same */
    int32_t fail_at_call;       /* test hook: the n-th engine call of rank `fail_rank` fails (every rank must then return an error, none may hang) */
    int32_t fail_rank;
    int32_t log_fraction_q8;    /* expected expanded parents / max_nodes in 1/256, sizes the record log (default 128 = 0.5; the log grows by doubling when the estimate is short) */
} acx_shard_opts;
typedef struct acx_shard_run_stats {
    int64_t nodes, expanded, levels, chunks, replicated_levels, local_nodes;
    int64_t all_to_all_calls, all_to_all_bytes, all_reduce_calls, all_reduce_bytes;
    int32_t min_len, reruns;
    double setup_seconds, loop_seconds;
} acx_shard_run_stats;
/* returns ACX_OK (solved / path as acx_search), ACX_E_ROWERR where the reference raises AssertionError (an invalid presentation, a move
 * that empties a relator), ACX_E_CAPACITY when the path needs more than path_cap entries (*path_n says how many), another negative code
 * when a rank failed (every rank returns one).  comm NULL = one rank. */
int acx_bfs_sharded(const int8_t *h_presentation, int L, int64_t max_nodes, int cyclical, const acx_comm *comm, const acx_shard_opts *opts,
                    int32_t *solved, int32_t *path_action, int32_t *path_len, int64_t path_cap, int64_t *path_n, acx_shard_run_stats *stats,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ACX_H */
