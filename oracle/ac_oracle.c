/*
 * ac_oracle.c -- CPU restatement of the Andrews-Curtis hot path of shehper/AC-Solver.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a plain-C, single-thread,
 * byte-for-byte restatement of the reference's NumPy algorithm.  Only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The product
 * (ac-solver_amd/) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  oracle/tools/make_golden.py imports the Python reference in the
 * build container and writes tests/golden/ (JSON); tests/test_oracle_golden.py checks
 * every function below against those vectors and against the data the reference's own
 * tests hold (tests/test_ac_env.py tables, test_bfs.py / test_gs.py / test_miller_schupp.py
 * paths, data/greedy_search_paths.txt, notebooks/Stable-AK3.ipynb cell 1).
 *
 * Each function cites the reference file:line (under /root/reference) it follows.
 * Letters are int8 in [-127, 127]; 0 is padding.  Error codes mirror the Python
 * exception the reference would raise:
 *     AC_E_ASSERT (-1)  AssertionError      AC_E_INDEX (-2)  IndexError
 *     AC_E_VALUE  (-3)  ValueError (np.pad with a negative width)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AC_OK 0
#define AC_E_ASSERT (-1)
#define AC_E_INDEX (-2)
#define AC_E_VALUE (-3)
#define AC_E_NOMEM (-4)
#define AC_MAXW 512 /* widest relator array handled by the fixed scratch buffers */

/* np.count_nonzero over a[0..n) */
static int count_nonzero(const int8_t *a, int n) {
    int c = 0;
    for (int k = 0; k < n; k++) c += (a[k] != 0);
    return c;
}

/* a[a != 0] -> w, returns len */
static int take_nonzero(const int8_t *a, int n, int8_t *w) {
    int c = 0;
    for (int k = 0; k < n; k++)
        if (a[k] != 0) w[c++] = a[k];
    return c;
}

/* ac_solver/envs/utils.py:13-54  is_array_valid_presentation (array of total length n) */
int ac_is_valid_presentation(const int8_t *a, int n) {
    if (n % 2 != 0) return 0; /* utils.py:33 */
    int L = n / 2;
    int l0 = count_nonzero(a, L), l1 = count_nonzero(a + L, L); /* :37-38 */
    for (int k = l0; k < L; k++) /* :40  array[first_word_length:L] == 0 */
        if (a[k] != 0) return 0;
    for (int k = L + l1; k < n; k++) /* :41 */
        if (a[k] != 0) return 0;
    return l0 > 0 && l1 > 0; /* :44-52 */
}

/* ac_solver/envs/utils.py:57-87  is_presentation_trivial */
int ac_is_trivial(const int8_t *a, int n) {
    if (!ac_is_valid_presentation(a, n)) return 0;
    int L = n / 2;
    if (count_nonzero(a, L) != 1 || count_nonzero(a + L, L) != 1) return 0;
    int u = a[0] < 0 ? -a[0] : a[0], v = a[L] < 0 ? -a[L] : a[L];
    return (u == 1 && v == 2) || (u == 2 && v == 1); /* sorted |nonzeros| == [1, 2], :84-86 */
}

/*
 * ac_solver/envs/utils.py:175-240  simplify_relator.
 * `rel` is an array of `n` int8 (n need not equal L); on success the simplified word is
 * written left-aligned into out[0..*out_n): *out_n == L when padded (zeros to the right),
 * else *out_n == *out_len.  `out` must hold max(n, L) bytes.
 */
int ac_simplify_relator(const int8_t *rel, int n, int L, int cyclical, int padded, int8_t *out,
                        int *out_n, int *out_len) {
    int8_t buf[AC_MAXW];
    if (n > AC_MAXW || L > AC_MAXW) return AC_E_NOMEM;
    memcpy(buf, rel, (size_t)n);
    int alen = n;                       /* len(relator): shrinks with every np.delete */
    int len = count_nonzero(buf, alen); /* :201 */
    for (int k = len; k < alen; k++)    /* :202-205 zeros must sit at the right end */
        if (buf[k] != 0) return AC_E_ASSERT;

    int pos = 0; /* :208-217 free reduction, delete pair and step back one */
    while (pos < len - 1) {
        if (buf[pos] == (int8_t)(-buf[pos + 1])) {
            memmove(buf + pos, buf + pos + 2, (size_t)(alen - pos - 2));
            alen -= 2;
            len -= 2;
            if (pos) pos -= 1;
        } else {
            pos += 1;
        }
    }

    if (cyclical && len > 0) { /* :220-229 strip mutually inverse end letters */
        pos = 0;
        while (buf[pos] == (int8_t)(-buf[len - pos - 1])) pos += 1; /* stops before the middle of a reduced word */
        if (pos) {
            /* delete indices [0,pos) and (len-1-pos, len-1] */
            memmove(buf + len - pos, buf + len, (size_t)(alen - len));        /* tail (zeros) slides left */
            memmove(buf, buf + pos, (size_t)(alen - 2 * pos));                /* drop the head */
            alen -= 2 * pos;
            len -= 2 * pos;
        }
    }

    if (padded) { /* :232-233 np.pad(relator, (0, L - len(relator))) */
        if (L - alen < 0) return AC_E_VALUE;
        memset(buf + alen, 0, (size_t)(L - alen));
        alen = L;
    }
    if (L < len) return AC_E_ASSERT; /* :235-238 */
    memcpy(out, buf, (size_t)alen);
    *out_n = alen;
    *out_len = len;
    return AC_OK;
}

/* ac_solver/envs/utils.py:243-280  simplify_presentation (in place on p[0..2L)) */
int ac_simplify_presentation(int8_t *p, int L, int cyclical, int *lengths) {
    if (!ac_is_valid_presentation(p, 2 * L)) return AC_E_ASSERT; /* :261-263 */
    for (int i = 0; i < 2; i++) {                                /* :267-278 */
        int8_t tmp[AC_MAXW];
        int tn, tl;
        int rc = ac_simplify_relator(p + i * L, L, L, cyclical, 1, tmp, &tn, &tl);
        if (rc != AC_OK) return rc;
        memcpy(p + i * L, tmp, (size_t)L);
        lengths[i] = tl;
    }
    return AC_OK;
}

/* ac_solver/envs/ac_moves.py:4-76  concatenate_relators: r_i <- r_i r_j^{sign} (in place) */
int ac_concatenate_relators(int8_t *p, int L, int i, int j, int sign, int *lengths) {
    if (!((i == 0 || i == 1) && (j == 0 || j == 1) && i == 1 - j)) return AC_E_ASSERT; /* :25-31 */
    if (sign != 1 && sign != -1) return AC_E_ASSERT;                                    /* :33 */
    int8_t r2[AC_MAXW], w1[AC_MAXW], w2[AC_MAXW];
    if (L > AC_MAXW) return AC_E_NOMEM;
    const int8_t *r1 = p + i * L; /* :37 */
    if (sign == 1) {              /* :41-48 r_j, or reversed and negated */
        memcpy(r2, p + j * L, (size_t)L);
    } else {
        for (int k = 0; k < L; k++) r2[k] = (int8_t)(-p[j * L + L - 1 - k]);
    }
    int len1 = take_nonzero(r1, L, w1), len2 = take_nonzero(r2, L, w2); /* :50-54 */
    int acc = 0, m = len1 < len2 ? len1 : len2;                        /* :56-60 junction cancellation */
    while (acc < m && w1[len1 - 1 - acc] == (int8_t)(-w2[acc])) acc += 1;
    int new_size = len1 + len2 - 2 * acc; /* :62 */
    if (new_size <= L) {                  /* :64-74 */
        lengths[i] = new_size;
        memcpy(p + i * L, w1, (size_t)(len1 - acc));
        memcpy(p + i * L + len1 - acc, w2 + acc, (size_t)(len2 - acc));
        memset(p + i * L + new_size, 0, (size_t)(L - new_size));
    }
    return AC_OK;
}

/* ac_solver/envs/ac_moves.py:79-156  conjugate: r_i <- g r_i g^-1, g = sign*j (in place) */
int ac_conjugate(int8_t *p, int L, int i, int j, int sign, int *lengths) {
    if (!((i == 0 || i == 1) && (j == 1 || j == 2))) return AC_E_ASSERT; /* :102-104 */
    if (sign != 1 && sign != -1) return AC_E_ASSERT;                      /* :106 */
    int8_t w[AC_MAXW];
    if (L > AC_MAXW) return AC_E_NOMEM;
    int8_t *r = p + i * L;
    int size = take_nonzero(r, L, w); /* :109-111 */
    int g = sign * j;                 /* :114 */
    if (size == 0) return AC_E_INDEX; /* :119 relator_nonzero[0] on an empty array */
    int sc = (w[0] == -g) ? 1 : 0;    /* :119-120 */
    int ec = (w[size - 1] == g) ? 1 : 0;
    int new_size = size + 2 - 2 * (sc + ec); /* :123 */
    if (new_size <= L) {                     /* :126-154; the tail is only cleared when both ends cancel */
        lengths[i] = new_size;
        memcpy(r + 1 - sc, w + sc, (size_t)(size - sc - ec));
        if (!sc) r[0] = (int8_t)g;
        if (!ec) r[size + 1 - 2 * sc] = (int8_t)(-g);
        if (sc && ec) {
            /* presentation[iL+new : iL+new+2] = 0 ; a NumPy slice clips at the array end (2L) */
            int lo = i * L + new_size, hi = lo + 2;
            if (hi > 2 * L) hi = 2 * L;
            for (int k = lo; k < hi; k++) p[k] = 0;
        }
    }
    return AC_OK;
}

/* ac_solver/envs/ac_moves.py:192-206  decode move_id -> (kind, i, j, sign) */
static void decode_move(int move_id, int *is_conj, int *i, int *j, int *sign) {
    int m = move_id + 1;
    *i = m % 2;
    if (move_id < 4) {
        *is_conj = 0;
        *j = 1 - *i;
        *sign = (((m - *i) / 2) % 2) ? -1 : 1;
    } else {
        *is_conj = 1;
        int jp = ((m - *i) / 2) % 2;
        *sign = (((m - *i - 2 * jp) / 4) % 2) ? -1 : 1;
        *j = jp + 1;
    }
}

/*
 * ac_solver/envs/ac_moves.py:159-231  ACMove.  `in` is not modified; `out` receives the new
 * presentation, `lengths` the recomputed relator lengths (the input lengths are dead).
 */
int ac_move(int move_id, const int8_t *in, int L, int cyclical, int8_t *out, int *lengths) {
    if (move_id < 0 || move_id >= 12) return AC_E_ASSERT; /* :188-190 */
    int is_conj, i, j, sign, rc;
    int scratch[2] = {0, 0};
    decode_move(move_id, &is_conj, &i, &j, &sign);
    memmove(out, in, (size_t)(2 * L)); /* presentation.copy(), :36 / :108 */
    rc = is_conj ? ac_conjugate(out, L, i, j, sign, scratch) : ac_concatenate_relators(out, L, i, j, sign, scratch);
    if (rc != AC_OK) return rc;
    return ac_simplify_presentation(out, L, cyclical, lengths); /* :224-229 */
}

/* batched ACMove over n independent presentations; err[k] = 0 or the negated error code */
int ac_move_batch(const int8_t *in, const uint8_t *action, int64_t n, int L, int cyclical, int8_t *out,
                  int32_t *len_out, uint8_t *err) {
    for (int64_t k = 0; k < n; k++) {
        int lens[2] = {0, 0};
        int rc = ac_move(action[k], in + k * 2 * L, L, cyclical, out + k * 2 * L, lens);
        err[k] = (uint8_t)(-rc);
        if (rc != AC_OK) { /* the reference raised: report the input unchanged */
            memmove(out + k * 2 * L, in + k * 2 * L, (size_t)(2 * L));
            lens[0] = count_nonzero(in + k * 2 * L, L);
            lens[1] = count_nonzero(in + k * 2 * L + L, L);
        }
        len_out[2 * k] = lens[0];
        len_out[2 * k + 1] = lens[1];
    }
    return AC_OK;
}

/*
 * ac_solver/envs/ac_env.py:95-113  ACEnv.step for n independent envs, T steps of an action tape
 * [T, n] (uint8), states updated in place.  reward/done/truncated are written per (t, env) when the
 * pointers are non-NULL.  No autoreset (the reference env keeps stepping after done).
 * count[k] is the env's step counter (ac_env.py:104).
 */
int ac_env_rollout(int8_t *state, int32_t *count, int64_t n, int L, int64_t horizon, const uint8_t *tape, int64_t T,
                   int32_t *reward, uint8_t *done, uint8_t *trunc, uint8_t *err) {
    const int32_t max_reward = (int32_t)(horizon * L * 2); /* ac_env.py:80 */
    int8_t nxt[2 * AC_MAXW];
    for (int64_t t = 0; t < T; t++) {
        for (int64_t k = 0; k < n; k++) {
            int lens[2];
            int8_t *s = state + k * 2 * L;
            int rc = ac_move(tape[t * n + k], s, L, 1, nxt, lens); /* cyclical=True default, ac_env.py:97 */
            if (rc != AC_OK) {
                if (err) err[k] = (uint8_t)(-rc);
                lens[0] = count_nonzero(s, L);
                lens[1] = count_nonzero(s + L, L);
            } else {
                memcpy(s, nxt, (size_t)(2 * L));
            }
            int tot = lens[0] + lens[1];
            int d = (tot == 2);                                            /* :101 */
            if (reward) reward[t * n + k] = d ? max_reward : -tot;         /* :102 */
            if (rc == AC_OK) count[k] += 1;                                /* :104; when ACMove raises, step() never gets here */
            if (done) done[t * n + k] = (uint8_t)d;
            if (trunc) trunc[t * n + k] = (uint8_t)(count[k] >= horizon);  /* :105 */
        }
    }
    return AC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Search: exact visited set (open addressing over the raw 2L-byte states) + node arena.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int L, W;          /* W = 2L bytes per state */
    int64_t n, cap;    /* nodes stored / arena capacity */
    int8_t *states;    /* [cap, W] */
    int64_t *parent;   /* [cap] */
    int8_t *action;    /* [cap] */
    int16_t *tlen;     /* [cap] total length */
    int32_t *depth;    /* [cap] */
    int64_t *slots;    /* hash table of node ids, -1 = empty */
    int64_t nslots;
} arena_t;

static uint64_t hash_bytes(const int8_t *s, int W) {
    uint64_t h = 1469598103934665603ull;
    for (int k = 0; k < W; k++) {
        h ^= (uint8_t)s[k];
        h *= 1099511628211ull;
    }
    return h ^ (h >> 29);
}

static int arena_init(arena_t *a, int L, int64_t cap) {
    memset(a, 0, sizeof(*a));
    a->L = L;
    a->W = 2 * L;
    a->cap = cap;
    a->nslots = 64;
    while (a->nslots < 2 * cap) a->nslots <<= 1;
    a->states = (int8_t *)malloc((size_t)cap * a->W);
    a->parent = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    a->action = (int8_t *)malloc((size_t)cap);
    a->tlen = (int16_t *)malloc((size_t)cap * sizeof(int16_t));
    a->depth = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    a->slots = (int64_t *)malloc((size_t)a->nslots * sizeof(int64_t));
    if (!a->states || !a->parent || !a->action || !a->tlen || !a->depth || !a->slots) return AC_E_NOMEM;
    memset(a->slots, 0xff, (size_t)a->nslots * sizeof(int64_t));
    return AC_OK;
}

static void arena_free(arena_t *a) {
    free(a->states); free(a->parent); free(a->action); free(a->tlen); free(a->depth); free(a->slots);
}

/* returns node id if present, else inserts and returns -(id+1)-... : we return 1 if inserted, 0 if seen */
static int arena_insert(arena_t *a, const int8_t *s, int64_t parent, int action, int tlen, int depth) {
    uint64_t m = (uint64_t)a->nslots - 1, h = hash_bytes(s, a->W) & m;
    while (a->slots[h] >= 0) {
        if (memcmp(a->states + a->slots[h] * a->W, s, (size_t)a->W) == 0) return 0;
        h = (h + 1) & m;
    }
    int64_t id = a->n++;
    a->slots[h] = id;
    memcpy(a->states + id * a->W, s, (size_t)a->W);
    a->parent[id] = parent;
    a->action[id] = (int8_t)action;
    a->tlen[id] = (int16_t)tlen;
    a->depth[id] = depth;
    return 1;
}

/* path of node `id` from the root: [(-1, len0), (a1, len1), ...]; returns entries written (or needed) */
static int64_t write_path(const arena_t *a, int64_t id, int32_t *pa, int32_t *pl, int64_t cap) {
    int64_t d = a->depth[id] + 1, k = d;
    for (int64_t v = id; v >= 0; v = a->parent[v]) {
        k--;
        if (k < cap) {
            pa[k] = a->action[v];
            pl[k] = a->tlen[v];
        }
    }
    return d;
}

typedef struct {
    int64_t nodes;    /* len(tree_nodes) at exit */
    int64_t expanded; /* parents popped */
    int64_t moves;    /* ACMove calls */
    int32_t min_len;  /* smallest total length generated */
} ac_stats_t;

/*
 * ac_solver/search/breadth_first.py:15-97  bfs.
 * Returns AC_OK; *solved / path as the reference returns them: path_n == 0 encodes `None`.
 */
int ac_bfs(const int8_t *presentation, int L, int64_t max_nodes, int cyclical, int32_t *solved, int32_t *path_action,
           int32_t *path_len, int64_t path_cap, int64_t *path_n, ac_stats_t *st) {
    if (!ac_is_valid_presentation(presentation, 2 * L)) return AC_E_ASSERT; /* :36-38 */
    arena_t a;
    int64_t cap = max_nodes + 13;
    if (cap < 32) cap = 32;
    if (arena_init(&a, L, cap) != AC_OK) { arena_free(&a); return AC_E_NOMEM; }
    int tot0 = count_nonzero(presentation, L) + count_nonzero(presentation + L, L); /* :48-52 */
    arena_insert(&a, presentation, -1, -1, tot0, 0);                                  /* :55-58 */
    int min_length = tot0, rc = AC_OK;
    int64_t head = 0, moves = 0;
    int8_t child[2 * AC_MAXW];
    *solved = 0;
    *path_n = 0;
    while (head < a.n) { /* :61 the arena in insertion order IS the FIFO queue */
        int64_t cur = head++;
        for (int act = 0; act < 12; act++) { /* :69 */
            int lens[2];
            rc = ac_move(act, a.states + cur * a.W, L, cyclical, child, lens); /* :70-76 */
            moves++;
            if (rc != AC_OK) goto out;
            int nl = lens[0] + lens[1];
            if (nl < min_length) min_length = nl; /* :79-82 */
            if (nl == 2) {                        /* :84-85 checked before dedup */
                int64_t d = write_path(&a, cur, path_action, path_len, path_cap);
                if (d < path_cap) { path_action[d] = act; path_len[d] = nl; }
                *path_n = d + 1;
                *solved = 1;
                goto out;
            }
            arena_insert(&a, child, cur, act, nl, a.depth[cur] + 1); /* :87-89 */
        }
        if (a.n >= max_nodes) break; /* :91-95 once per parent */
    }
out:
    if (st) { st->nodes = a.n; st->expanded = head; st->moves = moves; st->min_len = min_length; }
    arena_free(&a);
    return rc;
}

/* heap of node ids ordered by (total length, depth, state as a signed tuple): greedy.py:104-113 */
static int node_less(const arena_t *a, int64_t x, int64_t y) {
    if (a->tlen[x] != a->tlen[y]) return a->tlen[x] < a->tlen[y];
    if (a->depth[x] != a->depth[y]) return a->depth[x] < a->depth[y];
    const int8_t *sx = a->states + x * a->W, *sy = a->states + y * a->W;
    for (int k = 0; k < a->W; k++)
        if (sx[k] != sy[k]) return sx[k] < sy[k];
    return 0;
}

static void heap_push(const arena_t *a, int64_t *h, int64_t *n, int64_t id) {
    int64_t k = (*n)++;
    h[k] = id;
    while (k > 0) {
        int64_t p = (k - 1) / 2;
        if (!node_less(a, h[k], h[p])) break;
        int64_t t = h[k]; h[k] = h[p]; h[p] = t;
        k = p;
    }
}

static int64_t heap_pop(const arena_t *a, int64_t *h, int64_t *n) {
    int64_t top = h[0];
    h[0] = h[--(*n)];
    int64_t k = 0;
    for (;;) {
        int64_t l = 2 * k + 1, r = l + 1, m = k;
        if (l < *n && node_less(a, h[l], h[m])) m = l;
        if (r < *n && node_less(a, h[r], h[m])) m = r;
        if (m == k) break;
        int64_t t = h[k]; h[k] = h[m]; h[m] = t;
        k = m;
    }
    return top;
}

/*
 * ac_solver/search/greedy.py:15-121  greedy_search.  Unsolved return is the reference's odd
 * `(False, path_of_last_popped + [(11, last_child_len)])` (greedy.py:121).
 */
int ac_greedy(const int8_t *presentation, int L, int64_t max_nodes, int cyclical, int32_t *solved, int32_t *path_action,
              int32_t *path_len, int64_t path_cap, int64_t *path_n, ac_stats_t *st) {
    arena_t a;
    int64_t cap = max_nodes + 13;
    if (cap < 32) cap = 32;
    if (arena_init(&a, L, cap) != AC_OK) { arena_free(&a); return AC_E_NOMEM; }
    int64_t *heap = (int64_t *)malloc((size_t)cap * sizeof(int64_t)), hn = 0;
    if (!heap) { arena_free(&a); return AC_E_NOMEM; }
    int tot0 = count_nonzero(presentation, L) + count_nonzero(presentation + L, L); /* :47-51 */
    arena_insert(&a, presentation, -1, -1, tot0, 0);                                  /* :54-69 */
    heap_push(&a, heap, &hn, 0);
    int min_length = tot0, rc = AC_OK, last_act = -1, last_len = -1;
    int64_t cur = -1, popped = 0, moves = 0;
    int8_t child[2 * AC_MAXW];
    *solved = 0;
    *path_n = 0;
    while (hn > 0) { /* :71 */
        cur = heap_pop(&a, heap, &hn);
        popped++;
        for (int act = 0; act < 12; act++) { /* :76 */
            int lens[2];
            rc = ac_move(act, a.states + cur * a.W, L, cyclical, child, lens);
            moves++;
            if (rc != AC_OK) goto out;
            int nl = lens[0] + lens[1];
            last_act = act;
            last_len = nl;
            if (nl < min_length) min_length = nl; /* :86-89 */
            if (nl == 2) {                        /* :91-100 */
                int64_t d = write_path(&a, cur, path_action, path_len, path_cap);
                if (d < path_cap) { path_action[d] = act; path_len[d] = nl; }
                *path_n = d + 1;
                *solved = 1;
                goto out;
            }
            int64_t id = a.n;
            if (arena_insert(&a, child, cur, act, nl, a.depth[cur] + 1)) heap_push(&a, heap, &hn, id); /* :102-113 */
        }
        if (a.n >= max_nodes) break; /* :115-119 */
    }
    { /* :121 */
        int64_t d = write_path(&a, cur, path_action, path_len, path_cap);
        if (d < path_cap) { path_action[d] = last_act; path_len[d] = last_len; }
        *path_n = d + 1;
    }
out:
    if (st) { st->nodes = a.n; st->expanded = popped; st->moves = moves; st->min_len = min_length; }
    free(heap);
    arena_free(&a);
    return rc;
}
