/* ac_ball_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's neighbourhood program, barcode_analysis/5_steps_neibourhoods:
 *   AC_UTILS_no_hash.cpp  reduce_ :83   conj0_ :93   inv0_ :103   concat_ :111   sort_ :121   move :144-211
 *                         operator< on relators :19-38 (shorter first, then lexicographic on the integer letters)
 *   neibourhoods.cpp      neibourhood :18-54 (BFS to a given radius over SORTED pairs of freely reduced relators of
 *                         unbounded length; 14 "classic" or 12 "prime" moves; returns the number of distinct pairs)
 * Pinned against the reference itself (oracle/_ref/ball_ref, built from the reference's sources by oracle/Makefile)
 * and against the known answers of its README.txt:38-44 -- tests/test_oracle_golden.py, tests/golden/ball_sizes.json.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXW 4096 /* letters per relator this restatement can hold (radius 5 from 17-letter relators needs < 200) */

typedef struct {
    int n;
    int8_t w[MAXW];
} Rel;

/* free reduction (reduce_, :83-90: reduce0_ repeated to a fixed point = the unique freely reduced form) */
static void reduce(Rel* r) {
    int o = 0;
    for (int i = 0; i < r->n; i++) {
        if (o > 0 && r->w[o - 1] == -r->w[i]) o--;
        else r->w[o++] = r->w[i];
    }
    r->n = o;
}

static void conj0(const Rel* rel, int x, Rel* out) { /* :93-101: x^-1 rel x, reduced */
    out->n = rel->n + 2;
    out->w[0] = (int8_t)(-x);
    memcpy(out->w + 1, rel->w, (size_t)rel->n);
    out->w[out->n - 1] = (int8_t)x;
    reduce(out);
}

static void inv0(const Rel* rel, Rel* out) { /* :103-109 */
    out->n = rel->n;
    for (int i = 0; i < rel->n; i++) out->w[i] = (int8_t)(-rel->w[rel->n - 1 - i]);
}

static void concat(const Rel* a, const Rel* b, Rel* out) { /* :111-119 */
    out->n = a->n + b->n;
    memcpy(out->w, a->w, (size_t)a->n);
    memcpy(out->w + a->n, b->w, (size_t)b->n);
    reduce(out);
}

static int rel_less(const Rel* l, const Rel* r) { /* :19-38 */
    if (l->n != r->n) return l->n < r->n;
    for (int i = 0; i < l->n; i++)
        if (l->w[i] != r->w[i]) return l->w[i] < r->w[i];
    return 0;
}

typedef struct {
    Rel a, b; /* a < b or a == b ... sort_ keeps (first, second) with first < second, else swapped */
} Pres;

static void sort_pair(const Rel* r1, const Rel* r2, Pres* out) { /* :121-137: (r1, r2) if r1 < r2 else (r2, r1) */
    if (rel_less(r1, r2)) {
        out->a = *r1;
        out->b = *r2;
    } else {
        out->a = *r2;
        out->b = *r1;
    }
}

static void move(const Pres* p, int t, int classic, Pres* out) { /* :144-211 (the pair is already sorted) */
    const Rel *r1 = &p->a, *r2 = &p->b;
    static const int8_t A = -1, B = -2, a = 1, b = 2;
    Rel x, y;
    if (classic) {
        switch (t) {
            case 0: concat(r1, r2, &x); sort_pair(&x, r2, out); return;
            case 1: concat(r2, r1, &x); sort_pair(&x, r2, out); return;
            case 2: concat(r1, r2, &x); sort_pair(r1, &x, out); return;
            case 3: concat(r2, r1, &x); sort_pair(r1, &x, out); return;
            case 4: conj0(r2, a, &x); sort_pair(r1, &x, out); return;
            case 5: conj0(r2, b, &x); sort_pair(r1, &x, out); return;
            case 6: conj0(r2, A, &x); sort_pair(r1, &x, out); return;
            case 7: conj0(r2, B, &x); sort_pair(r1, &x, out); return;
            case 8: conj0(r1, a, &x); sort_pair(&x, r2, out); return;
            case 9: conj0(r1, b, &x); sort_pair(&x, r2, out); return;
            case 10: conj0(r1, A, &x); sort_pair(&x, r2, out); return;
            case 11: conj0(r1, B, &x); sort_pair(&x, r2, out); return;
            case 12: inv0(r1, &x); sort_pair(&x, r2, out); return;
            default: inv0(r2, &x); sort_pair(r1, &x, out); return; /* 13 */
        }
    }
    switch (t) {
        case 0: concat(r1, r2, &x); sort_pair(&x, r2, out); return;
        case 1: concat(r2, r1, &x); sort_pair(r1, &x, out); return;
        case 2: inv0(r2, &y); concat(r1, &y, &x); sort_pair(&x, r2, out); return;
        case 3: inv0(r1, &y); concat(r2, &y, &x); sort_pair(r1, &x, out); return;
        case 4: conj0(r1, B, &x); sort_pair(&x, r2, out); return;
        case 5: conj0(r1, A, &x); sort_pair(&x, r2, out); return;
        case 6: conj0(r1, a, &x); sort_pair(&x, r2, out); return;
        case 7: conj0(r1, b, &x); sort_pair(&x, r2, out); return;
        case 8: conj0(r2, B, &x); sort_pair(r1, &x, out); return;
        case 9: conj0(r2, A, &x); sort_pair(r1, &x, out); return;
        case 10: conj0(r2, a, &x); sort_pair(r1, &x, out); return;
        default: conj0(r2, b, &x); sort_pair(r1, &x, out); return; /* 11 */
    }
}

/* visited set: nodes packed as (na, nb, letters of a, letters of b) in a byte pool; open addressing on offsets */
typedef struct {
    uint8_t* pool;
    size_t used, cap;
    int64_t* slots; /* offset into pool, -1 empty */
    size_t nslots, count;
} Set;

static uint64_t hash_bytes(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h ^ (h >> 29);
}

static size_t pack(const Pres* p, uint8_t* buf) {
    buf[0] = (uint8_t)(p->a.n & 0xff);
    buf[1] = (uint8_t)(p->a.n >> 8);
    buf[2] = (uint8_t)(p->b.n & 0xff);
    buf[3] = (uint8_t)(p->b.n >> 8);
    memcpy(buf + 4, p->a.w, (size_t)p->a.n);
    memcpy(buf + 4 + p->a.n, p->b.w, (size_t)p->b.n);
    return 4 + (size_t)p->a.n + (size_t)p->b.n;
}

static void unpack(const uint8_t* buf, Pres* p) {
    p->a.n = buf[0] | (buf[1] << 8);
    p->b.n = buf[2] | (buf[3] << 8);
    memcpy(p->a.w, buf + 4, (size_t)p->a.n);
    memcpy(p->b.w, buf + 4 + p->a.n, (size_t)p->b.n);
}

static void set_grow(Set* s) {
    size_t n2 = s->nslots * 2;
    int64_t* ns = (int64_t*)malloc(n2 * sizeof(int64_t));
    for (size_t i = 0; i < n2; i++) ns[i] = -1;
    for (size_t i = 0; i < s->nslots; i++) {
        if (s->slots[i] < 0) continue;
        const uint8_t* k = s->pool + s->slots[i];
        size_t len = 4 + (size_t)(k[0] | (k[1] << 8)) + (size_t)(k[2] | (k[3] << 8));
        size_t h = hash_bytes(k, len) & (n2 - 1);
        while (ns[h] >= 0) h = (h + 1) & (n2 - 1);
        ns[h] = s->slots[i];
    }
    free(s->slots);
    s->slots = ns;
    s->nslots = n2;
}

/* returns the pool offset of the key if it was inserted, -1 if it was there already */
static int64_t set_insert(Set* s, const uint8_t* key, size_t len) {
    if (2 * (s->count + 1) > s->nslots) set_grow(s);
    size_t h = hash_bytes(key, len) & (s->nslots - 1);
    while (s->slots[h] >= 0) {
        const uint8_t* k = s->pool + s->slots[h];
        size_t kl = 4 + (size_t)(k[0] | (k[1] << 8)) + (size_t)(k[2] | (k[3] << 8));
        if (kl == len && memcmp(k, key, len) == 0) return -1;
        h = (h + 1) & (s->nslots - 1);
    }
    if (s->used + len > s->cap) {
        s->cap = 2 * (s->used + len);
        s->pool = (uint8_t*)realloc(s->pool, s->cap);
    }
    memcpy(s->pool + s->used, key, len);
    s->slots[h] = (int64_t)s->used;
    s->used += len;
    s->count++;
    return s->slots[h];
}

/* neibourhood (neibourhoods.cpp:18-54): number of distinct sorted pairs within `radius` moves of the presentation
 * given as a zero-padded int8 row of 2 * half entries.  *max_len receives the longest relator met (may be NULL).
 * Returns -1 when a relator outgrew MAXW. */
long long ac_ball_size(const int8_t* row, int half, int radius, int classic, int* max_len) {
    Rel r1 = {0, {0}}, r2 = {0, {0}};
    for (int i = 0; i < half; i++)
        if (row[i]) r1.w[r1.n++] = row[i];
    for (int i = 0; i < half; i++)
        if (row[half + i]) r2.w[r2.n++] = row[half + i];
    Pres start;
    sort_pair(&r1, &r2, &start);
    Set s;
    s.cap = 1 << 20;
    s.pool = (uint8_t*)malloc(s.cap);
    s.used = 0;
    s.nslots = 1 << 16;
    s.slots = (int64_t*)malloc(s.nslots * sizeof(int64_t));
    for (size_t i = 0; i < s.nslots; i++) s.slots[i] = -1;
    s.count = 0;
    uint8_t* buf = (uint8_t*)malloc(4 + 2 * MAXW);
    /* FIFO of pool offsets; levels are contiguous, so distances are tracked per level */
    size_t qcap = 1 << 16, qhead = 0, qtail = 0;
    int64_t* queue = (int64_t*)malloc(qcap * sizeof(int64_t));
    queue[qtail++] = set_insert(&s, buf, pack(&start, buf));
    const int nmoves = classic ? 14 : 12;
    int longest = start.a.n > start.b.n ? start.a.n : start.b.n;
    long long result = 0;
    Pres* cur = (Pres*)malloc(sizeof(Pres));
    Pres* child = (Pres*)malloc(sizeof(Pres));
    for (int dist = 0; dist < radius && qhead < qtail; dist++) {
        const size_t level_end = qtail;
        for (; qhead < level_end; qhead++) {
            unpack(s.pool + queue[qhead], cur);
            for (int t = 0; t < nmoves; t++) {
                if (cur->a.n + cur->b.n + 2 > MAXW) {
                    result = -1;
                    goto done;
                }
                move(cur, t, classic, child);
                if (child->a.n > longest) longest = child->a.n;
                if (child->b.n > longest) longest = child->b.n;
                const int64_t off = set_insert(&s, buf, pack(child, buf));
                if (off >= 0) {
                    if (qtail == qcap) {
                        qcap *= 2;
                        queue = (int64_t*)realloc(queue, qcap * sizeof(int64_t));
                    }
                    queue[qtail++] = off;
                }
            }
        }
    }
    result = (long long)s.count;
done:
    if (max_len) *max_len = longest;
    free(cur);
    free(child);
    free(queue);
    free(buf);
    free(s.slots);
    free(s.pool);
    return result;
}

/* ---- simplex data of the graph of presentations of total length <= n (barcode_analysis/simplex_data_generation) --------
 * Restatement of {prime,classic}_moves/ac_bfs.cpp:12-98: breadth-first search from <a, b> over sorted pairs, children of
 * total length > n are ignored; vertices are named in the order they are first met; for every (vertex, move) whose child
 * has total length <= n and a LARGER name one edge (vertex, child) is written, with filtration value max(size, size)
 * (an edge can be written more than once).  The moves are those of AC_UTILS_as_sets.h:298-364 -- the same tables as
 * 5_steps_neibourhoods (prime: 12, classic: 14).
 * node_size[cap_nodes], edges[2 * cap_edges], edge_filt[cap_edges]; returns 0, or -1 when a capacity is too small
 * (n_nodes / n_edges then hold what was needed so far), or -2 when a relator became empty (the reference would crash). */
int ac_simplex_graph(int n, int classic, long long cap_nodes, long long cap_edges, long long* n_nodes, uint8_t* node_size, long long* n_edges,
                     uint32_t* edges, uint8_t* edge_filt) {
    Rel r1 = {1, {1}}, r2 = {1, {2}};
    Pres start;
    sort_pair(&r1, &r2, &start);
    Set s;
    s.cap = 1 << 20;
    s.pool = (uint8_t*)malloc(s.cap);
    s.used = 0;
    s.nslots = 1 << 16;
    s.slots = (int64_t*)malloc(s.nslots * sizeof(int64_t));
    for (size_t i = 0; i < s.nslots; i++) s.slots[i] = -1;
    s.count = 0;
    uint8_t* buf = (uint8_t*)malloc(4 + 2 * MAXW);
    /* names: the k-th inserted key is vertex k; its pool offset is queue[k]; name lookup = position found through a second
       table from pool offset to name (offsets grow with insertion, so a binary search on `queue` does it) */
    size_t qcap = 1 << 16, qtail = 0;
    int64_t* queue = (int64_t*)malloc(qcap * sizeof(int64_t));
    queue[qtail++] = set_insert(&s, buf, pack(&start, buf));
    long long ne = 0;
    int rc = 0;
    if (cap_nodes > 0) node_size[0] = (uint8_t)(start.a.n + start.b.n);
    const int nmoves = classic ? 14 : 12;
    Pres* cur = (Pres*)malloc(sizeof(Pres));
    Pres* child = (Pres*)malloc(sizeof(Pres));
    for (size_t qhead = 0; qhead < qtail && rc == 0; qhead++) {
        unpack(s.pool + queue[qhead], cur);
        const int csize = cur->a.n + cur->b.n;
        for (int t = 0; t < nmoves; t++) {
            move(cur, t, classic, child);
            if (child->a.n == 0 || child->b.n == 0) {
                rc = -2;
                break;
            }
            const int size = child->a.n + child->b.n;
            if (size > n) continue;
            const size_t len = pack(child, buf);
            int64_t off = set_insert(&s, buf, len);
            long long cc;
            if (off >= 0) { /* new vertex */
                if (qtail == qcap) {
                    qcap *= 2;
                    queue = (int64_t*)realloc(queue, qcap * sizeof(int64_t));
                }
                cc = (long long)qtail;
                if (cc < cap_nodes) node_size[cc] = (uint8_t)size;
                else rc = -1;
                queue[qtail++] = off;
            } else { /* known: find its offset, then its name */
                size_t h = hash_bytes(buf, len) & (s.nslots - 1);
                for (;;) {
                    const uint8_t* k = s.pool + s.slots[h];
                    size_t kl = 4 + (size_t)(k[0] | (k[1] << 8)) + (size_t)(k[2] | (k[3] << 8));
                    if (kl == len && memcmp(k, buf, len) == 0) break;
                    h = (h + 1) & (s.nslots - 1);
                }
                const int64_t target = s.slots[h];
                size_t lo = 0, hi = qtail;
                while (lo + 1 < hi) {
                    size_t mid = (lo + hi) / 2;
                    if (queue[mid] <= target) lo = mid;
                    else hi = mid;
                }
                cc = (long long)lo;
            }
            if ((long long)qhead < cc) {
                if (ne < cap_edges) {
                    edges[2 * ne] = (uint32_t)qhead;
                    edges[2 * ne + 1] = (uint32_t)cc;
                    edge_filt[ne] = (uint8_t)(csize > size ? csize : size);
                } else {
                    rc = -1;
                }
                ne++;
            }
        }
    }
    *n_nodes = (long long)qtail;
    *n_edges = ne;
    free(cur);
    free(child);
    free(queue);
    free(buf);
    free(s.slots);
    free(s.pool);
    return rc;
}
