/*
 * ipk_oracle.c -- CPU restatement of IPK's phylo-k-mer scoring hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (ipk_amd/) never calls it.
 *
 * PARITY UNPINNED: the reference's hot-path translation units need the un-vendored i2l
 * headers (empty submodule) so they cannot be compiled here without stand-in headers, and
 * the reference's only goldens (tests/data/D652, D140) are git-LFS pointer stubs.  This file
 * therefore restates the algorithm from the reference *source text*; it is cross-checked
 * against an independently written numpy restatement (oracle/np_oracle.py), not against
 * outputs of the reference itself.
 *
 * Each function cites the reference lines (relative to /root/reference) it follows.
 * All score arithmetic is IEEE binary32, one operation per rounding (compile with
 * -ffp-contract=off; no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t key; float score; } pk_t;

/* ---------------------------------------------------------------------------------------
 * bit_length<seq_type>()  (i2l, un-vendored; constants documented in SURVEY.md App. B:
 * DNA: sigma 4 -> 2 bits, AA: sigma 20 -> 5 bits).  Used at ipk/src/pk_compute.cpp:99.
 * ------------------------------------------------------------------------------------- */
unsigned ipko_bits(unsigned sigma)
{
    unsigned b = 0;
    while ((1u << b) < sigma) ++b;
    return b;
}

/* ---------------------------------------------------------------------------------------
 * matrix::preprocess  -- ipk/src/window.cpp:16-27
 * best[0] = 0.0f; best[j+1] = best[j] + max_i m[j][i], accumulated sequentially in float.
 * std::max_element keeps the first largest element under operator<.
 * m is site-major: m[site*sigma + state]  (ipk/include/window.h:23, window.cpp:29-32).
 * ------------------------------------------------------------------------------------- */
void ipko_prefix_max(const float* m, size_t sites, unsigned sigma, float* best)
{
    float product = 0.0f;
    best[0] = 0.0f;
    for (size_t j = 0; j < sites; ++j) {
        const float* col = m + j * sigma;
        float largest = col[0];
        for (unsigned i = 1; i < sigma; ++i)
            if (largest < col[i]) largest = col[i];
        product += largest;
        best[j + 1] = product;
    }
}

/* matrix::range_max_sum -- ipk/src/window.cpp:69-72 (via window::range_max_product :134-137) */
static inline float range_max(const float* best, size_t start, size_t len)
{
    return best[start + len] - best[start];
}

/* ---------------------------------------------------------------------------------------
 * score threshold: db_builder.cpp:640  log_threshold = std::log10(score_threshold(omega, k)).
 * score_threshold lives in i2l (un-vendored); documented formula (omega/sigma)^k
 * (docs/source/usage.rst:224-229).  ASSUMPTION: evaluated in float (score_type) as
 * powf(omega / sigma, (float)k), then log10f.  The engine takes log_eps as an explicit input
 * so this assumption never enters kernel parity.
 * ------------------------------------------------------------------------------------- */
float ipko_score_threshold(float omega, unsigned sigma, unsigned k)
{
    return powf(omega / (float)sigma, (float)k);
}
float ipko_log_threshold(float omega, unsigned sigma, unsigned k)
{
    return log10f(ipko_score_threshold(omega, sigma, k));
}

/* --------------------------------------------------------------------------------------- */
typedef struct { pk_t* v; size_t n, cap; } vec_t;

static void vec_push(vec_t* a, uint32_t key, float score)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 16;
        a->v = (pk_t*)realloc(a->v, a->cap * sizeof(pk_t));
    }
    a->v[a->n].key = key;
    a->v[a->n].score = score;
    a->n++;
}

/* kmer_score_comparator -- pk_compute.cpp:8-11 (descending by score) */
static int cmp_score_desc(const void* a, const void* b)
{
    float x = ((const pk_t*)a)->score, y = ((const pk_t*)b)->score;
    return (x > y) ? -1 : (x < y) ? 1 : 0;
}

typedef struct {
    const float* m; const float* best;
    unsigned sigma, bits; size_t start;
} win_t;

/* as_column -- pk_compute.cpp:14-26: 1-mers of column j with score strictly above eps */
static vec_t as_column(const win_t* w, size_t j, float eps)
{
    vec_t col = {0, 0, 0};
    const float* c = w->m + (w->start + j) * w->sigma;     /* window::get window.cpp:114-117 */
    for (unsigned i = 0; i < w->sigma; ++i)
        if (c[i] > eps) vec_push(&col, i, c[i]);
    return col;
}

/* DCLA::DC -- pk_compute.cpp:42-114 */
static vec_t dc(const win_t* w, size_t j, size_t h, float eps)
{
    if (h == 1) return as_column(w, j, eps);

    vec_t result = {0, 0, 0};
    const size_t hl = h / 2, hr = h - h / 2;
    /* look-ahead bounds, pk_compute.cpp:54-55 */
    const float eps_l = eps - range_max(w->best, w->start + j + hl, hr);
    const float eps_r = eps - range_max(w->best, w->start + j, hl);

    vec_t l = dc(w, j, hl, eps_l);
    vec_t r = dc(w, j + hl, hr, eps_r);

    /* sort whichever side is smaller, pk_compute.cpp:61-70 */
    const int prefix_sort = l.n < r.n;
    vec_t* mn = prefix_sort ? &l : &r;
    vec_t* mx = prefix_sort ? &r : &l;
    const float eps_min = prefix_sort ? eps_l : eps_r;
    const float eps_max = prefix_sort ? eps_r : eps_l;

    if (mn->n != 0) {
        qsort(mn->v, mn->n, sizeof(pk_t), cmp_score_desc);
        for (size_t i = 0; i < mx->n; ++i) {
            const uint32_t a = mx->v[i].key; const float a_score = mx->v[i].score;
            if (a_score < eps_max) break;                   /* :76-79 */
            for (size_t i2 = 0; i2 < mn->n; ++i2) {
                const uint32_t b = mn->v[i2].key; const float b_score = mn->v[i2].score;
                if (b_score < eps_min) break;               /* :85-88 */
                const float score = a_score + b_score;
                if (score <= eps) break;                    /* :90-94 */
                uint32_t kmer;                              /* :96-104 */
                if (prefix_sort) kmer = (b << (hr * w->bits)) | a;
                else             kmer = (a << (hr * w->bits)) | b;
                vec_push(&result, kmer, score);
            }
        }
    }
    free(l.v); free(r.v);
    return result;
}

/* One window: DCLA(window,k).run(eps); get_result()  -- pk_compute.cpp:28-38,116-119.
 * Returns the number of scored phylo-k-mers; writes at most cap of them. */
size_t ipko_window(const float* m, const float* best, size_t sites, unsigned sigma, unsigned k,
                   size_t start, float eps, uint32_t* keys, float* scores, size_t cap)
{
    (void)sites;
    win_t w = { m, best, sigma, ipko_bits(sigma), start };
    vec_t r = dc(&w, 0, k, eps);
    for (size_t i = 0; i < r.n && i < cap; ++i) { keys[i] = r.v[i].key; scores[i] = r.v[i].score; }
    size_t n = r.n;
    free(r.v);
    return n;
}

/* ---------------------------------------------------------------------------------------
 * group_hash_map + ipk::put -- branch_group.h:23, branch_group.cpp:88-101.
 * Open addressing / linear probing stands in for i2l::hash_map (tsl::robin_map upstream,
 * docs/source/install.rst:36); only the (key -> max score, first-wins-on-tie) semantics matter.
 * ------------------------------------------------------------------------------------- */
typedef struct { uint32_t* keys; float* vals; uint8_t* used; size_t cap, n; } map_t;

static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static void map_init(map_t* m, size_t cap)
{
    m->cap = cap; m->n = 0;
    m->keys = (uint32_t*)malloc(cap * sizeof(uint32_t));
    m->vals = (float*)malloc(cap * sizeof(float));
    m->used = (uint8_t*)calloc(cap, 1);
}
static void map_free(map_t* m) { free(m->keys); free(m->vals); free(m->used); }
static void map_put(map_t* m, uint32_t key, float score);
static void map_grow(map_t* m)
{
    map_t big; map_init(&big, m->cap * 2);
    for (size_t i = 0; i < m->cap; ++i)
        if (m->used[i]) map_put(&big, m->keys[i], m->vals[i]);
    map_free(m); *m = big;
}
/* ipk::put -- branch_group.cpp:88-101: replace only if existing < new (strict) */
static void map_put(map_t* m, uint32_t key, float score)
{
    if ((m->n + 1) * 2 > m->cap) map_grow(m);
    size_t i = mix32(key) & (m->cap - 1);
    while (m->used[i]) {
        if (m->keys[i] == key) {
            if (m->vals[i] < score) m->vals[i] = score;
            return;
        }
        i = (i + 1) & (m->cap - 1);
    }
    m->used[i] = 1; m->keys[i] = key; m->vals[i] = score; m->n++;
}

static int cmp_key_asc(const void* a, const void* b)
{
    uint32_t x = ((const pk_t*)a)->key, y = ((const pk_t*)b)->key;
    return (x > y) - (x < y);
}

typedef struct ipko_group { pk_t* v; size_t n; uint64_t emitted; } ipko_group;

/* db_builder::explore_group (RAM mode) -- db_builder.cpp:629-698.
 * mats: n_mats site-major matrices of one branch group (X0 then X1), each [sites][sigma].
 * For every matrix, every window (to_windows, window.cpp:164-182: start = 0 .. sites-k),
 * run DCLA and put() every scored k-mer into group_map; `emitted` is the reference's `count`
 * (:664).  The result is returned sorted by key (the reference iterates in hash order,
 * :685-694; order is irrelevant to the (key -> score) set). */
ipko_group* ipko_explore_group(const float* mats, size_t n_mats, size_t sites, unsigned sigma,
                               unsigned k, float eps)
{
    ipko_group* g = (ipko_group*)calloc(1, sizeof(ipko_group));
    map_t map; map_init(&map, 1024);
    float* best = (float*)malloc((sites + 1) * sizeof(float));
    const unsigned bits = ipko_bits(sigma);
    for (size_t q = 0; q < n_mats; ++q) {
        const float* m = mats + q * sites * sigma;
        ipko_prefix_max(m, sites, sigma, best);          /* read_node -> preprocess, ar.cpp:268 */
        if (sites >= k) {
            for (size_t start = 0; start + k <= sites; ++start) {
                win_t w = { m, best, sigma, bits, start };
                vec_t r = dc(&w, 0, k, eps);
                for (size_t i = 0; i < r.n; ++i) map_put(&map, r.v[i].key, r.v[i].score);
                g->emitted += r.n;
                free(r.v);
            }
        }
    }
    g->n = map.n;
    g->v = (pk_t*)malloc((map.n ? map.n : 1) * sizeof(pk_t));
    size_t o = 0;
    for (size_t i = 0; i < map.cap; ++i)
        if (map.used[i]) { g->v[o].key = map.keys[i]; g->v[o].score = map.vals[i]; ++o; }
    qsort(g->v, g->n, sizeof(pk_t), cmp_key_asc);
    map_free(&map); free(best);
    return g;
}
size_t   ipko_group_size(const ipko_group* g)    { return g->n; }
uint64_t ipko_group_emitted(const ipko_group* g) { return g->emitted; }
void ipko_group_copy(const ipko_group* g, uint32_t* keys, float* scores)
{
    for (size_t i = 0; i < g->n; ++i) { keys[i] = g->v[i].key; scores[i] = g->v[i].score; }
}
void ipko_group_free(ipko_group* g) { if (g) { free(g->v); free(g); } }

/* ipk::kmer_batch -- branch_group.cpp:104-107 */
size_t ipko_kmer_batch(uint32_t key, size_t n_ranges) { return key % n_ranges; }

/* ---------------------------------------------------------------------------------------
 * mif0_filter::calc_filter_values for ONE k-mer -- filter.cpp:20-23,55-119 (double math).
 * log_scores: the k-mer's entries' log10 scores, n of them; N = total_num_groups
 * (= original tree node count, db_builder.cpp:261); threshold = score_threshold(omega,k) as float.
 * Returns the value assigned to kmer_fv.filter_value (stored as float in i2l -- ASSUMPTION, the
 * narrowing happens in un-vendored i2l; we return the double and let the caller narrow).
 * ------------------------------------------------------------------------------------- */
static double logscore_to_score(float log_score)
{
    /* filter.cpp:20-23: std::min(std::pow(10, log_score), 1.0) then narrowed to score_type */
    double s = pow(10.0, (double)log_score);
    if (1.0 < s) s = 1.0;                /* std::min(a, b) = (b < a) ? b : a */
    return (double)(float)s;
}
static double shannon(double x) { return -x * log2(x); }

double ipko_mif0(const float* log_scores, size_t n, size_t N, float threshold)
{
    double score_sum = 0;
    for (size_t i = 0; i < n; ++i) score_sum += logscore_to_score(log_scores[i]);
    score_sum += (double)(N - n) * (double)threshold;
    const double weighted_threshold = (double)threshold / score_sum;
    const double target_threshold = shannon(weighted_threshold);
    double HcBw1 = (double)N * target_threshold;
    for (size_t i = 0; i < n; ++i) {
        const double weighted_score = logscore_to_score(log_scores[i]) / score_sum;
        const double target_value = shannon(weighted_score);
        HcBw1 = HcBw1 - target_threshold + target_value;
    }
    const double Hc = log2((double)N);
    return score_sum * (HcBw1 - Hc);
}

/* ---------------------------------------------------------------------------------------
 * Timing leg for bench.py's cpu_baseline ("port"): explore n_groups groups of mats_per_group
 * matrices each, single thread (the reference is single-threaded, db_builder.cpp:602-606).
 * Returns the total emitted count; unique (key) entries summed over groups in *unique.
 * ------------------------------------------------------------------------------------- */
uint64_t ipko_explore_many(const float* mats, size_t n_groups, size_t mats_per_group, size_t sites,
                           unsigned sigma, unsigned k, float eps, uint64_t* unique)
{
    uint64_t emitted = 0, uniq = 0;
    for (size_t g = 0; g < n_groups; ++g) {
        ipko_group* r = ipko_explore_group(mats + g * mats_per_group * sites * sigma,
                                           mats_per_group, sites, sigma, k, eps);
        emitted += r->emitted; uniq += r->n;
        ipko_group_free(r);
    }
    if (unique) *unique = uniq;
    return emitted;
}

/* Diagnostic: sizes of the two top-level half lists of one window (l and r at pk_compute.cpp:57-58).
 * Used only to size the GPU kernel's list capacities in tests/benchmarks. */
void ipko_window_halves(const float* m, const float* best, unsigned sigma, unsigned k,
                        size_t start, float eps, size_t* nl, size_t* nr)
{
    win_t w = { m, best, sigma, ipko_bits(sigma), start };
    const size_t hl = k / 2, hr = k - k / 2;
    const float eps_l = eps - range_max(best, start + hl, hr);
    const float eps_r = eps - range_max(best, start, hl);
    vec_t l = dc(&w, 0, hl, eps_l);
    vec_t r = dc(&w, hl, hr, eps_r);
    *nl = l.n; *nr = r.n;
    free(l.v); free(r.v);
}

/* log10 in float exactly as the reference applies it (std::log10(float) -> log10f, ar.cpp:257-259). */
void ipko_log10f(const float* in, size_t n, float* out)
{
    for (size_t i = 0; i < n; ++i) out[i] = log10f(in[i]);
}

/* ---------------------------------------------------------------------------------------
 * KEEP_POSITIONS flavour of explore_group (db_builder.cpp:655-662,687-689) with the positions
 * variant of put (branch_group.cpp:73-86): the map value is {score, position}; an existing entry is
 * replaced only when its score is strictly smaller, so on equal scores the first window (matrices in
 * order, windows by ascending start) keeps its position.  position = window.get_position() = start.
 * ------------------------------------------------------------------------------------- */
typedef struct { uint32_t key; float score; uint32_t pos; } pkp_t;
typedef struct ipko_group_pos { pkp_t* v; size_t n; uint64_t emitted; } ipko_group_pos;

static int cmp_keyp_asc(const void* a, const void* b)
{
    uint32_t x = ((const pkp_t*)a)->key, y = ((const pkp_t*)b)->key;
    return (x > y) - (x < y);
}

ipko_group_pos* ipko_explore_group_pos(const float* mats, size_t n_mats, size_t sites, unsigned sigma,
                                       unsigned k, float eps)
{
    ipko_group_pos* g = (ipko_group_pos*)calloc(1, sizeof(ipko_group_pos));
    size_t cap = 1024, n = 0;
    uint32_t* keys = (uint32_t*)malloc(cap * 4); float* vals = (float*)malloc(cap * 4);
    uint32_t* poss = (uint32_t*)malloc(cap * 4); uint8_t* used = (uint8_t*)calloc(cap, 1);
    float* best = (float*)malloc((sites + 1) * sizeof(float));
    const unsigned bits = ipko_bits(sigma);
    for (size_t q = 0; q < n_mats; ++q) {
        const float* m = mats + q * sites * sigma;
        ipko_prefix_max(m, sites, sigma, best);
        for (size_t start = 0; start + k <= sites; ++start) {
            win_t w = { m, best, sigma, bits, start };
            vec_t r = dc(&w, 0, k, eps);
            for (size_t e = 0; e < r.n; ++e) {
                if ((n + 1) * 2 > cap) {                          /* grow + rehash */
                    size_t ncap = cap * 2;
                    uint32_t* nk = (uint32_t*)malloc(ncap * 4); float* nv = (float*)malloc(ncap * 4);
                    uint32_t* np = (uint32_t*)malloc(ncap * 4); uint8_t* nu = (uint8_t*)calloc(ncap, 1);
                    for (size_t i = 0; i < cap; ++i) if (used[i]) {
                        size_t j = mix32(keys[i]) & (ncap - 1);
                        while (nu[j]) j = (j + 1) & (ncap - 1);
                        nu[j] = 1; nk[j] = keys[i]; nv[j] = vals[i]; np[j] = poss[i];
                    }
                    free(keys); free(vals); free(poss); free(used);
                    keys = nk; vals = nv; poss = np; used = nu; cap = ncap;
                }
                size_t i = mix32(r.v[e].key) & (cap - 1);
                while (used[i] && keys[i] != r.v[e].key) i = (i + 1) & (cap - 1);
                if (used[i]) {
                    if (vals[i] < r.v[e].score) { vals[i] = r.v[e].score; poss[i] = (uint32_t)start; }   /* :77-80 */
                } else { used[i] = 1; keys[i] = r.v[e].key; vals[i] = r.v[e].score; poss[i] = (uint32_t)start; ++n; }
            }
            g->emitted += r.n;
            free(r.v);
        }
    }
    g->n = n;
    g->v = (pkp_t*)malloc((n ? n : 1) * sizeof(pkp_t));
    size_t o = 0;
    for (size_t i = 0; i < cap; ++i) if (used[i]) { g->v[o].key = keys[i]; g->v[o].score = vals[i]; g->v[o].pos = poss[i]; ++o; }
    qsort(g->v, g->n, sizeof(pkp_t), cmp_keyp_asc);
    free(keys); free(vals); free(poss); free(used); free(best);
    return g;
}
size_t ipko_group_pos_size(const ipko_group_pos* g) { return g->n; }
uint64_t ipko_group_pos_emitted(const ipko_group_pos* g) { return g->emitted; }
void ipko_group_pos_copy(const ipko_group_pos* g, uint32_t* keys, float* scores, uint32_t* pos)
{
    for (size_t i = 0; i < g->n; ++i) { keys[i] = g->v[i].key; scores[i] = g->v[i].score; pos[i] = g->v[i].pos; }
}
void ipko_group_pos_free(ipko_group_pos* g) { if (g) { free(g->v); free(g); } }
