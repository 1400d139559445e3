/*
 * finito_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See finito_oracle.h.
 *
 * Plain-C restatement of the reference's search-fmin path.  Parity: PINNED by the reference's own
 * known-answer tests (src/tests.cpp:62-317) -> tests/golden/reference_kat.json, tests/test_oracle_golden.py.
 *
 * Third-party semantics restated here because the dependency is absent from /root/reference (empty,
 * un-pinned submodule): algbio/SBWT plain-matrix SBWT (node set, colex order with $<A<C<G<T, suffix-group
 * edge marking, C array, update_sbwt_interval, search) and sdsl-lite rank_support_v5 / int_vector.  The
 * reference's own restatement of the extend formula is include/common.hh:26-36; the node order is pinned by the
 * 12-node table in src/tests.cpp:110-123.
 */
#define _GNU_SOURCE
#include "finito_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * small containers standing in for sdsl::bit_vector / rank_support_v5 / int_vector<> / int_vector<2>
 * ------------------------------------------------------------------------------------------------ */

typedef struct {
    uint64_t* w;     /* bits */
    int64_t n;       /* number of bits */
    uint64_t* sb;    /* rank_support_v5-shaped directory: per 2048-bit superblock {abs, 5 x 11-bit rel counts} */
} bitvec;

static void bv_init(bitvec* b, int64_t n) {
    b->n = n;
    b->w = (uint64_t*)calloc((size_t)(n / 64 + 2), 8);
    b->sb = NULL;
}
static inline int bv_get(const bitvec* b, int64_t i) { return (int)((b->w[i >> 6] >> (i & 63)) & 1); }
static inline void bv_set(bitvec* b, int64_t i) { b->w[i >> 6] |= 1ULL << (i & 63); }
static void bv_free(bitvec* b) { free(b->w); free(b->sb); b->w = b->sb = NULL; }

/* sdsl::rank_support_v5<>: 2048-bit superblocks, absolute count + relative counts at 384-bit boundaries. */
static void bv_build_rank(bitvec* b) {
    int64_t nsb = b->n / 2048 + 1;
    free(b->sb);
    b->sb = (uint64_t*)calloc((size_t)nsb * 2, 8);
    int64_t nwords = b->n / 64 + 1;
    uint64_t abs = 0;
    for (int64_t s = 0; s < nsb; s++) {
        b->sb[2 * s] = abs;
        uint64_t rel = 0, packed = 0;
        for (int j = 0; j < 32; j++) {
            int64_t wi = s * 32 + j;
            if (j > 0 && j % 6 == 0) packed |= rel << (11 * (j / 6 - 1));
            if (wi < nwords) rel += (uint64_t)__builtin_popcountll(b->w[wi]);
        }
        b->sb[2 * s + 1] = packed;
        abs += rel;
    }
}
/* number of ones in [0, i) */
static inline int64_t bv_rank(const bitvec* b, int64_t i) {
    int64_t s = i >> 11;
    int64_t r = (int64_t)b->sb[2 * s];
    int64_t blk = (i & 2047) / 384;
    if (blk > 0) r += (int64_t)((b->sb[2 * s + 1] >> (11 * (blk - 1))) & 2047);
    int64_t wi = s * 32 + blk * 6;
    int64_t wend = i >> 6;
    for (; wi < wend; wi++) r += __builtin_popcountll(b->w[wi]);
    if (i & 63) r += __builtin_popcountll(b->w[wend] & ((1ULL << (i & 63)) - 1));
    return r;
}

typedef struct {
    uint64_t* w;
    int64_t n;
    int width;
} intvec;   /* sdsl::int_vector<> with run-time width */

static int bits_needed(uint64_t x) { return x == 0 ? 1 : 64 - __builtin_clzll(x); }
static void iv_init(intvec* v, int64_t n, int width) {
    v->n = n; v->width = width;
    v->w = (uint64_t*)calloc((size_t)((n * width) / 64 + 2), 8);
}
static inline uint64_t iv_get(const intvec* v, int64_t i) {
    int64_t bit = i * v->width;
    int64_t wi = bit >> 6; int off = (int)(bit & 63);
    uint64_t x = v->w[wi] >> off;
    if (off + v->width > 64) x |= v->w[wi + 1] << (64 - off);
    return v->width == 64 ? x : (x & ((1ULL << v->width) - 1));
}
static inline void iv_set(intvec* v, int64_t i, uint64_t x) {
    int64_t bit = i * v->width;
    int64_t wi = bit >> 6; int off = (int)(bit & 63);
    uint64_t mask = v->width == 64 ? ~0ULL : ((1ULL << v->width) - 1);
    v->w[wi] = (v->w[wi] & ~(mask << off)) | ((x & mask) << off);
    if (off + v->width > 64) {
        int hi = off + v->width - 64;
        uint64_t m2 = (1ULL << hi) - 1;
        v->w[wi + 1] = (v->w[wi + 1] & ~m2) | ((x & mask) >> (64 - off));
    }
}
static void iv_free(intvec* v) { free(v->w); v->w = NULL; }
/* sdsl int_vector<2>::get_int(bit_offset, nbits) */
static inline uint64_t bits_get_int(const uint64_t* w, int64_t bit, int nbits) {
    int64_t wi = bit >> 6; int off = (int)(bit & 63);
    uint64_t x = w[wi] >> off;
    if (off + nbits > 64) x |= w[wi + 1] << (64 - off);
    return nbits == 64 ? x : (x & ((1ULL << nbits) - 1));
}

/* ------------------------------------------------------------------------------------------------ */

#define FO_LW 12  /* label words: 3 bits per char -> k <= 255 in fo_build */
typedef struct { uint64_t w[FO_LW]; } label_t;

struct fo_index {
    int64_t k, n_nodes, n_kmers;
    int64_t C[4];
    bitvec plane[4];
    intvec lcs;            /* packed to bits(k-1) bits (lcs_basic_parallel_algorithm.hpp:115) */
    bitvec fmin, ustart;
    intvec goff; int64_t n_fmin;
    uint64_t* concat; int64_t total_len;   /* int_vector<2> */
    intvec ends; int64_t n_unitigs;
    label_t* labels;       /* only from fo_build */
};

static int char_idx(char c) {   /* common.hh:50-58 */
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        default: return -1;
    }
}

/* ---- label arithmetic: value = sum code(L[j]) << 3j, code($)=0, A..T = 1..4; integer order == colex order ---- */
static int lab_cmp(const void* a, const void* b) {
    const label_t* x = (const label_t*)a; const label_t* y = (const label_t*)b;
    for (int i = FO_LW - 1; i >= 0; i--) {
        if (x->w[i] < y->w[i]) return -1;
        if (x->w[i] > y->w[i]) return 1;
    }
    return 0;
}
static label_t lab_shl(label_t x, int bits) {
    label_t r; memset(&r, 0, sizeof r);
    int ws = bits / 64, bs = bits % 64;
    for (int i = FO_LW - 1; i >= ws; i--) {
        uint64_t v = x.w[i - ws] << bs;
        if (bs && i - ws - 1 >= 0) v |= x.w[i - ws - 1] >> (64 - bs);
        r.w[i] = v;
    }
    return r;
}
static label_t lab_shr3(label_t x) {
    label_t r;
    for (int i = 0; i < FO_LW; i++) {
        uint64_t v = x.w[i] >> 3;
        if (i + 1 < FO_LW) v |= x.w[i + 1] << 61;
        r.w[i] = v;
    }
    return r;
}
static label_t lab_mask(label_t x, int bits) {
    for (int i = 0; i < FO_LW; i++) {
        int lo = i * 64;
        if (bits <= lo) x.w[i] = 0;
        else if (bits < lo + 64) x.w[i] &= (1ULL << (bits - lo)) - 1;
    }
    return x;
}
static void lab_setchar(label_t* x, int pos, int code) {
    int bit = 3 * pos;
    int wi = bit / 64, off = bit % 64;
    x->w[wi] |= (uint64_t)code << off;
    if (off > 61) x->w[wi + 1] |= (uint64_t)code >> (64 - off);
}
static int lab_getchar(const label_t* x, int pos) {
    int bit = 3 * pos;
    int wi = bit / 64, off = bit % 64;
    uint64_t v = x->w[wi] >> off;
    if (off > 61) v |= x->w[wi + 1] << (64 - off);
    return (int)(v & 7);
}
static int64_t lab_find(const label_t* arr, int64_t n, const label_t* key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        int c = lab_cmp(&arr[mid], key);
        if (c == 0) return mid;
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return -1;
}
static int64_t lab_sort_unique(label_t* a, int64_t n) {
    if (n == 0) return 0;
    qsort(a, (size_t)n, sizeof(label_t), lab_cmp);
    int64_t m = 1;
    for (int64_t i = 1; i < n; i++) if (lab_cmp(&a[i], &a[m - 1]) != 0) a[m++] = a[i];
    return m;
}

/* ---- SBWT extend: reference restates the formula at common.hh:26-36 ---- */
typedef struct { int64_t first, second; } ival;

static inline ival sbwt_extend(const fo_index* x, int c, ival I, fo_counters* ctr) {
    if (I.first == -1) return I;
    if (ctr) {
        ctr->extends++;
        ctr->rank_lines += ((I.first >> 9) == ((I.second + 1) >> 9)) ? 1 : 2;
    }
    ival r;
    r.first = x->C[c] + bv_rank(&x->plane[c], I.first);
    r.second = x->C[c] + bv_rank(&x->plane[c], I.second + 1) - 1;
    if (r.first > r.second) { r.first = r.second = -1; }
    return r;
}

/* plain_matrix_sbwt_t::search(kmer): colex rank of a k-mer or -1 (used at PackedStrings.hh:129) */
static int64_t sbwt_search(const fo_index* x, const char* kmer) {
    ival I = {0, x->n_nodes - 1};
    for (int64_t i = 0; i < x->k; i++) {
        int c = char_idx((char)(kmer[i] & ~32));
        if (c < 0) return -1;
        I = sbwt_extend(x, c, I, NULL);
        if (I.first == -1) return -1;
    }
    return I.first;
}

/* diagnostic: histogram of new_len over the drop_first_char calls that scan (fo_debug_newlen_hist) */
static int64_t g_newlen_hist[256];
void fo_debug_newlen_hist(int64_t* out) { for (int i = 0; i < 256; i++) { out[i] = g_newlen_hist[i]; g_newlen_hist[i] = 0; } }

/* common.hh:38-48 */
static inline ival drop_first_char(const fo_index* x, int64_t new_len, ival I, fo_counters* ctr) {
    if (I.first == -1) return I;
    if (ctr && new_len > 0 && new_len < 256) g_newlen_hist[new_len]++;
    if (new_len <= 0) { ival f = {0, x->n_nodes - 1}; return f; }
    ival r = I;
    /* entries read going down are LCS[dmin..I.first]; going up LCS[I.second+1..umax] */
    int64_t dmin = -1, umax = -1;
    while (r.first > 0) {
        dmin = r.first;
        if ((int64_t)iv_get(&x->lcs, r.first) >= new_len) r.first--; else break;
    }
    while (r.second < x->n_nodes - 1) {
        umax = r.second + 1;
        if ((int64_t)iv_get(&x->lcs, r.second + 1) >= new_len) r.second++; else break;
    }
    if (ctr) {
        ctr->drops++;
        int64_t lines = 0;
        if (dmin >= 0) { ctr->lcs_entries += I.first - dmin + 1; lines += (I.first >> 6) - (dmin >> 6) + 1; }
        if (umax >= 0) {
            ctr->lcs_entries += umax - I.second;
            lines += (umax >> 6) - ((I.second + 1) >> 6) + 1;
            if (dmin >= 0 && (I.first >> 6) == ((I.second + 1) >> 6)) lines--;   /* shared line */
        }
        ctr->lcs_lines += lines;
    }
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * LCS construction: lcs_basic_parallel_algorithm.hpp:52-120 (label propagation, k rounds)
 * ------------------------------------------------------------------------------------------------ */
static uint8_t* lcs_propagation(const fo_index* x) {
    int64_t n = x->n_nodes, k = x->k;
    char* buf1 = (char*)malloc((size_t)n);
    char* buf2 = (char*)malloc((size_t)n);
    /* populate_first_column :29-50 */
    {
        int64_t Cx[5] = {x->C[0], x->C[1], x->C[2], x->C[3], n};
        buf1[0] = '$';
        int64_t last_idx = 1;
        const char* ACGT = "ACGT";
        for (int s = 0; s < 4; s++)
            for (int64_t i = 0; i < Cx[s + 1] - Cx[s]; i++) buf1[last_idx++] = ACGT[s];
        if (last_idx != n) { fprintf(stderr, "oracle: BUG first column %ld %ld\n", (long)last_idx, (long)n); }
    }
    uint8_t* lcs = (uint8_t*)malloc((size_t)n);
    memset(lcs, (int)k, (size_t)n);
    for (int64_t round = 0; round < k; round++) {
        char* last = (round % 2 == 0) ? buf1 : buf2;
        char* prop = (round % 2 == 0) ? buf2 : buf1;
        prop[0] = '$';
        /* lcs_update_thread :21-27 */
        for (int64_t i = 0; i < n; i++)
            if (lcs[i] == k && (i == 0 || last[i] != last[i - 1])) lcs[i] = (uint8_t)round;
        /* lcs_propagate_thread :10-19 */
        char* out[4] = {prop + x->C[0], prop + x->C[1], prop + x->C[2], prop + x->C[3]};
        for (int64_t i = 0; i < n; i++)
            for (int s = 0; s < 4; s++)
                if (bv_get(&x->plane[s], i)) { *(out[s]) = last[i]; out[s]++; }
    }
    free(buf1); free(buf2);
    return lcs;
}

/* ------------------------------------------------------------------------------------------------
 * BoundedDeque.hh:5-75 over tuples {freq, len, colex, end}
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int64_t f, len, colex, end; } tup4;
static inline int tup_gt(tup4 a, tup4 b) {
    if (a.f != b.f) return a.f > b.f;
    if (a.len != b.len) return a.len > b.len;
    if (a.colex != b.colex) return a.colex > b.colex;
    return a.end > b.end;
}
typedef struct { tup4* buf; int64_t size, front_idx, back_idx, n_elements; } bdeque;
static inline int64_t dq_inc(const bdeque* d, int64_t i) { return (int64_t)(((uint64_t)(i + 1)) % (uint64_t)d->size); }
static inline int64_t dq_dec(const bdeque* d, int64_t i) { return (int64_t)(((uint64_t)(i - 1 + d->size)) % (uint64_t)d->size); }
static void dq_init(bdeque* d, int64_t max_size) {
    d->size = max_size;
    d->buf = (tup4*)calloc((size_t)(max_size > 0 ? max_size : 1), sizeof(tup4));
    d->front_idx = max_size - 1; d->back_idx = 0; d->n_elements = 0;
}
static inline tup4 dq_back(const bdeque* d) { return d->buf[dq_dec(d, d->back_idx)]; }
static inline tup4 dq_front(const bdeque* d) { return d->buf[dq_inc(d, d->front_idx)]; }
static inline void dq_push_back(bdeque* d, tup4 x) { d->buf[d->back_idx] = x; d->back_idx = dq_inc(d, d->back_idx); d->n_elements++; }
static inline void dq_pop_front(bdeque* d) { d->front_idx = dq_inc(d, d->front_idx); d->n_elements--; }
static inline void dq_pop_back(bdeque* d) { d->back_idx = dq_dec(d, d->back_idx); d->n_elements--; }
static inline void dq_clear(bdeque* d) { d->n_elements = 0; d->front_idx = d->size - 1; d->back_idx = 0; }

/* ------------------------------------------------------------------------------------------------
 * FinimizerIndexBuilder::add_sequence, FinimizerIndex.hh:321-389
 * ------------------------------------------------------------------------------------------------ */
typedef struct { uint64_t* a; int64_t n, cap; } u64vec;
static void u64vec_push(u64vec* v, uint64_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->a = (uint64_t*)realloc(v->a, (size_t)v->cap * 8); }
    v->a[v->n++] = x;
}
static int u64_cmp(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static void add_sequence(const fo_index* x, const char* seq, int64_t str_len, bitvec* fmin_bv, int64_t* fmin_found,
                         uint64_t* global_offsets, int64_t unitig_start, uint8_t* lastlen, u64vec* fin_set) {
    const int64_t n_nodes = x->n_nodes, k = x->k;
    int64_t freq;
    bdeque all_fmin; dq_init(&all_fmin, str_len);
    tup4 w_fmin = {n_nodes, k + 1, n_nodes, str_len};
    int64_t kmer = 0, start = 0, end;
    ival I = {0, n_nodes - 1};
    int64_t I_start;
    tup4 curr = {0, 0, 0, 0};
    for (end = 0; end < str_len; end++) {
        char c = (char)(seq[end] & ~32);
        I = sbwt_extend(x, char_idx(c), I, NULL);
        freq = I.second - I.first + 1;
        I_start = I.first;
        if (freq == 1) {
            while (freq == 1) {
                curr.f = freq; curr.len = end - start + 1; curr.colex = I_start; curr.end = end;
                start++;
                I = drop_first_char(x, end - start + 1, I, NULL);
                freq = I.second - I.first + 1;
                I_start = I.first;
            }
            if (tup_gt(w_fmin, curr)) { dq_clear(&all_fmin); w_fmin = curr; }
            else { while (tup_gt(dq_back(&all_fmin), curr)) dq_pop_back(&all_fmin); }
            dq_push_back(&all_fmin, curr);
        }
        if (end >= k - 1) {
            /* count_all_w_fmin.insert({len, freq, colex}) :368 -- only its size matters (:301) */
            if (w_fmin.colex >= 0 && w_fmin.colex < n_nodes && lastlen[w_fmin.colex] != (uint8_t)w_fmin.len) {
                lastlen[w_fmin.colex] = (uint8_t)w_fmin.len;
                u64vec_push(fin_set, ((uint64_t)w_fmin.colex << 8) | (uint64_t)w_fmin.len);
            }
            if (fmin_found[w_fmin.colex] == 0 || fmin_found[w_fmin.colex] < w_fmin.end) {
                bv_set(fmin_bv, w_fmin.colex);
                fmin_found[w_fmin.colex] = w_fmin.end;
                global_offsets[w_fmin.colex] = (uint64_t)(unitig_start + w_fmin.end);
            }
            kmer++;
            while (w_fmin.end - w_fmin.len + 1 < kmer) {
                dq_pop_front(&all_fmin);
                if (all_fmin.n_elements == 0) { w_fmin.f = n_nodes; w_fmin.len = k + 1; w_fmin.colex = kmer + 1; w_fmin.end = kmer + k; }
                else w_fmin = dq_front(&all_fmin);
            }
        }
    }
    free(all_fmin.buf);
}

/* ------------------------------------------------------------------------------------------------
 * fo_build
 * ------------------------------------------------------------------------------------------------ */
typedef struct { label_t lab; int64_t id; } firstkmer_t;
static int firstkmer_cmp(const void* a, const void* b) {
    const firstkmer_t* x = (const firstkmer_t*)a; const firstkmer_t* y = (const firstkmer_t*)b;
    int c = lab_cmp(&x->lab, &y->lab);
    if (c) return c;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);   /* std::pair comparison, PackedStrings.hh:121 */
}

static void finalize_dictionaries(fo_index* x) {
    for (int c = 0; c < 4; c++) bv_build_rank(&x->plane[c]);
    bv_build_rank(&x->fmin);
    bv_build_rank(&x->ustart);
}

fo_index* fo_build(const char* bases, const uint64_t* offsets, int64_t n_unitigs, int k) {
    if (k < 1 || 3 * k > 64 * FO_LW || k > 255) return NULL;
    for (int64_t u = 0; u < n_unitigs; u++) {
        if ((int64_t)(offsets[u + 1] - offsets[u]) < k) return NULL;
        for (uint64_t p = offsets[u]; p < offsets[u + 1]; p++) if (char_idx((char)(bases[p] & ~32)) < 0) return NULL;
    }
    fo_index* x = (fo_index*)calloc(1, sizeof(fo_index));
    x->k = k;

    /* 1. all k-mers, colex-sorted, distinct */
    int64_t tot = 0;
    for (int64_t u = 0; u < n_unitigs; u++) tot += (int64_t)(offsets[u + 1] - offsets[u]) - k + 1;
    label_t* kmers = (label_t*)malloc(sizeof(label_t) * (size_t)(tot > 0 ? tot : 1));
    int64_t m = 0;
    for (int64_t u = 0; u < n_unitigs; u++) {
        const char* s = bases + offsets[u];
        int64_t len = (int64_t)(offsets[u + 1] - offsets[u]);
        for (int64_t p = 0; p + k <= len; p++) {
            label_t L; memset(&L, 0, sizeof L);
            for (int j = 0; j < k; j++) lab_setchar(&L, j, 1 + char_idx((char)(s[p + j] & ~32)));
            kmers[m++] = L;
        }
    }
    m = lab_sort_unique(kmers, m);
    x->n_kmers = m;

    /* 2. node set = k-mers + $-padded proper prefixes of every k-mer without a predecessor + root */
    int64_t cap = m + 1024, n = 0;
    label_t* nodes = (label_t*)malloc(sizeof(label_t) * (size_t)cap);
    for (int64_t i = 0; i < m; i++) nodes[n++] = kmers[i];
    {
        label_t root; memset(&root, 0, sizeof root);
        nodes[n++] = root;
    }
    for (int64_t i = 0; i < m; i++) {
        int has_pred = 0;
        label_t shifted = lab_mask(lab_shl(kmers[i], 3), 3 * k);   /* Y[j+1] = X[j], Y[0] = 0 */
        for (int c = 1; c <= 4 && !has_pred; c++) {
            label_t y = shifted; y.w[0] |= (uint64_t)c;
            if (lab_find(kmers, m, &y) >= 0) has_pred = 1;
        }
        if (!has_pred) {
            for (int j = 1; j < k; j++) {   /* j real chars: $^(k-j) X[0..j) */
                if (n == cap) { cap *= 2; nodes = (label_t*)realloc(nodes, sizeof(label_t) * (size_t)cap); }
                nodes[n++] = lab_mask(lab_shl(kmers[i], 3 * (k - j)), 3 * k);
            }
        }
    }
    n = lab_sort_unique(nodes, n);
    x->n_nodes = n;
    x->labels = nodes;

    /* 3. planes: bit (c,i) set iff label_i[1:]+c is a node and i is the first node of its (k-1)-suffix group */
    for (int c = 0; c < 4; c++) bv_init(&x->plane[c], n);
    for (int64_t i = 0; i < n; i++) {
        label_t suf = lab_shr3(nodes[i]);
        if (i > 0) { label_t ps = lab_shr3(nodes[i - 1]); if (lab_cmp(&suf, &ps) == 0) continue; }
        for (int c = 0; c < 4; c++) {
            label_t t = suf; lab_setchar(&t, k - 1, c + 1);
            if (lab_find(nodes, n, &t) >= 0) bv_set(&x->plane[c], i);
        }
    }
    x->C[0] = 1;
    for (int c = 0; c < 3; c++) {
        int64_t pc = 0;
        for (int64_t wi = 0; wi <= n / 64; wi++) pc += __builtin_popcountll(x->plane[c].w[wi]);
        x->C[c + 1] = x->C[c] + pc;
    }
    for (int c = 0; c < 4; c++) bv_build_rank(&x->plane[c]);

    /* 4. LCS */
    uint8_t* lcs_bytes = lcs_propagation(x);
    iv_init(&x->lcs, n, bits_needed((uint64_t)(k - 1)));
    for (int64_t i = 0; i < n; i++) iv_set(&x->lcs, i, lcs_bytes[i]);
    free(lcs_bytes);

    /* 5. permute_unitigs, PackedStrings.hh:105-135 */
    firstkmer_t* fk = (firstkmer_t*)malloc(sizeof(firstkmer_t) * (size_t)(n_unitigs > 0 ? n_unitigs : 1));
    bv_init(&x->ustart, n);
    int64_t total_len = 0;
    for (int64_t u = 0; u < n_unitigs; u++) {
        const char* s = bases + offsets[u];
        label_t L; memset(&L, 0, sizeof L);
        for (int j = 0; j < k; j++) lab_setchar(&L, j, 1 + char_idx((char)(s[j] & ~32)));
        fk[u].lab = L; fk[u].id = u;
        int64_t colex = sbwt_search(x, s);
        if (colex >= 0) bv_set(&x->ustart, colex);
        total_len += (int64_t)(offsets[u + 1] - offsets[u]);
    }
    qsort(fk, (size_t)n_unitigs, sizeof(firstkmer_t), firstkmer_cmp);
    /* PackedStrings ctor :36-64 */
    x->total_len = total_len; x->n_unitigs = n_unitigs;
    x->concat = (uint64_t*)calloc((size_t)(total_len / 32 + 2), 8);
    iv_init(&x->ends, n_unitigs, bits_needed((uint64_t)total_len));
    {
        int64_t i = 0, end = 0;
        for (int64_t r = 0; r < n_unitigs; r++) {
            int64_t u = fk[r].id;
            const char* s = bases + offsets[u];
            int64_t len = (int64_t)(offsets[u + 1] - offsets[u]);
            for (int64_t j = 0; j < len; j++, i++)
                x->concat[i >> 5] |= (uint64_t)char_idx((char)(s[j] & ~32)) << (2 * (i & 31));
            end += len;
            iv_set(&x->ends, r, (uint64_t)end);
        }
    }

    /* 6. FinimizerIndexBuilder ctor :273-319 */
    bv_init(&x->fmin, n);
    int64_t* fmin_found = (int64_t*)calloc((size_t)n, 8);
    uint64_t* global_offsets = (uint64_t*)calloc((size_t)n, 8);
    uint8_t* lastlen = (uint8_t*)calloc((size_t)n, 1);
    u64vec fin_set = {0, 0, 0};
    {
        int64_t tl = 0;
        char* buf = NULL; int64_t bufcap = 0;
        for (int64_t r = 0; r < n_unitigs; r++) {
            int64_t u = fk[r].id;
            int64_t len = (int64_t)(offsets[u + 1] - offsets[u]);
            if (len + 1 > bufcap) { bufcap = len + 1; buf = (char*)realloc(buf, (size_t)bufcap); }
            for (int64_t j = 0; j < len; j++) buf[j] = (char)(bases[offsets[u] + j] & ~32);
            buf[len] = 0;
            add_sequence(x, buf, len, &x->fmin, fmin_found, global_offsets, tl, lastlen, &fin_set);
            tl += len;
        }
        free(buf);
    }
    qsort(fin_set.a, (size_t)fin_set.n, 8, u64_cmp);
    int64_t n_fin = 0;
    for (int64_t i = 0; i < fin_set.n; i++) if (i == 0 || fin_set.a[i] != fin_set.a[i - 1]) n_fin++;
    uint64_t maxoff = 0;
    for (int64_t i = 0; i < n; i++) if (global_offsets[i] > maxoff) maxoff = global_offsets[i];
    x->n_fmin = n_fin;
    iv_init(&x->goff, n_fin, bits_needed(maxoff));
    {
        int64_t idx = 0;
        for (int64_t i = 0; i < n; i++) if (bv_get(&x->fmin, i)) iv_set(&x->goff, idx++, global_offsets[i]);
    }
    free(fin_set.a); free(lastlen); free(global_offsets); free(fmin_found); free(fk); free(kmers);
    finalize_dictionaries(x);
    return x;
}

fo_index* fo_from_components(int k, int64_t n_nodes, const uint64_t* const planes[4], const uint8_t* lcs,
                             const uint64_t* fmin_bits, const uint64_t* ustart_bits, const int64_t* goff,
                             int64_t n_fmin, const uint8_t* concat_codes, int64_t total_len,
                             const int64_t* ends, int64_t n_unitigs) {
    fo_index* x = (fo_index*)calloc(1, sizeof(fo_index));
    x->k = k; x->n_nodes = n_nodes;
    int64_t nw = (n_nodes + 63) / 64;
    for (int c = 0; c < 4; c++) { bv_init(&x->plane[c], n_nodes); memcpy(x->plane[c].w, planes[c], (size_t)nw * 8); }
    x->C[0] = 1;
    for (int c = 0; c < 3; c++) {
        int64_t pc = 0;
        for (int64_t wi = 0; wi < nw; wi++) pc += __builtin_popcountll(x->plane[c].w[wi]);
        x->C[c + 1] = x->C[c] + pc;
    }
    iv_init(&x->lcs, n_nodes, bits_needed((uint64_t)(k - 1)));
    for (int64_t i = 0; i < n_nodes; i++) iv_set(&x->lcs, i, lcs[i]);
    bv_init(&x->fmin, n_nodes); memcpy(x->fmin.w, fmin_bits, (size_t)nw * 8);
    bv_init(&x->ustart, n_nodes); memcpy(x->ustart.w, ustart_bits, (size_t)nw * 8);
    uint64_t maxoff = 0;
    for (int64_t i = 0; i < n_fmin; i++) if ((uint64_t)goff[i] > maxoff) maxoff = (uint64_t)goff[i];
    x->n_fmin = n_fmin;
    iv_init(&x->goff, n_fmin, bits_needed(maxoff));
    for (int64_t i = 0; i < n_fmin; i++) iv_set(&x->goff, i, (uint64_t)goff[i]);
    x->total_len = total_len; x->n_unitigs = n_unitigs;
    x->concat = (uint64_t*)calloc((size_t)(total_len / 32 + 2), 8);
    for (int64_t i = 0; i < total_len; i++) x->concat[i >> 5] |= (uint64_t)(concat_codes[i] & 3) << (2 * (i & 31));
    iv_init(&x->ends, n_unitigs, bits_needed((uint64_t)total_len));
    for (int64_t i = 0; i < n_unitigs; i++) iv_set(&x->ends, i, (uint64_t)ends[i]);
    /* number_of_kmers: nodes whose label has no '$' -- not recoverable from the components alone; the caller
     * does not need it on the query path. */
    x->n_kmers = -1;
    finalize_dictionaries(x);
    return x;
}

void fo_free(fo_index* x) {
    if (!x) return;
    for (int c = 0; c < 4; c++) bv_free(&x->plane[c]);
    iv_free(&x->lcs); bv_free(&x->fmin); bv_free(&x->ustart); iv_free(&x->goff); iv_free(&x->ends);
    free(x->concat); free(x->labels);
    free(x);
}

int64_t fo_k(const fo_index* x) { return x->k; }
int64_t fo_n_nodes(const fo_index* x) { return x->n_nodes; }
int64_t fo_n_kmers(const fo_index* x) { return x->n_kmers; }
int64_t fo_n_unitigs(const fo_index* x) { return x->n_unitigs; }
int64_t fo_n_fmin(const fo_index* x) { return x->n_fmin; }
int64_t fo_total_len(const fo_index* x) { return x->total_len; }
void fo_get_C(const fo_index* x, int64_t out[4]) { for (int c = 0; c < 4; c++) out[c] = x->C[c]; }
void fo_get_plane(const fo_index* x, int c, uint8_t* out) { for (int64_t i = 0; i < x->n_nodes; i++) out[i] = (uint8_t)bv_get(&x->plane[c], i); }
void fo_get_lcs(const fo_index* x, uint8_t* out) { for (int64_t i = 0; i < x->n_nodes; i++) out[i] = (uint8_t)iv_get(&x->lcs, i); }
void fo_get_fmin(const fo_index* x, uint8_t* out) { for (int64_t i = 0; i < x->n_nodes; i++) out[i] = (uint8_t)bv_get(&x->fmin, i); }
void fo_get_ustart(const fo_index* x, uint8_t* out) { for (int64_t i = 0; i < x->n_nodes; i++) out[i] = (uint8_t)bv_get(&x->ustart, i); }
void fo_get_goff(const fo_index* x, int64_t* out) { for (int64_t i = 0; i < x->n_fmin; i++) out[i] = (int64_t)iv_get(&x->goff, i); }
void fo_get_ends(const fo_index* x, int64_t* out) { for (int64_t i = 0; i < x->n_unitigs; i++) out[i] = (int64_t)iv_get(&x->ends, i); }
void fo_get_concat(const fo_index* x, uint8_t* out) { for (int64_t i = 0; i < x->total_len; i++) out[i] = (uint8_t)((x->concat[i >> 5] >> (2 * (i & 31))) & 3); }
int fo_get_label(const fo_index* x, int64_t i, char* out) {
    if (!x->labels) return -1;
    const char* A = "$ACGT";
    for (int j = 0; j < x->k; j++) out[j] = A[lab_getchar(&x->labels[i], j)];
    return 0;
}
/* FinimizerIndex::size_in_bytes :244-258 (payload words; sdsl headers of 8-9 bytes per vector ignored) */
int64_t fo_size_in_bytes(const fo_index* x) {
    int64_t n = x->n_nodes, t = 0;
    t += (n * x->lcs.width + 63) / 64 * 8;
    t += 2 * ((n + 63) / 64 * 8) + 2 * ((n / 2048 + 1) * 16);
    t += (x->n_fmin * x->goff.width + 63) / 64 * 8;
    t += (x->total_len * 2 + 63) / 64 * 8;
    t += (x->n_unitigs * x->ends.width + 63) / 64 * 8;
    t += 4 * ((n + 63) / 64 * 8) + 4 * ((n / 2048 + 1) * 16);
    return t;
}

/* ------------------------------------------------------------------------------------------------
 * rarest_fmin_streaming_search, common.hh:78-186
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int has; int64_t a, b; } optpair;

static void streaming_search(const fo_index* x, const char* input, int64_t str_len, optpair* colex_ranks,
                             optpair* finimizers, optpair* best, fo_counters* ctr) {
    const int64_t n_nodes = x->n_nodes, k = x->k;
    bdeque all_fmin; dq_init(&all_fmin, str_len);
    tup4 w_fmin = {n_nodes, k + 1, n_nodes, str_len + 1};
    int64_t freq, start = 0, end, kmer_start = 0;
    ival I = {0, n_nodes - 1}, I_kmer = {0, n_nodes - 1}, I_new, I_kmer_new;
    int64_t I_start;
    tup4 curr = {0, 0, 0, 0};
    int64_t best_Ustart_first = -1, best_Ustart_second = -1;
    int64_t eager_stale = 0;   /* counters only: number of front entries already outside the window */

    for (end = 0; end < str_len; end++) {
        char c = (char)(input[end] & ~32);
        int ci = char_idx(c);
        if (ctr) ctr->base_strands++;
        if (ci == -1) {
            /* Reference: prints an error and returns empty vectors, which FinimizerIndex::search then indexes
             * out of bounds (UB, common.hh:108-111 / FinimizerIndex.hh:150).  Defined here (and identically in
             * the product) as: the bad base matches nothing -- the same state the reference's own
             * "start > end" reset (:118-122) produces -- so every k-mer overlapping it is (-1,-1). */
            start = end + 1; kmer_start = end + 1;
            I.first = 0; I.second = n_nodes - 1; I_kmer = I;
            continue;
        }
        /* 1) fmin interval */
        I_new = sbwt_extend(x, ci, I, ctr);
        while (I_new.first == -1) {
            kmer_start = ++start;
            if (start > end) {
                I_new.first = 0; I_new.second = n_nodes - 1;
                I_kmer = I_new;
                break;
            }
            I = drop_first_char(x, end - start, I, ctr);
            I_new = sbwt_extend(x, ci, I, ctr);
            I_kmer = I_new;
        }
        I = I_new;
        freq = I.second - I.first + 1;
        I_start = I.first;
        /* (2) k-mer interval */
        if (start != kmer_start) {
            I_kmer_new = sbwt_extend(x, ci, I_kmer, ctr);
            while (I_kmer_new.first == -1) {
                kmer_start++;
                I_kmer = drop_first_char(x, end - kmer_start, I_kmer, ctr);
                I_kmer_new = sbwt_extend(x, ci, I_kmer, ctr);
            }
            I_kmer = I_kmer_new;
        } else {
            I_kmer = I;
        }
        /* (2b) finimizer found */
        if (freq == 1) {
            while (freq == 1) {
                I_start = I.first;
                curr.f = freq; curr.len = end - start + 1; curr.colex = I_start; curr.end = end;
                start++;
                I = drop_first_char(x, end - start + 1, I, ctr);
                freq = I.second - I.first + 1;
            }
            if (tup_gt(w_fmin, curr)) {
                dq_clear(&all_fmin);
                w_fmin = curr;
                eager_stale = 0;
            } else {
                while (tup_gt(dq_back(&all_fmin), curr)) {
                    dq_pop_back(&all_fmin);
                    if (eager_stale > all_fmin.n_elements) eager_stale = all_fmin.n_elements;
                }
            }
            dq_push_back(&all_fmin, curr);
            if (ctr) {
                if (all_fmin.n_elements > ctr->max_deque) ctr->max_deque = all_fmin.n_elements;
                /* live size under eager popping: entries whose start >= kmer_start */
                while (eager_stale < all_fmin.n_elements) {
                    int64_t idx = (int64_t)(((uint64_t)(all_fmin.front_idx + 1 + eager_stale)) % (uint64_t)all_fmin.size);
                    tup4 t = all_fmin.buf[idx];
                    if (t.end - t.len + 1 < kmer_start) eager_stale++; else break;
                }
                if (all_fmin.n_elements - eager_stale > ctr->max_deque_eager) ctr->max_deque_eager = all_fmin.n_elements - eager_stale;
            }
        }
        /* Ustart */
        if (I_kmer.first == I_kmer.second && bv_get(&x->ustart, I_kmer.first) == 1) {
            best_Ustart_first = end; best_Ustart_second = I_kmer.first;
        }
        /* k-mer found */
        if (end - kmer_start + 1 == k) {
            while ((w_fmin.end - w_fmin.len + 1) < kmer_start) {
                dq_pop_front(&all_fmin);
                if (eager_stale > 0) eager_stale--;
                w_fmin = dq_front(&all_fmin);
            }
            int64_t pos = kmer_start + k - 1;
            colex_ranks[pos].has = 1; colex_ranks[pos].a = I_kmer.first;
            finimizers[pos].has = 1; finimizers[pos].a = w_fmin.end; finimizers[pos].b = w_fmin.colex;
            if (best_Ustart_first >= w_fmin.end) { best[pos].has = 1; best[pos].a = best_Ustart_first; best[pos].b = best_Ustart_second; }
            kmer_start++;
            I_kmer = drop_first_char(x, end - kmer_start + 1, I_kmer, ctr);
        }
    }
    free(all_fmin.buf);
}

/* common.hh:61-67 */
static inline int64_t lookup_from_branch_dictionary(const fo_index* x, int64_t kmer_colex) {
    int64_t unitig_rank = bv_rank(&x->ustart, kmer_colex);
    int64_t global_unitig_start = 0;
    if (unitig_rank > 0) global_unitig_start = (int64_t)iv_get(&x->ends, unitig_rank - 1);
    return global_unitig_start + x->k - 1;
}
/* common.hh:69-72 */
static inline int64_t lookup_from_finimizer_dictionary(const fo_index* x, int64_t finimizer_colex) {
    int64_t id = bv_rank(&x->fmin, finimizer_colex);
    return (int64_t)iv_get(&x->goff, id);
}
/* PackedStrings.hh:91-100 (std::upper_bound on ends) */
static inline void global_offset_to_local_offset(const fo_index* x, int64_t g, int64_t* uid, int64_t* off) {
    int64_t lo = 0, hi = x->n_unitigs;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if ((int64_t)iv_get(&x->ends, mid) <= g) lo = mid + 1; else hi = mid;
    }
    int64_t gs = (lo == 0) ? 0 : (int64_t)iv_get(&x->ends, lo - 1);
    *uid = lo; *off = g - gs;
}

typedef struct { int64_t* pairs; int64_t n; int64_t n_found; } qresult;

/* FinimizerIndex.hh:40-45 */
static inline void add_to_query_result(const fo_index* x, int64_t global_kmer_end, qresult* ans) {
    int64_t gks = global_kmer_end - x->k + 1;
    global_offset_to_local_offset(x, gks, &ans->pairs[2 * ans->n], &ans->pairs[2 * ans->n + 1]);
    ans->n++; ans->n_found++;
}

/* FinimizerIndex.hh:47-102.  Differences, both where the reference throws: bases are upper-cased
 * before comparison, and a non-ACGT base ends the walk like a mismatch. */
static void walk_in_unitigs(const fo_index* x, const char* query, int64_t qlen, int64_t global_kmer_end,
                            qresult* ans, int64_t* kmer_end, fo_counters* ctr) {
    int64_t unitig_id = ans->pairs[2 * (ans->n - 1)];
    int64_t u_end = (int64_t)iv_get(&x->ends, unitig_id);
    int64_t max_match = u_end - global_kmer_end - 1;
    if (qlen - *kmer_end - 1 < max_match) max_match = qlen - *kmer_end - 1;
    if (global_kmer_end > u_end || max_match <= 0) return;
    (*kmer_end)++;
    /* sdsl::int_vector<2> query_v(max_match) :56-67 */
    uint64_t* qv = (uint64_t*)calloc((size_t)(max_match / 32 + 2), 8);
    for (int64_t i = 0; i < max_match; i++) {
        int ci = char_idx((char)(query[*kmer_end + i] & ~32));
        if (ci < 0) { max_match = i; break; }
        qv[i >> 5] |= (uint64_t)ci << (2 * (i & 31));
    }
    int64_t word_start = 0;
    int64_t gke_copy = global_kmer_end;
    while (max_match > 0) {
        int word_len = (int)(max_match < 32 ? max_match : 32);
        uint64_t query_word = bits_get_int(qv, word_start, word_len * 2);
        uint64_t unitig_word = bits_get_int(x->concat, word_start + (gke_copy + 1) * 2, word_len * 2);
        uint64_t result = query_word ^ unitig_word;
        if (result) {
            int tz = __builtin_ctzll(result);
            for (int i = 0; i < tz / 2; i++) {
                global_kmer_end++;
                add_to_query_result(x, global_kmer_end, ans);
                (*kmer_end)++;
                if (ctr) ctr->walked++;
            }
            break;
        }
        for (int i = 0; i < word_len; i++) {
            global_kmer_end++;
            add_to_query_result(x, global_kmer_end, ans);
            (*kmer_end)++;
            if (ctr) ctr->walked++;
        }
        max_match -= word_len;
        word_start += word_len * 2;
    }
    (*kmer_end)--;
    free(qv);
}

/* FinimizerIndex::search, FinimizerIndex.hh:119-185 */
int64_t fo_search(const fo_index* x, const char* q, int64_t len, int64_t* pairs_out, int64_t* n_found, fo_counters* ctr) {
    const int64_t k = x->k;
    qresult ans = {pairs_out, 0, 0};
    optpair* colex_ranks = (optpair*)calloc((size_t)(len > 0 ? len : 1), sizeof(optpair));
    optpair* finimizers = (optpair*)calloc((size_t)(len > 0 ? len : 1), sizeof(optpair));
    optpair* rightmost = (optpair*)calloc((size_t)(len > 0 ? len : 1), sizeof(optpair));
    streaming_search(x, q, len, colex_ranks, finimizers, rightmost, ctr);
    for (int64_t kmer_end = k - 1; kmer_end < len; kmer_end++) {
        if (colex_ranks[kmer_end].has) {
            int64_t global_kmer_end;
            int64_t finimizer_end = finimizers[kmer_end].a;
            if (rightmost[kmer_end].has) {
                int64_t p = rightmost[kmer_end].a, colex = rightmost[kmer_end].b;
                global_kmer_end = lookup_from_branch_dictionary(x, colex);
                global_kmer_end += kmer_end - p;
            } else {
                int64_t p = finimizer_end, colex = finimizers[kmer_end].b;
                global_kmer_end = lookup_from_finimizer_dictionary(x, colex);
                global_kmer_end += kmer_end - p;
            }
            if (ctr) ctr->anchors++;
            add_to_query_result(x, global_kmer_end, &ans);
            if (kmer_end + 1 < len) walk_in_unitigs(x, q, len, global_kmer_end, &ans, &kmer_end, ctr);
        } else {
            ans.pairs[2 * ans.n] = -1; ans.pairs[2 * ans.n + 1] = -1; ans.n++;
        }
    }
    free(colex_ranks); free(finimizers); free(rightmost);
    if (n_found) *n_found = ans.n_found;
    return ans.n;
}

/* sbwt::get_rc on a string (external); non-ACGT bases are kept as they are */
static void reverse_complement(const char* q, int64_t len, char* out) {
    for (int64_t i = 0; i < len; i++) {
        char c = (char)(q[len - 1 - i] & ~32);
        switch (c) { case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break; default: break; }
        out[i] = c;
    }
}

/* search_fmin.hh:47-60 */
static int64_t search_merged_buf(const fo_index* x, const char* q, int64_t len, int64_t* pairs_out, int64_t* rev_pairs,
                                 char* rcbuf, fo_counters* ctr, int64_t* positives) {
    const int64_t k = x->k;
    int64_t nf, nr;
    int64_t tot = fo_search(x, q, len, pairs_out, &nf, ctr);
    reverse_complement(q, len, rcbuf);
    fo_search(x, rcbuf, len, rev_pairs, &nr, ctr);
    int64_t pos = 0;
    for (int64_t i = 0; i < tot; i++) {
        if (pairs_out[2 * i] == -1) {
            pairs_out[2 * i] = rev_pairs[2 * (len - k - i)];
            pairs_out[2 * i + 1] = rev_pairs[2 * (len - k - i) + 1];
        }
        if (pairs_out[2 * i] != -1) pos++;
    }
    if (positives) *positives += pos;
    return tot;
}

int64_t fo_search_merged(const fo_index* x, const char* q, int64_t len, int64_t* pairs_out, fo_counters* ctr) {
    int64_t nk = len - x->k + 1; if (nk < 0) nk = 0;
    int64_t* rev = (int64_t*)malloc((size_t)(2 * nk + 2) * 8);
    char* rc = (char*)malloc((size_t)len + 1);
    int64_t pos = 0;
    int64_t n = search_merged_buf(x, q, len, pairs_out, rev, rc, ctr, &pos);
    if (ctr) { ctr->kmers += n; ctr->found += pos; }
    free(rev); free(rc);
    return n;
}

/* out << '(' << unitig << ',' << pos << ')' with ' ' separators and '\n', search_fmin.hh:62-65 */
/* ------------------------------------------------------------------------------------------------
 * The statistics-only modes of build-fmin (build_fmin.hh:95-132 "verify", :134-200 "shortest", :203-214, :257-268):
 * the set of {length, frequency, colex rank} of the window finimizers of the input sequences with frequency threshold t.
 * out[0] = size of the set, out[1] = sum of frequencies, out[2] = sum of lengths (print_finimizer_stats, common.hh:188-206).
 * Returns 0, or -1 if a sequence leaves the index (the reference then loops forever or reads out of bounds).
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int64_t len, f, colex, end; } stup;
static inline int stup_gt(stup a, stup b) {   /* std::tuple order {len, freq, I start, end} */
    if (a.len != b.len) return a.len > b.len;
    if (a.f != b.f) return a.f > b.f;
    if (a.colex != b.colex) return a.colex > b.colex;
    return a.end > b.end;
}
typedef struct { int64_t* a; int64_t n, cap; } tripvec;
static void trip_push(tripvec* v, int64_t len, int64_t f, int64_t colex) {
    if (v->n + 3 > v->cap) { v->cap = v->cap ? v->cap * 2 : 3072; v->a = (int64_t*)realloc(v->a, (size_t)v->cap * 8); }
    v->a[v->n++] = len; v->a[v->n++] = f; v->a[v->n++] = colex;
}
static int trip_cmp(const void* a, const void* b) {
    const int64_t* x = (const int64_t*)a; const int64_t* y = (const int64_t*)b;
    for (int i = 0; i < 3; i++) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}

/* build_shortest_streaming_search, build_fmin.hh:134-200 (BoundedDeque semantics as at BoundedDeque.hh:5-75, stale reads included) */
static int shortest_streaming(const fo_index* x, const char* input, int64_t str_len, int64_t t, uint8_t* fmin_found, tripvec* out) {
    const int64_t n_nodes = x->n_nodes, k = x->k;
    int64_t dsize = str_len > 0 ? str_len : 1;
    stup* buf = (stup*)calloc((size_t)dsize, sizeof(stup));
    int64_t front_idx = dsize - 1, back_idx = 0, n_el = 0;
#define SD_INC(i) ((int64_t)(((uint64_t)((i) + 1)) % (uint64_t)dsize))
#define SD_DEC(i) ((int64_t)(((uint64_t)((i) - 1 + dsize)) % (uint64_t)dsize))
    stup w_fmin = {k + 2, n_nodes, n_nodes, str_len};
    stup curr = {0, 0, 0, 0};
    int64_t kmer = 0, start = 0;
    ival I = {0, n_nodes - 1};
    int rc = 0;
    for (int64_t end = 0; end < str_len; end++) {
        int c = char_idx((char)(input[end] & ~32));
        if (c < 0) { rc = -1; break; }
        I = sbwt_extend(x, c, I, NULL);
        if (I.first == -1) { rc = -1; break; }
        int64_t freq = I.second - I.first + 1, I_start = I.first;
        if (freq <= t) {
            while (freq <= t) {
                curr.len = end - start + 1; curr.f = freq; curr.colex = I_start; curr.end = end;
                start++;
                I = drop_first_char(x, end - start + 1, I, NULL);
                freq = I.second - I.first + 1; I_start = I.first;
            }
            if (stup_gt(w_fmin, curr)) { n_el = 0; front_idx = dsize - 1; back_idx = 0; w_fmin = curr; }
            else { while (stup_gt(buf[SD_DEC(back_idx)], curr)) { back_idx = SD_DEC(back_idx); n_el--; } }
            buf[back_idx] = curr; back_idx = SD_INC(back_idx); n_el++;
        }
        if (end >= k - 1) {
            if (!fmin_found[w_fmin.colex]) {
                trip_push(out, w_fmin.len, w_fmin.f, w_fmin.colex);
                if (w_fmin.end >= k - 1) fmin_found[w_fmin.colex] = 1;
            }
            kmer++;
            while (w_fmin.end - w_fmin.len + 1 < kmer) {
                front_idx = SD_INC(front_idx); n_el--;
                if (n_el == 0) { w_fmin.len = k + 1; w_fmin.f = n_nodes; w_fmin.colex = n_nodes; w_fmin.end = kmer + k; }
                else w_fmin = buf[SD_INC(front_idx)];
            }
        }
    }
#undef SD_INC
#undef SD_DEC
    free(buf);
    return rc;
}

/* verify_shortest_streaming_search, build_fmin.hh:95-132: every substring of every k-window, from scratch */
static int verify_windows(const fo_index* x, const char* input, int64_t str_len, int64_t t, tripvec* out) {
    const int64_t n_nodes = x->n_nodes, k = x->k;
    for (int64_t i = 0; i <= str_len - k; i++) {
        stup w = {k + 1, n_nodes, n_nodes, str_len};
        for (int64_t start = i; start < k + i; start++) {
            ival I = {0, n_nodes - 1};
            for (int64_t end = start; end < k + i; end++) {
                int c = char_idx((char)(input[end] & ~32));
                if (c < 0) return -1;
                I = sbwt_extend(x, c, I, NULL);
                int64_t freq = I.second - I.first + 1;   /* (-1,-1) counts as frequency 1, as in the reference */
                if (freq <= t) {
                    stup nf = {end - start + 1, freq, I.first, end};
                    if (stup_gt(w, nf)) w = nf;
                }
            }
        }
        trip_push(out, w.len, w.f, w.colex);
    }
    return 0;
}

int fo_finimizer_stats(const fo_index* x, const char* bases, const uint64_t* offsets, int64_t n_seqs, int type, int64_t t, int64_t out[3]) {
    tripvec v = {NULL, 0, 0};
    uint8_t* found = (uint8_t*)calloc((size_t)x->n_nodes + 2, 1);   /* (+1: the reference indexes fmin_found[n_nodes] when a window has no candidate) */
    int rc = 0;
    for (int64_t s = 0; s < n_seqs && rc == 0; s++) {
        const char* seq = bases + offsets[s];
        int64_t len = (int64_t)(offsets[s + 1] - offsets[s]);
        if (type == 1) rc = shortest_streaming(x, seq, len, t, found, &v);
        else {   /* remove_ns, build_fmin.hh:216-242: maximal ACGT stretches of length >= k */
            int64_t st = 0;
            for (int64_t i = 0; i <= len && rc == 0; i++) {
                if (i == len || char_idx((char)(seq[i] & ~32)) < 0) {
                    /* (the reference keeps the non-ACGT character at the end of a stretch that precedes it: substr(start, i-start+1);
                     *  verify would then hit it in its last windows -- defined here as: the stretch ends before it) */
                    if (i - st >= x->k) rc = verify_windows(x, seq + st, i - st, t, &v);
                    st = i + 1;
                }
            }
        }
    }
    if (rc == 0) {
        int64_t n = v.n / 3;
        qsort(v.a, (size_t)n, 24, trip_cmp);
        int64_t cnt = 0, sf = 0, sl = 0;
        for (int64_t i = 0; i < n; i++) {
            if (i && trip_cmp(v.a + 3 * i, v.a + 3 * (i - 1)) == 0) continue;
            cnt++; sl += v.a[3 * i]; sf += v.a[3 * i + 1];
        }
        out[0] = cnt; out[1] = sf; out[2] = sl;
    }
    free(v.a); free(found);
    return rc;
}

int64_t fo_format_pairs(const int64_t* pairs, int64_t n_pairs, char* out) {
    char* p = out;
    for (int64_t i = 0; i < n_pairs; i++) {
        if (i > 0) *p++ = ' ';
        p += sprintf(p, "(%ld,%ld)", (long)pairs[2 * i], (long)pairs[2 * i + 1]);
    }
    *p++ = '\n';
    return (int64_t)(p - out);
}

static double now_sec(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double fo_search_batch(const fo_index* x, const char* bases, const uint64_t* offsets, int64_t n_reads,
                       int64_t* pairs_out, int format_text, int n_threads, fo_counters* ctr,
                       uint64_t* text_checksum) {
    const int64_t k = x->k;
    int64_t maxlen = 0;
    for (int64_t r = 0; r < n_reads; r++) { int64_t l = (int64_t)(offsets[r + 1] - offsets[r]); if (l > maxlen) maxlen = l; }
    /* output offsets (in pairs) */
    int64_t* out_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
    out_off[0] = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        int64_t l = (int64_t)(offsets[r + 1] - offsets[r]);
        out_off[r + 1] = out_off[r] + (l >= k ? l - k + 1 : 0);
    }
    if (n_threads < 1) n_threads = 1;
    fo_counters* tctr = (fo_counters*)calloc((size_t)n_threads, sizeof(fo_counters));
    uint64_t* tsum = (uint64_t*)calloc((size_t)n_threads, 8);
    double t0 = now_sec();
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num(); nt = omp_get_num_threads();
#endif
        int64_t nk_max = maxlen - k + 1; if (nk_max < 0) nk_max = 0;
        int64_t* fwd = (int64_t*)malloc((size_t)(2 * nk_max + 2) * 8);
        int64_t* rev = (int64_t*)malloc((size_t)(2 * nk_max + 2) * 8);
        char* rc = (char*)malloc((size_t)maxlen + 1);
        char* text = (char*)malloc((size_t)(nk_max * 44 + 16));
        fo_counters* c = ctr ? &tctr[tid] : NULL;
        int64_t lo = n_reads * tid / nt, hi = n_reads * (tid + 1) / nt;
        for (int64_t r = lo; r < hi; r++) {
            const char* q = bases + offsets[r];
            int64_t len = (int64_t)(offsets[r + 1] - offsets[r]);
            int64_t* dst = pairs_out ? pairs_out + 2 * out_off[r] : fwd;
            int64_t pos = 0;
            int64_t n = search_merged_buf(x, q, len, dst, rev, rc, c, &pos);
            if (c) { c->kmers += n; c->found += pos; }
            if (format_text) {
                int64_t nb = fo_format_pairs(dst, n, text);
                uint64_t h = tsum[tid];
                for (int64_t i = 0; i < nb; i++) h = h * 1099511628211ULL + (uint8_t)text[i];
                tsum[tid] = h;
            }
        }
        free(fwd); free(rev); free(rc); free(text);
    }
    double t1 = now_sec();
    if (ctr) {
        for (int t = 0; t < n_threads; t++) {
            ctr->base_strands += tctr[t].base_strands; ctr->kmers += tctr[t].kmers; ctr->found += tctr[t].found;
            ctr->extends += tctr[t].extends; ctr->rank_lines += tctr[t].rank_lines; ctr->drops += tctr[t].drops;
            ctr->lcs_entries += tctr[t].lcs_entries; ctr->lcs_lines += tctr[t].lcs_lines; ctr->anchors += tctr[t].anchors;
            ctr->walked += tctr[t].walked;
            if (tctr[t].max_deque > ctr->max_deque) ctr->max_deque = tctr[t].max_deque;
            if (tctr[t].max_deque_eager > ctr->max_deque_eager) ctr->max_deque_eager = tctr[t].max_deque_eager;
        }
    }
    if (text_checksum) { uint64_t h = 0; for (int t = 0; t < n_threads; t++) h ^= tsum[t] + 0x9e3779b97f4a7c15ULL * (uint64_t)(t + 1); *text_checksum = h; }
    free(tctr); free(tsum); free(out_off);
    return t1 - t0;
}

/* ---- second restatement: the lazy algorithm of the product's default kernels (same pairs, its own byte count) ---- */
#include "finito_lazy.c"
