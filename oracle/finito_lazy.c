/*
 * finito_lazy.c -- CPU ORACLE, second restatement (test infrastructure, NOT the product; compiled into liboracle.so by being
 * included at the end of finito_oracle.c, whose index structures and primitives it uses).
 *
 * The product's default kernels do NOT run the reference algorithm base by base: they prove k-mers absent with short probes,
 * follow a hit along the unitig text, and restart the streaming search a bounded distance before the position where it is
 * needed again (CHANGELOG.md 4.6).  This file states that LAZY algorithm on the CPU, independently of the device code, for two
 * purposes:
 *   (a) a third check of the exactness argument: fo_search_batch_lazy must return the very pairs of the faithful restatement
 *       (fo_search_batch, which follows common.hh:78-186 / FinimizerIndex.hh:119-185 / search_fmin.hh:47-60 line by line) --
 *       tests/test_oracle_lazy.py, bench.py;
 *   (b) the ALGORITHMIC byte count of the algorithm that actually runs (fo_lazy_counters), from which bench.py derives
 *       roofline.achieved.  The model is stated at fo_lazy_counters in finito_oracle.h.
 *
 * What is restated, with the reference lines each piece answers to:
 *   lz_step        one base of rarest_fmin_streaming_search (common.hh:105-184) on explicit state, so that the search can be
 *                  (re)started anywhere in the read
 *   lz_probe       absence proofs: a substring that does not occur in the index rules out every k-mer containing it
 *   lz_strand      FinimizerIndex::search (FinimizerIndex.hh:119-185) for one strand: probes, streaming with verified restarts,
 *                  dictionary anchors (common.hh:61-72, PackedStrings.hh:91-100), walk (FinimizerIndex.hh:47-102)
 *   lz_read        run_fmin_queries_streaming's strand merge (search_fmin.hh:47-60)
 */

typedef struct { int64_t len, colex, end; } lz_cand;

typedef struct {
    const fo_index* x;
    ival I, K;                    /* finimizer-candidate interval q[start..end], k-mer interval q[kstart..end] */
    int64_t start, kstart, end;   /* end = next base to process */
    int64_t bu_end, bu_colex;     /* best_Ustart (common.hh:167) */
    lz_cand* dq; int dq_head, dq_cnt, dq_cap;
    int iskm;
    /* line accounting: distinct node blocks of this base step and of the previous one */
    int64_t cur[64], prev[64]; int ncur, nprev;
    fo_lazy_counters* ctr;
    int tainted;   /* this strand's search used the streaming search, or reported a place that does not spell its k-mer (an unverified anchor): its
                    * pairs prove nothing about the other strand (lz_read: the deferred sister is then searched in full) */
    unsigned char* rcwin;   /* flags bit 5: the index has k-mers whose reverse complement is in it too.  Per window of 64 text positions: 0 not asked
                             * yet, 1 no k-mer that ends in it has its reverse complement in the index, 2 one has (the device's FinDevIndex::rcwin) */
    int64_t stop;   /* a deferred strand (lz_read): the last k-mer end its probes, look-ups and comparisons decide -- a walk goes on past it to the read's end; -1: none */
    int probe_once; int64_t next_t0;   /* lz_probe asks ONE string (a pre-pass look); failed: next_t0 = the first k-mer end not proven absent, -1 none */
    int lean;                          /* flags bit 7: lean tables -- probes are exact occurrences of m-base strings (the directional string filter), no seeds by node */
    const struct lz_cbf_s* cbf;        /* flags bit 6: the canonical string filter of the fast path (built once per fo_search_batch_lazy call) */
} lz_state;

/* string length of the string filters: 20, less for short k -- three strings across a disagreeing base must reach over its k ends: 3 (k-m+1) >= k */
static inline int lz_cbf_m(int64_t k) { int m = (int)(k + 1 - (k + 2) / 3); if (m > 20) m = 20; if (m < 1) m = 1; return m; }

static inline void lz_touch(lz_state* s, int64_t node, int64_t* bucket) {
    if (!s->ctr) return;
    const int64_t b = node >> 6;
    for (int i = 0; i < s->ncur; i++) if (s->cur[i] == b) return;
    if (s->ncur < 64) s->cur[s->ncur++] = b;
    for (int i = 0; i < s->nprev; i++) if (s->prev[i] == b) return;   /* still in registers from the previous step */
    (*bucket)++;
}
static inline void lz_next_step(lz_state* s) {
    memcpy(s->prev, s->cur, (size_t)s->ncur * sizeof(int64_t)); s->nprev = s->ncur; s->ncur = 0;
}

static inline int lz_full(const fo_index* x, ival I) { return I.first == 0 && I.second == x->n_nodes - 1; }

/* update_sbwt_interval (formula at common.hh:26-36); the full interval is answered from the C array */
static inline ival lz_extend(lz_state* s, int c, ival I, int64_t* bucket) {
    const fo_index* x = s->x;
    if (s->ctr && !lz_full(x, I)) { lz_touch(s, I.first, bucket); lz_touch(s, I.second, bucket); }
    ival r;
    r.first = x->C[c] + bv_rank(&x->plane[c], I.first);
    r.second = x->C[c] + bv_rank(&x->plane[c], I.second + 1) - 1;
    if (r.first > r.second) r.first = r.second = -1;
    return r;
}
/* drop_first_char, common.hh:38-48 */
static inline ival lz_drop(lz_state* s, int64_t new_len, ival I, int64_t* bucket) {
    const fo_index* x = s->x;
    if (new_len <= 0) { ival f = {0, x->n_nodes - 1}; return f; }
    ival r = I;
    while (r.first > 0) { lz_touch(s, r.first, bucket); if ((int64_t)iv_get(&x->lcs, r.first) >= new_len) r.first--; else break; }
    while (r.second < x->n_nodes - 1) { lz_touch(s, r.second + 1, bucket); if ((int64_t)iv_get(&x->lcs, r.second + 1) >= new_len) r.second++; else break; }
    return r;
}

static void lz_cold_start(lz_state* s, int64_t c) {
    s->I.first = 0; s->I.second = s->x->n_nodes - 1; s->K = s->I;
    s->start = s->kstart = s->end = c; s->bu_end = -1; s->bu_colex = 0;
    s->dq_head = 0; s->dq_cnt = 0;
}
/* (Re)start of the streaming search at c with a JUMP: the state the search has after streaming the J bases q[c..c+J-1] from a cold
 * start is a function of their SBWT interval alone whenever that interval holds at least two nodes -- every prefix of the string
 * then occurs at least twice as well, so no candidate was pushed (the shrink loop only runs on a single node), start and kmer_start
 * are still c, the k-mer interval equals the finimizer interval (common.hh:142), no Ustart record was taken (it needs a single
 * node, :167) and no k-mer ended (J < k).  One lookup in a table of all J-base strings' intervals replaces J streamed bases.
 * Only positions before silent_until may be jumped over (nothing is reported there).  Falls back to the plain cold start when the
 * string has a non-ACGT base, does not occur, or is unique. */
static void lz_restart(lz_state* s, const char* q, int64_t c, int64_t silent_until, int J) {
    const fo_index* x = s->x;
    s->tainted = 1;
    lz_cold_start(s, c);
    if (J <= 0 || J >= x->k || c + J - 1 >= silent_until) return;
    ival I = {0, x->n_nodes - 1};
    for (int i = 0; i < J; i++) {
        const int ci = char_idx((char)(q[c + i] & ~32));
        if (ci < 0) return;
        ival r;
        r.first = x->C[ci] + bv_rank(&x->plane[ci], I.first);
        r.second = x->C[ci] + bv_rank(&x->plane[ci], I.second + 1) - 1;
        if (r.first > r.second) { if (s->ctr) s->ctr->jump_entries++; return; }
        I = r;
    }
    if (s->ctr) s->ctr->jump_entries++;
    if (I.second <= I.first) return;   /* a single node: candidates may have been pushed on the way */
    s->I = I; s->K = I; s->end = c + J;
    if (s->ctr) s->ctr->jumped_bases += J;
}

#define LZ_DQ(s, i) ((s)->dq[((s)->dq_head + (i)) % (s)->dq_cap])
/* tuple order (freq = 1, len, colex, end) of the reference (common.hh:155-163); the end never decides: the candidate being
 * inserted always has the largest one */
static inline int lz_gt(lz_cand a, lz_cand b) { return a.len != b.len ? a.len > b.len : a.colex > b.colex; }

/* One base of rarest_fmin_streaming_search (common.hh:105-184) at position s->end.  Afterwards s->iskm says whether a k-mer
 * ends here; its finimizer is then the front of the deque.  Candidates that start before the k-mer window are dropped as soon
 * as the window has moved (the reference drops them when it next reads the front, :173-176 -- same live part, CHANGELOG.md 4.3).
 * Does NOT advance s->end. */
static void lz_step(lz_state* s, const char* q) {
    const fo_index* x = s->x;
    const int64_t k = x->k, n = x->n_nodes, end = s->end;
    fo_lazy_counters* c = s->ctr;
    int64_t dummy = 0, *bk = c ? &c->stream_lines : &dummy;
    lz_next_step(s);
    if (c) c->stream_steps++;
    const int ci = char_idx((char)(q[end] & ~32));
    s->iskm = 0;
    if (ci < 0) {   /* defined behaviour for a non-ACGT base: the state of the reference's own `start > end` reset (:118-122) */
        s->start = s->kstart = end + 1; s->I.first = 0; s->I.second = n - 1; s->K = s->I; s->dq_cnt = 0;
        return;
    }
    /* (1) finimizer interval, :114-127 */
    ival In = lz_extend(s, ci, s->I, bk);
    while (In.first == -1) {
        s->kstart = ++s->start;
        if (s->start > end) { In.first = 0; In.second = n - 1; s->K = In; break; }
        s->I = lz_drop(s, end - s->start, s->I, bk);
        In = lz_extend(s, ci, s->I, bk);
        s->K = In;
    }
    s->I = In;
    /* (2) k-mer interval, :132-143 */
    if (s->start != s->kstart) {
        ival Kn = lz_extend(s, ci, s->K, bk);
        while (Kn.first == -1) {
            s->kstart++;
            s->K = lz_drop(s, end - s->kstart, s->K, bk);
            Kn = lz_extend(s, ci, s->K, bk);
        }
        s->K = Kn;
    } else s->K = s->I;
    s->iskm = end - s->kstart + 1 == k;
    while (s->dq_cnt && LZ_DQ(s, 0).end - LZ_DQ(s, 0).len + 1 < s->kstart) { s->dq_head = (s->dq_head + 1) % s->dq_cap; s->dq_cnt--; }
    /* Ustart probe, :167 */
    if (s->K.first == s->K.second) {
        lz_touch(s, s->K.first, bk);
        if (bv_get(&x->ustart, s->K.first)) { s->bu_end = end; s->bu_colex = s->K.first; }
    }
    /* (2b) shortest unique suffix, :145-164 */
    if (s->I.first == s->I.second) {
        lz_cand cur = {0, 0, 0};
        while (s->I.first == s->I.second) {
            cur.len = end - s->start + 1; cur.colex = s->I.first; cur.end = end;
            s->start++;
            s->I = lz_drop(s, end - s->start + 1, s->I, bk);
        }
        if (s->dq_cnt && lz_gt(LZ_DQ(s, 0), cur)) s->dq_cnt = 0;
        else while (s->dq_cnt && lz_gt(LZ_DQ(s, s->dq_cnt - 1), cur)) s->dq_cnt--;
        LZ_DQ(s, s->dq_cnt) = cur; s->dq_cnt++;
    }
    /* k-mer present: the window moves on, :180-181 (its finimizer is read by the caller, :170-179) */
    if (s->iskm) {
        s->kstart++;
        s->K = lz_drop(s, end - s->kstart + 1, s->K, bk);
    }
}

/* read chunks (32 bases, 16 bytes) a kernel loads: a two-entry cache like the kernels' current/next chunk registers */
typedef struct { int64_t c0, c1; } lz_chunks;
static inline void lz_chunk(lz_chunks* cc, int64_t pos, int64_t* bucket) {
    const int64_t ci = pos >> 5;
    if (ci == cc->c0 || ci == cc->c1) return;
    cc->c1 = cc->c0; cc->c0 = ci; (*bucket)++;
}

/* Absence proofs from k-mer end t0 on.  The string q[p..t0], p = t0-PM+1, is looked up: its first T bases in the prefix table
 * (every T-base string's SBWT interval), the rest by extends.  A failure (or a non-ACGT base) inside it rules out every k-mer
 * that contains the failing prefix, i.e. all k-mer ends up to p+k-1; the next probe asks about t0 = p+k.  Returns the first t0
 * whose probe passed (nothing is known about it), or -1 when every k-mer end from t0 on is proven absent. */
/* does the string q[p..p+n-1] (ACGT only) occur in a unitig?  (the device asks its absence filter: a bit per string of F bases) */
static int lz_occurs(const fo_index* x, const char* q, int64_t p, int n) {
    ival I = {0, x->n_nodes - 1};
    for (int i = 0; i < n; i++) {
        const int ci = char_idx((char)(q[p + i] & ~32));
        ival r;
        r.first = x->C[ci] + bv_rank(&x->plane[ci], I.first);
        r.second = x->C[ci] + bv_rank(&x->plane[ci], I.second + 1) - 1;
        if (r.first > r.second) return 0;
        I = r;
    }
    /* nodes that end with the string: a k-mer of the text, or a dummy node -- a dummy holds a PREFIX of a unitig, so the string occurs
     * in the text either way */
    return 1;
}

/* F > 0 (the pre-pass): before a probe at t0 the ABSENCE FILTER is asked about the two strings of F bases that end at t0 and at t0-1;
 * one that occurs in no unitig rules out every k-mer that contains it (ends t0..t0+k-F, or t0-1..t0+k-F-1). */
static int64_t lz_probe(lz_state* s, const char* q, int64_t len, int64_t t0, int T, int PM, lz_chunks* cc, int64_t* chunk_bucket,
                        int64_t* entries, int64_t* extends, int64_t* lines, int64_t* node, int F, int64_t* filter_checks) {
    const fo_index* x = s->x;
    const int64_t k = x->k;
    int64_t d_e = 0, d_x = 0, d_l = 0;
    if (s->lean) { entries = &d_e; extends = &d_x; lines = &d_l; F = 0; }   /* lean tables: a string costs one block of the directional filter, whatever T (0) */
    for (;;) {
        if (s->lean && s->ctr) { s->ctr->fbf_lookups++; if (s->probe_once) s->ctr->prepass_fbf++; }
        if (F > 0) {
            int ok = 1;
            for (int i = 0; i <= F; i++) if (char_idx((char)(q[t0 - F + i] & ~32)) < 0) ok = 0;
            if (ok) {
                lz_chunk(cc, t0 - F, chunk_bucket); lz_chunk(cc, t0, chunk_bucket);
                (*filter_checks)++;
                int64_t adv = 0;
                if (!lz_occurs(x, q, t0 - F + 1, F)) adv = k - F + 1;
                else if (!lz_occurs(x, q, t0 - F, F)) adv = k - F;
                if (adv) { t0 += adv; if (s->probe_once) { s->next_t0 = t0 < len ? t0 : -1; return -1; } if (t0 >= len) return -1; continue; }
            }
        }
        const int64_t p = t0 - PM + 1;
        lz_chunk(cc, p, chunk_bucket); lz_chunk(cc, t0, chunk_bucket);
        int fail = 0;
        ival I = {0, x->n_nodes - 1};
        int off = 0;
        if (T > 0) {
            for (; off < T; off++) if (char_idx((char)(q[p + off] & ~32)) < 0) { fail = 1; break; }
            if (!fail) {
                (*entries)++;   /* one table entry */
                for (int i = 0; i < T && !fail; i++) {
                    const int ci = char_idx((char)(q[p + i] & ~32));
                    ival r;
                    r.first = x->C[ci] + bv_rank(&x->plane[ci], I.first);
                    r.second = x->C[ci] + bv_rank(&x->plane[ci], I.second + 1) - 1;
                    if (r.first > r.second) fail = 1;
                    I = r;
                }
                off = T;
            }
        }
        for (; !fail && off < PM; off++) {
            const int ci = char_idx((char)(q[p + off] & ~32));
            if (ci < 0) { fail = 1; break; }
            lz_next_step(s);
            (*extends)++;
            I = lz_extend(s, ci, I, lines);
            if (I.first == -1) fail = 1;
        }
        if (!fail) { if (node) *node = (I.first == I.second && !s->lean) ? I.first : -1; return t0; }   /* (node: the string is the suffix of this node only; lean tables: a filter names no node) */
        t0 = p + k;
        if (s->probe_once) { s->next_t0 = t0 < len ? t0 : -1; return -1; }
        if (t0 >= len) return -1;
    }
}

/* Absence proofs ACROSS a known bad position E (a read base that disagrees with the unitig text, or a non-ACGT base): every k-mer
 * that contains E ends in [E, E+k-1].  Like lz_probe from k-mer end *t0 on, but every probe string is placed so that it contains
 * E (p = min(t0-PM+1, E)): a string with a wrong base in it almost never occurs in the index.  Returns 1 when all ends up to
 * E+k-1 are proven absent (*t0 >= E+k, or the read is over), 0 when a probe passed: nothing is known about end *t0 (*node: the
 * one node the string ends, or -1 if several; *last_out: where the string ends -- at *t0: a seed; short of it: a guess). */
static int lz_bridge(lz_state* s, const char* q, int64_t len, int64_t* t0, int64_t E, int T, int PM, lz_chunks* cc, int64_t* chunk_bucket,
                     int64_t* entries, int64_t* extends, int64_t* lines, int64_t* node, int64_t* last_out) {
    const fo_index* x = s->x;
    const int64_t k = x->k;
    int tried = 0;   /* the short string of this *t0 occurred: the full-length one is asked */
    int64_t d_e = 0, d_x = 0, d_l = 0;
    if (s->lean) { entries = &d_e; extends = &d_x; lines = &d_l; }
    while (*t0 <= E + k - 1 && *t0 < len) {
        if (s->lean && s->ctr) s->ctr->fbf_lookups++;
        int64_t p = *t0 - PM + 1; if (p > E) p = E;
        /* the first string asked starts T-1 bases before E at the earliest, so that E lies inside the prefix-table key: one table entry
         * settles it.  (A full-length string that ends at *t0 <= E+3 has E behind its key.)  Only if that short string occurs ... */
        /* ... and it starts AT E as soon as a table key fits between E and *t0: a failure then settles every end up to E+k-1 */
        int is_short = 0;
        if (!tried && T > 0) {
            if (k <= 32 && *t0 >= E + T - 1) { is_short = p < E; p = E; }
            else if (p < E - (T - 1)) { is_short = 1; p = E - (T - 1); }
        }
        int64_t last = *t0 < p + 31 ? *t0 : p + 31;   /* the string q[p..last]: it goes on to t0 as long as it matches, 32 bases at most */
        if (s->lean) last = p + PM - 1;                /* lean tables: the filter holds strings of exactly m = PM bases (p <= t0 - PM + 1: it fits) */
        const int n = (int)(last - p + 1);
        lz_chunk(cc, p, chunk_bucket); lz_chunk(cc, last, chunk_bucket);
        int fail = 0, off = 0;
        ival I = {0, x->n_nodes - 1};
        if (T > 0 && n >= T) {
            for (; off < T; off++) if (char_idx((char)(q[p + off] & ~32)) < 0) { fail = 1; break; }
            if (!fail) {
                (*entries)++;
                for (int i = 0; i < T && !fail; i++) {
                    const int ci = char_idx((char)(q[p + i] & ~32));
                    ival r;
                    r.first = x->C[ci] + bv_rank(&x->plane[ci], I.first);
                    r.second = x->C[ci] + bv_rank(&x->plane[ci], I.second + 1) - 1;
                    if (r.first > r.second) fail = 1;
                    I = r;
                }
                off = T;
            }
        }
        for (; !fail && off < n; off++) {
            const int ci = char_idx((char)(q[p + off] & ~32));
            if (ci < 0) { fail = 1; break; }
            lz_next_step(s);
            (*extends)++;
            I = lz_extend(s, ci, I, lines);
            if (I.first == -1) fail = 1;
        }
        if (!fail && is_short) { tried = 1; continue; }   /* ... is the full-length one asked */
        if (!fail) { if (node) *node = (I.first == I.second && !s->lean) ? I.first : -1; if (last_out) *last_out = last; return 0; }
        *t0 = p + k; tried = 0;
    }
    return 1;
}

/* THE REFERENCE'S ANCHOR ANSWER IS A FUNCTION OF THE K-MER (round 3; CHANGELOG.md 4.8).  When a present k-mer Q is not reached by a walk,
 * FinimizerIndex::search reports a place computed from the streaming state (FinimizerIndex.hh:148-174).  That place depends on Q alone:
 * the finimizer is the least candidate that starts inside Q's window, and such a candidate -- the shortest unique suffix ending at a
 * position of the window, recorded iff the longest repeated suffix one position earlier was shorter -- is decided by Q's own bases;
 * and for every window position p at or behind the finimizer's end the k-mer interval's string contains a unique string, so that interval
 * is ONE node whatever precedes Q in the read, and the branch record (common.hh:167) taken there does not depend on the history either.
 * So G(v), the answer for the k-mer of node v, can be had by handing the k-mer alone to the faithful search -- on ANY index, disjoint or
 * not.  What a non-disjoint index changes is that G(v) need not be a place where the text spells v's k-mer (the reference does not
 * check), and that a k-mer found in the text at g need not be reported there (it is reported at G).  Both are checked here, per k-mer:
 *   lz_node_pos(v)   = G(v), and whether the text there spells v's label inside one unitig ("verified": a seed may compare the read with
 *                      the text there; an unverified answer is only used once the k-mer's presence is known from a look-up of the whole k-mer);
 *   lz_text_safe(g)  = the k-mer that the text spells at g is reported at g, i.e. G(its node) == g.
 * The device keeps both as tables built at upload by streaming the unitig text through the plain search (FinDevIndex::pos, ::safe). */
static int64_t lz_kmer_answer(const fo_index* x, const char* lab) {   /* G: offset of the k-mer's last base in the concatenation, -1: not found */
    const int64_t k = x->k;
    int64_t pair[2] = {-1, -1}, nf = 0;
    fo_search(x, lab, k, pair, &nf, NULL);
    if (pair[0] < 0) return -1;
    const int64_t ustart = pair[0] == 0 ? 0 : (int64_t)iv_get(&x->ends, pair[0] - 1);
    return ustart + pair[1] + k - 1;
}
static inline int lz_text_code(const fo_index* x, int64_t g) { return (int)((x->concat[g >> 5] >> (2 * (g & 31))) & 3); }
/* does the text spell lab[0..k-1] at [g-k+1, g], inside one unitig? */
static int lz_text_spells(const fo_index* x, const char* lab, int64_t g) {
    const int64_t k = x->k, gs = g - (k - 1);
    if (gs < 0 || g >= x->total_len) return 0;
    int64_t lo = 0, hi = x->n_unitigs;
    while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if ((int64_t)iv_get(&x->ends, mid) <= gs) lo = mid + 1; else hi = mid; }
    if (g >= (int64_t)iv_get(&x->ends, lo)) return 0;   /* crosses a unitig end */
    for (int64_t j = 0; j < k; j++) if ("ACGT"[lz_text_code(x, gs + j)] != lab[j]) return 0;
    return 1;
}
/* The seed table's entry of node v: offset of the last base of its k-mer's reported place; -1: nothing known; -1-d for the dummy node
 * that holds d bases (d = 0: the root).  The node's label is spelled by walking its incoming edges backwards (the last base of a node is
 * the character whose C-array range holds it; its predecessor holds the edge mark of that rank). */
static int64_t lz_node_pos(const fo_index* x, int64_t v, int* verified) {
    const int64_t k = x->k, n = x->n_nodes;
    *verified = 0;
    char lab[256];
    for (int64_t j = k - 1; j >= 0; j--) {
        int c = -1;
        for (int cc = 0; cc < 4; cc++) { const int64_t hi = cc == 3 ? n : x->C[cc + 1]; if (v >= x->C[cc] && v < hi) c = cc; }
        if (c < 0) return -1 - (k - 1 - j);         /* the node ends with '$' here: a dummy node that holds k-1-j bases (0, the root: -1 = nothing known) */
        lab[j] = "ACGT"[c];
        const int64_t rank = v - x->C[c];           /* the edge with this many c-marks before it */
        int64_t lo = 0, hi = n - 1;
        while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (bv_rank(&x->plane[c], mid + 1) >= rank + 1) hi = mid; else lo = mid + 1; }
        v = lo;
    }
    const int64_t g = lz_kmer_answer(x, lab);
    *verified = g >= 0 && lz_text_spells(x, lab, g);
    return g >= 0 ? g : -1;
}
/* the k-mer the text spells at [g-k+1, g] (inside one unitig): is g the place the reference reports for it? */
static int lz_text_safe(const fo_index* x, int64_t g) {
    const int64_t k = x->k;
    char lab[256];
    for (int64_t j = 0; j < k; j++) lab[j] = "ACGT"[lz_text_code(x, g - (k - 1) + j)];
    return lz_kmer_answer(x, lab) == g;
}

/* The whole k-mer that ends at t: its node, or -1 if it is not in the index (table for the first T bases, extends for the rest) */
static int64_t lz_full_lookup(lz_state* s, const char* q, int64_t t, int T, lz_chunks* cc, int64_t* chunk_bucket, int64_t* entries, int64_t* extends, int64_t* lines) {
    const fo_index* x = s->x;
    const int64_t k = x->k, p = t - k + 1;
    lz_chunk(cc, p, chunk_bucket); lz_chunk(cc, t, chunk_bucket);
    ival I = {0, x->n_nodes - 1};
    int off = 0;
    if (T > 0) {
        for (; off < T; off++) if (char_idx((char)(q[p + off] & ~32)) < 0) return -1;
        (*entries)++;
        for (int i = 0; i < T; i++) {
            const int ci = char_idx((char)(q[p + i] & ~32));
            ival r;
            r.first = x->C[ci] + bv_rank(&x->plane[ci], I.first);
            r.second = x->C[ci] + bv_rank(&x->plane[ci], I.second + 1) - 1;
            if (r.first > r.second) return -1;
            I = r;
        }
    }
    for (; off < k; off++) {
        const int ci = char_idx((char)(q[p + off] & ~32));
        if (ci < 0) return -1;
        lz_next_step(s);
        (*extends)++;
        I = lz_extend(s, ci, I, lines);
        if (I.first == -1) return -1;
    }
    return I.first;
}

/* PackedStrings::global_offset_to_local_offset (PackedStrings.hh:91-100) */
static inline void lz_locate(const fo_index* x, int64_t gs, int64_t* u, int64_t* ustart, int64_t* uend) {
    int64_t lo = 0, hi = x->n_unitigs;
    while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if ((int64_t)iv_get(&x->ends, mid) <= gs) lo = mid + 1; else hi = mid; }
    *u = lo; *ustart = lo == 0 ? 0 : (int64_t)iv_get(&x->ends, lo - 1); *uend = (int64_t)iv_get(&x->ends, lo);
}

/* One strand of one read (FinimizerIndex::search, FinimizerIndex.hh:119-185).  Found pairs are written to out[2*slot(i)],
 * slot(i) = mirror ? nk-1-i : i; slots of absent k-mers are left as they are.  Returns the number of found k-mers. */
/* a pair is reported for the k-mer that ends at text position g: does its window hold a k-mer whose reverse complement is in the index? */
static int lz_rc_window(lz_state* s, int64_t w);
static void lz_rc_taint(lz_state* s, int64_t g) { if (lz_rc_window(s, g >> 6)) s->tainted = 1; }
/* does a k-mer that ends in window w (64 text positions) have its reverse complement in the index too?  (FinDevIndex::rcwin) */
static int lz_rc_window(lz_state* s, int64_t w) {
    const fo_index* x = s->x;
    const int64_t k = x->k;
    unsigned char st = s->rcwin[w];
    if (!st) {
        char buf[256];
        st = 1;
        for (int64_t p = w * 64; p < w * 64 + 64 && p < x->total_len && st == 1; p++) {
            int64_t u, ustart, uend;
            lz_locate(x, p, &u, &ustart, &uend);
            if (p - ustart + 1 < k) continue;
            for (int64_t j = 0; j < k; j++) buf[j] = "TGCA"[lz_text_code(x, p - j)];
            if (lz_occurs(x, buf, 0, (int)k)) st = 2;
        }
        s->rcwin[w] = st;   /* (threads may write the same value twice) */
    }
    return st == 2;
}

/* pre (may be NULL): the pre-pass verdict of this strand made by lz_read (deferred second strand: the pair pre-pass) -- the first k-mer end
 * not proven absent and the one node the string that ends there belongs to (-1: several) */
typedef struct lz_pre_s { int64_t t0, node; } lz_pre;
static int64_t lz_strand(lz_state* s, const char* q, int64_t len, int64_t* out, int mirror, int T, int J, int flags, const lz_pre* pre) {
    const int reanchor = flags & 1, seeds = (flags & 2) != 0, count_safe = (flags & 4) != 0, ktab = (flags & 8) != 0, F = (flags >> 8) & 0xFF;
    /* (internal) a DEFERRED strand: only the k-mer ends its sister strand left open are searched (lz_read), by the walk kernel -- its probes are
     * search-stage work, no pre-pass verdict exists for it; fill_only: its pairs only fill slots that still hold (-1,-1) */
    const int deferred = (flags & 0x10000) != 0, fill_only = (flags & 0x20000) != 0;
    const fo_index* x = s->x;
    fo_lazy_counters* c = s->ctr;
    fo_lazy_counters scratch; if (!c) { memset(&scratch, 0, sizeof scratch); }
    fo_lazy_counters* cc = c ? c : &scratch;
    const int64_t k = x->k, nk = len - k + 1;
    if (nk <= 0) return 0;
    const int PM = s->lean ? lz_cbf_m(k) : (int)((T + 4) < k ? (T + 4) : k);   /* lean tables: a probe string is what the directional string filter holds */
    const int64_t MARGIN = 2 * k, LEAVE = 2 * k;
    const int64_t DELTA = T > 0 ? ((T + 1) < (k - 1) ? (T + 1) : (k - 1)) : k - 1;
    int64_t found_n = 0;
    /* plen: k-mer ends below it are DECIDED by this strand's probes, look-ups and text comparisons -- the strand's length, or (a deferred
     * strand) the end of its stretch + 1: the device's t_stop.  A WALK is not bounded by it: it runs on to the read's end (len), as the
     * reference's does, and what it reports behind the stretch is written like any pair (B forward: it wins; B reverse: it only fills). */
    const int64_t plen = (deferred && s->stop >= 0 && s->stop + 1 < len) ? s->stop + 1 : len;
#define LZ_EMIT(pos, u, off) do { const int64_t sl_ = mirror ? nk - 1 - (pos) : (pos); if (!fill_only || out[2 * sl_] == -1) { out[2 * sl_] = (u); out[2 * sl_ + 1] = (off); } found_n++; \
        if (s->rcwin && !s->tainted) lz_rc_taint(s, ((u) ? (int64_t)iv_get(&x->ends, (u) - 1) : 0) + (off) + k - 1); } while (0)

    if (!deferred && !pre) cc->strands++;
    /* probe pre-pass (its own kernel on the device: its own chunk loads) */
    lz_chunks pch = {-1, -1};
    const int64_t te0 = cc->table_entries, pl0 = cc->probe_lines;
    /* SEEDS: a probe string that occurs, ends at t0 and is the suffix of ONE node names the only k-mer that can end at t0 (a present
     * k-mer ending there has that string as its suffix, so it is that node's label).  Its place is looked up (lz_node_pos) and the
     * read compared with the text there, by the re-anchoring comparison below -- entered as if the position in front of the k-mer had
     * been a bad one.  Equal: the k-mer is present, there (disjoint index: its only place); a base that differs: probes across it, then
     * the k-mer behind it, as after any sequencing error.  The streaming search is only needed where a probe string is not unique. */
    /* a string that ends at t0 but is not unique: the whole k-mer is looked up --
     * present: an anchor like any other, its place from the seed table; absent: probing goes on behind it */
    int64_t seed_node = -1, seed_t0 = 0, pnode = -1, full_t0 = -1;
    int64_t guessed_at = -1;   /* the unresolved end a guess (below) has been tried for: one guess per end */
    int64_t kf_run = 0;        /* consecutive k-mer ends the k-mer filter ruled out */
    int uend_mark = 0;         /* the next LZ_PROBE_ON is the one behind a unitig end (diagnostic counters) */
    int from_stream = 0;       /* the walk in progress began at an anchor of the streaming search, whose state is frozen at s->end (else that state is stale) */
    lz_chunks sch = {-1, -1};
    int64_t t0;
    if (pre) { t0 = pre->t0; pnode = pre->node; }
    else t0 = deferred ? lz_probe(s, q, plen, k - 1, T, PM, &sch, &cc->chunks_search, &cc->table_entries, &cc->probe_extends, &cc->probe_lines, &pnode, 0, NULL)
                       : lz_probe(s, q, len, k - 1, T, PM, &pch, &cc->chunks_probe, &cc->table_entries, &cc->probe_extends, &cc->probe_lines, &pnode, F, &cc->filter_checks);
    if (!deferred && !pre) { cc->prepass_entries += cc->table_entries - te0; cc->prepass_lines += cc->probe_lines - pl0; }
    if (t0 < 0) return 0;
    if (!deferred) { cc->strands_searched++; if (seeds) cc->seed_verdicts++; }

    int64_t silent_until = t0, last_pres = t0, exact_from = 0;
    int64_t place_node = -1;   /* lean tables: the pre-pass's look found the k-mer that ends at t0 in the k-mer table: its slot's answer comes with the item */
    if (seeds && s->lean && pnode >= 0) { place_node = pnode; }
    else if (seeds && pnode >= 0) { seed_node = pnode; seed_t0 = t0; }
    else if (seeds) { full_t0 = t0; }
    else lz_restart(s, q, t0 - MARGIN > 0 ? t0 - MARGIN : 0, silent_until, J);
    /* from k-mer end T0 on: absence proofs; where a probe passes, a seed or the streaming search restarted 2k before it (`continue`s or `break`s) */
#define LZ_PROBE_ON(T0) { \
        const int64_t ul0_ = cc->probe_lines, ue0_ = cc->table_entries; const int um_ = uend_mark; uend_mark = 0; \
        if ((T0) >= plen) break; \
        t0 = lz_probe(s, q, plen, (T0), T, PM, &sch, &cc->chunks_search, &cc->table_entries, &cc->probe_extends, &cc->probe_lines, &pnode, 0, NULL); \
        if (um_) { cc->uend_probes++; cc->uend_lines += cc->probe_lines - ul0_; cc->uend_entries += cc->table_entries - ue0_; } \
        if (t0 < 0) break; \
        if (seeds && pnode >= 0) { seed_node = pnode; seed_t0 = t0; continue; } \
        if (seeds) { full_t0 = t0; continue; } \
        silent_until = t0; last_pres = t0; exact_from = 0; lz_restart(s, q, t0 - MARGIN > 0 ? t0 - MARGIN : 0, silent_until, J); \
        continue; }
    for (;;) {
        int64_t u = 0, ustart = 0, uend = 0, wend = 0, wg = 0, last_win = -1, E = 0, tE = 0, unresolved = 0;
        int at_uend = 0, strand_over = 0, resume_stream = 0, from_seed = 0, uend_inside = 0, redo = 0;
        if (place_node >= 0) {
            /* a PLACE item (lean tables): the verified answer of the k-mer that ends at t0 -- an anchor like a k-mer-table hit: the unitig of the
             * place, the run, the walk.  An unverified answer travels as a probe item instead: the filter knows the string, the whole k-mer is looked up */
            int ver = 0;
            const int64_t g = lz_node_pos(x, place_node, &ver);
            place_node = -1;
            if (!ver || g < 0) { cc->fbf_lookups++; full_t0 = t0; continue; }
            cc->place_anchors++;
            lz_locate(x, g - (k - 1), &u, &ustart, &uend);
            if (s->rcwin) s->tainted = 1;   /* (reported without a text comparison: no window flag passes by) */
            LZ_EMIT(t0 - (k - 1), u, g - (k - 1) - ustart);
            from_stream = 0;
            wend = t0 + 1; wg = g;
            if (wend >= len) break;
            goto walk_on;
        }
        if (full_t0 >= 0) {
            const int64_t t = full_t0;
            full_t0 = -1;
            if (t >= plen) break;
            int64_t v = -2;   /* the k-mer's node, -1: not in the index, -2: not asked yet */
            if (ktab) {   /* (any k <= 255 since round 5: the device folds a long k-mer's words into the hash as its chunk cache brings them, fin_kernel_w.hip W_KF0B) */
                /* K-MER TABLE: a hash table from every k-mer of the text to its SBWT node (one 16-byte slot on the device) is asked instead
                 * of looking the whole k-mer up through the SBWT.  While it keeps saying "not there" the next ends are asked directly --
                 * the probe string of this stretch occurs all over the index (a repeat), a short probe would pass again -- except that
                 * every eighth end is probed first (a failing probe settles k-PM+1 ends at once: the way out of the stretch). */
                int valid = 1;
                for (int64_t j = t - k + 1; j <= t; j++) if (char_idx((char)(q[j] & ~32)) < 0) valid = 0;
                if (k >= 64) { for (int64_t j = t - k + 1; j < t; j += 32) lz_chunk(&sch, j, &cc->chunks_search); }   /* every chunk of the k-mer, one key word each */
                else lz_chunk(&sch, t - k + 1, &cc->chunks_search);
                lz_chunk(&sch, t, &cc->chunks_search);
                cc->ktab_lookups++;
                if (valid) { int64_t d0 = 0, d1 = 0, d2 = 0, d3 = 0; lz_chunks dch = {-1, -1}; lz_state* const s0 = s; fo_lazy_counters* const keep = s0->ctr; s0->ctr = NULL; v = lz_full_lookup(s0, q, t, T, &dch, &d0, &d1, &d2, &d3); s0->ctr = keep; }
                else v = -1;
                if (v < 0) {
                    if (t + 1 >= plen) break;
                    if (++kf_run % 8 == 0) LZ_PROBE_ON(t + 1)
                    full_t0 = t + 1;
                    continue;
                }
            }
            kf_run = 0;
            if (v == -2) {
                const int64_t fl0 = cc->probe_lines, fe0 = cc->table_entries;
                v = lz_full_lookup(s, q, t, T, &sch, &cc->chunks_search, &cc->table_entries, &cc->probe_extends, &cc->probe_lines);
                cc->full_lookups++; cc->full_lines += cc->probe_lines - fl0; cc->full_entries += cc->table_entries - fe0;
                if (v < 0) { if (t + 1 >= plen) break; LZ_PROBE_ON(t + 1) }
            }
            if (ktab) { cc->place_anchors++; if (k >= 64) cc->text_windows += (k + 63) / 64; }   /* (the k-mer table's slot holds the answer: the locate and the claim's comparison -- one window inside place_anchors' 36 bytes; a long k-mer's further windows beside it) */
            else cc->seed_lookups++;   /* the anchor table's entry */
            int ver = 0;
            const int64_t g = lz_node_pos(x, v, &ver);   /* (the k-mer is present: the reference's answer for its node, verified or not) */
            if (g < 0 || g - (k - 1) < 0 || g - (k - 1) >= x->total_len) { silent_until = t; last_pres = t; exact_from = 0; lz_restart(s, q, t - MARGIN > 0 ? t - MARGIN : 0, silent_until, J); continue; }   /* (an answer outside the text: the streaming search reports it as the reference does) */
            lz_locate(x, g - (k - 1), &u, &ustart, &uend);
            if (!ver || s->rcwin) s->tainted = 1;   /* the reference's answer, reported as the reference does -- but the text there does not spell the k-mer (an unverified entry); on an index with reverse-complement pairs: always (the device reports this k-mer without a text comparison: no window flag passes by) */
            LZ_EMIT(t - (k - 1), u, g - (k - 1) - ustart);
            cc->full_anchors++; from_stream = 0;
            wend = t + 1; wg = g;
            if (wend >= len) break;
            goto walk_on;
        }
        if (seed_node >= 0) {
            int ver = 0;
            const int64_t g = lz_node_pos(x, seed_node, &ver);
            seed_node = -1;
            cc->seed_lookups++;
            if (g < -1) {
                /* the seed string ends only a dummy node that holds d = -1-g bases: no k-mer ends with it, nor with an extension of it
                 * by fewer than k-d bases (their nodes are that dummy's descendants, still $-padded): probing goes on at seed_t0 + k - d */
                if (seed_t0 + k - (-1 - g) >= plen) break;
                LZ_PROBE_ON(seed_t0 + k - (-1 - g))
            }
            if (g < 0 || !ver) { full_t0 = seed_t0; continue; }   /* no place where the text spells the node's k-mer: the whole k-mer is looked up */
            lz_locate(x, g - (k - 1), &u, &ustart, &uend);
            E = seed_t0 - k; tE = g - k; unresolved = seed_t0; from_seed = 2;   /* (2: an exact seed -- if the comparison fails the k-mer at seed_t0 is absent) */
            goto after_walk;
        }
        /* ---- streaming search at s->end ---- */
        if (s->end >= len) break;
        lz_chunk(&sch, s->end, &cc->chunks_search);
        const int valid = char_idx((char)(q[s->end] & ~32)) >= 0;
        lz_step(s, q);
        const int64_t end = s->end;
        int found = 0; int64_t fin_end = 0, fin_colex = 0;
        if (valid) {
            if (s->iskm) last_pres = end;
            /* (exact_from < 0: a verified short restart whose check is due at the first position that reports) */
            const int check_due = exact_from < 0 && end == silent_until;
            if (check_due) exact_from = -exact_from;
            if (check_due && s->kstart <= end - DELTA) {
                /* the k-mer interval's string still reaches back to the restart point: its true start may lie before it, nothing
                 * after it is known exactly -- redo from k-1 bases back (presence is exact by the k-window alone) */
                cc->restarts_failed_check++;
                silent_until = end; exact_from = end + k; lz_restart(s, q, end - (k - 1), silent_until, J);
                continue;
            }
            if (s->iskm && end >= silent_until && end < exact_from) {
                /* a k-mer is present where only presence is known exactly: redo with the full margin, silently up to here */
                cc->restarts_full_margin++;
                silent_until = end; exact_from = 0; lz_restart(s, q, end - MARGIN > 0 ? end - MARGIN : 0, silent_until, J);
                continue;
            }
            if (s->iskm && s->dq_cnt && end >= silent_until) {
                found = 1; fin_end = LZ_DQ(s, 0).end; fin_colex = LZ_DQ(s, 0).colex;
            }
        }
        if (!found) {
            if (end - last_pres >= LEAVE && end >= silent_until) {
                /* a long stretch without any k-mer: back to absence proofs */
                if (end + 1 >= len) break;
                LZ_PROBE_ON(end + 1)
            }
            s->end++;
            continue;
        }
        /* ---- dictionary anchor (FinimizerIndex.hh:148-174): branch record if it is at or after the finimizer's end ---- */
        cc->anchors++;
        const int use_branch = s->bu_end >= fin_end;
        int64_t g;
        if (use_branch) g = lookup_from_branch_dictionary(x, s->bu_colex) + (end - s->bu_end);
        else g = lookup_from_finimizer_dictionary(x, fin_colex) + (end - fin_end);
        const int64_t gs = g - (k - 1);
        if (gs < 0 || gs >= x->total_len) { s->end++; continue; }   /* unreachable on a consistent index: reported absent */
        lz_locate(x, gs, &u, &ustart, &uend);
        LZ_EMIT(end - (k - 1), u, gs - ustart);
        s->end++;
        if (s->end >= len) break;
        /* ---- walk (walk_in_unitigs, FinimizerIndex.hh:47-102): the streaming state stays frozen at s->end ---- */
        wend = s->end; wg = g; from_stream = 1;
    walk_on:
        at_uend = 0;
        while (wend < len) {
            if (wg + 1 >= uend) { at_uend = 1; break; }
            const int ci = char_idx((char)(q[wend] & ~32));
            lz_chunk(&sch, wend, &cc->chunks_search);
            if (((wg + 1) >> 6) != last_win) { last_win = (wg + 1) >> 6; cc->text_windows++; }
            if (ci < 0 || (int)((x->concat[(wg + 1) >> 5] >> (2 * ((wg + 1) & 31))) & 3) != ci) break;
            wg++;
            LZ_EMIT(wend - (k - 1), u, wg - (k - 1) - ustart);
            cc->walk_bases++;
            wend++;
        }
        if (wend >= len) break;
        if (seeds && at_uend) { uend_mark = 1; LZ_PROBE_ON(wend) }   /* the unitig ended and the read goes on: the next k-mer end is probed (absent, a seed, or streaming) */
    after_walk:
        if (from_seed || (reanchor && !at_uend)) {
            /* TEXT RE-ANCHORING.  A k-mer found by comparing the read with the text is reported there only if that is the place the
             * reference reports for it (lz_text_safe: on a disjoint index every place is).  The read disagrees with the text at position E = wend.
             * (1) Every k-mer containing E ends in [E, E+k-1]: proven absent by probes across E.  (2) The k-mer after it,
             * q[E+1..E+k], is compared with the text right behind the disagreeing text base: if all k bases agree it is present,
             * there, and the walk goes on from it -- no streaming search, no dictionary.  A second disagreement inside those k
             * bases is the next E.  Whatever cannot be proven goes back to the streaming search, restarted with the full margin. */
            if (!from_seed) { E = wend; tE = wg + 1; unresolved = wend; }
            if (unresolved >= plen) break;   /* (a walk carried a deferred strand past the end of its stretch: done) */
            for (;;) {
                if (!from_seed) {
                    int64_t bnode = -1, blast = 0;
                    const int64_t bl0 = cc->probe_lines, be0 = cc->table_entries;
                    const int bridged = lz_bridge(s, q, plen, &unresolved, E, T, PM, &sch, &cc->chunks_search, &cc->table_entries, &cc->probe_extends, &cc->probe_lines, &bnode, &blast);
                    cc->bridge_lines += cc->probe_lines - bl0; cc->bridge_entries += cc->table_entries - be0;
                    if (!bridged) {
                        /* a string across the bad position occurs.  One node ends it and it ends at the unresolved end: a seed.  One node
                         * ends it but it stops short (k > 32: after an indel the strings behind it match, 32 bases do not reach the end):
                         * a GUESS of where the read lies now -- the k-mer at the unresolved end is compared with the text there (a
                         * comparison can only find k-mers: any guess is sound; a guess that fails is not repeated).  Else the whole k-mer. */
                        if (seeds && bnode >= 0 && blast == unresolved) { seed_node = bnode; seed_t0 = unresolved; redo = 1; break; }
                        if (seeds && bnode >= 0 && guessed_at != unresolved) {
                            guessed_at = unresolved;
                            cc->seed_lookups++;
                            int gver = 0;
                            int64_t g = lz_node_pos(x, bnode, &gver);
                            if (g >= 0 && gver) {
                                g += unresolved - blast;
                                const int64_t gs = g - (k - 1);
                                if (gs >= 0 && gs < x->total_len) {
                                    lz_locate(x, gs, &u, &ustart, &uend);
                                    if (g < uend) { E = unresolved - k; tE = g - k; from_seed = 1; continue; }
                                }
                            }
                        }
                        if (seeds) { full_t0 = unresolved; redo = 1; }   /* several nodes, or a guess that could not be used */
                        else resume_stream = 1;
                        break;
                    }
                    if (E + k >= plen) { strand_over = 1; break; }              /* no k-mer ends after E+k-1 */
                    if (tE + k >= uend) { unresolved = E + k; resume_stream = 1; uend_inside = 1; break; }   /* the unitig ends inside the next k-mer */
                }
                int64_t m = 0;
                for (; m < k; m++) {
                    const int ci = char_idx((char)(q[E + 1 + m] & ~32));
                    lz_chunk(&sch, E + 1 + m, &cc->chunks_search);
                    if (((tE + 1 + m) >> 6) != last_win) { last_win = (tE + 1 + m) >> 6; cc->text_windows++; }
                    if (ci < 0 || (int)((x->concat[(tE + 1 + m) >> 5] >> (2 * ((tE + 1 + m) & 31))) & 3) != ci) break;
                }
                if (m == k && from_seed != 2 && count_safe) cc->safe_checks++;
                if (m == k && from_seed != 2 && !lz_text_safe(x, tE + k)) {
                    /* the k-mer is in the text here, but this is not the place the reference reports for it (a k-mer with several
                     * places, or one whose finimizer's stored place lies elsewhere): the streaming search decides from its end on.
                     * (an exact seed's place IS the reference's answer: lz_node_pos) */
                    cc->unsafe_places++;
                    from_seed = 0;
                    if (seeds) { full_t0 = E + k; redo = 1; break; }   /* it is present: its node's entry is the reference's answer -- the whole k-mer is looked up */
                    unresolved = E + k; resume_stream = 1;
                    break;
                }
                if (m == k) {
                    if (from_seed) cc->seed_anchors++; else cc->text_anchors++;
                    from_seed = 0; from_stream = 0;   /* (the device gives the frozen streaming state up here) */
                    LZ_EMIT(E + 1, u, tE + 1 - ustart);
                    wg = tE + k; wend = E + k + 1;
                    break;
                }
                unresolved = E + k + (from_seed == 2);   /* ends [E+k, E2+k-1] all contain the next bad position E2; a seed's own k-mer is decided: absent */
                tE = tE + 1 + m; E = E + 1 + m; from_seed = 0;
                if (unresolved >= plen) { strand_over = 1; break; }
            }
            if (strand_over) break;
            if (redo) continue;
            if (!resume_stream) { if (wend >= len) break; goto walk_on; }
            resume_stream = 0;
            if (seeds && uend_inside) LZ_PROBE_ON(unresolved)
            cc->restarts_margin++;
            silent_until = unresolved; last_pres = unresolved; exact_from = 0;
            lz_restart(s, q, unresolved - MARGIN > 0 ? unresolved - MARGIN : 0, silent_until, J);
            continue;
        }
        /* the walk ended before position wend: the normal path applies there again, which needs the streaming state at wend */
        last_pres = wend - 1; exact_from = 0;
        const int64_t gap = from_stream ? wend - s->end : (int64_t)1 << 40;   /* (no frozen state to catch up from: restart) */
        if (gap > DELTA && !at_uend && DELTA < k - 1) {
            /* verified short restart: kmer_start and start of a search begun at c are max(c, true value) and only move forward;
             * checked when the search arrives at wend (above) */
            cc->restarts_short++;
            lz_restart(s, q, wend - DELTA, wend, J); exact_from = -(wend + k);
        } else if (gap > k - 1 && !at_uend) {
            cc->restarts_k1++;
            lz_restart(s, q, wend - (k - 1), wend, J); exact_from = wend + k;
        } else if (gap > MARGIN) {
            cc->restarts_margin++;
            lz_restart(s, q, wend - MARGIN > 0 ? wend - MARGIN : 0, wend, J);
        }
        silent_until = wend;
    }
#undef LZ_EMIT
#undef LZ_PROBE_ON
    return found_n;
}

static int lz_pstep(lz_state* s, const char* q, int64_t len, int T, int PM, int flags, struct lz_pre_s* pre);
/* the pair pre-pass's LOOK at a strand: is its first k-mer in the index -- asked of the k-mer table (k <= 31, flags bit 3: one slot) -- or,
 * without that table, does the string of PM bases that ends at its first k-mer end occur (one probe)?  1: yes, pre->t0 = k-1 and
 * pre->node = the k-mer's node / the one node the string ends (-1: several); 0: no, pre->t0 = the first k-mer end not proven absent (-1: none) */
static int lz_look(lz_state* s, const char* q, int64_t len, int T, int PM, int flags, lz_chunks* pch, lz_pre* pre) {
    const fo_index* x = s->x;
    const int64_t k = x->k;
    fo_lazy_counters scratch; memset(&scratch, 0, sizeof scratch);
    fo_lazy_counters* cc = s->ctr ? s->ctr : &scratch;
    const int F = (flags >> 8) & 0xFF;
    pre->node = -1;
    if ((flags & 8) && (k <= 31 || s->lean)) {
        int valid = 1;
        for (int64_t j = 0; j < k; j++) if (char_idx((char)(q[j] & ~32)) < 0) valid = 0;
        lz_chunk(pch, 0, &cc->chunks_probe);
        int64_t v = -1;
        if (valid) {
            cc->ktab_lookups++; cc->prepass_ktab++;
            int64_t d0 = 0, d1 = 0, d2 = 0, d3 = 0; lz_chunks dch = {-1, -1}; fo_lazy_counters* const keep = s->ctr; s->ctr = NULL;
            v = lz_full_lookup(s, q, k - 1, T, &dch, &d0, &d1, &d2, &d3); s->ctr = keep;
        }
        if (v >= 0) { pre->t0 = k - 1; pre->node = v; return 1; }
        pre->t0 = k < len ? k : -1;
        return 0;
    }
    (void)F;
    pre->t0 = k - 1;
    return lz_pstep(s, q, len, T, PM, flags, pre);
}
/* ... and one STEP of a strand's probing at k-mer end pre->t0 (fin_prepass.hip, probe_step): the absence filter, then the string of PM bases
 * that ends there.  1: it occurs (pre->node: the one node it ends, -1 several); 0: pre->t0 = the first k-mer end not proven absent (-1: none).
 * The device loads the step's chunks anew: those of t0 - max(PM-1, F) and of t0. */
static int lz_pstep(lz_state* s, const char* q, int64_t len, int T, int PM, int flags, lz_pre* pre) {
    fo_lazy_counters scratch; memset(&scratch, 0, sizeof scratch);
    fo_lazy_counters* cc = s->ctr ? s->ctr : &scratch;
    const int F = (flags >> 8) & 0xFF;
    const int64_t te0 = cc->table_entries, pl0 = cc->probe_lines, at = pre->t0;
    const int64_t span = (PM - 1) > F ? (PM - 1) : F;
    lz_chunks fresh = {-1, -1}, dummy = {-1, -1}; int64_t uncounted = 0;
    lz_chunk(&fresh, at - span, &cc->chunks_probe); lz_chunk(&fresh, at, &cc->chunks_probe);
    pre->node = -1;
    s->probe_once = 1; s->next_t0 = -1;
    const int64_t t0 = lz_probe(s, q, len, at, T, PM, &dummy, &uncounted, &cc->table_entries, &cc->probe_extends, &cc->probe_lines, &pre->node, F, &cc->filter_checks);
    s->probe_once = 0;
    cc->prepass_entries += cc->table_entries - te0; cc->prepass_lines += cc->probe_lines - pl0;
    if (t0 == at) return 1;
    pre->t0 = s->next_t0; pre->node = -1;
    return 0;
}

/* ---- THE FAST PATH of the pair pre-pass (round 4; the device's fin_prepass.hip, stated independently) -----------------------------------
 * The common read comes from one place of the indexed text and carries a few substitution errors.  Once a k-mer of strand A is known to be in
 * the index, with the reference's answer G for it at a place where the text spells it (lz_node_pos: "verified"), everything the reference
 * reports for the read follows from ONE comparison of strand A with the text behind that place, provided the read lies inside that unitig:
 *   - a k-mer end t whose k-mer holds no disagreeing base: the k-mer is in the text there.  The reference reaches it by its walk
 *     (walk_in_unitigs, FinimizerIndex.hh:47-102) from the k-mer before it, or -- the first k-mer of a stretch -- by its dictionaries, which
 *     answer that place iff the place is safe (lz_text_safe; the anchor k-mer's own place is its answer by definition);
 *   - a k-mer end t whose k-mer holds a disagreeing base E: absent if a string of m bases inside the k-mer occurs in no unitig in EITHER
 *     orientation (the canonical string filter: no false negative) -- and then the sister strand's k-mer of that slot, which holds the
 *     string's reverse complement, is absent too.  The slots A fills need nothing from the sister either, as with a deferred strand
 *     (lz_read): A forward wins; A reverse reports k-mers whose reverse complements are not in the index (no flagged window, lz_rc_window).
 * So the read is finished: pairs in A's stretches, (-1,-1) across the disagreeing bases, nothing left for either strand.  A read none of whose
 * looked-up k-mers is in the index is absent altogether if the filter knows none of the strings laid across it end to end.  Everything else
 * -- no place, an unverified answer, a unitig end inside the read, a non-ACGT base, more than LZ_FAST_MAXE disagreeing bases, a string the
 * filter knows (or takes for known), an unsafe place, a flagged window, more than 256 bases -- is left to lz_read's other routes, untouched. */
#define LZ_FAST_MAXE 4
#define LZ_FAST_MAXLEN 256
#define LZ_CBF_BITS 5
typedef struct lz_cbf_s { uint32_t* w; int log2_blocks; int m; } lz_cbf;
static inline uint64_t lz_cbf_hash(uint64_t key) {
    key ^= key >> 29; key *= 0xBF58476D1CE4E5B9ull; key ^= key >> 32; key *= 0x94D049BB133111EBull; key ^= key >> 29;
    return key;
}
static inline void lz_cbf_where(const lz_cbf* f, uint64_t canon, uint64_t* block, uint32_t m4[4]) {
    const uint64_t h = lz_cbf_hash(canon);
    *block = (h >> 35) & ((1ull << f->log2_blocks) - 1ull);
    m4[0] = m4[1] = m4[2] = m4[3] = 0;
    for (int i = 0; i < LZ_CBF_BITS; i++) { const uint32_t p = (uint32_t)(h >> (7 * i)) & 127u; m4[p >> 5] |= 1u << (p & 31u); }
}
/* every string of m bases inside one unitig, in canonical form: the smaller of its 2-bit key (first base in the low bits) and its reverse complement's */
static lz_cbf* lz_cbf_build(const fo_index* x) {
    lz_cbf* f = (lz_cbf*)calloc(1, sizeof(lz_cbf));
    f->m = lz_cbf_m(x->k);
    f->log2_blocks = 4;
    while ((8ull << f->log2_blocks) < (uint64_t)x->total_len && f->log2_blocks < 31) f->log2_blocks++;
    f->w = (uint32_t*)calloc((size_t)4 << f->log2_blocks, 4);
    const int m = f->m;
    const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int64_t u = 0; u < x->n_unitigs; u++) {
        const int64_t b = u ? (int64_t)iv_get(&x->ends, u - 1) : 0, e = (int64_t)iv_get(&x->ends, u);
        uint64_t fw = 0, rv = 0;
        for (int64_t g = b; g < e; g++) {
            const uint64_t c = (uint64_t)((x->concat[g >> 5] >> (2 * (g & 31))) & 3);
            fw = (fw >> 2) | (c << (2 * (m - 1)));
            rv = ((rv << 2) | (3ull - c)) & mask;
            if (g - b + 1 >= m) {
                uint64_t blk; uint32_t m4[4];
                lz_cbf_where(f, fw < rv ? fw : rv, &blk, m4);
                for (int i = 0; i < 4; i++) if (m4[i]) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                    f->w[4 * blk + i] |= m4[i];
                }
            }
        }
    }
    return f;
}
static void lz_cbf_free(lz_cbf* f) { if (f) { free(f->w); free(f); } }
/* does the filter know q[a .. a+m-1] (ACGT only)? */
static int lz_cbf_knows(const lz_cbf* f, const char* q, int64_t a) {
    const int m = f->m;
    uint64_t fw = 0, rv = 0;
    for (int i = 0; i < m; i++) {
        const uint64_t c = (uint64_t)char_idx((char)(q[a + i] & ~32));
        fw |= c << (2 * i);
        rv |= (3ull - c) << (2 * (m - 1 - i));
    }
    uint64_t blk; uint32_t m4[4];
    lz_cbf_where(f, fw < rv ? fw : rv, &blk, m4);
    for (int i = 0; i < 4; i++) if ((f->w[4 * blk + i] & m4[i]) != m4[i]) return 0;
    return 1;
}
typedef struct { int ok; int64_t u, off0; int nE; int64_t E[LZ_FAST_MAXE]; } lz_fast_res;
/* strand q, its k-mer that ends at t_anchor is in the index and the reference's answer for it is the text place that ends at g_ans (verified) */
static int lz_fast_try(lz_state* s, const char* q, int64_t len, int64_t t_anchor, int64_t g_ans, int count_safe, lz_fast_res* res) {
    const fo_index* x = s->x;
    fo_lazy_counters scratch; memset(&scratch, 0, sizeof scratch);
    fo_lazy_counters* cc = s->ctr ? s->ctr : &scratch;
    const int64_t k = x->k, m = s->cbf->m;
    res->ok = 0;
    if (len > LZ_FAST_MAXLEN || g_ans < t_anchor) return 0;
    const int64_t gs = g_ans - t_anchor;
    int64_t u, ustart, uend;
    cc->fast_tries++;
    lz_locate(x, gs, &u, &ustart, &uend);
    if (gs < ustart || gs + len > uend) return 0;   /* the unitig ends inside the read */
    /* the comparison: the strand's chunks, and a 32-base word of text more than chunks */
    cc->fast_chunks += (len + 31) / 32; cc->fast_text_words += (len + 31) / 32 + 1;
    int nE = 0, bad = 0; int64_t E[LZ_FAST_MAXE];
    for (int64_t i = 0; i < len; i++) {
        const int ci = char_idx((char)(q[i] & ~32));
        if (ci < 0) { bad = 1; continue; }
        if (ci != lz_text_code(x, gs + i)) { if (nE < LZ_FAST_MAXE) E[nE++] = i; else bad = 1; }
    }
    if (bad) return 0;
    if (s->rcwin) for (int64_t w = (gs + k - 1) >> 6; w <= (gs + len - 1) >> 6; w++) if (lz_rc_window(s, w)) return 0;
    if (t_anchor != k - 1 && (nE == 0 || E[0] >= k)) {   /* the first k-mer is not the anchor and holds no disagreeing base: reported here only if the place is safe */
        if (count_safe) cc->safe_checks++;
        if (!lz_text_safe(x, gs + k - 1)) return 0;
    }
    int64_t covered = k - 2;
    int known = 0, unsafe = 0;
    for (int e = 0; e < nE; e++) {
        int64_t lo = E[e] > k - 1 ? E[e] : k - 1; if (lo < covered + 1) lo = covered + 1;
        const int64_t hi = E[e] + k - 1 < len - 1 ? E[e] + k - 1 : len - 1;
        for (int n = 0; n < 3 && lo <= hi; n++) {
            const int64_t a = lo - (m - 1) < E[e] ? lo - (m - 1) : E[e];   /* q[a .. a+m-1] holds E and lies inside every k-mer that ends in [lo, a+k-1] */
            cc->fast_cbf++;
            if (lz_cbf_knows(s->cbf, q, a)) known = 1;
            covered = a + k - 1; lo = covered + 1;
        }
        if (lo <= hi) known = 1;
        const int64_t t = E[e] + k;
        const int last = e + 1 == nE || E[e + 1] > t;
        if (last && t < len) {
            if (count_safe) cc->safe_checks++;
            if (!lz_text_safe(x, gs + t)) unsafe = 1;
        }
    }
    if (known || unsafe) return 0;
    res->ok = 1; res->u = u; res->off0 = gs - ustart; res->nE = nE;
    for (int e = 0; e < nE; e++) res->E[e] = E[e];
    return 1;
}
/* the k-mer of strand q that ends at t: is it in the index?  (one slot of the k-mer table on the device; here the SBWT says) -- its answer and whether it is verified */
static int lz_fast_look(lz_state* s, const char* q, int64_t t, int T, int64_t* g_ans, int* ver) {
    const fo_index* x = s->x;
    const int64_t k = x->k;
    for (int64_t j = t - k + 1; j <= t; j++) if (char_idx((char)(q[j] & ~32)) < 0) return 0;
    int64_t d0 = 0, d1 = 0, d2 = 0, d3 = 0; lz_chunks dch = {-1, -1};
    fo_lazy_counters* const keep = s->ctr; s->ctr = NULL;
    const int64_t v = lz_full_lookup(s, q, t, T, &dch, &d0, &d1, &d2, &d3);
    s->ctr = keep;
    if (v < 0) return 0;
    *g_ans = lz_node_pos(x, v, ver);
    return 1;
}
static int lz_fast_all_absent(lz_state* s, const char* q, int64_t len) {
    fo_lazy_counters scratch; memset(&scratch, 0, sizeof scratch);
    fo_lazy_counters* cc = s->ctr ? s->ctr : &scratch;
    const int64_t k = s->x->k, m = s->cbf->m;
    if (len > LZ_FAST_MAXLEN) return 0;
    cc->fast_chunks += (len + 31) / 32;
    for (int64_t i = 0; i < len; i++) if (char_idx((char)(q[i] & ~32)) < 0) return 0;
    int known = 0;
    for (int64_t t = k - 1; t < len && !known; ) {   /* (the device asks four strings at a time and stops behind a group with a known one) */
        for (int i = 0; i < 4; i++) {
            const int have = t < len;
            const int64_t a = (have ? t : k - 1) - (m - 1);
            cc->fast_cbf++;
            if (lz_cbf_knows(s->cbf, q, a) && have) known = 1;
            if (have) t = a + k;
        }
    }
    return !known;
}
/* fills out[] and returns 1 when the fast path finishes the read.  f_hit / v_hit, f_node / v_node: what the pair pre-pass's first looks found
 * (lz_read made them: the forward strand's first k-mer, and -- only if that failed -- the reverse strand's) */
/* (ver_only: 32 <= k <= 63 -- the device's two-word anchor table holds the verified k-mers only, in 32-byte slots: a k-mer whose answer is not
 *  verified counts as not found, at every look; the first looks are the fast path's own: f_node / v_node are not used) */
static int lz_fast_read(lz_state* s, const char* q, const char* rcbuf, int64_t len, int64_t* out, int T, int flags, int f_hit, int v_hit, int64_t f_node, int64_t v_node, int ver_only) {
    const fo_index* x = s->x;
    fo_lazy_counters scratch; memset(&scratch, 0, sizeof scratch);
    fo_lazy_counters* cc = s->ctr ? s->ctr : &scratch;
    const int64_t k = x->k, nk = len - k + 1;
    const int count_safe = (flags & 4) != 0;
    lz_fast_res fr; fr.ok = 0;
    int rev = 0, absent = 0;
    int64_t g1 = -1; int ver1 = 0;
    if (ver_only) {   /* the first looks: the forward strand's first k-mer, then the reverse strand's */
        cc->fast_looks2++; cc->fast_chunks += 2;
        f_hit = lz_fast_look(s, q, k - 1, T, &g1, &ver1) && ver1;
        if (!f_hit) { cc->fast_looks2++; cc->fast_chunks += 2; v_hit = lz_fast_look(s, rcbuf, k - 1, T, &g1, &ver1) && ver1; } else v_hit = 0;
    }
    if (f_hit || v_hit) {
        int ver = ver1;
        const int64_t g = ver_only ? g1 : lz_node_pos(x, f_hit ? f_node : v_node, &ver);   /* (the table's slot holds it: no further load) */
        if (ver && g >= 0 && lz_fast_try(s, f_hit ? q : rcbuf, len, k - 1, g, count_safe, &fr)) rev = !f_hit;
    } else {
        /* neither first k-mer is in the index: the strands' LAST k-mers, then their MIDDLE ones; a k-mer that is found settles the attempt */
        int hit = 0;
        for (int w = 0; w < 2 && !hit; w++) {
            const int64_t t = w == 0 ? len - 1 : (len + k) / 2 - 1;
            cc->fast_redesc++;
            if (t <= k - 1 || (w == 1 && t >= len - 1)) continue;
            int64_t g = -1; int ver = 0;
            const int64_t nchk = ver_only ? 3 : 1 + (((t - k + 1) >> 5) != (t >> 5));   /* chunks a look loads */
            if (ver_only) cc->fast_looks2++; else cc->fast_looks++;
            cc->fast_chunks += nchk;
            if (lz_fast_look(s, q, t, T, &g, &ver) && (ver || !ver_only)) { hit = 1; if (ver && g >= 0) (void)lz_fast_try(s, q, len, t, g, count_safe, &fr); }
            else {
                if (ver_only) cc->fast_looks2++; else cc->fast_looks++;
                cc->fast_chunks += nchk;
                if (lz_fast_look(s, rcbuf, t, T, &g, &ver) && (ver || !ver_only)) { hit = 1; if (ver && g >= 0 && lz_fast_try(s, rcbuf, len, t, g, count_safe, &fr)) rev = 1; }
            }
        }
        if (!hit && lz_fast_all_absent(s, q, len)) absent = 1;
    }
    if (!fr.ok && !absent) return 0;
    for (int64_t sl = 0; sl < nk; sl++) {
        int gap = absent;
        for (int e = 0; e < fr.nE && !gap; e++) if (fr.E[e] >= sl && fr.E[e] <= sl + k - 1) gap = 1;
        const int64_t i = rev ? nk - 1 - sl : sl;
        out[2 * i] = gap ? -1 : fr.u; out[2 * i + 1] = gap ? -1 : fr.off0 + sl;
    }
    cc->fast_reads++; cc->fast_absent_reads += absent;
    return 1;
}

/* search(read), search(rc(read)), merge: a forward hit wins, else the reverse strand's pair at len-k-i (search_fmin.hh:47-60).
 * DEFERRED SECOND STRAND (flags bit 4; the caller asserts (1) that no k-mer of the index has its reverse complement in the index too -- true of
 * any set that holds every canonical k-mer once -- and (2) that every text place is the place the reference reports for its k-mer (no unsafe
 * place: a disjoint set), so that every reported pair is a k-mer of the index -- with duplicated k-mers the reference may walk along a place
 * where the read's k-mers are not (it compares one new base per step), and "found" then proves nothing; both are counted at upload on the
 * device): one strand A is searched first, and its sister B only where A left slots open -- between the first and the last of them: a
 * k-mer A found is final (A forward: a forward hit wins; A reverse: its reverse complement, the forward k-mer, is not in the index), and
 * B's k-mers outside that stretch are the reverse complements of k-mers A found, hence absent: nothing of B's search outside the
 * stretch can reach into it.  B's probes are then the walk kernel's (search stage), not the pre-pass's.
 * WHICH strand is A is a matter of cost only.  The pair pre-pass decides it read by read (fin_prepass.hip): the forward strand if its first
 * k-mer is in the index (lz_look); else the reverse strand if its first k-mer is; else the two strands are probed step by step in turn
 * (lz_step) and the first whose string occurs is A; the other is deferred if it has a k-mer end left, else it is absent altogether. */
static int64_t lz_read(lz_state* s, const char* q, int64_t len, char* rcbuf, int64_t* out, int T, int J, int flags, int64_t* positives) {
    const fo_index* x = s->x;
    const int64_t k = x->k, nk = len - k + 1;
    if (nk <= 0) return 0;
    for (int64_t i = 0; i < 2 * nk; i++) out[i] = -1;
    reverse_complement(q, len, rcbuf);
    const int PM = s->lean ? lz_cbf_m(k) : (int)((T + 4) < k ? (T + 4) : k);
    if ((flags & 2) && (flags & 16) && len >= 65536) {
        /* the pair pre-pass on a read of 65536 bases or more (a stretch's ends travel in 16 bits: nothing is deferred): both strands are
         * looked at and each is stepped to its own verdict (fin_prepass.hip, can_defer = false) */
        lz_chunks fch = {-1, -1}, vch = {-1, -1};
        lz_pre fp = {-1, -1}, vp = {-1, -1};
        if (s->ctr) s->ctr->strands += 2;
        if (!lz_look(s, q, len, T, PM, flags, &fch, &fp)) while (fp.t0 >= 0 && !lz_pstep(s, q, len, T, PM, flags, &fp)) {}
        if (!lz_look(s, rcbuf, len, T, PM, flags, &vch, &vp)) while (vp.t0 >= 0 && !lz_pstep(s, rcbuf, len, T, PM, flags, &vp)) {}
        if (vp.t0 >= 0) lz_strand(s, rcbuf, len, out, 1, T, J, flags, &vp);
        if (fp.t0 >= 0) lz_strand(s, q, len, out, 0, T, J, flags, &fp);
    } else
    if ((flags & 16) && (flags & 2) && len < 65536) {
        lz_chunks fch = {-1, -1}, vch = {-1, -1};
        lz_pre fp = {-1, -1}, vp = {-1, -1};
        int a = -1;   /* 0: A = forward, 1: A = reverse, -1: neither strand has a k-mer end left */
        int b_deferred = 1;
        if (s->ctr) s->ctr->strands += 2;
        /* 32 <= k <= 63: the fast path comes FIRST, with looks of its own (the two-word anchor table says nothing about the pipeline's verdicts) */
        if ((flags & 64) && (flags & 8) && k >= 32 && k <= 63 && !s->lean && s->cbf && lz_fast_read(s, q, rcbuf, len, out, T, flags, 0, 0, -1, -1, 1)) a = -2;
        if (a == -2) {}
        else if (lz_look(s, q, len, T, PM, flags, &fch, &fp)) a = 0;
        else if (lz_look(s, rcbuf, len, T, PM, flags, &vch, &vp)) a = 1;
        /* the fast path (flags bit 6: with the k-mer table's looks): a read it finishes is done -- nothing below runs for it */
        if (a != -2 && (flags & 64) && (flags & 8) && k <= 63 && s->cbf && lz_fast_read(s, q, rcbuf, len, out, T, flags, a == 0, a == 1, fp.node, vp.node, 0)) a = -2;
        if (a == -2 || a >= 0) {}
        else {
            /* neither first k-mer is there: steps of the two strands in turn until a string occurs */
            int f_alive = fp.t0 >= 0, v_alive = vp.t0 >= 0;
            while (a < 0 && (f_alive || v_alive)) {   /* (a step of both strands at once -- their loads run side by side on the device; both occur: forward first) */
                const int fo = f_alive ? lz_pstep(s, q, len, T, PM, flags, &fp) : 0;
                const int vo = v_alive ? lz_pstep(s, rcbuf, len, T, PM, flags, &vp) : 0;
                f_alive = fp.t0 >= 0; v_alive = vp.t0 >= 0;
                if (fo) a = 0; else if (vo) a = 1;
            }
            b_deferred = a == 0 ? v_alive : a == 1 ? f_alive : 0;   /* a strand without a k-mer end left is absent, not deferred */
        }
        if (a == 1 && fp.t0 < 0) b_deferred = 0;
        if (a >= 0) {
            s->tainted = 0;
            lz_strand(s, a == 0 ? q : rcbuf, len, out, a, T, J, flags, a == 0 ? &fp : &vp);
            const int a_tainted = s->tainted;
            if (b_deferred) {
                if (s->ctr) s->ctr->deferred_strands++;
                int64_t lo = 0, hi = nk - 1;
                if (!a_tainted) {   /* (a tainted A: every slot counts as open) */
                    while (lo < nk && out[2 * lo] != -1) lo++;
                    while (hi >= lo && out[2 * hi] != -1) hi--;
                }
                if (lo <= hi) {
                    if (s->ctr) s->ctr->deferred_slots += hi - lo + 1;
                    /* B from the first open slot's k-mer TO THE READ'S END: its probes, look-ups and comparisons stop at the stretch's end
                     * (s->stop), a walk that is under way there runs on -- into slots A filled.  B reverse only fills, so nothing changes there;
                     * B FORWARD WINS them, as the reference's forward search does (search_fmin.hh:54-60): with duplicated unitigs its walk may
                     * follow a text that does not spell the read's k-mers (an unverified place, common.hh:61-67 / FinimizerIndex.hh:47-102) and
                     * report pairs where A found the true place -- the merged answer is the forward one.  (Round 3 searched B as a sub-read
                     * that ENDS at the stretch: 1 index set in 1 700 differed from the faithful restatement) */
                    s->stop = hi - lo + k - 1;
                    if (a == 0) lz_strand(s, rcbuf + (nk - 1 - hi), hi + k, out, 1, T, J, flags | 0x10000 | 0x20000, NULL);   /* B = reverse: fills open slots only */
                    else lz_strand(s, q + lo, len - lo, out + 2 * lo, 0, T, J, flags | 0x10000, NULL);                        /* B = forward: its pairs win */
                    s->stop = -1;
                }
            }
        }
    } else {
        lz_strand(s, rcbuf, len, out, 1, T, J, flags, NULL);
        lz_strand(s, q, len, out, 0, T, J, flags, NULL);
    }
    int64_t pos = 0;
    for (int64_t i = 0; i < nk; i++) pos += out[2 * i] != -1;
    if (positives) *positives += pos;
    if (s->ctr) {
        s->ctr->reads++; s->ctr->kmers += nk; s->ctr->found += pos; s->ctr->bases += len;
        s->ctr->chunks_packed += 2 * ((len + 31) / 32);
    }
    return nk;
}

static void lz_ctr_add(fo_lazy_counters* a, const fo_lazy_counters* b) {
    int64_t* pa = (int64_t*)a; const int64_t* pb = (const int64_t*)b;
    for (size_t i = 0; i < sizeof(fo_lazy_counters) / 8; i++) pa[i] += pb[i];
}

int64_t fo_search_batch_lazy(const fo_index* x, const char* bases, const uint64_t* offsets, int64_t n_reads, int64_t* pairs_out,
                             int ptab_t, int jump_t, int flags, int n_threads, fo_lazy_counters* ctr) {
    const int64_t k = x->k;
    if (ptab_t < 0) ptab_t = 0;
    if (ptab_t > k) ptab_t = (int)k;
    int64_t maxlen = 0;
    int64_t* out_off = (int64_t*)malloc((size_t)(n_reads + 1) * 8);
    out_off[0] = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        int64_t l = (int64_t)(offsets[r + 1] - offsets[r]);
        if (l > maxlen) maxlen = l;
        out_off[r + 1] = out_off[r] + (l >= k ? l - k + 1 : 0);
    }
    if (n_threads < 1) n_threads = 1;
    fo_lazy_counters* tctr = (fo_lazy_counters*)calloc((size_t)n_threads, sizeof(fo_lazy_counters));
    unsigned char* rcwin = (flags & 32) ? (unsigned char*)calloc((size_t)(x->total_len / 64 + 2), 1) : NULL;
    const int lean = (flags & 128) && (flags & 2) && (flags & 8);   /* (any k since round 5: the device's walk kernel asks the k-mer table above 63 too) */
    if (lean) ptab_t = 0;   /* (no prefix table: a probe is an exact occurrence question) */
    lz_cbf* cbf = ((flags & 64) && (flags & 16) && (flags & 8) && (flags & 2) && k <= 63) ? lz_cbf_build(x) : NULL;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num(); nt = omp_get_num_threads();
#endif
        lz_state s; memset(&s, 0, sizeof s);
        s.x = x; s.dq_cap = (int)(2 * k + 8); s.dq = (lz_cand*)malloc((size_t)s.dq_cap * sizeof(lz_cand));
        s.ctr = ctr ? &tctr[tid] : NULL;
        s.rcwin = rcwin; s.stop = -1; s.cbf = cbf; s.lean = lean;
        int64_t nk_max = maxlen - k + 1; if (nk_max < 0) nk_max = 0;
        int64_t* tmp = (int64_t*)malloc((size_t)(2 * nk_max + 2) * 8);
        char* rc = (char*)malloc((size_t)maxlen + 1);
        int64_t lo = n_reads * tid / nt, hi = n_reads * (tid + 1) / nt;
        for (int64_t r = lo; r < hi; r++) {
            const int64_t len = (int64_t)(offsets[r + 1] - offsets[r]);
            lz_read(&s, bases + offsets[r], len, rc, pairs_out ? pairs_out + 2 * out_off[r] : tmp, ptab_t, jump_t, flags, NULL);
        }
        free(tmp); free(rc); free(s.dq);
    }
    if (ctr) for (int t = 0; t < n_threads; t++) lz_ctr_add(ctr, &tctr[t]);
    const int64_t total = out_off[n_reads];
    free(tctr); free(out_off); free(rcwin); lz_cbf_free(cbf);
    return total;
}

int fo_index_is_disjoint(const fo_index* x) {
    int64_t places = 0, prev = 0;
    for (int64_t u = 0; u < x->n_unitigs; u++) {
        const int64_t e = (int64_t)iv_get(&x->ends, u);
        if (e - prev >= x->k) places += e - prev - x->k + 1;
        prev = e;
    }
    return x->n_kmers == places;
}

int fo_index_rc_free(const fo_index* x) {
    const int64_t k = x->k;
    char* buf = (char*)malloc((size_t)k + 1);
    int64_t prev = 0; int ok = 1;
    for (int64_t u = 0; u < x->n_unitigs && ok; u++) {
        const int64_t e = (int64_t)iv_get(&x->ends, u);
        for (int64_t g = prev + k - 1; g < e && ok; g++) {
            for (int64_t j = 0; j < k; j++) buf[j] = "TGCA"[lz_text_code(x, g - j)];   /* reverse complement of the k-mer that ends at g */
            if (lz_occurs(x, buf, 0, (int)k)) ok = 0;
        }
        prev = e;
    }
    free(buf);
    return ok;
}

/* 1: for every k-mer of the text the place the reference reports for it (lz_kmer_answer) spells that k-mer -- every pair the reference
 * can report is then a k-mer of the index at that place (anchors are such places, walks go on from them base by base).  A disjoint set
 * has the property; a set with duplicated k-mers may or may not (a finimizer's stored place can lie in another copy's surroundings). */
int fo_index_all_verified(const fo_index* x) {
    const int64_t k = x->k;
    char* lab = (char*)malloc((size_t)k + 1);
    int64_t prev = 0; int ok = 1;
    for (int64_t u = 0; u < x->n_unitigs && ok; u++) {
        const int64_t e = (int64_t)iv_get(&x->ends, u);
        for (int64_t g = prev + k - 1; g < e && ok; g++) {
            for (int64_t j = 0; j < k; j++) lab[j] = "ACGT"[lz_text_code(x, g - (k - 1) + j)];
            const int64_t G = lz_kmer_answer(x, lab);
            if (G < 0 || !lz_text_spells(x, lab, G)) ok = 0;
        }
        prev = e;
    }
    free(lab);
    return ok;
}
