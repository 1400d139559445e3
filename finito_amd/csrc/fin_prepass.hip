// fin_prepass.hip -- the pair pre-pass of kernel 4 (merged searches with an anchor table: FinDevIndex::defer_ok): which strand of a read is searched first?
//
// A k-mer that one strand of a read reports at a place that spells it is in the index, so -- unless the index holds its reverse complement
// too -- the other strand's k-mer in that slot is not: one strand -- A -- is searched and its sister only between the first and the last
// slot A left open (verdict FIN_PASS_DEFERRED; the walk kernel searches the sister in full where A's reports do not prove that much:
// fin_kernel_w.hip "tainted", CHANGELOG.md 4.14).  WHICH strand is A is a matter of cost only; this kernel decides it read by read and, as
// fin_probe_kernel does, proves the k-mer ends in front of A's first anchor absent:
//   look   is the strand's first k-mer in the index?  k <= 31 with the k-mer table: one slot of that table, which also names the k-mer's
//          node (the seed); else one probe step at k-1.  The forward strand is asked first, the reverse strand only if it fails
//   step   one step of a strand's probing at k-mer end t0 (probe_step): the absence filter, then the string of PM bases that ends at t0
//          -- a prefix-table entry and up to PM - T extends.  It occurs: the strand is A (verdict t0, seed = the one node the string
//          ends).  It does not: every k-mer that contains it is absent, t0 moves behind them
//   A read whose looks both fail (a sequencing error in its first 31 bases, or in its last) alternates steps of the two strands until
//   one string occurs; the other strand is deferred if it has a k-mer end left, else absent.
// Plain SIMT code, not an epoch state machine (fin_kernel_v3.hip): the work per read is short and the same for nearly every read, and the
// state machine's lanes sat in different states -- rocprofv3 counted 15 active lanes per vector instruction in it, and the kernel was
// bound by instruction issue, not by memory (profiles/r03).  Here a block takes a segment of reads: every thread LOOKS at its reads in
// lockstep (three dependent loads); the reads whose looks fail -- about a quarter -- are collected in LDS and shared out again, so that
// the stepping loop runs with full waves too.
#include <cstdio>
#include <cstring>

#include "fin_device.h"
#include "fin_kernels.h"

#ifndef FIN_V3_PM_ADD
#define FIN_V3_PM_ADD 4      // (as in fin_kernel_v3.hip: probe length = prefix-table depth + this)
#endif
#define FIN_PP_SEG_MAX 1024  // reads per block at most (the LDS lists of a block's reads that go on to later phases)

namespace {
constexpr uint32_t NONE = 0xFFFFFFFFu;

struct PpConsts {
    const char* blk_base; const FinPrefixIval* ptab; const uint32_t* filt;
    const FinCbfBlock* fbf; uint32_t fbf_mask;   // lean tables: a probe string is asked of the directional string filter (PM = its string length)
    uint32_t n, C0, C1, C2, C3, C4, fmask;
    int k, PT, PM, F;
};

// 32 bases of a strand from position p on, from its chunks c0 (chunk ci0, holds p) and c1 (chunk ci0 + 1; = c0 when not needed)
__device__ __forceinline__ void pp_window(const uint4& c0, const uint4& c1, int p, uint64_t& w, uint32_t& v) {
    const uint32_t j = (uint32_t)p & 31u;
    const uint64_t b0 = c0.x | ((uint64_t)c0.y << 32), b1 = c1.x | ((uint64_t)c1.y << 32);
    w = b0 >> (2 * j); v = c0.z >> j;
    if (j) { w |= b1 << (64 - 2 * j); v |= c1.z << (32 - j); }
}

// lean tables: does the directional string filter know the PM bases `w` (first base in the low bits)?
__device__ __forceinline__ bool pp_fbf_knows(const PpConsts& K, uint64_t w) {
    const uint64_t f = w & (K.PM >= 32 ? ~0ull : ((1ull << (2 * K.PM)) - 1ull));
    const uint64_t h = fin_cbf_hash(f);
    const uint4 b = *(const uint4*)(K.fbf + ((h >> 35) & K.fbf_mask));
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
#pragma unroll
    for (int i = 0; i < FIN_CBF_BITS; i++) {
        const uint32_t pb = (uint32_t)(h >> (7 * i)) & 127u, bit = 1u << (pb & 31u);
        m0 |= (pb >> 5) == 0u ? bit : 0u; m1 |= (pb >> 5) == 1u ? bit : 0u; m2 |= (pb >> 5) == 2u ? bit : 0u; m3 |= (pb >> 5) == 3u ? bit : 0u;
    }
    return (b.x & m0) == m0 && (b.y & m1) == m1 && (b.z & m2) == m2 && (b.w & m3) == m3;
}

// One step of a strand's probing at k-mer end t0 (< r_len).  true: the string of PM bases that ends at t0 occurs, node = the one node
// it ends (NONE: several).  false: t0 = the first k-mer end not proven absent (NONE: none left).
__device__ __forceinline__ bool probe_step(const PpConsts& K, const uint4* chunks, uint32_t r_len, uint32_t& t0, uint32_t& node) {
    const int k = K.k;
    const int span = max(K.PM - 1, K.F);               // (<= 31: the bases asked lie in two chunks at most)
    const int p0 = (int)t0 - span;
    const int ci0 = p0 >> 5, ci1 = (int)t0 >> 5;
    const uint4 c0 = chunks[ci0];
    uint4 c1 = c0;
    if (ci1 != ci0) c1 = chunks[ci1];
    // a window from p on, p0 <= p <= t0 (it may begin in the second chunk)
    auto window = [&](int p, uint64_t& w, uint32_t& v) { if ((p >> 5) == ci0) pp_window(c0, c1, p, w, v); else pp_window(c1, c1, p, w, v); };
    if (K.F > 0) {
        // absence filter: the strings of F bases that end at t0 and at t0 - 1; one that occurs in no unitig rules out every k-mer with it
        uint64_t w; uint32_t v;
        window((int)t0 - K.F, w, v);
        const uint32_t inv = ~v;
        const uint32_t fi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
        if (fi > (uint32_t)K.F) {
            const uint32_t key0 = (uint32_t)w & K.fmask, key1 = (uint32_t)(w >> 2) & K.fmask;
            const uint32_t w1 = K.filt[key1 >> 5], w0 = K.filt[key0 >> 5];
            uint32_t adv = 0;
            if (!((w1 >> (key1 & 31u)) & 1u)) adv = (uint32_t)(k - K.F + 1);
            else if (!((w0 >> (key0 & 31u)) & 1u)) adv = (uint32_t)(k - K.F);
            if (adv) { t0 += adv; if (t0 >= r_len) t0 = NONE; return false; }
        }
    }
    const int p = (int)t0 - K.PM + 1;
    uint64_t w; uint32_t v;
    window(p, w, v);
    const uint32_t inv = ~v;
    const uint32_t pfi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;   // the first non-ACGT base of the string, if any
    if (K.fbf) {   // lean tables: one load; a string that occurs names no node (the walk kernel looks the whole k-mer up)
        if (pfi >= (uint32_t)K.PM && pp_fbf_knows(K, w)) { node = NONE; return true; }
        t0 = (uint32_t)(p + k);
        if (t0 >= r_len) t0 = NONE;
        return false;
    }
    uint32_t il = 0, ir = K.n - 1;
    int off = 0;
    bool ok = true;
    if (K.PT > 0) {
        if (pfi < (uint32_t)K.PT) ok = false;
        else {
            const FinPrefixIval iv = K.ptab[(uint32_t)w & ((1u << (2 * K.PT)) - 1u)];
            il = iv.l; ir = iv.r; off = K.PT;
            ok = il <= ir;
        }
    }
    // update_sbwt_interval (common.hh:26-36) on the records of the two blocks
    while (ok && off < K.PM) {
        if ((uint32_t)off >= pfi) { ok = false; break; }
        const uint32_t c = (uint32_t)(w >> (2 * off)) & 3u;
        if (il == 0 && ir == K.n - 1) {
            const uint32_t m0 = 0u - (uint32_t)(c == 0), m1 = 0u - (uint32_t)(c == 1), m2 = 0u - (uint32_t)(c == 2), m3 = 0u - (uint32_t)(c == 3);
            il = (K.C0 & m0) | (K.C1 & m1) | (K.C2 & m2) | (K.C3 & m3);
            ir = ((K.C1 & m0) | (K.C2 & m1) | (K.C3 & m2) | (K.C4 & m3)) - 1;
            ok = il <= ir;
        } else {
            const FinCharRec a = *(const FinCharRec*)(K.blk_base + (size_t)(il >> 6) * 128 + 64 + 12 * c);
            const FinCharRec b = *(const FinCharRec*)(K.blk_base + (size_t)(ir >> 6) * 128 + 64 + 12 * c);
            const uint64_t pa = a.plane_lo | ((uint64_t)a.plane_hi << 32), pb = b.plane_lo | ((uint64_t)b.plane_hi << 32);
            const uint32_t nl = a.base + (uint32_t)__popcll(pa & ~(~0ull << (il & 63u)));
            const uint32_t re = b.base + (uint32_t)__popcll(pb & (~0ull >> (63 - (ir & 63u))));
            ok = nl < re;
            il = nl; ir = re - 1;
        }
        off++;
    }
    if (ok) { node = il == ir ? il : NONE; return true; }
    t0 = (uint32_t)(p + k);   // every k-mer that contains the string is absent
    if (t0 >= r_len) t0 = NONE;
    return false;
}

// One step of BOTH strands' probing at once (the stepping loop): the two strands' chains of dependent loads -- chunks, filter words, table entry,
// up to PM - T pairs of rank records -- run side by side instead of one behind the other; what each load brings is used only after both
// strands' loads of that stage are on their way.  go[s]: strand s takes part; on return ok[s] = its string occurs (node[s]), else t0[s] moved on.
__device__ __forceinline__ void probe_step2(const PpConsts& K, const uint4* const chunks[2], uint32_t r_len, const bool go[2], uint32_t t0[2], uint32_t node[2], bool ok[2]) {
    if (K.fbf) {   // lean tables: a step is one filter block per strand
#pragma unroll
        for (int s = 0; s < 2; s++) { ok[s] = false; if (go[s]) ok[s] = probe_step(K, chunks[s], r_len, t0[s], node[s]); }
        return;
    }
    const int k = K.k;
    const int span = max(K.PM - 1, K.F);
    int ci0[2], ci1[2]; uint4 c0[2], c1[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        ok[s] = false;
        const uint32_t t = go[s] ? t0[s] : (uint32_t)span;   // (a strand that sits out loads its first chunk: harmless, and the code stays branch-free)
        ci0[s] = ((int)t - span) >> 5; ci1[s] = (int)t >> 5;
        c0[s] = chunks[s][ci0[s]];
        c1[s] = chunks[s][ci1[s]];
    }
    auto window = [&](int s, int p, uint64_t& w, uint32_t& v) { if ((p >> 5) == ci0[s]) pp_window(c0[s], ci1[s] != ci0[s] ? c1[s] : c0[s], p, w, v); else pp_window(c1[s], c1[s], p, w, v); };
    bool probe[2] = {go[0], go[1]};
    if (K.F > 0) {
        uint32_t key0[2] = {0, 0}, key1[2] = {0, 0}, w0[2] = {0, 0}, w1[2] = {0, 0}; bool ask[2] = {false, false};
#pragma unroll
        for (int s = 0; s < 2; s++) if (go[s]) {
            uint64_t w; uint32_t v;
            window(s, (int)t0[s] - K.F, w, v);
            const uint32_t inv = ~v;
            const uint32_t fi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
            if (fi > (uint32_t)K.F) { ask[s] = true; key0[s] = (uint32_t)w & K.fmask; key1[s] = (uint32_t)(w >> 2) & K.fmask; w1[s] = K.filt[key1[s] >> 5]; w0[s] = K.filt[key0[s] >> 5]; }
        }
#pragma unroll
        for (int s = 0; s < 2; s++) if (ask[s]) {
            uint32_t adv = 0;
            if (!((w1[s] >> (key1[s] & 31u)) & 1u)) adv = (uint32_t)(k - K.F + 1);
            else if (!((w0[s] >> (key0[s] & 31u)) & 1u)) adv = (uint32_t)(k - K.F);
            if (adv) { t0[s] += adv; if (t0[s] >= r_len) t0[s] = NONE; probe[s] = false; }
        }
    }
    int p[2] = {0, 0}; uint64_t w[2] = {0, 0}; uint32_t pfi[2] = {0, 0}, il[2] = {0, 0}, ir[2] = {0, 0}; bool alive[2] = {false, false};
    FinPrefixIval iv[2] = {{1u, 0u}, {1u, 0u}};
#pragma unroll
    for (int s = 0; s < 2; s++) if (probe[s]) {
        p[s] = (int)t0[s] - K.PM + 1;
        uint32_t v;
        window(s, p[s], w[s], v);
        const uint32_t inv = ~v;
        pfi[s] = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
        il[s] = 0; ir[s] = K.n - 1; alive[s] = true;
        if (K.PT > 0) { if (pfi[s] < (uint32_t)K.PT) alive[s] = false; else iv[s] = K.ptab[(uint32_t)w[s] & ((1u << (2 * K.PT)) - 1u)]; }
    }
    if (K.PT > 0) {
#pragma unroll
        for (int s = 0; s < 2; s++) if (alive[s]) { il[s] = iv[s].l; ir[s] = iv[s].r; alive[s] = il[s] <= ir[s]; }
    }
    for (int off = K.PT > 0 ? K.PT : 0; off < K.PM && (alive[0] || alive[1]); off++) {
        FinCharRec a[2], b[2]; uint32_t c[2] = {0, 0}; bool full[2] = {false, false};
#pragma unroll
        for (int s = 0; s < 2; s++) if (alive[s]) {
            if ((uint32_t)off >= pfi[s]) { alive[s] = false; continue; }
            c[s] = (uint32_t)(w[s] >> (2 * off)) & 3u;
            full[s] = il[s] == 0 && ir[s] == K.n - 1;
            if (!full[s]) {
                a[s] = *(const FinCharRec*)(K.blk_base + (size_t)(il[s] >> 6) * 128 + 64 + 12 * c[s]);
                b[s] = *(const FinCharRec*)(K.blk_base + (size_t)(ir[s] >> 6) * 128 + 64 + 12 * c[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; s++) if (alive[s]) {
            if (full[s]) {
                const uint32_t cc = c[s];
                const uint32_t m0 = 0u - (uint32_t)(cc == 0), m1 = 0u - (uint32_t)(cc == 1), m2 = 0u - (uint32_t)(cc == 2), m3 = 0u - (uint32_t)(cc == 3);
                il[s] = (K.C0 & m0) | (K.C1 & m1) | (K.C2 & m2) | (K.C3 & m3);
                ir[s] = ((K.C1 & m0) | (K.C2 & m1) | (K.C3 & m2) | (K.C4 & m3)) - 1;
                alive[s] = il[s] <= ir[s];
            } else {
                const uint64_t pa = a[s].plane_lo | ((uint64_t)a[s].plane_hi << 32), pb = b[s].plane_lo | ((uint64_t)b[s].plane_hi << 32);
                const uint32_t nl = a[s].base + (uint32_t)__popcll(pa & ~(~0ull << (il[s] & 63u)));
                const uint32_t re = b[s].base + (uint32_t)__popcll(pb & (~0ull >> (63 - (ir[s] & 63u))));
                alive[s] = nl < re;
                il[s] = nl; ir[s] = re - 1;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; s++) if (probe[s]) {
        if (alive[s]) { ok[s] = true; node[s] = il[s] == ir[s] ? il[s] : NONE; }
        else { t0[s] = (uint32_t)(p[s] + k); if (t0[s] >= r_len) t0[s] = NONE; }
    }
}

// The look through the k-mer table (FinDevIndex::kt3): does the table CLAIM the k-mer {k0, k1}?  One bucket of four slots {g, tag} per load; a tag match
// is a claim -- g_ans = the reference's answer for the k-mer, verified = the text there spells it -- that only a comparison with the text at g_ans
// proves (the fast path's whole-read comparison; the walk kernel's W_REANCH for the pipeline's place items).  No match up to the first empty slot:
// the k-mer is in no unitig.
__device__ __forceinline__ bool kt3_find_h(const FinDevIndex& ix, uint64_t h, uint32_t& g_ans, bool& verified) {
    const uint32_t tag = (uint32_t)h & FIN_KT3_TAGMASK;
    uint32_t b = fin_kt3_bucket(h, ix.kt3_buckets);
    for (;;) {
        const uint4* const p = (const uint4*)(ix.kt3 + b);
        const uint4 s0 = p[0], s1 = p[1];
        const uint32_t g[4] = {s0.x, s0.z, s1.x, s1.z}, m[4] = {s0.y, s0.w, s1.y, s1.w};
#pragma unroll
        for (int j = 0; j < FIN_KT3_SLOTS; j++) {
            if (m[j] == 0xFFFFFFFFu) return false;   // an empty slot: the chain ends
            if ((m[j] & FIN_KT3_TAGMASK) == tag) { g_ans = g[j]; verified = !(m[j] & FIN_KT3_UNVER); return true; }
        }
        b = b + 1u == ix.kt3_buckets ? 0u : b + 1u;   // the bucket is full of other k-mers: this one may have gone to the next
    }
}
// A PLACE seed of the pipeline is the look's claim about a strand's FIRST k-mer: "it lies at the text place that ends at g".  The walk kernel starts a run at a
// place item without asking again (W_DESC, fin_kernel_w.hip), so the claim is compared with the text here -- one or two 16-byte windows, and only for the
// reads the fast path does not finish (its own comparison covers the others).  false: another k-mer's tag -- the strand becomes a probe item, whose look-up
// meets the claim again and hands the read to kernel 3.
__device__ __forceinline__ bool pp_claim_holds(const FinDevIndex& ix, const uint4* ch, uint32_t g) {
    const uint32_t k = ix.k;
    if (g < k - 1u || g >= ix.total_len) return false;
    if (k >= 64u) {   // a wide key (lean tables above 63, round 5): 32 bases at a time -- the strand's chunk j against the text behind the place
        const uint32_t gs0 = g - (k - 1u);
        for (uint32_t j = 0; 32u * j < k; j++) {
            const uint32_t nb = k - 32u * j < 32u ? k - 32u * j : 32u, gp = gs0 + 32u * j, o2 = gp & 63u;
            const uint4* const t2 = (const uint4*)ix.concat + (gp >> 6);
            const uint4 wa2 = t2[0];
            uint4 wb2 = wa2;
            if (o2 + nb > 64u) wb2 = t2[1];
            uint64_t y0, y1;
            fin_text_kmer(wa2, wb2, o2, nb, y0, y1);
            const uint4 c = ch[j];
            uint64_t qj = c.x | ((uint64_t)c.y << 32);
            if (nb < 32u) qj &= (1ull << (2u * nb)) - 1ull;
            if (y0 != qj) return false;
        }
        return true;
    }
    const uint32_t gs = g - (k - 1u), o = gs & 63u;
    const uint4* const tw = (const uint4*)ix.concat + (gs >> 6);
    const uint4 wa = tw[0];
    uint4 wb = wa;
    if (o + k > 64u) wb = tw[1];
    uint64_t x0, x1;
    fin_text_kmer(wa, wb, o, k, x0, x1);
    const uint4 c0 = ch[0];
    uint64_t q0 = c0.x | ((uint64_t)c0.y << 32), q1 = 0ull;
    if (k < 32u) q0 &= (1ull << (2u * k)) - 1ull;
    if (k > 32u) { const uint4 c1 = ch[1]; q1 = (c1.x | ((uint64_t)c1.y << 32)) & ((1ull << (2u * (k - 32u))) - 1ull); }
    return x0 == q0 && x1 == q1;
}
__device__ __forceinline__ bool kt3_find(const FinDevIndex& ix, uint64_t k0, uint64_t k1, uint32_t& g_ans, bool& verified) { return kt3_find_h(ix, fin_kt3_hash(k0, k1), g_ans, verified); }
// k <= 32: the strand's first k-mer (c0: its first chunk)
__device__ __forceinline__ bool look_ktab(const PpConsts& K, const FinDevIndex& ix, const uint4& c0, uint32_t& g_ans, bool& verified) {
    const uint32_t need = K.k == 32 ? 0xFFFFFFFFu : (1u << K.k) - 1u;
    if ((c0.z & need) != need) return false;   // a non-ACGT base: no k-mer
    const uint64_t key = (c0.x | ((uint64_t)c0.y << 32)) & (K.k == 32 ? ~0ull : (1ull << (2 * K.k)) - 1ull);
    return kt3_find(ix, key, 0ull, g_ans, verified);
}

// ... and the k-mer that ENDS at position t of a strand (its bases lie in one or two chunks)
__device__ __forceinline__ bool look_ktab_at(const PpConsts& K, const FinDevIndex& ix, const uint4* ch, uint32_t t, uint32_t& g_ans, bool& verified) {
    const uint32_t p = t - (uint32_t)(K.k - 1), j0 = p >> 5, j1 = t >> 5, o = p & 31u;
    const uint4 a = ch[j0];
    uint4 c;
    if (j1 == j0) {
        const uint64_t wa = a.x | ((uint64_t)a.y << 32);
        const uint64_t w = wa >> (2 * o);
        c.x = (uint32_t)w; c.y = (uint32_t)(w >> 32); c.z = a.z >> o; c.w = 0;
    } else {
        const uint4 b = ch[j1];
        const uint64_t wa = a.x | ((uint64_t)a.y << 32), wb = b.x | ((uint64_t)b.y << 32);
        const uint64_t w = (wa >> (2 * o)) | (wb << (64 - 2 * o));   // (o > 0: the k-mer spans two chunks)
        c.x = (uint32_t)w; c.y = (uint32_t)(w >> 32); c.z = (a.z >> o) | (b.z << (32 - o)); c.w = 0;
    }
    return look_ktab(K, ix, c, g_ans, verified);
}

// 33 <= k <= 63: the k-mer that ends at position t of a strand -- two key words; its bases lie in up to three chunks
__device__ __forceinline__ bool look_ktab2_at(const FinDevIndex& ix, const uint4* ch, uint32_t t, uint32_t r_len, uint32_t& g_ans, bool& verified) {
    const uint32_t k = ix.k, p = t - (k - 1u), j0 = p >> 5, o = p & 31u, jl = (r_len - 1u) >> 5;
    const uint4 a = ch[j0], b = ch[j0 + 1u <= jl ? j0 + 1u : jl], c = ch[j0 + 2u <= jl ? j0 + 2u : jl];
    const uint64_t wa = a.x | ((uint64_t)a.y << 32), wb = b.x | ((uint64_t)b.y << 32), wc = c.x | ((uint64_t)c.y << 32);
    const uint64_t w0 = o ? (wa >> (2 * o)) | (wb << (64 - 2 * o)) : wa, w1 = o ? (wb >> (2 * o)) | (wc << (64 - 2 * o)) : wb;
    const uint32_t v0 = o ? (a.z >> o) | (b.z << (32 - o)) : a.z, v1 = o ? (b.z >> o) | (c.z << (32 - o)) : b.z;
    const uint32_t n1 = k - 32u;                       // bases in the second word
    const uint32_t need1 = n1 ? (1u << n1) - 1u : 0u;  // (n1 <= 31)
    if (v0 != 0xFFFFFFFFu || (v1 & need1) != need1) return false;   // a non-ACGT base: no k-mer
    return kt3_find(ix, w0, n1 ? w1 & ((1ull << (2 * n1)) - 1ull) : 0ull, g_ans, verified);
}

// 64 <= k <= 255: the k-mer that ends at position t of a strand, its ceil(k / 32) key words folded into the hash as they are made from the strand's chunks (the
// table holds no k-mer, so none is kept here either: the fast path's comparison of the whole read with the text is what proves the claim).  (The walk kernel
// folds the same words one epoch each as its chunk cache brings them, fin_kernel_w.hip W_KF0B.)
__device__ __forceinline__ bool look_ktabN_at(const FinDevIndex& ix, const uint4* ch, uint32_t t, uint32_t r_len, uint32_t& g_ans, bool& verified) {
    const uint32_t k = ix.k, p = t - (k - 1u), j0 = p >> 5, o = p & 31u, jl = (r_len - 1u) >> 5, nw = (k + 31u) >> 5;
    uint4 a = ch[j0];
    uint64_t key = 0;
    for (uint32_t w = 0; w < nw; w++) {
        const uint4 b = ch[j0 + w + 1u <= jl ? j0 + w + 1u : jl];
        const uint64_t wa = a.x | ((uint64_t)a.y << 32), wb = b.x | ((uint64_t)b.y << 32);
        uint64_t word = o ? (wa >> (2u * o)) | (wb << (64u - 2u * o)) : wa;
        const uint32_t valid = o ? (a.z >> o) | (b.z << (32u - o)) : a.z;
        const uint32_t nb = k - 32u * w < 32u ? k - 32u * w : 32u;
        const uint32_t need = nb == 32u ? 0xFFFFFFFFu : (1u << nb) - 1u;
        if ((valid & need) != need) return false;   // a non-ACGT base: no k-mer
        if (nb < 32u) word &= (1ull << (2u * nb)) - 1ull;
        key = w == 0u ? word : fin_kt3_fold(key, word, w);
        a = b;
    }
    return kt3_find_h(ix, fin_mix64(key), g_ans, verified);
}

// ---- the FAST PATH (round 4): a whole read against one unitig's text, in plain SIMT code --------------------------------------------
// The common read comes from one place of the indexed text and carries a few substitution errors.  Once the look has found the first k-mer
// of strand A in the k-mer table -- with the reference's answer G for it, a place where the text spells it -- everything the reference
// reports for the read follows from ONE comparison of the read with the text behind that place:
//   * a k-mer end t whose k-mer holds no disagreeing base: the k-mer is in the text there.  The reference reaches it by its walk
//     (walk_in_unitigs, FinimizerIndex.hh:47-102: one base at a time along the unitig) from the k-mer before it, or -- the first k-mer behind a
//     disagreeing base -- by its dictionaries, which answer that place iff the place is SAFE (FinDevIndex::safe; every place of a disjoint set is);
//   * a k-mer end t whose k-mer holds a disagreeing base E: absent iff proven so -- by a string of cbf_m bases inside the k-mer that the
//     CANONICAL string filter does not know (FinDevIndex::cbf): then neither strand's k-mer in that slot is in the index.  Two or three such
//     strings settle the k ends around E, for both strands: the sister strand needs no search at all, for the slots A fills hold k-mers
//     whose reverse complements are not in the index (no reverse-complement window flagged, FinDevIndex::rcwin -- the deferral's own argument,
//     CHANGELOG.md 4.14), and its k-mers in A's open slots contain the reverse complements of strings the filter does not know.
// A read that does not fit -- its look fails, the answer is unverified, the unitig ends inside the read, a non-ACGT base, more than
// FIN_FAST_MAXE disagreeing bases, a string the filter knows (or takes for known: it has false positives, never false negatives), an unsafe
// place, a flagged window, more than FIN_FAST_CHUNKS chunks -- keeps the verdict the look gave it and goes the pipeline's way, untouched.
#define FIN_FAST_MAXE 4
#ifdef FIN_PP_STATS   // diagnostic build: why reads leave the fast path (fin_debug_dump_pp)
__device__ unsigned long long g_fin_ppdbg[16];
#define PPDBG(i) atomicAdd(&g_fin_ppdbg[i], 1ull)
#else
#define PPDBG(i) ((void)0)
#endif
#define FIN_FAST_CHUNKS 8      // chunks of strand A kept in LDS for the strings (reads of up to 256 bases)
struct FastRun { uint32_t ok, u, off0, nE; uint64_t Es, Es2; };   // Es, Es2: the disagreeing positions, 16 bits each, ascending (four in each word)

__device__ __forceinline__ uint64_t pp_revcomp(uint64_t f, uint32_t m, uint64_t mask) {
    uint64_t r = __brevll(f);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    return (r >> (64u - 2u * m)) ^ mask;
}
// the canonical string filter's block of q[a .. a+m-1] and the bits the string sets in it (lds: the lane's chunk codes, stride FIN_TPB)
struct CbfAsk { uint4 blk; uint32_t mw[4]; };
__device__ __forceinline__ void cbf_ask(const FinDevIndex& ix, const uint64_t* lds, uint32_t a, uint32_t m, uint64_t mask, CbfAsk& q) {
    const uint32_t j = a >> 5, o = a & 31u;
    uint64_t w = lds[j * FIN_TPB] >> (2 * o);
    if (o + m > 32u) w |= lds[(j + 1) * FIN_TPB] << (64u - 2u * o);
    const uint64_t f = w & mask, v = pp_revcomp(f, m, mask);
    const uint64_t h = fin_cbf_hash(f < v ? f : v);
    q.blk = *(const uint4*)(ix.cbf + ((h >> 35) & ((1ull << ix.cbf_log2) - 1ull)));
    q.mw[0] = q.mw[1] = q.mw[2] = q.mw[3] = 0u;
#pragma unroll
    for (int i = 0; i < FIN_CBF_BITS; i++) {
        const uint32_t p = (uint32_t)(h >> (7 * i)) & 127u, bit = 1u << (p & 31u);
        q.mw[0] |= (p >> 5) == 0u ? bit : 0u; q.mw[1] |= (p >> 5) == 1u ? bit : 0u; q.mw[2] |= (p >> 5) == 2u ? bit : 0u; q.mw[3] |= (p >> 5) == 3u ? bit : 0u;
    }
}
__device__ __forceinline__ bool cbf_known(const CbfAsk& q) {
    return (q.blk.x & q.mw[0]) == q.mw[0] && (q.blk.y & q.mw[1]) == q.mw[1] && (q.blk.z & q.mw[2]) == q.mw[2] && (q.blk.w & q.mw[3]) == q.mw[3];
}
// Strand A's chunks `ch`, the k-mer the look found ends at position t_anchor of the strand (k-1: its first k-mer), g_ans = the reference's
// answer for it (verified).  true: res describes every slot of the read.
// (written for memory-level parallelism: a lane's loads of one stage -- the read's chunks and the text beside them; the strings across one
//  disagreeing base -- are issued together and used afterwards; nothing in a stage returns early)
__device__ __forceinline__ bool fast_try(const PpConsts& K, const FinDevIndex& ix, const uint4* ch, uint32_t t_anchor, uint32_t r_len, uint32_t g_ans, uint64_t* lds, FastRun& res) {
    const uint32_t k = (uint32_t)K.k, m = ix.cbf_m;
    const uint32_t nch = (r_len + 31u) >> 5;
    if (nch > FIN_FAST_CHUNKS || g_ans < t_anchor) { PPDBG(1); return false; }
    const uint32_t gs = g_ans - t_anchor;           // the text position of the read's first base
    uint32_t u, ustart, uend;
    {   // PackedStrings::global_offset_to_local_offset (PackedStrings.hh:91-100) through the sampled ends
        uint32_t idx = ix.samp[gs >> ix.samp_shift];
        uint4 e4; __builtin_memcpy(&e4, ix.ends + idx, 16);   // ends_p[idx .. idx+3] (the array ends with eight 0xFFFFFFFF)
        while (e4.w <= gs) { idx += 3u; __builtin_memcpy(&e4, ix.ends + idx, 16); }
        if (gs < e4.y) { u = idx; ustart = e4.x; uend = e4.y; }
        else if (gs < e4.z) { u = idx + 1u; ustart = e4.y; uend = e4.z; }
        else { u = idx + 2u; ustart = e4.z; uend = e4.w; }
    }
    if (gs < ustart || gs + r_len > uend) { PPDBG(2); return false; }   // the unitig ends inside the read: the pipeline's business
    // ---- the comparison: the read's chunks and the text beside them, four chunks at a time ----
    const uint64_t* const text = (const uint64_t*)ix.concat;
    const uint32_t tw = gs >> 5, sh = (gs & 31u) * 2u;
    uint64_t Es = 0, Es2 = 0; uint32_t nE = 0; bool bad = false;
    auto E_at = [&](uint32_t e) -> uint32_t { return (uint32_t)((e < 4u ? Es : Es2) >> (16u * (e & 3u))) & 0xFFFFu; };
    uint64_t w0 = text[tw];
    for (uint32_t jb = 0; jb < nch; jb += 4u) {
        uint4 c[4]; uint64_t tx[4];
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) {
            const uint32_t j = jb + i < nch ? jb + i : nch - 1u;   // (past the end: the last chunk again -- a load, no branch)
            c[i] = ch[j]; tx[i] = text[tw + j + 1u];
        }
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) {
            const uint32_t j = jb + i;
            if (j < nch) {
                const uint64_t w1 = tx[i];
                const uint64_t tb = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
                w0 = w1;
                const uint64_t rb = c[i].x | ((uint64_t)c[i].y << 32);
                lds[j * FIN_TPB] = rb;
                const uint32_t nb = r_len - 32u * j < 32u ? r_len - 32u * j : 32u;
                const uint32_t vm = nb == 32u ? 0xFFFFFFFFu : (1u << nb) - 1u;
                bad = bad || (c[i].z & vm) != vm;         // a non-ACGT base
                const uint64_t x = rb ^ tb;
                uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
                if (nb < 32u) y &= (1ull << (2u * nb)) - 1ull;
                while (y && nE < FIN_FAST_MAXE) {
                    const uint32_t p = (uint32_t)(__ffsll((long long)y) - 1) >> 1;
                    y &= y - 1ull;
                    if (nE < 4u) Es |= (uint64_t)(32u * j + p) << (16u * nE); else Es2 |= (uint64_t)(32u * j + p) << (16u * (nE - 4u));
                    nE++;
                }
                bad = bad || y != 0ull;                    // more than FIN_FAST_MAXE disagreeing bases
            }
        }
    }
    if (bad) { PPDBG(3); return false; }
    // ---- reverse-complement windows (indexes that hold a k-mer and its reverse complement): a report from a flagged window proves nothing about the sister ----
    if (ix.rcwin) {
        for (uint32_t w = (gs + k - 1u) >> 6; w <= (gs + r_len - 1u) >> 6; w++) if ((ix.rcwin[w >> 3] >> (w & 7u)) & 1u) { PPDBG(5); return false; }
    }
    // (an anchor that is not the first k-mer: when the first k-mer holds no disagreeing base it is reported where the text has it only if that place is safe)
    if (ix.safe && t_anchor != k - 1u && (nE == 0u || E_at(0) >= k) && !((ix.safe[(gs + k - 1u) >> 6] >> ((gs + k - 1u) & 63u)) & 1ull)) { PPDBG(7); return false; }
    // ---- the k-mer ends across the disagreeing bases: absent on both strands iff the filter does not know a string inside each ----
    // A string q[a .. a+m-1] that holds E lies inside every k-mer that ends in [a+m-1, a+k-1]: from the first unsettled end lo on, strings at
    // a = min(lo-m+1, E), then k-m+1 further on each time -- three at most for the k ends around E (m <= k, 3 (k-m+1) >= k for m = min(k, 20), k <= 31)
    const uint64_t mask = m >= 32u ? ~0ull : ((1ull << (2u * m)) - 1ull);
    uint32_t covered = k - 2u;   // every k-mer end up to here is settled
    bool known = false, unsafe = false;
    for (uint32_t e = 0; e < nE; e++) {
        const uint32_t E = E_at(e);
        uint32_t lo = E > k - 1u ? E : k - 1u; if (lo < covered + 1u) lo = covered + 1u;
        const uint32_t hi = E + k - 1u < r_len - 1u ? E + k - 1u : r_len - 1u;
        CbfAsk q0, q1, q2; bool h0 = false, h1 = false, h2 = false;
        if (lo <= hi) { const uint32_t a = lo - (m - 1u) < E ? lo - (m - 1u) : E; cbf_ask(ix, lds, a, m, mask, q0); h0 = true; covered = a + k - 1u; lo = covered + 1u; }
        if (lo <= hi) { const uint32_t a = lo - (m - 1u) < E ? lo - (m - 1u) : E; cbf_ask(ix, lds, a, m, mask, q1); h1 = true; covered = a + k - 1u; lo = covered + 1u; }
        if (lo <= hi) { const uint32_t a = lo - (m - 1u) < E ? lo - (m - 1u) : E; cbf_ask(ix, lds, a, m, mask, q2); h2 = true; covered = a + k - 1u; lo = covered + 1u; }
        // the first k-mer behind this stretch of absent ends is reported where the text has it only if that place is safe
        const uint32_t t = E + k;
        const bool last = e + 1u == nE || E_at(e + 1u) > t;
        unsigned long long sw = ~0ull;
        if (ix.safe && last && t < r_len) sw = ix.safe[(gs + t) >> 6];
        known = known || (h0 && cbf_known(q0)) || (h1 && cbf_known(q1)) || (h2 && cbf_known(q2)) || lo <= hi;   // (lo <= hi: three strings did not reach -- m far below k; not with the default m)
        unsafe = unsafe || !((sw >> ((gs + t) & 63u)) & 1ull);
    }
    if (known) { PPDBG(6); return false; }
    if (unsafe) { PPDBG(7); return false; }
    PPDBG(0);
    res.ok = 1u; res.u = u; res.off0 = gs - ustart; res.nE = nE; res.Es = Es; res.Es2 = Es2;
    return true;
}
// A read none of whose looks found a k-mer: is EVERY k-mer of it absent, on both strands?  Strings of m bases that end at the first unsettled
// k-mer end, one after the other (each settles k-m+1 ends), asked of the canonical string filter -- a read from nowhere takes about
// len / (k-m+1) loads, all of them independent, where the pipeline proves each strand by itself with a prefix-table entry and node blocks per
// string.  false: a string the filter knows (or a non-ACGT base): the pipeline decides.
__device__ __forceinline__ bool fast_all_absent(const PpConsts& K, const FinDevIndex& ix, const uint4* ch, uint32_t r_len, uint64_t* lds) {
    const uint32_t k = (uint32_t)K.k, m = ix.cbf_m;
    const uint32_t nch = (r_len + 31u) >> 5;
    if (nch > FIN_FAST_CHUNKS) return false;
    bool bad = false;
    for (uint32_t j = 0; j < nch; j++) {
        const uint4 c = ch[j];
        const uint32_t nb = r_len - 32u * j < 32u ? r_len - 32u * j : 32u;
        const uint32_t vm = nb == 32u ? 0xFFFFFFFFu : (1u << nb) - 1u;
        bad = bad || (c.z & vm) != vm;
        lds[j * FIN_TPB] = c.x | ((uint64_t)c.y << 32);
    }
    if (bad) return false;
    const uint64_t mask = m >= 32u ? ~0ull : ((1ull << (2u * m)) - 1ull);
    bool known = false;
    for (uint32_t t = k - 1u; t < r_len; ) {   // (four strings' loads at a time)
        CbfAsk q[4]; bool h[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { h[i] = t < r_len; const uint32_t a = (h[i] ? t : k - 1u) - (m - 1u); cbf_ask(ix, lds, a, m, mask, q[i]); if (h[i]) t = a + k; }
#pragma unroll
        for (int i = 0; i < 4; i++) known = known || (h[i] && cbf_known(q[i]));
        if (known) break;
    }
    if (known) { PPDBG(10); return false; }
    PPDBG(11);
    return true;
}
}  // namespace

// defer = 0 (a read of 65536 bases or more -- a stretch's ends travel in 16 bits --; the launcher passes 1 whenever the run defers second strands,
// which since round 3 is any index with an anchor table: taints and window flags make it exact, fin_kernel_w.hip): both strands are looked at,
// and each is stepped to its own verdict.
// KT2 (with FAST): the fast path's looks go to the k-mer table but the PIPELINE's verdicts do not come from them -- k >= 33 (this flow's looks reach
// any k-mer end through up to three chunks), or round 3's tables (seeds are nodes, which the compact table does not hold): verdicts come from probe
// steps as before, made afterwards and only for the reads the fast path did not finish (list L); under lean tables 2 (k >= 33) the looks are the verdicts
template <bool FAST, bool KT2>
__device__ __forceinline__ void fin_pair_prepass_body(const FinDevIndex& ix, const uint4* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t seg,
                                                      uint32_t* pass, uint32_t* seed, int defer, int2* out, uint32_t* n_fast) {
    // lists of reads (numbers inside the block's segment).  lds_list: from the front, list A -- both first looks failed, the fast path goes on
    // with other k-mers of the read (phase 2) --; from the back, the stepping loop's reads (phase 4).  lds_b: list B, phase 3.
    __shared__ uint16_t lds_list[FIN_PP_SEG_MAX];
    __shared__ uint16_t lds_b[FAST ? FIN_PP_SEG_MAX : 1];
    // list L: reads the fast path's later phases did not finish.  KT2: their verdicts are made by phase L.  Else they join the stepping loop's list --
    // once phases 2 and 3 are over: pushed to the back of lds_list while those phases still read list A from its front, they overwrote unread
    // entries of A as soon as n_a + pushes passed FIN_PP_SEG_MAX (every read of a 1024-read segment missing both first looks)
    __shared__ uint16_t lds_l[FAST ? FIN_PP_SEG_MAX : 1];
    __shared__ uint32_t lds_n, lds_na, lds_nb, lds_nl;
    __shared__ uint64_t lds_ck[FAST ? FIN_FAST_CHUNKS * FIN_TPB : 1];   // the fast path: strand A's chunk codes, per lane
    PpConsts K;
    K.blk_base = (const char*)ix.blocks; K.ptab = ix.ptab; K.filt = ix.filt;
    K.n = ix.n_nodes; K.C0 = ix.C[0]; K.C1 = ix.C[1]; K.C2 = ix.C[2]; K.C3 = ix.C[3]; K.C4 = ix.C[4];
    K.k = (int)ix.k; K.PT = (int)ix.ptab_t; K.PM = min(K.PT + FIN_V3_PM_ADD, K.k);
    K.fbf = ix.fbf; K.fbf_mask = ix.fbf ? (uint32_t)((1ull << ix.cbf_log2) - 1ull) : 0u;
    if (K.fbf) K.PM = (int)ix.cbf_m;
    K.F = ix.filt ? (int)ix.filt_f : 0;
    K.fmask = K.F ? (K.F == 16 ? 0xFFFFFFFFu : (1u << (2 * K.F)) - 1u) : 0u;
    // (this flow's looks ARE the pipeline's verdicts and its seeds are places: lean tables.  With round 3's tables seeds are nodes, which the compact k-mer
    //  table does not hold: the KT2 flow -- the table for the fast path, verdicts by probe steps -- or, without the fast path, probe steps alone)
    const bool look_kt = ix.kt3 != nullptr && K.k <= 32 && K.fbf != nullptr;
    const uint32_t k1 = (uint32_t)(K.k - 1);
    // a strand's slot of pass[] between the two loops: k-1 = its look succeeded (final), NONE = absent (final), FIN_PASS_DEFERRED (final),
    // anything else = the k-mer end its stepping starts at (>= k: a failed look proves end k-1 absent)
    auto is_final = [&](uint32_t v) { return v == k1 || v == NONE || v == FIN_PASS_DEFERRED || v == FIN_PASS_DONE; };

    if (threadIdx.x == 0) { lds_n = 0; lds_na = 0; lds_nb = 0; lds_nl = 0; }
    __syncthreads();
    const uint32_t r_lo = blockIdx.x * seg, r_hi = r_lo + seg < n_reads ? r_lo + seg : n_reads;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t fast_done = 0;
    // the reads a wave's lanes have finished: their slots, written by the whole wave read by read -- a pair where no disagreeing base lies inside
    // the k-mer, (-1,-1) where one does.  Wave-converged.
    auto write_out = [&](const FastRun& fr, bool fr_rev, uint32_t r, uint32_t out_off, uint32_t r_len) {
        uint64_t mdone = __ballot(fr.ok != 0u);
        fast_done += (uint32_t)__popcll(mdone);
        if (ix.frec) {   // text modes: the read's record, for fin_text.hip; text only: its pairs are never materialised
            if (fr.ok) {
                uint4* const q = (uint4*)(ix.frec + r);
                q[0] = make_uint4(fr.u, fr.off0, fr.nE | ((uint32_t)fr_rev << 8) | (fr.ok << 16), r_len - k1);
                q[1] = make_uint4((uint32_t)fr.Es, (uint32_t)(fr.Es >> 32), (uint32_t)fr.Es2, (uint32_t)(fr.Es2 >> 32));
            }
            if (ix.text_only) return;
        }
        while (mdone) {
            const int src = __ffsll((long long)mdone) - 1;
            mdone &= mdone - 1ull;
            auto lane_of = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
            const uint32_t o_base = lane_of(out_off), o_nk = lane_of(r_len) - k1;
            const uint32_t p_u = lane_of(fr.u), p_off = lane_of(fr.off0), p_nE = lane_of(fr.nE), p_rev = lane_of((uint32_t)fr_rev), p_ok = lane_of(fr.ok);
            const uint32_t e_lo = lane_of((uint32_t)fr.Es), e_hi = lane_of((uint32_t)(fr.Es >> 32)), f_lo = lane_of((uint32_t)fr.Es2), f_hi = lane_of((uint32_t)(fr.Es2 >> 32));
            const uint32_t E0 = e_lo & 0xFFFFu, E1 = e_lo >> 16, E2 = e_hi & 0xFFFFu, E3 = e_hi >> 16, E4 = f_lo & 0xFFFFu, E5 = f_lo >> 16, E6 = f_hi & 0xFFFFu, E7 = f_hi >> 16;
            for (uint32_t i = lane; i < o_nk; i += 64u) {
                const uint32_t sl = p_rev ? o_nk - 1u - i : i;   // the slot in strand A's own order: its k-mer is A[sl .. sl+k-1]
                bool gap = p_ok == 2u;   // (2: every k-mer of the read is absent)
                if (p_nE > 0u) gap = gap || (E0 - sl <= k1);     // (unsigned: false when E < sl too)
                if (p_nE > 1u) gap = gap || (E1 - sl <= k1);
                if (p_nE > 2u) gap = gap || (E2 - sl <= k1);
                if (p_nE > 3u) gap = gap || (E3 - sl <= k1);
                if (p_nE > 4u) gap = gap || (E4 - sl <= k1) || (p_nE > 5u && E5 - sl <= k1) || (p_nE > 6u && E6 - sl <= k1) || (p_nE > 7u && E7 - sl <= k1);
                const int2 val = gap ? make_int2(-1, -1) : make_int2((int)p_u, (int)(p_off + sl));
                __builtin_nontemporal_store(*(const unsigned long long*)&val, (unsigned long long*)&out[(size_t)o_base + i]);
            }
        }
    };
    // the fast path's look at the k-mer that ends at position t of a strand: found (hit), its answer g, whether the text there spells it
    auto flook = [&](const uint4* ch, uint32_t t, uint32_t r_len, uint32_t& g, bool& ver) -> bool {
        if (K.k >= 64) return look_ktabN_at(ix, ch, t, r_len, g, ver);
        if (K.k >= 33) return look_ktab2_at(ix, ch, t, r_len, g, ver);
        return t == k1 ? look_ktab(K, ix, ch[0], g, ver) : look_ktab_at(K, ix, ch, t, g, ver);
    };
    auto to_tail = [&](uint32_t r) { lds_list[FIN_PP_SEG_MAX - 1u - atomicAdd(&lds_n, 1u)] = (uint16_t)(r - r_lo); };   // (phase 1, phase L and behind: the two ends of lds_list never meet -- a read is in one of them)
    auto to_l = [&](uint32_t r) { lds_l[atomicAdd(&lds_nl, 1u)] = (uint16_t)(r - r_lo); };     // not finished by the fast path (KT2: its verdicts are still to be made)
    // the pipeline's verdicts of a read by probe steps (no k-mer table for this k): look at the forward strand, the reverse one only if that fails
    auto probe_looks = [&](uint32_t r, uint32_t r_len, const uint4* cf, const uint4* cv) {
        uint2 verdict = make_uint2(NONE, NONE), sd = make_uint2(NONE, NONE);
        if (r_len >= (uint32_t)K.k) {
            const bool can_defer = defer && r_len < 65536u;
            uint32_t f_t0 = k1, v_t0 = k1;
            const bool f_hit = probe_step(K, cf, r_len, f_t0, sd.x);
            if (f_hit && can_defer) v_t0 = FIN_PASS_DEFERRED;
            else {
                const bool v_hit = probe_step(K, cv, r_len, v_t0, sd.y);
                if (v_hit && can_defer && f_t0 != NONE) f_t0 = FIN_PASS_DEFERRED;
            }
            verdict = make_uint2(f_t0, v_t0);
        }
        *(uint2*)(pass + 2 * (size_t)r) = verdict;
        if (seed) *(uint2*)(seed + 2 * (size_t)r) = sd;
        if (!is_final(verdict.x) || !is_final(verdict.y)) to_tail(r);
    };
    // ---- phase 1, every read of the segment: the looks at the strands' first k-mers; the fast path where one is found ----
    for (uint32_t rb = r_lo; rb < r_hi; rb += FIN_TPB) {   // (whole waves: the fast path's write-out is the wave's)
        const uint32_t r = rb + threadIdx.x;
        FinReadDesc d = {0, 0, 0};
        if (r < r_hi) d = desc[r];
        const uint32_t r_len = d.len, r_nch = (r_len + 31u) >> 5;
        const uint4* const cf = packed + d.off, *const cv = cf + r_nch;
        uint2 verdict = make_uint2(NONE, NONE), sd = make_uint2(NONE, NONE);
        FastRun fr = {0u, 0u, 0u, 0u, 0ull, 0ull}; bool fr_rev = false, to_a = false;
        if (KT2) {
            if (r < r_hi) {
                bool hit = false, settled = false;
                if (r_len >= (uint32_t)K.k && defer && r_len < 65536u) {
                    uint32_t g_f = NONE, g_v = NONE; bool ver_f = false, ver_v = false;
                    const bool f_hit = flook(cf, k1, r_len, g_f, ver_f);
                    const bool v_hit = !f_hit && flook(cv, k1, r_len, g_v, ver_v);
                    hit = f_hit || v_hit;
                    if (((f_hit && ver_f) || (v_hit && ver_v)) && fast_try(K, ix, v_hit ? cv : cf, k1, r_len, v_hit ? g_v : g_f, lds_ck + threadIdx.x, fr)) fr_rev = v_hit;
                    if (fr.ok) *(uint2*)(pass + 2 * (size_t)r) = make_uint2(FIN_PASS_DONE, FIN_PASS_DONE);
                    else if (!hit) lds_list[atomicAdd(&lds_na, 1u)] = (uint16_t)(r - r_lo);
                    else if (K.fbf) {
                        // lean tables: the table's look IS the pipeline's verdict (a k-mer it has is in the index), and a verified answer is the
                        // strand's seed -- a PLACE; nothing is left for the probe steps of phase L
                        const uint32_t other = (uint32_t)K.k < r_len ? FIN_PASS_DEFERRED : NONE;   // (a strand without a k-mer end behind its first is absent, not deferred)
                        *(uint2*)(pass + 2 * (size_t)r) = f_hit ? make_uint2(k1, FIN_PASS_DEFERRED) : make_uint2(other, k1);
                        if (seed) *(uint2*)(seed + 2 * (size_t)r) = make_uint2((f_hit && ver_f && pp_claim_holds(ix, cf, g_f)) ? g_f : NONE, (v_hit && ver_v && pp_claim_holds(ix, cv, g_v)) ? g_v : NONE);
                        settled = true;
                    }
                }
                if (!fr.ok && !settled && (hit || !(r_len >= (uint32_t)K.k && defer && r_len < 65536u))) to_l(r);
            }
            write_out(fr, fr_rev, r, d.out_off, r_len);
            continue;
        }
        if (r < r_hi && r_len >= (uint32_t)K.k) {
            const bool can_defer = defer && r_len < 65536u;
            const uint32_t after = (uint32_t)K.k < r_len ? (uint32_t)K.k : NONE;   // where a strand goes on when the table does not have its first k-mer
            uint32_t f_t0 = k1, v_t0 = k1;
            if (look_kt) {
                uint32_t g_f = NONE, g_v = NONE; bool ver_f = false, ver_v = false, v_hit = false;
                const bool f_hit = look_ktab(K, ix, cf[0], g_f, ver_f);
                sd.x = (f_hit && ver_f) ? g_f : NONE;   // lean tables: a seed is the k-mer's verified PLACE (no anchor table to ask a node's); unverified: a probe item
                if (!f_hit) f_t0 = after;
                if (f_hit && can_defer) v_t0 = FIN_PASS_DEFERRED;   // A = forward; the reverse strand is not looked at
                else {
                    v_hit = look_ktab(K, ix, cv[0], g_v, ver_v);
                    sd.y = (v_hit && ver_v) ? g_v : NONE;
                    if (!v_hit) v_t0 = after;
                    if (v_hit && can_defer && f_t0 != NONE) f_t0 = FIN_PASS_DEFERRED;   // A = reverse (a forward strand without an end left is absent)
                }
                if (FAST && can_defer) {
                    // ONE attempt per read, for whichever strand's look succeeded (the lanes of a wave run it together)
                    const bool af = f_hit && ver_f, av = !f_hit && v_hit && ver_v;
                    if ((af || av) && fast_try(K, ix, av ? cv : cf, k1, r_len, av ? g_v : g_f, lds_ck + threadIdx.x, fr)) fr_rev = av;
                    if ((f_hit && !ver_f) || (!f_hit && v_hit && !ver_v)) PPDBG(8);
                    // neither first k-mer is in the index (a sequencing error in the first 31 bases of the strand that matches -- the other
                    // strand's k-mers are not in the table at all -- or a read from nowhere): phase 2 looks at other k-mers of the read
                    to_a = !f_hit && !v_hit;
                    if (to_a) PPDBG(9);
                }
            } else {
                const bool f_hit = probe_step(K, cf, r_len, f_t0, sd.x);
                if (f_hit && can_defer) v_t0 = FIN_PASS_DEFERRED;
                else {
                    const bool v_hit = probe_step(K, cv, r_len, v_t0, sd.y);
                    if (v_hit && can_defer && f_t0 != NONE) f_t0 = FIN_PASS_DEFERRED;
                }
            }
            if (look_kt && !fr.ok) {   // the place seeds this read takes into the pipeline: claims, compared with the text
                if (sd.x != NONE && !pp_claim_holds(ix, cf, sd.x)) sd.x = NONE;
                if (sd.y != NONE && !pp_claim_holds(ix, cv, sd.y)) sd.y = NONE;
            }
            verdict = fr.ok ? make_uint2(FIN_PASS_DONE, FIN_PASS_DONE) : make_uint2(f_t0, v_t0);
        }
        if (r < r_hi) {
            *(uint2*)(pass + 2 * (size_t)r) = verdict;
            if (seed) *(uint2*)(seed + 2 * (size_t)r) = sd;
            if (to_a) lds_list[atomicAdd(&lds_na, 1u)] = (uint16_t)(r - r_lo);
            else if (!is_final(verdict.x) || !is_final(verdict.y)) lds_list[FIN_PP_SEG_MAX - 1u - atomicAdd(&lds_n, 1u)] = (uint16_t)(r - r_lo);
        }
        if (FAST) write_out(fr, fr_rev, r, d.out_off, r_len);
    }
    __syncthreads();
    if (FAST) {
        // ---- phase 2, list A: the strands' LAST k-mers (verdicts and seeds stay what the first looks made them: a read the fast path does not
        //      finish goes to the stepping loop as before).  A look that finds a k-mer settles the attempt, whichever way it ends; none: list B ----
        const uint32_t n_a = lds_na;
        for (uint32_t ib = 0; ib < n_a; ib += FIN_TPB) {
            const uint32_t i = ib + threadIdx.x;
            const bool on = i < n_a;
            const uint32_t r = r_lo + (on ? lds_list[i] : 0u);
            FinReadDesc d = {0, 0, 0};
            if (on) d = desc[r];
            const uint32_t r_len = d.len, r_nch = (r_len + 31u) >> 5;
            const uint4* const cf = packed + d.off, *const cv = cf + r_nch;
            FastRun fr = {0u, 0u, 0u, 0u, 0ull, 0ull}; bool fr_rev = false;
            if (on) {
                bool hit = false, to_b = true;
                const uint32_t t = r_len - 1u;
                if (t > k1) {
                    uint32_t g_f = NONE, g_v = NONE; bool ver_f = false, ver_v = false;
                    const bool f_hit = flook(cf, t, r_len, g_f, ver_f);
                    const bool v_hit = !f_hit && flook(cv, t, r_len, g_v, ver_v);
                    hit = f_hit || v_hit; to_b = !hit;
                    const bool af = f_hit && ver_f, av = v_hit && ver_v;
                    if ((af || av) && fast_try(K, ix, av ? cv : cf, t, r_len, av ? g_v : g_f, lds_ck + threadIdx.x, fr)) fr_rev = av;
                }
                if (fr.ok) *(uint2*)(pass + 2 * (size_t)r) = make_uint2(FIN_PASS_DONE, FIN_PASS_DONE);
                else if (to_b) lds_b[atomicAdd(&lds_nb, 1u)] = (uint16_t)(r - r_lo);
                else to_l(r);
            }
            write_out(fr, fr_rev, r, d.out_off, r_len);
        }
        __syncthreads();
        // ---- phase 3, list B: the strands' MIDDLE k-mers; no k-mer found anywhere: every k-mer of the read absent on both strands, if the
        //      canonical string filter knows none of the strings across it ----
        const uint32_t n_b = lds_nb;
        for (uint32_t ib = 0; ib < n_b; ib += FIN_TPB) {
            const uint32_t i = ib + threadIdx.x;
            const bool on = i < n_b;
            const uint32_t r = r_lo + (on ? lds_b[i] : 0u);
            FinReadDesc d = {0, 0, 0};
            if (on) d = desc[r];
            const uint32_t r_len = d.len, r_nch = (r_len + 31u) >> 5;
            const uint4* const cf = packed + d.off, *const cv = cf + r_nch;
            FastRun fr = {0u, 0u, 0u, 0u, 0ull, 0ull}; bool fr_rev = false;
            if (on) {
                bool hit = false;
                // (the middle k-mer, in both strands.  More positions, and more than FIN_FAST_MAXE = 4 disagreeing bases, were measured: they finish
                //  more reads -- chr1 93.3 instead of 90.7 %, k = 63 86.7 instead of 76.7 % -- and cost the pre-pass more than the pipeline gets back:
                //  a wave waits for its lane with the most disagreeing bases)
                for (uint32_t w = 0; w < 1u && !hit; w++) {
                    const uint32_t t = (r_len + (uint32_t)K.k) / 2u - 1u;
                    if (t <= k1 || t >= r_len - 1u) continue;
                    uint32_t g_f = NONE, g_v = NONE; bool ver_f = false, ver_v = false;
                    const bool f_hit = flook(cf, t, r_len, g_f, ver_f);
                    const bool v_hit = !f_hit && flook(cv, t, r_len, g_v, ver_v);
                    hit = f_hit || v_hit;
                    const bool af = f_hit && ver_f, av = v_hit && ver_v;
                    if ((af || av) && fast_try(K, ix, av ? cv : cf, t, r_len, av ? g_v : g_f, lds_ck + threadIdx.x, fr)) fr_rev = av;
                }
                if (!hit && fast_all_absent(K, ix, cf, r_len, lds_ck + threadIdx.x)) { fr.ok = 2u; fr.nE = 0u; }
                if (fr.ok) *(uint2*)(pass + 2 * (size_t)r) = make_uint2(FIN_PASS_DONE, FIN_PASS_DONE);
                else to_l(r);
            }
            write_out(fr, fr_rev, r, d.out_off, r_len);
        }
        if (n_fast && lane == 0 && fast_done) atomicAdd(n_fast, fast_done);
        __syncthreads();
        if (KT2) {
            // ---- phase L (32 <= k <= 63): the pipeline's verdicts, by probe steps, of the reads the fast path did not finish ----
            const uint32_t n_l = lds_nl;
            for (uint32_t i = threadIdx.x; i < n_l; i += FIN_TPB) {
                const uint32_t r = r_lo + lds_l[i];
                const FinReadDesc d = desc[r];
                const uint4* const cf = packed + d.off;
                probe_looks(r, d.len, cf, cf + ((d.len + 31u) >> 5));
            }
            __syncthreads();
        } else {
            // list L joins the stepping loop's reads (list A is dead now; a read is in one list only: lds_n + n_l <= seg)
            const uint32_t n_l = lds_nl, n_t = lds_n;
            for (uint32_t i = threadIdx.x; i < n_l; i += FIN_TPB) lds_list[FIN_PP_SEG_MAX - 1u - (n_t + i)] = lds_l[i];
            __syncthreads();
            if (threadIdx.x == 0) lds_n = n_t + n_l;
            __syncthreads();
        }
    }
    // ---- the stepping loop: the reads with a strand whose look failed, shared out again ----
    const uint32_t n_tail = lds_n;
    for (uint32_t i = threadIdx.x; i < n_tail; i += FIN_TPB) {
        const uint32_t r = r_lo + lds_list[FIN_PP_SEG_MAX - 1u - i];
        const FinReadDesc d = desc[r];
        const uint32_t r_len = d.len, r_nch = (r_len + 31u) >> 5;
        const uint4* const cf = packed + d.off, *const cv = cf + r_nch;
        const bool can_defer = defer && r_len < 65536u;
        const uint2 at = *(const uint2*)(pass + 2 * (size_t)r);
        uint32_t f_t0 = at.x, v_t0 = at.y, f_node = NONE, v_node = NONE;
        const bool f_step = !is_final(f_t0), v_step = !is_final(v_t0);
        // a strand is done when its verdict stands: its string occurred (t0 = that end), it has no end left (NONE), or it is deferred
        bool f_done = !f_step, v_done = !v_step;
        const uint4* const ch[2] = {cf, cv};
        while (!(f_done && v_done)) {
            // a step of both strands at once (their loads side by side); when both strings occur the forward strand is searched first
            const bool go[2] = {!f_done, !v_done};
            uint32_t t[2] = {f_t0, v_t0}, nd[2] = {NONE, NONE}; bool okk[2];
            probe_step2(K, ch, r_len, go, t, nd, okk);
            if (go[0]) { f_t0 = t[0]; if (okk[0]) { f_node = nd[0]; f_done = true; } else f_done = f_t0 == NONE; }
            if (go[1]) { v_t0 = t[1]; if (okk[1]) { v_node = nd[1]; v_done = true; } else v_done = v_t0 == NONE; }
            if (can_defer) {
                if (go[0] && okk[0] && !v_done) { v_t0 = FIN_PASS_DEFERRED; v_done = true; }
                else if (go[0] && okk[0] && go[1] && okk[1]) { v_t0 = FIN_PASS_DEFERRED; v_node = NONE; }   // (both occur: the reverse strand's look-up is dropped)
                else if (go[1] && okk[1] && !f_done) { f_t0 = FIN_PASS_DEFERRED; f_done = true; }
            }
        }
        if (f_step) { pass[2 * (size_t)r] = f_t0; if (seed) seed[2 * (size_t)r] = f_node; }
        if (v_step) { pass[2 * (size_t)r + 1] = v_t0; if (seed) seed[2 * (size_t)r + 1] = v_node; }
    }
}

__global__ __launch_bounds__(FIN_TPB) void fin_pair_prepass_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t seg,
                                                                   uint32_t* pass, uint32_t* seed, int defer) {
    fin_pair_prepass_body<false, false>(ix, packed, desc, n_reads, seg, pass, seed, defer, nullptr, nullptr);
}
// ... with the fast path: the reads it finishes are written to `out` and get FIN_PASS_DONE
__global__ __launch_bounds__(FIN_TPB) void fin_fast_prepass_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t seg,
                                                                   uint32_t* pass, uint32_t* seed, int defer, int2* out, uint32_t* n_fast) {
    fin_pair_prepass_body<true, false>(ix, packed, desc, n_reads, seg, pass, seed, defer, out, n_fast);
}
// ... for 32 <= k <= 63 (two-word anchor table; verdicts by probe steps for what the fast path leaves)
__global__ __launch_bounds__(FIN_TPB) void fin_fast2_prepass_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t seg,
                                                                    uint32_t* pass, uint32_t* seed, int defer, int2* out, uint32_t* n_fast) {
    fin_pair_prepass_body<true, true>(ix, packed, desc, n_reads, seg, pass, seed, defer, out, n_fast);
}

// reads per block: whole iterations of the block's threads, FIN_PP_SEG_MAX at most; small batches get smaller segments so that the grid
// still fills the chip
extern "C" int fin_launch_pair_prepass(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t* pass, uint32_t* seed,
                                       int defer, uint32_t grid_hint, void* out, uint32_t* n_fast, hipStream_t stream) {
    if (n_reads == 0) return 0;
    uint32_t seg = (n_reads + grid_hint - 1) / (grid_hint ? grid_hint : 1u);
    seg = (seg + FIN_TPB - 1) / FIN_TPB * FIN_TPB;
    if (seg > FIN_PP_SEG_MAX) seg = FIN_PP_SEG_MAX;
    if (n_reads >= 512u * FIN_PP_SEG_MAX) seg = FIN_PP_SEG_MAX;   // (long segments keep the phases' lists full: measured on 1 M and 10 M reads, 1024 beats 768 / 512 / 256)
    if (ix->pp_seg >= FIN_TPB && ix->pp_seg <= FIN_PP_SEG_MAX && ix->pp_seg % FIN_TPB == 0) seg = ix->pp_seg;   // (option "debug_pp_seg": tests reach the longest segments with small batches)
    // the fast path: merged searches with a deferred strand on an index with the k-mer table (any k since round 5) and the canonical string filter.  Lean tables at
    // k <= 32: the looks are the verdicts (fin_fast_prepass_kernel); else they serve the fast path alone and probe steps make the verdicts (fin_fast2_...)
    if (out && defer && ix->kt3 && ix->cbf && ix->cbf_m >= 1 && ix->cbf_m <= ix->k && (ix->k >= 33 || !ix->fbf))
        hipLaunchKernelGGL(fin_fast2_prepass_kernel, dim3((n_reads + seg - 1) / seg), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, n_reads, seg, pass, seed, defer, (int2*)out, n_fast);
    else if (out && defer && ix->kt3 && ix->fbf && ix->cbf && ix->k <= 32 && ix->cbf_m >= 1 && ix->cbf_m <= ix->k)
        hipLaunchKernelGGL(fin_fast_prepass_kernel, dim3((n_reads + seg - 1) / seg), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, n_reads, seg, pass, seed, defer, (int2*)out, n_fast);
    else
        hipLaunchKernelGGL(fin_pair_prepass_kernel, dim3((n_reads + seg - 1) / seg), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, n_reads, seg, pass, seed, defer);
    return (int)hipGetLastError();
}

extern "C" void fin_debug_dump_pp(void) {
#ifdef FIN_PP_STATS
    unsigned long long h[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fin_ppdbg), sizeof h);
    fprintf(stderr, "[fin_ppdbg] fast ok %llu | too long / no answer %llu  unitig ends inside %llu  non-ACGT %llu  > %d errors %llu  rc window %llu  filter knows %llu  unsafe place %llu | unverified %llu  both first looks fail %llu | all-absent: a string known %llu  proven %llu\n",
            h[0], h[1], h[2], h[3], FIN_FAST_MAXE, h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_ppdbg), h, sizeof h);
#endif
}
