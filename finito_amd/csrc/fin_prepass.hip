// fin_prepass.hip -- the pair pre-pass of kernel 4 (merged searches with an anchor table: FinDevIndex::defer_ok): which strand of a read is searched first?
//
// A k-mer that one strand of a read reports at a place that spells it is in the index, so -- unless the index holds its reverse complement
// too -- the other strand's k-mer in that slot is not: one strand -- A -- is searched and its sister only between the first and the last
// slot A left open (verdict FIN_PASS_DEFERRED; the walk kernel searches the sister in full where A's reports do not prove that much:
// fin_kernel_w.hip "tainted").  WHICH strand is A is a matter of cost only; this kernel decides it read by read and, as
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
#include "fin_device.h"
#include "fin_kernels.h"

#ifndef FIN_V3_PM_ADD
#define FIN_V3_PM_ADD 4      // (as in fin_kernel_v3.hip: probe length = prefix-table depth + this)
#endif
#define FIN_PP_SEG_MAX 2048  // reads per block at most (the LDS list of a block's reads that go on to the stepping loop)

namespace {
constexpr uint32_t NONE = 0xFFFFFFFFu;

struct PpConsts {
    const char* blk_base; const FinPrefixIval* ptab; const uint32_t* filt; const FinKtabSlot* ktab;
    uint32_t n, C0, C1, C2, C3, C4, kt_mask, fmask;
    int k, PT, PM, F;
};

// 32 bases of a strand from position p on, from its chunks c0 (chunk ci0, holds p) and c1 (chunk ci0 + 1; = c0 when not needed)
__device__ __forceinline__ void pp_window(const uint4& c0, const uint4& c1, int p, uint64_t& w, uint32_t& v) {
    const uint32_t j = (uint32_t)p & 31u;
    const uint64_t b0 = c0.x | ((uint64_t)c0.y << 32), b1 = c1.x | ((uint64_t)c1.y << 32);
    w = b0 >> (2 * j); v = c0.z >> j;
    if (j) { w |= b1 << (64 - 2 * j); v |= c1.z << (32 - j); }
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

// The look through the k-mer table (k <= 31): is the strand's first k-mer in the index?  true: node = its SBWT node
__device__ __forceinline__ bool look_ktab(const PpConsts& K, const uint4* chunks, uint32_t& node) {
    const uint4 c0 = chunks[0];
    const uint32_t need = K.k == 32 ? 0xFFFFFFFFu : (1u << K.k) - 1u;
    if ((c0.z & need) != need) return false;   // a non-ACGT base: no k-mer
    const uint64_t key = (c0.x | ((uint64_t)c0.y << 32)) & ((1ull << (2 * K.k)) - 1ull);
    uint32_t slot = fin_ktab_hash(key) & K.kt_mask;
    for (;;) {
        const uint4 s = *(const uint4*)(K.ktab + slot);
        const uint64_t skey = s.x | ((uint64_t)s.y << 32);
        if (skey == key) { node = s.z; return true; }
        if (skey == FIN_KTAB_EMPTY) return false;
        slot = (slot + 1u) & K.kt_mask;   // another k-mer's slot: linear probing (the table is at most half full)
    }
}
}  // namespace

// defer = 0 (an index on which nothing may be deferred -- reverse-complement pairs, unsafe places -- or a read of 65536 bases or more: a
// stretch's ends travel in 16 bits): both strands are looked at, and each is stepped to its own verdict.
__global__ __launch_bounds__(FIN_TPB) void fin_pair_prepass_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t seg,
                                                                   uint32_t* pass, uint32_t* seed, int defer) {
    __shared__ uint32_t lds_tail[FIN_PP_SEG_MAX];
    __shared__ uint32_t lds_n;
    PpConsts K;
    K.blk_base = (const char*)ix.blocks; K.ptab = ix.ptab; K.filt = ix.filt; K.ktab = ix.ktab;
    K.n = ix.n_nodes; K.C0 = ix.C[0]; K.C1 = ix.C[1]; K.C2 = ix.C[2]; K.C3 = ix.C[3]; K.C4 = ix.C[4];
    K.k = (int)ix.k; K.PT = (int)ix.ptab_t; K.PM = min(K.PT + FIN_V3_PM_ADD, K.k);
    K.F = ix.filt ? (int)ix.filt_f : 0;
    K.fmask = K.F ? (K.F == 16 ? 0xFFFFFFFFu : (1u << (2 * K.F)) - 1u) : 0u;
    const bool look_kt = ix.ktab != nullptr && K.k <= 31;
    K.kt_mask = look_kt ? (1u << ix.ktab_log2) - 1u : 0u;
    const uint32_t k1 = (uint32_t)(K.k - 1);
    // a strand's slot of pass[] between the two loops: k-1 = its look succeeded (final), NONE = absent (final), FIN_PASS_DEFERRED (final),
    // anything else = the k-mer end its stepping starts at (>= k: a failed look proves end k-1 absent)
    auto is_final = [&](uint32_t v) { return v == k1 || v == NONE || v == FIN_PASS_DEFERRED; };

    if (threadIdx.x == 0) lds_n = 0;
    __syncthreads();
    const uint32_t r_lo = blockIdx.x * seg, r_hi = r_lo + seg < n_reads ? r_lo + seg : n_reads;
    // ---- the looks: every read of the segment ----
    for (uint32_t r = r_lo + threadIdx.x; r < r_hi; r += FIN_TPB) {
        const FinReadDesc d = desc[r];
        const uint32_t r_len = d.len, r_nch = (r_len + 31u) >> 5;
        const uint4* const cf = packed + d.off, *const cv = cf + r_nch;
        uint2 verdict = make_uint2(NONE, NONE), sd = make_uint2(NONE, NONE);
        if (r_len >= (uint32_t)K.k) {
            const bool can_defer = defer && r_len < 65536u;
            const uint32_t after = (uint32_t)K.k < r_len ? (uint32_t)K.k : NONE;   // where a strand goes on when the table does not have its first k-mer
            uint32_t f_t0 = k1, v_t0 = k1;
            const bool f_hit = look_kt ? look_ktab(K, cf, sd.x) : probe_step(K, cf, r_len, f_t0, sd.x);
            if (look_kt && !f_hit) f_t0 = after;
            if (f_hit && can_defer) v_t0 = FIN_PASS_DEFERRED;   // A = forward; the reverse strand is not looked at
            else {
                const bool v_hit = look_kt ? look_ktab(K, cv, sd.y) : probe_step(K, cv, r_len, v_t0, sd.y);
                if (look_kt && !v_hit) v_t0 = after;
                if (v_hit && can_defer && f_t0 != NONE) f_t0 = FIN_PASS_DEFERRED;   // A = reverse (a forward strand without an end left is absent)
            }
            verdict = make_uint2(f_t0, v_t0);
        }
        *(uint2*)(pass + 2 * (size_t)r) = verdict;
        if (seed) *(uint2*)(seed + 2 * (size_t)r) = sd;
        if (!is_final(verdict.x) || !is_final(verdict.y)) lds_tail[atomicAdd(&lds_n, 1u)] = r;
    }
    __syncthreads();
    // ---- the stepping loop: the reads with a strand whose look failed, shared out again ----
    const uint32_t n_tail = lds_n;
    for (uint32_t i = threadIdx.x; i < n_tail; i += FIN_TPB) {
        const uint32_t r = lds_tail[i];
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

// reads per block: whole iterations of the block's threads, FIN_PP_SEG_MAX at most; small batches get smaller segments so that the grid
// still fills the chip
extern "C" int fin_launch_pair_prepass(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t* pass, uint32_t* seed,
                                       int defer, uint32_t grid_hint, hipStream_t stream) {
    if (n_reads == 0) return 0;
    uint32_t seg = (n_reads + grid_hint - 1) / (grid_hint ? grid_hint : 1u);
    seg = (seg + FIN_TPB - 1) / FIN_TPB * FIN_TPB;
    if (seg > FIN_PP_SEG_MAX) seg = FIN_PP_SEG_MAX;
    hipLaunchKernelGGL(fin_pair_prepass_kernel, dim3((n_reads + seg - 1) / seg), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, n_reads, seg, pass, seed, defer);
    return (int)hipGetLastError();
}
