// fin_device.h -- device helpers shared by the gfx950 kernels (index primitives over the 128-B node blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fin_format.h"

#define FIN_TPB 256

// ---- device-side index primitives ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t d_nodebyte(const FinDevIndex& ix, uint32_t i) {
    return ((const uint8_t*)ix.blocks)[(size_t)(i >> 6) * sizeof(FinNodeBlock) + (i & 63)];
}
// (k > 128: the node byte holds min(LCS, 127), the exact value is in ix.lcs8)
__device__ __forceinline__ uint32_t d_lcs(const FinDevIndex& ix, uint32_t i) { return ix.lcs8 ? (uint32_t)ix.lcs8[i] : (d_nodebyte(ix, i) & FIN_LCS_MASK); }

// update_sbwt_interval (formula: common.hh:26-36) on [l, r]; false = (-1,-1)
__device__ __forceinline__ bool d_extend(const FinDevIndex& ix, uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) {
    const FinNodeBlock* bl = ix.blocks + (l >> 6);
    const FinNodeBlock* br = ix.blocks + (r >> 6);
    uint32_t ol = l & 63, orr = r & 63;
    uint64_t ml = ol == 0 ? 0ull : (~0ull >> (64 - ol));
    uint64_t mr = ~0ull >> (63 - orr);
    nl = bl->rec[c].base + (uint32_t)__popcll(fin_plane(bl->rec[c]) & ml);
    uint32_t re = br->rec[c].base + (uint32_t)__popcll(fin_plane(br->rec[c]) & mr);   // exclusive end
    nr = re - 1;
    return nl < re;
}

// drop_first_char (common.hh:38-48) for new_len >= 1 on a valid interval
__device__ __forceinline__ void d_drop(const FinDevIndex& ix, int new_len, uint32_t& l, uint32_t& r) {
    if (new_len <= 0) { l = 0; r = ix.n_nodes - 1; return; }
    while (l > 0 && (int)d_lcs(ix, l) >= new_len) l--;
    while (r < ix.n_nodes - 1 && (int)d_lcs(ix, r + 1) >= new_len) r++;
}

__device__ __forceinline__ uint32_t d_concat(const FinDevIndex& ix, uint32_t g) {
    return (ix.concat[g >> 4] >> (2 * (g & 15))) & 3u;
}

// PackedStrings::global_offset_to_local_offset: smallest idx with ends[idx] > gs
__device__ __forceinline__ void d_locate(const FinDevIndex& ix, uint32_t gs, uint32_t& u, uint32_t& ustart, uint32_t& uend) {
    uint32_t idx = ix.samp[gs >> ix.samp_shift];
    uint32_t e = ix.ends[idx + 1];
    while (e <= gs) { idx++; e = ix.ends[idx + 1]; }
    u = idx; uend = e;
    ustart = ix.ends[idx];
}

// The k (<= 63) bases of the unitig text from offset o (< 64) of the 128 bases in the two 64-base windows wa, wb (wb is only looked at when o + k > 64):
// x0 = the first 32 (all of them, k <= 32), x1 = the rest -- 2-bit codes, first base in the low bits, as the k-mer table hashes a k-mer.  What a claim of
// that table is compared with (fin_kernel_w.hip W_RES4 / W_KFV, fin_prepass.hip).
__device__ __forceinline__ void fin_text_kmer(const uint4& wa, const uint4& wb, uint32_t o, uint32_t k, uint64_t& x0, uint64_t& x1) {
    const uint64_t W0 = wa.x | ((uint64_t)wa.y << 32), W1 = wa.z | ((uint64_t)wa.w << 32), W2 = wb.x | ((uint64_t)wb.y << 32), W3 = wb.z | ((uint64_t)wb.w << 32);
    const uint32_t s = (o & 31u) * 2u;
    const bool hi = (o >> 5) != 0u;
    const uint64_t a = hi ? W1 : W0, b = hi ? W2 : W1, c = hi ? W3 : W2;
    x0 = s ? (a >> s) | (b << (64u - s)) : a;
    x1 = s ? (b >> s) | (c << (64u - s)) : b;
    if (k < 32u) x0 &= (1ull << (2u * k)) - 1ull;
    if (k <= 32u) x1 = 0ull; else x1 &= (1ull << (2u * (k - 32u))) - 1ull;   // (k - 32 <= 31)
}

__device__ __forceinline__ uint32_t d_base_code(const uint8_t* bases, uint64_t o, uint32_t len, uint32_t pos, bool rev) {
    uint8_t ch = rev ? bases[o + (len - 1 - pos)] : bases[o + pos];
    ch &= (uint8_t)~32u;
    uint32_t c = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
    if (rev && c < 4) c = 3u - c;
    return c;
}

// deque entry: len(8) | colex(32) | end mod 2^24; order of (len, colex) decides (the end never does: the
// candidate being inserted always has the largest end, see CHANGELOG.md 4.3 ("deque"))
__device__ __forceinline__ uint64_t dq_pack(uint32_t len, uint32_t colex, uint32_t end) {
    return ((uint64_t)len << 56) | ((uint64_t)colex << 24) | (uint64_t)(end & 0xFFFFFFu);
}
__device__ __forceinline__ uint32_t dq_end(uint64_t e, uint32_t cur_end) { return cur_end - ((cur_end - (uint32_t)e) & 0xFFFFFFu); }
__device__ __forceinline__ uint32_t dq_len(uint64_t e) { return (uint32_t)(e >> 56); }
__device__ __forceinline__ uint32_t dq_colex(uint64_t e) { return (uint32_t)(e >> 24); }


// a read for the overflow kernel's list (bounded: see FinDevIndex::ovf_cap)
__device__ __forceinline__ void fin_ovf_push(const FinDevIndex& ix, uint32_t* ovf_list, uint32_t* ovf_count, uint32_t r) {
    const uint32_t slot = atomicAdd(ovf_count, 1u);
    if (slot < ix.ovf_cap) ovf_list[slot] = r;
}

// first word of a kernel-4 item (fin_kernel_w.hip): bit 31 strand, bit 30 "pairs only fill slots that still hold (-1,-1)", bit 29 the lane
// also writes its strand's absent slots, bit 28 the read's other strand is deferred, bits 0..27 the read
#define FIN_ITEM_READ 0x0FFFFFFFu

// ---- queues between kernels (kernel 4's pipeline) ---------------------------------------------------------------------------
// A wave appends to a queue in HBM through slots it reserves 64 at a time: one atomic on the queue's counter per 64 items instead
// of one per item (a single word takes about 88 atomics per microsecond, MI355X_MICROARCH.md "dequeue": ten million items would
// cost 110 ms).  Slots a wave reserved but did not use are filled with FIN_Q_EMPTY when it exits; consumers skip those.  The
// counter therefore counts RESERVED slots, and a queue needs room for its items + 64 per producing wave.
#define FIN_Q_EMPTY 0xFFFFFFFFu
struct FinWaveQueue { uint32_t base = 0, left = 0; };   // wave-uniform
template <typename T>
__device__ __forceinline__ void fin_wq_push(FinWaveQueue& w, bool emit, const T& item, T* queue, uint32_t* count, uint32_t lane) {
    const uint64_t m = __ballot(emit);
    if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m), rk = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t take1 = cnt < w.left ? cnt : w.left;
        if (emit && rk < take1) queue[w.base + rk] = item;
        w.base += take1; w.left -= take1;
        if (cnt > take1) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(count, 64u);
            w.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b); w.left = 64u;
            if (emit && rk >= take1) queue[w.base + (rk - take1)] = item;
            w.base += cnt - take1; w.left -= cnt - take1;
        }
    }
}
template <typename T>
__device__ __forceinline__ void fin_wq_flush(const FinWaveQueue& w, const T& empty, T* queue, uint32_t lane) {
    if (lane < w.left) queue[w.base + lane] = empty;
}

// ---- pieces the epoch kernels share (fin_kernel_v3.hip: search body and probe pre-pass; fin_kernel_w.hip: walk kernel) ----------------
#define FIN_NONE 0xFFFFFFFFu
#define FIN_Q_RA 2u   // request flags of an epoch: rank record A / B of FinRecCache (the kernels' own flags continue from 8)
#define FIN_Q_RB 4u

// Which item a lane works on next.  Items come from a global counter in ranges of 64 per wave; the returning atomic is issued one
// epoch before its value is needed (its latency hides behind that epoch's loads): the wave holds a current range and a prefetched
// next one.  Every wave starts with the range of its own number, without touching the counter -- a launch with little or nothing to
// do costs no atomic storm; the counter hands out the ranges behind those.  Wave-uniform except `val`.
template <uint32_t R>   // R: items per range (64 for the kernels that take whole reads; the walk kernel's items are shorter-lived: FIN_WALK_RANGE)
struct FinWorkRangesT {
    uint32_t base, cnt, nbase, val, wpb;
    bool nhave, inflight, exhausted;
    __device__ __forceinline__ void init(uint32_t waves_per_block = FIN_TPB / 64u) {   // (the sorted walk kernel's blocks are larger than FIN_TPB)
        wpb = waves_per_block;
        base = (blockIdx.x * wpb + (threadIdx.x >> 6)) * R; cnt = R; nbase = 0; val = 0;
        nhave = false; inflight = false; exhausted = false;
    }
    // Once per epoch, wave-converged; need: this lane wants an item.  1 = id is the lane's next item, 2 = no item is left (the lane is
    // done), 0 = nothing yet (or not asked).
    __device__ __forceinline__ int take(bool need, uint32_t lane, uint32_t n_items, uint32_t* counter, uint32_t& id) {
        int res = 0;
        if (inflight) { nbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)val) + gridDim.x * wpb * R; nhave = true; inflight = false; }
        const uint64_t m = __ballot(need);
        if (m) {
            const uint32_t n = (uint32_t)__popcll(m), rk = (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (cnt == 0 && nhave) { base = nbase; cnt = R; nhave = false; }
            const uint32_t take1 = n < cnt ? n : cnt;
            id = base + rk; bool got = rk < take1;
            base += take1; cnt -= take1;
            const uint32_t rest = n - take1;
            if (rest && nhave) {
                base = nbase; cnt = R; nhave = false;
                const uint32_t take2 = (R >= 64u || rest < cnt) ? rest : cnt;   // (a range shorter than a wave may hold fewer items than lanes ask: the others ask again next epoch)
                if (!got && rk - take1 < take2) { id = base + (rk - take1); got = true; }
                base += take2; cnt -= take2;
            }
            if (need && (got || exhausted)) res = (got && id < n_items) ? 1 : 2;
        }
        if (base >= n_items) { exhausted = true; cnt = 0; }
        if (nhave && nbase >= n_items) { exhausted = true; nhave = false; }
        if (!nhave && !inflight && !exhausted) {
            if (lane == 0) val = atomicAdd(counter, R);
            inflight = true;
        }
        return res;
    }
};
typedef FinWorkRangesT<64u> FinWorkRanges;

// The two rank records (12 bytes of a node block: plane of 64 edge marks + rank base, FinCharRec) a lane extends an interval with,
// cached by tag = block * 4 + character.  A record asked for in an epoch (FIN_Q_RA / FIN_Q_RB in the epoch's request word) is there
// from the next epoch on.  (value selects, no conditional stores to different variables: keeps every tag in a register)
struct FinRecCache {
    uint32_t tagA = FIN_NONE, tagB = FIN_NONE; uint64_t plA = 0, plB = 0; uint32_t bsA = 0, bsB = 0;
    __device__ __forceinline__ void request(uint32_t l, uint32_t r, uint32_t c, uint32_t& q) {
        const uint32_t ta = ((l >> 6) << 2) | c, tb = ((r >> 6) << 2) | c;
        const bool ta_inA = tagA == ta, ta_inB = tagB == ta;
        const bool ldA_ta = !ta_inA && !ta_inB;
        const bool ta_atA = ta_inA || ldA_ta;
        const bool tb_toB = tb != ta && ta_atA && tagB != tb;
        const bool tb_toA = tb != ta && !ta_atA && tagA != tb;
        tagA = ldA_ta ? ta : (tb_toA ? tb : tagA);
        tagB = tb_toB ? tb : tagB;
        q |= ((ldA_ta || tb_toA) ? FIN_Q_RA : 0u) | (tb_toB ? FIN_Q_RB : 0u);
    }
    // the loads of this epoch's requests
    __device__ __forceinline__ void serve(uint32_t q, const char* blk_base) {
        if (q & FIN_Q_RA) { const FinCharRec v = *(const FinCharRec*)(blk_base + (size_t)(tagA >> 2) * 128 + 64 + 12 * (tagA & 3u)); plA = v.plane_lo | ((uint64_t)v.plane_hi << 32); bsA = v.base; }
        if (q & FIN_Q_RB) { const FinCharRec v = *(const FinCharRec*)(blk_base + (size_t)(tagB >> 2) * 128 + 64 + 12 * (tagB & 3u)); plB = v.plane_lo | ((uint64_t)v.plane_hi << 32); bsB = v.base; }
    }
    // requests that will not be served (the lane drops its work): their tags must not name data that never arrived
    __device__ __forceinline__ void drop(uint32_t q) { if (q & FIN_Q_RA) tagA = FIN_NONE; if (q & FIN_Q_RB) tagB = FIN_NONE; }
    // update_sbwt_interval (formula: common.hh:26-36) on [l, r] with the cached records: 0 = data missing (requested), 1 = ok,
    // 2 = (-1,-1).  The full interval is answered from the C array (C[4] = number of nodes).
    __device__ __forceinline__ int extend(uint32_t c, uint32_t l, uint32_t r, uint32_t n, uint32_t C0, uint32_t C1, uint32_t C2, uint32_t C3, uint32_t C4,
                                          uint32_t& q, uint32_t& nl, uint32_t& nr) {
        if (l == 0 && r == n - 1) {
            // masks, not `c == 0 ? C0 : ...`: the compiler folds a select of loads into a load through a selected ADDRESS, which
            // turns the operands into memory (kernarg loads in mid-epoch, or scratch)
            const uint32_t m0 = 0u - (uint32_t)(c == 0), m1 = 0u - (uint32_t)(c == 1), m2 = 0u - (uint32_t)(c == 2), m3 = 0u - (uint32_t)(c == 3);
            nl = (C0 & m0) | (C1 & m1) | (C2 & m2) | (C3 & m3);
            nr = ((C1 & m0) | (C2 & m1) | (C3 & m2) | (C4 & m3)) - 1;
            return nl <= nr ? 1 : 2;
        }
        if (q & (FIN_Q_RA | FIN_Q_RB)) return 0;   // requested this epoch, not there yet
        const uint32_t tl = ((l >> 6) << 2) | c, tr = ((r >> 6) << 2) | c;
        const bool lA = tl == tagA, lB = tl == tagB, rA = tr == tagA, rB = tr == tagB;
        if (!((lA || lB) && (rA || rB))) { request(l, r, c, q); return 0; }
        const uint64_t pl = lA ? plA : plB, pr = rA ? plA : plB;
        const uint32_t bl = lA ? bsA : bsB, br = rA ? bsA : bsB;
        nl = bl + (uint32_t)__popcll(pl & ~(~0ull << (l & 63u)));
        const uint32_t re = br + (uint32_t)__popcll(pr & (~0ull >> (63 - (r & 63u))));
        nr = re - 1;
        return nl < re ? 1 : 2;
    }
};

// The packed chunks of the read a lane works on (32 bases = {u64 2-bit codes, u32 validity, u32 0}, fin_pack.hip): the current chunk
// and one more -- a probe string or a filter window may span two.  The current chunk arrives through the kernel's 16-byte AUX load
// (FIN_Q_AUX | FIN_Q_CURCHUNK), the second one has its own load (FIN_Q_NEXTCHUNK), so both come in one epoch.  A tag is set when its
// load is REQUESTED; the data is there from the next epoch on.
#define FIN_Q_AUX 8u
#define FIN_Q_NEXTCHUNK 16u
#define FIN_Q_CURCHUNK 64u
struct FinChunkCache {
    int cur = -1, nxt = -1; uint64_t bcodes = 0, ncodes = 0; uint32_t bvalid = 0, nvalid = 0;
    __device__ __forceinline__ void reset() { cur = -1; nxt = -1; }
    // make chunk ci of the strand the current one (strand(): where the strand's chunks start -- only evaluated when an address is
    // needed); false = it has been requested, or the AUX slot is taken, and the caller retries next epoch
    template <class S>
    __device__ __forceinline__ bool need(int ci, S&& strand, uint32_t& q, const void*& q_aux) {
        if (cur == ci) return !(q & FIN_Q_CURCHUNK);
        // (not while a load of the CURRENT chunk is under way: it would land in bcodes after the promotion, under the promoted chunk's number --
        //  a walk that asks for the chunk behind its anchor and ends at once, followed in the same epoch by a probe whose first chunk is `nxt`;
        //  never met while every look-up moved the cache along, found with the look-ups between epochs of fin_kernel_w.hip)
        if (nxt == ci) { if (q & (FIN_Q_NEXTCHUNK | FIN_Q_CURCHUNK)) return false; bcodes = ncodes; bvalid = nvalid; cur = ci; nxt = -1; return true; }
        if (!(q & FIN_Q_AUX)) { q_aux = (const void*)(strand() + ci); q |= FIN_Q_AUX | FIN_Q_CURCHUNK; cur = ci; }
        return false;
    }
    // as need(), and when chunk ci has to be fetched the one behind it (if the strand has one: ci + 1 < n_chunks) is asked for in the same
    // epoch: the two 16-byte loads go out back to back and mostly share a 128-byte line, so the second costs no memory request of its
    // own -- asked for an epoch later, when the probes have moved on, it is an L2 miss again (the pre-pass's L2 hit rate is 2 %)
    template <class S>
    __device__ __forceinline__ bool need_ahead(int ci, int n_chunks, S&& strand, uint32_t& q, const void*& q_aux) {
        const bool fetch = cur != ci && nxt != ci && !(q & FIN_Q_AUX);
        const bool ready = need(ci, strand, q, q_aux);
        if (fetch && ci + 1 < n_chunks && !(q & FIN_Q_NEXTCHUNK)) { nxt = ci + 1; q |= FIN_Q_NEXTCHUNK; }
        return ready;
    }
    // chunks ci0 (current) and, if different, ci1 (next) both there?
    template <class S>
    __device__ __forceinline__ bool need2(int ci0, int ci1, S&& strand, uint32_t& q, const void*& q_aux, int n_chunks = 0) {
        bool ready = n_chunks ? need_ahead(ci0, n_chunks, strand, q, q_aux) : need(ci0, strand, q, q_aux);
        if (ci1 != ci0) {
            if (nxt != ci1 && !(q & FIN_Q_NEXTCHUNK)) { nxt = ci1; q |= FIN_Q_NEXTCHUNK; }
            if (nxt != ci1 || (q & FIN_Q_NEXTCHUNK)) ready = false;
        }
        return ready;
    }
    // the loads of this epoch's requests (aux: what the kernel's AUX load brought)
    template <class S>
    __device__ __forceinline__ void serve(uint32_t q, const uint4& aux, S&& strand) {
        if (q & FIN_Q_NEXTCHUNK) { uint4 nv; __builtin_memcpy(&nv, strand() + nxt, 16); ncodes = nv.x | ((uint64_t)nv.y << 32); nvalid = nv.z; }
        if (q & FIN_Q_CURCHUNK) { bcodes = aux.x | ((uint64_t)aux.y << 32); bvalid = aux.z; }
    }
    // up to 32 bases from read position p on (p in chunk ci0 = current, the rest in ci1 = next if different): codes and validity bits
    __device__ __forceinline__ void window(int p, int ci0, int ci1, uint64_t& w, uint32_t& v) const {
        const uint32_t j = (uint32_t)p & 31u;
        w = bcodes >> (2 * j); v = bvalid >> j;
        if (ci1 != ci0) { w |= ncodes << (64 - 2 * j); v |= nvalid << (32 - j); }   // (j > 0 here: a window that starts a chunk does not need the next)
    }
};

