// fin_kernel_b.hip -- upload-time kernels: the ANCHOR TABLE (FinDevIndex::pos), the SAFE-PLACE bitmap (FinDevIndex::safe) and the K-MER TABLE
// (FinDevIndex::kt3, round 5's compact form: every text k-mer -> {the reference's answer for it, a tag of its hash}, entered by the same pass).
//
// What they hold (CHANGELOG.md 4.8/4.9, round 3).  When a present k-mer Q is not reached by a walk, FinimizerIndex::search reports a place
// computed from the streaming state: the finimizer dictionary's offset of the least candidate of Q's window, or the branch dictionary's
// unitig start when a Ustart record lies at or behind that candidate's end (FinimizerIndex.hh:148-174).  That answer G is a function of
// Q ALONE: a candidate that starts inside Q's window is the shortest unique suffix ending at its position, recorded iff the longest
// repeated suffix one position earlier was shorter -- both decided by Q's own bases; and at every window position at or behind the
// finimizer's end the k-mer interval's string contains a unique string, so that interval is ONE node whatever precedes Q in the read, and
// the branch record taken there does not depend on the history either.  So G can be tabulated per SBWT node, on ANY index:
//
//   pos[v].g      = G(v) for the node v of every k-mer of the text (0xFFFFFFFF: none; FIN_POS_DUMMY | d: a dummy node)
//   pos[v].u ...  = the unitig of that place and its bounds -- written only when the text AT G(v) spells v's k-mer inside one unitig
//                   (a "verified" entry; u keeps its top bit set otherwise).  On a set of disjoint unitigs every entry is verified; with
//                   duplicated k-mers the reference may report a place where the k-mer is not (it never checks), and such an entry can
//                   only be used once the k-mer's presence is known by other means (look-up of the whole k-mer).
//   safe bit g    = the k-mer that the text spells at [g-k+1, g] is reported AT g, i.e. G(its node) == g.  A k-mer found by comparing a
//                   read with the text (walk kernel: seeds, text re-anchoring) is reported there only if this bit is set; else the
//                   streaming search decides, as in the reference.  The bitmap is dropped when every bit of a k-mer position is set.
//
// How: a lane streams FIN_ANCH_SEG consecutive text positions through the PLAIN streaming search (the obviously-faithful form of
// rarest_fmin_streaming_search, common.hh:78-186, as in fin_kernels.hip: two SBWT intervals, drop_first_char on the LCS bytes, the
// sliding-window deque in LDS) with the unitig text as its read, started 2k bases earlier -- or at its unitig's start: a cold start is
// exact from 2k-1 bases on (CHANGELOG.md 4.6) -- and evaluates the two dictionaries at every k-mer end, with the walk switched off.  The
// node of the k-mer is the k-mer interval itself.  Segments whose candidate deque outgrows the LDS slots are redone with the deque in
// global memory.  One pass over the text at upload: 250 Mbp in tens of milliseconds, beside a 99-ms prefix-table build.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fin_device.h"
#include "fin_kernels.h"

#define FIN_ANCH_SEG 512u   // text positions per lane (a multiple of 64: a lane owns whole words of the bitmap)

namespace {
struct BLdsDeque {
    static constexpr uint32_t CAP = 16;
    uint64_t* base; uint32_t limit;
    __device__ __forceinline__ uint64_t get(uint32_t i) const { return base[(i & (CAP - 1)) * FIN_TPB]; }
    __device__ __forceinline__ void set(uint32_t i, uint64_t v) { base[(i & (CAP - 1)) * FIN_TPB] = v; }
};
struct BGlobalDeque {
    static constexpr uint32_t CAP = 256;   // >= k: with eager popping at most k entries are live
    uint64_t* base; uint64_t stride; uint32_t limit;
    __device__ __forceinline__ uint64_t get(uint32_t i) const { return base[(uint64_t)(i & (CAP - 1)) * stride]; }
    __device__ __forceinline__ void set(uint32_t i, uint64_t v) { base[(uint64_t)(i & (CAP - 1)) * stride] = v; }
};

// One segment [s0, s1) of text positions.  false: the deque overflowed (nothing of the segment is final: redo it).
template <typename DQ>
__device__ bool anchor_segment(const FinDevIndex& ix, uint32_t s0, uint32_t s1, FinSeedEntry* pos, unsigned long long* safe, FinKt3Bucket* kt3, uint32_t kt3_buckets, DQ dq, uint32_t& n_unsafe, uint32_t* ktab_full, FinKtxSlot* ulist, uint32_t ulist_cap) {
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    uint32_t u = ix.samp[s0 >> ix.samp_shift];
    while (ix.ends[u + 1] <= s0) u++;
    uint32_t ustart = ix.ends[u], uend = ix.ends[u + 1];
    uint32_t g = ustart;
    if (s0 >= (uint32_t)(2 * k) && s0 - (uint32_t)(2 * k) > g) g = s0 - (uint32_t)(2 * k);
    // the streaming search's state (common.hh:79-101), positions = offsets in the concatenation
    uint32_t il = 0, ir = n - 1, kl = 0, kr = n - 1;
    uint32_t start = g, kstart = g;
    int64_t bu_end = -1; uint32_t bu_colex = 0;
    uint32_t dq_head = 0, dq_cnt = 0;
    unsigned long long bits[FIN_ANCH_SEG / 64];
    for (uint32_t i = 0; i < FIN_ANCH_SEG / 64; i++) bits[i] = 0ull;
    uint32_t unsafe = 0;
    // the 2-bit codes of the last k bases (k <= 64), first base in the low bits: bases 0..31 in key0, the rest in key1 -- what the k-mer table hashes
    uint64_t key0 = 0, key1 = 0;
    // (k > 64: the words beyond the second, rolled the same way; all of them fold into the hash -- fin_kt3_fold)
    uint64_t keyx[6] = {0, 0, 0, 0, 0, 0};
    const int nwx = k > 64 ? (k - 64 + 31) / 32 : 0;   // words 2 .. 2 + nwx - 1
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    // one entry of the k-mer table: {answer, tag | flags}.  Slots of a bucket fill in order and nothing is ever removed, so a look-up may stop at the first empty
    // slot; a value that is already there (the places of an unverified k-mer all compute the same one) is not entered twice
    auto kt3_insert = [&](uint32_t G, bool ver) {
        if (*(volatile uint32_t*)ktab_full) return;
        uint64_t h;
        if (nwx == 0) h = fin_kt3_hash(key0, key1);
        else {
            uint64_t key = fin_kt3_fold(key0, key1, 1u);
            for (int j = 0; j < nwx; j++) key = fin_kt3_fold(key, keyx[j], (uint32_t)(j + 2));
            h = fin_mix64(key);
        }
        const unsigned long long val = (unsigned long long)G | ((unsigned long long)(((uint32_t)h & FIN_KT3_TAGMASK) | (ver ? 0u : FIN_KT3_UNVER)) << 32);
        uint32_t b = fin_kt3_bucket(h, kt3_buckets);
        for (uint32_t tries = 0; ; tries++) {
            unsigned long long* const sl = (unsigned long long*)(kt3 + b);
            for (int j = 0; j < FIN_KT3_SLOTS; j++) {
                const unsigned long long old = atomicCAS(&sl[j], (unsigned long long)FIN_KT3_EMPTY, val);
                if (old == FIN_KT3_EMPTY || old == val) return;
            }
            if (tries >= kt3_buckets) { atomicExch(ktab_full, 1u); return; }   // (table full: the host sized it for a load of 55 % and fails the upload -- never a wrong answer)
            b = b + 1u == kt3_buckets ? 0u : b + 1u;
        }
    };
    // a k-mer whose answer is unverified also goes, with its whole key, on the list the exact side table (FinDevIndex::ktx) is made from once their number
    // is known (ktab_full[1] counts them; entries beyond the list's room are counted and dropped: the table is then marked partial)
    auto ulist_push = [&](uint32_t G) {
        if (k > 64) return;   // (the side table's keys are two words: the walk kernel, which asks it, looks whole k-mers up through the k-mer table for k <= 63 only)
        const uint32_t at = atomicAdd(ktab_full + 1, 1u);
        if (at < ulist_cap) ulist[at] = FinKtxSlot{(uint32_t)key0, (uint32_t)(key0 >> 32), (uint32_t)key1, (uint32_t)(key1 >> 32), G, 1u, 0u, 0u};
    };

    for (; g < s1; g++) {
        if (g >= uend) {   // the next unitig begins: the state the search has before its first base
            do { u++; ustart = uend; uend = ix.ends[u + 1]; } while (g >= uend);
            il = 0; ir = n - 1; kl = 0; kr = n - 1; start = g; kstart = g; bu_end = -1; dq_head = 0; dq_cnt = 0;
        }
        const uint32_t c = d_concat(ix, g);
        if (k > 64) {   // every word moves down a base; the new base is base k-1, in the last word
            key0 = (key0 >> 2) | (key1 << 62); key1 = (key1 >> 2) | (keyx[0] << 62);
            for (int j = 0; j < nwx; j++) keyx[j] = (keyx[j] >> 2) | (j + 1 < nwx ? keyx[j + 1] << 62 : 0ull);
            keyx[nwx - 1] |= (uint64_t)c << (2 * ((k - 1) & 31));
        } else
        if (k >= 33) { key0 = (key0 >> 2) | (key1 << 62); key1 = (key1 >> 2) | ((uint64_t)c << (2 * ((k - 33) & 31))); }   // (the new base is base k-1: in the second word ...
        else key0 = ((key0 >> 2) | ((uint64_t)c << (2 * (k - 1)))) & kmask;                                                  //  ... or, k <= 32, the first one's last)
        // (1) finimizer interval, common.hh:114-127
        uint32_t nl, nr;
        bool ok = d_extend(ix, c, il, ir, nl, nr);
        while (!ok) {
            kstart = ++start;
            if (start > g) { nl = 0; nr = n - 1; kl = nl; kr = nr; break; }
            d_drop(ix, (int)(g - start), il, ir);
            ok = d_extend(ix, c, il, ir, nl, nr);
            kl = nl; kr = nr;
        }
        il = nl; ir = nr;
        // (2) k-mer interval, :132-143
        if (start != kstart) {
            uint32_t nkl, nkr;
            bool okk = d_extend(ix, c, kl, kr, nkl, nkr);
            while (!okk) {
                kstart++;
                d_drop(ix, (int)(g - kstart), kl, kr);
                okk = d_extend(ix, c, kl, kr, nkl, nkr);
            }
            kl = nkl; kr = nkr;
        } else { kl = il; kr = ir; }
        // candidates that start before the k-mer window (eager form of the pop_front loop, :173-176; CHANGELOG.md 4.3)
        while (dq_cnt) {
            const uint64_t f = dq.get(dq_head);
            if (dq_end(f, g) - dq_len(f) + 1 < kstart) { dq_head++; dq_cnt--; } else break;
        }
        // (2b) shortest unique suffix -> candidate, :145-164
        if (il == ir) {
            uint32_t cl = 0, cc = 0;
            do {
                cl = g - start + 1; cc = il;
                start++;
                d_drop(ix, (int)(g - start + 1), il, ir);
            } while (il == ir);
            const uint64_t cand = dq_pack(cl, cc, g);
            if (dq_cnt && (dq.get(dq_head) >> 24) > (cand >> 24)) dq_cnt = 0;
            else { while (dq_cnt && (dq.get(dq_head + dq_cnt - 1) >> 24) > (cand >> 24)) dq_cnt--; }
            if (dq_cnt >= dq.limit) return false;
            dq.set(dq_head + dq_cnt, cand); dq_cnt++;
        }
        // Ustart probe, :167
        if (kl == kr && (d_nodebyte(ix, kl) & FIN_USTART_BIT)) { bu_end = (int64_t)g; bu_colex = kl; }
        // a k-mer ends here, :170-182 -- with the dictionary look-ups of FinimizerIndex.hh:148-174 in place of the recorded optionals
        if (g - kstart + 1 == (uint32_t)k) {
            if (g >= s0 && dq_cnt && kl == kr) {
                const uint64_t w = dq.get(dq_head);
                const uint32_t fin_end = dq_end(w, g), fin_colex = dq_colex(w);
                uint32_t G;
                if (bu_end >= (int64_t)fin_end) {   // lookup_from_branch_dictionary, common.hh:61-67
                    const uint32_t o = bu_colex & 63u;
                    const FinBlockInfo bi = ix.blkinfo[bu_colex >> 6];
                    const uint32_t rank = bi.ustart_rank + (uint32_t)__popcll((bi.ustart_mask_lo | ((uint64_t)bi.ustart_mask_hi << 32)) & (o ? (~0ull >> (64 - o)) : 0ull));
                    G = ix.ends[rank] + (uint32_t)(k - 1) + (g - (uint32_t)bu_end);
                } else {                            // lookup_from_finimizer_dictionary, common.hh:69-72
                    const uint32_t o = fin_colex & 63u;
                    const FinBlockInfo bi = ix.blkinfo[fin_colex >> 6];
                    const uint32_t rank = bi.fmin_rank + (uint32_t)__popcll((bi.fmin_mask_lo | ((uint64_t)bi.fmin_mask_hi << 32)) & (o ? (~0ull >> (64 - o)) : 0ull));
                    G = ix.goff[rank] + g - fin_end;
                }
                // every place of this node's k-mer computes the same G; the place that IS G writes the whole entry -- and enters the k-mer in the k-mer table
                // (FinDevIndex::kt3) as VERIFIED: the text at its answer spells it.  Another place of a k-mer (G != g) enters it only when NO place will: the
                // text at G does not spell the k-mer ("unverified": the reference reports a place where the k-mer is not -- it never checks)
                if (kt3) {
                    if (G == g) kt3_insert(G, true);
                    else {
                        bool ver = false;
                        if (G >= (uint32_t)(k - 1) && G < ix.total_len) {
                            uint32_t uu = ix.samp[(G - (uint32_t)(k - 1)) >> ix.samp_shift];
                            while (ix.ends[uu + 1] <= G - (uint32_t)(k - 1)) uu++;
                            ver = G < ix.ends[uu + 1];
                            for (int j = 0; ver && j < k; j++) ver = d_concat(ix, G - (uint32_t)j) == d_concat(ix, g - (uint32_t)j);
                        }
                        if (!ver) { kt3_insert(G, false); ulist_push(G); }
                    }
                }
                if (G == g) {
                    if (pos) pos[kl] = FinSeedEntry{g, u, ustart, uend};
                    bits[(g - s0) >> 6] |= 1ull << ((g - s0) & 63u);
                } else {
                    if (pos && G < FIN_POS_DUMMY) pos[kl].g = G;   // (an answer outside the table's range cannot be kept: the entry stays "none")
                    unsafe++;
                }
            } else if (g >= s0) { unsafe++; if (kt3 && kl == kr) { kt3_insert(0xFFFFFFFFu, false); ulist_push(0xFFFFFFFFu); } }   // (unreachable on a consistent index: a text k-mer without a candidate -- "present, no answer known")
            kstart++;
            d_drop(ix, (int)(g - kstart + 1), kl, kr);
        }
    }
    if (safe) for (uint32_t i = 0; i < FIN_ANCH_SEG / 64 && s0 + 64u * i < s1; i++) safe[(s0 >> 6) + i] = bits[i];
    n_unsafe = unsafe;
    return true;
}
}  // namespace

__global__ __launch_bounds__(FIN_TPB) void fin_build_anchor_kernel(FinDevIndex ix, FinSeedEntry* pos, unsigned long long* safe, FinKt3Bucket* kt3, uint32_t kt3_buckets,
                                                                   uint32_t n_seg, uint32_t* ovf_list, uint32_t* ovf_count, unsigned long long* unsafe_total, FinKtxSlot* ulist, uint32_t ulist_cap) {
    __shared__ uint64_t lds_dq[BLdsDeque::CAP * FIN_TPB];
    const uint32_t seg = blockIdx.x * FIN_TPB + threadIdx.x;
    if (seg >= n_seg) return;
    const uint64_t s0 = (uint64_t)seg * FIN_ANCH_SEG;
    const uint32_t s1 = (uint32_t)(s0 + FIN_ANCH_SEG < ix.total_len ? s0 + FIN_ANCH_SEG : ix.total_len);
    BLdsDeque dq{lds_dq + threadIdx.x, BLdsDeque::CAP};
    uint32_t unsafe = 0;
    // (a segment that overflows is redone from scratch: what it has entered so far is entered again -- the same values, which the table takes once; the list
    //  may hold such a segment's unverified k-mers twice, which only costs slots)
    if (!anchor_segment<BLdsDeque>(ix, (uint32_t)s0, s1, pos, safe, kt3, kt3_buckets, dq, unsafe, (uint32_t*)(unsafe_total + 2), ulist, ulist_cap)) { ovf_list[atomicAdd(ovf_count, 1u)] = seg; return; }
    if (unsafe) atomicAdd(unsafe_total, (unsigned long long)unsafe);
}
// segments whose candidate deque outgrew the LDS slots, with the deque in a global scratch ring
__global__ __launch_bounds__(FIN_TPB) void fin_build_anchor_overflow_kernel(FinDevIndex ix, FinSeedEntry* pos, unsigned long long* safe, FinKt3Bucket* kt3, uint32_t kt3_buckets,
                                                                            const uint32_t* ovf_list, const uint32_t* ovf_count, uint64_t* scratch, unsigned long long* unsafe_total, FinKtxSlot* ulist, uint32_t ulist_cap) {
    const uint32_t nthreads = gridDim.x * FIN_TPB, tid = blockIdx.x * FIN_TPB + threadIdx.x;
    const uint32_t cnt = *ovf_count;
    BGlobalDeque dq{scratch + tid, nthreads, BGlobalDeque::CAP};
    for (uint32_t i = tid; i < cnt; i += nthreads) {
        const uint64_t s0 = (uint64_t)ovf_list[i] * FIN_ANCH_SEG;
        const uint32_t s1 = (uint32_t)(s0 + FIN_ANCH_SEG < ix.total_len ? s0 + FIN_ANCH_SEG : ix.total_len);
        uint32_t unsafe = 0;
        // (k <= 255 < CAP live candidates at most: cannot fail)
        if (anchor_segment<BGlobalDeque>(ix, (uint32_t)s0, s1, pos, safe, kt3, kt3_buckets, dq, unsafe, (uint32_t*)(unsafe_total + 2), ulist, ulist_cap) && unsafe) atomicAdd(unsafe_total, (unsigned long long)unsafe);
    }
}

// The dummy nodes ($-padded prefixes of the k-mers that have no predecessor, i.e. of unitig starts): a lane follows the first k-1
// bases of a unitig from the root node (node 0, "$$..$") along single edges; the node after d bases -- if the path exists -- is
// the dummy "$..$ U[0..d-1]", and gets FIN_POS_DUMMY | d: a string that ends only that node ends no k-mer, nor does any extension
// of it by fewer than k-d bases (their nodes are the dummy's descendants, still $-padded).  True on any index.
__global__ __launch_bounds__(FIN_TPB) void fin_build_pos_dummies_kernel(FinDevIndex ix, FinSeedEntry* pos) {
    const uint32_t u = blockIdx.x * FIN_TPB + threadIdx.x;
    if (u >= ix.n_unitigs) return;
    const char* const blk_base = (const char*)ix.blocks;
    const uint32_t ustart = ix.ends[u], uend = ix.ends[u + 1];
    uint32_t v = 0;
    for (uint32_t d = 1; d < ix.k && ustart + d - 1 < uend; d++) {
        const uint32_t g = ustart + d - 1;
        const uint32_t c = (ix.concat[g >> 4] >> (2 * (g & 15u))) & 3u;
        const FinCharRec a = *(const FinCharRec*)(blk_base + (size_t)(v >> 6) * 128 + 64 + 12 * c);
        const uint64_t pa = a.plane_lo | ((uint64_t)a.plane_hi << 32);
        if (!((pa >> (v & 63u)) & 1ull)) break;   // no such edge: this unitig's start has predecessors, or the path belongs to others from here on
        v = a.base + (uint32_t)__popcll(pa & ~(~0ull << (v & 63u)));
        pos[v].g = FIN_POS_DUMMY | d;
    }
}

// ---- reverse-complement pairs: how many k-mers of the text have their reverse complement in the index too (a k-mer that is its own counts)
// A set that holds every canonical k-mer once -- the unitigs of a bidirected de Bruijn graph -- has none; then a k-mer found on one strand
// of a read is certainly absent on the other, which lets the pipeline search a read's second strand only where the first left slots open
// (CHANGELOG.md 4.14).  A lane takes FIN_ANCH_SEG text positions; the reverse complement of the k-mer that ends at g begins with the
// complements of text[g], text[g-1], ...: its first T bases through the prefix table (a rolling key), the rest by extends.
// rcwin (may be null): a bit per window of 64 text positions -- does a k-mer that ends in it have its reverse complement in the index?  A lane's
// FIN_ANCH_SEG = 512 positions are eight windows: one byte, rcwin[s0 / 512].
__global__ __launch_bounds__(FIN_TPB) void fin_count_rc_pairs_kernel(FinDevIndex ix, unsigned long long* count, uint8_t* rcwin) {
    const uint64_t s0 = ((uint64_t)blockIdx.x * FIN_TPB + threadIdx.x) * FIN_ANCH_SEG;
    if (s0 >= ix.total_len) return;
    const uint32_t s1 = (uint32_t)(s0 + FIN_ANCH_SEG < ix.total_len ? s0 + FIN_ANCH_SEG : ix.total_len);
    const uint32_t k = ix.k, T = ix.ptab_t <= k ? ix.ptab_t : 0u, n = ix.n_nodes;
    uint32_t u = ix.samp[s0 >> ix.samp_shift];
    while (ix.ends[u + 1] <= (uint32_t)s0) u++;
    uint32_t ustart = ix.ends[u], uend = ix.ends[u + 1];
    uint32_t g = ustart;
    if (s0 >= k - 1 && (uint32_t)s0 - (k - 1) > g) g = (uint32_t)s0 - (k - 1);
    uint64_t key = 0;   // codes of comp(text[g]), comp(text[g-1]), ... : the first bases of the reverse complement, first base in the low bits
    const uint64_t tmask = T ? ((1ull << (2 * T)) - 1ull) : 0ull;
    unsigned long long found = 0;
    uint32_t wins = 0;
    for (; g < s1; g++) {
        while (g >= uend) { u++; ustart = uend; uend = ix.ends[u + 1]; }
        key = ((key << 2) | (uint64_t)(3u - d_concat(ix, g))) & tmask;
        if (g - ustart + 1 < k || g < (uint32_t)s0) continue;
        uint32_t l = 0, r = n - 1; bool ok = true;
        uint32_t i = 0;
        if (T) { const FinPrefixIval iv = ix.ptab[(uint32_t)key]; l = iv.l; r = iv.r; ok = l <= r; i = T; }
        for (; ok && i < k; i++) { uint32_t nl, nr; ok = d_extend(ix, 3u - d_concat(ix, g - i), l, r, nl, nr); l = nl; r = nr; }
        found += ok ? 1ull : 0ull;
        if (ok) wins |= 1u << ((g - (uint32_t)s0) >> 6);
    }
    if (rcwin) rcwin[s0 / FIN_ANCH_SEG] = (uint8_t)wins;
    if (found) atomicAdd(count, found);
}
extern "C" uint64_t fin_rcwin_bytes(uint64_t total_len) { return (total_len + FIN_ANCH_SEG - 1) / FIN_ANCH_SEG + 16; }
// rcwin: null, or fin_rcwin_bytes(total_len) bytes (every lane writes its byte)
extern "C" int fin_launch_count_rc_pairs(const FinDevIndex* ix, void* tmp8, uint64_t* n_pairs, void* rcwin, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(tmp8, 0, 8, stream);
    if (e != hipSuccess) return (int)e;
    const uint64_t lanes = ((uint64_t)ix->total_len + FIN_ANCH_SEG - 1) / FIN_ANCH_SEG;
    if (lanes) hipLaunchKernelGGL(fin_count_rc_pairs_kernel, dim3((uint32_t)((lanes + FIN_TPB - 1) / FIN_TPB)), dim3(FIN_TPB), 0, stream, *ix, (unsigned long long*)tmp8, (uint8_t*)rcwin);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    unsigned long long h = 0;
    if ((e = hipMemcpyAsync(&h, tmp8, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
    if (n_pairs) *n_pairs = (uint64_t)h;
    return 0;
}

// pos: n_nodes + 1 entries (null: lean tables); safe: fin_anchor_safe_words() u64 (zeroed here); kt3: null, or kt3_buckets buckets of 32 bytes (emptied here; k <= 64);
// tmp: fin_anchor_tmp_bytes() of scratch; *n_unsafe_out: k-mer positions of the text that are not the place the reference reports for their k-mer.  Synchronises the stream.
extern "C" uint64_t fin_anchor_safe_words(uint64_t total_len) { return (total_len + 63) / 64 + FIN_ANCH_SEG / 64 + 2; }
// (the list of unverified k-mers: room for one in 64 text positions, 65536 at least -- a set with more is served by a partial side table)
extern "C" uint32_t fin_anchor_ulist_cap(uint64_t total_len) { const uint64_t c = total_len / 64 + 65536; return (uint32_t)(c > 0x7FFFFFFFull ? 0x7FFFFFFFull : c); }
extern "C" uint64_t fin_anchor_tmp_bytes(uint64_t total_len) {
    const uint64_t n_seg = (total_len + FIN_ANCH_SEG - 1) / FIN_ANCH_SEG;
    return (n_seg + 4) * 4 + 64 + 64 + 64ull * FIN_TPB * BGlobalDeque::CAP * 8 + 64 + (uint64_t)fin_anchor_ulist_cap(total_len) * sizeof(FinKtxSlot);
}
// where fin_launch_build_anchors left the list of the unverified k-mers inside tmp
extern "C" void* fin_anchor_ulist(void* tmp, uint64_t total_len) {
    const uint64_t n_seg = (total_len + FIN_ANCH_SEG - 1) / FIN_ANCH_SEG;
    const uint64_t off = 64 + ((n_seg + 4) * 4 + 63) / 64 * 64 + 64ull * FIN_TPB * BGlobalDeque::CAP * 8;
    return (char*)tmp + (off + 63) / 64 * 64;
}
// the exact side table of the unverified k-mers from that list: ktx = 2^log2 slots of 32 bytes (emptied here), n list entries
__global__ __launch_bounds__(FIN_TPB) void fin_build_ktx_kernel(const FinKtxSlot* ulist, uint32_t n, FinKtxSlot* ktx, uint32_t log2) {
    const uint32_t i = blockIdx.x * FIN_TPB + threadIdx.x;
    if (i >= n) return;
    const FinKtxSlot e = ulist[i];
    const uint64_t k0 = e.k0_lo | ((uint64_t)e.k0_hi << 32), k1 = e.k1_lo | ((uint64_t)e.k1_hi << 32);
    uint32_t slot = (uint32_t)(fin_kt3_hash(k0, k1) >> 32) & ((1u << log2) - 1u);
    for (uint32_t tries = 0; tries < (1u << log2); tries++) {   // (one writer per slot: `claim`; the table is at most half full)
        if (atomicCAS(&ktx[slot].claim, 0xFFFFFFFFu, 1u) == 0xFFFFFFFFu) {
            ktx[slot].k0_lo = e.k0_lo; ktx[slot].k0_hi = e.k0_hi; ktx[slot].k1_lo = e.k1_lo; ktx[slot].k1_hi = e.k1_hi; ktx[slot].g = e.g;
            return;
        }
        slot = (slot + 1u) & ((1u << log2) - 1u);
    }
}
extern "C" int fin_launch_build_ktx(const void* ulist, uint32_t n, void* ktx, uint32_t log2, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(ktx, 0xFF, (32ull << log2) + 32, stream);
    if (e != hipSuccess) return (int)e;
    if (n) hipLaunchKernelGGL(fin_build_ktx_kernel, dim3((n + FIN_TPB - 1) / FIN_TPB), dim3(FIN_TPB), 0, stream, (const FinKtxSlot*)ulist, n, (FinKtxSlot*)ktx, log2);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    return (int)hipStreamSynchronize(stream);
}
// *n_unver_out: k-mer places whose k-mer has an unverified answer (entered in the list fin_anchor_ulist(tmp) up to its room)
extern "C" int fin_launch_build_anchors(const FinDevIndex* ix, FinSeedEntry* pos, void* safe, void* kt3, uint32_t kt3_buckets, void* tmp, uint64_t* n_unsafe_out, hipStream_t stream, uint64_t* n_unver_out) {
    hipError_t e = pos ? hipMemsetAsync(pos, 0xFF, ((size_t)ix->n_nodes + 1) * sizeof(FinSeedEntry), stream) : hipSuccess;   // (pos null: "lean tables" -- only the k-mer table, the bitmap and the count)
    if (e != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(safe, 0, fin_anchor_safe_words(ix->total_len) * 8, stream)) != hipSuccess) return (int)e;
    if (kt3 && (e = hipMemsetAsync(kt3, 0xFF, 32ull * kt3_buckets, stream)) != hipSuccess) return (int)e;   // every slot empty
    const uint64_t n_seg = ((uint64_t)ix->total_len + FIN_ANCH_SEG - 1) / FIN_ANCH_SEG;
    if (n_unsafe_out) *n_unsafe_out = 0;
    if (n_seg == 0) return 0;
    // tmp: [0,8) unsafe total, [8,12) overflow count, [16,20) "the k-mer table is full", [20,24) unverified k-mers counted, [64, 64 + 4 n_seg) overflow list, then the
    // global deque rings, then the list of unverified k-mers (fin_anchor_ulist)
    unsigned long long* const d_unsafe = (unsigned long long*)tmp;
    uint32_t* const d_cnt = (uint32_t*)((char*)tmp + 8);
    uint32_t* const d_list = (uint32_t*)((char*)tmp + 64);
    uint64_t* const d_scratch = (uint64_t*)((char*)tmp + 64 + ((n_seg + 4) * 4 + 63) / 64 * 64);
    if ((e = hipMemsetAsync(tmp, 0, 64, stream)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fin_build_anchor_kernel, dim3((uint32_t)((n_seg + FIN_TPB - 1) / FIN_TPB)), dim3(FIN_TPB), 0, stream, *ix, pos, (unsigned long long*)safe,
                       (FinKt3Bucket*)kt3, kt3_buckets, (uint32_t)n_seg, d_list, d_cnt, d_unsafe, (FinKtxSlot*)fin_anchor_ulist(tmp, ix->total_len), fin_anchor_ulist_cap(ix->total_len));
    hipLaunchKernelGGL(fin_build_anchor_overflow_kernel, dim3(64), dim3(FIN_TPB), 0, stream, *ix, pos, (unsigned long long*)safe, (FinKt3Bucket*)kt3, kt3_buckets, d_list, d_cnt, d_scratch, d_unsafe, (FinKtxSlot*)fin_anchor_ulist(tmp, ix->total_len), fin_anchor_ulist_cap(ix->total_len));
    if (pos && ix->C[0] >= 1)   // (a root node exists: node 0 is "$$..$")
        hipLaunchKernelGGL(fin_build_pos_dummies_kernel, dim3((ix->n_unitigs + FIN_TPB - 1) / FIN_TPB), dim3(FIN_TPB), 0, stream, *ix, pos);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    unsigned long long h[3] = {0, 0, 0};
    if ((e = hipMemcpyAsync(h, d_unsafe, 24, hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
    if (n_unsafe_out) *n_unsafe_out = (uint64_t)h[0];
    if (n_unver_out) *n_unver_out = (uint64_t)(h[2] >> 32);
    if ((uint32_t)h[2]) return (int)hipErrorOutOfMemory;   // the k-mer table filled up (the caller sized it for a load of 55 %)
    return 0;
}

// ---- canonical string filter (FinDevIndex::cbf, round 4) -----------------------------------------------------------------------------
// Every string of m bases that lies inside ONE unitig is entered in canonical form (the smaller of its 2-bit key and its reverse complement's;
// a key holds the string's first base in its low bits, as the read windows of the search kernels do).  A k-mer of the index lies inside one
// unitig, so every m-base substring of every k-mer is entered: a string the filter does not know is a substring of no k-mer of the index, in
// either orientation.  A lane takes FIN_ANCH_SEG text positions (string ENDS) and rolls both keys along the text.
__device__ __forceinline__ void cbf_insert(uint32_t* words, uint32_t log2_blocks, uint64_t canon) {
    const uint64_t h = fin_cbf_hash(canon);
    uint32_t* const blk = words + 4 * (size_t)((h >> 35) & ((1ull << log2_blocks) - 1ull));
    uint32_t m[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < FIN_CBF_BITS; i++) { const uint32_t p = (uint32_t)(h >> (7 * i)) & 127u; m[p >> 5] |= 1u << (p & 31u); }
#pragma unroll
    for (int w = 0; w < 4; w++) if (m[w] && (blk[w] & m[w]) != m[w]) atomicOr(&blk[w], m[w]);
}
// (words_f, may be null: the DIRECTIONAL filter -- the same strings entered as they stand, FinDevIndex::fbf)
__global__ __launch_bounds__(FIN_TPB) void fin_build_cbf_kernel(FinDevIndex ix, uint32_t* words, uint32_t* words_f, uint32_t log2_blocks, uint32_t m, uint32_t n_seg) {
    const uint32_t seg = blockIdx.x * FIN_TPB + threadIdx.x;
    if (seg >= n_seg) return;
    const uint64_t s0_64 = (uint64_t)seg * FIN_ANCH_SEG;
    const uint32_t s0 = (uint32_t)s0_64, s1 = (uint32_t)(s0_64 + FIN_ANCH_SEG < ix.total_len ? s0_64 + FIN_ANCH_SEG : ix.total_len);
    uint32_t u = ix.samp[s0 >> ix.samp_shift];
    while (ix.ends[u + 1] <= s0) u++;
    uint32_t ustart = ix.ends[u], uend = ix.ends[u + 1];
    uint32_t g = ustart;
    if (s0 >= m - 1u && s0 - (m - 1u) > g) g = s0 - (m - 1u);
    const uint64_t mask = m >= 32u ? ~0ull : ((1ull << (2 * m)) - 1ull);
    uint64_t f = 0, v = 0; uint32_t have = 0;
    for (; g < s1; g++) {
        if (g >= uend) { do { u++; ustart = uend; uend = ix.ends[u + 1]; } while (g >= uend); have = 0; }
        const uint64_t c = d_concat(ix, g);
        f = (f >> 2) | (c << (2 * (m - 1u)));
        v = ((v << 2) | (3ull - c)) & mask;
        have++;
        if (have >= m && g >= s0) { cbf_insert(words, log2_blocks, f < v ? f : v); if (words_f) cbf_insert(words_f, log2_blocks, f); }
    }
}
// words: (16 << log2_blocks) bytes, zeroed here
extern "C" int fin_launch_build_cbf(const FinDevIndex* ix, void* words, void* words_f, uint32_t log2_blocks, uint32_t m, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(words, 0, 16ull << log2_blocks, stream);
    if (e != hipSuccess) return (int)e;
    if (words_f && (e = hipMemsetAsync(words_f, 0, 16ull << log2_blocks, stream)) != hipSuccess) return (int)e;
    const uint64_t n_seg = ((uint64_t)ix->total_len + FIN_ANCH_SEG - 1) / FIN_ANCH_SEG;
    if (n_seg == 0 || m < 1 || m > 32) return 0;
    hipLaunchKernelGGL(fin_build_cbf_kernel, dim3((uint32_t)((n_seg + FIN_TPB - 1) / FIN_TPB)), dim3(FIN_TPB), 0, stream, *ix, (uint32_t*)words, (uint32_t*)words_f, log2_blocks, m, (uint32_t)n_seg);
    return (int)hipGetLastError();
}
