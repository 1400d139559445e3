// fin_kernel_presence.hip -- strand pre-filter of the search-fmin path.
//
// The reference loop searches every read twice, as given and reverse-complemented (search_fmin.hh:47-51), and merges.  A read
// comes from one strand, so one of the two searches (and both, for unrelated reads) finds no k-mer at all -- yet it costs as
// much as the one that does, because rarest_fmin_streaming_search (common.hh:78-186) keeps all its machinery running.  Whether
// a strand holds ANY present k-mer needs far less: `end - kmer_start + 1 == k` (common.hh:170) only depends on the longest
// suffix of q[..end] that exists in the SBWT, capped at k -- matching statistics of ONE interval (extend; on failure
// kmer_start++ / drop_first_char / retry, which is what loops (1) and (2) of the reference do to that quantity).  This kernel
// computes just that per strand, stops at the first present k-mer, and hands the full kernel a copy of the read descriptors
// whose `off` field carries the strands to SKIP in its top two bits (bit 62: forward, bit 63: reverse).  A strand without a present k-mer yields nothing but (-1,-1) in the reference (walk hits need an anchor
// hit first, FinimizerIndex.hh:148-183), which is what the pre-filled output already holds -- results are unchanged.
//
// Same execution model as fin_kernel_v2.hip (epochs, guarded blocks, work queue), with a fraction of the state: one interval,
// no deque (no LDS), no candidate/Ustart/walk/output blocks; ~60 VGPRs.
#include "fin_device.h"
#include "fin_kernels.h"

namespace {
enum : uint32_t { S_DONE = 0, S_READ0, S_READ1, S_STRAND_END, S_CHUNKWAIT, S_BDROP, S_BASE, S_EXT, S_ARRIVE };
enum : uint32_t { Q_W = 1, Q_RA = 2, Q_RB = 4, Q_AUX = 8, Q_NEXTCHUNK = 16, Q_C = 32 };
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint4 load16u_(const void* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint32_t movemask8_(uint64_t t) {
    uint64_t x = (t >> 7) & 0x0101010101010101ull;
    x |= x >> 7; x |= x >> 14; x |= x >> 28;
    return (uint32_t)x & 0xFFu;
}
}  // namespace

__global__ __launch_bounds__(FIN_TPB) void fin_presence_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, uint32_t n_reads,
                                                                int strands, FinReadDesc* desc_out, uint32_t* work_counter) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    const char* const blk_base = (const char*)ix.blocks;

    uint32_t pc = S_READ0;
    uint32_t il = 0, ir = 0;
    int kstart = 0, end = 0;
    uint64_t r_pk = 0; uint32_t r_len = 0, r_id = 0, r_nch = 0, r_out = 0; bool rev = false;
    uint32_t hit = 0;   // bit 0: forward strand holds a k-mer, bit 1: reverse strand
    uint32_t cur_c = 0;
    int ch_idx = -1, nx_idx = -1; uint64_t bcodes = 0, ncodes = 0; uint32_t bvalid = 0, nvalid = 0;
    uint32_t dflags = 0; int dlen = 0;
    uint32_t budget = 0;
    const uint32_t WNONE = n + 64u;
    uint32_t wtag = WNONE, q_wtag = 0; uint64_t wlo = 0, whi = 0;
    uint32_t ctag = NONE, q_ctag = 0; uint64_t cth0 = 0, cth1 = 0;
    uint32_t rtagA = NONE, rtagB = NONE; uint64_t rplA = 0, rplB = 0; uint32_t rbsA = 0, rbsB = 0;
    uint4 aux = make_uint4(0, 0, 0, 0);
    const void* q_aux = nullptr;
    uint32_t q = 0;

    auto win_place = [&](uint32_t pos, uint32_t below) -> uint32_t {
        const int bs = (int)(pos & ~63u);
        return (uint32_t)min(max((int)pos - (int)below, bs), bs + 48);
    };
    auto req_win = [&](uint32_t ws) { q_wtag = ws; wtag = WNONE; q |= Q_W; };
    auto in_win = [&](uint32_t pos) -> bool { return pos - wtag < 16u; };
    auto req_recs = [&](uint32_t l, uint32_t r, uint32_t c) {
        const uint32_t ta = ((l >> 6) << 2) | c, tb = ((r >> 6) << 2) | c;
        const bool ta_inA = rtagA == ta, ta_inB = rtagB == ta;
        const bool ldA_ta = !ta_inA && !ta_inB;
        const bool ta_atA = ta_inA || ldA_ta;
        const bool tb_toB = tb != ta && ta_atA && rtagB != tb;
        const bool tb_toA = tb != ta && !ta_atA && rtagA != tb;
        rtagA = ldA_ta ? ta : (tb_toA ? tb : rtagA);
        rtagB = tb_toB ? tb : rtagB;
        q |= ((ldA_ta || tb_toA) ? (uint32_t)Q_RA : 0u) | (tb_toB ? (uint32_t)Q_RB : 0u);
    };
    // update_sbwt_interval (formula common.hh:26-36): 0 = data missing (requested), 1 = ok, 2 = (-1,-1)
    auto extend_try = [&](uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) -> int {
        if (l == 0 && r == n - 1) {
            nl = c == 0 ? ix.C[0] : c == 1 ? ix.C[1] : c == 2 ? ix.C[2] : ix.C[3];
            nr = (c == 0 ? ix.C[1] : c == 1 ? ix.C[2] : c == 2 ? ix.C[3] : ix.C[4]) - 1;
            return nl <= nr ? 1 : 2;
        }
        if (q & (Q_RA | Q_RB)) return 0;
        const uint32_t tl = ((l >> 6) << 2) | c, tr = ((r >> 6) << 2) | c;
        const bool lA = tl == rtagA, lB = tl == rtagB, rA = tr == rtagA, rB = tr == rtagB;
        if (!((lA || lB) && (rA || rB))) { req_recs(l, r, c); return 0; }
        const uint64_t pl = lA ? rplA : rplB, pr = rA ? rplA : rplB;
        const uint32_t bl = lA ? rbsA : rbsB, br = rA ? rbsA : rbsB;
        nl = bl + (uint32_t)__popcll(pl & ~(~0ull << (l & 63u)));
        const uint32_t re = br + (uint32_t)__popcll(pr & (~0ull >> (63 - (r & 63u))));
        nr = re - 1;
        return nl < re ? 1 : 2;
    };
    // drop_first_char (common.hh:38-48), byte-window step; see fin_kernel_v2.hip
    auto drop_step = [&](uint32_t& l, uint32_t& r, int new_len) -> bool {
        const bool avail = !(q & Q_W);
        const uint64_t trep = (uint64_t)(uint32_t)new_len * 0x0101010101010101ull;
        const uint32_t lt = (movemask8_(~(((wlo & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) |
                             (movemask8_(~(((whi & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) << 8));
        const uint32_t jd = l - wtag;
        const bool d_in = avail && jd < 16u;
        const uint32_t md = lt & (0xFFFFu >> (15u - (jd & 15u)));
        const bool d_open = !(dflags & 1u) && l != 0;
        const uint32_t l_new = md ? wtag + (31u - (uint32_t)__clz((int)md)) : wtag - 1u;
        const bool d_done = !d_open || (d_in && md != 0);
        l = (d_open && d_in) ? l_new : l;
        const uint32_t ju = r + 1u - wtag;
        const bool u_in = avail && ju < 16u;
        const uint32_t mu = (lt & (0xFFFFu << (ju & 15u))) & 0xFFFFu;
        const bool u_open = !(dflags & 2u) && r < n - 1u;
        uint32_t r_new = mu ? wtag + ((uint32_t)__ffs((int)mu) - 1u) - 1u : wtag + 15u;
        const bool u_clamp = r_new >= n - 1u;
        r_new = u_clamp ? n - 1u : r_new;
        const bool u_done = !u_open || (u_in && (mu != 0 || u_clamp));
        r = (u_open && u_in) ? r_new : r;
        dflags = (d_done ? 1u : 0u) | (u_done ? 2u : 0u);
        const bool done = d_done && u_done;
        const bool want = !done && avail;
        const uint32_t ws = !d_done ? win_place(l, 15) : win_place(r + 1u, 0);
        q_wtag = want ? ws : q_wtag;
        wtag = want ? WNONE : wtag;
        q |= want ? (uint32_t)Q_W : 0u;
        return done;
    };
    // thermometer-plane form for thresholds lcs_t0+1..lcs_t0+3; see fin_kernel_v2.hip
    auto drop_coarse = [&](uint32_t& l, uint32_t& r, int new_len) -> bool {
        const int d = new_len - (int)ix.lcs_t0;
        const bool rng = (uint32_t)(d - 1) < 3u;
        const uint64_t lt = ~(d <= 1 ? (cth1 | cth0) : (d == 2 ? cth1 : (cth1 & cth0)));
        const bool d_open = l != 0, d_can = rng && (l >> 6) == ctag;
        const uint64_t md = lt & (~0ull >> (63u - (l & 63u)));
        const uint32_t l_new = md ? (l & ~63u) + 63u - (uint32_t)__clzll((long long)md) : (l & ~63u) - 1u;
        const bool d_done = !d_open || (d_can && md != 0);
        l = (d_open && d_can) ? l_new : l;
        const uint32_t p = r + 1u;
        const bool u_open = r < n - 1u, u_can = rng && (p >> 6) == ctag;
        const uint64_t mu = lt & (~0ull << (p & 63u));
        uint32_t r_new = mu ? (p & ~63u) + (uint32_t)__ffsll((long long)mu) - 2u : (p & ~63u) + 63u;
        const bool u_edge = !mu && r_new >= n - 1u;
        r_new = u_edge ? n - 1u : r_new;
        const bool u_done = !u_open || (u_can && (mu != 0 || u_edge));
        r = (u_open && u_can) ? r_new : r;
        dflags = (d_done ? 1u : 0u) | (u_done ? 2u : 0u);
        return d_done && u_done;
    };
    auto chunk_addr = [&](int ci) -> const void* { return (const void*)(packed + r_pk + (rev ? r_nch : 0u) + (uint32_t)ci); };
    // one extend attempt for the current base; on failure one step of the recovery (kmer_start++, drop_first_char)
    auto ext_block = [&]() {
        if (pc == S_EXT) {
            uint32_t nl, nr;
            const int rc = extend_try(cur_c, il, ir, nl, nr);
            if (rc == 1) {
                il = nl; ir = nr;
                if (end - kstart + 1 == k) { hit |= rev ? 2u : 1u; pc = S_STRAND_END; }   // a present k-mer: this strand needs the full search
                else { end++; pc = end == (int)r_len ? S_STRAND_END : S_ARRIVE; }
            } else if (rc == 2) {
                if (il == 0 && ir == n - 1) { kstart = end + 1; end++; pc = end == (int)r_len ? S_STRAND_END : S_BASE; }   // the base itself is absent
                else {
                    kstart++;
                    const int nlen = end - kstart;
                    if (nlen <= 0) { il = 0; ir = n - 1; }
                    else { dflags = 0; if (!drop_coarse(il, ir, nlen)) { dlen = nlen; pc = S_BDROP; const uint32_t pos = !(dflags & 1u) ? il : ir + 1u;
                                                                       if (!in_win(pos) && !(q & Q_W)) req_win(!(dflags & 1u) ? win_place(il, 15) : win_place(ir + 1u, 0)); } }
                }
            }
        }
    };

    for (;;) {
        if (q & Q_W) { wtag = q_wtag; const uint4 v = load16u_(blk_base + (size_t)(wtag >> 6) * 128 + (wtag & 63u)); wlo = v.x | ((uint64_t)v.y << 32); whi = v.z | ((uint64_t)v.w << 32); }
        if (q & Q_RA) { const FinCharRec v = *(const FinCharRec*)(blk_base + (size_t)(rtagA >> 2) * 128 + 64 + 12 * (rtagA & 3u)); rplA = v.plane_lo | ((uint64_t)v.plane_hi << 32); rbsA = v.base; }
        if (q & Q_RB) { const FinCharRec v = *(const FinCharRec*)(blk_base + (size_t)(rtagB >> 2) * 128 + 64 + 12 * (rtagB & 3u)); rplB = v.plane_lo | ((uint64_t)v.plane_hi << 32); rbsB = v.base; }
        if (q & Q_C) { ctag = q_ctag; const uint4 v = *(const uint4*)(blk_base + (size_t)ctag * 128 + 112); cth0 = v.x | ((uint64_t)v.y << 32); cth1 = v.z | ((uint64_t)v.w << 32); }
        if (q & Q_AUX) aux = load16u_(q_aux);
        if (q & Q_NEXTCHUNK) { ncodes = aux.x | ((uint64_t)aux.y << 32); nvalid = aux.z; }
        q = 0;

        if (pc == S_STRAND_END) {
            if (rev) { rev = false; il = 0; ir = n - 1; kstart = 0; end = 0; ch_idx = -1; nx_idx = -1; pc = S_BASE; }
            else {
                const uint32_t skip = (strands == 1 ? 3u : 1u) & ~hit;
                const uint64_t off = r_pk | ((uint64_t)skip << 62);
                *(uint4*)(desc_out + r_id) = make_uint4((uint32_t)off, (uint32_t)(off >> 32), r_len, r_out);
                pc = S_READ0;
            }
        }
        if (pc == S_READ1) {
            r_pk = aux.x | ((uint64_t)aux.y << 32); r_len = aux.z; r_out = aux.w;
            r_nch = (r_len + 31u) >> 5;
            hit = 0; rev = false;
            if ((int)r_len < k) pc = S_STRAND_END;   // nothing to search: both strands skipped
            else {
                rev = strands == 1;
                il = 0; ir = n - 1; kstart = 0; end = 0; ch_idx = -1; nx_idx = -1;
                budget = r_len > 0x1FFFF00u ? 0xFFFFFFFFu : 64u * r_len + 4096u;
                pc = S_BASE;
            }
        }
        if (pc == S_CHUNKWAIT) { bcodes = aux.x | ((uint64_t)aux.y << 32); bvalid = aux.z; ch_idx = end >> 5; pc = S_BASE; }
        if (pc == S_BDROP) { if (drop_step(il, ir, dlen)) pc = S_EXT; }
        if (pc == S_BASE) {
            const int ci = end >> 5;
            if (ci != ch_idx) {
                if (nx_idx == ci) { bcodes = ncodes; bvalid = nvalid; ch_idx = ci; nx_idx = -1; }
                else { q_aux = chunk_addr(ci); q |= Q_AUX; pc = S_CHUNKWAIT; }
            }
            if (pc == S_BASE) {
                const uint32_t j = (uint32_t)end & 31u;
                if ((bvalid >> j) & 1u) { cur_c = (uint32_t)(bcodes >> (2 * j)) & 3u; pc = S_EXT; }
                else { kstart = end + 1; il = 0; ir = n - 1; end++; if (end == (int)r_len) pc = S_STRAND_END; }   // non-ACGT base: matches nothing
            }
        }
        ext_block();
        ext_block();
        if (pc == S_ARRIVE) {
            pc = S_BASE;
            if (!(il == 0 && ir == n - 1)) {
                if ((il >> 6) != ctag) { q_ctag = il >> 6; ctag = NONE; q |= Q_C; }
                const int ci = end >> 5; const uint32_t j = (uint32_t)end & 31u;
                uint32_t cn = 4;
                if (ci == ch_idx) { if ((bvalid >> j) & 1u) cn = (uint32_t)(bcodes >> (2 * j)) & 3u; }
                else if (ci == nx_idx) { if ((nvalid >> j) & 1u) cn = (uint32_t)(ncodes >> (2 * j)) & 3u; }
                if (cn < 4) req_recs(il, ir, cn);
            }
            if (nx_idx < 0 && ch_idx >= 0 && (uint32_t)(ch_idx + 1) < r_nch) { nx_idx = ch_idx + 1; q_aux = chunk_addr(nx_idx); q |= Q_AUX | Q_NEXTCHUNK; }
        }
        // exit condition every lane reaches: a read that exceeds its epoch budget is passed on unfiltered (the full kernel decides)
        if (pc > S_STRAND_END) {
            if (budget == 0) { hit = 3u; rev = false; q = 0; pc = S_STRAND_END; }
            else budget--;
        }
        {
            const bool need = pc == S_READ0;
            const uint64_t m = __ballot(need);
            if (m) {
                uint32_t basev = 0;
                const int leader = __ffsll((long long)m) - 1;
                if ((int)lane == leader) basev = atomicAdd(work_counter, (uint32_t)__popcll(m));
                basev = __shfl(basev, leader);
                if (need) {
                    r_id = basev + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                    if (r_id < n_reads) { q_aux = (const void*)(desc + r_id); q |= Q_AUX; pc = S_READ1; }
                    else pc = S_DONE;
                }
            }
        }
        if (!__any(pc != S_DONE)) break;
    }
}

extern "C" int fin_launch_presence(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t n_reads, int strands,
                                   FinReadDesc* desc_out, uint32_t* work_counter, uint32_t grid_blocks, hipStream_t stream) {
    if (n_reads == 0) return 0;
    hipError_t e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    const uint32_t need = (n_reads + FIN_TPB - 1) / FIN_TPB;
    const uint32_t grid = grid_blocks < need ? grid_blocks : need;
    hipLaunchKernelGGL(fin_presence_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, n_reads, strands, desc_out,
                       work_counter);
    return (int)hipGetLastError();
}

extern "C" int fin_presence_blocks_per_cu(void) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fin_presence_kernel, FIN_TPB, 0) != hipSuccess || nb < 1) nb = 4;
    return nb;
}
