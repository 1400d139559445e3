// fin_kernel_v1.hip -- the tuned gfx950 kernel of the search-fmin path ("v1").
//
// Same reference semantics as the plain kernel in fin_kernels.hip (rarest_fmin_streaming_search common.hh:78-186,
// FinimizerIndex::search FinimizerIndex.hh:119-185 with walk_in_unitigs :47-102 fused in streaming form, strand
// merge search_fmin.hh:54-60), restructured for how CDNA4 executes it:
//
//  * The profile of the plain kernel (profiles/r01_v0) shows a wave spending 77 % of its cycles parked on
//    ~180 serialized, divergent loads per base.  Here every lane is a small state machine and the wave runs
//    "epochs": at the top of an epoch every lane issues the (at most six) loads its next piece of work needs,
//    the wave waits ONCE, then every lane runs ALU-only phases until it needs memory again.  Lanes are not in
//    lockstep per base, so one lane's slow base (mismatch recovery, dictionary lookup) does not stall 63 others,
//    and each lane always has a load in flight: memory-level parallelism is 64 per wave instead of a handful.
//  * Steady state is ONE epoch per base touching ONE 128-byte node block: right after an extend lands on its new
//    interval the lane requests the 16 LCS bytes around it (for this base's drop_first_char scans and the Ustart
//    probe) together with the bit-plane word + rank base of the NEXT base's character in the same block.
//  * drop_first_char scans 16 LCS bytes at a time in registers (SWAR compare + clz/ctz) instead of a byte loop.
//  * Results leave as runs: a hit produced by the unitig walk only lengthens the lane's current run; when a run
//    ends the whole wave writes it out cooperatively as contiguous 8-byte pairs (512-B bursts).  The output is
//    pre-filled with (-1,-1); the reverse strand is processed first so forward hits overwrite (the merge rule).
//  * Lanes pull reads from a global counter (work queue), so a wave never idles on its slowest read; the grid
//    is sized to the chip, every lane exits when the counter passes n_reads.
//  * The sliding-window candidate deque lives in LDS, 16 slots per lane ([slot][lane] layout, conflict free);
//    a read that needs more is redone by the overflow kernel with the deque in global memory.
#include "fin_device.h"
#include "fin_kernels.h"

namespace {

enum : uint32_t {
    P_DONE = 0, P_READ0, P_READ1, P_STRAND_END, P_CHUNKWAIT, P_BASE, P_EXTI, P_EXTI_DROP, P_EXTK, P_EXTK_DROP,
    P_SHRINK, P_SHRINK_DROP, P_USTART, P_KMER, P_KMER_DROP, P_OUT, P_TEXTWAIT, P_RES0, P_RES1, P_RES2, P_RES3, P_RES4, P_NEXT
};
enum : uint32_t { Q_WA = 1, Q_WB = 2, Q_RA = 4, Q_RB = 8, Q_AUX = 16, Q_AUX2 = 32, Q_NEXTCHUNK = 64 };
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t lowbytes(uint32_t nb) { return nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1); }

// 16 ASCII bases -> 2-bit codes (A0 C1 G2 T3, base j at bits 2j) + validity bits; case-insensitive
__device__ __forceinline__ uint4 load16u(const void* p) {   // 16 bytes from any byte address (one global_load_dwordx4)
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

__device__ __forceinline__ void decode_chunk(const uint4 v, bool rev, uint32_t& codes, uint32_t& valid) {
    uint32_t c = 0, ok = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t wsel = (i >> 2) == 0 ? v.x : (i >> 2) == 1 ? v.y : (i >> 2) == 2 ? v.z : v.w;
        uint32_t b = (wsel >> (8 * (i & 3))) & 0xDFu;
        uint32_t y = (b >> 1) & 3u;
        y ^= y >> 1;
        uint32_t good = ((0x0010008Au >> (b & 31u)) & 1u) & (uint32_t)((b & 0xE0u) == 0x40u);
        c |= y << (2 * i);
        ok |= good << i;
    }
    if (rev) {   // position j of rc(read) is byte 15-j complemented
        uint32_t r = __brev(~c);
        c = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
        ok = __brev(ok) >> 16;
    }
    codes = c; valid = ok;
}

}  // namespace

__global__ __launch_bounds__(FIN_TPB) void fin_search_v1_kernel(FinDevIndex ix, const uint8_t* bases, const FinReadDesc* desc, int2* out,
                                                                 uint32_t n_reads, int strands, uint32_t dq_limit, uint32_t* ovf_list,
                                                                 uint32_t* ovf_count, uint32_t* work_counter) {
    __shared__ uint64_t lds_dq[16 * FIN_TPB];
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t* const dq = lds_dq + threadIdx.x;
#define DQ(i) dq[((i) & 15u) * FIN_TPB]
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    const char* const blk_base = (const char*)ix.blocks;

    // ---- per-lane state -------------------------------------------------------------------------------------
    uint32_t pc = P_READ0;
    uint32_t il = 0, ir = 0, kl = 0, kr = 0;
    int start = 0, kstart = 0, end = 0, bu_end = -1;
    uint32_t bu_colex = 0, dq_head = 0, dq_cnt = 0;
    bool walk = false; uint32_t wg = 0, w_u = 0, w_ustart = 0, w_uend = 0;
    uint32_t run_pos = 0, run_len = 0, run_u = 0, run_off = 0;
    bool pend = false, pend_rev = false; uint32_t pend_pos = 0, pend_len = 0, pend_u = 0, pend_off = 0;
    uint64_t r_off = 0; uint32_t r_len = 0, r_out = 0, r_id = 0; int r_nk = 0; bool rev = false;
    uint32_t cur_c = 0;
    int ch_idx = -1, nx_idx = -1; uint32_t bcodes = 0, bvalid = 0, ncodes = 0, nvalid = 0;
    bool found = false, use_branch = false; uint32_t fin_end = 0, fin_colex = 0;
    bool have_cand = false; uint32_t cand_len = 0, cand_colex = 0;
    uint32_t dflags = 0, res_g = 0, res_idx = 0;
    // register caches of index data
    uint32_t wtagA = NONE, wtagB = NONE; uint64_t wAlo = 0, wAhi = 0, wBlo = 0, wBhi = 0;   // 16 node bytes of group tag
    uint32_t rtagA = NONE, rtagB = NONE; uint64_t rplA = 0, rplB = 0; uint32_t rbsA = 0, rbsB = 0;   // tag = block*4 + char
    uint32_t ttag = NONE; uint4 wt = make_uint4(0, 0, 0, 0);   // 64 bases of unitig text, tag = position >> 6
    uint4 aux = make_uint4(0, 0, 0, 0), aux2 = make_uint4(0, 0, 0, 0);
    const void* q_aux = nullptr; const void* q_aux2 = nullptr;
    uint32_t q = 0;

    auto win_lookup = [&](uint32_t g, uint64_t& lo, uint64_t& hi) -> bool {
        if (g == wtagA) { lo = wAlo; hi = wAhi; return true; }
        if (g == wtagB) { lo = wBlo; hi = wBhi; return true; }
        return false;
    };
    auto rec_lookup = [&](uint32_t tag, uint64_t& pl, uint32_t& bs) -> bool {
        if (tag == rtagA) { pl = rplA; bs = rbsA; return true; }
        if (tag == rtagB) { pl = rplB; bs = rbsB; return true; }
        return false;
    };
    // request the windows a scan of [l, r] needs: group of l into A, group of r+1 into B
    // (value selects, no conditional stores to different variables: keeps every cache tag in a register)
    auto req_wins = [&](uint32_t l, uint32_t r) {
        const uint32_t gd = l >> 4, gu = (r + 1 < n ? r + 1 : r) >> 4;
        const bool gd_inA = wtagA == gd, gd_inB = wtagB == gd;
        const bool ldA_gd = !gd_inA && !gd_inB;
        const bool gd_atA = gd_inA || ldA_gd;
        const bool gu_toB = gu != gd && gd_atA && wtagB != gu;
        const bool gu_toA = gu != gd && !gd_atA && wtagA != gu;
        wtagA = ldA_gd ? gd : (gu_toA ? gu : wtagA);
        wtagB = gu_toB ? gu : wtagB;
        q |= ((ldA_gd || gu_toA) ? (uint32_t)Q_WA : 0u) | (gu_toB ? (uint32_t)Q_WB : 0u);
    };
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
    // update_sbwt_interval on [l, r] with the cached records; returns 0 = data missing (requested), 1 = ok, 2 = (-1,-1)
    auto extend_try = [&](uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) -> int {
        if (l == 0 && r == n - 1) {
            nl = c == 0 ? ix.C[0] : c == 1 ? ix.C[1] : c == 2 ? ix.C[2] : ix.C[3];
            nr = (c == 0 ? ix.C[1] : c == 1 ? ix.C[2] : c == 2 ? ix.C[3] : ix.C[4]) - 1;
            return nl <= nr ? 1 : 2;
        }
        uint64_t pl, pr; uint32_t bl, br;
        bool okl = rec_lookup(((l >> 6) << 2) | c, pl, bl), okr = rec_lookup(((r >> 6) << 2) | c, pr, br);
        if (!(okl && okr)) { req_recs(l, r, c); return 0; }
        uint32_t ol = l & 63u, orr = r & 63u;
        nl = bl + (uint32_t)__popcll(pl & (ol ? (~0ull >> (64 - ol)) : 0ull));
        uint32_t re = br + (uint32_t)__popcll(pr & (~0ull >> (63 - orr)));
        nr = re - 1;
        return nl < re ? 1 : 2;
    };
    // drop_first_char (common.hh:38-48), new_len >= 1, resumable: progress lives in l, r and dflags
    auto drop_try = [&](uint32_t& l, uint32_t& r, int new_len) -> bool {
        const uint64_t trep = (uint64_t)(uint32_t)new_len * 0x0101010101010101ull;
        while (!(dflags & 1u)) {
            if (l == 0) { dflags |= 1u; break; }
            uint32_t g = l >> 4; uint64_t lo, hi;
            if (!win_lookup(g, lo, hi)) { req_wins(l, r); return false; }
            uint64_t ltl = (~(((lo & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) & 0x8080808080808080ull;
            uint64_t lth = (~(((hi & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) & 0x8080808080808080ull;
            uint32_t j0 = l & 15u;
            uint64_t mh = j0 >= 8 ? (lth & lowbytes(j0 - 7)) : 0ull;
            uint64_t ml = j0 >= 8 ? ltl : (ltl & lowbytes(j0 + 1));
            if (mh) { l = (g << 4) + 8 + ((63 - (uint32_t)__clzll((long long)mh)) >> 3); dflags |= 1u; }
            else if (ml) { l = (g << 4) + ((63 - (uint32_t)__clzll((long long)ml)) >> 3); dflags |= 1u; }
            else l = (g << 4) - 1;   // LCS[0] = 0 stops the scan in group 0, so g > 0 here
        }
        while (!(dflags & 2u)) {
            if (r >= n - 1) { dflags |= 2u; break; }
            uint32_t p = r + 1, g = p >> 4; uint64_t lo, hi;
            if (!win_lookup(g, lo, hi)) { req_wins(l, r); return false; }
            uint64_t ltl = (~(((lo & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) & 0x8080808080808080ull;
            uint64_t lth = (~(((hi & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) & 0x8080808080808080ull;
            uint32_t j0 = p & 15u;
            uint64_t ml = j0 < 8 ? (ltl & ~lowbytes(j0)) : 0ull;
            uint64_t mh = j0 < 8 ? lth : (lth & ~lowbytes(j0 - 8));
            if (ml) { r = (g << 4) + ((uint32_t)__ffsll((long long)ml) - 1) / 8 - 1; dflags |= 2u; }
            else if (mh) { r = (g << 4) + 8 + ((uint32_t)__ffsll((long long)mh) - 1) / 8 - 1; dflags |= 2u; }
            else { r = (g << 4) + 15; if (r >= n - 1) { r = n - 1; dflags |= 2u; } }
        }
        return true;
    };
    auto close_run = [&]() {
        if (run_len) { pend = true; pend_rev = rev; pend_pos = run_pos; pend_len = run_len; pend_u = run_u; pend_off = run_off; run_len = 0; }
    };
    auto strand_init = [&]() {
        il = 0; ir = n - 1; kl = 0; kr = n - 1; start = 0; kstart = 0; end = 0; bu_end = -1;
        dq_head = 0; dq_cnt = 0; walk = false; run_len = 0; ch_idx = -1; nx_idx = -1;
    };
    auto chunk_addr = [&](int ci) -> const void* {
        return rev ? (const void*)(bases + r_off + r_len - 16u * (uint32_t)(ci + 1)) : (const void*)(bases + r_off + 16u * (uint32_t)ci);
    };

    for (;;) {
        // ================= 1. serve this epoch's requests: all loads issue back to back, one wait =================
        if (q & Q_WA) { uint4 v = *(const uint4*)(blk_base + (size_t)(wtagA >> 2) * 128 + (wtagA & 3u) * 16); wAlo = v.x | ((uint64_t)v.y << 32); wAhi = v.z | ((uint64_t)v.w << 32); }
        if (q & Q_WB) { uint4 v = *(const uint4*)(blk_base + (size_t)(wtagB >> 2) * 128 + (wtagB & 3u) * 16); wBlo = v.x | ((uint64_t)v.y << 32); wBhi = v.z | ((uint64_t)v.w << 32); }
        if (q & Q_RA) { const char* b = blk_base + (size_t)(rtagA >> 2) * 128; rplA = *(const uint64_t*)(b + 64 + 8 * (rtagA & 3u)); rbsA = *(const uint32_t*)(b + 96 + 4 * (rtagA & 3u)); }
        if (q & Q_RB) { const char* b = blk_base + (size_t)(rtagB >> 2) * 128; rplB = *(const uint64_t*)(b + 64 + 8 * (rtagB & 3u)); rbsB = *(const uint32_t*)(b + 96 + 4 * (rtagB & 3u)); }
        if (q & Q_AUX) aux = load16u(q_aux);
        if (q & Q_AUX2) aux2 = load16u(q_aux2);
        if (q & Q_NEXTCHUNK) { decode_chunk(aux, rev, ncodes, nvalid); }
        q = 0;

        // ================= 2. phases, in the order a base flows through them =================
        if (pc == P_STRAND_END) {
            close_run();
            if (rev) { rev = false; strand_init(); pc = P_BASE; }
            else pc = P_READ0;
        }
        if (pc == P_READ1) {   // descriptor arrived
            r_off = aux.x | ((uint64_t)aux.y << 32); r_len = aux.z; r_out = aux.w;
            r_nk = (int)r_len - k + 1;
            if (r_nk <= 0) pc = P_READ0;
            else { rev = strands == 1; strand_init(); pc = P_BASE; }
        }
        if (pc == P_CHUNKWAIT) { decode_chunk(aux, rev, bcodes, bvalid); ch_idx = end >> 4; pc = P_BASE; }

        // ---- shortest-unique shrink loop + candidate insertion (common.hh:145-164) ----
        if (pc == P_SHRINK) {
            // window bookkeeping first: drop candidates that start before the k-mer window (eager form of :173-176)
            while (dq_cnt) {
                uint64_t f = DQ(dq_head);
                int fs = (int)dq_end(f, (uint32_t)end) - (int)dq_len(f) + 1;
                if (fs < kstart) { dq_head++; dq_cnt--; } else break;
            }
            have_cand = false;
        }
        while (pc == P_SHRINK || pc == P_SHRINK_DROP) {
            if (pc == P_SHRINK) {
                if (il != ir) {
                    if (have_cand) {
                        uint64_t cand = dq_pack(cand_len, cand_colex, (uint32_t)end);
                        if (dq_cnt && (DQ(dq_head) >> 24) > (cand >> 24)) dq_cnt = 0;
                        else while (dq_cnt && (DQ(dq_head + dq_cnt - 1) >> 24) > (cand >> 24)) dq_cnt--;
                        if (dq_cnt >= dq_limit) {   // more live candidates than LDS slots: hand the read to the overflow kernel
                            uint32_t slot = atomicAdd(ovf_count, 1u); ovf_list[slot] = r_id;
                            run_len = 0; pc = P_READ0; break;
                        }
                        DQ(dq_head + dq_cnt) = cand; dq_cnt++;
                    }
                    pc = P_USTART; break;
                }
                have_cand = true; cand_len = (uint32_t)(end - start + 1); cand_colex = il;
                start++;
                if (end - start + 1 <= 0) { il = 0; ir = n - 1; continue; }
                dflags = 0; pc = P_SHRINK_DROP;
            }
            if (!drop_try(il, ir, end - start + 1)) break;
            pc = P_SHRINK;
        }
        // ---- Ustart probe (common.hh:167) ----
        if (pc == P_USTART) {
            if (kl == kr) {
                uint64_t lo, hi;
                if (win_lookup(kl >> 4, lo, hi)) {
                    uint32_t j = kl & 15u;
                    uint32_t byte = (uint32_t)((j < 8 ? lo : hi) >> (8 * (j & 7u))) & 0xFFu;
                    if (byte & FIN_USTART_BIT) { bu_end = end; bu_colex = kl; }
                    pc = P_KMER;
                } else req_wins(kl, kr);
            } else pc = P_KMER;
        }
        // ---- k-mer present? (common.hh:170-182) ----
        if (pc == P_KMER) {
            found = false;
            if (end - kstart + 1 == k) {
                if (dq_cnt) {
                    uint64_t w = DQ(dq_head);
                    found = true; fin_end = dq_end(w, (uint32_t)end); fin_colex = dq_colex(w);
                    use_branch = bu_end >= (int)fin_end;
                }
                kstart++;
                if (end - kstart + 1 <= 0) { kl = 0; kr = n - 1; pc = P_OUT; }
                else { dflags = 0; pc = P_KMER_DROP; }
            } else pc = P_OUT;
        }
        if (pc == P_KMER_DROP) { if (drop_try(kl, kr, end - kstart + 1)) pc = P_OUT; }

        // ---- resolve + walk (FinimizerIndex.hh:148-183, :47-102) ----
        if (pc == P_TEXTWAIT) { wt = aux; pc = P_OUT; }
        if (pc == P_OUT) {
            if (end >= k - 1) {
                bool walk_hit = false, need_text = false;
                if (walk && wg + 1 < w_uend && cur_c < 4) {
                    uint32_t g1 = wg + 1;
                    if ((g1 >> 6) != ttag) { need_text = true; ttag = g1 >> 6; q_aux = (const void*)(ix.concat + ((size_t)(g1 >> 6) << 2)); q |= Q_AUX; pc = P_TEXTWAIT; }
                    else {
                        uint32_t wsel = (g1 >> 4) & 3u;
                        uint32_t word = wsel == 0 ? wt.x : wsel == 1 ? wt.y : wsel == 2 ? wt.z : wt.w;
                        walk_hit = ((word >> (2 * (g1 & 15u))) & 3u) == cur_c;
                    }
                }
                if (!need_text) {
                    if (walk_hit) { wg++; run_len++; pc = P_NEXT; }
                    else if (found) pc = P_RES0;
                    else { walk = false; close_run(); pc = P_NEXT; }
                }
            } else pc = P_NEXT;
        }
        if (pc == P_RES4) {   // aux = ends_p[res_idx .. res_idx+3]
            const uint32_t gs = res_g - (uint32_t)(k - 1);
            bool done = true;
            if (gs < aux.y) { w_u = res_idx; w_ustart = aux.x; w_uend = aux.y; }
            else if (gs < aux.z) { w_u = res_idx + 1; w_ustart = aux.y; w_uend = aux.z; }
            else if (gs < aux.w) { w_u = res_idx + 2; w_ustart = aux.z; w_uend = aux.w; }
            else { res_idx += 3; q_aux = (const void*)(ix.ends + res_idx); q |= Q_AUX; done = false; }
            if (done) {
                close_run();
                run_pos = (uint32_t)(end - (k - 1)); run_len = 1; run_u = w_u; run_off = gs - w_ustart;
                walk = true; wg = res_g;
                pc = P_NEXT;
            }
        }
        if (pc == P_RES3) { res_idx = aux.x; q_aux = (const void*)(ix.ends + res_idx); q |= Q_AUX; pc = P_RES4; }
        if (pc == P_RES2) {   // aux.x = global_offsets[rank] (common.hh:71) or the unitig start (common.hh:65)
            res_g = use_branch ? aux.x + (uint32_t)(k - 1) + (uint32_t)(end - bu_end) : aux.x + (uint32_t)end - fin_end;
            const uint32_t gs = res_g - (uint32_t)(k - 1);
            if (gs < ix.total_len) { q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = P_RES3; }
            else { walk = false; close_run(); pc = P_NEXT; }   // unreachable on a consistent index (the reference reads out of bounds here): reported as absent
        }
        if (pc == P_RES1) {   // aux = {fmin_mask, ustart_mask} of the block, aux2 = its {ustart_rank, fmin_rank}
            const uint32_t colex = use_branch ? bu_colex : fin_colex;
            const uint32_t o = colex & 63u;
            const uint64_t below = o ? (~0ull >> (64 - o)) : 0ull;
            const uint64_t fm = aux.x | ((uint64_t)aux.y << 32), um = aux.z | ((uint64_t)aux.w << 32);
            const uint32_t rank = use_branch ? aux2.x + (uint32_t)__popcll(um & below) : aux2.y + (uint32_t)__popcll(fm & below);
            q_aux = use_branch ? (const void*)(ix.ends + rank) : (const void*)(ix.goff + rank);
            q |= Q_AUX; pc = P_RES2;
        }
        if (pc == P_RES0) {
            const uint32_t colex = use_branch ? bu_colex : fin_colex;
            q_aux = (const void*)(blk_base + (size_t)(colex >> 6) * 128 + 112);
            q_aux2 = (const void*)(ix.blkrank + (colex >> 6));
            q |= Q_AUX | Q_AUX2; pc = P_RES1;
        }

        if (pc == P_NEXT) {
            end++;
            pc = end == (int)r_len ? P_STRAND_END : P_BASE;
        }
        // ---- next base ----
        if (pc == P_BASE) {
            const int ci = end >> 4;
            if (ci != ch_idx) {
                if (nx_idx == ci) { bcodes = ncodes; bvalid = nvalid; ch_idx = ci; nx_idx = -1; }
                else { q_aux = chunk_addr(ci); q |= Q_AUX; pc = P_CHUNKWAIT; }
            }
            if (pc == P_BASE) {
                const uint32_t j = (uint32_t)end & 15u;
                if ((bvalid >> j) & 1u) { cur_c = (bcodes >> (2 * j)) & 3u; pc = P_EXTI; }
                else {
                    // non-ACGT base: defined behaviour (reference: UB) = matches nothing, the state the reference's own
                    // `start > end` reset produces (common.hh:118-122)
                    cur_c = 4; start = end + 1; kstart = end + 1; il = 0; ir = n - 1; kl = 0; kr = n - 1; dq_cnt = 0;
                    found = false; pc = P_OUT;
                }
            }
        }
        // ---- (1) finimizer interval (common.hh:114-127) ----
        while (pc == P_EXTI || pc == P_EXTI_DROP) {
            if (pc == P_EXTI) {
                uint32_t nl, nr;
                int rc = extend_try(cur_c, il, ir, nl, nr);
                if (rc == 0) break;
                if (rc == 1) { il = nl; ir = nr; pc = P_EXTK; break; }
                kstart = ++start;
                if (start > end) { il = 0; ir = n - 1; pc = P_EXTK; break; }
                if (end - start <= 0) { il = 0; ir = n - 1; continue; }
                dflags = 0; pc = P_EXTI_DROP;
            }
            if (!drop_try(il, ir, end - start)) break;
            pc = P_EXTI;
        }
        // ---- (2) k-mer interval (common.hh:132-143) ----
        while (pc == P_EXTK || pc == P_EXTK_DROP) {
            if (pc == P_EXTK) {
                if (start == kstart) { kl = il; kr = ir; pc = P_SHRINK; break; }
                uint32_t nl, nr;
                int rc = extend_try(cur_c, kl, kr, nl, nr);
                if (rc == 0) break;
                if (rc == 1) { kl = nl; kr = nr; pc = P_SHRINK; break; }
                kstart++;
                if (end - kstart <= 0) { kl = 0; kr = n - 1; continue; }
                dflags = 0; pc = P_EXTK_DROP;
            }
            if (!drop_try(kl, kr, end - kstart)) break;
            pc = P_EXTK;
        }
        // ---- arrival at the new interval: ask for everything the rest of this base and the next extend need ----
        if (pc == P_SHRINK && q == 0) {
            if (!(il == 0 && ir == n - 1)) {
                req_wins(il, ir);
                const int e1 = end + 1;
                if (e1 < (int)r_len) {
                    const int ci = e1 >> 4; const uint32_t j = (uint32_t)e1 & 15u;
                    uint32_t cn = 4;
                    if (ci == ch_idx) { if ((bvalid >> j) & 1u) cn = (bcodes >> (2 * j)) & 3u; }
                    else if (ci == nx_idx) { if ((nvalid >> j) & 1u) cn = (ncodes >> (2 * j)) & 3u; }
                    if (cn < 4) req_recs(il, ir, cn);
                }
            }
            if (nx_idx < 0 && ch_idx >= 0 && (uint32_t)(ch_idx + 1) * 16u < r_len && !(q & Q_AUX)) {
                nx_idx = ch_idx + 1; q_aux = chunk_addr(nx_idx); q |= Q_AUX | Q_NEXTCHUNK;
            }
        }

        // ================= 3. cooperative write-out of finished runs (wave-wide, converged) =================
        {
            uint64_t m = __ballot(pend);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t o_base = __shfl(r_out, src), o_nk = (uint32_t)__shfl(r_nk, src);
                const uint32_t p_pos = __shfl(pend_pos, src), p_len = __shfl(pend_len, src);
                const uint32_t p_u = __shfl(pend_u, src), p_off = __shfl(pend_off, src);
                const bool p_rev = __shfl((int)pend_rev, src) != 0;
                for (uint32_t i = lane; i < p_len; i += 64) {
                    uint32_t idx = p_rev ? (o_nk - 1 - (p_pos + i)) : (p_pos + i);
                    out[(size_t)o_base + idx] = make_int2((int)p_u, (int)(p_off + i));
                }
            }
            pend = false;
        }
        // ================= 4. work queue =================
        {
            const bool need = pc == P_READ0;
            const uint64_t m = __ballot(need);
            if (m) {
                uint32_t basev = 0;
                const int leader = __ffsll((long long)m) - 1;
                if ((int)lane == leader) basev = atomicAdd(work_counter, (uint32_t)__popcll(m));
                basev = __shfl(basev, leader);
                if (need) {
                    r_id = basev + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                    if (r_id < n_reads) { q_aux = (const void*)(desc + r_id); q |= Q_AUX; pc = P_READ1; }
                    else pc = P_DONE;
                }
            }
        }
        if (!__any(pc != P_DONE)) break;
    }
#undef DQ
}

extern "C" int fin_launch_search_v1(const FinDevIndex* ix, const uint8_t* bases, const FinReadDesc* desc, const uint64_t* offs,
                                    const uint64_t* out_offs, void* out, uint64_t n_kmers, uint32_t n_reads, int strands,
                                    uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count, uint32_t* work_counter,
                                    uint64_t* ovf_scratch, uint32_t ovf_blocks, uint32_t grid_blocks, hipStream_t stream,
                                    hipEvent_t ev0, hipEvent_t ev1) {
    if (n_reads == 0) return 0;
    hipError_t e = hipMemsetAsync(ovf_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(out, 0xFF, n_kmers * 8, stream);   // every slot (-1,-1); runs overwrite
    if (e != hipSuccess) return (int)e;
    uint32_t need = (n_reads + FIN_TPB - 1) / FIN_TPB;
    uint32_t grid = grid_blocks < need ? grid_blocks : need;
    if (ev0) (void)hipEventRecord(ev0, stream);
    hipLaunchKernelGGL(fin_search_v1_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, bases, desc, (int2*)out, n_reads, strands,
                       lds_deque_limit, ovf_list, ovf_count, work_counter);
    if (ev1) (void)hipEventRecord(ev1, stream);
    return fin_launch_overflow(ix, bases, offs, out_offs, out, strands, ovf_list, ovf_count, ovf_scratch, ovf_blocks, stream);
}

// resident blocks per CU the hardware admits for the tuned kernel (LDS: 32 KiB per block; registers)
extern "C" int fin_v1_blocks_per_cu(void) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fin_search_v1_kernel, FIN_TPB, 0) != hipSuccess || nb < 1) nb = 2;
    return nb;
}
