// fin_kernel_v2.hip -- the tuned gfx950 kernel of the search-fmin path ("v2").
//
// Reference semantics: rarest_fmin_streaming_search (common.hh:78-186), FinimizerIndex::search
// (FinimizerIndex.hh:119-185) with walk_in_unitigs (:47-102) in streaming form, strand merge (search_fmin.hh:54-60).
//
// How it maps onto CDNA4 (the measurements that drove each choice are in profiles/ and CHANGELOG.md):
//  * Epochs.  Every lane is a small state machine; at the top of an epoch each lane issues the few loads its next
//    piece of work needs, the wave waits once, then all lanes run ALU-only blocks.  Lanes are not in lockstep per
//    base, every lane always has a load in flight (64-way memory parallelism per wave), and a lane's slow base
//    does not stall the other 63.  (v0, lockstep per base: 180 serialized loads per base, 77 % of cycles parked.)
//  * One 128-byte node block per base.  Right after an extend lands on its new interval the lane asks for the 16 LCS
//    bytes around it (this base's drop_first_char scans + the Ustart probe) and for the plane word + rank base of
//    the NEXT base's character in the same block.
//  * Flat control flow.  The epoch body is a fixed sequence of guarded straight-line blocks in the order a base flows
//    through them (shrink x2, Ustart, k-mer, output, next base, extend x2, k-mer extend); what does not fit (a third
//    shrink step, a scan leaving its window) simply resumes at the same block next epoch.  (v1 used loops with
//    breaks inside the blocks: 35 % of its instructions were v_mov / exec-mask bookkeeping and it was ALU-bound.)
//  * drop_first_char: thresholds lcs_t0+1..lcs_t0+3 (about 88 % of the scanning calls; chosen per index) are answered from the
//    block's thermometer planes -- a 64-bit stop mask for the whole block, no scan, no extra line; the rest takes a SWAR
//    step on an unaligned 16-byte window of LCS bytes (compare all 16, movemask, clz/ffs) in one shared block per epoch.
//  * Mismatch recovery of the k-mer interval jumps: while the interval is a single node p the reference's loop
//    (common.hh:134-139) cannot succeed until new_len <= max(LCS[p], LCS[p+1]), so kmer_start moves there at once.
//  * Reads arrive packed (2 bits/base + validity, both strands: fin_pack.hip, the first kernel of every step) so the hot loop never decodes ASCII.
//  * Results leave as runs written cooperatively by the wave (512-byte bursts) into a (-1,-1)-prefilled buffer;
//    reverse strand first, forward hits overwrite (the merge rule).  Lanes pull reads from a global work counter.
//  * The candidate deque lives in LDS ([slot][lane]); front and back are mirrored in registers.
#include "fin_device.h"
#include "fin_kernels.h"
#include <cstdio>

// Diagnostic build (-DFIN_STATS): per-lane counters of where epochs go, summed into `stats` at exit.  Never on in the product.
#ifdef FIN_STATS
#define STAT(i) (st[(i)]++)
enum { ST_EPOCH = 0, ST_ARRIVE, ST_REC_I, ST_REC_K, ST_WIN_SHRINK, ST_WIN_KMER, ST_WIN_EXTI, ST_WIN_EXTK, ST_WIN_USTART, ST_WIN_JUMP,
       ST_CHUNK, ST_TEXT, ST_RES, ST_SHRINK4, ST_EXTI4, ST_EXTK_AGAIN, ST_READ, ST_STRAND, ST_N };
#define TSTAMP(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); tacc[(i)] += t_ - tprev; tprev = t_; } while (0)
enum { T_SERVE = 0, T_HEAD, T_USTART_KDROP, T_SHRINK, T_KMERREC, T_OUT_RES, T_BASE, T_EXTI, T_EXTK, T_ARRIVE, T_TAIL, T_N };
#else
#define STAT(i) ((void)0)
#define TSTAMP(i) ((void)0)
#endif
// Second diagnostic build (-DFIN_BLOCKS): for every guarded block, how often a wave executes it and with how many lanes.
#ifdef FIN_BLOCKS
enum { B_STRAND_END = 0, B_READ1, B_BDROP, B_CHUNKWAIT, B_USTART, B_USTART_PROBE, B_KDROP, B_KDROP_ISKM, B_KDROP_SCAN, B_SHRINK1, B_SHRINK2,
       B_SHRINK_CAND, B_SHRINK_POPBACK, B_SHRINK_DROP, B_SHRINK_BDROP, B_KMER, B_TEXTWAIT, B_OUT, B_OUT_WALK, B_OUT_CLOSE, B_RES5, B_RES4, B_RES3, B_RES1, B_RES0,
       B_BASE, B_BASE_CHUNK, B_EXTI1, B_EXTI2, B_EXTI_FAIL, B_EXTI_BDROP, B_EXTK, B_EXTK_EXT, B_EXTK_FAIL, B_EXTK_BDROP, B_ARRIVE, B_ARRIVE_POP, B_WRITEOUT, B_QUEUE, B_EPOCH, B_N };
#define WB(i) do { bl[(i)]++; if ((uint32_t)(__ffsll((long long)__ballot(1)) - 1) == lane) bw[(i)]++; } while (0)
#else
#define WB(i) ((void)0)
#endif

namespace {

enum : uint32_t {
    P_DONE = 0, P_READ0, P_READ1, P_STRAND_END, P_CHUNKWAIT, P_BDROP, P_BASE, P_EXTI, P_EXTK,
    P_ARRIVE, P_SHRINK, P_USTART, P_KMER, P_KMER_DROP0, P_OUT, P_TEXTWAIT, P_RES0, P_RES1, P_RES3, P_RES4, P_RES5
};
enum : uint32_t { Q_W = 1, Q_RA = 2, Q_RB = 4, Q_AUX = 8, Q_NEXTCHUNK = 16, Q_C = 32 };
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint4 load16u(const void* p) {   // 16 bytes from any byte address (one global_load_dwordx4)
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// bit 7 of each byte -> one bit per byte (byte i -> bit i)
__device__ __forceinline__ uint32_t movemask8(uint64_t t) {
    uint64_t x = (t >> 7) & 0x0101010101010101ull;
    x |= x >> 7; x |= x >> 14; x |= x >> 28;
    return (uint32_t)x & 0xFFu;
}

}  // namespace

#ifndef FIN_V2_SHRINK_REPS
#define FIN_V2_SHRINK_REPS 3   // shrink-loop iterations a lane may do per epoch
#endif
#ifndef FIN_V2_EXTI_REPS
#define FIN_V2_EXTI_REPS 2     // extend attempts (failure recovery steps) a lane may do per epoch
#endif
#ifndef FIN_V2_EXTK2
#define FIN_V2_EXTK2 0         // second k-mer-interval extend attempt in the same epoch (no gain since the rejoin case moved into the first)
#endif
#ifndef FIN_V2_RESGUARD
#define FIN_V2_RESGUARD 1   // one test skips all dictionary-lookup stages when no lane is in them
#endif
#ifndef FIN_V2_BELOW
#define FIN_V2_BELOW 7          // LCS bytes the arrival window keeps below the interval's lower end (16 in all)
#endif
#ifndef FIN_V2_WINALWAYS
#define FIN_V2_WINALWAYS 0      // 1: every arrival asks for the LCS window, 0: only lanes whose k-mer interval is a single node
#endif
#ifndef FIN_V2_MINWAVES
#define FIN_V2_MINWAVES 4   // waves per SIMD the register allocator must leave room for
#endif
__global__ __launch_bounds__(FIN_TPB, FIN_V2_MINWAVES) void fin_search_v2_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                                 uint32_t n_reads, int strands, uint32_t dq_limit, uint32_t* ovf_list,
                                                                 uint32_t* ovf_count, uint32_t* work_counter
#if defined(FIN_STATS) || defined(FIN_BLOCKS)
                                                                 , unsigned long long* stats
#endif
                                                                 ) {
    __shared__ uint64_t lds_dq[16 * FIN_TPB];
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t* const dq = lds_dq + threadIdx.x;
#define DQ(i) dq[((i) & 15u) * FIN_TPB]
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    const char* const blk_base = (const char*)ix.blocks;
    // the C array as scalar values (left as `ix.C[c]` the compiler selects a kernarg OFFSET and issues two dependent global loads
    // in the middle of the epoch)
    const uint32_t C0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[0]), C1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[1]),
                   C2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[2]), C3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[3]),
                   C4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[4]);
#ifdef FIN_BLOCKS
    uint32_t bl[B_N] = {0}, bw[B_N] = {0};
#endif
#ifdef FIN_STATS
    uint32_t st[ST_N] = {0};
    uint64_t tacc[T_N] = {0}; uint64_t tprev = __builtin_amdgcn_s_memtime();
#endif

    // ---- per-lane state -------------------------------------------------------------------------------------
    uint32_t pc = P_READ0;
    uint32_t il = 0, ir = 0, kl = 0, kr = 0;
    int start = 0, kstart = 0, end = 0, bu_end = -1;
    uint32_t bu_colex = 0, dq_head = 0, dq_cnt = 0;
    uint64_t dq_front = 0, dq_back = 0;   // register mirrors of DQ(dq_head) and DQ(dq_head + dq_cnt - 1)
    bool walk = false; uint32_t wg = 0, w_u = 0, w_ustart = 0, w_uend = 0;
    uint32_t run_pos = 0, run_len = 0, run_u = 0, run_off = 0;
    bool pend = false, pend_rev = false; uint32_t pend_pos = 0, pend_len = 0, pend_u = 0, pend_off = 0;
    uint64_t r_pk = 0; uint32_t r_len = 0, r_out = 0, r_id = 0, r_nch = 0; int r_nk = 0; bool rev = false;
    uint32_t cur_c = 0;
    int ch_idx = -1, nx_idx = -1; uint64_t bcodes = 0, ncodes = 0; uint32_t bvalid = 0, nvalid = 0;
    bool found = false, use_branch = false, iskm = false; uint32_t fin_end = 0, fin_colex = 0;
    bool have_cand = false; uint32_t cand_len = 0, cand_colex = 0;
    uint32_t dflags = 0, res_g = 0, res_idx = 0;
    uint32_t budget = 0;   // epochs this read may still use; a read that runs out is handed to the overflow kernel
    // register caches of index data
    const uint32_t WNONE = n + 64u;   // a window tag no node position can match (n_nodes < 2^32 - 64)
    uint32_t wtag = WNONE, q_wtag = 0; uint64_t wlo = 0, whi = 0;   // node bytes [wtag, wtag+16), inside one block
    uint32_t ctag = NONE, q_ctag = 0; uint64_t cth0 = 0, cth1 = 0;   // thermometer planes of block ctag (NONE while in flight)
    uint32_t dsel = 0, dret = 0; int dlen = 0;                       // byte-window drop in progress: interval (0 = I, 1 = k-mer), new_len, state to return to
    uint32_t rtagA = NONE, rtagB = NONE; uint64_t rplA = 0, rplB = 0; uint32_t rbsA = 0, rbsB = 0;   // tag = block*4 + char
    uint32_t ttag = NONE; uint4 wt = make_uint4(0, 0, 0, 0);   // 64 bases of unitig text, tag = position >> 6
    uint4 aux = make_uint4(0, 0, 0, 0);
    const void* q_aux = nullptr;
    uint32_t q = 0;
    // work queue (wave-uniform): current range [rs_base, rs_base + rs_cnt), prefetched next range, refill in flight
    uint32_t rs_base = 0, rs_cnt = 0, rs_nbase = 0, rs_val = 0;
    bool rs_nhave = false, rs_inflight = false, rs_exhausted = false;

    // window placement: [ws, ws+16) inside the block of `pos`, `below` bytes of room under pos when possible
    auto win_place = [&](uint32_t pos, uint32_t below) -> uint32_t {
        const int bs = (int)(pos & ~63u);
        return (uint32_t)min(max((int)pos - (int)below, bs), bs + 48);
    };
    auto req_win = [&](uint32_t ws) { q_wtag = ws; wtag = WNONE; q |= Q_W; };   // nothing is in the window until it lands
    auto in_win = [&](uint32_t pos) -> bool { return pos - wtag < 16u; };
    auto win_byte = [&](uint32_t pos) -> uint32_t {
        const uint32_t j = pos - wtag;
        return (uint32_t)((j < 8 ? wlo : whi) >> (8 * (j & 7u))) & 0xFFu;
    };
    // (value selects, no conditional stores to different variables: keeps every cache tag in a register)
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
    // update_sbwt_interval on [l, r] with the cached records: 0 = data missing (requested), 1 = ok, 2 = (-1,-1)
    auto extend_try = [&](uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) -> int {
        if (l == 0 && r == n - 1) {
            // masks, not `c == 0 ? C0 : ...`: the compiler folds a select of loads into a load through a selected ADDRESS, which
            // turns the operands into memory (kernarg loads in mid-epoch, or scratch)
            const uint32_t m0 = 0u - (uint32_t)(c == 0), m1 = 0u - (uint32_t)(c == 1), m2 = 0u - (uint32_t)(c == 2), m3 = 0u - (uint32_t)(c == 3);
            nl = (C0 & m0) | (C1 & m1) | (C2 & m2) | (C3 & m3);
            nr = ((C1 & m0) | (C2 & m1) | (C3 & m2) | (C4 & m3)) - 1;
            return nl <= nr ? 1 : 2;
        }
        if (q & (Q_RA | Q_RB)) return 0;   // requested this epoch, not there yet
        const uint32_t tl = ((l >> 6) << 2) | c, tr = ((r >> 6) << 2) | c;
        const bool lA = tl == rtagA, lB = tl == rtagB, rA = tr == rtagA, rB = tr == rtagB;
        if (!((lA || lB) && (rA || rB))) { req_recs(l, r, c); return 0; }
        const uint64_t pl = lA ? rplA : rplB, pr = rA ? rplA : rplB;
        const uint32_t bl = lA ? rbsA : rbsB, br = rA ? rbsA : rbsB;
        const uint32_t ol = l & 63u, orr = r & 63u;
        nl = bl + (uint32_t)__popcll(pl & ~(~0ull << ol));
        const uint32_t re = br + (uint32_t)__popcll(pr & (~0ull >> (63 - orr)));
        nr = re - 1;
        return nl < re ? 1 : 2;
    };
    // One step of drop_first_char (common.hh:38-48) with the window in registers; new_len >= 1.  Progress lives in
    // l, r, dflags (bit 0: lower end final, bit 1: upper end final).  Returns true when both ends are final; otherwise a
    // window has been requested and the caller stays in its state.
    auto drop_step = [&](uint32_t& l, uint32_t& r, int new_len) -> bool {
        // written with selects only (no branches): both SIMD issue ports are the limit of this kernel, and every divergent
        // `if` costs scalar exec-mask bookkeeping
        const bool avail = !(q & Q_W);   // a window requested this epoch is not there yet
        const uint64_t trep = (uint64_t)(uint32_t)new_len * 0x0101010101010101ull;
        // bit per byte: LCS < new_len (a scan stops there)
        const uint32_t lt = (movemask8(~(((wlo & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) |
                             (movemask8(~(((whi & 0x7f7f7f7f7f7f7f7full) | 0x8080808080808080ull) - trep)) << 8));
        // lower end: highest stop at or below l
        const uint32_t jd = l - wtag;
        const bool d_in = avail && jd < 16u;
        const uint32_t md = lt & (0xFFFFu >> (15u - (jd & 15u)));
        const bool d_open = !(dflags & 1u) && l != 0;
        const bool d_move = d_open && d_in;
        const uint32_t l_new = md ? wtag + (31u - (uint32_t)__clz((int)md)) : wtag - 1u;   // nothing stops: continue below the window
        const bool d_done = !d_open || (d_in && md != 0);
        l = d_move ? l_new : l;
        // upper end: lowest stop at or above r+1
        const uint32_t ju = r + 1u - wtag;
        const bool u_in = avail && ju < 16u;
        const uint32_t mu = (lt & (0xFFFFu << (ju & 15u))) & 0xFFFFu;
        const bool u_open = !(dflags & 2u) && r < n - 1u;
        const bool u_move = u_open && u_in;
        uint32_t r_new = mu ? wtag + ((uint32_t)__ffs((int)mu) - 1u) - 1u : wtag + 15u;
        const bool u_clamp = r_new >= n - 1u;
        r_new = u_clamp ? n - 1u : r_new;
        const bool u_done = !u_open || (u_in && (mu != 0 || u_clamp));
        r = u_move ? r_new : r;
        dflags = (d_done ? 1u : 0u) | (u_done ? 2u : 0u);
        const bool done = d_done && u_done;
        // not finished: ask for the window the unfinished scan continues in (the lower one first)
        const bool want = !done && avail;
        const uint32_t ws = !d_done ? win_place(l, 15) : win_place(r + 1u, 0);
        q_wtag = want ? ws : q_wtag;
        wtag = want ? WNONE : wtag;
        q |= want ? (uint32_t)Q_W : 0u;
        return done;
    };
    // drop_first_char from the thermometer planes: exact for new_len in lcs_t0+1 .. lcs_t0+3 when the scan starts in the cached
    // block; an end that cannot be decided here (other threshold, other block, scan leaves the block) stays open for the byte path
    auto drop_coarse = [&](uint32_t& l, uint32_t& r, int new_len) -> bool {
        const int d = new_len - (int)ix.lcs_t0;
        const bool rng = (uint32_t)(d - 1) < 3u;
        const uint64_t lt = ~(d <= 1 ? (cth1 | cth0) : (d == 2 ? cth1 : (cth1 & cth0)));   // bit i: LCS[block*64 + i] < new_len
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
    // hand an unfinished drop to the shared byte-window block (top of the next epoch) and ask for the window it will need
    auto enter_bdrop = [&](uint32_t sel, uint32_t l, uint32_t r, int new_len, uint32_t ret) {
        dsel = sel; dlen = new_len; dret = ret; pc = P_BDROP;
        const uint32_t pos = !(dflags & 1u) ? l : r + 1u;
        if (!in_win(pos) && !(q & Q_W)) req_win(!(dflags & 1u) ? win_place(l, 15) : win_place(r + 1u, 0));
    };
    auto close_run = [&]() {
        if (run_len) { pend = true; pend_rev = rev; pend_pos = run_pos; pend_len = run_len; pend_u = run_u; pend_off = run_off; run_len = 0; }
    };
    auto strand_init = [&]() {
        il = 0; ir = n - 1; kl = 0; kr = n - 1; start = 0; kstart = 0; end = 0; bu_end = -1;
        dq_head = 0; dq_cnt = 0; walk = false; run_len = 0; ch_idx = -1; nx_idx = -1;
    };
    auto chunk_addr = [&](int ci) -> const void* { return (const void*)(packed + r_pk + (rev ? r_nch : 0u) + (uint32_t)ci); };

    // shrink step: one iteration of the `while (freq == 1)` loop (common.hh:146-154) or, when the interval is no longer a
    // single node, the candidate insertion (:155-163).  Called twice per epoch for pc == P_SHRINK.
    auto shrink_block = [&](int rep) {
        if (pc == P_SHRINK) {
            WB(rep == 0 ? B_SHRINK1 : B_SHRINK2);
            if (il != ir) {
                if (have_cand) {
                    WB(B_SHRINK_CAND);
                    const uint64_t cand = dq_pack(cand_len, cand_colex, (uint32_t)end);
                    if (dq_cnt && (dq_front >> 24) > (cand >> 24)) dq_cnt = 0;
                    else if (dq_cnt && (dq_back >> 24) > (cand >> 24)) {
                        // the front is <= cand here, so the pops stop at the front at the latest and every slot read below is live
                        // when its value is used; the two entries under the back are fetched together (one LDS latency, not two)
                        const uint64_t b1 = DQ(dq_head + dq_cnt - 2), b2 = DQ(dq_head + dq_cnt - 3);
                        WB(B_SHRINK_POPBACK);
                        dq_cnt--; dq_back = b1;
                        if ((b1 >> 24) > (cand >> 24)) {
                            dq_cnt--; dq_back = b2;
                            while ((dq_back >> 24) > (cand >> 24)) { WB(B_SHRINK_POPBACK); dq_cnt--; dq_back = DQ(dq_head + dq_cnt - 1); }
                        }
                    }
                    if (dq_cnt >= dq_limit) {   // more live candidates than LDS slots: the overflow kernel redoes this read
                        fin_ovf_push(ix, ovf_list, ovf_count, r_id);
                        run_len = 0; pc = P_READ0;
                    } else {
                        DQ(dq_head + dq_cnt) = cand;
                        if (dq_cnt == 0) dq_front = cand;
                        dq_back = cand; dq_cnt++;
                        pc = P_KMER;
                    }
                } else pc = P_KMER;
            } else {
                WB(B_SHRINK_DROP);
                have_cand = true; cand_len = (uint32_t)(end - start + 1); cand_colex = il;
                start++;
                const int nlen = end - start + 1;
                if (nlen <= 0) { il = 0; ir = n - 1; }
                else { dflags = 0; if (!drop_coarse(il, ir, nlen)) { WB(B_SHRINK_BDROP); enter_bdrop(0, il, ir, nlen, P_SHRINK); } }
            }
        }
    };
    // one attempt of the finimizer-interval extend and, on failure, one step of its recovery (common.hh:114-126)
    auto exti_block = [&](int rep) {
        if (pc == P_EXTI) {
            WB(rep == 0 ? B_EXTI1 : B_EXTI2);
            uint32_t nl, nr;
            const int rc = extend_try(cur_c, il, ir, nl, nr);
            if (rc == 1) { il = nl; ir = nr; pc = P_EXTK; }
            else if (rc == 2) {
                WB(B_EXTI_FAIL);
                kstart = ++start;
                if (start > end) { il = 0; ir = n - 1; pc = P_EXTK; }
                else if (end - start <= 0) { il = 0; ir = n - 1; }
                else { dflags = 0; if (!drop_coarse(il, ir, end - start)) { WB(B_EXTI_BDROP); enter_bdrop(0, il, ir, end - start, P_EXTI); } }
            }
        }
    };

    for (;;) {
        // ================= 1. serve this epoch's requests: all loads issue back to back, one wait =================
        // (issue order = order of first use in the body below: the waits are counter-based and loads return in order, so what is
        // needed last -- the rank records, at the extend blocks -- is issued last and is still in flight while the head runs)
        if (q & Q_AUX) aux = load16u(q_aux);
        if (q & Q_W) { wtag = q_wtag; const uint4 v = load16u(blk_base + (size_t)(wtag >> 6) * 128 + (wtag & 63u)); wlo = v.x | ((uint64_t)v.y << 32); whi = v.z | ((uint64_t)v.w << 32); }
        if (q & Q_C) { ctag = q_ctag; const uint4 v = *(const uint4*)(blk_base + (size_t)ctag * 128 + 112); cth0 = v.x | ((uint64_t)v.y << 32); cth1 = v.z | ((uint64_t)v.w << 32); }
        if (q & Q_RA) { const FinCharRec v = *(const FinCharRec*)(blk_base + (size_t)(rtagA >> 2) * 128 + 64 + 12 * (rtagA & 3u)); rplA = v.plane_lo | ((uint64_t)v.plane_hi << 32); rbsA = v.base; }
        if (q & Q_RB) { const FinCharRec v = *(const FinCharRec*)(blk_base + (size_t)(rtagB >> 2) * 128 + 64 + 12 * (rtagB & 3u)); rplB = v.plane_lo | ((uint64_t)v.plane_hi << 32); rbsB = v.base; }
        const bool got_nextchunk = (q & Q_NEXTCHUNK) != 0;
        q = 0;

        // force the wait for this epoch's loads here so that it is charged to T_SERVE
#ifdef FIN_STATS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TSTAMP(T_SERVE);
        // ================= 2. guarded blocks, in the order a base flows through them =================
        if (pc == P_STRAND_END) {
            STAT(ST_STRAND); WB(B_STRAND_END);
            close_run();
            if (rev) { rev = false; strand_init(); pc = P_BASE; }
            else pc = P_READ0;
        }
        if (pc == P_READ1) {   // descriptor arrived
            STAT(ST_READ); WB(B_READ1);
            r_pk = aux.x | ((uint64_t)aux.y << 32); r_len = aux.z; r_out = aux.w;
            r_nk = (int)r_len - k + 1; r_nch = (r_len + 31u) >> 5;
            budget = r_len > 0x3FFFF00u ? 0xFFFFFFFFu : ix.budget_mult * r_len + ix.budget_add;   // a healthy read needs about 3 epochs per base (both strands)
            if (r_nk <= 0) pc = P_READ0;
            else { rev = strands == 1; strand_init(); pc = P_BASE; }
        }
        // ---- the shared byte-window step of drop_first_char (thresholds or blocks the thermometer planes do not cover) ----
        if (pc == P_BDROP) {
            WB(B_BDROP);
            uint32_t l = dsel ? kl : il, r = dsel ? kr : ir;
            const bool done = drop_step(l, r, dlen);
            il = dsel ? il : l; ir = dsel ? ir : r; kl = dsel ? l : kl; kr = dsel ? r : kr;
            if (done) pc = dret; else STAT(ST_WIN_SHRINK);
        }
        if (pc == P_CHUNKWAIT) { STAT(ST_CHUNK); WB(B_CHUNKWAIT); bcodes = aux.x | ((uint64_t)aux.y << 32); bvalid = aux.z; ch_idx = end >> 5; pc = P_BASE; }

        TSTAMP(T_HEAD);
        // The blocks that only need the arrival window come first (Ustart probe, the k-mer interval's drop); the shrink loop,
        // whose scans may replace the window, comes after them.  Same results as the reference order (:145-182): the probe
        // and the drop do not depend on the candidate insertion, and `found` is read after it.
        // ---- Ustart probe (common.hh:167) ----
        if (pc == P_USTART) {
            WB(B_USTART);
            if (kl == kr) {
                WB(B_USTART_PROBE);
                if (in_win(kl)) {
                    if (win_byte(kl) & FIN_USTART_BIT) { bu_end = end; bu_colex = kl; }
                    pc = P_KMER_DROP0;
                } else { STAT(ST_WIN_USTART); if (!(q & Q_W)) req_win(win_place(kl, 6)); }
            } else pc = P_KMER_DROP0;
        }
        // ---- k-mer present: advance kmer_start and drop the first char of the k-mer interval (common.hh:180-181) ----
        if (pc == P_KMER_DROP0) {
            WB(B_KDROP);
            pc = P_SHRINK;
            if (iskm) {
                WB(B_KDROP_ISKM);
                kstart++;
                const int nlen = end - kstart + 1;
                if (nlen <= 0) { kl = 0; kr = n - 1; }
                else {
                    // the interval of a present k-mer is one node p; it only grows if a neighbour shares its (k-1)-suffix,
                    // i.e. LCS[p] or LCS[p+1] >= new_len: two byte tests settle the usual case without the window-wide scan
                    const bool up = kl + 1 < n;
                    const bool quick = kl == kr && in_win(kl) && (!up || in_win(kl + 1));
                    const bool stay = quick && (int)(win_byte(kl) & FIN_LCS_MASK) < nlen && (!up || (int)(win_byte(kl + 1) & FIN_LCS_MASK) < nlen) && kl != 0;
                    if (!stay) { WB(B_KDROP_SCAN); dflags = 0; if (!drop_coarse(kl, kr, nlen)) enter_bdrop(1, kl, kr, nlen, P_SHRINK); }
                }
            }
        }
        TSTAMP(T_USTART_KDROP);
        // ---- shortest-unique shrink (common.hh:145-164): up to two loop iterations per epoch ----
        shrink_block(0);
        shrink_block(1);
#if FIN_V2_SHRINK_REPS >= 3
        shrink_block(2);
#endif
#if FIN_V2_SHRINK_REPS >= 4
        shrink_block(3);
#endif
        TSTAMP(T_SHRINK);
        // ---- k-mer present: its finimizer is the front of the deque (common.hh:170-179) ----
        if (pc == P_KMER) {
            WB(B_KMER);
            found = false;
            if (iskm && dq_cnt) {
                found = true; fin_end = dq_end(dq_front, (uint32_t)end); fin_colex = dq_colex(dq_front);
                use_branch = bu_end >= (int)fin_end;
            }
            pc = P_OUT;
        }

        TSTAMP(T_KMERREC);
        // ---- resolve + walk (FinimizerIndex.hh:148-183, :47-102) ----
        if (pc == P_TEXTWAIT) { STAT(ST_TEXT); WB(B_TEXTWAIT); wt = aux; pc = P_OUT; }
        if (pc == P_OUT) {
            WB(B_OUT);
            uint32_t npc = P_BASE;
            if (end >= k - 1) {
                bool walk_hit = false, need_text = false;
                if (walk && wg + 1 < w_uend && cur_c < 4) {
                    WB(B_OUT_WALK);
                    const uint32_t g1 = wg + 1;
                    if ((g1 >> 6) != ttag) { need_text = true; ttag = g1 >> 6; q_aux = (const void*)(ix.concat + ((size_t)(g1 >> 6) << 2)); q |= Q_AUX; }
                    else {
                        const uint32_t wsel = (g1 >> 4) & 3u;
                        const uint32_t word = wsel == 0 ? wt.x : wsel == 1 ? wt.y : wsel == 2 ? wt.z : wt.w;
                        walk_hit = ((word >> (2 * (g1 & 15u))) & 3u) == cur_c;
                    }
                }
                if (need_text) npc = P_TEXTWAIT;
                else if (walk_hit) { wg++; run_len++; }
                else if (found) npc = P_RES0;
                else { WB(B_OUT_CLOSE); walk = false; close_run(); }
            }
            if (npc == P_BASE) { end++; if (end == (int)r_len) npc = P_STRAND_END; }
            pc = npc;
        }
        // dictionary lookups: one dependent load per epoch (their states are the largest pc values: one test skips them all)
#if FIN_V2_RESGUARD
        if (pc >= P_RES0)
#endif
        {
        if (pc == P_RES5) {   WB(B_RES5);   // aux = ends_p[res_idx .. res_idx+3]
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
                end++;
                pc = end == (int)r_len ? P_STRAND_END : P_BASE;
            }
        }
        if (pc == P_RES4) { WB(B_RES4); res_idx = aux.x; q_aux = (const void*)(ix.ends + res_idx); q |= Q_AUX; pc = P_RES5; }
        if (pc == P_RES3) {   WB(B_RES3);   // aux.x = global_offsets[rank] (common.hh:71) or the unitig start (common.hh:65)
            res_g = use_branch ? aux.x + (uint32_t)(k - 1) + (uint32_t)(end - bu_end) : aux.x + (uint32_t)end - fin_end;
            const uint32_t gs = res_g - (uint32_t)(k - 1);
            if (gs < ix.total_len) { q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = P_RES4; }
            else {   // unreachable on a consistent index (the reference reads out of bounds here): reported as absent
                walk = false; close_run(); end++;
                pc = end == (int)r_len ? P_STRAND_END : P_BASE;
            }
        }
        if (pc == P_RES1) {   WB(B_RES1);   // aux = the 16 bytes of FinBlockInfo that hold this dictionary's mask and rank
            const uint32_t colex = use_branch ? bu_colex : fin_colex;
            const uint64_t below = ~(~0ull << (colex & 63u));
            // finimizer dictionary: bytes [0,16) = {fmin_rank, mask lo, mask hi, -}; branch dictionary: bytes [8,24) = {-, mask lo, mask hi, ustart_rank}
            const uint64_t mask = aux.y | ((uint64_t)aux.z << 32);
            const uint32_t rank = (use_branch ? aux.w : aux.x) + (uint32_t)__popcll(mask & below);
            q_aux = use_branch ? (const void*)(ix.ends + rank) : (const void*)(ix.goff + rank);
            q |= Q_AUX; pc = P_RES3;
        }
        if (pc >= P_RES0 && pc <= P_RES5) STAT(ST_RES);
        if (pc == P_RES0) {
            WB(B_RES0);
            const uint32_t colex = use_branch ? bu_colex : fin_colex;
            q_aux = (const void*)((const char*)(ix.blkinfo + (colex >> 6)) + (use_branch ? 8 : 0)); q |= Q_AUX; pc = P_RES1;
        }
        }

        TSTAMP(T_OUT_RES);
        // ---- next base ----
        if (got_nextchunk) { ncodes = aux.x | ((uint64_t)aux.y << 32); nvalid = aux.z; }   // the prefetched chunk (nothing re-used aux since)
        if (pc == P_BASE) {
            WB(B_BASE);
            const int ci = end >> 5;
            if (ci != ch_idx) {
                WB(B_BASE_CHUNK);
                if (nx_idx == ci) { bcodes = ncodes; bvalid = nvalid; ch_idx = ci; nx_idx = -1; }
                else { q_aux = chunk_addr(ci); q |= Q_AUX; pc = P_CHUNKWAIT; }
            }
            if (pc == P_BASE) {
                const uint32_t j = (uint32_t)end & 31u;
                if ((bvalid >> j) & 1u) { cur_c = (uint32_t)(bcodes >> (2 * j)) & 3u; pc = P_EXTI; }
                else {
                    // non-ACGT base: defined behaviour (reference: UB) = matches nothing, the state the reference's own
                    // `start > end` reset produces (common.hh:118-122)
                    cur_c = 4; start = end + 1; kstart = end + 1; il = 0; ir = n - 1; kl = 0; kr = n - 1; dq_cnt = 0;
                    found = false; pc = P_OUT;
                }
            }
        }
        TSTAMP(T_BASE);
        // ---- (1) finimizer interval (common.hh:114-127): up to three attempts per epoch ----
        exti_block(0);
        exti_block(1);
#if FIN_V2_EXTI_REPS >= 3
        exti_block(2);
#endif
        TSTAMP(T_EXTI);
        if (pc == P_EXTI) { if (q & (Q_RA | Q_RB)) STAT(ST_REC_I); else STAT(ST_EXTI4); }
        if (pc == P_BDROP) STAT(ST_WIN_EXTI);
        // ---- (2) k-mer interval (common.hh:132-143) ----
        if (pc == P_EXTK) {
            WB(B_EXTK);
            if (start == kstart) { kl = il; kr = ir; pc = P_ARRIVE; }
            else {
                WB(B_EXTK_EXT);
                uint32_t nl, nr;
                const int rc = extend_try(cur_c, kl, kr, nl, nr);
                if (rc == 1) { kl = nl; kr = nr; pc = P_ARRIVE; }
                else if (rc == 2) {
                    WB(B_EXTK_FAIL);
                    // the reference advances kmer_start one base at a time, re-deriving the interval each time; while the
                    // interval is the single node p it cannot change before new_len <= max(LCS[p], LCS[p+1]), and the
                    // extend keeps failing on the same node, so jump there (needs the two LCS bytes in the window)
                    int nks = kstart + 1;
                    bool can = true;
                    // (a window never spans two blocks: with p the last node of its block use the plain one-base step)
                    if (kl == kr && !(kl + 1 < n && (kl & 63u) == 63u)) {
                        const bool up = kl + 1 < n;
                        if (in_win(kl) && (!up || in_win(kl + 1))) {
                            const uint32_t m = max(win_byte(kl) & FIN_LCS_MASK, up ? (win_byte(kl + 1) & FIN_LCS_MASK) : 0u);
                            nks = max(nks, end - (int)m);
                            nks = min(nks, start);
                        } else { can = false; STAT(ST_WIN_JUMP); if (!(q & Q_W)) req_win(win_place(kl, 6)); }
                    }
                    if (can) {
                        kstart = nks;
                        if (start == kstart) { kl = il; kr = ir; pc = P_ARRIVE; }   // the usual end of a sequencing error: the k-mer interval rejoins I
                        else if (end - kstart <= 0) { kl = 0; kr = n - 1; }
                        else { dflags = 0; if (!drop_coarse(kl, kr, end - kstart)) { WB(B_EXTK_BDROP); enter_bdrop(1, kl, kr, end - kstart, P_EXTK); } }
                    }
                }
            }
        }
#if FIN_V2_EXTK2
        if (pc == P_EXTK && q == 0) {   // one more attempt right away (typical: after the jump the extend succeeds)
            if (start == kstart) { kl = il; kr = ir; pc = P_ARRIVE; }
            else {
                uint32_t nl, nr;
                const int rc = extend_try(cur_c, kl, kr, nl, nr);
                if (rc == 1) { kl = nl; kr = nr; pc = P_ARRIVE; }
                else if (rc == 2 && kl != kr) {
                    kstart++;
                    if (start != kstart) {
                        if (end - kstart <= 0) { kl = 0; kr = n - 1; }
                        else { dflags = 0; if (!drop_coarse(kl, kr, end - kstart)) enter_bdrop(1, kl, kr, end - kstart, P_EXTK); }
                    }
                }
            }
        }
#endif
        TSTAMP(T_EXTK);
        if (pc == P_EXTK) { if (q & (Q_RA | Q_RB)) STAT(ST_REC_K); else if (!(q & Q_W)) STAT(ST_EXTK_AGAIN); }
        // ---- arrival at the new interval: ask for everything the rest of this base and the next extend need ----
        if (pc == P_ARRIVE) {
            STAT(ST_ARRIVE); WB(B_ARRIVE);
            pc = P_USTART;
            have_cand = false;
            iskm = end - kstart + 1 == k;
            // drop candidates that start before the k-mer window (eager form of the pop_front loop, common.hh:173-176)
            auto stale = [&](uint64_t e) -> bool { return (int)dq_end(e, (uint32_t)end) - (int)dq_len(e) + 1 < kstart; };
            if (dq_cnt && stale(dq_front)) {
                // the two entries behind the front are fetched together (one LDS latency); a slot's value is only used while live
                const uint64_t f1 = DQ(dq_head + 1), f2 = DQ(dq_head + 2);
                WB(B_ARRIVE_POP);
                dq_head++; dq_cnt--; dq_front = f1;
                if (dq_cnt && stale(f1)) {
                    dq_head++; dq_cnt--; dq_front = f2;
                    while (dq_cnt && stale(dq_front)) { WB(B_ARRIVE_POP); dq_head++; dq_cnt--; dq_front = DQ(dq_head); }
                }
            }
            if (!(il == 0 && ir == n - 1)) {
                // the LCS bytes around the interval serve the Ustart probe and the k-mer drop's two-byte test, both only for a
                // single-node k-mer interval; other lanes ask for a window when a scan needs one
                const uint32_t ws = win_place(il, FIN_V2_BELOW);
                if (FIN_V2_WINALWAYS || kl == kr) { if (ws != wtag) req_win(ws); }
                if ((il >> 6) != ctag) { q_ctag = il >> 6; ctag = NONE; q |= Q_C; }
                const int e1 = end + 1;
                if (e1 < (int)r_len) {
                    const int ci = e1 >> 5; const uint32_t j = (uint32_t)e1 & 31u;
                    uint32_t cn = 4;
                    if (ci == ch_idx) { if ((bvalid >> j) & 1u) cn = (uint32_t)(bcodes >> (2 * j)) & 3u; }
                    else if (ci == nx_idx) { if ((nvalid >> j) & 1u) cn = (uint32_t)(ncodes >> (2 * j)) & 3u; }
                    if (cn < 4) req_recs(il, ir, cn);
                }
            }
            if (nx_idx < 0 && ch_idx >= 0 && (uint32_t)(ch_idx + 1) < r_nch) {
                nx_idx = ch_idx + 1; q_aux = chunk_addr(nx_idx); q |= Q_AUX | Q_NEXTCHUNK;
            }
        }

        TSTAMP(T_ARRIVE);
        if (pc == P_SHRINK) STAT(ST_SHRINK4);
        if (pc != P_DONE) { STAT(ST_EPOCH); WB(B_EPOCH); }
        // exit condition every lane reaches: a read that exceeds its epoch budget is redone by the (loop-free) overflow kernel
        if (pc > P_READ1) {
            if (budget == 0) {   // (its requests are dropped: no cache tag may claim data that never arrives)
                fin_ovf_push(ix, ovf_list, ovf_count, r_id); run_len = 0; pend = false;
                if (q & Q_RA) rtagA = NONE;
                if (q & Q_RB) rtagB = NONE;
                q = 0; pc = P_READ0;
            }
            else budget--;
        }

        // ================= 3. cooperative write-out of finished runs (wave-wide, converged) =================
        {
            uint64_t m = __ballot(pend);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                WB(B_WRITEOUT);
                // (ds_bpermute via __shfl measured faster here than v_readlane with a scalar lane index: 134 vs 142 ms)
                const uint32_t o_base = __shfl(r_out, src), o_nk = (uint32_t)__shfl(r_nk, src);
                const uint32_t p_pos = __shfl(pend_pos, src), p_len = __shfl(pend_len, src);
                const uint32_t p_u = __shfl(pend_u, src), p_off = __shfl(pend_off, src);
                const bool p_rev = __shfl((int)pend_rev, src) != 0;
                for (uint32_t i = lane; i < p_len; i += 64) {
                    const uint32_t idx = p_rev ? (o_nk - 1 - (p_pos + i)) : (p_pos + i);
                    out[(size_t)o_base + idx] = make_int2((int)p_u, (int)(p_off + i));
                }
            }
            pend = false;
        }
        // ================= 4. work queue =================
        // Reads come from a global counter in ranges of 64 per wave.  The returning atomic is issued one epoch before its value
        // is needed (its latency hides behind that epoch's loads): the wave holds a current range and a prefetched next one.
        {
            if (rs_inflight) { rs_nbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)rs_val); rs_nhave = true; rs_inflight = false; }
            const bool need = pc == P_READ0;
            const uint64_t m = __ballot(need);
            if (m) {
                WB(B_QUEUE);
                const uint32_t cnt = (uint32_t)__popcll(m), rk = (uint32_t)__popcll(m & ((1ull << lane) - 1));
                if (rs_cnt == 0 && rs_nhave) { rs_base = rs_nbase; rs_cnt = 64; rs_nhave = false; }
                const uint32_t take1 = min(cnt, rs_cnt);
                uint32_t id = rs_base + rk; bool got = rk < take1;
                rs_base += take1; rs_cnt -= take1;
                const uint32_t rest = cnt - take1;
                if (rest && rs_nhave) {
                    rs_base = rs_nbase; rs_cnt = 64; rs_nhave = false;
                    if (!got) { id = rs_base + (rk - take1); got = true; }
                    rs_base += rest; rs_cnt -= rest;
                }
                if (need && (got || rs_exhausted)) {
                    r_id = id;
                    if (got && id < n_reads) { q_aux = (const void*)(desc + id); q |= Q_AUX; pc = P_READ1; }
                    else pc = P_DONE;
                }
            }
            if (rs_base >= n_reads && (rs_cnt || rs_exhausted)) { rs_exhausted = true; rs_cnt = 0; }
            if (rs_nhave && rs_nbase >= n_reads) { rs_exhausted = true; rs_nhave = false; }
            if (!rs_nhave && !rs_inflight && !rs_exhausted) {
                if (lane == 0) rs_val = atomicAdd(work_counter, 64u);
                rs_inflight = true;
            }
        }
        TSTAMP(T_TAIL);
        if (!__any(pc != P_DONE)) break;
    }
#ifdef FIN_BLOCKS
    for (int i = 0; i < B_N; i++) { atomicAdd(&stats[2 * i], (unsigned long long)bl[i]); if (bw[i]) atomicAdd(&stats[2 * i + 1], (unsigned long long)bw[i]); }
#endif
#ifdef FIN_STATS
    for (int i = 0; i < ST_N; i++) atomicAdd(&stats[i], (unsigned long long)st[i]);
    if (lane == 0) for (int i = 0; i < T_N; i++) atomicAdd(&stats[ST_N + i], (unsigned long long)tacc[i]);
#endif
#undef DQ
}

extern "C" int fin_launch_search_v2(const FinDevIndex* ix, const uint8_t* bases, const void* packed, const FinReadDesc* desc,
                                    const uint64_t* offs, const uint64_t* out_offs, void* out, uint64_t n_kmers, uint32_t n_reads,
                                    int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                                    uint32_t* work_counter, uint64_t* ovf_scratch, uint32_t ovf_blocks, uint32_t grid_blocks,
                                    hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (n_reads == 0) return 0;
    hipError_t e = hipMemsetAsync(ovf_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(out, 0xFF, n_kmers * 8, stream);   // every slot (-1,-1); runs overwrite
    if (e != hipSuccess) return (int)e;
    const uint32_t need = (n_reads + FIN_TPB - 1) / FIN_TPB;
    const uint32_t grid = grid_blocks < need ? grid_blocks : need;
    if (ev0) (void)hipEventRecord(ev0, stream);
#ifdef FIN_BLOCKS
    static unsigned long long* d_bstats = nullptr;
    if (!d_bstats) { (void)hipMalloc((void**)&d_bstats, 2 * B_N * 8); }
    (void)hipMemsetAsync(d_bstats, 0, 2 * B_N * 8, stream);
    hipLaunchKernelGGL(fin_search_v2_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, n_reads,
                       strands, lds_deque_limit, ovf_list, ovf_count, work_counter, d_bstats);
    {
        unsigned long long h[2 * B_N];
        (void)hipMemcpy(h, d_bstats, 2 * B_N * 8, hipMemcpyDeviceToHost);
        static const char* names[B_N] = {"strand_end", "read1", "bdrop", "chunkwait", "ustart", "ustart_probe", "kdrop", "kdrop_iskm", "kdrop_scan", "shrink1", "shrink2",
                                         "shrink_cand", "shrink_popback", "shrink_drop", "shrink_bdrop", "kmer", "textwait", "out", "out_walk", "out_close", "res5", "res4", "res3", "res1", "res0",
                                         "base", "base_chunk", "exti1", "exti2", "exti_fail", "exti_bdrop", "extk", "extk_ext", "extk_fail", "extk_bdrop", "arrive", "arrive_pop", "writeout", "queue", "epoch"};
        const double we = (double)h[2 * B_EPOCH + 1];
        fprintf(stderr, "[fin_blocks] %-16s %14s %14s %8s %8s\n", "block", "lane_execs", "wave_execs", "lanes/ex", "ex/epoch");
        for (int i = 0; i < B_N; i++)
            fprintf(stderr, "[fin_blocks] %-16s %14llu %14llu %8.2f %8.3f\n", names[i], h[2 * i], h[2 * i + 1], h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] : 0.0,
                    we > 0 ? (double)h[2 * i + 1] / we : 0.0);
    }
#elif defined(FIN_STATS)
    static unsigned long long* d_stats = nullptr;
    if (!d_stats) { (void)hipMalloc((void**)&d_stats, (ST_N + T_N) * 8); }
    (void)hipMemsetAsync(d_stats, 0, (ST_N + T_N) * 8, stream);
    hipLaunchKernelGGL(fin_search_v2_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, n_reads,
                       strands, lds_deque_limit, ovf_list, ovf_count, work_counter, d_stats);
    {
        unsigned long long h[ST_N + T_N];
        (void)hipMemcpy(h, d_stats, (ST_N + T_N) * 8, hipMemcpyDeviceToHost);
        static const char* names[ST_N] = {"epoch", "arrive", "rec_i", "rec_k", "win_shrink", "win_kmer", "win_exti", "win_extk", "win_ustart", "win_jump",
                                          "chunk", "text", "res", "shrink4", "exti4", "extk_again", "read", "strand"};
        fprintf(stderr, "[fin_stats]");
        for (int i = 0; i < ST_N; i++) fprintf(stderr, " %s=%llu", names[i], h[i]);
        static const char* tn[T_N] = {"serve+wait", "head", "ustart_kdrop", "shrink", "kmerrec", "out_res", "base", "exti", "extk", "arrive", "tail"};
        unsigned long long tt = 0;
        for (int i = 0; i < T_N; i++) tt += h[ST_N + i];
        fprintf(stderr, "\n[fin_time] wave-cycles share:");
        for (int i = 0; i < T_N; i++) fprintf(stderr, " %s=%.1f%%", tn[i], 100.0 * (double)h[ST_N + i] / (double)(tt ? tt : 1));
        fprintf(stderr, "\n");
    }
#else
    hipLaunchKernelGGL(fin_search_v2_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, n_reads,
                       strands, lds_deque_limit, ovf_list, ovf_count, work_counter);
#endif
    if (ev1) (void)hipEventRecord(ev1, stream);
    return fin_launch_overflow(ix, bases, offs, out_offs, out, strands, ovf_list, ovf_count, ovf_scratch, ovf_blocks, stream);
}

// resident blocks per CU the hardware admits for the tuned kernel (LDS: 32 KiB per block; registers)
extern "C" int fin_v2_blocks_per_cu(void) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fin_search_v2_kernel, FIN_TPB, 0) != hipSuccess || nb < 1) nb = 2;
    return nb;
}
