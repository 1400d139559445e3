// fin_kernel_v3.hip -- the search-fmin kernel with lazy streaming ("v3"): the v2 machine (fin_kernel_v2.hip: epochs, guarded
// blocks, thermometer/byte-window drops, LDS deque, run write-out, work queue -- read its header first) plus three things that
// let a lane skip work the reference does but whose results cannot matter:
//
//  * WALK mode.  After an anchor hit the reference extends the match along the unitig text without consulting the streaming
//    search (walk_in_unitigs, FinimizerIndex.hh:47-102).  v2 kept the streaming state moving underneath, one base per epoch;
//    here the lane compares up to 32 read bases against the text per epoch (XOR of the 2-bit codes, count trailing zeros) and
//    leaves the streaming state frozen where the anchor was found.
//  * Cold restart.  When a walk ends at read position e the streaming state at e is needed again.  Everything the reference
//    reports at positions >= e is a function of the last 2k-1 bases only (CHANGELOG.md 4.6: kmer_start and start are pure
//    functions of the k-window; a deque entry or a branch record older than that is stale by the reference's own pop rule), so
//    the lane either catches up from the frozen state (gap <= 2k) or restarts the streaming search 2k bases before e, silently
//    (no output) up to e.  Never more streaming than v2 does, usually far less.
//  * PROBE mode.  A strand that matches nothing (the other strand of every read, unrelated reads) needs no streaming at
//    all as long as every k-mer can be PROVEN absent: a substring q[p..f] that does not occur in the index rules out every k-mer
//    containing it.  The lane looks up the interval of the T bases q[p..p+T-1] in a prefix table (T = 15 for a 250 Mbp index: 8 GiB,
//    built on the device when the index is uploaded), extends it base by base up to q[p..t0] where t0 is the first unresolved
//    k-mer end and p = t0-(T+4)+1; a failure at f <= t0 resolves every k-mer ending in [f, p+k-1] as absent and the next probe
//    starts k-(T+4)+1 bases further on; a probe that reaches t0 without failing hands over to the streaming search (cold restart
//    before t0).  The streaming search hands back to probing after 2k positions without a present k-mer.
//
//  * Short restarts.  A walk that ends on a base disagreeing with the text almost always hit a sequencing error: the next k k-mers
//    are absent (only their PRESENCE matters, which needs the k-window alone) and the next anchor is k positions on.  The lane
//    restarts T+1 bases before the mismatching base and checks, when it gets there, that kmer_start has moved past the restart
//    point (then kmer_start and start are the true values from there on); else k-1 back; and with the full margin if a k-mer
//    turns out to be present before the state is known exact.
//  * Probe pre-pass.  fin_probe_kernel (below) probes every strand from its start in a light kernel of its own and tells this one
//    where to start each strand, or to skip it.
//  * Jump table, text re-anchoring (round 2; CHANGELOG.md 4.6, 4.8): a (re)started search takes its state after J bases from a table
//    when their interval holds two nodes or more; on a disjoint index the k-mer behind a bad position is found by comparing the read
//    with the unitig text after probes have proven the k-mers across it absent.
// The same body, as ROLE_STREAM, is the stream kernel of kernel 4's pipeline (fin_kernel_w.hip).  Also in this file: the probe pre-pass
// and the kernels that build the prefix / jump tables, the seed table and the absence filter when an index is uploaded.
//
// Results are bit-identical to v2 / the oracle; only the amount of work differs.  Nothing is carried from one launch to the next.
#include "fin_device.h"
#include "fin_kernels.h"
#include <cstdio>
#include <cstring>
#ifdef FIN_V3_STATS
#define MST(i) (mst[(i)]++)
#else
#define MST(i) ((void)0)
#endif
// -DFIN_V3_TIME: wave-cycles per segment of the epoch (s_memtime stamps by lane 0, summed into g_fin_tacc; diagnostic build only)
#ifdef FIN_V3_TIME
enum { T_SERVE = 0, T_READ, T_BDROP, T_USTART, T_KDROP, T_SHRINK, T_PUSH, T_KMER, T_OUT, T_LOOKUP, T_BASE, T_EXTI, T_EXTK, T_ARRIVE, T_BUDGET, T_WRITEOUT, T_QUEUE, T_N };
__device__ unsigned long long g_fin_tacc[2 * T_N];
#define TS(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tacc[(i)] += t_ - tprev; tprev = t_; } while (0)
#else
#define TS(i) ((void)0)
#endif
#ifdef FIN_V3_TRACE
#define TR(...) do { if (r_id == (uint32_t)(FIN_V3_TRACE) && !rev) printf(__VA_ARGS__); } while (0)
#else
#define TR(...) ((void)0)
#endif

namespace {

enum : uint32_t {
    P_DONE = 0, P_READ0, P_READL, P_READ1, P_READ2, P_STRAND_END, P_BDROP, P_JUMP1, P_JUMP0, P_BASE, P_EXTI, P_EXTK,
    P_ARRIVE, P_SHRINK, P_USTART, P_KMER, P_KMER_DROP0, P_OUT, P_WALK, P_PROBE1, P_PROBEX, P_PROBE0, P_REANCH, P_SAFE, P_RES0, P_RES1, P_RES3, P_RES4, P_RES5
};
// Q_AUX: one 16-byte load per lane and epoch; the CUR/NEXT/TEXT flags say which cache it fills (else `aux` is read by the lane's state)
static_assert(FIN_Q_RA == 2u && FIN_Q_RB == 4u, "request flags");
enum : uint32_t { Q_W = 1, Q_RA = FIN_Q_RA, Q_RB = FIN_Q_RB, Q_AUX = FIN_Q_AUX, Q_NEXTCHUNK = FIN_Q_NEXTCHUNK, Q_C = 32, Q_CURCHUNK = FIN_Q_CURCHUNK, Q_TEXT = 128 };
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

#ifndef FIN_V3_BELOW
#define FIN_V3_BELOW 7          // LCS bytes the arrival window keeps below the interval's lower end (16 in all)
#endif
#ifndef FIN_V3_PM_ADD
#define FIN_V3_PM_ADD 4      // probe length = prefix-table depth + this (a random string of that length must almost never occur in the index)
#endif
#ifndef FIN_V3_DELTA_ADD
#define FIN_V3_DELTA_ADD 1   // verified short restart: prefix-table depth + this many bases before the mismatching base
#endif
#ifndef FIN_V3_MINWAVES
#define FIN_V3_MINWAVES 4   // waves per SIMD the register allocator must leave room for
#endif
// ROLE_ALL: the whole search of a read in one lane (reverse strand, forward strand; probes, streaming, lookups, walks, output).
// ROLE_STREAM: the streaming search alone, for the kernel pipeline of fin_kernel_w.hip ("kernel 4"): a lane takes a stream item
// {read|strand, restart position, silent_until, exact_from}, streams until the first k-mer it must report and hands that over as an
// anchor item (or a probe item after a long absent stretch) -- no lookups, no walks, no output, none of their code or registers.
enum : int { ROLE_ALL = 0, ROLE_STREAM = 1 };
struct FinPipeArgs {            // ROLE_STREAM (and the list mode of ROLE_ALL)
    const uint4* items_in;      // stream items
    const uint32_t* n_in;       // their number (device memory: written by the kernel before)
    uint4* items_out;           // anchor / probe items for the walk kernel
    uint32_t* n_out;
    const uint32_t* read_list;  // ROLE_ALL: search these reads only (n_in of them) instead of all n_reads
};
template <int ROLE>
__device__ __forceinline__ void fin_search_body(const FinDevIndex& ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                uint32_t n_reads_arg, int strands, uint32_t dq_limit, uint32_t* ovf_list,
                                                uint32_t* ovf_count, uint32_t* work_counter, const uint32_t* pass, const FinPipeArgs& pa
#ifdef FIN_V3_STATS
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
    // work items: reads (ROLE_ALL; through pa.read_list when given), stream items (ROLE_STREAM); a count in device memory wins
    const uint32_t n_reads = pa.n_in ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*pa.n_in) : n_reads_arg;
    // the C array as scalar values (left as `ix.C[c]` the compiler selects a kernarg OFFSET and issues two dependent global loads
    // in the middle of the epoch)
    const uint32_t C0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[0]), C1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[1]),
                   C2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[2]), C3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[3]),
                   C4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[4]);

    // probing (uniform): table depth T (0: no table), probe length PM = min(T + 4, k), cold-restart margin and hand-back distance 2k
    const int PT = (int)ix.ptab_t;
    const int PM = min(PT + FIN_V3_PM_ADD, k);
    const int MARGIN = 2 * k, LEAVE = 2 * k;
    // verified short restart (walk block): this many bases before a mismatching base; a string ending in a wrong base rarely matches
    // longer than log4(index size) + a few, which is about PT
    const int DELTA = PT > 0 ? min(k - 1, PT + FIN_V3_DELTA_ADD) : k - 1;
    // jump table depth: a (re)start takes the state after its first JT bases from the table (P_JUMP0/1); the table's strings must be
    // shorter than k (no k-mer may end inside them)
    const int JT = (int)ix.jtab_t < k ? (int)ix.jtab_t : 0;

#ifdef FIN_V3_STATS
    uint32_t mst[12] = {0};
#endif
    // ---- per-lane state -------------------------------------------------------------------------------------
    uint32_t pc = P_READ0;
    uint32_t il = 0, ir = 0, kl = 0, kr = 0;
    int start = 0, kstart = 0, end = 0, bu_end = -1;
    uint32_t bu_colex = 0, dq_head = 0, dq_cnt = 0;
    uint64_t dq_front = 0, dq_back = 0;   // register mirrors of DQ(dq_head) and DQ(dq_head + dq_cnt - 1)
    uint32_t wg = 0, w_u = 0, w_ustart = 0, w_uend = 0;   // walk: global text position of the last matched base, its unitig
    int silent_until = 0, last_pres = 0;                   // streaming: no output before this position; last position with a present k-mer
    int& wend = silent_until;                              // walk: next k-mer end position to test (the streaming state stays at `end`);
                                                           // shares a register with silent_until, which a walk always sets to wend when it ends
    uint32_t pass_fwd = 0;                                 // pre-pass result of the forward strand (used when the reverse strand is done)
    int exact_from = 0;                                    // streaming: after an optimistic (short) restart the state is only known exact from here on

    uint32_t run_pos = 0, run_len = 0, run_u = 0, run_off = 0;
    bool pend = false, pend_rev = false; uint32_t pend_pos = 0, pend_len = 0, pend_u = 0, pend_off = 0;
    uint64_t r_pk = 0; uint32_t r_len = 0, r_out = 0, r_id = 0, r_nch = 0; int r_nk = 0; bool rev = false;
    uint32_t cur_c = 0;
    int ch_idx = -1, nx_idx = -1; uint64_t bcodes = 0, ncodes = 0; uint32_t bvalid = 0, nvalid = 0;
    bool found = false, use_branch = false, iskm = false; uint32_t fin_end = 0, fin_colex = 0;
    // PROBE mode keeps its few values in registers of the streaming search, which is dead while a lane probes (every way out of
    // PROBE mode goes through cold_start or ends the strand): first unresolved k-mer end, probe start, next base, the probe string's
    // codes from pp on, offset of its first non-ACGT base.  (The kernel sits at the 128-VGPR limit of 4 waves per SIMD.)
    uint32_t& t0 = fin_end; int& pp = kstart; int& pe = start; uint64_t& pcode = dq_front; uint32_t& pfi = bu_colex;
    // TEXT RE-ANCHORING behind a bad read position (disjoint indexes, see the walk block): the bad position and the text position
    // aligned with it live in the k-mer interval's registers, which are dead until the streaming search is restarted
    uint32_t& br_E = kl; uint32_t& br_tE = kr;
    bool bridging = false;   // PROBE mode is proving the k-mers across br_E absent; P_REANCH follows
    bool have_cand = false; uint32_t cand_len = 0, cand_colex = 0;
    uint32_t dflags = 0, res_g = 0, res_idx = 0;
    uint32_t budget = 0;   // epochs this read may still use; a read that runs out is handed to the overflow kernel
    // register caches of index data
    const uint32_t WNONE = n + 64u;   // a window tag no node position can match (n_nodes < 2^32 - 64)
    uint32_t wtag = WNONE, q_wtag = 0; uint64_t wlo = 0, whi = 0;   // node bytes [wtag, wtag+16), inside one block
    uint32_t ctag = NONE, q_ctag = 0; uint64_t cth0 = 0, cth1 = 0;   // thermometer planes of block ctag (NONE while in flight)
    uint32_t dsel = 0, dret = 0; int dlen = 0;                       // byte-window drop in progress: interval (0 = I, 1 = k-mer), new_len, state to return to
    FinRecCache rc;   // the two cached rank records (tag = block*4 + char)
    uint32_t ttag = NONE; uint4 wt = make_uint4(0, 0, 0, 0);   // 64 bases of unitig text, tag = position >> 6
    uint4 aux = make_uint4(0, 0, 0, 0);
    const void* q_aux = nullptr;
    uint32_t q = 0;
    FinWaveQueue oq;   // ROLE_STREAM: this wave's slots in the queue it hands items to
    uint32_t it_cas = 0;   // ROLE_STREAM: bit 30 of the item's first word
    FinWorkRanges wr; wr.init();   // which read / item a lane takes next

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
    auto req_recs = [&](uint32_t l, uint32_t r, uint32_t c) { rc.request(l, r, c, q); };
    // update_sbwt_interval on [l, r] with the cached records: 0 = data missing (requested), 1 = ok, 2 = (-1,-1)
    auto extend_try = [&](uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) -> int { return rc.extend(c, l, r, n, C0, C1, C2, C3, C4, q, nl, nr); };
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
    auto chunk_addr = [&](int ci) -> const void* { return (const void*)(packed + r_pk + (rev ? r_nch : 0u) + (uint32_t)ci); };
    // (re)start the streaming search at read position c: the state the reference has before its first base, shifted to c
    auto cold_start = [&](int c) {
        il = 0; ir = n - 1; kl = 0; kr = n - 1; start = c; kstart = c; end = c; bu_end = -1;
        dq_head = 0; dq_cnt = 0;
    };
    // the streaming search goes on at `end`; after a cold start the first JT bases may come from the jump table: the state after
    // streaming q[end..end+JT-1] from cold is a function of their SBWT interval alone when that interval holds at least two nodes
    // (every prefix of the string then occurs twice as well: no candidate was pushed, start = kmer_start = the restart position, the
    // k-mer interval equals the finimizer interval, no Ustart record, no k-mer ended) -- provided nothing is reported there
    auto begin_stream = [&](bool cold) { pc = (cold && JT > 0 && end + JT - 1 < silent_until) ? (uint32_t)P_JUMP0 : (uint32_t)P_BASE; };
    // a strand begins by probing for its first k-mer (k-mer end k-1)
    auto strand_init = [&]() {
        cold_start(0); run_len = 0; ch_idx = -1; nx_idx = -1;
        silent_until = 0; last_pres = 0; exact_from = 0; t0 = (uint32_t)(k - 1); bridging = false;
    };
    // make chunk ci the current read chunk; false = it has been requested (or the load slot is taken) and the caller retries
    auto need_chunk = [&](int ci) -> bool {
        // (a tag is set when its load is REQUESTED; the data is there from the next epoch on)
        if (ch_idx == ci) return !(q & Q_CURCHUNK);
        if (nx_idx == ci) { if (q & Q_NEXTCHUNK) return false; bcodes = ncodes; bvalid = nvalid; ch_idx = ci; nx_idx = -1; return true; }
        if (!(q & Q_AUX)) { q_aux = chunk_addr(ci); q |= Q_AUX | Q_CURCHUNK; ch_idx = ci; }
        return false;
    };
    // a probe proved every k-mer ending in [.., pp+k-1] absent
    auto probe_fail = [&]() {
        TR("probe fail t0=%u pp=%d pe=%d\n", t0, pp, pe);
        t0 = (uint32_t)(pp + k);
        if (t0 >= r_len) pc = P_STRAND_END;
        else if (bridging && t0 > br_E + (uint32_t)(k - 1)) { pe = 0; pc = P_REANCH; }   // every k-mer that contains the bad position is proven absent
        else pc = P_PROBE0;
    };
    // q[pp..t0] occurs in the index: the streaming search takes over, restarted far enough back to be exact from t0 on
    auto probe_pass = [&]() {
        TR("probe pass t0=%u pp=%d\n", t0, pp);
        bridging = false;
        cold_start(max(0, (int)t0 - MARGIN));
        silent_until = (int)t0; last_pres = (int)t0; exact_from = 0; begin_stream(true);
    };

    // shrink step: one iteration of the `while (freq == 1)` loop (common.hh:146-154); several copies per epoch
    // While the interval is the single node p, dropping to new_len leaves it {p} as long as new_len > m = max(LCS[p], LCS[p+1]); every
    // such iteration only overwrites the candidate (same colex rank, one base shorter).  So the iterations down to new_len = m are
    // taken in one step when the two LCS bytes are in the arrival window (same reasoning as the k-mer interval's jump):
    // the candidate has length m+1, start = end-m+1, and the one real scan has threshold m -- an LCS value, which is what the
    // thermometer planes are centred on.
    auto shrink_block = [&](int rep) {
        if (pc == P_SHRINK && il == ir) {
            int nlen = end - start;   // the threshold of the plain next iteration
            const bool up = il + 1 < n;
            if (!(up && (il & 63u) == 63u) && in_win(il) && (!up || in_win(il + 1))) {
                const int m = max(il ? (int)(win_byte(il) & FIN_LCS_MASK) : 0, up ? (int)(win_byte(il + 1) & FIN_LCS_MASK) : 0);
                nlen = min(nlen, m);
            }
            have_cand = true; cand_len = (uint32_t)(nlen + 1); cand_colex = il;
            start = end - nlen + 1;
            if (nlen <= 0) { il = 0; ir = n - 1; }
            else { dflags = 0; if (!drop_coarse(il, ir, nlen)) { enter_bdrop(0, il, ir, nlen, P_SHRINK); } }
        }
    };
    // the loop has ended (the interval is no longer a single node): monotone-deque insertion of the last candidate (:155-163).
    // One copy per epoch, after the shrink steps: whichever step ended a lane's loop, the insertion runs once for all of them.
    auto shrink_push = [&]() {
        if (pc == P_SHRINK && il != ir) {
            pc = P_KMER;
            if (have_cand) {
                const uint64_t cand = dq_pack(cand_len, cand_colex, (uint32_t)end);
                if (dq_cnt && (dq_front >> 24) > (cand >> 24)) dq_cnt = 0;
                else if (dq_cnt && (dq_back >> 24) > (cand >> 24)) {
                    // the front is <= cand here, so the pops stop at the front at the latest and every slot read below is live
                    // when its value is used; the two entries under the back are fetched together (one LDS latency, not two)
                    const uint64_t b1 = DQ(dq_head + dq_cnt - 2), b2 = DQ(dq_head + dq_cnt - 3);
                    dq_cnt--; dq_back = b1;
                    if ((b1 >> 24) > (cand >> 24)) {
                        dq_cnt--; dq_back = b2;
                        while ((dq_back >> 24) > (cand >> 24)) { dq_cnt--; dq_back = DQ(dq_head + dq_cnt - 1); }
                    }
                }
                if (dq_cnt >= dq_limit) {   // more live candidates than LDS slots: the overflow kernel redoes this read
                    fin_ovf_push(ix, ovf_list, ovf_count, r_id);
                    run_len = 0; pc = P_READ0;
                } else {
                    DQ(dq_head + dq_cnt) = cand;
                    if (dq_cnt == 0) dq_front = cand;
                    dq_back = cand; dq_cnt++;
                }
            }
        }
    };
    // one attempt of the finimizer-interval extend and, on failure, one step of its recovery (common.hh:114-126)
    auto exti_block = [&](int rep) {
        if (pc == P_EXTI) {
            uint32_t nl, nr;
            const int rc = extend_try(cur_c, il, ir, nl, nr);
            if (rc == 1) { il = nl; ir = nr; pc = P_EXTK; }
            else if (rc == 2) {
                kstart = ++start;
                if (start > end) { il = 0; ir = n - 1; pc = P_EXTK; }
                else if (end - start <= 0) { il = 0; ir = n - 1; }
                else { dflags = 0; if (!drop_coarse(il, ir, end - start)) { enter_bdrop(0, il, ir, end - start, P_EXTI); } }
            }
        }
    };

#ifdef FIN_V3_TIME
    unsigned long long tacc[T_N] = {0}, tprev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#endif
    for (;;) {
        // ================= 1. serve this epoch's requests: all loads issue back to back, one wait =================
        // (issue order = order of first use in the body below: the waits are counter-based and loads return in order, so what is
        // needed last -- the rank records, at the extend blocks -- is issued last and is still in flight while the head runs)
        if (q & Q_AUX) aux = load16u(q_aux);
        if (q & Q_W) { wtag = q_wtag; const uint4 v = load16u(blk_base + (size_t)(wtag >> 6) * 128 + (wtag & 63u)); wlo = v.x | ((uint64_t)v.y << 32); whi = v.z | ((uint64_t)v.w << 32); }
        if (q & Q_C) { ctag = q_ctag; const uint4 v = *(const uint4*)(blk_base + (size_t)ctag * 128 + 112); cth0 = v.x | ((uint64_t)v.y << 32); cth1 = v.z | ((uint64_t)v.w << 32); }
        rc.serve(q, blk_base);
        if (q & Q_NEXTCHUNK) { ncodes = aux.x | ((uint64_t)aux.y << 32); nvalid = aux.z; }
        if (q & Q_CURCHUNK) { bcodes = aux.x | ((uint64_t)aux.y << 32); bvalid = aux.z; }
        if (q & Q_TEXT) wt = aux;
        q = 0;
#ifdef FIN_V3_TIME
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TS(T_SERVE);
        // force the wait for this epoch's loads here so that it is charged to T_SERVE
        // ================= 2. guarded blocks, in the order a base flows through them =================
        // A strand starts either by probing (no pre-pass) or from what the probe pre-pass (fin_probe_kernel) found for it: the first
        // k-mer end it could not prove absent (the streaming search starts there, as after a passed probe), or none at all.
        auto strand_begin = [&](uint32_t first) {
            strand_init();
            if (first == FIN_PASS_DEFERRED) first = (uint32_t)(k - 1);   // (a strand the pipeline deferred and then gave the read up: from its first k-mer)
            if (!pass) pc = P_PROBE0;
            else if (first == NONE) pc = P_STRAND_END;
            else { t0 = first; probe_pass(); }
        };
        if constexpr (ROLE == ROLE_STREAM) {
            if (pc == P_STRAND_END) pc = P_READ0;   // nothing left of this item
            if (pc == P_READ2) {   // descriptor of the item's read arrived: the search starts at the item's restart position
                r_pk = aux.x | ((uint64_t)aux.y << 32); r_len = aux.z; r_out = aux.w;
                r_nk = (int)r_len - k + 1; r_nch = (r_len + 31u) >> 5;
                budget = r_len > 0x3FFFF00u ? 0xFFFFFFFFu : ix.budget_mult * r_len + ix.budget_add;
                const int c = kstart;   // (held there since the item arrived)
                run_len = 0; ch_idx = -1; nx_idx = -1;
                cold_start(c); last_pres = silent_until; begin_stream(true);
            }
            if (pc == P_READ1) {   // stream item arrived: {read | strand << 31, restart position, silent_until, exact_from}
                r_id = aux.x & 0x3FFFFFFFu; rev = (aux.x >> 31) != 0u; it_cas = aux.x & 0x40000000u;   // (bit 30 travels with the item: see fin_route_kernel)
                kstart = (int)aux.y; silent_until = (int)aux.z; exact_from = (int)aux.w;
                budget = 0xFFFFFFFFu;
                if (aux.x == FIN_Q_EMPTY) pc = P_READ0;   // a slot its producer reserved and did not use
                else { q_aux = (const void*)(desc + r_id); q |= Q_AUX; pc = P_READ2; }
            }
        } else {
        if (pc == P_STRAND_END) {
            close_run();
            if (rev) { rev = false; strand_begin(pass_fwd); }
            else pc = P_READ0;
        }
        if (pc == P_READ2) {   // pre-pass results of this read: {forward, reverse}
            pass_fwd = aux.x;
            if (strands == 1 && aux.y != NONE) { rev = true; strand_begin(aux.y); }   // (a reverse strand without any k-mer is not even begun; FIN_PASS_DEFERRED != NONE)
            else { rev = false; strand_begin(aux.x); }
        }
        if (pc == P_READ1) {   // descriptor arrived
            r_pk = aux.x | ((uint64_t)aux.y << 32); r_len = aux.z; r_out = aux.w;
            r_nk = (int)r_len - k + 1; r_nch = (r_len + 31u) >> 5;
            budget = r_len > 0x3FFFF00u ? 0xFFFFFFFFu : ix.budget_mult * r_len + ix.budget_add;   // a healthy read needs about 3 epochs per base (both strands)
            if (r_nk <= 0) pc = P_READ0;
            else if (pass) { q_aux = (const void*)(pass + 2 * (size_t)r_id); q |= Q_AUX; pc = P_READ2; }
            else { rev = strands == 1; strand_init(); pc = P_PROBE0; }
        }
        if (pc == P_READL) {   // list mode: the read number arrived
            r_id = aux.x;
            if (aux.x == FIN_Q_EMPTY) pc = P_READ0;   // a slot its producer reserved and did not use
            else { q_aux = (const void*)(desc + r_id); q |= Q_AUX; pc = P_READ1; }
        }
        }
        TS(T_READ);
        // ---- the shared byte-window step of drop_first_char (thresholds or blocks the thermometer planes do not cover) ----
        if (pc == P_BDROP) {
            uint32_t l = dsel ? kl : il, r = dsel ? kr : ir;
            const bool done = drop_step(l, r, dlen);
            il = dsel ? il : l; ir = dsel ? ir : r; kl = dsel ? l : kl; kr = dsel ? r : kr;
            if (done) pc = dret;
        }

        TS(T_BDROP);
        // The blocks that only need the arrival window come first (Ustart probe, the k-mer interval's drop); the shrink loop,
        // whose scans may replace the window, comes after them.  Same results as the reference order (:145-182): the probe
        // and the drop do not depend on the candidate insertion, and `found` is read after it.
        // ---- Ustart probe (common.hh:167) ----
        if (pc == P_USTART) {
            if (kl == kr) {
                if (in_win(kl)) {
                    if (win_byte(kl) & FIN_USTART_BIT) { bu_end = end; bu_colex = kl; }
                    pc = P_KMER_DROP0;
                } else { if (!(q & Q_W)) req_win(win_place(kl, 6)); }
            } else pc = P_KMER_DROP0;
        }
        TS(T_USTART);
        // ---- k-mer present: advance kmer_start and drop the first char of the k-mer interval (common.hh:180-181) ----
        if (pc == P_KMER_DROP0) {
            pc = P_SHRINK;
            if (iskm) {
                kstart++;
                const int nlen = end - kstart + 1;
                if (nlen <= 0) { kl = 0; kr = n - 1; }
                else {
                    // the interval of a present k-mer is one node p; it only grows if a neighbour shares its (k-1)-suffix,
                    // i.e. LCS[p] or LCS[p+1] >= new_len: two byte tests settle the usual case without the window-wide scan
                    const bool up = kl + 1 < n;
                    const bool quick = kl == kr && in_win(kl) && (!up || in_win(kl + 1));
                    const bool stay = quick && (int)(win_byte(kl) & FIN_LCS_MASK) < nlen && (!up || (int)(win_byte(kl + 1) & FIN_LCS_MASK) < nlen) && kl != 0;
                    if (!stay) { dflags = 0; if (!drop_coarse(kl, kr, nlen)) enter_bdrop(1, kl, kr, nlen, P_SHRINK); }
                }
            }
        }
        TS(T_KDROP);
        // ---- shortest-unique shrink (common.hh:145-164): one loop iteration per epoch (more were measured: no gain), then the insertion ----
        shrink_block(0);
        TS(T_SHRINK);
        shrink_push();
        TS(T_PUSH);
        // ---- k-mer present: its finimizer is the front of the deque (common.hh:170-179) ----
        if (pc == P_KMER) {
            found = false;
            if (iskm) last_pres = end;
            // (exact_from < 0: a verified short restart whose check is still due; it comes due at the first position that is not
            //  silent, and no k-mer can be present before that: the restarted search is at most DELTA < k bases long there)
            const bool check_due = exact_from < 0 && end == silent_until;
            if (check_due) exact_from = -exact_from;
            if (check_due && kstart <= end - DELTA) {
                // verified short restart: the k-mer interval's string still reaches back to the restart point, so its true start may lie
                // before it -- nothing after it is known exactly; redo from k-1 bases back (presence exact by the k-window alone)
                const int e0 = end;
                cold_start(e0 - (k - 1)); silent_until = e0; exact_from = e0 + k; MST(9);
                begin_stream(true);
            } else
            if (iskm && end >= silent_until && end < exact_from) {
                // a k-mer is present where only its presence is known exactly (optimistic restart, see the walk block): redo with the
                // full margin, silently up to this position
                const int e0 = end;
                cold_start(max(0, e0 - MARGIN)); silent_until = e0; exact_from = 0; MST(11);
                begin_stream(true);
            } else
            if (iskm && dq_cnt && end >= silent_until) {
                found = true; fin_end = dq_end(dq_front, (uint32_t)end); fin_colex = dq_colex(dq_front);
                use_branch = bu_end >= (int)fin_end;
            }
            if (pc == P_KMER) pc = P_OUT;
        }

        // ---- resolve (FinimizerIndex.hh:148-183).  No walk is armed while the streaming search runs (an anchor hands over to
        //      WALK mode, a walk that ends comes back here with the walk disarmed), so a k-mer is either found -> dictionary
        //      lookups, or absent -> the prefilled (-1,-1) stands.  Positions before silent_until only rebuild state. ----
        TS(T_KMER);
        bool emit = false; uint4 emit_item = make_uint4(0, 0, 0, 0);   // ROLE_STREAM: what this lane hands to the walk kernel
        if (pc == P_OUT) {
            uint32_t npc = P_BASE;
            TR("out end=%d found=%d silent_until=%d iskm=%d\n", end, (int)found, silent_until, (int)iskm);
            if constexpr (ROLE == ROLE_STREAM) {
                const uint32_t who = r_id | (rev ? 0x80000000u : 0u) | it_cas;
                if (found) {   // anchor item: the dictionary to look in, the node, and how far the k-mer's end lies behind the record's position
                    emit = true;
                    emit_item = make_uint4(who, (uint32_t)end, use_branch ? bu_colex : fin_colex,
                                           (uint32_t)(use_branch ? end - bu_end : end - (int)fin_end) | (use_branch ? 0x80000000u : 0u));
                    npc = P_READ0;
                } else if (end - last_pres >= LEAVE && end >= silent_until) {   // probe item: prove the rest absent, or find where to go on
                    if ((uint32_t)end + 1u < r_len) { emit = true; emit_item = make_uint4(who, (uint32_t)end + 1u, NONE, 0u); }
                    npc = P_READ0;
                }
            } else {
            if (found) npc = P_RES0;
            else if (end - last_pres >= LEAVE && end >= silent_until) {   // a long stretch without any k-mer: back to probing
                TR("leave end=%d last_pres=%d\n", end, last_pres);
                t0 = (uint32_t)end + 1u;
                npc = t0 < r_len ? (uint32_t)P_PROBE0 : (uint32_t)P_STRAND_END;
            }
            }
            if (npc == P_BASE) { end++; if (end == (int)r_len) npc = P_STRAND_END; }
            pc = npc;
        }
        if constexpr (ROLE == ROLE_STREAM) fin_wq_push(oq, emit, emit_item, pa.items_out, pa.n_out, lane);   // hand-over
        TS(T_OUT);
        if constexpr (ROLE == ROLE_ALL) {
        // dictionary lookups: one dependent load per epoch (their states are the largest pc values: one test skips them all)
        if (pc >= P_RES0)
        {
        if (pc == P_RES5) {     // aux = ends_p[res_idx .. res_idx+3]
            const uint32_t gs = res_g - (uint32_t)(k - 1);
            bool done = true;
            if (gs < aux.y) { w_u = res_idx; w_ustart = aux.x; w_uend = aux.y; }
            else if (gs < aux.z) { w_u = res_idx + 1; w_ustart = aux.y; w_uend = aux.z; }
            else if (gs < aux.w) { w_u = res_idx + 2; w_ustart = aux.z; w_uend = aux.w; }
            else { res_idx += 3; q_aux = (const void*)(ix.ends + res_idx); q |= Q_AUX; done = false; }
            if (done) {
                close_run();
                run_pos = (uint32_t)(end - (k - 1)); run_len = 1; run_u = w_u; run_off = gs - w_ustart;
                wg = res_g;
                TR("anchor end=%d u=%u off=%u g=%u uend=%u\n", end, w_u, run_off, res_g, w_uend);
                end++; wend = end;   // the streaming state is complete through the anchor's position and stays there
                pc = end == (int)r_len ? P_STRAND_END : P_WALK;
                // the walk's first step compares against the text right after the anchor: ask for it now (this lookup's load slot is free)
                if (pc == P_WALK && ((res_g + 1u) >> 6) != ttag && res_g + 1u < w_uend && !(q & Q_AUX)) {
                    ttag = (res_g + 1u) >> 6; q_aux = (const void*)(ix.concat + ((size_t)ttag << 2)); q |= Q_AUX | Q_TEXT;
                }
            }
        }
        if (pc == P_RES4) { res_idx = aux.x; q_aux = (const void*)(ix.ends + res_idx); q |= Q_AUX; pc = P_RES5; }
        if (pc == P_RES3) {     // aux.x = global_offsets[rank] (common.hh:71) or the unitig start (common.hh:65)
            res_g = use_branch ? aux.x + (uint32_t)(k - 1) + (uint32_t)(end - bu_end) : aux.x + (uint32_t)end - fin_end;
            const uint32_t gs = res_g - (uint32_t)(k - 1);
            if (gs < ix.total_len) { q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = P_RES4; }
            else {   // unreachable on a consistent index (the reference reads out of bounds here): reported as absent
                end++;
                pc = end == (int)r_len ? P_STRAND_END : P_BASE;
            }
        }
        if (pc == P_RES1) {     // aux = the 16 bytes of FinBlockInfo that hold this dictionary's mask and rank
            const uint32_t colex = use_branch ? bu_colex : fin_colex;
            const uint64_t below = ~(~0ull << (colex & 63u));
            // finimizer dictionary: bytes [0,16) = {fmin_rank, mask lo, mask hi, -}; branch dictionary: bytes [8,24) = {-, mask lo, mask hi, ustart_rank}
            const uint64_t mask = aux.y | ((uint64_t)aux.z << 32);
            const uint32_t rank = (use_branch ? aux.w : aux.x) + (uint32_t)__popcll(mask & below);
            q_aux = use_branch ? (const void*)(ix.ends + rank) : (const void*)(ix.goff + rank);
            q |= Q_AUX; pc = P_RES3;
        }
        if (pc == P_RES0) {
            const uint32_t colex = use_branch ? bu_colex : fin_colex;
            q_aux = (const void*)((const char*)(ix.blkinfo + (colex >> 6)) + (use_branch ? 8 : 0)); q |= Q_AUX; pc = P_RES1;
        }
        }

        // ---- WALK mode: the match runs on along the unitig text (walk_in_unitigs, FinimizerIndex.hh:47-102), up to 32 bases per epoch ----
        if (pc == P_WALK) {
            const uint32_t g1 = wg + 1u;
            const uint32_t lim_u = w_uend - g1;   // text left in this unitig
            bool brk = g1 >= w_uend;             // (>: an anchor whose k-mer ends beyond its unitig, FinimizerIndex.hh:51-53)
            bool at_uend = brk;
            if (!brk) {
                bool ready = need_chunk(wend >> 5) && !(q & Q_TEXT);   // (text asked for earlier in this epoch, at the anchor: not there yet)
                if (ready && (g1 >> 6) != ttag) {
                    ready = false;
                    if (!(q & Q_AUX)) { ttag = g1 >> 6; q_aux = (const void*)(ix.concat + ((size_t)(g1 >> 6) << 2)); q |= Q_AUX | Q_TEXT; }
                }
                if (ready) {
                    const uint32_t j = (uint32_t)wend & 31u, t = g1 & 63u;
                    const uint64_t rb = bcodes >> (2 * j);
                    const uint32_t inv = ~(bvalid >> j) | (j ? 0xFFFFFFFFu << (32 - j) : 0u);
                    const uint64_t lo = wt.x | ((uint64_t)wt.y << 32), hi = wt.z | ((uint64_t)wt.w << 32);
                    const uint64_t tb = t < 32 ? ((lo >> (2 * t)) | (t ? hi << (64 - 2 * t) : 0ull)) : (hi >> (2 * (t - 32)));
                    const uint32_t tav = t < 32 ? 32u : 64u - t;
                    const uint32_t nmax = min(min(32u - j, tav), min(lim_u, r_len - (uint32_t)wend));
                    const uint64_t x = rb ^ tb;
                    const uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
                    const uint32_t mm = y ? (uint32_t)(__ffsll((long long)y) - 1) >> 1 : 32u;
                    const uint32_t fi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
                    const uint32_t nadv = min(min(mm, fi), nmax);
                    TR("walk wend=%d j=%u t=%u nmax=%u mm=%u fi=%u lim_u=%u nadv=%u run_len=%u\n", wend, j, t, nmax, mm, fi, lim_u, nadv, run_len);
                    run_len += nadv; wg += nadv; wend += (int)nadv;
                    if (wend == (int)r_len) pc = P_STRAND_END;
                    else { brk = nadv < nmax || nadv == lim_u; at_uend = nadv == lim_u; }   // mismatch / non-ACGT base / end of the unitig; else a chunk or text boundary: go on
                }
            }
            if (brk) {
                // the walk ends before position wend: the normal path applies there (FinimizerIndex.hh:148-183), which needs the
                // streaming state at wend -- caught up from where it was left, or restarted MARGIN bases back (see header)
                TR("walk break wend=%d end=%d run_pos=%u run_len=%u\n", wend, end, run_pos, run_len);
                close_run();
                last_pres = wend - 1;
                exact_from = 0;
                MST(7);
                bool cold = true;
                if (!at_uend && ix.text_anchors) {
                    // TEXT RE-ANCHORING.  A k-mer found by comparing the read with the text is reported there if that is the place the
                    // reference reports for it (ix.safe; every place of a disjoint unitig set is).  The read disagrees with the text at E = wend:
                    // the k-mers that contain E (ends E .. E+k-1) are proven absent by probes across E, then q[E+1..E+k] is compared
                    // with the text behind the disagreeing base (P_REANCH) -- no streaming search, no dictionary lookup.  The frozen
                    // streaming state is given up (its k-mer interval registers hold E and the text position aligned with it).
                    br_E = (uint32_t)wend; br_tE = wg + 1u; t0 = (uint32_t)wend; bridging = true; end = -(1 << 29);
                    pc = P_PROBE0;
                } else {
                if (wend - end > DELTA && !at_uend && DELTA < k - 1) {
                    // Verified short restart: DELTA bases back only.  kmer_start and start of a search started at c are max(c, true value),
                    // and both only move forward; if at position wend (the mismatching base, where matches are short) kmer_start has
                    // moved past c, it and start are the true values from there on, and so is everything computed from them: presence
                    // from wend on, candidates and branch records from wend+1 on, i.e. the full state from wend+k on.  The k-mer block
                    // checks this when it gets to wend (marked by a negative exact_from) and falls back to the k-1 restart otherwise.
                    cold_start(wend - DELTA);
                    exact_from = -(wend + k); MST(8);
                } else if (wend - end > k - 1 && !at_uend) {
                    // Optimistic restart.  A read base that disagrees with the text is nearly always a sequencing error, so the k
                    // k-mers containing it are absent and the next anchor is k positions on.  Restarting k-1 bases back makes k-mer
                    // PRESENCE exact from wend on (it only needs the k-window), which is all an absent position needs; everything
                    // else is exact from wend+k on (2k-1 bases after the restart, 4.6 of CHANGELOG.md).  Should a k-mer be present
                    // before that, the k-mer block falls back to the full margin.
                    cold_start(wend - (k - 1));
                    exact_from = wend + k; MST(10);
                } else if (wend - end > MARGIN) cold_start(wend - MARGIN);
                else cold = false;   // the frozen state is close enough: catch up from it
                silent_until = wend;
                begin_stream(cold);
                }
            }
        }
        // ---- PROBE mode (see header): prefix-table entry arrived ----
        if (pc == P_PROBE1) {
            if (aux.x > aux.y) probe_fail();
            else {
                il = aux.x; ir = aux.y; pe = pp + PT;
                if (pe > pp + PM - 1) probe_pass();   // (the probe string is q[pp .. pp+PM-1]; it ends at t0 unless a bad position pulled it back)
                else {
                    pc = P_PROBEX;
                    const uint32_t off = (uint32_t)(pe - pp);
                    if (off < pfi) req_recs(il, ir, (uint32_t)(pcode >> (2 * off)) & 3u);
                }
            }
        }
        // ---- one more base of the probe string ----
        if (pc == P_PROBEX) {
            const uint32_t off = (uint32_t)(pe - pp);
            if (off >= pfi) probe_fail();   // a non-ACGT base: no k-mer contains it
            else {
                uint32_t nl, nr;
                const int rc = extend_try((uint32_t)(pcode >> (2 * off)) & 3u, il, ir, nl, nr);
                if (rc == 2) probe_fail();
                else if (rc == 1) {
                    il = nl; ir = nr; pe++;
                    if (pe > pp + PM - 1) probe_pass();
                    else if (off + 1 < pfi) req_recs(il, ir, (uint32_t)(pcode >> (2 * (off + 1))) & 3u);
                }
            }
        }
        // ---- start a probe for the first unresolved k-mer end t0: the string q[t0-PM+1 .. t0]; across a bad position E the string is
        //      pulled back so that it contains E (q[E .. E+PM-1] at most): a string with a wrong base in it almost never occurs ----
        if (pc == P_PROBE0) {
            int p = (int)t0 - PM + 1;
            if (bridging && p > (int)br_E) p = (int)br_E;
            const int ci0 = p >> 5, ci1 = (p + PM - 1) >> 5;
            bool ready = need_chunk(ci0);
            if (ready && ci1 != ci0 && nx_idx != ci1) {
                ready = false;
                if (!(q & Q_AUX)) { q_aux = chunk_addr(ci1); q |= Q_AUX | Q_NEXTCHUNK; nx_idx = ci1; }
            }
            if (ready) {
                const uint32_t j = (uint32_t)p & 31u;
                uint64_t w = bcodes >> (2 * j); uint32_t v = bvalid >> j;
                if (ci1 != ci0) { w |= ncodes << (64 - 2 * j); v |= nvalid << (32 - j); }   // (j > 0 here: PM <= 32)
                const uint32_t inv = ~v;
                pfi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
                pcode = w; pp = p;
                if (PT > 0) {
                    if (pfi < (uint32_t)PT) probe_fail();
                    else {
                        const uint32_t key = (uint32_t)w & ((1u << (2 * PT)) - 1u);
                        q_aux = (const void*)(ix.ptab + key); q |= Q_AUX; pc = P_PROBE1;
                    }
                } else { il = 0; ir = n - 1; pe = p; pc = P_PROBEX; }
            }
        }
        // ---- text re-anchoring: is q[E+1..E+k] the text behind the bad position?  up to 32 bases per epoch, pe = bases found equal ----
        if (pc == P_REANCH) {
            const int E = (int)br_E;   // (t0 = E + k < r_len here: the k-mer lies inside the read)
            if (br_tE + (uint32_t)k >= w_uend) probe_pass();   // the unitig ends inside that k-mer: the streaming search decides from t0 = E+k on
            else {
                const int rp = E + 1 + pe;
                const uint32_t tp = br_tE + 1u + (uint32_t)pe;
                bool ready = need_chunk(rp >> 5) && !(q & Q_TEXT);
                if (ready && (tp >> 6) != ttag) {
                    ready = false;
                    if (!(q & Q_AUX)) { ttag = tp >> 6; q_aux = (const void*)(ix.concat + ((size_t)(tp >> 6) << 2)); q |= Q_AUX | Q_TEXT; }
                }
                if (ready) {
                    const uint32_t j = (uint32_t)rp & 31u, t = tp & 63u;
                    const uint64_t rb = bcodes >> (2 * j);
                    const uint32_t inv = ~(bvalid >> j) | (j ? 0xFFFFFFFFu << (32 - j) : 0u);
                    const uint64_t lo = wt.x | ((uint64_t)wt.y << 32), hi = wt.z | ((uint64_t)wt.w << 32);
                    const uint64_t tb = t < 32 ? ((lo >> (2 * t)) | (t ? hi << (64 - 2 * t) : 0ull)) : (hi >> (2 * (t - 32)));
                    const uint32_t tav = t < 32 ? 32u : 64u - t;
                    const uint32_t nmax = min(min(32u - j, tav), (uint32_t)(k - pe));
                    const uint64_t x = rb ^ tb;
                    const uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
                    const uint32_t mm = y ? (uint32_t)(__ffsll((long long)y) - 1) >> 1 : 32u;
                    const uint32_t fi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
                    const uint32_t nadv = min(min(mm, fi), nmax);
                    pe += (int)nadv;
                    if (nadv < nmax) {
                        // the next bad position: the k-mers ending in [E+k, E2+k-1] all contain it (E2 <= E+k) -- prove them absent next
                        br_E = (uint32_t)rp + nadv; br_tE = tp + nadv;
                        pc = P_PROBE0;
                    } else if (pe == k) {
                        // present, and in the text here.  Is this the place the reference reports for it?  (an index with duplicated
                        // k-mers: the bit of this text position, next epoch)
                        if (ix.safe && !(q & Q_AUX)) { q_aux = (const void*)(ix.safe + ((br_tE + (uint32_t)k) >> 6)); q |= Q_AUX; pc = P_SAFE; }
                        else if (!ix.safe) pc = P_SAFE;
                    }
                }
            }
        }
        if (pc == P_SAFE && !(q & Q_AUX)) {
            const uint32_t tp = br_tE + (uint32_t)k;
            const uint64_t word = aux.x | ((uint64_t)aux.y << 32);
            if (!ix.safe || ((word >> (tp & 63u)) & 1ull)) {
                // the run starts with this k-mer and the walk goes on behind it
                const int E = (int)br_E;
                run_pos = (uint32_t)(E + 1); run_len = 1; run_u = w_u; run_off = br_tE + 1u - w_ustart;
                wg = br_tE + (uint32_t)k; wend = E + k + 1; bridging = false;
                pc = wend == (int)r_len ? (uint32_t)P_STRAND_END : (uint32_t)P_WALK;
            } else probe_pass();   // reported elsewhere (or from a walk): the streaming search decides from t0 = E+k on
        }
        }   // ROLE_ALL: lookups, walk, probes
        TS(T_LOOKUP);
        // ---- jump start of a (re)started streaming search (see begin_stream) ----
        if (pc == P_JUMP1) {   // table entry arrived: at least two nodes -> that is the state after JT bases
            if (aux.y > aux.x) { il = aux.x; ir = aux.y; kl = aux.x; kr = aux.y; end += JT; }
            pc = P_BASE;
        }
        if (pc == P_JUMP0) {
            const int p = end;
            const int ci0 = p >> 5, ci1 = (p + JT - 1) >> 5;
            bool ready = need_chunk(ci0);
            if (ready && ci1 != ci0 && nx_idx != ci1) {
                ready = false;
                if (!(q & Q_AUX)) { q_aux = chunk_addr(ci1); q |= Q_AUX | Q_NEXTCHUNK; nx_idx = ci1; }
            }
            if (ready && !(q & (Q_AUX | Q_NEXTCHUNK))) {
                const uint32_t j = (uint32_t)p & 31u;
                uint64_t w = bcodes >> (2 * j); uint32_t v = bvalid >> j;
                if (ci1 != ci0) { w |= ncodes << (64 - 2 * j); v |= nvalid << (32 - j); }   // (j > 0 here)
                const uint32_t all = (1u << JT) - 1u;
                if ((v & all) != all) pc = P_BASE;   // a non-ACGT base among them: plain cold start
                else { q_aux = (const void*)(ix.jtab + (uint32_t)(w & ((1ull << (2 * JT)) - 1ull))); q |= Q_AUX; pc = P_JUMP1; }
            }
        }
        // ---- next base ----
        if (pc == P_BASE) {
            if (need_chunk(end >> 5)) {
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
        TS(T_BASE);
        // ---- (1) finimizer interval (common.hh:114-127): one attempt per epoch ----
        exti_block(0);
        TS(T_EXTI);
        // ---- (2) k-mer interval (common.hh:132-143) ----
        if (pc == P_EXTK) {
            if (start == kstart) { kl = il; kr = ir; pc = P_ARRIVE; }
            else {
                uint32_t nl, nr;
                const int rc = extend_try(cur_c, kl, kr, nl, nr);
                if (rc == 1) { kl = nl; kr = nr; pc = P_ARRIVE; }
                else if (rc == 2) {
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
                        } else { can = false; if (!(q & Q_W)) req_win(win_place(kl, 6)); }
                    }
                    if (can) {
                        kstart = nks;
                        if (start == kstart) { kl = il; kr = ir; pc = P_ARRIVE; }   // the usual end of a sequencing error: the k-mer interval rejoins I
                        else if (end - kstart <= 0) { kl = 0; kr = n - 1; }
                        else { dflags = 0; if (!drop_coarse(kl, kr, end - kstart)) { enter_bdrop(1, kl, kr, end - kstart, P_EXTK); } }
                    }
                }
            }
        }
        TS(T_EXTK);
        // ---- arrival at the new interval: ask for everything the rest of this base and the next extend need ----
        if (pc == P_ARRIVE) {
            pc = P_USTART;
            have_cand = false;
            iskm = end - kstart + 1 == k;
            // drop candidates that start before the k-mer window (eager form of the pop_front loop, common.hh:173-176)
            auto stale = [&](uint64_t e) -> bool { return (int)dq_end(e, (uint32_t)end) - (int)dq_len(e) + 1 < kstart; };
            if (dq_cnt && stale(dq_front)) {
                // the two entries behind the front are fetched together (one LDS latency); a slot's value is only used while live
                const uint64_t f1 = DQ(dq_head + 1), f2 = DQ(dq_head + 2);

                dq_head++; dq_cnt--; dq_front = f1;
                if (dq_cnt && stale(f1)) {
                    dq_head++; dq_cnt--; dq_front = f2;
                    while (dq_cnt && stale(dq_front)) { dq_head++; dq_cnt--; dq_front = DQ(dq_head); }
                }
            }
            if (!(il == 0 && ir == n - 1)) {
                // the LCS bytes around the interval serve the Ustart probe and the k-mer drop's two-byte test, both only for a
                // single-node k-mer interval; other lanes ask for a window when a scan needs one
                const uint32_t ws = win_place(il, FIN_V3_BELOW);
                if (kl == kr) { if (ws != wtag) req_win(ws); }   // (only lanes whose k-mer interval is a single node ask for the LCS window)
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
            if (nx_idx < 0 && ch_idx >= 0 && (uint32_t)(ch_idx + 1) < r_nch && !(q & Q_AUX)) {
                nx_idx = ch_idx + 1; q_aux = chunk_addr(nx_idx); q |= Q_AUX | Q_NEXTCHUNK;
            }
        }

#ifdef FIN_V3_STATS
        {   // lane-epochs by mode (the state a lane ends the epoch in)
            const uint32_t cls = pc == P_DONE ? 7u : pc <= P_STRAND_END ? 0u : (pc >= P_RES0 ? 4u : (pc == P_WALK ? 3u : (pc >= P_PROBE1 && pc <= P_PROBE0 ? 2u : (end < silent_until ? 1u : 5u))));
            if (cls < 7) mst[cls]++;
            if (pc == P_BDROP) mst[6]++;
        }
#endif
        TS(T_ARRIVE);
        // exit condition every lane reaches: a read that exceeds its epoch budget is redone by the (loop-free) overflow kernel
        if (pc > P_READ1) {
            if (budget == 0) {   // (its requests are dropped: no cache tag may claim data that never arrives)
                fin_ovf_push(ix, ovf_list, ovf_count, r_id); run_len = 0; pend = false;
                if (q & Q_TEXT) ttag = NONE;
                if (q & Q_W) wtag = WNONE;
                if (q & Q_C) ctag = NONE;
                rc.drop(q);
                q = 0; pc = P_READ0;
            }
            else budget--;
        }

        TS(T_BUDGET);
        // ================= 3. cooperative write-out of finished runs (wave-wide, converged) =================
        if constexpr (ROLE == ROLE_ALL) {
            uint64_t m = __ballot(pend);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;

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
        TS(T_WRITEOUT);
        // ================= 4. work queue (FinWorkRanges) =================
        {
            uint32_t id = 0;
            const int wk = wr.take(pc == P_READ0, lane, n_reads, work_counter, id);
            if (wk) r_id = id;
            if (wk == 1) {
                if constexpr (ROLE == ROLE_STREAM) { q_aux = (const void*)(pa.items_in + id); pc = P_READ1; }
                else if (pa.read_list) { q_aux = (const void*)(pa.read_list + id); pc = P_READL; }
                else { q_aux = (const void*)(desc + id); pc = P_READ1; }
                q |= Q_AUX;
            } else if (wk == 2) pc = P_DONE;
        }
        TS(T_QUEUE);
        if (!__any(pc != P_DONE)) break;
    }
#ifdef FIN_V3_TIME
    if (lane == 0) for (int i = 0; i < T_N; i++) atomicAdd(&g_fin_tacc[(ROLE == ROLE_STREAM ? T_N : 0) + i], tacc[i]);
#endif
    if constexpr (ROLE == ROLE_STREAM) fin_wq_flush(oq, make_uint4(FIN_Q_EMPTY, FIN_Q_EMPTY, FIN_Q_EMPTY, FIN_Q_EMPTY), pa.items_out, lane);
#ifdef FIN_V3_STATS
    if (stats) for (int i = 0; i < 12; i++) atomicAdd(&stats[i], (unsigned long long)mst[i]);
#endif
#undef DQ
}

__global__ __launch_bounds__(FIN_TPB, FIN_V3_MINWAVES) void fin_search_v3_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                                 uint32_t n_reads, int strands, uint32_t dq_limit, uint32_t* ovf_list,
                                                                 uint32_t* ovf_count, uint32_t* work_counter, const uint32_t* pass
#ifdef FIN_V3_STATS
                                                                 , unsigned long long* stats
#endif
                                                                 ) {
    const FinPipeArgs pa{nullptr, nullptr, nullptr, nullptr, nullptr};
    fin_search_body<ROLE_ALL>(ix, packed, desc, out, n_reads, strands, dq_limit, ovf_list, ovf_count, work_counter, pass, pa
#ifdef FIN_V3_STATS
                              , stats
#endif
                              );
}
// the same over a list of reads whose length sits in device memory (what kernel 4's pipeline leaves to kernel 3)
__global__ __launch_bounds__(FIN_TPB, FIN_V3_MINWAVES) void fin_search_v3_list_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                                      int strands, uint32_t dq_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                                                                      uint32_t* work_counter, const uint32_t* pass, const uint32_t* read_list,
                                                                      const uint32_t* n_list) {
    const FinPipeArgs pa{nullptr, n_list, nullptr, nullptr, read_list};
#ifdef FIN_V3_STATS
    fin_search_body<ROLE_ALL>(ix, packed, desc, out, 0u, strands, dq_limit, ovf_list, ovf_count, work_counter, pass, pa, nullptr);
#else
    fin_search_body<ROLE_ALL>(ix, packed, desc, out, 0u, strands, dq_limit, ovf_list, ovf_count, work_counter, pass, pa);
#endif
}

#ifndef FIN_STREAM_MINWAVES
#define FIN_STREAM_MINWAVES 4   // (round 3: at 5 waves -- 96 registers -- the body spilled 5 registers; chr1 without seeds 25.7 ms per step at 4, 26.8 at 5)
#endif
// kernel 4's streaming stage (ROLE_STREAM of the body above)
__global__ __launch_bounds__(FIN_TPB, FIN_STREAM_MINWAVES) void fin_stream_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, uint32_t dq_limit,
                                                                uint32_t* ovf_list, uint32_t* ovf_count, uint32_t* work_counter,
                                                                const uint4* items_in, const uint32_t* n_in, uint4* items_out, uint32_t* n_out) {
    FinPipeArgs pa{items_in, n_in, items_out, n_out, nullptr};
#ifdef FIN_V3_STATS
    fin_search_body<ROLE_STREAM>(ix, packed, desc, nullptr, 0u, 0, dq_limit, ovf_list, ovf_count, work_counter, nullptr, pa, nullptr);
#else
    fin_search_body<ROLE_STREAM>(ix, packed, desc, nullptr, 0u, 0, dq_limit, ovf_list, ovf_count, work_counter, nullptr, pa);
#endif
}

// ---- probe pre-pass: PROBE mode of the kernel above, alone, one strand per work item ------------------------------------------
// For every strand: the first k-mer end t0 that a probe could not prove absent (pass[2*read + strand] = t0; the search kernel starts
// its streaming search there), or NONE if every k-mer of the strand is proven absent (the search kernel skips the strand).  Same
// proofs as PROBE mode (see the header); doing them here keeps the non-matching strands -- half of all strands -- out of the big
// kernel's waves, whose every epoch pays for the streaming blocks whether a lane needs them or not.  ~50 VGPRs, 8 waves per SIMD.
__global__ __launch_bounds__(FIN_TPB) void fin_probe_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, uint32_t n_reads, int strands,
                                                            uint32_t* pass, uint32_t* seed, uint32_t* work_counter) {
    enum : uint32_t { Z_DONE = 0, Z_READ0, Z_READ1, Z_PROBE1, Z_PROBEX, Z_PROBE0, Z_FILT0, Z_FILT1 };
    constexpr uint32_t Q_F2 = 256;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    const char* const blk_base = (const char*)ix.blocks;
    const uint32_t C0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[0]), C1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[1]),
                   C2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[2]), C3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[3]),
                   C4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[4]);
    const int PT = (int)ix.ptab_t;
    const int PM = min(PT + FIN_V3_PM_ADD, k);
    const uint32_t n_items = strands == 1 ? 2u * n_reads : n_reads;
    // absence filter: before a probe at t0 the two strings of F bases that end at t0 and at t0-1 are looked up in the bit set; one that
    // does not occur rules out every k-mer that contains it -- without a prefix-table line or a node block from HBM
    const int F = ix.filt ? (int)ix.filt_f : 0;
    const uint32_t fmask = F ? (F == 16 ? 0xFFFFFFFFu : (1u << (2 * F)) - 1u) : 0u;
    uint32_t f2 = 0;

    uint32_t pc = Z_READ0, item = 0;
    uint32_t il = 0, ir = 0;
    uint64_t r_pk = 0; uint32_t r_len = 0, r_nch = 0; bool rev = false;
    FinChunkCache ck;
    uint32_t t0 = 0; int pp = 0, pe = 0; uint64_t pcode = 0; uint32_t pfi = 0;
    uint32_t budget = 0;
    FinRecCache rc;
    uint4 aux = make_uint4(0, 0, 0, 0);
    const void* q_aux = nullptr;
    uint32_t q = 0;
    FinWorkRanges wr; wr.init();

    auto req_recs = [&](uint32_t l, uint32_t r, uint32_t c) { rc.request(l, r, c, q); };
    auto extend_try = [&](uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) -> int { return rc.extend(c, l, r, n, C0, C1, C2, C3, C4, q, nl, nr); };
    auto strand_chunks = [&]() -> const uint4* { return packed + r_pk + (rev ? r_nch : 0u); };
    // (seed: when the probe string q[t0-PM+1..t0] matched completely and is the suffix of exactly one node, that node -- the only k-mer
    //  that can end at t0 is its label; the walk kernel looks its place up in ix.pos.  NONE otherwise.)
    auto finish = [&](uint32_t result, uint32_t node) { pass[item] = result; if (seed && result != NONE) seed[item] = node; pc = Z_READ0; };
    auto probe_fail = [&]() { t0 = (uint32_t)(pp + k); if (t0 < r_len) pc = F ? (uint32_t)Z_FILT0 : (uint32_t)Z_PROBE0; else finish(NONE, NONE); };

    for (;;) {
        if (q & Q_AUX) aux = load16u(q_aux);
        rc.serve(q, blk_base);
        ck.serve(q, aux, strand_chunks);
        if (q & Q_F2) f2 = ix.filt[(uint32_t)(pcode >> 32) >> 5];
        q = 0;

        if (pc == Z_READ1) {
            r_pk = aux.x | ((uint64_t)aux.y << 32); r_len = aux.z;
            r_nch = (r_len + 31u) >> 5;
            ck.reset();
            budget = r_len > 0x3FFFF00u ? 0xFFFFFFFFu : (ix.budget_mult >> 1) * r_len + ix.budget_add;
            if ((int)r_len < k) finish(NONE, NONE);
            else { t0 = (uint32_t)(k - 1); pc = F ? (uint32_t)Z_FILT0 : (uint32_t)Z_PROBE0; }
        }
        if (pc == Z_FILT1) {   // aux.x / f2: the filter words of the strings that end at t0 / at t0-1 (their keys: pcode low / high)
            const uint32_t key1 = (uint32_t)pcode, key0 = (uint32_t)(pcode >> 32);
            if (!((aux.x >> (key1 & 31u)) & 1u)) t0 += (uint32_t)(k - F + 1);        // q[t0-F+1..t0] occurs nowhere: ends t0 .. t0+k-F are absent
            else if (!((f2 >> (key0 & 31u)) & 1u)) t0 += (uint32_t)(k - F);         // q[t0-F..t0-1] occurs nowhere: ends t0-1 .. t0+k-F-1 are absent
            else pc = Z_PROBE0;                                                       // both occur: the prefix-table probe decides
            if (pc == Z_FILT1) { if (t0 < r_len) pc = Z_FILT0; else finish(NONE, NONE); }
        }
        if (pc == Z_PROBE1) {
            if (aux.x > aux.y) probe_fail();
            else {
                il = aux.x; ir = aux.y; pe = pp + PT;
                if (pe > (int)t0) finish(t0, il == ir ? il : NONE);
                else {
                    pc = Z_PROBEX;
                    const uint32_t off = (uint32_t)(pe - pp);
                    if (off < pfi) req_recs(il, ir, (uint32_t)(pcode >> (2 * off)) & 3u);
                }
            }
        }
        if (pc == Z_PROBEX) {
            const uint32_t off = (uint32_t)(pe - pp);
            if (off >= pfi) probe_fail();
            else {
                uint32_t nl, nr;
                const int rc = extend_try((uint32_t)(pcode >> (2 * off)) & 3u, il, ir, nl, nr);
                if (rc == 2) probe_fail();
                else if (rc == 1) {
                    il = nl; ir = nr; pe++;
                    if (pe > (int)t0) finish(t0, il == ir ? il : NONE);
                    else if (off + 1 < pfi) req_recs(il, ir, (uint32_t)(pcode >> (2 * (off + 1))) & 3u);
                }
            }
        }
        if (pc == Z_FILT0) {
            const int p = (int)t0 - F;   // the F+1 bases q[p..t0]
            const int ci0 = p >> 5, ci1 = (int)t0 >> 5;
            if (ck.need2(ci0, ci1, strand_chunks, q, q_aux)) {
                uint64_t w; uint32_t v;
                ck.window(p, ci0, ci1, w, v);
                const uint32_t inv = ~v;
                const uint32_t fi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
                if (fi <= (uint32_t)F) pc = Z_PROBE0;   // a non-ACGT base among them: the probe below deals with it
                else if (!(q & Q_AUX)) {
                    const uint32_t key0 = (uint32_t)w & fmask, key1 = (uint32_t)(w >> 2) & fmask;
                    pcode = key1 | ((uint64_t)key0 << 32);
                    q_aux = (const void*)(ix.filt + (key1 >> 5)); q |= Q_AUX | Q_F2; pc = Z_FILT1;
                }
            }
        }
        if (pc == Z_PROBE0) {
            const int p = (int)t0 - PM + 1;
            const int ci0 = p >> 5, ci1 = (int)t0 >> 5;
            if (ck.need2(ci0, ci1, strand_chunks, q, q_aux, (int)r_nch)) {   // (a strand's chunks are fetched two at a time: FinChunkCache::need_ahead)
                uint64_t w; uint32_t v;
                ck.window(p, ci0, ci1, w, v);
                const uint32_t inv = ~v;
                pfi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
                pcode = w; pp = p;
                if (PT > 0) {
                    if (pfi < (uint32_t)PT) probe_fail();
                    else {
                        const uint32_t key = (uint32_t)w & ((1u << (2 * PT)) - 1u);
                        q_aux = (const void*)(ix.ptab + key); q |= Q_AUX; pc = Z_PROBE1;
                    }
                } else { il = 0; ir = n - 1; pe = p; pc = Z_PROBEX; }
            }
        }
        // exit condition every lane reaches: a strand that runs out of epochs is handed to the search kernel from its first k-mer
        if (pc > Z_READ1) {
            if (budget == 0) {
                rc.drop(q);
                q = 0; finish((uint32_t)(k - 1), NONE);
            } else budget--;
        }
        {   // work queue (FinWorkRanges)
            uint32_t id = 0;
            const int wk = wr.take(pc == Z_READ0, lane, n_items, work_counter, id);
            if (wk == 1) {
                item = strands == 1 ? id : 2u * id;   // pass[] always has two slots per read: {forward, reverse}
                rev = strands == 1 && (id & 1u);
                q_aux = (const void*)(desc + (strands == 1 ? id >> 1 : id)); q |= Q_AUX; pc = Z_READ1;
            } else if (wk == 2) pc = Z_DONE;
        }
        if (!__any(pc != Z_DONE)) break;
    }
}

// ---- prefix table: the SBWT interval of every string of T bases (update_sbwt_interval T times from the full interval) ----
__global__ __launch_bounds__(FIN_TPB) void fin_build_ptab_kernel(FinDevIndex ix, FinPrefixIval* tab, int T) {
    const uint64_t key = (uint64_t)blockIdx.x * FIN_TPB + threadIdx.x;
    if (key >> (2 * T)) return;
    const char* const blk_base = (const char*)ix.blocks;
    uint32_t l = 0, r = ix.n_nodes - 1;
    bool ok = true;
    for (int i = 0; i < T && ok; i++) {
        const uint32_t c = (uint32_t)(key >> (2 * i)) & 3u;
        const FinCharRec a = *(const FinCharRec*)(blk_base + (size_t)(l >> 6) * 128 + 64 + 12 * c);
        const FinCharRec b = *(const FinCharRec*)(blk_base + (size_t)(r >> 6) * 128 + 64 + 12 * c);
        const uint64_t pa = a.plane_lo | ((uint64_t)a.plane_hi << 32), pb = b.plane_lo | ((uint64_t)b.plane_hi << 32);
        const uint32_t nl = a.base + (uint32_t)__popcll(pa & ~(~0ull << (l & 63u)));
        const uint32_t re = b.base + (uint32_t)__popcll(pb & (~0ull >> (63 - (r & 63u))));
        ok = nl < re;
        l = nl; r = re - 1;
    }
    tab[key] = ok ? FinPrefixIval{l, r} : FinPrefixIval{1u, 0u};
}

extern "C" int fin_launch_build_ptab(const FinDevIndex* ix, void* tab, int T, hipStream_t stream) {
    if (T <= 0) return 0;
    const uint64_t n = 1ull << (2 * T);
    hipLaunchKernelGGL(fin_build_ptab_kernel, dim3((uint32_t)((n + FIN_TPB - 1) / FIN_TPB)), dim3(FIN_TPB), 0, stream, *ix, (FinPrefixIval*)tab, T);
    return (int)hipGetLastError();
}

#define FIN_POS_SEG 256   // text positions per lane of the filter build
// ---- absence filter: a bit for every string of F bases that occurs in a unitig (FinDevIndex::filt) ----
__global__ __launch_bounds__(FIN_TPB) void fin_build_filter_kernel(FinDevIndex ix, uint32_t* filt, int F) {
    const uint64_t s0 = ((uint64_t)blockIdx.x * FIN_TPB + threadIdx.x) * FIN_POS_SEG;
    if (s0 >= ix.total_len) return;
    const uint32_t s1 = (uint32_t)(s0 + FIN_POS_SEG < ix.total_len ? s0 + FIN_POS_SEG : ix.total_len);
    uint32_t u = ix.samp[s0 >> ix.samp_shift];
    while (ix.ends[u + 1] <= (uint32_t)s0) u++;
    uint32_t uend = ix.ends[u + 1];
    uint32_t g = ix.ends[u];
    if (s0 >= (uint32_t)(F - 1) && (uint32_t)s0 - (uint32_t)(F - 1) > g) g = (uint32_t)s0 - (uint32_t)(F - 1);
    uint32_t key = 0, depth = 0;
    for (; g < s1; g++) {
        while (g >= uend) { u++; uend = ix.ends[u + 1]; depth = 0; }
        const uint32_t c = (ix.concat[g >> 4] >> (2 * (g & 15u))) & 3u;
        key = (key >> 2) | (c << (2 * (F - 1)));   // first base of the string in the low bits, as the prefix table's key
        depth++;
        if (depth >= (uint32_t)F && g >= (uint32_t)s0) atomicOr(&filt[key >> 5], 1u << (key & 31u));
    }
}
extern "C" int fin_launch_build_filter(const FinDevIndex* ix, uint32_t* filt, int F, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(filt, 0, ((1ull << (2 * F)) / 32 + 8) * 4, stream);
    if (e != hipSuccess) return (int)e;
    const uint64_t lanes = ((uint64_t)ix->total_len + FIN_POS_SEG - 1) / FIN_POS_SEG;
    if (lanes == 0) return 0;
    hipLaunchKernelGGL(fin_build_filter_kernel, dim3((uint32_t)((lanes + FIN_TPB - 1) / FIN_TPB)), dim3(FIN_TPB), 0, stream, *ix, filt, F);
    return (int)hipGetLastError();
}
extern "C" int fin_launch_search_v3(const FinDevIndex* ix, const uint8_t* bases, const void* packed, const FinReadDesc* desc,
                                    const uint64_t* offs, const uint64_t* out_offs, void* out, uint64_t n_kmers, uint32_t n_reads,
                                    int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                                    uint32_t* work_counter, uint64_t* ovf_scratch, uint32_t ovf_blocks, uint32_t grid_blocks,
                                    uint32_t* pass, uint32_t grid_blocks_probe,
                                    hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev_mid) {
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
    if (pass) {   // probe pre-pass over all strands (its own, much lighter kernel); the search kernel then starts where it says
        const uint64_t items = strands == 1 ? 2ull * n_reads : n_reads;
        const uint32_t need_p = (uint32_t)((items + FIN_TPB - 1) / FIN_TPB);
        hipLaunchKernelGGL(fin_probe_kernel, dim3(grid_blocks_probe < need_p ? grid_blocks_probe : need_p), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc,
                           n_reads, strands, pass, (uint32_t*)nullptr, work_counter);
        e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return (int)e;
        if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    }
#ifdef FIN_V3_STATS
    static unsigned long long* d_stats = nullptr;
    if (!d_stats) (void)hipMalloc((void**)&d_stats, 12 * 8);
    (void)hipMemsetAsync(d_stats, 0, 12 * 8, stream);
    hipLaunchKernelGGL(fin_search_v3_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, n_reads,
                       strands, lds_deque_limit, ovf_list, ovf_count, work_counter, pass, d_stats);
    {
        unsigned long long h[12];
        (void)hipMemcpy(h, d_stats, 12 * 8, hipMemcpyDeviceToHost);
        fprintf(stderr, "[fin_v3_stats] per read: walk breaks %.2f  short restarts %.2f  failed checks %.2f  k-1 restarts %.2f  full-margin fallbacks %.2f\n", (double)h[7] / n_reads, (double)h[8] / n_reads, (double)h[9] / n_reads, (double)h[10] / n_reads, (double)h[11] / n_reads);
        unsigned long long tot = 0;
        for (int i = 0; i < 6; i++) tot += h[i];
        fprintf(stderr, "[fin_v3_stats] lane-epochs %llu (%.1f per read): read/strand %.1f%%  stream silent %.1f%%  stream %.1f%%  probe %.1f%%  walk %.1f%%  lookups %.1f%%  (byte-window drops %.1f%%)\n",
                tot, (double)tot / n_reads, 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[5] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[4] / tot, 100.0 * h[6] / tot);
    }
#else
    hipLaunchKernelGGL(fin_search_v3_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, n_reads,
                       strands, lds_deque_limit, ovf_list, ovf_count, work_counter, pass);
#endif
    if (ev1) (void)hipEventRecord(ev1, stream);
    return fin_launch_overflow(ix, bases, offs, out_offs, out, strands, ovf_list, ovf_count, ovf_scratch, ovf_blocks, stream);
}

// ---- single-stage launchers for the kernel pipeline of fin_kernel_w.hip (kernels are launched from the file that defines them) ----
extern "C" int fin_launch_probe_stage(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t n_reads, int strands, uint32_t* pass,
                                      uint32_t* seed, uint32_t* work_counter, uint32_t grid_blocks, void* fast_out, uint32_t* n_fast, hipStream_t stream) {
    // a merged search whose second strands are deferred (kernel 4 with an anchor table): fin_prepass.hip's plain kernel decides which strand is
    // searched first.  (Where nothing is deferred -- option defer_strand 0, no anchor table -- every read has a strand that takes some nine
    // steps to prove absent: the state machine below, whose lanes take new strands as they finish, was 1 ms faster at that than the plain
    // kernel's stepping loop: chr1_dups before its second strands were deferred, CHANGELOG.md 5.6)
    if (strands == 1 && ix->defer_ok) return fin_launch_pair_prepass(ix, packed, desc, n_reads, pass, seed, 1, grid_blocks, fast_out, n_fast, stream);
    const uint64_t items = strands == 1 ? 2ull * n_reads : n_reads;
    const uint32_t need = (uint32_t)((items + FIN_TPB - 1) / FIN_TPB);
    hipLaunchKernelGGL(fin_probe_kernel, dim3(grid_blocks < need ? grid_blocks : need), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, n_reads, strands, pass, seed, work_counter);
    return (int)hipGetLastError();
}
extern "C" int fin_launch_stream_stage(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t lds_deque_limit, uint32_t* ovf_list,
                                       uint32_t* ovf_count, uint32_t* work_counter, const void* items_in, const uint32_t* n_in, void* items_out,
                                       uint32_t* n_out, uint32_t grid_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(fin_stream_kernel, dim3(grid_blocks), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, lds_deque_limit, ovf_list, ovf_count,
                       work_counter, (const uint4*)items_in, n_in, (uint4*)items_out, n_out);
    return (int)hipGetLastError();
}
// kernel 3 over a list of reads (count in device memory)
extern "C" int fin_launch_v3_list(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, void* out, int strands, uint32_t lds_deque_limit,
                                  uint32_t* ovf_list, uint32_t* ovf_count, uint32_t* work_counter, const uint32_t* pass, const uint32_t* read_list,
                                  const uint32_t* n_list, uint32_t grid_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(fin_search_v3_list_kernel, dim3(grid_blocks), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, strands, lds_deque_limit,
                       ovf_list, ovf_count, work_counter, pass, read_list, n_list);
    return (int)hipGetLastError();
}
// diagnostic (-DFIN_V3_TIME): print and reset the per-segment wave-cycle sums; synchronises the device
extern "C" void fin_debug_dump_time(void) {
#ifdef FIN_V3_TIME
    static const char* tn[T_N] = {"serve+wait", "read/item", "bdrop", "ustart", "kdrop", "shrink", "push", "kmer", "out+emit", "lookup/walk/probe", "base", "exti", "extk", "arrive", "budget", "writeout", "queue"};
    unsigned long long h[2 * T_N];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fin_tacc), sizeof h);
    for (int r = 0; r < 2; r++) {
        unsigned long long tt = 0;
        for (int i = 0; i < T_N; i++) tt += h[r * T_N + i];
        if (!tt) continue;
        fprintf(stderr, "[fin_time %s] wave-cycles share:", r ? "stream kernel" : "kernel 3");
        for (int i = 0; i < T_N; i++) fprintf(stderr, " %s=%.1f%%", tn[i], 100.0 * (double)h[r * T_N + i] / (double)tt);
        fprintf(stderr, "\n");
    }
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_tacc), h, sizeof h);
#endif
}
extern "C" int fin_stream_blocks_per_cu(void) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fin_stream_kernel, FIN_TPB, 0) != hipSuccess || nb < 1) nb = 2;
    return nb;
}

// resident blocks per CU the hardware admits for the tuned kernel (LDS: 32 KiB per block; registers)
extern "C" int fin_probe_blocks_per_cu(void) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fin_probe_kernel, FIN_TPB, 0) != hipSuccess || nb < 1) nb = 4;
    return nb;
}
extern "C" int fin_v3_blocks_per_cu(void) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fin_search_v3_kernel, FIN_TPB, 0) != hipSuccess || nb < 1) nb = 2;
    return nb;
}
