// fin_kernel_w.hip -- "kernel 4": the search-fmin hot path as a PIPELINE of specialised kernels.
//
// Kernel 3 (fin_kernel_v3.hip) gives every lane a whole read: probes, streaming search, dictionary lookups, walks, output.  Its
// wave runs every block of that state machine in every epoch for whichever handful of lanes is in it (17.6 of 64 lanes per vector
// instruction, rocprofv3 round 1): a third of its instructions belong to blocks that a lane is in for 16 % of its epochs.  Here the
// same work -- the same blocks, the same exactness arguments (CHANGELOG.md 4.6) -- is cut where a lane changes its kind of work, and
// the pieces are handed from kernel to kernel through queues in HBM, so that every wave runs ONE kind of work with full lanes:
//
//   fin_pair_prepass_kernel (fin_prepass.hip; merged searches with an anchor table) | fin_probe_kernel (fin_kernel_v3.hip)
//                                           every strand: absence proofs from its start; verdict = first k-mer end not proven absent,
//                                           and the SEED node when the last probe string ends exactly one node.  The pair pre-pass
//                                           also decides which strand of a read is searched first; its sister is DEFERRED (CHANGELOG.md 4.14)
//   fin_route_kernel                        an item for every strand not ruled out: a seed / probe item for the walk kernel (index with a
//                                           seed table), else a stream item.  When both strands of a read are searched the reverse
//                                           strand's pairs only fill slots that still hold (-1,-1): the forward pair wins; when one is,
//                                           its lane writes the read's absent slots too and nothing prefills the output (CHANGELOG.md 4.10)
//   fin_stream_kernel  (fin_kernel_v3.hip, ROLE_STREAM)  stream item {read|strand, restart position, silent_until, exact_from}: the
//                                           streaming search (rarest_fmin_streaming_search, common.hh:78-186) from the restart position
//                                           to the first k-mer it has to report -> anchor item {read|strand, end, node, distance};
//                                           after 2k absent positions -> probe item
//   fin_walk_kernel    (this file)          seed item: the place of the seed's k-mer (seed table), comparison with the unitig text, walk
//                                           (walk_in_unitigs, FinimizerIndex.hh:47-102), runs and absent slots written out; behind a bad
//                                           position: probes across it, then the k-mer behind it compared with the text (re-anchoring);
//                                           at unitig ends and wherever a probe string is not unique: further probes, seeds, look-ups
//                                           of the whole k-mer (CHANGELOG.md 4.8, 4.9) -- nearly everywhere the whole search of a strand.
//                                           When a strand with a deferred sister is done its lane goes on with the sister, inside the
//                                           stretch of slots the strand left open (all of them if what it reported proves nothing
//                                           about the sister: "tainted", CHANGELOG.md 4.14).
//                                           anchor item (from the stream kernel): dictionary lookups (common.hh:61-72,
//                                           PackedStrings.hh:91-100), then the walk; where it ends -> stream item (verified short restart
//                                           T+1 bases back; at a unitig end 2k back).  probe item: absence proofs -> seed, stream item or nothing
//   ... stream / walk alternate FIN_V4_ROUNDS times (a round per sequencing error of the longest-lived reads when anchors come from the
//   streaming search); what is left then goes through kernel 3 in list mode; deque overflows go to the overflow kernel.
//
// No state travels with an item except what is in it: a walk never resumes a frozen streaming search (kernel 3 does when the walk
// was short), it always restarts it -- by the rules kernel 3 uses when the frozen state is too far back, which are exact for any
// distance.  Results are bit-identical to kernels 3 / 2 / 0 and the oracle (the whole GPU suite runs on kernel 4 too).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fin_device.h"
#include "fin_kernels.h"

#ifndef FIN_V4_ROUNDS
#define FIN_V4_ROUNDS 8
#endif
#ifndef FIN_WALK_MINWAVES
#define FIN_WALK_MINWAVES 5   // waves per SIMD the register allocator must leave room for (96 VGPRs)
#endif
#ifndef FIN_V3_PM_ADD
#define FIN_V3_PM_ADD 4      // (as in fin_kernel_v3.hip: probe length = prefix-table depth + this)
#endif
#ifndef FIN_WALK_RANGE
#define FIN_WALK_RANGE 32u   // items per range of the walk kernel's work queue (fin_device.h FinWorkRangesT)
#endif
#ifndef FIN_W_KF_LEAN_EVERY
#define FIN_W_KF_LEAN_EVERY 8   // lean tables: behind a k-mer the k-mer table does not have, every how-many-th end is probed first (a power of two)
#endif
#ifndef FIN_V3_DELTA_ADD
#define FIN_V3_DELTA_ADD 1   // (as in fin_kernel_v3.hip: verified short restart this far + table depth before the mismatching base)
#endif

#ifndef FIN_W_KT2_PAIR
#define FIN_W_KT2_PAIR 1
#endif
#ifndef FIN_W_KT_PAIR
#define FIN_W_KT_PAIR 1
#endif
#ifndef FIN_W_BACKSCAN
#define FIN_W_BACKSCAN 1
#endif
#ifdef FIN_W_DEBUG
__device__ unsigned long long g_fin_wdbg[16];
__device__ unsigned long long g_fin_kfv[80];      // claims the text did not bear out, by the place's offset in its window; [64] first word differs, [65] second, [66] via a rolled key, [67] pp > 0
__device__ unsigned long long g_fin_wstate[40];   // [s]: lane-epochs that began in state s; [32]: wave-epochs; [33]: states present, summed over wave-epochs; [34]: live lanes, summed
__device__ unsigned long long g_fin_witem[96];    // [b]: items (a strand and the deferred sister its lane went on with) that took [2^b, 2^(b+1)) epochs; [40] the longest; [41] epochs summed; [48+b]: waves that ran [2^b, 2^(b+1)) epochs; [88] the longest wave; [90] lane-epochs of deferred strands; [91] sisters gone on with, [92] their stretches' slots, [93] their reads' slots
__device__ unsigned long long g_fin_wtime[8192 + 8];   // [w]: when wave w of the launch with the most work ended, in ticks of the 100 MHz clock since the launch's first wave began ([8192]: that beginning; [8193]: waves recorded)
#define WDBG(i) atomicAdd(&g_fin_wdbg[i], 1ull)
#else
#define WDBG(i) ((void)0)
#endif
namespace {
// The walk kernel's states, in the order of a lane's life (VERDICT r3 asked for a map: the body below is one loop whose blocks are guarded by `pc`).
// An EPOCH = serve the loads the lanes asked for last epoch (aux: one 16-byte load per lane; the read-chunk cache; rank records; the text window), then
// run, top to bottom, the block of every state some lane is in.  A block that moves a lane to a state whose block stands BELOW it hands it on within the
// epoch; one that needs data sets the request bits (q) and the lane waits an epoch.  aux = what the state finds loaded when it runs.
//   W_ITEM0   no item: draw one from the queue (FinWorkRanges)                                   -> W_ITEM1 | W_DONE
//   W_ITEM1   aux = the item {who = read | strand | flags, end, a_colex, a_dl}                   -> W_DESC (asks for the read's descriptor)
//   W_DESC    aux = the descriptor.  By item kind: a PLACE (lean tables: the pre-pass's verified look) -> W_RES4; a seed node -> W_RES3 (pos[node]);
//             a probe item (a_colex = NONE: first unresolved k-mer end) -> W_PROBE0; the stream kernel's anchor -> W_RES1
//   W_RES1    aux = dictionary block (mask + rank, FinBlockInfo)                                  -> W_RES3 (global_offsets / unitig start, common.hh:61-72)
//   W_RES3    aux = the offset, or a seed's anchor-table entry {g, unitig, bounds}: a verified seed -> W_REANCH (compare the k-mer with the text);
//             an anchor -> W_RES4; a dummy node / unusable seed -> W_PROBE0 | W_KF0
//   W_RES4    aux = samp[] (where in ends_p[] the place lies)                                     -> W_RES5
//   W_RES5    aux = four unitig ends: the place's unitig and its bounds (PackedStrings.hh:91-100); the run starts -> W_WALK (or the item ends)
//   W_WALK    read chunk against text window, 32 bases an epoch (walk_in_unitigs): a disagreeing base -> bridging probes W_PROBE0 (then W_REANCH);
//             the unitig's end -> W_PROBE0 at the next k-mer end; the read's end -> run closed, W_ITEM0
//   W_PROBE0  makes the probe string that ends at t0 from the chunk cache (the PM bases; across a bad position: pulled back over it).  Lean tables: asks
//             the directional string filter -> W_PROBEF; else the prefix table -> W_PROBE1.  W_KF0 shares this block: the string is the whole k-mer,
//             asked of the k-mer table -> W_KF1 (two-word keys: the second word first, W_KF0B)
//   W_PROBEF  aux = filter block: the string occurs in no unitig -> every k-mer that holds it is absent, t0 moves on (-> W_PROBE0 | W_REANCH | the item
//             ends); it may occur -> nothing proven: the whole k-mer is looked up (W_KF0)
//   W_PROBE1  aux = prefix-table interval: empty -> absent as above; else W_PROBEX extends it base by base (rank records) to PM bases; a string that
//             ends one node is a SEED -> W_RES3; several nodes -> the whole k-mer (W_KF0, or W_PROBE0 with pfull when there is no k-mer table)
//   W_KF1     aux (+ the text window's register) = a bucket of the k-mer table, four slots {answer, tag}: a tag match CLAIMS the k-mer with its answer g -- the
//             text window(s) at g are asked for and compared with the k-mer's codes (they are in registers) before anything else happens: one window -> with the
//             locate's first load, W_RES4 compares; two -> W_KFV compares, then W_RES4.  Equal: an anchor like any other; not equal (a shared tag): kernel 3.
//             An UNVERIFIED claim (the answer spells another k-mer) -> W_KFX, the exact side table; the chain's first empty slot -> the k-mer is absent, next
//             end (every 8th: a probe first; k >= 40 under lean tables: one back-scan per stretch, W_PROBE0 with fl.bs; two-word keys roll by a base: kf_roll2);
//             a bucket full of other k-mers -> the next
//   W_REANCH  the k-mer behind a bad position (or a seed's k-mer) against the text, 32 bases an epoch: equal -> the run starts there, W_WALK; a base
//             differs -> bridging probes again; an unsafe place -> W_SAFE (bitmap) first
//   every state: an item out of epochs (budget) gives its read to kernel 3; what the streaming search must do is handed on as a stream item (hand_on).
//   When an item ends its strand's open slots are written; a deferred sister strand is then searched by the same lane as a probe item (to_sister).
enum : uint32_t { W_DONE = 0, W_ITEM0, W_ITEM1, W_DESC, W_RES1, W_RES3, W_RES4, W_RES5, W_WALK, W_PROBE1, W_PROBEX, W_PROBE0, W_REANCH, W_SAFE, W_KF0, W_KF1, W_PROBEF, W_KF0B, W_KFX, W_KFV };
static_assert(FIN_Q_RA == 2u && FIN_Q_RB == 4u, "request flags");
enum : uint32_t { Q_RA = FIN_Q_RA, Q_RB = FIN_Q_RB, Q_AUX = FIN_Q_AUX, Q_NEXTCHUNK = FIN_Q_NEXTCHUNK, Q_CURCHUNK = FIN_Q_CURCHUNK, Q_TEXT = 128, Q_AUX2 = 256 };   // Q_AUX2 (with Q_AUX): the k-mer table's NEXT slot too
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t FIN_WHO_GAPS = 0x20000000u;    // first word of an item, bit 29: this lane also writes the (-1,-1) of every slot of its strand that no pair fills
constexpr uint32_t FIN_WHO_DEFER = 0x10000000u;   // ... bit 28: the read's other strand is deferred -- when this one is done, the lane searches it inside the stretch of slots left open
constexpr uint32_t FIN_WHO_READ = 0x0FFFFFFFu;    // ... bits 0..27: the read
constexpr uint32_t FIN_SEED_MARK = 0x7FFFFFFEu;   // fourth word of a seed item (an anchor item has distance | use_branch << 31 there, a distance is below the read length)
constexpr uint32_t FIN_PLACE_MARK = 0x7FFFFFFDu;  // ... of a PLACE item ("lean tables": no anchor table): the third word is not a node but the verified answer g of the k-mer that ends at t0 (a look's k-mer-table slot)

__device__ __forceinline__ uint4 load16u(const void* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }

}  // namespace

// ---- route: an item for every strand the pre-pass could not rule out -------------------------------------------------------------
// A strand whose verdict comes with a seed node (the last probe string matched completely and is the suffix of one node only; the
// index has a seed table) goes straight to the walk kernel as a SEED item {read|strand, t0, node, FIN_SEED_MARK}; every other one
// becomes a stream item.  When both strands of a read are searched, the reverse strand's pairs are written with "only if the slot
// still holds (-1,-1)" (flag bit 30 of the item's first word): the strands need not wait for each other and the forward pair still
// wins the merge (search_fmin.hh:54-60).  A block counts the items of its reads, reserves exactly that many slots of either queue with
// one atomic each, then writes them (a queue's counter takes about 88 atomics per microsecond: one per wave would cost more than the
// kernel's memory traffic).
// probe_items != 0: a strand without a seed goes to the walk kernel too, as a probe item {read|strand, t0, NONE, 0} -- its probes end in
// a look-up of the whole k-mer when a string is not unique, so that nothing is left for the streaming search.
__global__ __launch_bounds__(FIN_TPB) void fin_route_kernel(const uint32_t* pass, const uint32_t* seed, uint32_t n_reads, int strands, int k, uint4* items,
                                                            uint32_t* n_items, uint4* aitems, uint32_t* n_aitems, int probe_items,
                                                            const FinReadDesc* desc, int2* out, int lean) {
    __shared__ uint32_t lds[2][FIN_TPB / 64 + 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per = ((n_reads + gridDim.x - 1) / gridDim.x + FIN_TPB - 1) / FIN_TPB * FIN_TPB;   // reads per block, whole iterations
    const uint32_t r_lo = blockIdx.x * per, r_hi = r_lo + per < n_reads ? r_lo + per : n_reads;
    // verdicts f, v of the forward / reverse strand and their seed nodes sf, sv (NONE: none)
    // (defer: 1 / 2 = the forward / reverse strand's verdict is FIN_PASS_DEFERRED: no item for it now -- its sister's item carries
    //  FIN_WHO_DEFER and the walk kernel's lane goes on with the deferred strand when the sister is done)
    auto verdicts = [&](uint32_t r, uint32_t& f, uint32_t& v, uint32_t& sf, uint32_t& sv, uint32_t& defer) {
        f = NONE; v = NONE; sf = NONE; sv = NONE; defer = 0;
        if (r < r_hi) {
            const uint2 p = *(const uint2*)(pass + 2 * (size_t)r); f = p.x; v = strands == 1 ? p.y : NONE;
            if (f == FIN_PASS_DEFERRED) { f = NONE; defer = 1; }
            if (v == FIN_PASS_DEFERRED) { v = NONE; defer = 2; }
            if (p.x == FIN_PASS_DONE) { f = NONE; v = NONE; defer = 4; }   // (the fast path wrote every slot of this read, fin_prepass.hip: no item, no fill)
            if (seed) { const uint2 sd = *(const uint2*)(seed + 2 * (size_t)r); if (f != NONE) sf = sd.x; if (v != NONE) sv = sd.y; }
        }
    };
    // pass 1: how many items of either kind
    uint32_t cs = 0, ca = 0;
    for (uint32_t r0 = r_lo; r0 < r_hi; r0 += FIN_TPB) {
        uint32_t f, v, sf, sv, df; verdicts(r0 + threadIdx.x, f, v, sf, sv, df);
        const uint32_t nf = (uint32_t)(f != NONE), nv = (uint32_t)(v != NONE);
        const uint32_t af = probe_items ? nf : (uint32_t)(sf != NONE), av = probe_items ? nv : (uint32_t)(sv != NONE);
        ca += af + av; cs += nf + nv - af - av;
    }
    for (int d = 32; d >= 1; d >>= 1) { cs += (uint32_t)__shfl_xor((int)cs, d); ca += (uint32_t)__shfl_xor((int)ca, d); }
    if (lane == 0) { lds[0][wave] = cs; lds[1][wave] = ca; }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t t = 0; for (uint32_t w = 0; w < FIN_TPB / 64; w++) t += lds[threadIdx.x][w];
        lds[threadIdx.x][FIN_TPB / 64] = t ? atomicAdd(threadIdx.x ? n_aitems : n_items, t) : 0u;
    }
    __syncthreads();
    uint32_t base_s = lds[0][FIN_TPB / 64], base_a = lds[1][FIN_TPB / 64];
    __syncthreads();
    // pass 2: write them
    for (uint32_t r0 = r_lo; r0 < r_hi; r0 += FIN_TPB) {
        const uint32_t r = r0 + threadIdx.x;
        uint32_t f, v, sf, sv, df; verdicts(r, f, v, sf, sv, df);
        const bool a_f = f != NONE && (probe_items || sf != NONE), a_v = v != NONE && (probe_items || sv != NONE);
        const uint32_t mine_a = (uint32_t)a_f + (uint32_t)a_v;
        const uint32_t mine_s = (uint32_t)(f != NONE) + (uint32_t)(v != NONE) - mine_a;
        uint32_t xs = mine_s, xa = mine_a;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ys = (uint32_t)__shfl_up((int)xs, d), ya = (uint32_t)__shfl_up((int)xa, d);
            if ((int)lane >= d) { xs += ys; xa += ya; }
        }
        if (lane == 63u) { lds[0][wave] = xs; lds[1][wave] = xa; }
        __syncthreads();
        uint32_t before_s = 0, total_s = 0, before_a = 0, total_a = 0;
        for (uint32_t w = 0; w < FIN_TPB / 64; w++) {
            const uint32_t ts = lds[0][w], ta = lds[1][w];
            if (w < wave) { before_s += ts; before_a += ta; }
            total_s += ts; total_a += ta;
        }
        __syncthreads();
        uint32_t at_s = base_s + before_s + xs - mine_s, at_a = base_a + before_a + xa - mine_a;
        const bool both = f != NONE && v != NONE;
        const int cf = (int)f - 2 * k, cv = (int)v - 2 * k;
        // out != null: the output is NOT prefilled.  A read with one strand to search: that strand's lane writes every slot, pairs and
        // (-1,-1) alike (FIN_WHO_GAPS).  A read with none or both: its slots are prefilled here, a wave per read.
        const uint32_t gaps = ((out && !both && (f != NONE || v != NONE)) ? FIN_WHO_GAPS : 0u) | ((df & 3u) ? FIN_WHO_DEFER : 0u);   // (a deferred sister: only with `out`, fin_launch_search_v4)
        const uint32_t who_v = r | 0x80000000u | (both ? 0x40000000u : 0u) | gaps;
        if (out) {
            const bool fill = r < r_hi && !(gaps & FIN_WHO_GAPS) && df != 4u;
            FinReadDesc d = {0, 0, 0};
            if (fill) d = desc[r];
            uint64_t m = __ballot(fill);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t o_base = (uint32_t)__shfl((int)d.out_off, src), o_len = (uint32_t)__shfl((int)d.len, src);
                const uint32_t o_nk = o_len >= (uint32_t)k ? o_len - (uint32_t)(k - 1) : 0u;
                for (uint32_t i = lane; i < o_nk; i += 64) out[(size_t)o_base + i] = make_int2(-1, -1);
            }
        }
        const uint32_t mark = lean ? FIN_PLACE_MARK : FIN_SEED_MARK;   // (lean tables: the pre-pass's seeds are places, not nodes)
        if (a_f) aitems[at_a++] = make_uint4(r | gaps, f, sf, sf != NONE ? mark : 0u);   // (node NONE: a probe item)
        else if (f != NONE) items[at_s++] = make_uint4(r, (uint32_t)(cf > 0 ? cf : 0), f, 0u);   // (stream items only exist with a prefilled output)
        if (a_v) aitems[at_a] = make_uint4(who_v, v, sv, sv != NONE ? mark : 0u);
        else if (v != NONE) items[at_s] = make_uint4(who_v, (uint32_t)(cv > 0 ? cv : 0), v, 0u);
        base_s += total_s; base_a += total_a;
    }
}

// ---- walk kernel: anchor items -> lookups, walk, output, next stream item; probe items -> absence proofs -> next stream item ----
// (LONGK: k > 32 -- a probe string may be longer than the 32 bases a lane keeps in registers and then reads the rest from the read's chunks;
//  the kernel for short k does not carry that path)
// LEAN (round 5; only with k <= 32): the instantiation for lean tables -- no prefix-table / rank-record states (W_PROBE1, W_PROBEX: a probe is a block of the
// directional string filter), whose registers hold a SECOND bucket of the k-mer table instead: behind a k-mer the table does not have, the next end's k-mer is
// the old one shifted by a base, already in the window the key was made from -- both look-ups go out in one epoch, and a run of absent k-mers (a third of
// chr1's lane-epochs, 60 % of a repeat-rich genome's) moves two ends per epoch instead of one.
template <bool LONGK, bool LEAN = false>
__device__ __forceinline__ void fin_walk_body(const FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                              const uint4* items_in, const uint32_t* n_in, uint4* items_out, uint32_t* n_out,
                                              uint32_t* list, uint32_t* n_list, int last_round, uint32_t* work_counter, uint32_t* n_sister_out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    const char* const blk_base = (const char*)ix.blocks;
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)*n_in);
    const uint32_t C0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[0]), C1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[1]),
                   C2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[2]), C3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[3]),
                   C4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ix.C[4]);
    const int PT = (int)ix.ptab_t;
    const int kf_every = ix.fbf ? FIN_W_KF_LEAN_EVERY - 1 : 7;   // (a probe is one load with lean tables, a table entry and up to four node blocks without)
    // a k-mer table this kernel can ask.  k <= 63: the look-up registers hold the key's two words and a claim is compared in registers (W_RES4 / W_KFV);
    // k >= 64 (round 5, LONGK only): the key's ceil(k/32) words are folded into the hash one epoch each as the chunk cache brings them (W_KF0B), nothing of
    // the k-mer is kept, and a claim is borne out by the re-anchoring block's comparison of the k bases with the text at the claimed place (W_REANCH)
    const bool have_kt = ix.kt3 != nullptr && (k <= 63 || LONGK);
    const bool kt_wide = LONGK && k >= 64;
    const bool has_anchor = ix.pos != nullptr || have_kt;                // an anchor table, or (lean tables) the k-mer table alone
    const int PM = ix.fbf ? (int)ix.cbf_m : min(PT + FIN_V3_PM_ADD, k);  // (lean tables: a probe string is what the directional string filter holds)
    const int MARGIN = 2 * k;
    const int DELTA = PT > 0 ? min(k - 1, PT + FIN_V3_DELTA_ADD) : k - 1;

    // ---- per-lane state ----
    uint32_t pc = W_ITEM0;
    uint32_t who = 0;                                   // read | strand << 31
    uint32_t r_pk = 0, r_len = 0, r_out = 0;   // (first packed chunk of the read: a batch has fewer than 2^28 chunks; strand and write mode are read from `who` where needed: bit 31, bit 30)
    int end = 0;                                        // anchor: its k-mer end; afterwards the next position
    uint32_t a_colex = 0, a_dl = 0;                     // anchor: node, distance | use_branch << 31
    uint32_t res_g = 0, res_idx = 0;
    uint32_t wg = 0, w_u = 0, w_ustart = 0, w_uend = 0;
    int& wend = end;            // the walk's next read position: lives from the anchor's resolution on, when `end` has done its duty
    uint32_t run_pos = 0, run_len = 0, run_off = 0;
    uint32_t& run_u = w_u;      // a run lies in the unitig of its anchor; it is closed before the next anchor is resolved
    // a finished run waiting for this epoch's write-out (r_out / r_len / who are the lane's own: a new item's descriptor
    // arrives two epochs after the old item is done at the earliest)
    // (the run itself stays in run_* until then; with FIN_WHO_GAPS the write-out also covers the absent slots in front of it -- gap0 -- and,
    //  when the item ends, behind it -- gap1; w_next = first slot of the strand not written yet)
    uint32_t w_next = 0, gap0 = 0, gap1 = 0;
    // FIN_WHO_DEFER: the stretch of this strand's slots that no pair of this lane fills, first | last << 16 (first > last: none) -- where the
    // deferred sister strand has to be searched; t_stop: the last k-mer end this item resolves (a deferred strand's item: the end of its
    // stretch; else the strand's last)
    uint32_t hull = 0x0000FFFFu;   // (a deferred strand's own item keeps its t_stop here: it has no sister to report a stretch to)
    FinChunkCache ck;
    uint32_t ttag = NONE; uint4 wt = make_uint4(0, 0, 0, 0);
    // probe items
    // (probes: the interval, the first unresolved k-mer end and the first non-ACGT offset live in the anchor's registers, which are
    //  dead once the lookups are done -- a probe item has no anchor, a bridging probe comes after its anchor's walk)
    uint32_t& il = a_colex; uint32_t& ir = a_dl; uint32_t& t0 = res_g; uint32_t& pfi = res_idx; int pp = 0, pe = 0; uint64_t pcode = 0;
    // text re-anchoring behind a bad read position (as in kernel 3): the bad position, the text position aligned with it
    uint32_t br_E = 0, br_tE = 0;
    // a probe string is q[pp..plim]: the PM bases that end at t0; across a bad position pulled back to contain it and then as long as it
    // goes on matching, up to t0; pfull: the whole k-mer that ends at t0 -- asked when a string that ends at t0 is not unique
    // ptried: across a bad position E the first string asked is the SHORT one that starts T-1 bases before E, so that E lies inside the
    // prefix-table key and one table entry settles it (the full-length string that ends at t0 <= E+3 has E behind its key: a table entry
    // plus up to four node blocks); only if that short string occurs is the full-length one asked (ptried)
    // (a whole-k-mer look-up decides one k-mer end; in a stretch whose probe strings are all repeated and whose k-mers are absent that
    //  is k look-ups of k bases: such a strand runs into the item's epoch budget and goes to kernel 3)
    // pguessed: a string across a bad position that stops short of t0 but ends one node has been used as a GUESS of where the read
    // lies now (k > 32: after an indel the strings behind it match, 32 bases do not reach t0) -- the k-mer at t0 is compared with the
    // text there; a comparison can only find k-mers, so any guess is sound, and a guess that fails is not repeated
    // the lane's flags, in ONE register (as separate bools each took one): pend = a finished run waits for this epoch's write-out;
    // bridging = the probes in progress are those across the bad position br_E; pfull / ptried / pguessed as described above
    // (n_sister: the deferred strands this lane went on with -- a statistic, summed into *n_sister_out when the wave ends)
    // win_rc: a k-mer that ends in the text window in `wt` has its reverse complement in the index too (FinDevIndex::rcwin) -- reporting from
    // that window taints.  tainted: this item used the streaming search (hand_on) or an anchor that is not a seed (a whole-k-mer look-up, whose entry may name a place
    // that does not spell the k-mer): what it reports proves nothing about the other strand -- a deferred sister is then searched in full
    // kt_claim: the anchor W_RES4 is about to locate is a CLAIM of the k-mer table (a tag match, FinDevIndex::kt3) whose text window `wt` holds: W_RES4 compares the
    // k-mer's codes (pcode, il | ir << 32) with the text at the claimed place first -- equal: an anchor as on round 4's exact-key tables; not equal (another
    // k-mer with the same 30-bit tag): the read goes to kernel 3, which asks no table.  (The pre-pass's place items are compared there, fin_prepass.hip.)
    struct { uint32_t pend : 1, bridging : 1, pfull : 1, ptried : 1, pguessed : 1, bounded : 1, tainted : 1, win_rc : 1, tabent : 1, kt_claim : 1, bs : 3, bs_off : 1, n_sister : 18; } fl = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define t_stop (fl.bounded ? hull : r_len - 1u)
#define pend fl.pend
#define bridging fl.bridging
#define pfull fl.pfull
#define ptried fl.ptried
#define pguessed fl.pguessed
    // (plim is not kept: it is min(t0, pp + 31) while the string lies across a bad position, else t0 -- plim_now())
    auto plim_now = [&]() -> int { return bridging ? min((int)t0, pp + 31) : (int)t0; };
    // who bit 30: this strand's pairs may only fill slots that are still (-1,-1) (the reverse strand of a read whose two strands are both searched)
    FinRecCache rc;
    // LEAN: the second look-up of an epoch -- its bucket (kb0, kb1), where it lies (kb_idx; NONE: no second look-up under way) and its k-mer's tag
    uint4 kb0 = make_uint4(0, 0, 0, 0), kb1 = kb0; uint32_t kb_idx = NONE, tag2 = 0;
    constexpr uint32_t Q_KB = 512u;
    uint32_t budget = 0;
    uint4 aux = make_uint4(0, 0, 0, 0);
    const void* q_aux = nullptr;
    uint32_t q = 0;
    FinWorkRangesT<FIN_WALK_RANGE> wr; wr.init();
    FinWaveQueue oq, lq;   // this wave's slots in the stream-item queue and in kernel 3's list
#ifdef FIN_W_DEBUG
    uint32_t dbg_ep = 0, dbg_wave = 0;
    if (lane == 0 && n_items > 100000u) (void)atomicCAS(&g_fin_wtime[8192], 0ull, (unsigned long long)wall_clock64());   // (the launch's first wave sets it)
#endif

    auto req_recs = [&](uint32_t l, uint32_t r, uint32_t c) { rc.request(l, r, c, q); };
    // update_sbwt_interval (formula: common.hh:26-36) with the cached rank records: 0 = data requested, 1 = ok, 2 = (-1,-1)
    auto extend_try = [&](uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) -> int { return rc.extend(c, l, r, n, C0, C1, C2, C3, C4, q, nl, nr); };
    // the chunks of this item's strand: [forward | reverse complement] per read
    auto strand_chunks = [&]() -> const uint4* { return packed + r_pk + ((who >> 31) ? (r_len + 31u) >> 5 : 0u); };
    // (fetching the chunk behind along with one that is needed, as the pre-pass does, measured no difference here: CHANGELOG.md 5.6)
    auto need_chunk = [&](int ci) -> bool { return ck.need(ci, strand_chunks, q, q_aux); };
    auto hull_add = [&](uint32_t first, uint32_t last) {   // slots [first, last] stay open
        const uint32_t lo = min(hull & 0xFFFFu, first), hi = max(hull >> 16, last);
        hull = lo | (hi << 16);
    };
    auto close_run = [&]() {
        if (run_len && !pend) {
            pend = true; gap0 = (who & FIN_WHO_GAPS) ? run_pos - w_next : 0u;
            if ((who & FIN_WHO_DEFER) && run_pos > w_next) hull_add(w_next, run_pos - 1u);
            w_next = run_pos + run_len;
        }
    };

    for (;;) {
        // ================= 1. serve this epoch's requests =================
        if (q & Q_AUX) aux = load16u(q_aux);
        // (the bucket's second half travels in the text window's register -- the kernel has none to spare: 96 of 102, and four more spilled -- ; no walk is under way
        //  while a k-mer is looked up, and the comparison behind a claim asks for its window again)
        if (q & Q_AUX2) { wt = load16u((const char*)q_aux + 16); ttag = NONE; }   // (a bucket of the k-mer table is 32 bytes: slots 2 and 3)
        if (!LEAN) rc.serve(q, blk_base);
        if (LEAN && (q & Q_KB)) { const char* const a = (const char*)(ix.kt3 + kb_idx); kb0 = load16u(a); kb1 = load16u(a + 16); }
        ck.serve(q, aux, strand_chunks);
        if (q & Q_TEXT) wt = load16u((const void*)(ix.concat + ((size_t)ttag << 2)));   // (the text window has its own load: a walk step needs read chunk and text together)
        if ((q & Q_TEXT) && ix.rcwin) fl.win_rc = (ix.rcwin[ttag >> 3] >> (ttag & 7u)) & 1u;   // (indexes with reverse-complement pairs only)
        q = 0;

        // ================= 2. blocks =================
        const uint32_t pc0 = pc;
#ifdef FIN_W_DEBUG
        dbg_wave++;
        if (pc0 > W_DESC) { dbg_ep++; if (fl.bounded) atomicAdd(&g_fin_witem[90], 1ull); }
        {
            atomicAdd(&g_fin_wstate[pc0 < 32u ? pc0 : 31u], 1ull);
            uint32_t present = 0;
            for (uint32_t st = 1; st < 24u; st++) if (__any(pc0 == st)) present++;
            const uint64_t live = __ballot(pc0 != W_DONE);
            if ((threadIdx.x & 63u) == 0u) { atomicAdd(&g_fin_wstate[32], 1ull); atomicAdd(&g_fin_wstate[33], (unsigned long long)present); atomicAdd(&g_fin_wstate[34], (unsigned long long)__popcll(live)); }
        }
#endif
        bool emit = false; uint4 emit_item = make_uint4(0, 0, 0, 0);   // the stream item this lane hands on
        bool give_up = false;                                          // the read goes to kernel 3 instead
        // where the streaming search goes on after position e is reached: restart point, silence, what is exact from where
        auto hand_on = [&](int c, int silent, int exact) {
            emit = !last_round; give_up = last_round != 0;
            fl.tainted = 1;
            // (what this lane leaves unwritten it fills with (-1,-1) now: the kernels behind only write pairs.  A REVERSE strand whose forward
            //  sister is deferred: the sister is searched by this lane in a moment, the rest of this strand rounds later -- its pairs may
            //  then only fill, so that a forward pair stays whichever comes first)
            emit_item = make_uint4((who & ~(FIN_WHO_GAPS | FIN_WHO_DEFER)) | (((who & FIN_WHO_DEFER) && (who >> 31)) ? 0x40000000u : 0u), (uint32_t)c, (uint32_t)silent, (uint32_t)exact);
            pc = W_ITEM0;
        };
        // an anchor table entry that names no place where the text spells the k-mer at `end` (a guess that cannot be used; an unverified
        // entry: the reference's answer for that node is not a place of its k-mer, or nothing is known): the whole k-mer is looked up --
        // if it is present its node's entry is the reference's answer, verified or not
        auto seed_unusable = [&]() {
            WDBG(0);
            t0 = (uint32_t)end;   // (t0 is res_g's register)
            bridging = false;
            if (have_kt) { pe = 0; pc = W_KF0; } else { pfull = true; pc = W_PROBE0; }
        };
        // text re-anchoring / seed verification found q[E+1..E+k] in the text behind br_tE: the run starts with this k-mer and the walk goes
        // on behind it (true: the walk has text left to compare)
        auto reanch_found = [&]() -> bool {
            const int E = (int)br_E;
            if (fl.win_rc) fl.tainted = 1;   // (the k-mer's last base was compared with the window in `wt`: its end lies in that window)
            run_pos = (uint32_t)(E + 1); run_len = 1; run_u = w_u; run_off = br_tE + 1u - w_ustart;
            wg = br_tE + (uint32_t)k; wend = E + k + 1; bridging = false;
            if (wend == (int)r_len) { close_run(); pc = W_ITEM0; return false; }
            pc = W_WALK;
            return wg + 1u < w_uend;
        };
        // ---- dictionary lookups (FinimizerIndex.hh:148-174), one dependent load per epoch ----
        if (pc >= W_RES1 && pc <= W_RES5) {
        if (pc == W_RES5) {     // aux = ends_p[res_idx .. res_idx+3]
            const uint32_t gs = res_g - (uint32_t)(k - 1);
            bool done = true;
            if (gs < aux.y) { w_u = res_idx; w_ustart = aux.x; w_uend = aux.y; }
            else if (gs < aux.z) { w_u = res_idx + 1; w_ustart = aux.y; w_uend = aux.z; }
            else if (gs < aux.w) { w_u = res_idx + 2; w_ustart = aux.z; w_uend = aux.w; }
            else { res_idx += 3; q_aux = (const void*)(ix.ends + res_idx); q |= Q_AUX; done = false; }
            if (done && kt_wide && fl.kt_claim) {
                // a wide key's claim: the k bases against the text at the claimed place, by the re-anchoring comparison -- entered as a seed's is, as if
                // the position in front of the k-mer had been a bad one.  Equal: the run starts there; anything else: another k-mer's tag (the plain kernel decides)
                if (gs >= w_ustart && res_g < w_uend) { br_E = (uint32_t)(end - k); br_tE = gs - 1u; pe = 0; t0 = (uint32_t)end; pc = W_REANCH; }
                else { fl.kt_claim = 0; give_up = true; pc = W_ITEM0; }
            } else
            if (done) {
                run_pos = (uint32_t)(end - (k - 1)); run_len = 1; run_u = w_u; run_off = gs - w_ustart;
                wg = res_g;
                end++; wend = end;
                if (end == (int)r_len) { close_run(); pc = W_ITEM0; }
                else {
                    pc = W_WALK;
                    // the walk's first step compares the read with the text right after the anchor: ask for both now
                    if (((res_g + 1u) >> 6) != ttag && res_g + 1u < w_uend && !(q & Q_TEXT)) { ttag = (res_g + 1u) >> 6; q |= Q_TEXT; }
                    (void)need_chunk(wend >> 5);
                }
            }
        }
        if (pc == W_RES4 && fl.kt_claim && !kt_wide) {   // wt = the text window that holds the claimed place [res_g - k + 1, res_g] whole: is it this lane's k-mer?
            fl.kt_claim = 0;
            uint64_t x0, x1;
            fin_text_kmer(wt, wt, (res_g - (uint32_t)(k - 1)) & 63u, (uint32_t)k, x0, x1);
            if (x0 != pcode || (LONGK && x1 != ((uint64_t)il | ((uint64_t)ir << 32)))) { WDBG(13); give_up = true; pc = W_ITEM0; }   // another k-mer with this tag: kernel 3 decides
            else a_dl = 0u;   // (an anchor of distance 0 from its answer; il | ir have done their duty)
        }
        if (pc == W_RES4) { res_idx = aux.x; q_aux = (const void*)(ix.ends + res_idx); q |= Q_AUX; pc = W_RES5; }
        if (pc == W_RES3) {     // aux.x = global_offsets[rank] (common.hh:71) or the unitig start (common.hh:65)
            const bool ub = (a_dl >> 31) != 0u; const uint32_t dl = a_dl & 0x7FFFFFFFu;
            const uint32_t raw = aux.x;   // a seed: pos[node] (a_dl: 0, or how far short of `end` a guess's string stopped)
            res_g = bridging ? raw + a_dl : ub ? raw + (uint32_t)(k - 1) + dl : raw + dl;
            const uint32_t gs = res_g - (uint32_t)(k - 1);
            if (bridging && raw >= FIN_POS_DUMMY && raw != NONE && a_dl == 0u) {
                // the seed string ends only a dummy node that holds d bases: no k-mer ends at `end`, nor at the next k-d-1 positions
                // (nodes of the extensions are that dummy's descendants).  Probing goes on at end + k - d.
                bridging = false;
                t0 = (uint32_t)end + (uint32_t)k - (raw & 0xFFu);   // (t0 is res_g's register)
                pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (uint32_t)W_PROBE0;
            } else
            if (bridging && raw < FIN_POS_DUMMY && (aux.y & FIN_POS_UNVERIFIED)) seed_unusable();
            else
            if (bridging && raw < FIN_POS_DUMMY) {
                // A SEED: the place the reference reports for the k-mer that ends at `end` if that k-mer is its node's (a guess: a place it
                // may have); the text there spells the node's k-mer (a verified entry), and the entry holds the unitig and its bounds too.
                // Is the k-mer there?  The comparison of its k bases with the text is the re-anchoring block's, entered as if
                // the position in front of the k-mer had been a bad one: equal -> the run starts here; a base that differs -> the k-mers
                // across it are proven absent by probes and the k-mer behind it is compared next
                w_u = aux.y; w_ustart = aux.z; w_uend = aux.w;
                if (res_g >= raw && gs >= w_ustart && res_g < w_uend) { br_E = (uint32_t)(end - k); br_tE = gs - 1u; pe = 0; t0 = (uint32_t)end; pc = W_REANCH; }
                else seed_unusable();   // (a guess whose k-mer would cross its unitig's end)
            } else
            if (!bridging && gs < ix.total_len) {
                // (an anchor that is not a seed: the streaming search's -- dictionary look-ups -- or a whole k-mer's entry of the anchor table, which
                //  may name a place that does not spell the k-mer: those taint, §4.14; a verified entry is a place like a seed's)
                // (on an index with reverse-complement pairs every such anchor taints: its k-mer is reported without a text comparison, so no
                //  window flag passes by -- found by tools/fuzz_defer.py)
                if (!fl.tabent || (aux.y & FIN_POS_UNVERIFIED) || ix.rcwin) fl.tainted = 1;
                fl.tabent = 0;
                q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = W_RES4;
            }
            else if (bridging) seed_unusable();
            else { give_up = true; pc = W_ITEM0; }   // unreachable on a consistent index (the reference reads out of bounds): kernel 3 reports it as absent
        }
        if (pc == W_RES1) {     // aux = the 16 bytes of FinBlockInfo that hold this dictionary's mask and rank
            const bool ub = (a_dl >> 31) != 0u;
            const uint64_t below = ~(~0ull << (a_colex & 63u));
            const uint64_t mask = aux.y | ((uint64_t)aux.z << 32);
            const uint32_t rank = (ub ? aux.w : aux.x) + (uint32_t)__popcll(mask & below);
            q_aux = ub ? (const void*)(ix.ends + rank) : (const void*)(ix.goff + rank);
            q |= Q_AUX; pc = W_RES3;
        }
        }
        // ---- probe items: absence proofs from k-mer end t0 on (see fin_kernel_v3.hip, PROBE mode) ----
        auto probe_fail = [&]() {
            pfull = false; ptried = false; pguessed = false;
            t0 = (uint32_t)(pp + k);
            if (t0 > t_stop) pc = W_ITEM0;
            else if (bridging && t0 > br_E + (uint32_t)(k - 1)) { pe = 0; pc = W_REANCH; }   // every k-mer that contains the bad position is proven absent
            else pc = W_PROBE0;
        };
        // a probe string occurs: nothing is proven about end t0.  SEED (index with a seed table, string ending at t0, suffix of exactly
        // one node): that node's k-mer is the only one that can end at t0 -- look its place up and compare (W_RES3 .. W_REANCH);
        // otherwise the streaming search takes over, restarted 2k before t0
        //   * the string was the whole k-mer (pfull): its node is the k-mer's -- an anchor like any other, its place from the seed table;
        //   * the string ends at t0 but several nodes end with it, or it is a string across a bad position that stops short of t0: the whole
        //     k-mer that ends at t0 is looked up next
        auto probe_pass = [&]() {
            if (bridging && !ptried && (int)t0 - pp + 1 < PM) { ptried = true; pc = W_PROBE0; return; }   // the short string occurs: nothing proven, ask the full-length one
            ptried = false;
            const int plim = plim_now();
            const bool at_t0 = plim == (int)t0;
            if (!LEAN && ix.pos && il == ir && (at_t0 || (bridging && !pguessed))) {
                // a seed (the string ends at t0), or a guess (it stops plim short of t0: the read is taken to lie t0 - plim bases further on)
                if (!at_t0) pguessed = true;
                end = (int)t0;
                q_aux = (const void*)(ix.pos + il); q |= Q_AUX; pc = W_RES3;
                if (pfull) { pfull = false; bridging = false; a_dl = 0u; fl.tabent = 1; }   // (distance 0 from "the dictionary's" offset, which is pos[node])
                else { bridging = true; a_dl = t0 - (uint32_t)plim; }        // (ir, the interval's end, has done its duty)
            } else if ((at_t0 || bridging) && has_anchor && !pfull) {
                // the whole k-mer that ends at t0 is asked next -- of the k-mer table where there is one (k <= 31): one 16-byte load says
                // whether it is there and which node it is; else by a look-up through the SBWT (prefix table + k-T extends)
                bridging = false;
                if (have_kt) { pe = 0; pc = W_KF0; } else { pfull = true; pc = W_PROBE0; }
            }
            else { WDBG(bridging ? 1 : (il != ir ? 2 : 3)); pfull = false; bridging = false; hand_on(max(0, (int)t0 - MARGIN), (int)t0, 0); }
        };
        if (pc == W_PROBEF) {   // aux = the directional string filter's block of the string q[pp .. pp+m-1] (pcode: its codes)
            const uint32_t m = ix.cbf_m;
            const uint64_t f = pcode & (m >= 32u ? ~0ull : ((1ull << (2u * m)) - 1ull));
            const uint64_t h = fin_cbf_hash(f);
            uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
#pragma unroll
            for (int i = 0; i < FIN_CBF_BITS; i++) {
                const uint32_t pb = (uint32_t)(h >> (7 * i)) & 127u, bit = 1u << (pb & 31u);
                m0 |= (pb >> 5) == 0u ? bit : 0u; m1 |= (pb >> 5) == 1u ? bit : 0u; m2 |= (pb >> 5) == 2u ? bit : 0u; m3 |= (pb >> 5) == 3u ? bit : 0u;
            }
            const bool known = (aux.x & m0) == m0 && (aux.y & m1) == m1 && (aux.z & m2) == m2 && (aux.w & m3) == m3;
            if (known && fl.bs) {   // (looking for the absent string of a k-mer the table does not have: this one occurs too)
                // the k-mer's first bases: every string of it occurs -- the next end, and no further back-scans in this stretch (a repeat: its k-mers are
                // absent though their strings occur; scanning behind every one of them cost k63_repeats 18 ms of 46 under lean tables)
                if (pp <= (int)t0 - k + 1) { fl.bs = 0; fl.bs_off = 1; t0++; pe++; pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (uint32_t)W_KF0; }
                else { fl.bs++; pc = W_PROBE0; }
            } else if (known) { WDBG(10); il = 0; ir = 1; probe_pass(); }   // it occurs (or the filter takes it to): nothing proven; several nodes for all we know
            else { WDBG(11); fl.bs = 0; fl.bs_off = 0; probe_fail(); }
        }
        if (!LEAN && pc == W_PROBE1) {
            if (aux.x > aux.y) probe_fail();
            else {
                il = aux.x; ir = aux.y; pe = pp + PT;
                if (pe > plim_now()) probe_pass();
                else {
                    pc = W_PROBEX;
                    const uint32_t off = (uint32_t)(pe - pp);
                    if (off < pfi) req_recs(il, ir, (uint32_t)(pcode >> (2 * off)) & 3u);
                }
            }
        }
        if (!LEAN && pc == W_PROBEX) {
            // the string's first 32 bases are in pcode (pfi: the first non-ACGT one among them); a longer string -- the whole k-mer of a
            // long k, a string that goes on to a far t0 -- reads the rest from the read's chunks as it goes
            const uint32_t off = (uint32_t)(pe - pp);
            uint32_t code; bool have = true, bad;
            if (!LONGK || off < 32u) { code = (uint32_t)(pcode >> (2 * off)) & 3u; bad = off >= pfi; }
            else {
                have = need_chunk(pe >> 5);
                const uint32_t j = (uint32_t)pe & 31u;
                code = (uint32_t)(ck.bcodes >> (2 * j)) & 3u; bad = !((ck.bvalid >> j) & 1u);
            }
            if (have) {
                if (bad) probe_fail();   // a non-ACGT base: no k-mer contains it
                else {
                    uint32_t nl, nr;
                    const int rc = extend_try(code, il, ir, nl, nr);
                    if (rc == 2) probe_fail();
                    else if (rc == 1) {
                        il = nl; ir = nr; pe++;
                        if (pe > plim_now()) probe_pass();
                        else if ((!LONGK || off + 1 < 32u) && off + 1 < pfi) req_recs(il, ir, (uint32_t)(pcode >> (2 * (off + 1))) & 3u);
                    }
                }
            }
        }
        // ---- comparison of the read with the unitig text, up to 32 bases per epoch: ONE block for its two users ----
        //  W_WALK    the match runs on along the unitig text (walk_in_unitigs, FinimizerIndex.hh:47-102): every further equal base is a pair
        //  W_REANCH  text re-anchoring: is q[E+1..E+k] the text behind the bad position E?  (pe = bases found equal so far; also the
        //            verification of a seed, entered with E = the position in front of its k-mer)
        if (pc == W_SAFE) {   // aux = the word of the safe-place bitmap that holds the bit of text position br_tE + k
            const uint32_t tp = br_tE + (uint32_t)k;
            const uint64_t word = aux.x | ((uint64_t)aux.y << 32);
            if ((word >> (tp & 63u)) & 1ull) {
                if (reanch_found()) {   // ask now for what the walk's first step compares
                    if (((wg + 1u) >> 6) != ttag) { ttag = (wg + 1u) >> 6; q |= Q_TEXT; }
                    (void)need_chunk(wend >> 5);
                }
            } else {
                // the k-mer is in the text here, but the reference reports it elsewhere (a k-mer with several places, or one whose
                // finimizer's stored place lies elsewhere).  It is PRESENT: its node's entry of the anchor table is the reference's answer --
                // the whole k-mer that ends at t0 = E + k is looked up (k-mer table, or through the SBWT), an anchor like any other.
                // (round 2 and the first forms of round 3 sent such a strand back to the streaming search: 2.6 % of chr1_dups' reads)
                WDBG(4);
                bridging = false;
                if (have_kt) { pe = 0; pc = W_KF0; } else { pfull = true; pc = W_PROBE0; }
            }
        }
        {
            const bool is_walk = pc == W_WALK, is_re = pc == W_REANCH;
            bool brk = false, at_uend = false, go = false;
            int c_rp = 0; uint32_t c_tp = 0, c_lim = 0;   // read position, text position, bases left to compare
            if (is_walk) {
                c_tp = wg + 1u; c_rp = wend;
                if (c_tp >= w_uend) { brk = true; at_uend = true; }   // (>: an anchor whose k-mer ends beyond its unitig, FinimizerIndex.hh:51-53)
                else { go = true; c_lim = min(w_uend - c_tp, r_len - (uint32_t)wend); }
            }
            if (is_re) {   // (t0 = E + k < r_len here: the k-mer lies inside the read)
                if (pe == k) {   // (all k bases equal, the bitmap's word could not be asked for in that epoch)
                    if (!(q & Q_AUX)) { q_aux = (const void*)(ix.safe + ((br_tE + (uint32_t)k) >> 6)); q |= Q_AUX; pc = W_SAFE; }
                } else
                if (br_tE + (uint32_t)k >= w_uend) {   // the unitig ends inside that k-mer: a probe at t0 = E+k (seed), or the streaming search, decides
                    if (kt_wide && fl.kt_claim) { fl.kt_claim = 0; give_up = true; pc = W_ITEM0; }   // (a claim's place lies inside its unitig: not this one's)
                    else if (has_anchor) { bridging = false; pc = W_PROBE0; } else probe_pass();
                } else { go = true; c_rp = (int)br_E + 1 + pe; c_tp = br_tE + 1u + (uint32_t)pe; c_lim = (uint32_t)(k - pe); }
            }
            if (go) {
                const bool chunk_ok = need_chunk(c_rp >> 5);
                if ((c_tp >> 6) != ttag && !(q & Q_TEXT)) { ttag = c_tp >> 6; q |= Q_TEXT; }
                if (chunk_ok && (c_tp >> 6) == ttag && !(q & Q_TEXT)) {
                    const uint32_t j = (uint32_t)c_rp & 31u, t = c_tp & 63u;
                    const uint64_t rb = ck.bcodes >> (2 * j);
                    const uint32_t inv = ~(ck.bvalid >> j) | (j ? 0xFFFFFFFFu << (32 - j) : 0u);
                    const uint64_t lo = wt.x | ((uint64_t)wt.y << 32), hi = wt.z | ((uint64_t)wt.w << 32);
                    const uint64_t tb = t < 32 ? ((lo >> (2 * t)) | (t ? hi << (64 - 2 * t) : 0ull)) : (hi >> (2 * (t - 32)));
                    const uint32_t tav = t < 32 ? 32u : 64u - t;
                    const uint32_t nmax = min(min(32u - j, tav), c_lim);
                    const uint64_t x = rb ^ tb;
                    const uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
                    const uint32_t mm = y ? (uint32_t)(__ffsll((long long)y) - 1) >> 1 : 32u;
                    const uint32_t fi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
                    const uint32_t nadv = min(min(mm, fi), nmax);
                    bool more = false;   // the comparison goes on next epoch
                    if (is_walk) {
                        const uint32_t lim_u = w_uend - c_tp;   // text left in this unitig
                        if (nadv && fl.win_rc) fl.tainted = 1;
                        run_len += nadv; wg += nadv; wend += (int)nadv;
                        if (wend == (int)r_len) { close_run(); pc = W_ITEM0; }
                        else { brk = nadv < nmax || nadv == lim_u; at_uend = nadv == lim_u; more = !brk; }
                    } else {
                        pe += (int)nadv;
                        if (nadv < nmax && kt_wide && fl.kt_claim) { fl.kt_claim = 0; WDBG(13); give_up = true; pc = W_ITEM0; }   // the text there does not spell the k-mer: a shared tag
                        else
                        if (nadv < nmax) {   // the next bad position
                            // (the comparison was a seed's own -- it began in front of the k-mer that ends at `end` -- and the seed was exact:
                            //  that k-mer is decided, absent; a lane with nothing left to resolve is done)
                            if ((int)br_E + k == end && a_dl == 0u && t0 == (uint32_t)end) t0++;
                            br_E = (uint32_t)c_rp + nadv; br_tE = c_tp + nadv;
                            pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (uint32_t)W_PROBE0;
                        }
                        else if (pe == k) {
                            // present, and in the text here.  On an index with duplicated k-mers (ix.safe) that is where the reference
                            // reports it only if the bit of this text position says so -- except for an exact seed's own k-mer, whose
                            // entry is the reference's answer (verified: the bit is set)
                            const bool exact_seed = (int)br_E + k == end && a_dl == 0u;
                            if (kt_wide) fl.kt_claim = 0;   // (a claim borne out is the reference's answer for the k-mer: an exact seed)
                            if (!ix.safe || exact_seed) more = reanch_found();
                            else if (!(q & Q_AUX)) { q_aux = (const void*)(ix.safe + ((br_tE + (uint32_t)k) >> 6)); q |= Q_AUX; pc = W_SAFE; }
                        } else more = true;
                    }
                    if (more) {   // ask now for what the next step compares (this step's chunk and window are dead): a step per epoch
                        const bool w = pc == W_WALK;
                        const uint32_t tp2 = w ? wg + 1u : br_tE + 1u + (uint32_t)pe;
                        if ((tp2 >> 6) != ttag) { ttag = tp2 >> 6; q |= Q_TEXT; }
                        (void)need_chunk((w ? wend : (int)br_E + 1 + pe) >> 5);
                    }
                }
            }
            if (brk) {
                // The walk ends before position wend; the normal path applies there again (FinimizerIndex.hh:148-183), which needs the
                // streaming state at wend.  It is rebuilt, never resumed (CHANGELOG.md 4.6):
                //  * a base that disagrees with the text: verified short restart DELTA bases back -- kmer_start and start of a search begun
                //    at c are max(c, true value) and only move forward, so once kmer_start has passed c (checked by the stream kernel when
                //    it arrives at wend, marked by a negative exact_from) both are true from there on;
                //  * without a prefix table (DELTA = k-1): k-1 back, presence exact from wend, everything from wend+k (2k-1 bases on);
                //  * the unitig ended (the read goes on in another one, the next k-mer is usually present at once): full margin 2k.
                close_run();
                if (!at_uend && ix.text_anchors) {
                    // TEXT RE-ANCHORING (see kernel 3's walk block): prove the k-mers across the bad position absent, then compare the
                    // k-mer behind it with the text -- the streaming search is not needed again unless that fails
                    br_E = (uint32_t)wend; br_tE = wg + 1u; t0 = (uint32_t)wend; bridging = true; pc = W_PROBE0;
                } else
                if (at_uend && has_anchor) {
                    // the unitig ended and the read goes on (in another unitig, if anywhere): a probe at the next k-mer end either proves
                    // it absent or yields a seed
                    t0 = (uint32_t)wend; bridging = false; pc = W_PROBE0;
                } else
                if (!at_uend && DELTA < k - 1) hand_on(wend - DELTA, wend, -(wend + k));
                else if (!at_uend) hand_on(wend - (k - 1), wend, wend + k);
                else hand_on(max(0, wend - MARGIN), wend, 0);
            }
        }
        // ---- k-mer table (FinDevIndex::ktab): is the k-mer that ends at t0 in the index, and which node is it? ----
        // The k-mer table does not have the k-mer that ends at t0.  Lean tables: its last PM bases usually DO occur (that is why it was asked), so a string
        // that occurs in no unitig lies further back in it -- found by asking the filter about the PM bases in front of the known ones, then the PM
        // before those ... (fl.bs = how many strings back; the k-mer's first PM bases at the latest): the string that fails rules out every k-mer
        // that holds it and t0 moves behind them all (probe_fail), instead of one end at a time -- a k-mer holds k-PM+1 k-mer ends' worth of such
        // steps (63-mers: 18 misses per sequencing error in a deferred strand's stretch, measured).  All of them occur: the next end, as before.
        auto kf_miss = [&]() {
            // (k < 2 PM: the one string in front overlaps the known one and settles little -- measured slower at k = 31.  fl.bs counts strings in three bits:
            //  a k-mer of more than 7 strings -- a user-set cbf_m below k / 7 -- is not scanned: the counter would wrap to 0 = "no scan" in mid-scan)
            if (FIN_W_BACKSCAN && ix.fbf && k >= 2 * PM && (k + PM - 1) / PM <= 7 && !fl.bs_off) { fl.bs = 1; pc = W_PROBE0; }
            else { t0++; pe++; pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (pe & kf_every) == 0 ? (uint32_t)W_PROBE0 : (uint32_t)W_KF0; }
        };
        // (the k-mer that is looked up: its first 32 bases -- all of them, k <= 32 -- in pcode, the rest (k >= 33) in il | ir << 32.  While it is looked up pp holds the
        //  k-mer's tag: the hash -- two 64-bit finalisers for a two-word key -- is made once, for the address of the chain's first bucket; a further bucket
        //  is the one behind the bucket that has just arrived, whose address q_aux still holds)
        auto kt3_addr = [&]() -> const char* {
            const uint64_t h = kt_wide ? fin_mix64(pcode) : fin_kt3_hash(pcode, LONGK ? ((uint64_t)il | ((uint64_t)ir << 32)) : 0ull);   // (wide: pcode holds the folded words)
            pp = (int)((uint32_t)h & FIN_KT3_TAGMASK);
            return (const char*)(ix.kt3 + fin_kt3_bucket(h, ix.kt3_buckets));
        };
        // Behind a miss the next end's k-mer is the old one shifted by a base: both key words roll, and the one new base is in the chunk cache nearly always --
        // instead of W_KF0 and W_KF0B making the two words again from up to three chunks of which the cache holds two (every look-up of a run reloaded one:
        // k63_repeats' walk kernel began as many epochs in those two states as in W_KF1).  Not across a bad position (W_KF0 places that string itself).
        auto kf_roll2 = [&]() {
            if (!LONGK || kt_wide || pc != W_KF0 || bridging || (q & Q_AUX)) return;   // (k >= 33: the kernel for k <= 32 has no register to spare for it; k >= 64: no key is kept that could roll)
            const int ci = (int)(t0 >> 5); const uint32_t j = t0 & 31u;
            uint32_t b = 4u;
            if (ci == ck.cur && !(q & Q_CURCHUNK)) { if ((ck.bvalid >> j) & 1u) b = (uint32_t)(ck.bcodes >> (2u * j)) & 3u; }
            else if (ci == ck.nxt && !(q & Q_NEXTCHUNK)) { if ((ck.nvalid >> j) & 1u) b = (uint32_t)(ck.ncodes >> (2u * j)) & 3u; }
            if (b == 4u) return;   // (not at hand, or no base: W_KF0 next epoch)
            const uint32_t n2 = (uint32_t)k - 32u;
            uint64_t k1w = (uint64_t)il | ((uint64_t)ir << 32);
            if (n2) { pcode = (pcode >> 2) | ((k1w & 3ull) << 62); k1w = (k1w >> 2) | ((uint64_t)b << (2u * (n2 - 1u))); }
            else pcode = (pcode >> 2) | ((uint64_t)b << 62);
            il = (uint32_t)k1w; ir = (uint32_t)(k1w >> 32);
            q_aux = (const void*)kt3_addr(); q |= Q_AUX | Q_AUX2; pc = W_KF1;
        };
        if (pc == W_KFV) {   // wt, aux = the two text windows the claimed place straddles
            uint64_t x0, x1;
            fin_text_kmer(wt, aux, (res_g - (uint32_t)(k - 1)) & 63u, (uint32_t)k, x0, x1);
            if (x0 != pcode || (LONGK && x1 != ((uint64_t)il | ((uint64_t)ir << 32)))) {
                WDBG(13);
#ifdef FIN_W_DEBUG
                atomicAdd(&g_fin_kfv[(res_g - (uint32_t)(k - 1)) & 63u], 1ull);
                if (x0 != pcode) atomicAdd(&g_fin_kfv[64], 1ull);
                if (LONGK && x1 != ((uint64_t)il | ((uint64_t)ir << 32))) atomicAdd(&g_fin_kfv[65], 1ull);
                (void)0;
                atomicAdd(&g_fin_kfv[68 + ((uint32_t)__popcll(x0 ^ pcode) > 4u ? 1 : 0)], 1ull);
#endif
                give_up = true; pc = W_ITEM0;
            }
            else { a_dl = 0u; q_aux = (const void*)(ix.samp + ((res_g - (uint32_t)(k - 1)) >> ix.samp_shift)); q |= Q_AUX; pc = W_RES4; }
        }
        if (pc == W_KFX) {   // aux = {k0, k1} of slot pp of the exact side table's chain, wt = its {g, claim}: the k-mers whose answer is unverified, whole
            const uint64_t s0 = aux.x | ((uint64_t)aux.y << 32), s1 = aux.z | ((uint64_t)aux.w << 32);
            const uint64_t k1w = LONGK ? ((uint64_t)il | ((uint64_t)ir << 32)) : 0ull;
            if (wt.y == 0xFFFFFFFFu) { give_up = true; pc = W_ITEM0; }   // not there: the claim was another k-mer's (or the upload's list overran): kernel 3 decides
            else if (s0 == pcode && s1 == k1w) {
                // present, and the reference reports it at wt.x, a place that does not spell it: an anchor as round 4's unverified slot was -- it taints
                end = (int)t0; bridging = false; a_dl = 0u; fl.bs_off = 0; fl.tainted = 1;
                res_g = wt.x;
                const uint32_t gs = res_g - (uint32_t)(k - 1);
                if (gs < ix.total_len) { q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = W_RES4; }
                else { give_up = true; pc = W_ITEM0; }   // (no answer: unreachable on a consistent index -- kernel 3 reports it as the reference's restatement does)
            } else {   // (pp counts this chain's slots here)
                pp++;
                const uint64_t h = fin_kt3_hash(pcode, k1w);
                q_aux = (const void*)(ix.ktx + (((uint32_t)(h >> 32) + (uint32_t)pp) & ((1u << ix.ktx_log2) - 1u))); q |= Q_AUX | Q_AUX2;
            }
        }
        // (a look-up fetches a whole bucket -- four slots, 32 bytes -- per epoch: the table is 55 % full, an absent k-mer's chain ends in its first bucket three
        //  times in four, and a repeat-rich read asks about a hundred absent k-mers one epoch each -- W_KF1 was 60 % of chr1_repeats' lane-epochs in round 4)
        if (pc == W_KF1) {   // aux, wt = the bucket's four slots {g, tag | flags}; pcode (il | ir << 32) = the k-mer, pp = buckets looked at so far
            const uint32_t tag = (uint32_t)pp;
            uint32_t verdict = 0, hit_g = 0;   // 0 = the bucket is full of other k-mers (the next one), 1 = a claim, 2 = an unverified claim, 3 = the chain ends: absent
            auto slot = [&](uint32_t g_, uint32_t m_) {   // (in slot order: the first empty slot ends the chain, the first matching tag is the claim)
                if (verdict == 0u) {
                    if (m_ == 0xFFFFFFFFu) verdict = 3u;
                    else if ((m_ & FIN_KT3_TAGMASK) == tag) { verdict = (m_ & FIN_KT3_UNVER) ? 2u : 1u; hit_g = g_; }
                }
            };
            slot(aux.x, aux.y); slot(aux.z, aux.w); slot(wt.x, wt.y); slot(wt.z, wt.w);
            if (LEAN && verdict != 3u && verdict != 0u) kb_idx = NONE;   // (a claim: the second look-up of the epoch is dropped)
            if (verdict == 1u) {
                // the table claims the k-mer, with the reference's answer for it -- a place where the text spells it: an anchor like any other once the text there
                // has borne the claim out.  The window(s) of the place are asked for now; one window: the locate's first load goes out beside it and W_RES4
                // compares (no epoch more than round 4's exact keys took); the place straddles two windows: W_KFV compares, then the locate
                WDBG(8);
                end = (int)t0; bridging = false; fl.bs_off = 0;
                if (!LONGK) a_dl = 0u;
                res_g = hit_g;   // (t0's register: t0 has done its duty)
                const uint32_t gs = res_g - (uint32_t)(k - 1);
                if (kt_wide && gs < ix.total_len && res_g < ix.total_len) {
                    if (ix.rcwin) fl.tainted = 1;
                    a_dl = 0u; fl.kt_claim = 1; q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = W_RES4;   // locate, then compare (W_RES5 -> W_REANCH)
                } else
                if (gs < ix.total_len && res_g < ix.total_len) {
                    if (ix.rcwin) fl.tainted = 1;   // (as round 4's tables: on an index with reverse-complement pairs every whole-k-mer anchor taints)
                    ttag = gs >> 6; q |= Q_TEXT;
                    if ((gs & 63u) + (uint32_t)k <= 64u) { fl.kt_claim = 1; q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = W_RES4; }
                    else { q_aux = (const void*)(ix.concat + ((size_t)(ttag + 1u) << 2)); q |= Q_AUX; pc = W_KFV; }
                } else { give_up = true; pc = W_ITEM0; }   // (no place: a false claim)
            } else if (verdict == 2u) {
                // a k-mer with this tag is in the index and the reference reports it at a place that does not spell it (duplicated k-mers): nothing to compare
                // the read's k-mer with -- the exact side table holds such k-mers whole (none: the upload found no such k-mer, a shared tag -- kernel 3 decides)
                if (ix.ktx && !kt_wide) {   // (the side table holds two key words: a wide key's unverified claim goes to the plain kernel)
                    const uint64_t h = fin_kt3_hash(pcode, LONGK ? ((uint64_t)il | ((uint64_t)ir << 32)) : 0ull);
                    pp = 0; q_aux = (const void*)(ix.ktx + ((uint32_t)(h >> 32) & ((1u << ix.ktx_log2) - 1u))); q |= Q_AUX | Q_AUX2; pc = W_KFX;
                }
                else { give_up = true; pc = W_ITEM0; }
            } else if (verdict == 3u) {
                // not there.  The next end is asked directly -- a short probe would pass again in this stretch --, every eighth one is probed first: a failing
                // probe settles k-PM+1 ends at once (k >= 40 under lean tables: a back-scan, kf_miss)
                WDBG(9);
                bool two = false;
                if (LEAN && kb_idx != NONE) {   // the next end's look-up went out with this one: a clean miss too?
                    uint32_t v2 = 0;
                    auto slot2 = [&](uint32_t m_) { if (v2 == 0u) { if (m_ == 0xFFFFFFFFu) v2 = 3u; else if ((m_ & FIN_KT3_TAGMASK) == tag2) v2 = 1u; } };
                    slot2(kb0.y); slot2(kb0.w); slot2(kb1.y); slot2(kb1.w);
                    two = v2 == 3u;   // (a claim, or a bucket full of other k-mers: that end is looked up again in its own epoch)
                    kb_idx = NONE;
                }
                kf_miss();
                if (two && pc == W_KF0) { WDBG(9); kf_miss(); }   // (the first miss leads straight to the next end's look-up -- the issue made sure of it: that one missed as well)
                kf_roll2();
            } else {   // the next bucket of the chain: the one behind the bucket that has just arrived (q_aux is its address still; the table wraps)
                WDBG(12);
                const char* nb = (const char*)q_aux + sizeof(FinKt3Bucket);
                if (nb == (const char*)(ix.kt3 + ix.kt3_buckets)) nb = (const char*)ix.kt3;
                q_aux = (const void*)nb; q |= Q_AUX | Q_AUX2;
            }
        }
        if ((pc == W_PROBE0 || pc == W_KF0) && t0 > t_stop) pc = W_ITEM0;   // (a deferred strand's item: its stretch is done -- a walk may have carried it past the end)
        if (pc == W_PROBE0 || pc == W_KF0) {
            const bool kf = pc == W_KF0;   // the string is the whole k-mer, for the k-mer table (k <= 31)
            int p = (int)t0 - ((pfull || kf) ? k : PM) + 1;
            if (!kf && fl.bs) p = max((int)t0 - k + 1, (int)t0 - ((int)fl.bs + 1) * PM + 1);   // (the PM bases fl.bs strings in front of the known ones; the k-mer's first at the latest)
            if (bridging && p > (int)br_E) p = (int)br_E;   // across a bad position the string is pulled back so that it contains it ...
            if (bridging && !ptried && PT > 0) {   // ... and is placed so that the table key contains the bad position E:
                if (!LONGK && (int)t0 >= (int)br_E + PT - 1) p = (int)br_E;          // it starts AT E as soon as a key fits between E and t0 (a failure then settles everything up to E+k-1; k <= 32: for longer k the rule changes nothing, CHANGELOG.md 5.6),
                else if (p < (int)br_E - (PT - 1)) p = (int)br_E - (PT - 1);         // and T-1 bases before E at the earliest
            }
            // ... and goes on to t0 as long as it matches, 32 bases at most: a string that starts at a bad position and still matches that
            // far says the read has left this place for another (an indel, a chimera), not that one base is wrong -- going on base by
            // base from a stale alignment would cost k probes of k bases; the whole k-mer at t0 is looked up instead (probe_pass)
            const int last = bridging ? min((int)t0, p + 31) : (int)t0;
            const int ci0 = p >> 5, ci1 = min(last, p + 31) >> 5;   // (the chunks of its first 32 bases)
            if (ck.need2(ci0, ci1, strand_chunks, q, q_aux)) {
                uint64_t w; uint32_t v;
                ck.window(p, ci0, ci1, w, v);
                const uint32_t inv = ~v;
                pfi = inv ? (uint32_t)(__ffs((int)inv) - 1) : 32u;
                pcode = w; pp = p;   // (plim_now() == last)
                if (kf && LONGK) {   // k >= 33: the first 32 bases are here, the rest next (W_KF0B)
                    if (pfi < 32u) {   // a non-ACGT base: no k-mer contains it
                        t0++; pe++;
                        pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (pe & kf_every) == 0 ? (uint32_t)W_PROBE0 : (uint32_t)W_KF0;
                    } else { pc = W_KF0B; ir = 1u; }   // (k >= 64: the next word's number)
                } else
                if (kf) {   // (k <= 32)
                    if (pfi < (uint32_t)k) {   // a non-ACGT base: no k-mer contains it
                        t0++; pe++;
                        pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (pe & kf_every) == 0 ? (uint32_t)W_PROBE0 : (uint32_t)W_KF0;
                    } else if (!(q & Q_AUX)) {
                        pcode = k >= 32 ? w : w & ((1ull << (2 * k)) - 1ull);
                        q_aux = (const void*)kt3_addr(); q |= Q_AUX | Q_AUX2; pc = W_KF1;
                        if (LEAN) {
                            // the next end's k-mer -- bases p+1 .. p+k, inside the window's 32 bases for k <= 31 -- in the same epoch, when that end would be asked
                            // directly anyway (inside the item's stretch, not the eighth of a run: that one is probed first)
                            kb_idx = NONE;
                            if (k < 32 && pfi > (uint32_t)k && t0 + 1u <= t_stop && ((pe + 1) & kf_every) != 0 && !bridging) {
                                const uint64_t h2 = fin_kt3_hash((w >> 2) & ((1ull << (2 * k)) - 1ull), 0ull);
                                tag2 = (uint32_t)h2 & FIN_KT3_TAGMASK; kb_idx = fin_kt3_bucket(h2, ix.kt3_buckets); q |= Q_KB;
                            }
                        }
                    }
                } else
                if (ix.fbf && !pfull) {   // lean tables: the string's first m bases in the directional string filter (one 16-byte load)
                    if (pfi < ix.cbf_m) { fl.bs = 0; probe_fail(); }
                    else if (!(q & Q_AUX)) {
                        const uint32_t m = ix.cbf_m;
                        const uint64_t h = fin_cbf_hash(w & (m >= 32u ? ~0ull : ((1ull << (2u * m)) - 1ull)));
                        q_aux = (const void*)(ix.fbf + ((h >> 35) & ((1ull << ix.cbf_log2) - 1ull))); q |= Q_AUX; pc = W_PROBEF;
                    }
                } else
                if (LEAN) { give_up = true; pc = W_ITEM0; }   // (unreachable: under lean tables every probe string is asked of the filter, every whole k-mer of the table)
                else if (PT > 0) {
                    if (pfi < (uint32_t)PT) probe_fail();
                    else {
                        const uint32_t key = (uint32_t)w & ((1u << (2 * PT)) - 1u);
                        q_aux = (const void*)(ix.ptab + key); q |= Q_AUX; pc = W_PROBE1;
                    }
                } else { il = 0; ir = n - 1; pe = p; pc = W_PROBEX; }
            }
        }
        // (this block stands BEHIND the W_KF0 block: a look-up whose first word W_KF0 has just made goes on here in the same epoch -- the second word's chunk is
        //  nearly always in the cache -- instead of the next one: a two-word look-up is one epoch like a one-word one, not two; k63_repeats' walk kernel spent
        //  as many lane-epochs in this state as in W_KF1)
        if (pc == W_KF0B && kt_wide) {   // k >= 64: word ir of the key (ir = 1 .. ceil(k/32) - 1; pcode = the words so far, folded: fin_prepass.hip look_ktabN_at)
            const int wj = (int)ir, p2 = (int)t0 - k + 1 + 32 * wj, n2 = min(32, k - 32 * wj);
            const int ci0 = p2 >> 5, ci1 = (p2 + n2 - 1) >> 5;
            if (ck.need2(ci0, ci1, strand_chunks, q, q_aux)) {
                uint64_t w; uint32_t v;
                ck.window(p2, ci0, ci1, w, v);
                const uint32_t needv = n2 >= 32 ? 0xFFFFFFFFu : (1u << n2) - 1u;
                if ((v & needv) != needv) {   // a non-ACGT base: no k-mer contains it
                    t0++; pe++;
                    pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (pe & kf_every) == 0 ? (uint32_t)W_PROBE0 : (uint32_t)W_KF0;
                } else {
                    const uint64_t word = n2 >= 32 ? w : w & ((1ull << (2 * n2)) - 1ull);
                    const bool last = 32 * (wj + 1) >= k;
                    if (!last) { pcode = fin_kt3_fold(pcode, word, (uint32_t)wj); ir = (uint32_t)wj + 1u; }
                    else if (!(q & Q_AUX)) { pcode = fin_kt3_fold(pcode, word, (uint32_t)wj); q_aux = (const void*)kt3_addr(); q |= Q_AUX | Q_AUX2; pc = W_KF1; }
                }
            }
        } else
        if (pc == W_KF0B) {   // two-word keys, k > 32: the k-mer's bases 32 .. k-1 (its first 32 are in pcode)
            const int p2 = (int)t0 - k + 1 + 32, n2 = k - 32;
            const int ci0 = p2 >> 5, ci1 = (p2 + n2 - 1) >> 5;
            if (ck.need2(ci0, ci1, strand_chunks, q, q_aux)) {
                uint64_t w; uint32_t v;
                ck.window(p2, ci0, ci1, w, v);
                const uint32_t needv = (1u << n2) - 1u;   // (n2 <= 31)
                if ((v & needv) != needv) {   // a non-ACGT base: no k-mer contains it
                    t0++; pe++;
                    pc = t0 > t_stop ? (uint32_t)W_ITEM0 : (pe & kf_every) == 0 ? (uint32_t)W_PROBE0 : (uint32_t)W_KF0;
                } else if (!(q & Q_AUX)) {
                    const uint64_t k1w = w & ((1ull << (2 * n2)) - 1ull);
                    il = (uint32_t)k1w; ir = (uint32_t)(k1w >> 32);
                    q_aux = (const void*)kt3_addr(); q |= Q_AUX | Q_AUX2; pc = W_KF1;
                }
            }
        }
        // ---- a new item (these blocks come last: a state that has just asked for data must not run on this epoch's `aux`) ----
        if (pc == W_DESC) {   // descriptor arrived
            r_pk = aux.x; r_len = aux.z; r_out = aux.w;
            ck.reset(); run_len = 0; w_next = 0; hull = 0x0000FFFFu;
            budget = r_len > 0x3FFFF00u ? 0xFFFFFFFFu : (ix.budget_mult >> 1) * r_len + ix.budget_add;
            kb_idx = NONE;
            fl.bounded = 0; fl.tainted = 0; fl.tabent = 0; fl.bs = 0; fl.bs_off = 0; fl.kt_claim = 0;   // (tabent: the anchor being resolved is a whole k-mer's entry of the anchor table)
            if (a_colex == NONE) { WDBG(5); t0 = (uint32_t)end; if (a_dl) { fl.bounded = 1; hull = a_dl - 1u; } pc = W_PROBE0; }   // probe item: `end` is its first unresolved k-mer end (a deferred strand's: a_dl - 1 its last)
            else if (a_dl == FIN_PLACE_MARK) {
                // the pre-pass's look found the k-mer that ends at `end` in the k-mer table, with its verified answer (a_colex): an anchor like a
                // k-mer-table hit of this kernel (W_KF1): the unitig of the place, then the run and the walk
                bridging = false; a_dl = 0u; res_g = a_colex;   // (the look's claim was compared with the text by the pre-pass, fin_prepass.hip)
                if (ix.rcwin) fl.tainted = 1;   // (as a whole-k-mer anchor of this kernel)
                const uint32_t gs = res_g - (uint32_t)(k - 1);
                if (gs < ix.total_len) { q_aux = (const void*)(ix.samp + (gs >> ix.samp_shift)); q |= Q_AUX; pc = W_RES4; }
                else { give_up = true; pc = W_ITEM0; }   // (not a text place: kernel 3 searches the read)
            }
            else if (a_dl == FIN_SEED_MARK) { WDBG(6); bridging = true; a_dl = 0u; q_aux = (const void*)(ix.pos + a_colex); q |= Q_AUX; pc = W_RES3; }   // seed item: node -> pos[node]
            else {
                WDBG(7);
                const bool ub = (a_dl >> 31) != 0u;
                q_aux = (const void*)((const char*)(ix.blkinfo + (a_colex >> 6)) + (ub ? 8 : 0)); q |= Q_AUX; pc = W_RES1;
            }
        }
        if (pc == W_ITEM1) {   // item arrived
            who = aux.x; end = (int)aux.y; a_colex = aux.z; a_dl = aux.w;
            bridging = false; pfull = false; ptried = false; pguessed = false;
            if (aux.x == FIN_Q_EMPTY && aux.y == FIN_Q_EMPTY) pc = W_ITEM0;   // a slot its producer reserved and did not use
            else { q_aux = (const void*)(desc + (who & FIN_WHO_READ)); q |= Q_AUX; pc = W_DESC; }
        }
        // exit condition every lane reaches: an item that runs out of epochs sends its read to kernel 3
        if (pc > W_DESC) {
            if (budget == 0) {
                if (q & Q_TEXT) ttag = NONE;
                if (!LEAN) rc.drop(q);
                WDBG(14);
                q = 0; if (!pend) run_len = 0; give_up = true; pc = W_ITEM0;
            } else budget--;
        }
        // an item that ended in this epoch -- by whichever path -- and writes its strand's gaps: everything behind the last run is absent,
        // or left to the kernels behind (hand-over, give-up), which only write pairs
        bool to_sister = false;
        if (pc == W_ITEM0 && pc0 > W_DESC && (who & FIN_WHO_GAPS)) {
            const uint32_t nk_ = r_len - (uint32_t)(k - 1);
            if (!pend) { run_len = 0; run_pos = w_next; gap0 = 0; }
            gap1 = nk_ - w_next;
            // the deferred sister strand: this lane searches it next, inside the stretch of this strand's slots left open (below, once
            // the last run is written).  Nothing open: nothing to search.  (a read given up goes to kernel 3 whole, which searches a
            // deferred strand from its first k-mer)
            if ((who & FIN_WHO_DEFER) && !give_up) {
                if (fl.tainted) hull = (nk_ - 1u) << 16;   // (every slot counts as open)
                else if (gap1) hull_add(w_next, nk_ - 1u);
                to_sister = (hull & 0xFFFFu) <= (hull >> 16);
            }
            w_next = nk_;
            pend = pend || gap1 != 0u;
        }
        // ================= 3. hand-over (wave-wide, converged) =================
        fin_wq_push(oq, emit, emit_item, items_out, n_out, lane);
        fin_wq_push(lq, give_up, who & FIN_WHO_READ, list, n_list, lane);
        // ================= 4. cooperative write-out of finished runs (and of the absent slots around them) =================
        {
            uint64_t m = __ballot(pend);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                // (v_readlane into scalar registers: `src` is the same in every lane -- a shuffle through the LDS crossbar cost nine LDS
                //  instructions and their wait per pending lane)
                auto lane_of = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
                const uint32_t o_base = lane_of(r_out), o_nk = lane_of(r_len) - (uint32_t)(k - 1);
                const uint32_t p_g0 = lane_of(gap0), p_len = lane_of(run_len), p_g1 = lane_of(gap1);
                const uint32_t p_pos = lane_of(run_pos) - p_g0;   // first slot of the region: gap, run, gap
                const uint32_t p_u = lane_of(w_u), p_off = lane_of(run_off);
                const uint32_t p_who = lane_of(who);
                const bool p_rev = (p_who >> 31) != 0u, p_cas = (p_who & 0x40000000u) != 0u;
                const uint32_t total = p_g0 + p_len + p_g1;
                if (!p_cas) {
                    for (uint32_t i = lane; i < total; i += 64) {
                        const uint32_t idx = p_rev ? (o_nk - 1 - (p_pos + i)) : (p_pos + i);
                        const bool pair = i - p_g0 < p_len;   // (unsigned: false in front of the run too)
                        // (nontemporal: the pairs are written once and nobody reads them before the step ends -- chr1 search 7.05 -> 6.64 ms)
                        const int2 val = pair ? make_int2((int)p_u, (int)(p_off + i - p_g0)) : make_int2(-1, -1);
                        __builtin_nontemporal_store(*(const unsigned long long*)&val, (unsigned long long*)&out[(size_t)o_base + idx]);
                    }
                } else {
                    // The reverse strand of a read whose forward strand is searched too (possibly at this moment, by another lane): the
                    // merge rule lets the forward pair win (search_fmin.hh:54-60).  The forward strand stores plainly; this one only
                    // fills slots that still hold the prefilled (-1,-1), atomically: whichever comes first, the forward pair stays.
                    // (such a read's slots are prefilled: p_g0 = p_g1 = 0)
                    for (uint32_t i = lane; i < p_len; i += 64) {
                        const uint32_t idx = p_rev ? (o_nk - 1 - (p_pos + i)) : (p_pos + i);
                        const unsigned long long v = (unsigned long long)p_u | ((unsigned long long)(p_off + i) << 32);
                        (void)atomicCAS((unsigned long long*)&out[(size_t)o_base + idx], 0xFFFFFFFFFFFFFFFFull, v);
                    }
                }
            }
            if (pend) { run_len = 0; gap0 = 0; gap1 = 0; pend = false; }
        }
        // the sister strand of a finished one (FIN_WHO_DEFER): a probe item in this lane's own registers -- the read's descriptor is here
        // and its chunks lie beside this strand's.  Slot s of this strand is the sister's k-mer end r_len - 1 - s: the stretch [lo, hi] of
        // open slots is its ends r_len - 1 - hi .. r_len - 1 - lo.  Its pairs only fill: the absent slots are written (FIN_WHO_GAPS above).
        if (to_sister) {
            const uint32_t lo = hull & 0xFFFFu, hi = hull >> 16;
#ifdef FIN_W_DEBUG
            atomicAdd(&g_fin_witem[91], 1ull); atomicAdd(&g_fin_witem[92], (unsigned long long)(hi - lo + 1u)); atomicAdd(&g_fin_witem[93], (unsigned long long)(r_len - (uint32_t)(k - 1)));
#endif
            const uint32_t b_rev = (who >> 31) ^ 1u;
            who = (who & FIN_WHO_READ) | (b_rev << 31) | (b_rev ? 0x40000000u : 0u);
            t0 = r_len - 1u - hi; hull = r_len - 1u - lo; fl.bounded = 1;
            a_colex = NONE; a_dl = 0u;
            bridging = false; pfull = false; ptried = false; pguessed = false; fl.bs = 0; fl.bs_off = 0; fl.kt_claim = 0;
            ck.reset(); w_next = 0; fl.n_sister++;
            pc = W_PROBE0;
        }
        // ================= 5. work queue (FinWorkRanges) =================
#ifdef FIN_W_DEBUG
        if (pc == W_ITEM0 && dbg_ep) {
            atomicAdd(&g_fin_witem[31 - __clz((int)dbg_ep)], 1ull); atomicMax(&g_fin_witem[40], (unsigned long long)dbg_ep); atomicAdd(&g_fin_witem[41], (unsigned long long)dbg_ep);
            dbg_ep = 0;
        }
#endif
        {
            uint32_t id = 0;
            const int wk = wr.take(pc == W_ITEM0, lane, n_items, work_counter, id);
            if (wk == 1) { q_aux = (const void*)(items_in + id); q |= Q_AUX; pc = W_ITEM1; }
            else if (wk == 2) pc = W_DONE;
        }
        if (!__any(pc != W_DONE)) break;
    }
    fin_wq_flush(oq, make_uint4(FIN_Q_EMPTY, FIN_Q_EMPTY, FIN_Q_EMPTY, FIN_Q_EMPTY), items_out, lane);
    fin_wq_flush(lq, (uint32_t)FIN_Q_EMPTY, list, lane);
#ifdef FIN_W_DEBUG
    if (lane == 0 && dbg_wave) { atomicAdd(&g_fin_witem[48 + 31 - __clz((int)dbg_wave)], 1ull); atomicMax(&g_fin_witem[88], (unsigned long long)dbg_wave); }
    if (lane == 0 && n_items > 100000u) {
        const uint32_t wv = (blockIdx.x * FIN_TPB + threadIdx.x) >> 6;
        if (wv < 8192u) { g_fin_wtime[wv] = (unsigned long long)wall_clock64() - g_fin_wtime[8192]; atomicAdd(&g_fin_wtime[8193], 1ull); }
    }
#endif
    {
        uint32_t ns = fl.n_sister;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ns += (uint32_t)__shfl_xor((int)ns, d);
        if (lane == 0 && ns) atomicAdd(n_sister_out, ns);
    }
#undef t_stop
#undef pend
#undef bridging
#undef pfull
#undef ptried
#undef pguessed
}

__global__ __launch_bounds__(FIN_TPB, FIN_WALK_MINWAVES) void fin_walk_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                           const uint4* items_in, const uint32_t* n_in, uint4* items_out, uint32_t* n_out,
                                                           uint32_t* list, uint32_t* n_list, int last_round, uint32_t* work_counter, uint32_t* n_sister_out) {
    fin_walk_body<false>(ix, packed, desc, out, items_in, n_in, items_out, n_out, list, n_list, last_round, work_counter, n_sister_out);
}
__global__ __launch_bounds__(FIN_TPB, FIN_WALK_MINWAVES) void fin_walk_lean_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                           const uint4* items_in, const uint32_t* n_in, uint4* items_out, uint32_t* n_out,
                                                           uint32_t* list, uint32_t* n_list, int last_round, uint32_t* work_counter, uint32_t* n_sister_out) {
    fin_walk_body<false, true>(ix, packed, desc, out, items_in, n_in, items_out, n_out, list, n_list, last_round, work_counter, n_sister_out);
}
__global__ __launch_bounds__(FIN_TPB, FIN_WALK_MINWAVES) void fin_walk_long_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                                const uint4* items_in, const uint32_t* n_in, uint4* items_out, uint32_t* n_out,
                                                                uint32_t* list, uint32_t* n_list, int last_round, uint32_t* work_counter, uint32_t* n_sister_out) {
    fin_walk_body<true>(ix, packed, desc, out, items_in, n_in, items_out, n_out, list, n_list, last_round, work_counter, n_sister_out);
}
// ... and the lean instantiation for 33 <= k <= 63 (no second look-up there: the key has two words; what it gains is the registers of the states it drops)
__global__ __launch_bounds__(FIN_TPB, FIN_WALK_MINWAVES) void fin_walk_long_lean_kernel(FinDevIndex ix, const uint4* packed, const FinReadDesc* desc, int2* out,
                                                                const uint4* items_in, const uint32_t* n_in, uint4* items_out, uint32_t* n_out,
                                                                uint32_t* list, uint32_t* n_list, int last_round, uint32_t* work_counter, uint32_t* n_sister_out) {
    fin_walk_body<true, true>(ix, packed, desc, out, items_in, n_in, items_out, n_out, list, n_list, last_round, work_counter, n_sister_out);
}

// ---- host side: one step of kernel 4 ------------------------------------------------------------------------------------------
extern "C" int fin_walk_blocks_per_cu(void) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fin_walk_kernel, FIN_TPB, 0) != hipSuccess || nb < 1) nb = 4;
    return nb;
}
// 1: with this index and these buffers the pipeline can do without a prefilled output (every strand's first item is the walk kernel's)
extern "C" int fin_v4_writes_gaps(const FinDevIndex* ix, const uint32_t* seed) { return (ix->pos != nullptr || (ix->kt3 != nullptr && ix->fbf != nullptr)) && seed != nullptr; }
extern "C" uint32_t fin_v4_counter_words(void) { return 4u * FIN_V4_ROUNDS + 16u; }
extern "C" uint32_t fin_v4_max_rounds(void) { return (uint32_t)FIN_V4_ROUNDS; }

// Queue capacity (slots): a queue holds at most one item per strand plus the slots its producing waves reserved and did not use (64
// per wave of the largest grid) -- fin_v4_queue_slots.  Kernel 3's list is appended to by every one of the FIN_V4_ROUNDS walk launches
// (a give-up may come in any round, each launch may strand 63 slots per wave) and is never reset in between: it has that slack
// FIN_V4_ROUNDS times -- fin_v4_list_slots (the same count bounds the plain kernel's list when the walk kernel feeds it: k > 128).
// ws: 3 item queues of fin_v4_queue_slots uint4, then kernel 3's list of fin_v4_list_slots u32 (+ 64 bytes: list entries are fetched with
// 16-byte loads).  ctr: fin_v4_counter_words() u32, zeroed here.
extern "C" uint64_t fin_v4_queue_slots(uint32_t n_reads, uint32_t max_grid_blocks) { return 2ull * n_reads + 64ull * (FIN_TPB / 64) * max_grid_blocks + 128; }   // (an item per strand)
extern "C" uint64_t fin_v4_list_slots(uint32_t n_reads, uint32_t max_grid_blocks) { return 2ull * n_reads + (uint64_t)FIN_V4_ROUNDS * 64ull * (FIN_TPB / 64) * max_grid_blocks + 128; }
extern "C" uint64_t fin_v4_workspace_bytes(uint32_t n_reads, uint32_t max_grid_blocks) { return fin_v4_queue_slots(n_reads, max_grid_blocks) * 3 * 16 + fin_v4_list_slots(n_reads, max_grid_blocks) * 4 + 64; }
extern "C" int fin_launch_search_v4(const FinDevIndex* ix, const uint8_t* bases, const void* packed, const FinReadDesc* desc,
                                    const uint64_t* offs, const uint64_t* out_offs, void* out, uint64_t n_kmers, uint32_t n_reads,
                                    int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                                    uint64_t* ovf_scratch, uint32_t ovf_blocks, uint32_t* pass, uint32_t* seed, void* ws, uint64_t q_slots, uint32_t* ctr,
                                    uint32_t grid_probe, uint32_t grid_stream, uint32_t grid_walk, uint32_t grid_v3,
                                    hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev_mid, hipEvent_t out_ready, int no_prefill, uint32_t rounds) {
    if (n_reads == 0) return 0;
    // no_prefill (the caller checked fin_v4_writes_gaps): no (-1,-1) pass over the output -- every first item goes to the walk kernel,
    // whose lanes write the absent slots of their strands with the pairs, and the route kernel fills the reads nobody searches
    if (no_prefill && !fin_v4_writes_gaps(ix, seed)) return (int)hipErrorInvalidValue;
    // k > 128 (FIN_FAST_K): the streaming kernels read 7-bit LCS values and cannot be used.  With a seed table the walk kernel needs none
    // of them: one round, and whatever it hands on or gives up goes straight to the plain kernel's list (ovf_list: FinWaveQueue slots, so
    // that list needs the capacity of a queue) instead of the stream kernel / kernel 3.  Without a seed table the caller uses kernel 0.
    const bool longk = ix->k > FIN_FAST_K;
    if (longk && !fin_v4_writes_gaps(ix, seed)) return (int)hipErrorInvalidValue;
    // rounds: how many stream / walk rounds to launch (1 .. FIN_V4_ROUNDS; the caller's guess from this batch's previous runs -- whatever is
    // left after the last one goes to kernel 3's list, so any number is exact; an empty round still costs two launches)
    const uint32_t R = longk ? 1u : (rounds < 1u ? 1u : rounds > (uint32_t)FIN_V4_ROUNDS ? (uint32_t)FIN_V4_ROUNDS : rounds);
    hipError_t e = hipMemsetAsync(ovf_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(ctr, 0, fin_v4_counter_words() * sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    if (!out_ready && !no_prefill) {   // every slot (-1,-1); runs overwrite.  (out_ready: the caller does that on another stream and this event says when it is done)
        e = hipMemsetAsync(out, 0xFF, n_kmers * 8, stream);
        if (e != hipSuccess) return (int)e;
    }
    if (ev0) (void)hipEventRecord(ev0, stream);
    // counters: [0] probe work, [1] kernel-3 work, [2] list count, [3] unused, then per round r: [4+4r] stream work, [5+4r] walk work,
    //           [6+4r] stream items of round r, [7+4r] anchor items of round r   (stream items of round R land in [6+4R])
    //           [4*FIN_V4_ROUNDS+8] deferred strands the walk kernel's lanes went on with (a statistic)
    //           [4*FIN_V4_ROUNDS+9] reads the pre-pass's fast path finished (a statistic)
    uint32_t* const wc_probe = ctr, *const wc_v3 = ctr + 1, *const n_list = ctr + 2;
    uint4* const sq0 = (uint4*)ws, *const sq1 = sq0 + q_slots, *const aq = sq1 + q_slots;
    uint32_t* const list = (uint32_t*)(aq + q_slots);
    if (!ix->pos && !(ix->kt3 && ix->fbf)) seed = nullptr;
    // (the fast path of the pair pre-pass writes the reads it finishes itself -- only when nothing prefills the output behind it)
    int rc = fin_launch_probe_stage(ix, packed, desc, n_reads, strands, pass, seed, wc_probe, grid_probe, (no_prefill && ix->fast_path) ? out : nullptr, ctr + 4 * FIN_V4_ROUNDS + 9, stream);
    if (rc) return rc;
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    {
        const uint32_t need = (n_reads + FIN_TPB - 1) / FIN_TPB;
        // lean tables: a seed is a PLACE, and only the pair pre-pass makes those (its looks' k-mer-table slots); the probe kernel's seeds are nodes,
        // which nothing can turn into places without the anchor table: every strand it leaves becomes a probe item
        const bool lean = ix->pos == nullptr && ix->fbf != nullptr;
        const uint32_t* const route_seed = (lean && !(strands == 1 && ix->defer_ok)) ? nullptr : seed;
        hipLaunchKernelGGL(fin_route_kernel, dim3(need < grid_probe ? need : grid_probe), dim3(FIN_TPB), 0, stream, pass, route_seed, n_reads, strands, (int)ix->k,
                           // with seeds the few strands without one wait for round 1's stream launch (round 0's would run a handful of long chains alone)
                           seed ? sq1 : sq0, seed ? ctr + 10 : ctr + 6, aq, ctr + 7, (int)(seed != nullptr), desc, no_prefill ? (int2*)out : (int2*)nullptr, (int)lean);
    }
    if ((rc = (int)hipGetLastError()) != 0) return rc;
    if (out_ready && (e = hipStreamWaitEvent(stream, out_ready, 0)) != hipSuccess) return (int)e;   // the walk kernels are the first to write pairs
    for (uint32_t r = 0; r < R; r++) {
        uint4* const s_in = (r & 1u) ? sq1 : sq0, *const s_out = (r & 1u) ? sq0 : sq1;
        uint32_t* const c = ctr + 4 + 4 * r;
        if (!longk) {
            rc = fin_launch_stream_stage(ix, packed, desc, lds_deque_limit, ovf_list, ovf_count, c + 0, s_in, c + 2, aq, c + 3, grid_stream, stream);
            if (rc) return rc;
        }
        if (ix->k <= 32 && ix->kt3 && ix->fbf && !ix->pos && !ix->ptab && ix->lean_walk)   // lean tables: the instantiation with two look-ups per epoch
            hipLaunchKernelGGL(fin_walk_lean_kernel, dim3(grid_walk), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, (const uint4*)aq, c + 3,
                               s_out, c + 6, list, n_list, (int)(r + 1 == R), c + 1, ctr + 4 * FIN_V4_ROUNDS + 8);
        else if (ix->k >= 33 && ix->kt3 && ix->fbf && !ix->pos && !ix->ptab && ix->lean_walk)   // ... and the lean instantiation for two-word keys
            hipLaunchKernelGGL(fin_walk_long_lean_kernel, dim3(grid_walk), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, (const uint4*)aq, c + 3,
                               s_out, c + 6, longk ? ovf_list : list, longk ? ovf_count : n_list, (int)(r + 1 == R), c + 1, ctr + 4 * FIN_V4_ROUNDS + 8);   // (k > 128: what it gives up goes to the plain kernel's list, as the general instantiation's)
        else if (ix->k <= 32)
            hipLaunchKernelGGL(fin_walk_kernel, dim3(grid_walk), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, (const uint4*)aq, c + 3,
                               s_out, c + 6, list, n_list, (int)(r + 1 == R), c + 1, ctr + 4 * FIN_V4_ROUNDS + 8);
        else
            hipLaunchKernelGGL(fin_walk_long_kernel, dim3(grid_walk), dim3(FIN_TPB), 0, stream, *ix, (const uint4*)packed, desc, (int2*)out, (const uint4*)aq, c + 3,
                               s_out, c + 6, longk ? ovf_list : list, longk ? ovf_count : n_list, (int)(r + 1 == R), c + 1, ctr + 4 * FIN_V4_ROUNDS + 8);
        if ((rc = (int)hipGetLastError()) != 0) return rc;
    }
    // what the pipeline kept back or did not finish: whole reads through kernel 3 (their pre-pass verdicts still stand)
    if (!longk) {
        rc = fin_launch_v3_list(ix, packed, desc, out, strands, lds_deque_limit, ovf_list, ovf_count, wc_v3, pass, list, n_list, grid_v3, stream);
        if (rc) return rc;
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
    return fin_launch_overflow(ix, bases, offs, out_offs, out, strands, ovf_list, ovf_count, ovf_scratch, ovf_blocks, stream);
}
extern "C" void fin_debug_dump_w(void) {
#ifdef FIN_W_DEBUG
    unsigned long long h[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fin_wdbg), sizeof h);
    fprintf(stderr, "[fin_wdbg] seed->no place %llu  pass while bridging %llu  pass non-unique %llu  pass unique-but-no-seed %llu  unsafe place %llu  probe items %llu  seed items %llu  anchor items %llu | two-word table: hits %llu misses %llu further slots %llu | string filter: known %llu absent %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[12], h[10], h[11]);
    fprintf(stderr, "[fin_wdbg] k-mer table claims the text did not bear out %llu  items out of epochs %llu\n", h[13], h[14]);
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_wdbg), h, sizeof h);
    {
        unsigned long long f[80];
        (void)hipMemcpyFromSymbol(f, HIP_SYMBOL(g_fin_kfv), sizeof f);
        fprintf(stderr, "[fin_kfv] x0 differs %llu  x1 differs %llu  pp>0 %llu  few bits %llu  many bits %llu | by offset:", f[64], f[65], f[67], f[68], f[69]);
        for (int i = 0; i < 64; i++) fprintf(stderr, " %llu", f[i]);
        fprintf(stderr, "\n");
        memset(f, 0, sizeof f);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_kfv), f, sizeof f);
    }
    {
        unsigned long long w[40];
        (void)hipMemcpyFromSymbol(w, HIP_SYMBOL(g_fin_wstate), sizeof w);
        static const char* names[] = {"DONE", "ITEM0", "ITEM1", "DESC", "RES1", "RES3", "RES4", "RES5", "WALK", "PROBE1", "PROBEX", "PROBE0", "REANCH", "SAFE", "KF0", "KF1", "PROBEF", "KF0B", "KFX", "KFV"};
        fprintf(stderr, "[fin_wstate] wave-epochs %llu  states present per wave-epoch %.2f  live lanes per wave-epoch %.1f | lane-epochs by state:", w[32], w[32] ? (double)w[33] / (double)w[32] : 0.0, w[32] ? (double)w[34] / (double)w[32] : 0.0);
        for (int i = 0; i < 20; i++) fprintf(stderr, " %s %llu", names[i], w[i]);
        fprintf(stderr, "\n");
        memset(w, 0, sizeof w);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_wstate), w, sizeof w);
    }
    {
        unsigned long long t[96];
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fin_witem), sizeof t);
        fprintf(stderr, "[fin_witem] epochs per item (a strand + the sister its lane went on with), by power of two:");
        for (int b = 0; b < 20; b++) if (t[b]) fprintf(stderr, " 2^%d: %llu", b, t[b]);
        fprintf(stderr, " | longest %llu, summed %llu\n[fin_witem] epochs per wave, by power of two:", t[40], t[41]);
        for (int b = 0; b < 24; b++) if (t[48 + b]) fprintf(stderr, " 2^%d: %llu", b, t[48 + b]);
        fprintf(stderr, " | longest wave %llu\n", t[88]);
        fprintf(stderr, "[fin_witem] deferred strands: %llu lane-epochs; %llu sisters gone on with, %llu slots in their stretches of %llu in their reads\n", t[90], t[91], t[92], t[93]);
        memset(t, 0, sizeof t);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_witem), t, sizeof t);
    }
    {
        static unsigned long long tm[8192 + 8];
        (void)hipMemcpyFromSymbol(tm, HIP_SYMBOL(g_fin_wtime), sizeof tm);
        std::vector<unsigned long long> e;
        for (int i = 0; i < 8192; i++) if (tm[i]) e.push_back(tm[i]);
        if (!e.empty()) {
            std::sort(e.begin(), e.end());
            auto at = [&](double f) { return (double)e[(size_t)(f * (double)(e.size() - 1))] * 0.01; };
            fprintf(stderr, "[fin_wtime] when the waves of the last big launch ended, microseconds after its first wave began (%zu waves): min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f\n",
                    e.size(), at(0.0), at(0.1), at(0.5), at(0.9), at(0.99), at(1.0));
        }
        memset(tm, 0, sizeof tm);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_wtime), tm, sizeof tm);
    }
#endif
}
