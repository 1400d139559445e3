// fin_format.h -- the HBM layout of the finimizer index, shared by host and device code.
//
// MI355X-first design (DESIGN.md "Data layout"): the reference keeps seven separately allocated succinct
// structures (4 bit-planes + rank_support_v5 each, packed LCS, fmin/Ustart bit-vectors + rank, offsets, ends,
// packed text; FinimizerIndex.hh:108-115).  On the GPU every dependent access is a random 128-B line from
// HBM/Infinity Cache, so everything the streaming search needs about 64 consecutive SBWT nodes lives in ONE
// 128-byte line: the LCS bytes the next drop_first_char scans, the plane words and pre-added C[c]+rank bases
// the next extend needs, and the Ustart/fmin flags and their ranks for the dictionary lookups.
#pragma once
#include <stdint.h>

#define FIN_BLOCK_NODES 64
#define FIN_LCS_MASK 0x7Fu      // node byte bits 0-6: min(LCS, 127) -- exact for k <= 128; longer k: the exact values are in FinDevIndex::lcs8
#define FIN_USTART_BIT 0x80u    // node byte bit 7: Ustart[i] (probed every step next to the LCS bytes, common.hh:167)
#define FIN_MAX_K 255           // the reference's range (LCS in a byte, lcs_basic_parallel_algorithm.hpp:54-57)
#define FIN_FAST_K 128          // up to here LCS values (<= k-1) fit the 7 bits of a node byte, which is what the streaming kernels 2 / 3 read

struct FinCharRec {         // what an extend by one character needs from a block: ONE 12-byte load
    uint32_t plane_lo, plane_hi;   // outgoing-edge marks of the 64 nodes for this character
    uint32_t base;                 // C[c] + rank_c(64*b): start of the target interval of an extend from this block
};
struct alignas(128) FinNodeBlock {
    uint8_t node[64];       // per node: LCS | Ustart<<7                                          [0,64)
    FinCharRec rec[4];      // A, C, G, T                                                          [64,112)
    // Thermometer copy of the LCS for the three thresholds almost every drop_first_char uses (lcs_t0+1 .. lcs_t0+3, chosen
    // per index from its LCS histogram): th1:th0 of node i = min(3, max(0, LCS[i] - lcs_t0)).  One 16-byte load gives the
    // stop mask of the WHOLE block for such a threshold, so those scans need no byte window and never leave the line.
    uint64_t th0, th1;      //                                                                     [112,128)
};
static_assert(sizeof(FinNodeBlock) == 128, "one block = one 128-B line");
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t fin_plane(const FinCharRec& r) { return r.plane_lo | ((uint64_t)r.plane_hi << 32); }

// Dictionary side of a block, only needed at dictionary lookups (about 0.5 % of the bases): kept out of the hot line.
// Laid out so that ONE 16-byte load yields a mask together with its rank: bytes [0,12) = fmin rank + mask,
// bytes [12,24) = Ustart mask + rank (a load at offset 8 covers them).
struct FinBlockInfo {
    uint32_t fmin_rank;                       // ones of fmin before this block
    uint32_t fmin_mask_lo, fmin_mask_hi;      // fmin bits of the 64 nodes
    uint32_t ustart_mask_lo, ustart_mask_hi;  // Ustart bits of the 64 nodes
    uint32_t ustart_rank;                     // ones of Ustart before this block
};
static_assert(sizeof(FinBlockInfo) == 24, "packed");

// What a kernel needs to know about the index (passed by value).
struct FinDevIndex {
    const FinNodeBlock* blocks;
    const FinBlockInfo* blkinfo; // per block: fmin / Ustart masks and ranks (dictionary lookups only)
    const uint32_t* goff;        // global_offsets in fmin-rank order
    const uint32_t* ends;        // ends_p: ends_p[0] = 0, ends_p[u+1] = exclusive end of unitig u, then 8 x 0xFFFFFFFF
    const uint32_t* samp;        // samp[g >> samp_shift] = number of ends <= (g >> samp_shift) << samp_shift
    const uint32_t* concat;      // 2-bit packed unitig text, 16 bases per word, base i at bits 2*(i&15)
    uint32_t n_nodes;
    uint32_t n_unitigs;
    uint32_t total_len;
    uint32_t k;
    uint32_t samp_shift;
    uint32_t n_samp;
    uint32_t C[5];               // C[0..3], C[4] = n_nodes
    uint32_t lcs_t0;             // thresholds lcs_t0+1..lcs_t0+3 are answered by th0/th1
    uint32_t ptab_t;             // prefix table depth T (0: none)
    const struct FinPrefixIval* ptab;  // 4^T intervals: entry key = sum code(s[i]) << 2i of a T-base string s; l > r if s does not occur (device-built)
    // jump table: the same for J-base strings, J chosen so that nearly every J-base string of the indexed text occurs at least
    // twice (4^J <= n_nodes / 3).  A (re)started streaming search takes its state after J bases from here (fin_kernel_v3.hip).
    uint32_t jtab_t;             // J (0: none)
    const struct FinPrefixIval* jtab;
    // epochs a read may use before it is handed to the overflow kernel: budget_mult * length + budget_add (64, 4096 by default; a
    // healthy read needs about 3 per base.  Tests shrink it to force that path: fin_set_option "epoch_budget_mult")
    uint32_t budget_mult, budget_add;
    // capacity (entries) of the batch's overflow list: a push beyond it is dropped and the overflow kernel reads no further (set per run;
    // the list is sized so that this never binds -- fin_capi.cpp, batch_load -- the bound only keeps a miscount from leaving the allocation)
    uint32_t ovf_cap;
    // 1: text re-anchoring is on (option "text_anchors"; needs the tables below): behind a read base that disagrees with the unitig text the
    // walk proves the k-mers across it absent and compares the k-mer behind it with the text (fin_kernel_w.hip, fin_kernel_v3.hip).
    uint32_t text_anchors;
    // Anchor table (device-built at upload by fin_kernel_b.hip, null when absent or switched off): one 16-byte entry per node,
    // pos[v] = {g, u, ustart, uend}.  g = the reference's ANSWER for node v's k-mer when that k-mer is not reached by a walk: the offset
    // in the concatenation of the last base of the place FinimizerIndex::search computes from its dictionaries (a function of the k-mer
    // alone, fin_kernel_b.hip).  u = the unitig of that place and ustart/uend its bounds -- everything the walk needs to start there, in
    // one load -- written only when the text AT g spells v's k-mer inside one unitig ("verified"; else u keeps its top bit set: with
    // duplicated k-mers the reference may report a place where the k-mer is not, and such an entry is only used once the k-mer's presence
    // is known from a look-up of the whole k-mer).  g = FIN_POS_DUMMY | d for the dummy node that holds the first d < k bases of a unitig
    // behind k-d '$' (no k-mer ends with a string that only such a node ends, nor with an extension of it by fewer than k-d bases);
    // g = 0xFFFFFFFF: nothing known.  A probe string that matched completely and is the suffix of exactly one node v names the only k-mer
    // that can end there: the walk kernel compares the read with the text at pos[v] instead of running the streaming search to find the
    // first anchor (fin_kernel_w.hip).
    const struct FinSeedEntry* pos;
    // Safe-place bitmap (null: every place is safe -- a set of disjoint unitigs): bit g = the k-mer that the text spells at [g-k+1, g] is
    // reported AT g by the reference (pos[its node].g == g).  A k-mer found by comparing a read with the text is reported there only if
    // its bit is set; else the streaming search decides.  One u64 per 64 text positions.
    const unsigned long long* safe;
    // K-mer table (round 5: the COMPACT form; device-built at upload for every k <= 255 -- the pre-pass's looks and the walk kernel's look-ups; above 63 the
    // walk kernel keeps no key: the words are folded into the hash as the chunk cache brings them; null: none): a bucketed hash table over the k-mers of the unitig text.
    // A slot is 8 bytes {g, meta}: g = the reference's ANSWER for the k-mer (what the anchor table holds for its node: the offset in the concatenation of
    // the last base of the place FinimizerIndex::search reports), meta = a 30-bit TAG of the k-mer's hash | FIN_KT3_UNVER.  A bucket = 4 slots = 32 bytes, one
    // load; kt3_buckets buckets (any number: bucket = high hash word * kt3_buckets >> 32), filled to 55 %; a k-mer whose bucket is full lies in the next.
    // The table holds no k-mer: a tag match is a CLAIM that the read's k-mer is in the index with its answer at g, and the text at g -- which a verified
    // answer spells -- is the proof.  Every user compares: the fast path lays the whole read beside that text anyway (fin_prepass.hip), the walk kernel
    // compares the k bases before the run starts there (W_REANCH, fin_kernel_w.hip).  A k-mer without a matching tag in its bucket chain (up to the
    // first empty slot) is absent for certain.  A false match (2^-30 per slot looked at) fails the comparison and sends the read to kernel 3, which asks
    // no table.  FIN_KT3_UNVER: the answer of this k-mer is NOT a place that spells it (duplicated k-mers, FIN_POS_UNVERIFIED) -- nothing can be
    // compared: such k-mers (a few per thousand on a set with duplicated stretches, none on a disjoint one) are kept a second time with their whole
    // keys in the small exact table ktx, which a look-up asks behind such a claim.  14.5 bytes per indexed k-mer whatever k is (round 4: 34 bytes for
    // k <= 31, 68 for k <= 63), and no power-of-two sizing: the table exists for any text below 2^32 bases.
    const struct FinKt3Bucket* kt3;
    uint32_t kt3_buckets;
    // the exact side table of the k-mers whose answer is unverified: 2^ktx_log2 slots of 32 bytes {k0, k1, g, claim} (claim 0xFFFFFFFF: empty), linear probing
    // from the high word of the k-mer's hash, at most half full; null: the index has no such k-mer.  A k-mer with an unverified claim in kt3 that is not
    // found here (a tag shared with another k-mer; or the upload's list of such k-mers overran, ktx_partial) is decided by kernel 3
    const struct FinKtxSlot* ktx;
    uint32_t ktx_log2;
    // Canonical string filter (round 4; device-built at upload, null: none): a blocked Bloom filter over the strings of cbf_m bases (20; fewer for k < 29: 3 (k-cbf_m+1) >= k) that
    // occur inside a unitig, entered in CANONICAL form -- the smaller of the string and its reverse complement -- 2^cbf_log2 blocks of 128 bits,
    // FIN_CBF_BITS bits per string inside ONE block: one 16-byte load says "this string occurs in no unitig, and neither does its reverse
    // complement" (no false negative: every bit of a string that was entered is set).  A string that does not occur rules out every k-mer that
    // contains it -- on BOTH strands of a read at once: the fast path (fin_prepass.hip) proves the k-mer ends across a sequencing error absent with
    // two or three such loads, where the probes of the walk kernel need a prefix-table entry and up to four node blocks per strand.
    const struct FinCbfBlock* cbf;
    uint32_t cbf_log2, cbf_m;
    // Directional string filter (round 4, "lean tables"; null: none): the same blocked Bloom filter over the strings of cbf_m bases inside unitigs, each
    // entered AS IT STANDS (not canonically): "this string occurs in no unitig in this orientation" -- what a probe asks.  With it and the k-mer
    // table (k <= 31) the walk kernel and the pre-pass's stepping loop need neither the prefix table (a probe = one 16-byte load instead of a table
    // entry and up to four node blocks; a string that occurs is followed by a look-up of the whole k-mer in the k-mer table, not by a seed) nor
    // the anchor table (a look's slot holds the place).  Same geometry as cbf (cbf_log2 blocks, cbf_m bases).
    const struct FinCbfBlock* fbf;
    uint32_t fast_path;          // 1 (set per run, option "fast_path"): the pair pre-pass may finish reads by itself (fin_prepass.hip)
    // 1 (set per run): the second strand of a read is DEFERRED -- searched only where the first strand left slots open (kernel 4;
    // fin_prepass.hip, fin_kernel_w.hip; CHANGELOG.md 4.14): a k-mer the first strand reports AT A PLACE THAT SPELLS IT is in the index, so its
    // reverse complement -- the other strand's k-mer in that slot -- is not, unless the index holds both.  rcwin (null: no k-mer of the index
    // has its reverse complement in it) marks the windows of 64 text positions in which such a k-mer ends: a strand that reports from one
    // of them, or through anything but a seed and walks, has its sister searched in full.
    uint32_t defer_ok;
    const uint8_t* rcwin;
    // Absence filter (device-built at upload; null: none): one bit per string of filt_f bases, set iff the string occurs in a unitig;
    // bit index = sum code(s[i]) << 2i, as the prefix table's key.  4^filt_f bits -- 32 MB at 250 Mbp, small enough to stay in the
    // Infinity Cache -- so the pre-pass can rule out most k-mer ends of a strand that matches nothing without touching HBM.
    const uint32_t* filt;
    uint32_t filt_f;
    // k > FIN_FAST_K only (else null): the exact LCS array, a byte per node.  The node bytes then hold min(LCS, 127); the streaming
    // kernels 2 / 3 (which compare 7-bit values 16 at a time) are not used for such an index: kernel 4's pre-pass and walk kernel need
    // no LCS at all, and what they cannot finish goes to the plain kernel (kernel 0), which reads this array.
    const uint8_t* lcs8;
    // Text modes of a batch (set per run: fin_batch_text_mode).  frec (null: none): the fast path leaves a 32-byte record per read it finishes --
    // unitig, first offset, the disagreeing positions -- from which fin_text.hip makes that read's text without reading its pairs back;
    // text_only: the pairs of such reads are not written at all (the text is the batch's only product, as in search_fmin.hh:62-65)
    struct FinFastRec* frec;
    uint32_t text_only;
    uint32_t lean_walk;          // host side only (set per run, option "lean_walk"): 1 = under lean tables the walk kernel's lean instantiations (k <= 31: two k-mer-table look-ups per epoch in a run of misses)
    uint32_t pp_seg;             // host side only (set per run, option "debug_pp_seg"; 0: by batch size): reads per block of the pair pre-pass
};
// What the fast path knows about a read it finished (fin_prepass.hip: FastRun): strand A (meta bit 8: the reverse strand) lies in unitig u with its
// first base at offset off0 and disagrees with the text at positions E (meta bits 0..7: how many; 16 bits each, Es then Es2); meta >> 16 = 2: every
// k-mer of the read is absent.  Slot sl of strand A is (u, off0 + sl) unless a disagreeing position lies in [sl, sl + k - 1]
struct FinFastRec { uint32_t u, off0, meta, nk; uint64_t Es, Es2; };
// The compact k-mer table's bucket (FinDevIndex::kt3): four slots {g, meta}.  An empty slot is all ones; a used slot's meta has bit 31 clear.
struct FinKt3Bucket { uint32_t w[8]; };
struct FinKtxSlot { uint32_t k0_lo, k0_hi, k1_lo, k1_hi, g, claim, pad0, pad1; };
#define FIN_KT3_SLOTS 4
#define FIN_KT3_TAGMASK 0x3FFFFFFFu
#define FIN_KT3_UNVER 0x40000000u
#define FIN_KT3_EMPTY 0xFFFFFFFFFFFFFFFFull
// (linear probing over buckets clusters: a miss reads 0.23 / 0.34 / 0.49 further buckets at a load of 50 / 55 / 60 %, and one chain in 30 000 is longer than
//  sixteen buckets at 60 % -- simulation and the device's own counters agree; a further bucket is a further epoch of the walk kernel's look-up)
#define FIN_KT3_LOAD_PCT 55
// hash of a k-mer given as two words of 2-bit codes (first base in the low bits; k0 = bases 0..31, k1 = bases 32..k-1, 0 for k <= 32): the high word picks
// the bucket, the low 30 bits are the tag.  A full 64-bit finaliser (xor-shift / multiply / xor-shift / multiply / xor-shift) on each word: the k-mers of a
// repeat family differ in a base or two, and a table whose proof is "the text at the answer spells the k-mer" pays for every pair of such siblings that shares
// bucket and tag with a read sent to kernel 3 -- the first form (32-bit multiplies, the second word folded in by one multiply and a rotation) let 411 of the
// 19 million 63-mers of a 20 Mbp repeat-rich set collide, 410 of them siblings a few bits apart (differences in the high bits of a word only travel upwards
// through a multiply): 4.5 % of k63_repeats' claims failed.  This one: none (tools/kt3_collisions.cpp).
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t fin_mix64(uint64_t x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
    return x;
}
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t fin_kt3_hash(uint64_t k0, uint64_t k1) {
    uint64_t key = k0;
    if (k1) key ^= fin_mix64(k1 + 0x9E3779B97F4A7C15ull);   // (= fin_kt3_fold(k0, k1, 1))
    return fin_mix64(key);
}
// ... of a k-mer of any length, its key words folded one by one (words 0 and 1 as above, so both forms agree for k <= 64): word j >= 1 enters as a finaliser of
// itself plus j times the constant -- the order of the words matters, a word of zeros (a run of A) adds nothing, as above
#define FIN_KT3_FOLD 0x9E3779B97F4A7C15ull
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t fin_kt3_fold(uint64_t key, uint64_t word, uint32_t j) { return word ? key ^ fin_mix64(word + FIN_KT3_FOLD * (uint64_t)j) : key; }
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint32_t fin_kt3_bucket(uint64_t h, uint32_t n_buckets) { return (uint32_t)(((h >> 32) * (uint64_t)n_buckets) >> 32); }
struct FinCbfBlock { uint32_t w[4]; };   // 128 bits of the canonical string filter (FinDevIndex::cbf)
// the canonical string filter's block and bits of a string's canonical 2-bit key (shared by the build kernel, the fast path and the tests)
#define FIN_CBF_BITS 5
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t fin_cbf_hash(uint64_t key) {
    key ^= key >> 29; key *= 0xBF58476D1CE4E5B9ull; key ^= key >> 32; key *= 0x94D049BB133111EBull; key ^= key >> 29;
    return key;
}
#define FIN_PASS_DONE 0xFFFFFFFDu       // pre-pass verdict of BOTH strands of a read the fast path finished: every output slot of the read is written (fin_prepass.hip)
#define FIN_PASS_DEFERRED 0xFFFFFFFEu   // pre-pass verdict of a strand whose search waits for its sister strand's result (FinDevIndex::defer_ok)
struct FinPrefixIval { uint32_t l, r; };
struct FinSeedEntry { uint32_t g, u, ustart, uend; };
#define FIN_POS_DUMMY 0xFFFFFF00u   // anchor-table entries at or above this (and below 0xFFFFFFFF): a dummy node, low byte = its number of bases
                                    // (the table is only built for indexes whose text is shorter than this)
#define FIN_POS_UNVERIFIED 0x80000000u   // FinSeedEntry::u, top bit: the text at g does not spell the node's k-mer (or nothing is known): g is an answer, not a place

// One read of a batch as the tuned kernel sees it (16 bytes, one load)
struct FinReadDesc { uint64_t off; uint32_t len; uint32_t out_off; };   // byte offset of the bases, length, first output pair

// Container file <prefix>.finamd
#define FIN_MAGIC 0x31444d414e4946ull   // "FINAMD1"
struct FinFileHeader {
    uint64_t magic;
    uint32_t version;   // 4; 5 = k > 128: the exact LCS array (n_nodes bytes) follows the other sections
    uint32_t k;
    uint64_t n_nodes, n_kmers, n_unitigs, total_len, n_fmin;
    uint64_t C[4];
    uint32_t samp_shift, n_samp;
    uint64_t n_blocks, n_concat_words;
    uint64_t lcs_t0;
    uint64_t reserved[3];
};
