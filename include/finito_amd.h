/*
 * finito_amd.h -- C ABI of the MI355X-native search-fmin path (libfinito_amd.so).
 *
 * The reference (ElenaBiagi/Finito) has no FFI/plugin layer: its boundary for this path is the C++ class
 * FinimizerIndex (include/FinimizerIndex.hh:26-259) and the two commands build_fmin / search_fmin
 * (include/build_fmin.hh:302, include/search_fmin.hh:130).  Every entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no C++ or torch types.  INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Conventions
 *  - every function returning int returns FIN_OK (0) or a negative FIN_E* code and, when err != NULL, writes
 *    a NUL-terminated message into err[0..errlen).  Nothing here throws or aborts.
 *  - sequences are ASCII, upper or lower case ACGT; offsets[i]..offsets[i+1] delimit record i (n+1 entries).
 *  - a (unitig, offset) result is two int32 on the batch path (-1,-1 = k-mer absent), two int64 on the
 *    single-read path (as FinimizerIndex::QueryResult::local_offsets, FinimizerIndex.hh:30-33).
 *  - search entry points need a HIP device; there is NO CPU fallback: without a device they fail with
 *    FIN_ENODEV.  Building, saving and loading an index need no device.
 *  - handles are not copyable; a handle may be searched from several host threads at once (each call brings
 *    its own stream/scratch), matching FinimizerIndex::search being const (FinimizerIndex.hh:119).  Adding a
 *    replica (fin_index_to_device) and freeing are NOT concurrent-safe: do them before sharing / after joining.
 *  - lifetime: a fin_batch borrows the HBM replica of its index; free every batch before fin_index_free.
 */
#ifndef FINITO_AMD_H
#define FINITO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIN_OK 0
#define FIN_EINVAL (-1)   /* bad argument (k out of range, unitig shorter than k, non-ACGT base in a unitig ...) */
#define FIN_EIO (-2)      /* file could not be read/written or is not a finito-amd container */
#define FIN_ENODEV (-3)   /* no HIP device / HIP call failed */
#define FIN_ENOMEM (-4)
#define FIN_ELIMIT (-5)   /* size limit of this build (n_nodes or total unitig length >= 2^32, k > 255: LCS values are kept in a byte, as in the reference) */

typedef struct fin_index fin_index;   /* index: host copy + (after fin_index_to_device) one HBM replica */
typedef struct fin_batch fin_batch;   /* a batch of reads resident in HBM with its output buffer */

const char* fin_version(void);

/* Process-wide tuning/debug switches (no reference counterpart).  Returns FIN_OK or FIN_EINVAL.  A change is seen by every handle that
 * has no value of its own (fin_index_set_option below) the next time a batch is loaded or run -- to change the behaviour of one handle
 * while others are in use, use the per-handle form.
 *   "lds_deque_limit" 1..16 : live candidates a lane keeps in LDS before the read is redone with the deque in
 *                             global memory (default 16; tests lower it to exercise that path)
 *   "kernel"        0|2|3|4 : 4 (default) = the lazy search as a pipeline of specialised kernels (probe -> stream -> walk, items handed on
 *                             through queues in HBM); 3 = the same lazy algorithm with a whole read per lane (walk mode, restarts,
 *                             probing -- what the pipeline leaves over runs on it); 2 = the kernel that streams every base of both
 *                             strands like the reference; 0 = plain lane-per-read kernel.  Same results from all; applies to
 *                             batches loaded afterwards.  An index with k > 128: 2 and 3 mean 0 (their LCS scans are 7-bit), 4 runs
 *                             when the index has an anchor table and otherwise means 0
 *   "probe_prepass"   0|1   : kernel 3: 1 (default) = all strands are probed by a separate light kernel first and the search kernel
 *                             starts each strand where that says; 0 = probing happens inside the search kernel
 *   "ptab_t"          -1..15: depth of the prefix table that fin_index_to_device builds for kernel 3's probes (-1 = by index
 *                             size, the default; 0 = none); applies to replicas uploaded afterwards
 *   "epoch_budget_mult" 0..64, "epoch_budget_add" 1..2^20 : epochs a read may use in the tuned kernels before it is handed to the
 *                             overflow kernel = mult * length + add (64, 4096; tests shrink them to force that path)
 *   "text_anchors"    0|1   : 1 (default) = kernels 4 and 3 prove the k-mers across a sequencing error absent and find the k-mer behind it
 *                             by comparing the read with the unitig text -- at places the upload found "safe": the k-mer the text spells
 *                             there is reported there by the reference (every place of a disjoint unitig set; fin_index_unsafe_places);
 *                             0 = they restart the streaming search there (same results)
 *   "seed_anchors"    0|1   : 1 (default) = fin_index_to_device builds the anchor table (per SBWT node the place the reference reports for
 *                             its k-mer, with that unitig's bounds, 16 bytes per node) and kernel 4 finds a strand's anchors through it: a
 *                             probe string that matched completely and ends exactly one node names the only k-mer that can end there, the
 *                             read is compared with the text at its place (CHANGELOG.md 4.9); 0 = anchors come from the streaming search (same
 *                             results).  Any index qualifies, duplicated k-mers or not.  Applies to replicas uploaded afterwards (table)
 *                             and to later runs (use)
 *   "kmer_table"      0|1   : 1 (default) = fin_index_to_device also builds, in the anchor pass, the COMPACT k-mer table (round 5; any k <= 255): a bucketed
 *                             hash table over the k-mers of the unitig text, 8-byte slots {the reference's answer for the k-mer, a 30-bit tag of its
 *                             hash, "answer unverified"}, four slots to a 32-byte bucket, 55 % full -- 14.5 bytes per indexed k-mer whatever k is (round 4:
 *                             34 bytes at k <= 31, 68 at k <= 63), for any text below 2^32 bases.  The table holds no k-mer: a tag match is a claim
 *                             that the text at the answer proves or refutes -- the fast path compares the whole read there anyway, the walk kernel
 *                             compares the k bases before a run starts; a k-mer without a match up to the first empty slot of its chain is absent
 *                             for certain; a false match (2^-30 per slot) sends the read to kernel 3; a k-mer whose answer is unverified (duplicated k-mers) is kept whole in a small exact side table.
 *                             The pair pre-pass asks it for a read's first, last and middle k-mers, kernel 4's walk kernel wherever a probe string
 *                             that occurs leaves a k-mer end undecided: one 32-byte load instead of a look-up of the whole k-mer through the SBWT
 *                             (a prefix-table entry and k-T node blocks).  0 = whole-k-mer look-ups (same results).  Upload and run time
 *   "defer_strand"    0|1   : 1 (default) = kernel 4 searches the second strand of a read only between the first and the last slot the first
 *                             strand left open -- on ANY index: a first strand that reports through the streaming search or a whole-k-mer
 *                             look-up (a place that may not spell its k-mer: duplicated k-mers), or from a text window that holds a k-mer
 *                             whose reverse complement is in the index too (fin_index_rc_pairs > 0), has its sister searched in full; a
 *                             deferred FORWARD strand's walk runs on past its stretch to the read's end and wins the slots it reaches, as
 *                             the reference's forward search does (search_fmin.hh:54-60).  0 = both strands in full (same results)
 *   "fast_path"       0|1|2 : 1 (default) = with deferred second strands and a k-mer table (k <= 63) the pair pre-pass finishes by itself
 *                             the reads that lie inside one unitig with up to four substitutions -- one comparison with the text behind
 *                             the place of one of the read's k-mers; the k-mer ends across a disagreeing base proven absent on both strands
 *                             by strings the canonical string filter does not know -- and the reads none of whose k-mers it finds, when
 *                             that filter knows none of the strings laid across them; 2 = for every k <= 255 (not the default above 63: the walk kernel
 *                             asks the k-mer table there too -- a long k-mer's key words folded into the hash as its chunks arrive -- and at k = 127 the
 *                             fast path, which finishes 55 % of the benchmark's reads, makes the step 11 % slower); 0 = every read through the pipeline
 *                             (same results)
 *   "cbf_m"           -1..32: string length of the string filters built at upload (-1 = 20, less for k < 29; 0 = none)
 *   "lean_tables"     0..3  : at upload (with "kmer_table", "seed_anchors", "text_anchors" on, "cbf_m" not 0 and "ptab_t" -1): NO prefix table and NO anchor
 *                             table -- the compact k-mer table, the canonical and the directional string filter and the jump table only.  A probe asks the
 *                             directional filter about a string of 20 bases (one 16-byte load instead of a table entry and up to four node blocks), a
 *                             string that occurs is followed by a look-up of the whole k-mer in the k-mer table (whose slot holds the place), the
 *                             pre-pass hands on places, not nodes.  2 (default since round 5) = for k <= 63: 21 bytes of
 *                             tables per indexed base at 250 Mbp whatever k is (round 4: 41 at k <= 31, 124 at k = 63; round 3: 89); 3 = for every k <= 255
 *                             (above 63: 21 instead of 70 bytes per base, and at k = 127 a step of 17.0 instead of 10.3 ms -- without seeds by node every
 *                             anchor is a whole-k-mer look-up -- so not the default there); 1 = for k <= 31 only
 *                             (32 <= k <= 63 then keeps round 3's tables: 68 bytes per base, 6 % faster on iid reads at k = 63, 32 % slower on a
 *                             repeat-rich genome -- DESIGN.md §7); 0 = round 3's tables
 *   "lean_walk"       0|1   : kernel 4 under lean tables: 1 (default) = the walk kernel's lean instantiations -- without the prefix-table / rank-record
 *                             states (fewer registers, no scratch); k <= 31: behind a k-mer the k-mer table does not have, the next end's k-mer
 *                             (the old one shifted by a base) is looked up in the same epoch, so a run of absent k-mers moves two ends per epoch;
 *                             0 = the general instantiation (same results)
 *   "write_gaps"      0|1   : kernel 4 on an index with a seed table: 1 (default) = the output is not prefilled with (-1,-1); the
 *                             lane that searches a read's only strand writes the absent slots with the pairs, the route kernel fills the
 *                             reads nobody searches (every slot is written once); 0 = prefill, pairs overwrite
 *   "overlap_prefill" 0|1   : kernel 4: 1 (default) = the (-1,-1) prefill of the output runs on a side stream beside the ingest kernel and the
 *                             pre-pass, joined before the first pairs are written; 0 = on the launch stream, in front of them
 *   "filt_f"          -1..16: depth of the pre-pass's absence filter (-1 = by index size, the default: the smallest F in 8..12 with 4^F >= text
 *                             length -- a bit set that stays in the L2 --, none for larger indexes; 0 = none); applies to replicas
 *                             uploaded afterwards
 *   "jtab_t"          -1..14: depth of the jump table (-1 = by index size: 4^J <= nodes / 3, the default; 0 = none); applies to replicas
 *                             uploaded afterwards
 *   "max_batch_kmers" n     : fin_search_batch processes inputs with more k-mers than this as consecutive device
 *                             batches (default 2^30; tests lower it)
 *   "pipeline_kmers"  n     : k-mers per sub-batch of fin_search_batch's copy/compute pipeline (default 2^26)
 *   "pipeline_depth"  1..8  : sub-batches in flight per device (default 3: upload, search and download overlap)
 *   "stage_pageable"  0|1   : 1 (default) = pageable caller buffers are staged through pooled page-locked memory by the
 *                             pipeline's threads; 0 = handed to the runtime as they are (also frees the pool) */
int fin_set_option(const char* name, int64_t value);
/* The same switches for ONE index handle: the handle uses its own value, every other handle keeps following the process-wide one
 * (fin_index_clear_option: this handle follows it again).  Touches nothing but the handle, so it is the form to use when handles are
 * shared between threads; upload-time options (ptab_t, jtab_t, filt_f, seed_anchors, text_anchors, kmer_table) must be set before
 * fin_index_to_device.  FIN_EINVAL for an unknown name or a value out of range. */
int fin_index_set_option(fin_index* idx, const char* name, int64_t value);
int fin_index_clear_option(fin_index* idx, const char* name);
/* usable host cores: affinity mask capped by the cgroup CPU quota and by $FINITO_THREADS (default cap 64) */
int fin_host_threads(void);

/* ---- index construction and persistence ------------------------------------------------------------------ */

/* Replaces the whole build-fmin chain for type "rarest", t = 1: `sbwt build` (external, README.md:33-35),
 * lcs_basic_parallel_algorithm (lcs_basic_parallel_algorithm.hpp:52), permute_unitigs (PackedStrings.hh:105)
 * and the FinimizerIndexBuilder constructor (FinimizerIndex.hh:273-319).  unitigs must be a spectrum-preserving
 * string set, every unitig at least k long.  n_threads <= 0 means all cores. */
int fin_index_build(const char* unitig_bases, const uint64_t* unitig_offsets, uint64_t n_unitigs, int k,
                    int n_threads, fin_index** out, char* err, size_t errlen);

/* The same construction on a HIP device (every k <= 255: two-word keys above 32, four above 64, eight above 128; finito_amd/csrc/fin_build_gpu.hip): k-mer extraction, radix sort, dummy nodes, LCS from
 * neighbouring keys, edge marks, permute_unitigs and the finimizer pass as kernels -- the index it returns (host side, like
 * fin_index_build's) is bit-identical to the host builder's (same container file).
 * phase_ms: NULL, or 8 doubles that receive the device time of its stages (upload+k-mers, sort, dummies, SBWT, unitigs, finimizers,
 * dictionaries, copy back). */
int fin_index_build_device(const char* unitig_bases, const uint64_t* unitig_offsets, uint64_t n_unitigs, int k, int device,
                           fin_index** out, double* phase_ms, char* err, size_t errlen);

/* FinimizerIndex::serialize(prefix) (FinimizerIndex.hh:187-207): writes <prefix>.finamd (one container file). */
int fin_index_save(const fin_index* idx, const char* prefix, char* err, size_t errlen);
/* FinimizerIndex::load(prefix) (FinimizerIndex.hh:209-241): <prefix>.finamd if it exists, else the reference's own seven files
 * <prefix>.{O,FBV,packed_unitigs,unitig_endpoints,Ustart,LCS}.sdsl + <prefix>.sbwt (an index built by the reference's tools). */
int fin_index_load(const char* prefix, fin_index** out, char* err, size_t errlen);
/* The reference's on-disk layout itself, written / read explicitly (FinimizerIndex::serialize / load, FinimizerIndex.hh:187-241;
 * byte layouts of sdsl::int_vector / bit_vector and sbwt::plain_matrix_sbwt_t::serialize restated in finito_amd/csrc/fin_sdsl.cpp).
 * PARITY UNPINNED: the reference tree ships no index file and no serialization test; round trip and layout are tested here. */
int fin_index_save_reference_layout(const fin_index* idx, const char* prefix, char* err, size_t errlen);
int fin_index_load_reference_layout(const char* prefix, fin_index** out, char* err, size_t errlen);
/* The SBWT alone as the file `sbwt build` writes and build-fmin -i reads (string "plain-matrix", then plain_matrix_sbwt_t::serialize;
 * build_fmin.hh:346-364); fin_sbwt_file_info reads k and the node / k-mer counts from such a file. */
int fin_index_save_sbwt(const fin_index* idx, const char* path, char* err, size_t errlen);
int fin_sbwt_file_info(const char* path, int64_t* k, int64_t* n_nodes, int64_t* n_kmers, char* err, size_t errlen);
/* build-fmin's -i <x.sbwt> and --lcs <file> (build_fmin.hh:346-383): here the SBWT and the LCS are functions of the unitigs and k and are
 * rebuilt, so the files can only be CHECKED against what was built: FIN_EINVAL with a message if either differs.  NULL / "" = skip. */
int fin_index_check_against_files(const fin_index* idx, const char* sbwt_path, const char* lcs_path, char* err, size_t errlen);
void fin_index_free(fin_index* idx);

/* sbwt->get_k(), number_of_subsets(), number_of_kmers() (search_fmin.hh:187-189), unitigs.number_of_strings(),
 * FinimizerIndex::size_in_bytes() (FinimizerIndex.hh:244-258; here: bytes of the HBM-resident layout). */
int64_t fin_index_k(const fin_index* idx);
int64_t fin_index_n_nodes(const fin_index* idx);
int64_t fin_index_n_kmers(const fin_index* idx);
int64_t fin_index_n_unitigs(const fin_index* idx);
int64_t fin_index_n_finimizers(const fin_index* idx);
int64_t fin_index_total_len(const fin_index* idx);
int64_t fin_index_size_in_bytes(const fin_index* idx);
/* The statistics-only modes of build-fmin (build_fmin.hh:95-214, 252-268): the distinct {length, frequency, colex rank} window
 * finimizers of the given sequences (normally the indexed unitigs) with frequency threshold t >= 1 -- their number, the sum of
 * their frequencies and of their lengths, which is what print_finimizer_stats (common.hh:188-206) reports.  Host code. */
#define FIN_STATS_SHORTEST 1   /* --type shortest: streaming (build_shortest_streaming_search) */
#define FIN_STATS_VERIFY 2     /* --type verify: every substring of every k-window (verify_shortest_streaming_search); O(k^2) per window */
int fin_index_finimizer_stats(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_seqs, int type, int64_t t,
                              int64_t* n_finimizers, int64_t* sum_freq, int64_t* sum_len, char* err, size_t errlen);
/* depth T of the prefix table built for the replica on `device` (4^T entries of 8 bytes; 0 = none, -1 = no replica there) */
int fin_index_prefix_table_depth(const fin_index* idx, int device);
/* depth J of the jump table of the replica on `device` (4^J entries of 8 bytes: the SBWT interval of every J-base string; a (re)started
 * streaming search takes its state after J bases from it; 0 = none, -1 = no replica there) */
int fin_index_jump_table_depth(const fin_index* idx, int device);
/* depth F of the absence filter of the replica on `device` (4^F bits: which strings of F bases occur in the unitigs; the pre-pass asks
 * it before it spends a prefix-table probe; 0 = none, -1 = no replica there) */
int fin_index_filter_depth(const fin_index* idx, int device);
/* bytes of the anchor table of the replica on `device` (16 per SBWT node: the place the reference reports for every node's k-mer; built
 * unless option "seed_anchors" is 0; 0 = none, -1 = no replica there) */
int64_t fin_index_seed_table_bytes(const fin_index* idx, int device);
/* k-mer places of the text whose k-mer has an UNVERIFIED answer on the replica on `device`: the reference reports a place that does not spell the k-mer
 * (duplicated k-mers; 0 on a disjoint set).  The compact k-mer table cannot prove such a k-mer by comparison: they are kept with whole keys in its exact
 * side table.  -1: no replica / no anchor pass */
int64_t fin_index_unverified_kmers(const fin_index* idx, int device);
/* bytes of the k-mer table of the replica on `device` (option "kmer_table"; 0 = none, -1 = no replica there) */
int64_t fin_index_kmer_table_bytes(const fin_index* idx, int device);
/* bytes of the canonical string filter of the replica on `device` (round 4: built with the k-mer table; option "cbf_m"; 0 = none) */
int64_t fin_index_string_filter_bytes(const fin_index* idx, int device);
/* HBM the replica on `device` occupies BEYOND the index arrays (fin_index_size_in_bytes, the reference's size_in_bytes,
 * FinimizerIndex.hh:244-258): every derived table, filter and bitmap the upload built -- prefix, jump, anchor and k-mer tables, string
 * filter, safe-place bitmap, reverse-complement windows.  The command's "bytes:" / bits-per-k-mer lines report both. -1 = no replica there */
int64_t fin_index_replica_table_bytes(const fin_index* idx, int device);
/* the device of the handle's first replica -- the one fin_search / fin_search_batch / fin_batch_create use -- or -1 */
int fin_index_first_device(const fin_index* idx);
/* 1 iff every k-mer of the index has exactly one place in the unitigs: the number of distinct k-mers equals the number of k-mer
 * positions (sum of max(0, length - k + 1)) -- unitigs of a compacted de Bruijn graph, any disjoint spectrum-preserving string set.
 * Informative: what the kernels may take from the text is decided per k-mer at upload (next function). */
int fin_index_is_disjoint(const fin_index* idx);
/* number of k-mer positions of the unitig text that are NOT the place the reference reports for the k-mer they spell (a duplicated k-mer's
 * other places; a k-mer whose finimizer's stored offset belongs to another k-mer), counted on the device when the replica on `device`
 * was uploaded: 0 on a set of disjoint unitigs.  A k-mer found by text comparison at such a place is PRESENT but reported elsewhere: its
 * whole k-mer is looked up (k-mer table, or through the SBWT) and its node's anchor-table entry -- the reference's answer,
 * FinimizerIndex.hh:148-174 -- is used; everywhere else kernels 3 / 4 report it from the text.  -1: no replica there, or options
 * "seed_anchors" and "text_anchors" were both 0 at upload.  fin_index_anchor_build_ms: device time of that pass. */
int64_t fin_index_unsafe_places(const fin_index* idx, int device);
double fin_index_anchor_build_ms(const fin_index* idx, int device);
/* number of k-mers of the unitig text whose reverse complement is in the index too (a k-mer that is its own reverse complement counts),
 * counted on the device when the replica on `device` was uploaded: 0 for a set that holds every canonical k-mer once.  Where it is not 0
 * the upload also marks the windows of 64 text positions such a k-mer ends in, and a strand that reports from one of them has its deferred
 * sister searched in full (option "defer_strand").  -1: not counted (then nothing is deferred). */
int64_t fin_index_rc_pairs(const fin_index* idx, int device);
/* diagnostic (tests): the anchor table of the replica on `device` (option "seed_anchors"): out[2v] = the reference's answer for node v's
 * k-mer (offset in the concatenated unitigs of its last base), 0xFFFFFFFF for nodes that are no k-mer of the unitigs, 0xFFFFFF00 | d for
 * the dummy node that holds d bases; out[2v+1] = the entry's unitig, top bit set when the text at that place does not spell the k-mer
 * (unverified); 2 * n_nodes entries.  FIN_EINVAL if there is none. */
int fin_index_debug_seed_table(const fin_index* idx, int device, uint32_t* out, char* err, size_t errlen);

/* Read-only views of the members FinimizerIndex exposes publicly (FinimizerIndex.hh:108-115), decoded from the
 * HBM layout into plain arrays.  `what` selects the member; out must hold fin_index_export_size(idx, what) bytes. */
#define FIN_X_C 0          /* int64[4]  sbwt C array */
#define FIN_X_PLANE_A 1    /* uint64[ceil(n/64)] bit-plane words, bit i of word i/64 = node i (also _C,_G,_T = 2,3,4) */
#define FIN_X_LCS 5        /* uint8[n_nodes] */
#define FIN_X_FMIN 6       /* uint64[ceil(n/64)] */
#define FIN_X_USTART 7     /* uint64[ceil(n/64)] */
#define FIN_X_GOFF 8       /* int64[n_finimizers] global_offsets */
#define FIN_X_ENDS 9       /* int64[n_unitigs] unitigs.ends */
#define FIN_X_CONCAT 10    /* uint8[total_len] unitigs.concat, one 0..3 code per base */
int64_t fin_index_export_size(const fin_index* idx, int what);
int fin_index_export(const fin_index* idx, int what, void* out, uint64_t out_bytes, char* err, size_t errlen);

/* "The FinimizerIndex loads into HBM once": upload (or re-use) the replica on HIP device `device`.  A handle may hold one
 * replica per device; the first one is the default for fin_search / fin_search_batch / fin_batch_create. */
int fin_index_to_device(fin_index* idx, int device, char* err, size_t errlen);

/* ---- queries ---------------------------------------------------------------------------------------------- */

/* FinimizerIndex::search(const std::string&) (FinimizerIndex.hh:119-185): one strand of one read.
 * pairs_out receives 2*max(0,len-k+1) int64; *n_found = QueryResult::n_found.  Exists for API parity and tests:
 * a single read on a GPU is latency-bound, the batch calls are the fast path. */
int fin_search(const fin_index* idx, const char* seq, int64_t len, int64_t* pairs_out, int64_t* n_found,
               char* err, size_t errlen);

/* The streaming loop run_fmin_queries_streaming (search_fmin.hh:43-72) over host buffers: for every read,
 * search(read), search(rc(read)), merge (forward hit wins, else the reverse strand's hit at len-k-i).
 * strands: FIN_FWD = forward search only (FinimizerIndex::search semantics), FIN_MERGED = the reference loop.
 * pairs_out: 2 int32 per k-mer, reads back to back (read r starts at pair index sum_{q<r} max(0,len_q-k+1)).
 * n_positive (may be NULL) = the reference's "Total found kmers" (search_fmin.hh:61,77). */
#define FIN_FWD 0
#define FIN_MERGED 1
int fin_search_batch(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads,
                     int strands, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen);

/* The same loop with the reference's OUTPUT TEXT as its result: "(u,p) (u,p) ...\n" per read (search_fmin.hh:62-65), made on the GPU
 * next to the pairs and brought back instead of them -- the text is what search-fmin prints, and formatting it on the host is the
 * slowest stage of the whole command.  `out` is a page-locked, growable text buffer owned by the library (fin_text_*); it is
 * valid until the next call with the same buffer.  Every read must have at least one k-mer (FIN_EINVAL otherwise: a shorter read
 * prints an empty line that belongs to no pair -- format such batches with fin_search_batch + fin_format_pairs). */
typedef struct fin_text fin_text;
fin_text* fin_text_create(void);
void fin_text_free(fin_text* t);
int fin_text_reserve(fin_text* t, uint64_t bytes);   /* page-lock room ahead of time (costs about 0.15 s per GB; optional) */
const char* fin_text_data(const fin_text* t);
uint64_t fin_text_size(const fin_text* t);
int fin_search_batch_text(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads, int strands, fin_text* out,
                          uint64_t* n_positive, char* err, size_t errlen);

/* The same loop sharded by record over several GPUs of one node (BASELINE: "reads sharded by record across the 8 GPUs,
 * index replicated, no collective"): uploads a replica to every listed device that has none, cuts the reads into
 * n_devices contiguous shards balanced by bases, runs one host thread per device; pairs_out is in input order. */
int fin_search_batch_multi(fin_index* idx, const int* devices, int n_devices, const char* bases, const uint64_t* offsets,
                           uint64_t n_reads, int strands, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen);
int fin_device_count(void);   /* visible HIP devices (0 without a driver/device) */

/* Pinned (page-locked) host memory for the bases / pairs buffers of fin_search_batch: PCIe copies from and to pinned
 * buffers run at link speed (pageable buffers are staged by the runtime, several times slower).  Optional. */
void* fin_host_alloc(size_t bytes);
void fin_host_free(void* p);

/* Device-resident form of the same loop, for pipelines that keep reads and results in HBM:
 * create uploads the reads (ASCII) once; run enqueues one step -- ingest kernel, probe pre-pass, search kernel -- on `hip_stream` (a hipStream_t, NULL = default
 * stream) without synchronising; results stay in HBM until fin_batch_download / fin_batch_device_pairs. */
int fin_batch_create(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads,
                     fin_batch** out, char* err, size_t errlen);
int fin_batch_create_on(const fin_index* idx, int device, const char* bases, const uint64_t* offsets, uint64_t n_reads,
                        fin_batch** out, char* err, size_t errlen);   /* on a given replica; fin_batch_create uses the first */
/* replace the reads of an existing batch (after its results have been fetched): device buffers are kept and only grow, so a
 * caller streaming read sets of similar size through one batch allocates once.  A failed reload leaves an empty batch. */
int fin_batch_reload(fin_batch* b, const char* bases, const uint64_t* offsets, uint64_t n_reads, char* err, size_t errlen);
int fin_batch_run(fin_batch* b, int strands, void* hip_stream, char* err, size_t errlen);
uint64_t fin_batch_n_kmers(const fin_batch* b);      /* number_of_queries of search_fmin.hh:69 */
uint64_t fin_batch_n_base_strands(const fin_batch* b);
void* fin_batch_device_pairs(const fin_batch* b);    /* device pointer: int32 pairs, layout as pairs_out above */
/* diagnostic (tests of the text formatter): overwrite the batch's pairs in HBM with its n_kmers pairs from `pairs` */
int fin_batch_set_pairs(fin_batch* b, const int32_t* pairs, char* err, size_t errlen);
int fin_batch_download(fin_batch* b, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen);
/* the reference's output text of the batch's pairs, made on the device (every read must have a k-mer); then its download */
int fin_batch_format_text(fin_batch* b, uint64_t* text_bytes, char* err, size_t errlen);
/* Text modes of a batch, for the runs that follow (search_fmin.hh:62-65 prints the pairs and keeps nothing else).  0 (default): pairs.
 * 1: pairs, and the fast path leaves a 32-byte record per read it finishes, from which fin_batch_format_text makes those reads' text
 * without reading their pairs back.  2: text only -- the pairs of such reads are never written (fin_batch_download of pairs returns
 * FIN_EINVAL; the number of found pairs is available after fin_batch_format_text).  fin_search_batch_text runs its batches in mode 2.
 * The text is byte-identical in all three. */
int fin_batch_text_mode(fin_batch* b, int mode);
int fin_batch_download_text(fin_batch* b, char* text_out, char* err, size_t errlen);
/* pairs [first_pair, first_pair + n_pairs) of the batch's output only (ordered behind the most recent run) */
int fin_batch_download_range(fin_batch* b, uint64_t first_pair, uint64_t n_pairs, int32_t* pairs_out, char* err, size_t errlen);
/* Device time of a step (= one fin_batch_run), from HIP events recorded on the stream the step was launched on, averaged over the
 * runs since create/reload after skipping the first `skip_first` (warm-up).  ms_parts[0] = ingest (ASCII -> 2-bit chunks of both
 * strands: the reference's get_rc + base decoding, inside its timed region search_fmin.hh:46-71) + output prefill,
 * [1] = probe pre-pass kernel (kernel 3), [2] = search kernel, [3] = overflow redo + tail, [4] = the whole step. */
int fin_batch_step_time(const fin_batch* b, uint64_t skip_first, double ms_parts[5], uint64_t* n_runs);
/* ms_parts[4] of the above over all runs */
int fin_batch_kernel_time(const fin_batch* b, double* ms_avg, uint64_t* n_runs);
/* diagnostic: reads of the last run that the tuned kernel handed to the overflow kernel (candidate deque beyond its
 * LDS slots, or epoch budget exhausted); waits for that run.  -1 on error. */
int64_t fin_batch_overflow_reads(fin_batch* b);
/* diagnostic, kernel 4: the pipeline's counters of the last run (waits for it): [2] = reads left to kernel 3 (queue slots, some empty),
 * [6+4r] / [7+4r] = stream / anchor+probe queue slots of round r.  Zeros for the other kernels. */
int fin_batch_pipeline_counts(fin_batch* b, uint32_t* out, uint32_t n_words);
/* diagnostic: what the most recent fin_batch_run decided for this batch -- out[0] the kernel that ran, [1] 1 = nothing prefilled the output,
 * [2] 1 = second strands were deferred (option "defer_strand" and the replica's tables allowing), [3] 1 = the pre-pass's fast path was on
 * (option "fast_path"; k <= 63 with the k-mer table and the canonical string filter) */
int fin_batch_run_info(const fin_batch* b, uint32_t out[4]);

/* ---- partitioned indexes (fin_pindex): unitig sets beyond 2^32 nodes (round 5) ----------------------------------------------------------------------
 * The reference counts in int64_t (common.hh:79-93, FinimizerIndex.hh:30-33); one fin_index holds fewer than 2^32 nodes / text bases (FIN_ELIMIT above: a
 * 4.1 Gbp unitig set is the largest measured).  A SET splits the input unitigs, in input order, into parts of at most max_part_bases bases (0: 3.2e9),
 * builds an ordinary index of each on the device, with a replica there,, and searches a read in every part.  Results
 * are those of ONE index of all the unitigs -- pair for pair, the unitig numbers being permute_unitigs' over the whole set (PackedStrings.hh:105-135) --
 * provided the input is what the reference requires (README.md:79-80), a disjoint spectrum-preserving string set: no k-mer, nor its reverse complement,
 * a second time anywhere.  verify != 0 checks exactly that on the device (no part holds a k-mer twice; every part's unitigs searched in the parts behind
 * it) and refuses a set that fails with FIN_EINVAL: which occurrence of a shared k-mer the reference reports depends on the WHOLE index.  A step costs the
 * parts' steps plus a merge pass each.  At most 2^31-1 unitigs. */
typedef struct fin_pindex fin_pindex;
typedef struct fin_pbatch fin_pbatch;
int fin_pindex_build_device(const char* unitig_bases, const uint64_t* unitig_offsets, uint64_t n_unitigs, int k, int device, uint64_t max_part_bases,
                               int verify, fin_pindex** out, char* err, size_t errlen);
void fin_pindex_free(fin_pindex* s);
/* persistence: <prefix>.finparts (a text manifest), <prefix>.p<i>.finamd (every part's container, as fin_index_save) and <prefix>.p<i>.gid (its table of
 * set-wide unitig numbers).  fin_pindex_load uploads every part's replica to `device`; fin_pindex_exists: 1 iff <prefix>.finparts is there */
int fin_pindex_save(const fin_pindex* s, const char* prefix, char* err, size_t errlen);
int fin_pindex_load(const char* prefix, int device, fin_pindex** out, char* err, size_t errlen);
int fin_pindex_exists(const char* prefix);
uint32_t fin_pindex_parts(const fin_pindex* s);
const fin_index* fin_pindex_part(const fin_pindex* s, uint32_t part);   /* owned by the set */
int64_t fin_pindex_k(const fin_pindex* s);
int64_t fin_pindex_n_nodes(const fin_pindex* s);      /* sums over the parts: may pass 2^32 */
int64_t fin_pindex_n_kmers(const fin_pindex* s);
int64_t fin_pindex_n_unitigs(const fin_pindex* s);
int64_t fin_pindex_total_len(const fin_pindex* s);
int64_t fin_pindex_size_in_bytes(const fin_pindex* s);
int64_t fin_pindex_replica_table_bytes(const fin_pindex* s);
int64_t fin_pindex_shared_kmers(const fin_pindex* s);   /* what verify counted (0 for a set that was built with it); -1: not checked */
double fin_pindex_verify_seconds(const fin_pindex* s);
/* out[n] (n = the part's number of unitigs): the set's number of each of the part's unitigs */
int fin_pindex_unitig_ids(const fin_pindex* s, uint32_t part, uint32_t* out, uint64_t n);
/* merged search (search_fmin.hh:46-60) of a flat read set, host buffers, as fin_search_batch with FIN_MERGED (the set keeps its device batches from call
 * to call; one search at a time per set) */
int fin_pindex_search_batch(const fin_pindex* s, const char* bases, const uint64_t* offsets, uint64_t n_reads, int32_t* pairs_out, uint64_t* n_positive,
                               char* err, size_t errlen);
/* device-resident form, as fin_batch_*: run = every part's step and its merge on `hip_stream`; the pairs stay in HBM (fin_pbatch_device_pairs) */
int fin_pbatch_create(const fin_pindex* s, const char* bases, const uint64_t* offsets, uint64_t n_reads, fin_pbatch** out, char* err, size_t errlen);
int fin_pbatch_reload(fin_pbatch* b, const char* bases, const uint64_t* offsets, uint64_t n_reads, char* err, size_t errlen);   /* as fin_batch_reload, every part's batch */
int fin_pbatch_run(fin_pbatch* b, void* hip_stream, char* err, size_t errlen);
uint64_t fin_pbatch_n_kmers(const fin_pbatch* b);
void* fin_pbatch_device_pairs(const fin_pbatch* b);
int fin_pbatch_download(fin_pbatch* b, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen);
int fin_pbatch_step_time(const fin_pbatch* b, uint64_t skip_first, double* ms_avg, uint64_t* n_runs);
void fin_pbatch_free(fin_pbatch* b);

/* ---- results as records (round 5; no reference counterpart: the reference's QueryResult holds a pair per k-mer, FinimizerIndex.hh:30-33) ------------
 * Nine reads in ten of a sequencing run lie in one unitig with a few substitutions; the pair pre-pass finishes them by itself and knows each as 32 bytes
 * (DESIGN.md 4.3).  A caller who takes RECORDS gets those 32 bytes instead of the read's pairs (960 bytes at 150 bp, k = 31) -- and the pairs of the other
 * reads, the ones the pipeline searched, back to back in one stream: what crosses PCIe shrinks eight-fold.  fin_expand_records (host, threads) makes
 * fin_search_batch's pairs from both, bit for bit.
 *   kind = meta >> 16:  0 = the read's nk pairs are the next nk of the stream;  2 = every slot is (-1,-1);
 *                       1 = one run: slot sl of strand A (meta bit 8 set: the reverse strand -- output slot i is strand slot nk-1-i) is (u, off0 + sl) unless
 *                           one of the meta & 0xFF disagreeing positions (16 bits each, ascending, four in Es then four in Es2) lies in [sl, sl + k - 1]: (-1,-1) */
typedef struct fin_read_record { uint32_t u, off0, meta, nk; uint64_t Es, Es2; } fin_read_record;
/* merged search of a flat read set (as fin_search_batch with FIN_MERGED), results as records: recs_out[n_reads]; stream_pairs_out receives the pairs of the
 * kind-0 reads in read order (room for stream_cap_pairs pairs; the input's total number of k-mers always suffices; FIN_ELIMIT if it does not fit),
 * *n_stream_pairs how many came.  Sub-batches are pipelined as in fin_search_batch. */
int fin_search_batch_records(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads, fin_read_record* recs_out,
                             int32_t* stream_pairs_out, uint64_t stream_cap_pairs, uint64_t* n_stream_pairs, char* err, size_t errlen);
/* host: pairs_out[2 * sum of nk] = what fin_search_batch delivers for the same reads; *n_positive (may be NULL) the pairs found.  n_threads <= 0: all
 * cores.  FIN_EINVAL: records and stream do not belong together (the records ask for another number of stream pairs) */
int fin_expand_records(const fin_read_record* recs, uint64_t n_reads, const int32_t* stream_pairs, uint64_t n_stream_pairs, int k, int32_t* pairs_out,
                       uint64_t* n_positive, int n_threads);
/* the same for a resident batch: after fin_batch_run (text mode 2 leaves the fast path's records; any other run: every read is kind 0 and the stream is
 * the batch's pairs) fin_batch_records gathers the stream on the device and says how long it is; fin_batch_download_records copies both to the host */
int fin_batch_records(fin_batch* b, uint64_t* n_stream_pairs, char* err, size_t errlen);
int fin_batch_download_records(fin_batch* b, fin_read_record* recs_out, int32_t* stream_pairs_out, char* err, size_t errlen);

/* diagnostic (tests): the compact k-mer table of the replica on `device` asked about n k-mers, each given as its two key words (2-bit codes A=0 C=1 G=2 T=3, first
 * base in the low bits; k0 = bases 0..31, k1 = bases 32..k-1, 0 for k <= 32): out[2 i] = the answer g the table claims, out[2 i + 1] = flags -- 0 no claim (the
 * k-mer is in no unitig), 1 a verified claim, 2 an unverified one (| 8: the exact side table has the k-mer, g is its answer), | 4 the text at [g-k+1, g] spells
 * the k-mer.  FIN_EINVAL: no k-mer table there */
int fin_index_debug_kmer_table(const fin_index* idx, int device, const uint64_t* k0, const uint64_t* k1, uint64_t n, uint32_t* out, char* err, size_t errlen);

/* diagnostic (tests): drives the epoch kernels' read-chunk cache through "a load of the current chunk under way, then the next chunk asked for" on the
 * device (the hazard fixed in round 4: the next chunk must not be promoted while that load is pending).  FIN_OK and *fail_bits == 0: every step behaved */
int fin_debug_chunk_cache_selftest(uint32_t* fail_bits);
void fin_batch_free(fin_batch* b);

/* The reference's output text for n_pairs results of one read: "(u,p) (u,p) ...\n" (search_fmin.hh:62-65).
 * Returns the number of bytes written (no NUL).  out must hold 24*n_pairs+2 bytes. */
int64_t fin_format_pairs(const int32_t* pairs, int64_t n_pairs, char* out);

#ifdef __cplusplus
}
#endif
#endif
