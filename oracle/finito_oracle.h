/*
 * finito_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's search-fmin localization path and of the index
 * construction it depends on.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use anything under oracle/.  The product (finito_amd/) never links or calls this.
 *
 * Parity status: PINNED by the reference's own 9 known-answer tests (src/tests.cpp:62-317), committed as
 * tests/golden/reference_kat.json and checked by tests/test_oracle_golden.py.  The reference binary itself
 * cannot be built here (its SBWT/sdsl-lite submodule is an empty directory), so "bit-exact vs reference"
 * everywhere in this repo means "bit-exact vs this restatement, which reproduces every reference vector".
 *
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef FINITO_ORACLE_H
#define FINITO_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fo_index fo_index;

/* Operation counters used to derive the ALGORITHMIC bytes of SURVEY.md section 8(d). */
typedef struct fo_counters {
    int64_t base_strands;   /* bases processed by rarest_fmin_streaming_search (both strands) */
    int64_t kmers;          /* reported k-mers: sum over reads of max(0, len-k+1) (search_fmin.hh:69) */
    int64_t found;          /* reported k-mers with unitig != -1 after the strand merge */
    int64_t extends;        /* update_sbwt_interval calls that reach the rank structure */
    int64_t rank_lines;     /* distinct 512-bit plane blocks touched, summed over extends */
    int64_t drops;          /* drop_first_char calls that scan LCS */
    int64_t lcs_entries;    /* LCS entries read */
    int64_t lcs_lines;      /* distinct 64-entry LCS lines touched, summed over drop calls */
    int64_t anchors;        /* dictionary lookups (branch or finimizer) */
    int64_t walked;         /* hits produced by walk_in_unitigs */
    int64_t max_deque;      /* max live BoundedDeque size seen */
    int64_t max_deque_eager;/* max live size if stale entries were popped eagerly */
} fo_counters;

/* ---- construction (SBWT semantics SURVEY 8a-5; lcs_basic_parallel_algorithm.hpp:52-120;
 *      PackedStrings.hh:105-135; FinimizerIndex.hh:273-389) ---- */
fo_index* fo_build(const char* bases, const uint64_t* offsets, int64_t n_unitigs, int k);

/* Assemble an index from raw components (used at sizes where fo_build would take too long; the
 * components then come from the product builder, whose equality with fo_build is tested separately). */
fo_index* fo_from_components(int k, int64_t n_nodes, const uint64_t* const planes[4], const uint8_t* lcs,
                             const uint64_t* fmin_bits, const uint64_t* ustart_bits, const int64_t* goff,
                             int64_t n_fmin, const uint8_t* concat_codes, int64_t total_len,
                             const int64_t* ends, int64_t n_unitigs);
void fo_free(fo_index*);

int64_t fo_k(const fo_index*);
int64_t fo_n_nodes(const fo_index*);
int64_t fo_n_kmers(const fo_index*);
int64_t fo_n_unitigs(const fo_index*);
int64_t fo_n_fmin(const fo_index*);
int64_t fo_total_len(const fo_index*);
int64_t fo_size_in_bytes(const fo_index*);
void fo_get_C(const fo_index*, int64_t out[4]);
void fo_get_plane(const fo_index*, int c, uint8_t* out_one_byte_per_node);
void fo_get_lcs(const fo_index*, uint8_t* out);
void fo_get_fmin(const fo_index*, uint8_t* out);
void fo_get_ustart(const fo_index*, uint8_t* out);
void fo_get_goff(const fo_index*, int64_t* out);
void fo_get_ends(const fo_index*, int64_t* out);
void fo_get_concat(const fo_index*, uint8_t* out_codes);
/* node label of node i as k chars over $ACGT (only for indexes made by fo_build) */
int fo_get_label(const fo_index*, int64_t i, char* out_k_chars);

/* ---- query path ---- */
/* FinimizerIndex::search (FinimizerIndex.hh:119-185). pairs_out has room for 2*max(0,len-k+1) int64.
 * Returns the number of pairs written; *n_found as QueryResult::n_found. */
int64_t fo_search(const fo_index*, const char* q, int64_t len, int64_t* pairs_out, int64_t* n_found,
                  fo_counters* ctr);
/* search(read), search(rc(read)), merge (search_fmin.hh:47-60). Returns number of pairs. */
int64_t fo_search_merged(const fo_index*, const char* q, int64_t len, int64_t* pairs_out, fo_counters* ctr);
/* Streaming loop over a batch (search_fmin.hh:43-72). pairs_out may be NULL (timing only); if non-NULL it
 * receives the merged pairs of all reads back to back, as int64 (u,p).  format_text != 0 also formats the
 * "(u,p) (u,p)\n" text (into a scratch buffer) as the reference's timed region does.  Returns seconds spent
 * in the timed region.  n_threads > 1 shards reads over OpenMP threads (the reference is single-threaded). */
double fo_search_batch(const fo_index*, const char* bases, const uint64_t* offsets, int64_t n_reads,
                       int64_t* pairs_out, int format_text, int n_threads, fo_counters* ctr,
                       uint64_t* text_checksum);
/* build-fmin --type shortest (type 1, build_fmin.hh:134-214) / verify (type 2, :95-132, :257-268) with frequency threshold t over
 * the given sequences: out[0] = number of distinct {length, frequency, colex} finimizers, out[1] = sum of frequencies,
 * out[2] = sum of lengths (what print_finimizer_stats, common.hh:188-206, reports).  -1 if a sequence leaves the index. */
int fo_finimizer_stats(const fo_index*, const char* bases, const uint64_t* offsets, int64_t n_seqs, int type, int64_t t, int64_t out[3]);
/* ---- the LAZY algorithm of the product's default kernels, restated (finito_lazy.c; CHANGELOG.md 4.6) ----
 * Same pairs as fo_search_batch, by construction of the algorithm -- tests assert it -- with far fewer index accesses.
 * ALGORITHMIC BYTES of a step (what bench.py's roofline.achieved is built from), per read set:
 *     128 * (probe_lines + stream_lines)     node blocks: the 128-byte line that holds the LCS bytes, the four rank records and the
 *                                            thermometer planes of 64 nodes (finito_amd/csrc/fin_format.h).  Counted per unit of
 *                                            work -- one probe extend, one streamed base -- as the DISTINCT blocks it touches that the
 *                                            previous unit of the same strand did not touch (a lane keeps one step's data in registers)
 *   +   8 * (table_entries + jump_entries)   prefix-table lookups of the probes; jump-table lookups of the (re)starts
 *   +  40 * anchors                          dictionary lookups: 16 B block record + 4 B offset + 4 B sample + 16 B unitig ends
 *   +  16 * seed_lookups + 8 * seed_verdicts seeds: one 16 B seed-table entry (place, unitig, its bounds); seed node written + read with a verdict
 *   +  16 * text_windows                     64-base windows of 2-bit unitig text compared by walks
 *   +  32 * ktab_lookups                     one bucket of the compact k-mer table per whole k-mer asked (round 5: four 8-byte slots {answer, tag}; k <= 63)
 *   +   8 * safe_checks                      one word of the 'reported here' bitmap per k-mer placed by text comparison (indexes with unsafe places only)
 *   +  16 * (chunks_probe + chunks_search)   packed read chunks (32 bases) loaded by the pre-pass / by the search kernel
 *   +   8 * filter_checks                    pre-pass: two words of the absence filter per check
 *   +   8 * strands + 16 * reads             pre-pass verdict written + read per strand; read descriptor
 *   +       bases + 16 * chunks_packed       ingest: ASCII in, 2-bit chunks of both strands out
 *   +   8 * kmers                            one (unitig, offset) pair per k-mer
 *   +  16 * (fast_chunks + fast_cbf + fast_redesc) + 32 * (fast_looks + fast_looks2) + 8 * fast_text_words + 20 * fast_tries     the pre-pass's fast path (round 4; a look = one bucket)
 *   +  16 * fbf_lookups + 36 * place_anchors     lean tables (round 4); a claimed anchor: the locate + 16 bytes of text the k-mer is compared with (round 5)
 * Payload bytes only (no line rounding for the small records), nothing counted twice: a lower bound of what the step must move. */
typedef struct fo_lazy_counters {
    int64_t reads, strands, strands_searched;   /* strands_searched: not ruled out entirely by the probe pre-pass */
    int64_t bases, kmers, found, chunks_packed;
    int64_t table_entries, probe_extends, probe_lines, chunks_probe;
    int64_t stream_steps, stream_lines, chunks_search;
    int64_t anchors, walk_bases, text_windows;
    int64_t restarts_short, restarts_failed_check, restarts_k1, restarts_full_margin, restarts_margin;
    int64_t jump_entries, jumped_bases;   /* (re)starts that looked the jump table up; bases they did not have to stream */
    int64_t text_anchors;                 /* k-mers placed by comparing them with the unitig text behind a sequencing error (disjoint indexes) */
    int64_t prepass_entries, prepass_lines;   /* the share of table_entries / probe_lines spent by the probe pre-pass (its own kernel on the device) */
    int64_t filter_checks;  /* pre-pass: pairs of absence-filter look-ups (two 4-byte words of a bit set that stays in cache) */
    int64_t full_anchors;   /* anchors from a look-up of the whole k-mer (a probe string that was not unique) */
    int64_t seed_lookups, seed_anchors, seed_verdicts;   /* seeds: places looked up in the seed table; k-mers found there; pre-pass verdicts that carry a seed slot */
    int64_t unsafe_places;  /* k-mers found in the text at a place the reference does not report for them (non-disjoint indexes): left to the streaming search */
    int64_t safe_checks;    /* look-ups of the per-position 'reported here' bit (8 bytes of the bitmap; only counted when the index has any unsafe place) */
    int64_t ktab_lookups;   /* look-ups of the k-mer table (32 bytes: one bucket of four slots {answer, tag} of the hash table over the text's k-mers) */
    int64_t deferred_strands, deferred_slots;   /* second strands searched only where the first left slots open; the slots of those stretches */
    /* where the probe work of the search goes (shares of probe_lines / table_entries; diagnostics, not in the byte model a second time) */
    int64_t full_lookups, full_lines, full_entries;       /* look-ups of a whole k-mer (a probe string that occurs more than once) */
    int64_t bridge_lines, bridge_entries;                 /* probes across a bad position */
    int64_t uend_lines, uend_entries, uend_probes;        /* probes for the next k-mer end after a unitig ended */
    int64_t prepass_ktab;   /* the share of ktab_lookups spent by the pre-pass: the first k-mer of a strand asked for directly (deferred second strand, k <= 31) */
    /* THE FAST PATH of the pair pre-pass (round 4; flags bit 6): a read that lies in one unitig with a few substitutions is finished by the
     * pre-pass itself -- whole read against the text behind the place of one of its k-mers, the k-mer ends across a disagreeing base proven
     * absent on both strands by strings the canonical string filter does not know (finito_lazy.c, lz_fast_read).  Its own byte terms: */
    int64_t fast_reads, fast_absent_reads;   /* reads it finished; of them, reads proven absent altogether (no k-mer of either strand in the index) */
    int64_t fast_tries;      /* comparisons with the text begun (a place found; one locate each: 4-byte sample + 16 bytes of unitig ends) */
    int64_t fast_looks;      /* k-mer table buckets asked beyond the two first-k-mer looks (last / middle k-mers): 32 bytes each */
    int64_t fast_chunks;     /* packed chunks loaded by looks beyond the first, comparisons and the all-absent proof: 16 bytes each */
    int64_t fast_text_words; /* 32-base words of unitig text compared with: 8 bytes each */
    int64_t fast_cbf;        /* blocks of the canonical string filter asked: 16 bytes each */
    int64_t fast_redesc;     /* read descriptors loaded again by the later phases: 16 bytes each */
    int64_t fast_looks2;     /* 32 <= k <= 63 with round 3's tables: buckets of the k-mer table asked (every look of the fast path): 32 bytes each */
    /* LEAN TABLES (round 4; flags bit 7; the device's default for k <= 31): no prefix table, no anchor table -- a probe asks the directional
     * string filter about a string of m bases (one 16-byte block), a string that occurs is followed by a look-up of the whole k-mer in the
     * k-mer table, the pre-pass's seeds are places */
    int64_t fbf_lookups, prepass_fbf;   /* blocks of the directional string filter asked: 16 bytes each; the pre-pass's share */
    int64_t place_anchors;   /* anchors whose place came with a k-mer-table slot (a look's, or the walk kernel's own look-up): the locate -- 4-byte sample + 16 bytes of unitig ends -- and the 16 bytes of text its k-mer is compared with */
} fo_lazy_counters;
/* pairs_out (may be NULL): merged pairs of all reads back to back, int64 (u,p).  ptab_t = depth of the probes' prefix table, jump_t =
 * depth of the jump table of the (re)starts (the device replica's: fin_index_prefix_table_depth, fin_index_jump_table_depth;
 * 0 = none).  Returns the number of pairs. */
/* flags bit 0: text re-anchoring behind sequencing errors (a k-mer found by comparing the read with the unitig text is reported there
 * iff that is the place the reference reports for it -- checked per k-mer, so any index qualifies); bit 1: seeds -- a strand's anchors
 * come from unique probe strings and the reference's answer for their node's k-mer wherever that answer is a place of the k-mer, not
 * from the streaming search (finito_lazy.c, lz_strand); bit 2: count safe_checks (the index has unsafe places: the device reads the
 * bitmap); bit 3: the k-mer table (k <= 31) is asked instead of a look-up of the whole k-mer; bit 4: the second strand of a read is
 * deferred (finito_lazy.c, lz_read); bit 5: with bit 4 -- the index has k-mers whose reverse complement is in it too (not fo_index_rc_free): a first
 * strand that reports from a text window with such a k-mer has its sister searched in full; bit 6: with bits 1, 3, 4 and k <= 31 -- the pre-pass's FAST PATH
 * (lz_fast_read; the canonical string filter is built for the call); bit 7: LEAN TABLES -- probes ask the directional string filter (exact
 * occurrence of m-base strings here), no seeds by node, the pre-pass's seeds are places; ptab_t is taken as 0; bits 8..15: depth F of the pre-pass's absence filter (0: none). */
int64_t fo_search_batch_lazy(const fo_index*, const char* bases, const uint64_t* offsets, int64_t n_reads, int64_t* pairs_out,
                             int ptab_t, int jump_t, int flags, int n_threads, fo_lazy_counters* ctr);
/* 1 iff no k-mer of the unitigs has its reverse complement among them too (O(text length * k): small indexes) */
int fo_index_rc_free(const fo_index*);
/* 1: every place the reference reports for a k-mer of the text spells that k-mer (finito_lazy.c) */
int fo_index_all_verified(const fo_index*);
/* 1 iff the number of distinct k-mers equals the number of k-mer positions in the unitigs (sum of max(0, length - k + 1)) */
int fo_index_is_disjoint(const fo_index*);
/* text of one read in the reference's output format; returns bytes written (no NUL) */
int64_t fo_format_pairs(const int64_t* pairs, int64_t n_pairs, char* out);

#ifdef __cplusplus
}
#endif
#endif
