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
/* text of one read in the reference's output format; returns bytes written (no NUL) */
int64_t fo_format_pairs(const int64_t* pairs, int64_t n_pairs, char* out);

#ifdef __cplusplus
}
#endif
#endif
