/*
 * finito_synth.h -- seeded synthetic inputs and a ground-truth checker: TOOLING exported by libfinito_amd.so for bench.py and the tests
 * (finito_amd/synth.py binds them).  NOT part of the drop-in boundary (include/finito_amd.h): nothing here has a counterpart in the
 * reference, whose tree ships no generator (SURVEY.md 8d prescribes the inputs; finito_amd/csrc/fin_synth.cpp makes them).
 * Plain C, like the boundary itself.
 */
#ifndef FINITO_SYNTH_H
#define FINITO_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* genome: n bases iid uniform ACGT (splitmix64-seeded xoshiro256**) */
void fin_synth_genome(uint64_t n, uint64_t seed, char* out);
/* the same with repeats written over it: interspersed families in both orientations whose copies diverged by div_lo..div_hi, tandem
 * arrays, segmental duplications -- about repeat_frac of the bases */
void fin_synth_repeat_genome(uint64_t n, uint64_t seed, double repeat_frac, double div_lo, double div_hi, char* out);
/* unitigs: the genome cut into pieces of uniform length [k, max_len] overlapping by k-1, each reverse-complemented with p = 1/2, shuffled.
 * Returns the number of pieces, or -(needed) if a capacity is too small. */
int64_t fin_synth_unitigs(const char* genome, uint64_t n, int k, uint32_t max_len, uint64_t seed, char* out_bases, uint64_t out_cap,
                          uint64_t* out_offsets, uint64_t* piece_gstart, uint32_t* piece_glen, uint8_t* piece_rc, int64_t cap_pieces);
/* a DISJOINT spectrum-preserving string set (k <= 32): every canonical k-mer at its first occurrence only; dup_pos / dup_first: the k-mer
 * starts that are not a first occurrence, ascending, each with that first occurrence; multi (n bytes, may be null): k-mer starts whose
 * canonical k-mer occurs more than once.  Returns the number of pieces; negative if a capacity is too small (*n_dups is set; -np with
 * out_offsets[0] = bases needed). */
int64_t fin_synth_spss(const char* genome, uint64_t n, int k, uint32_t max_len, uint64_t seed, char* out_bases, uint64_t out_cap,
                       uint64_t* out_offsets, uint64_t* piece_gstart, uint32_t* piece_glen, uint8_t* piece_rc, int64_t cap_pieces,
                       uint32_t* dup_pos, uint32_t* dup_first, uint64_t cap_dups, uint64_t* n_dups, uint8_t* multi);
/* reads: start uniform, fixed length, strand p = 1/2, iid substitutions, a fraction of fully random reads (read_gstart -1) */
void fin_synth_reads(const char* genome, uint64_t n, uint64_t n_reads, uint32_t read_len, double err_rate, double random_frac,
                     uint64_t seed, char* out_bases, uint64_t* out_offsets, int64_t* read_gstart, uint8_t* read_rc, uint8_t* err_mask);
/* ... records [first_record, first_record + n_reads) of that set (a record depends on its number only: the set is the same however it is cut) */
void fin_synth_reads_at(const char* genome, uint64_t n, uint64_t first_record, uint64_t n_reads, uint32_t read_len, double err_rate, double random_frac,
                        uint64_t seed, char* out_bases, uint64_t* out_offsets, int64_t* read_gstart, uint8_t* read_rc, uint8_t* err_mask);
/* ground truth at any size: every error-free k-mer of a genome-derived read must localize to the piece that holds it (check2: to the
 * piece that holds its first occurrence; skip = k-mer starts not checked).  Returns the number of wrong pairs. */
int64_t fin_synth_check(uint64_t n_pieces, const uint64_t* piece_gstart, const uint32_t* piece_glen, const uint8_t* piece_rc,
                        const uint32_t* unitig_id, int k, uint64_t n_reads, uint32_t read_len, const int64_t* read_gstart,
                        const uint8_t* read_rc, const uint8_t* err_mask, const int32_t* pairs, uint64_t* n_checked, int64_t* first_bad_read);
int64_t fin_synth_check2(uint64_t n_pieces, const uint64_t* piece_gstart, const uint32_t* piece_glen, const uint8_t* piece_rc,
                         const uint32_t* unitig_id, int k, uint64_t n_reads, uint32_t read_len, const int64_t* read_gstart,
                         const uint8_t* read_rc, const uint8_t* err_mask, const int32_t* pairs, const uint32_t* dup_pos,
                         const uint32_t* dup_first, uint64_t n_dups, const uint8_t* skip, uint64_t* n_checked, int64_t* first_bad_read);

#ifdef __cplusplus
}
#endif
#endif
