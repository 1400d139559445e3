// fin_kernels.h -- host-callable launchers of the HIP kernels (implemented in fin_kernels.hip)
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "fin_format.h"

#ifdef __cplusplus
extern "C" {
#endif
int fin_launch_search_v0(const FinDevIndex* ix, const uint8_t* bases, const uint64_t* offs, const uint64_t* out_offs,
                         void* out, uint32_t n_reads, int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                         uint64_t* ovf_scratch, uint32_t ovf_blocks, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
int fin_launch_overflow(const FinDevIndex* ix, const uint8_t* bases, const uint64_t* offs, const uint64_t* out_offs, void* out,
                        int strands, const uint32_t* ovf_list, const uint32_t* ovf_count, uint64_t* ovf_scratch, uint32_t ovf_blocks,
                        hipStream_t stream);
int fin_launch_pack_reads(const uint8_t* bases, const uint64_t* offs, const FinReadDesc* desc, void* packed, uint32_t n_reads, uint64_t n_chunks, hipStream_t stream);
int fin_launch_search_v2(const FinDevIndex* ix, const uint8_t* bases, const void* packed, const FinReadDesc* desc,
                         const uint64_t* offs, const uint64_t* out_offs, void* out, uint64_t n_kmers, uint32_t n_reads,
                         int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                         uint32_t* work_counter, uint64_t* ovf_scratch, uint32_t ovf_blocks, uint32_t grid_blocks,
                         hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);
int fin_v2_blocks_per_cu(void);
// v3 = v2 + walk mode, cold restart and probing (fin_kernel_v3.hip); same arguments
int fin_launch_search_v3(const FinDevIndex* ix, const uint8_t* bases, const void* packed, const FinReadDesc* desc,
                         const uint64_t* offs, const uint64_t* out_offs, void* out, uint64_t n_kmers, uint32_t n_reads,
                         int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                         uint32_t* work_counter, uint64_t* ovf_scratch, uint32_t ovf_blocks, uint32_t grid_blocks,
                         uint32_t* pass /* 2 * n_reads + 4 words for the probe pre-pass, or NULL: probe inside the search kernel */,
                         uint32_t grid_blocks_probe, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev_mid /* between pre-pass and search */);
int fin_v3_blocks_per_cu(void);
// single-stage launchers used by kernel 4's pipeline (fin_kernel_w.hip)
int fin_launch_probe_stage(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t n_reads, int strands, uint32_t* pass,
                           uint32_t* seed /* 2 * n_reads + 4 words: the seed node of every verdict, or NULL */, uint32_t* work_counter, uint32_t grid_blocks,
                           void* fast_out /* the batch's pairs when the fast path may write them (nothing prefills the output), else NULL */, uint32_t* n_fast, hipStream_t stream);
int fin_launch_stream_stage(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t lds_deque_limit, uint32_t* ovf_list,
                            uint32_t* ovf_count, uint32_t* work_counter, const void* items_in, const uint32_t* n_in, void* items_out,
                            uint32_t* n_out, uint32_t grid_blocks, hipStream_t stream);
int fin_launch_v3_list(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, void* out, int strands, uint32_t lds_deque_limit,
                       uint32_t* ovf_list, uint32_t* ovf_count, uint32_t* work_counter, const uint32_t* pass, const uint32_t* read_list,
                       const uint32_t* n_list, uint32_t grid_blocks, hipStream_t stream);
// the pair pre-pass (fin_prepass.hip): verdicts and seeds of both strands of every read; defer: one of them FIN_PASS_DEFERRED where possible
// out (may be NULL): the batch's pairs -- with it, a read the FAST PATH finishes (whole read against one unitig's text, gaps proven absent by
// the canonical string filter) is written here and gets the verdict FIN_PASS_DONE on both strands; n_fast (may be NULL): how many
int fin_launch_pair_prepass(const FinDevIndex* ix, const void* packed, const FinReadDesc* desc, uint32_t n_reads, uint32_t* pass, uint32_t* seed,
                            int defer, uint32_t grid_hint, void* out, uint32_t* n_fast, hipStream_t stream);
int fin_stream_blocks_per_cu(void);
void fin_debug_dump_time(void);   // -DFIN_V3_TIME builds: per-segment wave-cycle shares to stderr
int fin_walk_blocks_per_cu(void);
uint32_t fin_v4_counter_words(void);
uint64_t fin_v4_queue_slots(uint32_t n_reads, uint32_t max_grid_blocks);
uint64_t fin_v4_list_slots(uint32_t n_reads, uint32_t max_grid_blocks);
uint64_t fin_v4_workspace_bytes(uint32_t n_reads, uint32_t max_grid_blocks);
// kernel 4 = the pipeline probe -> route -> (stream -> walk) x rounds -> kernel 3 on what is left (fin_kernel_w.hip)
int fin_launch_search_v4(const FinDevIndex* ix, const uint8_t* bases, const void* packed, const FinReadDesc* desc,
                         const uint64_t* offs, const uint64_t* out_offs, void* out, uint64_t n_kmers, uint32_t n_reads,
                         int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                         uint64_t* ovf_scratch, uint32_t ovf_blocks, uint32_t* pass, uint32_t* seed /* as pass, or NULL */, void* ws /* fin_v4_workspace_bytes */, uint64_t q_slots /* fin_v4_queue_slots */,
                         uint32_t* ctr /* fin_v4_counter_words() u32 */, uint32_t grid_probe, uint32_t grid_stream, uint32_t grid_walk, uint32_t grid_v3,
                         hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev_mid,
                         hipEvent_t out_ready /* NULL: the launcher prefills the output itself; else the prefill is done when this event fires */,
                         int no_prefill /* 1 (only if fin_v4_writes_gaps): nobody prefills, the pipeline writes every slot itself */,
                         uint32_t rounds /* stream / walk rounds to launch, 1 .. fin_v4_max_rounds() */);
uint32_t fin_v4_max_rounds(void);
int fin_v4_writes_gaps(const FinDevIndex* ix, const uint32_t* seed);
int fin_probe_blocks_per_cu(void);
// fills the prefix table of depth T (4^T entries) from the uploaded node blocks
int fin_launch_build_ptab(const FinDevIndex* ix, void* tab, int T, hipStream_t stream);
// fills the anchor table pos[n_nodes + 1] (FinDevIndex::pos) and the safe-place bitmap safe[fin_anchor_safe_words()] (FinDevIndex::safe)
// from the uploaded index (fin_kernel_b.hip); tmp: fin_anchor_tmp_bytes() of device scratch; *n_unsafe: k-mer positions of the text that
// are not the place the reference reports for their k-mer (0: the bitmap is all ones and can be dropped).  Synchronises the stream.
uint64_t fin_anchor_safe_words(uint64_t total_len);
uint64_t fin_anchor_tmp_bytes(uint64_t total_len);
// kt3 (may be null; k <= 63): the compact k-mer table, kt3_buckets buckets of 32 bytes, filled by the same pass
int fin_launch_build_anchors(const FinDevIndex* ix, struct FinSeedEntry* pos, void* safe, void* kt3, uint32_t kt3_buckets, void* tmp, uint64_t* n_unsafe, hipStream_t stream, uint64_t* n_unver);
// the k-mers with an unverified answer, listed by that pass inside tmp (room for fin_anchor_ulist_cap() of them), and the exact side table made from the list
uint32_t fin_anchor_ulist_cap(uint64_t total_len);
void* fin_anchor_ulist(void* tmp, uint64_t total_len);
int fin_launch_build_ktx(const void* ulist, uint32_t n, void* ktx, uint32_t log2, hipStream_t stream);
// counts the k-mers of the text whose reverse complement is in the index too (fin_kernel_b.hip); tmp8: 8 bytes of device scratch.  Synchronises.
uint64_t fin_rcwin_bytes(uint64_t total_len);
int fin_launch_count_rc_pairs(const FinDevIndex* ix, void* tmp8, uint64_t* n_pairs, void* rcwin, hipStream_t stream);
// fills the canonical string filter (FinDevIndex::cbf; fin_kernel_b.hip): 2^log2_blocks blocks of 16 bytes over the unitigs' strings of m bases (m <= 32)
// (words_f, may be NULL: the directional filter FinDevIndex::fbf, same size, filled by the same pass)
int fin_launch_build_cbf(const FinDevIndex* ix, void* words, void* words_f, uint32_t log2_blocks, uint32_t m, hipStream_t stream);
// fills the absence filter filt[4^F / 32 + 8] (FinDevIndex::filt) from the uploaded text
int fin_launch_build_filter(const FinDevIndex* ix, uint32_t* filt, int F, hipStream_t stream);
int fin_launch_count_positive(const void* out, uint64_t n_pairs, unsigned long long* d_result, hipStream_t stream);
// index sets: a part's pairs into the set's result, unitigs renumbered by gid[] (fin_records.hip)
int fin_launch_set_merge(void* dst, const void* src, const uint32_t* gid, uint64_t n_pairs, int first, hipStream_t stream);
uint32_t fin_overflow_deque_cap(void);
// the reference's output text on the device (fin_text.hip)
uint32_t fin_text_blocks(uint64_t n_pairs);
uint64_t fin_text_off_words(uint64_t n_pairs);   // u64 words of d_blk_off
int fin_launch_text_lengths(const void* pairs, uint64_t n_pairs, const uint64_t* out_offs, uint32_t n_reads, uint32_t* d_last_bits,
                            uint32_t* d_blk_sum, uint64_t* d_blk_off, uint64_t* d_total, hipStream_t stream);
int fin_launch_text_write(const void* pairs, uint64_t n_pairs, const uint64_t* d_blk_off, const uint32_t* d_last_bits, char* d_text, hipStream_t stream);
uint64_t fin_text3_off_words(uint64_t n_seg);
uint32_t fin_text3_seg_pairs(void);
int fin_launch_text3_lengths(const void* pairs, const uint64_t* out_offs, const void* frec, const void* seg, uint32_t n_seg, uint32_t k,
                             uint32_t* d_seg_sum, uint64_t* d_seg_off, uint64_t* d_total, unsigned long long* d_found, hipStream_t stream);
int fin_launch_text3_write(const void* pairs, const uint64_t* out_offs, const void* frec, const void* seg, uint32_t n_seg, uint32_t k,
                           const uint64_t* d_seg_off, char* d_text, hipStream_t stream);
// diagnostic: the compact k-mer table asked about n k-mers {k0[i], k1[i]}: out[i] = {g, flags} (fin_kernels.hip)
int fin_launch_kt3_query(const FinDevIndex* ix, const uint64_t* k0, const uint64_t* k1, uint32_t n, void* out, hipStream_t stream);
// fin_records.hip: the pairs of the reads whose fast-path record stayed zero gathered into one dense stream (read order kept), their records stamped with nk
uint32_t fin_rec_blocks(uint32_t n_reads);
int fin_launch_rec_count(const void* frec, const uint64_t* out_offs, uint32_t n_reads, uint32_t* blk_sum, uint64_t* blk_off, uint64_t* total, hipStream_t stream);
int fin_launch_rec_compact(void* frec, const uint64_t* out_offs, const void* pairs, uint32_t n_reads, const uint64_t* blk_off, void* stream_out, hipStream_t stream);
#ifdef __cplusplus
}
#endif
