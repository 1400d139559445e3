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
#define FIN_LCS_MASK 0x3Fu      // node byte bits 0-5: LCS (k <= 64)
#define FIN_USTART_BIT 0x40u    // node byte bit 6: Ustart[i]
#define FIN_FMIN_BIT 0x80u      // node byte bit 7: fmin[i]
#define FIN_MAX_K 64

struct alignas(128) FinNodeBlock {
    uint8_t node[64];       // per node: LCS | Ustart<<6 | fmin<<7
    uint64_t plane[4];      // A,C,G,T outgoing-edge marks of the 64 nodes
    uint32_t base[4];       // C[c] + rank_c(64*b): start of the target interval for an extend from this block
    uint32_t ustart_rank;   // ones of Ustart before this block
    uint32_t fmin_rank;     // ones of fmin before this block
    uint32_t pad[2];
};
static_assert(sizeof(FinNodeBlock) == 128, "one block = one 128-B line");

// What a kernel needs to know about the index (passed by value).
struct FinDevIndex {
    const FinNodeBlock* blocks;
    const uint32_t* goff;        // global_offsets in fmin-rank order
    const uint32_t* ends;        // exclusive unitig ends in the concatenation
    const uint32_t* samp;        // samp[g >> samp_shift] = number of ends <= (g >> samp_shift) << samp_shift
    const uint32_t* concat;      // 2-bit packed unitig text, 16 bases per word, base i at bits 2*(i&15)
    uint32_t n_nodes;
    uint32_t n_unitigs;
    uint32_t total_len;
    uint32_t k;
    uint32_t samp_shift;
    uint32_t n_samp;
    uint32_t C[5];               // C[0..3], C[4] = n_nodes
};

// Container file <prefix>.finamd
#define FIN_MAGIC 0x31444d414e4946ull   // "FINAMD1"
struct FinFileHeader {
    uint64_t magic;
    uint32_t version;
    uint32_t k;
    uint64_t n_nodes, n_kmers, n_unitigs, total_len, n_fmin;
    uint64_t C[4];
    uint32_t samp_shift, n_samp;
    uint64_t n_blocks, n_concat_words;
    uint64_t reserved[4];
};
