// fin_device.h -- device helpers shared by the gfx950 kernels (index primitives over the 128-B node blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fin_format.h"

#define FIN_TPB 256

// ---- device-side index primitives ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t d_nodebyte(const FinDevIndex& ix, uint32_t i) {
    return ((const uint8_t*)ix.blocks)[(size_t)(i >> 6) * sizeof(FinNodeBlock) + (i & 63)];
}
__device__ __forceinline__ uint32_t d_lcs(const FinDevIndex& ix, uint32_t i) { return d_nodebyte(ix, i) & FIN_LCS_MASK; }

// update_sbwt_interval (formula: common.hh:26-36) on [l, r]; false = (-1,-1)
__device__ __forceinline__ bool d_extend(const FinDevIndex& ix, uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr) {
    const FinNodeBlock* bl = ix.blocks + (l >> 6);
    const FinNodeBlock* br = ix.blocks + (r >> 6);
    uint32_t ol = l & 63, orr = r & 63;
    uint64_t ml = ol == 0 ? 0ull : (~0ull >> (64 - ol));
    uint64_t mr = ~0ull >> (63 - orr);
    nl = bl->rec[c].base + (uint32_t)__popcll(fin_plane(bl->rec[c]) & ml);
    uint32_t re = br->rec[c].base + (uint32_t)__popcll(fin_plane(br->rec[c]) & mr);   // exclusive end
    nr = re - 1;
    return nl < re;
}

// drop_first_char (common.hh:38-48) for new_len >= 1 on a valid interval
__device__ __forceinline__ void d_drop(const FinDevIndex& ix, int new_len, uint32_t& l, uint32_t& r) {
    if (new_len <= 0) { l = 0; r = ix.n_nodes - 1; return; }
    while (l > 0 && (int)d_lcs(ix, l) >= new_len) l--;
    while (r < ix.n_nodes - 1 && (int)d_lcs(ix, r + 1) >= new_len) r++;
}

__device__ __forceinline__ uint32_t d_concat(const FinDevIndex& ix, uint32_t g) {
    return (ix.concat[g >> 4] >> (2 * (g & 15))) & 3u;
}

// PackedStrings::global_offset_to_local_offset: smallest idx with ends[idx] > gs
__device__ __forceinline__ void d_locate(const FinDevIndex& ix, uint32_t gs, uint32_t& u, uint32_t& ustart, uint32_t& uend) {
    uint32_t idx = ix.samp[gs >> ix.samp_shift];
    uint32_t e = ix.ends[idx + 1];
    while (e <= gs) { idx++; e = ix.ends[idx + 1]; }
    u = idx; uend = e;
    ustart = ix.ends[idx];
}

__device__ __forceinline__ uint32_t d_base_code(const uint8_t* bases, uint64_t o, uint32_t len, uint32_t pos, bool rev) {
    uint8_t ch = rev ? bases[o + (len - 1 - pos)] : bases[o + pos];
    ch &= (uint8_t)~32u;
    uint32_t c = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
    if (rev && c < 4) c = 3u - c;
    return c;
}

// deque entry: len(8) | colex(32) | end mod 2^24; order of (len, colex) decides (the end never does: the
// candidate being inserted always has the largest end, see DESIGN.md "deque")
__device__ __forceinline__ uint64_t dq_pack(uint32_t len, uint32_t colex, uint32_t end) {
    return ((uint64_t)len << 56) | ((uint64_t)colex << 24) | (uint64_t)(end & 0xFFFFFFu);
}
__device__ __forceinline__ uint32_t dq_end(uint64_t e, uint32_t cur_end) { return cur_end - ((cur_end - (uint32_t)e) & 0xFFFFFFu); }
__device__ __forceinline__ uint32_t dq_len(uint64_t e) { return (uint32_t)(e >> 56); }
__device__ __forceinline__ uint32_t dq_colex(uint64_t e) { return (uint32_t)(e >> 24); }

