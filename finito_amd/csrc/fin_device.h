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


// ---- queues between kernels (kernel 4's pipeline) ---------------------------------------------------------------------------
// A wave appends to a queue in HBM through slots it reserves 64 at a time: one atomic on the queue's counter per 64 items instead
// of one per item (a single word takes about 88 atomics per microsecond, MI355X_MICROARCH.md "dequeue": ten million items would
// cost 110 ms).  Slots a wave reserved but did not use are filled with FIN_Q_EMPTY when it exits; consumers skip those.  The
// counter therefore counts RESERVED slots, and a queue needs room for its items + 64 per producing wave.
#define FIN_Q_EMPTY 0xFFFFFFFFu
struct FinWaveQueue { uint32_t base = 0, left = 0; };   // wave-uniform
template <typename T>
__device__ __forceinline__ void fin_wq_push(FinWaveQueue& w, bool emit, const T& item, T* queue, uint32_t* count, uint32_t lane) {
    const uint64_t m = __ballot(emit);
    if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m), rk = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t take1 = cnt < w.left ? cnt : w.left;
        if (emit && rk < take1) queue[w.base + rk] = item;
        w.base += take1; w.left -= take1;
        if (cnt > take1) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(count, 64u);
            w.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b); w.left = 64u;
            if (emit && rk >= take1) queue[w.base + (rk - take1)] = item;
            w.base += cnt - take1; w.left -= cnt - take1;
        }
    }
}
template <typename T>
__device__ __forceinline__ void fin_wq_flush(const FinWaveQueue& w, const T& empty, T* queue, uint32_t lane) {
    if (lane < w.left) queue[w.base + lane] = empty;
}
