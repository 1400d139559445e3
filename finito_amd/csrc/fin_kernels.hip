// fin_kernels.hip -- gfx950 kernels of the search-fmin path.
//
// Path covered (reference): run_fmin_queries_streaming (search_fmin.hh:43-72) -> FinimizerIndex::search
// (FinimizerIndex.hh:119-185) -> rarest_fmin_streaming_search (common.hh:78-186) with update_sbwt_interval,
// drop_first_char (common.hh:38-48), BoundedDeque (BoundedDeque.hh), the two dictionary lookups
// (common.hh:61-72), PackedStrings::global_offset_to_local_offset (PackedStrings.hh:91-100) and
// walk_in_unitigs (FinimizerIndex.hh:47-102).
//
// Kernel "v0" below is the plain mapping: one lane = one read, reverse strand first, forward strand second
// (forward hits overwrite, which is exactly the merge rule of search_fmin.hh:54-60), the walk fused into the
// stream (SURVEY.md 8a-7 streaming form) and the sliding-window deque in LDS.  It is kept as the simple,
// obviously-faithful device version; the tuned kernel lives next to it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fin_device.h"
#include "fin_kernels.h"

struct LdsDeque {
    static constexpr uint32_t CAP = 16;
    uint64_t* base;   // &lds[0][tid]; entry i at base[i * FIN_TPB]
    uint32_t limit;   // <= CAP
    __device__ __forceinline__ uint64_t get(uint32_t i) const { return base[(i & (CAP - 1)) * FIN_TPB]; }
    __device__ __forceinline__ void set(uint32_t i, uint64_t v) { base[(i & (CAP - 1)) * FIN_TPB] = v; }
};
struct GlobalDeque {
    static constexpr uint32_t CAP = 256;   // >= k: with eager popping at most k entries are live
    uint64_t* base; uint64_t stride; uint32_t limit;
    __device__ __forceinline__ uint64_t get(uint32_t i) const { return base[(uint64_t)(i & (CAP - 1)) * stride]; }
    __device__ __forceinline__ void set(uint32_t i, uint64_t v) { base[(uint64_t)(i & (CAP - 1)) * stride] = v; }
};

// One strand of one read.  Returns false if the deque overflowed (nothing useful written).
template <typename DQ>
__device__ bool search_strand(const FinDevIndex& ix, const uint8_t* bases, uint64_t o, uint32_t len, bool rev,
                              bool write_miss, int2* out, DQ dq) {
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    uint32_t il = 0, ir = n - 1, kl = 0, kr = n - 1;
    int start = 0, kstart = 0;
    int bu_end = -1; uint32_t bu_colex = 0;
    uint32_t dq_head = 0, dq_cnt = 0;
    bool walk = false; uint32_t wg = 0, w_u = 0, w_ustart = 0, w_uend = 0;
    const int nk = (int)len - k + 1;

    for (int end = 0; end < (int)len; end++) {
        const uint32_t c = d_base_code(bases, o, len, (uint32_t)end, rev);
        bool found = false;
        uint32_t fin_end = 0, fin_colex = 0; bool use_branch = false;
        if (c > 3) {
            // defined behaviour for a non-ACGT base (reference: UB): matches nothing, state as after the
            // reference's own `start > end` reset (common.hh:118-122)
            start = end + 1; kstart = end + 1; il = 0; ir = n - 1; kl = 0; kr = n - 1;
            dq_cnt = 0;   // every candidate now starts before the window
        } else {
            // (1) finimizer interval
            uint32_t nl, nr;
            bool ok = d_extend(ix, c, il, ir, nl, nr);
            while (!ok) {
                kstart = ++start;
                if (start > end) { nl = 0; nr = n - 1; kl = nl; kr = nr; break; }
                d_drop(ix, end - start, il, ir);
                ok = d_extend(ix, c, il, ir, nl, nr);
                kl = nl; kr = nr;
            }
            il = nl; ir = nr;
            // (2) k-mer interval
            if (start != kstart) {
                uint32_t nkl, nkr;
                bool okk = d_extend(ix, c, kl, kr, nkl, nkr);
                while (!okk) {
                    kstart++;
                    d_drop(ix, end - kstart, kl, kr);
                    okk = d_extend(ix, c, kl, kr, nkl, nkr);
                }
                kl = nkl; kr = nkr;
            } else { kl = il; kr = ir; }
            // window bookkeeping: drop candidates that start before the current k-mer window ("eager" form of
            // the pop_front loop of common.hh:173-176; equivalence argued in CHANGELOG.md 4.3)
            while (dq_cnt) {
                uint64_t f = dq.get(dq_head);
                int fs = (int)dq_end(f, (uint32_t)end) - (int)dq_len(f) + 1;
                if (fs < kstart) { dq_head++; dq_cnt--; } else break;
            }
            // (2b) shortest unique suffix -> candidate
            if (il == ir) {
                uint32_t cl = 0, cc = 0;
                do {
                    cl = (uint32_t)(end - start + 1); cc = il;
                    start++;
                    d_drop(ix, end - start + 1, il, ir);
                } while (il == ir);
                uint64_t cand = dq_pack(cl, cc, (uint32_t)end);
                if (dq_cnt && (dq.get(dq_head) >> 24) > (cand >> 24)) { dq_cnt = 0; }
                else { while (dq_cnt && (dq.get(dq_head + dq_cnt - 1) >> 24) > (cand >> 24)) dq_cnt--; }
                if (dq_cnt >= dq.limit) return false;
                dq.set(dq_head + dq_cnt, cand); dq_cnt++;
            }
            // Ustart probe (common.hh:167)
            if (kl == kr && (d_nodebyte(ix, kl) & FIN_USTART_BIT)) { bu_end = end; bu_colex = kl; }
            // k-mer present (common.hh:170-182)
            if (end - kstart + 1 == k) {
                if (dq_cnt) {
                    uint64_t w = dq.get(dq_head);
                    found = true;
                    fin_end = dq_end(w, (uint32_t)end); fin_colex = dq_colex(w);
                    use_branch = bu_end >= (int)fin_end;
                }
                kstart++;
                d_drop(ix, end - kstart + 1, kl, kr);
            }
        }
        // FinimizerIndex::search resolve loop + walk_in_unitigs in streaming form
        if (end >= k - 1) {
            const int pos = end - k + 1;
            const int opos = rev ? (nk - 1 - pos) : pos;
            int2 res = make_int2(-1, -1);
            if (walk && wg + 1 < w_uend && c < 4 && c == d_concat(ix, wg + 1)) {
                wg++;
                res = make_int2((int)w_u, (int)(wg - (uint32_t)(k - 1) - w_ustart));
            } else if (found) {
                uint32_t g;
                if (use_branch) {
                    // lookup_from_branch_dictionary, common.hh:61-67
                    const uint32_t o = bu_colex & 63;
                    const FinBlockInfo bi = ix.blkinfo[bu_colex >> 6];
                    uint32_t rank = bi.ustart_rank + (uint32_t)__popcll((bi.ustart_mask_lo | ((uint64_t)bi.ustart_mask_hi << 32)) & (o ? (~0ull >> (64 - o)) : 0ull));
                    uint32_t us = ix.ends[rank];
                    g = us + (uint32_t)(k - 1) + (uint32_t)(end - bu_end);
                } else {
                    // lookup_from_finimizer_dictionary, common.hh:69-72
                    const uint32_t o = fin_colex & 63;
                    const FinBlockInfo bi = ix.blkinfo[fin_colex >> 6];
                    uint32_t rank = bi.fmin_rank + (uint32_t)__popcll((bi.fmin_mask_lo | ((uint64_t)bi.fmin_mask_hi << 32)) & (o ? (~0ull >> (64 - o)) : 0ull));
                    g = ix.goff[rank] + (uint32_t)end - fin_end;
                }
                uint32_t gs = g - (uint32_t)(k - 1);
                if (gs < ix.total_len) {
                    d_locate(ix, gs, w_u, w_ustart, w_uend);
                    res = make_int2((int)w_u, (int)(gs - w_ustart));
                    walk = true; wg = g;
                } else walk = false;   // unreachable on a consistent index (the reference reads out of bounds here): reported as absent
            } else {
                walk = false;
            }
            if (res.x != -1 || write_miss) out[opos] = res;
        }
    }
    return true;
}

template <typename DQ>
__device__ void search_read(const FinDevIndex& ix, const uint8_t* bases, const uint64_t* offs, const uint64_t* out_offs,
                            int2* out, int strands, uint32_t r, DQ dq, uint32_t* ovf_list, uint32_t* ovf_count) {
    const uint64_t o = offs[r];
    const uint32_t len = (uint32_t)(offs[r + 1] - o);
    if (len < ix.k) return;
    int2* dst = out + out_offs[r];
    bool ok = true;
    if (strands == 1) {
        ok = search_strand<DQ>(ix, bases, o, len, true, true, dst, dq);          // rc(read): writes every slot
        if (ok) ok = search_strand<DQ>(ix, bases, o, len, false, false, dst, dq);   // read: hits overwrite
    } else {
        ok = search_strand<DQ>(ix, bases, o, len, false, true, dst, dq);
    }
    if (!ok && ovf_list) fin_ovf_push(ix, ovf_list, ovf_count, r);
}

__global__ __launch_bounds__(FIN_TPB) void fin_search_v0_kernel(FinDevIndex ix, const uint8_t* bases, const uint64_t* offs,
                                                                 const uint64_t* out_offs, int2* out, uint32_t n_reads, int strands,
                                                                 uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count) {
    __shared__ uint64_t lds_dq[LdsDeque::CAP * FIN_TPB];
    uint32_t r = blockIdx.x * FIN_TPB + threadIdx.x;
    if (r >= n_reads) return;
    LdsDeque dq{lds_dq + threadIdx.x, lds_deque_limit};
    search_read<LdsDeque>(ix, bases, offs, out_offs, out, strands, r, dq, ovf_list, ovf_count);
}

// Reads whose candidate deque outgrew the LDS slots (more than 16 live candidates; bound is k) are redone here
// with the deque in a global scratch ring.
__global__ __launch_bounds__(FIN_TPB) void fin_search_overflow_kernel(FinDevIndex ix, const uint8_t* bases, const uint64_t* offs,
                                                                       const uint64_t* out_offs, int2* out, int strands,
                                                                       const uint32_t* ovf_list, const uint32_t* ovf_count,
                                                                       uint64_t* scratch) {
    const uint32_t nthreads = gridDim.x * FIN_TPB;
    const uint32_t tid = blockIdx.x * FIN_TPB + threadIdx.x;
    const uint32_t cnt = *ovf_count < ix.ovf_cap ? *ovf_count : ix.ovf_cap;
    GlobalDeque dq{scratch + tid, nthreads, GlobalDeque::CAP};
    for (uint32_t i = tid; i < cnt; i += nthreads)
        if (ovf_list[i] != FIN_Q_EMPTY)   // (a slot a wave of the walk kernel reserved and did not use: k > 128, see fin_launch_search_v4)
            search_read<GlobalDeque>(ix, bases, offs, out_offs, out, strands, ovf_list[i], dq, nullptr, nullptr);
}

// "Total found kmers" (search_fmin.hh:61,77)
__global__ __launch_bounds__(FIN_TPB) void fin_count_positive_kernel(const int2* out, uint64_t n, unsigned long long* result) {
    uint64_t i = (uint64_t)blockIdx.x * FIN_TPB + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * FIN_TPB;
    unsigned long long c = 0;
    for (; i < n; i += stride) c += out[i].x != -1;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(result, c);
}

// ---- host launchers -------------------------------------------------------------------------------------------
extern "C" {

int fin_launch_search_v0(const FinDevIndex* ix, const uint8_t* bases, const uint64_t* offs, const uint64_t* out_offs,
                         void* out, uint32_t n_reads, int strands, uint32_t lds_deque_limit, uint32_t* ovf_list, uint32_t* ovf_count,
                         uint64_t* ovf_scratch, uint32_t ovf_blocks, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (n_reads == 0) return 0;
    hipError_t e = hipMemsetAsync(ovf_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    uint32_t grid = (n_reads + FIN_TPB - 1) / FIN_TPB;
    if (ev0) (void)hipEventRecord(ev0, stream);
    hipLaunchKernelGGL(fin_search_v0_kernel, dim3(grid), dim3(FIN_TPB), 0, stream, *ix, bases, offs, out_offs, (int2*)out, n_reads,
                       strands, lds_deque_limit, ovf_list, ovf_count);
    if (ev1) (void)hipEventRecord(ev1, stream);
    return fin_launch_overflow(ix, bases, offs, out_offs, out, strands, ovf_list, ovf_count, ovf_scratch, ovf_blocks, stream);
}

int fin_launch_overflow(const FinDevIndex* ix, const uint8_t* bases, const uint64_t* offs, const uint64_t* out_offs, void* out,
                        int strands, const uint32_t* ovf_list, const uint32_t* ovf_count, uint64_t* ovf_scratch, uint32_t ovf_blocks,
                        hipStream_t stream) {
    hipLaunchKernelGGL(fin_search_overflow_kernel, dim3(ovf_blocks), dim3(FIN_TPB), 0, stream, *ix, bases, offs, out_offs,
                       (int2*)out, strands, ovf_list, ovf_count, ovf_scratch);
    return (int)hipGetLastError();
}

int fin_launch_count_positive(const void* out, uint64_t n_pairs, unsigned long long* d_result, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_result, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return (int)e;
    if (n_pairs == 0) return 0;
    uint64_t blocks = (n_pairs + FIN_TPB - 1) / FIN_TPB;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fin_count_positive_kernel, dim3((uint32_t)blocks), dim3(FIN_TPB), 0, stream, (const int2*)out, n_pairs, d_result);
    return (int)hipGetLastError();
}

uint32_t fin_overflow_deque_cap(void) { return GlobalDeque::CAP; }
}

// ---- self-test of the read-chunk cache the epoch kernels share (FinChunkCache, fin_device.h) -- VERDICT r4 #9b ------------------------------------
// The hazard of commit 81fcdfd: within ONE epoch a lane asks for a chunk it does not hold (the load is requested into the CURRENT slot) and then for the
// chunk in its NEXT slot.  Promoting that chunk at once would let the pending load land under the promoted chunk's number -- wrong bases under a valid tag.
// need() must refuse (return false, change nothing) until the load has been served.  A lane per scenario; bit i of *fail = scenario i went wrong.
namespace {
__device__ __forceinline__ uint4 sc_load(const void* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
}
__global__ void fin_chunk_cache_selftest_kernel(const uint4* chunks, uint32_t* fail) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto strand = [&]() -> const uint4* { return chunks; };
    auto codes_of = [&](int ci) -> uint64_t { const uint4 c = chunks[ci]; return c.x | ((uint64_t)c.y << 32); };
    uint32_t bad = 0;
    FinChunkCache ck; uint32_t q = 0; const void* q_aux = nullptr; uint4 aux = make_uint4(0, 0, 0, 0);
    auto serve = [&]() { if (q & FIN_Q_AUX) aux = sc_load(q_aux); ck.serve(q, aux, strand); q = 0; };
    // epoch 1: chunks 0 (current) and 1 (next) are asked for; nothing is there yet
    if (ck.need2(0, 1, strand, q, q_aux)) bad |= 1u;
    serve();
    // epoch 2: both are there
    if (!ck.need2(0, 1, strand, q, q_aux) || ck.bcodes != codes_of(0) || ck.ncodes != codes_of(1)) bad |= 2u;
    // ... the same epoch: chunk 2 is asked for (a load into the current slot), then chunk 1 (the next slot's): NO promotion while that load is under way
    if (ck.need(2, strand, q, q_aux)) bad |= 4u;
    if (ck.need(1, strand, q, q_aux)) bad |= 8u;
    if (ck.cur != 2 || ck.nxt != 1) bad |= 16u;
    serve();
    // epoch 3: chunk 2 arrived under its own number; chunk 1 is still in the next slot and is promoted now
    if (!ck.need(2, strand, q, q_aux) || ck.bcodes != codes_of(2)) bad |= 32u;
    if (!ck.need(1, strand, q, q_aux) || ck.cur != 1 || ck.bcodes != codes_of(1)) bad |= 64u;
    // a load of the NEXT slot under way: the chunk is not promoted either
    FinChunkCache c2; q = 0;
    if (c2.need2(0, 1, strand, q, q_aux)) bad |= 128u;
    if (c2.need(1, strand, q, q_aux)) bad |= 256u;       // (chunk 1 was requested into the next slot this very epoch)
    if (c2.cur != 0 || c2.nxt != 1) bad |= 512u;
    serve(); ck = c2;
    *fail = bad;
}
extern "C" int fin_debug_chunk_cache_selftest(uint32_t* fail_bits) {
    uint4 h[3]; uint32_t* d_fail = nullptr; uint4* d_chunks = nullptr;
    for (uint32_t i = 0; i < 3; i++) h[i] = make_uint4(0x11111111u * (i + 1), 0x01234567u + i, 0xFFFFFFFFu, 0u);
    if (hipMalloc((void**)&d_chunks, sizeof h) != hipSuccess || hipMalloc((void**)&d_fail, 4) != hipSuccess) { (void)hipFree(d_chunks); return -3; }
    (void)hipMemcpy(d_chunks, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMemset(d_fail, 0xFF, 4);
    hipLaunchKernelGGL(fin_chunk_cache_selftest_kernel, dim3(1), dim3(64), 0, nullptr, d_chunks, d_fail);
    uint32_t f = 0xFFFFFFFFu;
    const hipError_t e = hipMemcpy(&f, d_fail, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_chunks); (void)hipFree(d_fail);
    if (fail_bits) *fail_bits = f;
    return e == hipSuccess ? 0 : -3;
}

// ---- diagnostic (tests): the compact k-mer table asked directly -- what it claims about a list of k-mers, and whether the text bears each claim out --------------
// out[i] = {g, flags}: flags 0 = no claim (the chain ended: the k-mer is in no unitig); 1 = a verified claim with answer g; 2 = an unverified claim (the exact side
// table is asked: | 8 = found there, g = its answer); | 4 = the text at [g-k+1, g] spells the k-mer.  The look-up the kernels make (fin_prepass.hip kt3_find,
// fin_kernel_w.hip W_KF1), stated once more in the plainest form.
__global__ void fin_kt3_query_kernel(FinDevIndex ix, const uint64_t* k0, const uint64_t* k1, uint32_t n, uint2* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = k0[i], b = k1[i];
    const uint64_t h = fin_kt3_hash(a, b);
    const uint32_t tag = (uint32_t)h & FIN_KT3_TAGMASK;
    uint32_t bk = fin_kt3_bucket(h, ix.kt3_buckets), g = 0xFFFFFFFFu, flags = 0;
    for (uint32_t tries = 0; tries <= ix.kt3_buckets && !flags; tries++) {
        const FinKt3Bucket bu = ix.kt3[bk];
        bool ended = false;
        for (int j = 0; j < FIN_KT3_SLOTS && !flags && !ended; j++) {
            const uint32_t m = bu.w[2 * j + 1];
            if (m == 0xFFFFFFFFu) ended = true;
            else if ((m & FIN_KT3_TAGMASK) == tag) { g = bu.w[2 * j]; flags = (m & FIN_KT3_UNVER) ? 2u : 1u; }
        }
        if (ended) break;
        bk = bk + 1u == ix.kt3_buckets ? 0u : bk + 1u;
    }
    if (flags == 2u && ix.ktx) {
        uint32_t slot = (uint32_t)(h >> 32) & ((1u << ix.ktx_log2) - 1u);
        for (uint32_t tries = 0; tries < (1u << ix.ktx_log2); tries++) {
            const FinKtxSlot e = ix.ktx[slot];
            if (e.claim == 0xFFFFFFFFu) break;
            if ((e.k0_lo | ((uint64_t)e.k0_hi << 32)) == a && (e.k1_lo | ((uint64_t)e.k1_hi << 32)) == b) { g = e.g; flags |= 8u; break; }
            slot = (slot + 1u) & ((1u << ix.ktx_log2) - 1u);
        }
    }
    if (flags && g != 0xFFFFFFFFu && g >= ix.k - 1u && g < ix.total_len) {
        bool eq = true;
        for (uint32_t j = 0; j < ix.k && eq; j++) {
            const uint32_t c = d_concat(ix, g - (ix.k - 1u) + j);
            const uint32_t q = j < 32u ? (uint32_t)(a >> (2u * j)) & 3u : (uint32_t)(b >> (2u * (j - 32u))) & 3u;
            eq = c == q;
        }
        if (eq) flags |= 4u;
    }
    out[i] = make_uint2(g, flags);
}
extern "C" int fin_launch_kt3_query(const FinDevIndex* ix, const uint64_t* k0, const uint64_t* k1, uint32_t n, void* out, hipStream_t stream) {
    if (!ix->kt3 || n == 0) return 0;
    hipLaunchKernelGGL(fin_kt3_query_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, *ix, k0, k1, n, (uint2*)out);
    return (int)hipGetLastError();
}
