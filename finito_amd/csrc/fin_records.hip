// fin_records.hip -- a batch's results as RECORDS (round 5; include/finito_amd.h: fin_read_record, fin_search_batch_records).
//
// A read the pair pre-pass's fast path finished is a function of 32 bytes -- unitig, first offset, strand, the disagreeing positions (FinFastRec,
// fin_prepass.hip) -- and nine reads in ten of the benchmark's are such reads: their 120 pairs (960 bytes) need not be made on the device, cross PCIe
// and be read again by a caller who wants runs anyway.  In text-only mode (fin_batch_text_mode 2) the step already writes the record INSTEAD of the pairs;
// here the pairs of the OTHER reads -- the ones the pipeline searched, whose records stayed zero -- are gathered into one dense stream, read order kept,
// and their records are stamped {0, 0, 0, nk, 0, 0}: records + stream are the batch's whole result (fin_expand_records, fin_capi.cpp, makes the pairs).
#include "fin_device.h"
#include "fin_kernels.h"

#define FIN_REC_BLK 1024u   // reads per block

namespace {
__device__ __forceinline__ uint32_t rec_block_sum(uint32_t v, uint32_t* lds) {   // sum over the block's 256 threads, in every thread
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    if ((threadIdx.x & 63u) == 0u) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t t = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return t;
}
}  // namespace

// 1. per block of FIN_REC_BLK reads: pairs of the reads whose record is zero (meta >> 16 == 0: the pipeline searched them)
__global__ __launch_bounds__(256) void fin_rec_count_kernel(const FinFastRec* frec, const uint64_t* out_offs, uint32_t n_reads, uint32_t* blk_sum) {
    __shared__ uint32_t lds[4];
    uint32_t s = 0;
    const uint32_t r0 = blockIdx.x * FIN_REC_BLK;
    for (uint32_t i = threadIdx.x; i < FIN_REC_BLK; i += 256u) {
        const uint32_t r = r0 + i;
        if (r < n_reads && (frec[r].meta >> 16) == 0u) s += (uint32_t)(out_offs[r + 1] - out_offs[r]);
    }
    s = rec_block_sum(s, lds);
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = s;
}
// 2. exclusive prefix of the block sums (one block; n_blk <= 2^16 for a batch of 2^26 reads) and the stream's length
__global__ __launch_bounds__(1024) void fin_rec_scan_kernel(const uint32_t* blk_sum, uint32_t n_blk, uint64_t* blk_off, uint64_t* total) {
    __shared__ uint64_t lds[1024];
    const uint32_t per = (n_blk + 1023u) / 1024u, b0 = threadIdx.x * per;
    uint64_t s = 0;
    for (uint32_t i = 0; i < per; i++) if (b0 + i < n_blk) s += blk_sum[b0 + i];
    lds[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t y = threadIdx.x >= d ? lds[threadIdx.x - d] : 0ull;
        __syncthreads();
        lds[threadIdx.x] += y;
        __syncthreads();
    }
    uint64_t at = lds[threadIdx.x] - s;
    for (uint32_t i = 0; i < per; i++) if (b0 + i < n_blk) { blk_off[b0 + i] = at; at += blk_sum[b0 + i]; }
    if (threadIdx.x == 1023u) *total = lds[1023];
}
// 3. per block: where each of its unfinished reads' pairs go in the stream; their records stamped with nk; their pairs copied, a wave per read
__global__ __launch_bounds__(256) void fin_rec_compact_kernel(FinFastRec* frec, const uint64_t* out_offs, const int2* pairs, uint32_t n_reads, const uint64_t* blk_off, int2* stream) {
    __shared__ uint32_t lds[4], lds_w[4];
    __shared__ uint32_t dst[FIN_REC_BLK];   // offset of read i's pairs inside the block's stretch of the stream; 0xFFFFFFFF: a finished read
    const uint32_t r0 = blockIdx.x * FIN_REC_BLK, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // a thread's four consecutive reads, then the exclusive prefix over the block's threads
    uint32_t nk[4], mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t r = r0 + threadIdx.x * 4u + j;
        nk[j] = 0xFFFFFFFFu;
        if (r < n_reads && (frec[r].meta >> 16) == 0u) { nk[j] = (uint32_t)(out_offs[r + 1] - out_offs[r]); mine += nk[j]; }
    }
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, d); if ((int)lane >= d) inc += y; }
    if (lane == 63u) lds_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += lds_w[w];
    uint32_t at = before + inc - mine;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t i = threadIdx.x * 4u + j;
        dst[i] = nk[j] == 0xFFFFFFFFu ? 0xFFFFFFFFu : at;
        if (nk[j] != 0xFFFFFFFFu) { frec[r0 + i].nk = nk[j]; at += nk[j]; }
    }
    __syncthreads();
    (void)lds;
    int2* const out = stream + blk_off[blockIdx.x];
    for (uint32_t i = wave; i < FIN_REC_BLK; i += 4u) {
        const uint32_t d = dst[i];
        if (d == 0xFFFFFFFFu) continue;
        const uint64_t src = out_offs[r0 + i];
        const uint32_t n = (uint32_t)(out_offs[r0 + i + 1] - src);
        for (uint32_t j = lane; j < n; j += 64u) out[d + j] = pairs[src + j];
    }
}

extern "C" uint32_t fin_rec_blocks(uint32_t n_reads) { return (n_reads + FIN_REC_BLK - 1u) / FIN_REC_BLK; }
// blk_sum: fin_rec_blocks() u32; blk_off: as many u64; total: one u64 (the stream's pairs).  Launches kernels 1 and 2.
extern "C" int fin_launch_rec_count(const void* frec, const uint64_t* out_offs, uint32_t n_reads, uint32_t* blk_sum, uint64_t* blk_off, uint64_t* total, hipStream_t stream) {
    const uint32_t nb = fin_rec_blocks(n_reads);
    if (nb == 0) return (int)hipMemsetAsync(total, 0, 8, stream);
    hipLaunchKernelGGL(fin_rec_count_kernel, dim3(nb), dim3(256), 0, stream, (const FinFastRec*)frec, out_offs, n_reads, blk_sum);
    hipLaunchKernelGGL(fin_rec_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t*)blk_sum, nb, blk_off, total);
    return (int)hipGetLastError();
}
extern "C" int fin_launch_rec_compact(void* frec, const uint64_t* out_offs, const void* pairs, uint32_t n_reads, const uint64_t* blk_off, void* stream_out, hipStream_t stream) {
    const uint32_t nb = fin_rec_blocks(n_reads);
    if (nb == 0) return 0;
    hipLaunchKernelGGL(fin_rec_compact_kernel, dim3(nb), dim3(256), 0, stream, (FinFastRec*)frec, out_offs, (const int2*)pairs, n_reads, blk_off, (int2*)stream_out);
    return (int)hipGetLastError();
}

// ---- index sets (fin_pindex, fin_capi.cpp): the pairs a PART of the set found, into the set's result --------------------------------------------
// A part numbers its unitigs by itself; gid[] gives each the number the whole set's permute_unitigs (PackedStrings.hh:105-135) gives it.  No k-mer lies in two
// parts (checked when the set is built), so at most one part finds a slot's k-mer: first != 0 (the first part; dst may be src): every slot is written --
// a pair with its unitig renumbered, or (-1,-1); else only the slots this part found.  HBM-streaming bound: 8 B in, up to 8 B out per k-mer and part.
__global__ __launch_bounds__(256) void fin_set_merge_kernel(int2* dst, const int2* src, const uint32_t* gid, uint64_t n, int first) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        int2 p = src[i];
        if (p.x >= 0) { p.x = (int)gid[p.x]; dst[i] = p; }
        else if (first && dst != src) dst[i] = p;
    }
}
extern "C" int fin_launch_set_merge(void* dst, const void* src, const uint32_t* gid, uint64_t n_pairs, int first, hipStream_t stream) {
    if (n_pairs == 0) return 0;
    const uint64_t want = (n_pairs + 255) / 256;
    hipLaunchKernelGGL(fin_set_merge_kernel, dim3((uint32_t)(want < 65536 ? want : 65536)), dim3(256), 0, stream, (int2*)dst, (const int2*)src, gid, n_pairs, first);
    return (int)hipGetLastError();
}
