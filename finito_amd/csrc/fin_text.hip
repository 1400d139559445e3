// fin_text.hip -- the reference's output text, made on the GPU.
//
// run_fmin_queries_streaming prints every k-mer's pair as "(unitig,offset)", single spaces between the pairs of a read and '\n'
// after its last one (search_fmin.hh:62-65) -- inside its timed region, through an ostream.  At 10^9 k-mers/s that text is the
// bottleneck of the whole command (12 bytes per k-mer, more than the pairs themselves), so it is produced where the pairs are:
//   fin_text_mark_kernel   one bit per pair: is it the last pair of its read (then '\n' follows it, else ' ')
//   fin_text_len_kernel    bytes of every pair's text, summed per block of FIN_TEXT_PAIRS pairs
//   fin_text_scan1/2       exclusive prefix sum of the block sums (per chunk of 4096 blocks, then over the chunks), total length
//   fin_text_write_kernel  every block formats its pairs into LDS at their offsets and copies the bytes out side by side
// HBM-streaming bound: 8 B in, about 12 B out per k-mer, twice over the pairs.  Every read of the batch must have at least one
// k-mer (a read shorter than k prints an empty line that belongs to no pair; the host formats such batches itself).
#include "fin_device.h"
#include "fin_kernels.h"

#ifndef FIN_TEXT_PER_THREAD
#define FIN_TEXT_PER_THREAD 4   // (even, <= 8.  4: 24.6 KB of staging per block, six blocks per CU -- 8.3 ms per chr1 batch; 8: three blocks, 10.5 ms; 2: 8.6 ms)
#endif
#define FIN_TEXT_PAIRS (FIN_TPB * FIN_TEXT_PER_THREAD)   // pairs per block
static_assert(FIN_TEXT_PER_THREAD % 2 == 0 && FIN_TEXT_PER_THREAD <= 8, "the write kernel loads its pairs two at a time");
#define FIN_TEXT_MAX_PAIR 24                              // "(2147483647,2147483647)" + separator

namespace {
__device__ __forceinline__ uint32_t ndigits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u :
           v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
// bytes of "(u,p)" plus one separator
__device__ __forceinline__ uint32_t pair_len(int2 pr) { return pr.x < 0 ? 8u : ndigits((uint32_t)pr.x) + ndigits((uint32_t)pr.y) + 4u; }
__device__ __forceinline__ char* put_number(char* p, uint32_t v, uint32_t n) {   // n = ndigits(v)
    for (uint32_t i = n; i-- > 0;) { p[i] = (char)('0' + v % 10u); v /= 10u; }
    return p + n;
}
// block-wide exclusive prefix sum of one value per thread (FIN_TPB threads); returns the block total through `total`
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d); if ((int)lane >= d) x += y; }
    if (lane == 63u) lds_wave[wave] = x;
    __syncthreads();
    uint32_t before = 0; total = 0;
    for (uint32_t w = 0; w < FIN_TPB / 64; w++) { const uint32_t s = lds_wave[w]; if (w < wave) before += s; total += s; }
    __syncthreads();
    return before + x - v;
}
}  // namespace

__global__ __launch_bounds__(FIN_TPB) void fin_text_mark_kernel(const uint64_t* out_offs, uint32_t n_reads, uint32_t* last_bits) {
    const uint32_t r = blockIdx.x * FIN_TPB + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t a = out_offs[r], e = out_offs[r + 1];
    if (e > a) atomicOr(&last_bits[(e - 1) >> 5], 1u << ((e - 1) & 31u));
}

__global__ __launch_bounds__(FIN_TPB) void fin_text_len_kernel(const int2* pairs, uint64_t n_pairs, uint32_t* blk_sum) {
    __shared__ uint32_t lds_wave[FIN_TPB / 64];
    // (only the block's sum is wanted: the threads take the pairs side by side, 512 contiguous bytes per wave and load)
    const uint64_t g0 = (uint64_t)blockIdx.x * FIN_TEXT_PAIRS + threadIdx.x;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < FIN_TEXT_PER_THREAD; i++) if (g0 + (uint64_t)i * FIN_TPB < n_pairs) s += pair_len(pairs[g0 + (uint64_t)i * FIN_TPB]);
    uint32_t total;
    (void)block_exclusive_scan(s, lds_wave, total);
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = total;
}

// Exclusive prefix sum of the block sums, in two steps (one workgroup over all 586 000 sums of a chr1 batch took 1.9 ms):
//   fin_text_scan1_kernel  a workgroup per FIN_TEXT_SCAN_CHUNK sums: blk_off[b] = sum of the chunk's sums in front of b, chunk_sum[c] = its total
//   fin_text_scan2_kernel  one workgroup: chunk_base[c] = sum of the chunks in front of c (64-bit: a batch's text may pass 4 GB), *total
// The write kernel adds the two: blk_off[b] + chunk_base[b / FIN_TEXT_SCAN_CHUNK] (chunk_base = blk_off + n_blocks).
#define FIN_TEXT_SCAN_PER 16
#define FIN_TEXT_SCAN_CHUNK (FIN_TPB * FIN_TEXT_SCAN_PER)
__global__ __launch_bounds__(FIN_TPB) void fin_text_scan1_kernel(const uint32_t* blk_sum, uint32_t n_blocks, uint64_t* blk_off, uint64_t* chunk_sum) {
    __shared__ uint32_t lds_wave[FIN_TPB / 64];
    const uint32_t b0 = blockIdx.x * FIN_TEXT_SCAN_CHUNK + threadIdx.x * FIN_TEXT_SCAN_PER;
    uint32_t v[FIN_TEXT_SCAN_PER];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < FIN_TEXT_SCAN_PER; i++) { v[i] = b0 + i < n_blocks ? blk_sum[b0 + i] : 0u; s += v[i]; }
    uint32_t total;   // (a chunk's sums stay below 2^32: 4096 blocks of at most FIN_TEXT_PAIRS * 24 bytes)
    uint32_t run = block_exclusive_scan(s, lds_wave, total);
#pragma unroll
    for (int i = 0; i < FIN_TEXT_SCAN_PER; i++) { if (b0 + i < n_blocks) blk_off[b0 + i] = run; run += v[i]; }
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(FIN_TPB) void fin_text_scan2_kernel(uint64_t* chunk_base, uint32_t n_chunks, uint64_t* total) {   // in place: sums in, bases out
    __shared__ uint64_t part[FIN_TPB];
    const uint32_t per = (n_chunks + FIN_TPB - 1) / FIN_TPB;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
    uint64_t s = 0;
    for (uint32_t c = lo; c < hi; c++) s += chunk_base[c];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t run = 0; for (int t = 0; t < FIN_TPB; t++) { const uint64_t v = part[t]; part[t] = run; run += v; } *total = run; }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t c = lo; c < hi; c++) { const uint64_t v = chunk_base[c]; chunk_base[c] = run; run += v; }
}

__global__ __launch_bounds__(FIN_TPB) void fin_text_write_kernel(const int2* pairs, uint64_t n_pairs, const uint64_t* blk_off, uint32_t n_blocks, const uint32_t* last_bits, char* text) {
    __shared__ uint32_t lds_wave[FIN_TPB / 64];
    // (the block's text is staged at the same offset modulo 16 as its place in the output, so that the copy-out moves aligned 16-byte
    //  pieces -- LDS reads as well as global stores; byte-wise reads of the staging buffer were 2/3 of this kernel's LDS traffic)
    __shared__ __attribute__((aligned(16))) char stage_raw[FIN_TEXT_PAIRS * FIN_TEXT_MAX_PAIR + 16];
    const uint64_t g0 = (uint64_t)blockIdx.x * FIN_TEXT_PAIRS + (uint64_t)threadIdx.x * FIN_TEXT_PER_THREAD;
    int2 pr[FIN_TEXT_PER_THREAD];
    uint32_t s = 0;
    uint64_t nds = 0;
    // (a thread's pairs are 8 * FIN_TEXT_PER_THREAD contiguous bytes; g0 + i is even -- FIN_TEXT_PER_THREAD is -- and `pairs` comes from hipMalloc: 16-byte loads of two pairs)
#pragma unroll
    for (int i = 0; i < FIN_TEXT_PER_THREAD; i += 2) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g0 + i + 1 < n_pairs) v = *(const uint4*)(pairs + g0 + i);
        else if (g0 + i < n_pairs) { const int2 one = pairs[g0 + i]; v.x = (uint32_t)one.x; v.y = (uint32_t)one.y; }
        pr[i] = make_int2((int)v.x, (int)v.y); pr[i + 1] = make_int2((int)v.z, (int)v.w);
#pragma unroll
        for (int j = i; j < i + 2; j++) {   // the digit counts are kept for the formatting below, a byte per pair (found pairs only)
            const uint32_t nu = pr[j].x < 0 ? 2u : ndigits((uint32_t)pr[j].x), np = pr[j].x < 0 ? 2u : ndigits((uint32_t)pr[j].y);
            nds |= (uint64_t)(nu | (np << 4)) << (8 * j);
            if (g0 + j < n_pairs) s += nu + np + 4u;
        }
    }
    uint32_t total;
    const uint32_t at = block_exclusive_scan(s, lds_wave, total);
    char* dst = text + blk_off[blockIdx.x] + blk_off[n_blocks + blockIdx.x / FIN_TEXT_SCAN_CHUNK];
    const uint32_t mis = (uint32_t)((uintptr_t)dst & 15u);
    char* const stage = stage_raw + mis;
    char* p = stage + at;
#pragma unroll
    for (int i = 0; i < FIN_TEXT_PER_THREAD; i++) {
        if (g0 + i >= n_pairs) break;
        const uint64_t g = g0 + i;
        *p++ = '(';
        if (pr[i].x < 0) { *p++ = '-'; *p++ = '1'; *p++ = ','; *p++ = '-'; *p++ = '1'; }
        else { const uint32_t nd = (uint32_t)(nds >> (8 * i)); p = put_number(p, (uint32_t)pr[i].x, nd & 15u); *p++ = ','; p = put_number(p, (uint32_t)pr[i].y, (nd >> 4) & 15u); }
        *p++ = ')';
        *p++ = ((last_bits[g >> 5] >> (g & 31u)) & 1u) ? '\n' : ' ';
    }
    __syncthreads();
    // bytes side by side: up to the first 16-byte boundary of the output singly, then 16 at a time (both sides aligned), then the rest
    const uint32_t head = mis == 0u ? 0u : (16u - mis < total ? 16u - mis : total);
    if (threadIdx.x < head) dst[threadIdx.x] = stage[threadIdx.x];
    const uint32_t chunks = (total - head) >> 4;
    for (uint32_t c = threadIdx.x; c < chunks; c += FIN_TPB)
    {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(*(const u4*)(stage + head + 16u * c), (u4*)(dst + head + 16u * c));
    }
    const uint32_t tail0 = head + 16u * chunks;
    if (threadIdx.x < total - tail0) dst[tail0 + threadIdx.x] = stage[tail0 + threadIdx.x];
}

extern "C" uint32_t fin_text_blocks(uint64_t n_pairs) { return (uint32_t)((n_pairs + FIN_TEXT_PAIRS - 1) / FIN_TEXT_PAIRS); }
// u64 words of d_blk_off: an offset per block, then a base per chunk of FIN_TEXT_SCAN_CHUNK blocks
extern "C" uint64_t fin_text_off_words(uint64_t n_pairs) { const uint64_t nb = fin_text_blocks(n_pairs); return nb + (nb + FIN_TEXT_SCAN_CHUNK - 1) / FIN_TEXT_SCAN_CHUNK + 1; }

// Enqueues mark + length + scan; *d_total (device) holds the text length afterwards.  d_last_bits: (n_pairs + 31) / 32 + 1 words;
// d_blk_sum: fin_text_blocks(n_pairs) u32; d_blk_off: fin_text_off_words(n_pairs) u64.
extern "C" int fin_launch_text_lengths(const void* pairs, uint64_t n_pairs, const uint64_t* out_offs, uint32_t n_reads, uint32_t* d_last_bits,
                                       uint32_t* d_blk_sum, uint64_t* d_blk_off, uint64_t* d_total, hipStream_t stream) {
    if (n_pairs == 0) return (int)hipMemsetAsync(d_total, 0, 8, stream);
    hipError_t e = hipMemsetAsync(d_last_bits, 0, ((n_pairs + 31) / 32 + 1) * 4, stream);
    if (e != hipSuccess) return (int)e;
    const uint32_t nb = fin_text_blocks(n_pairs);
    hipLaunchKernelGGL(fin_text_mark_kernel, dim3((n_reads + FIN_TPB - 1) / FIN_TPB), dim3(FIN_TPB), 0, stream, out_offs, n_reads, d_last_bits);
    hipLaunchKernelGGL(fin_text_len_kernel, dim3(nb), dim3(FIN_TPB), 0, stream, (const int2*)pairs, n_pairs, d_blk_sum);
    const uint32_t nc = (nb + FIN_TEXT_SCAN_CHUNK - 1) / FIN_TEXT_SCAN_CHUNK;
    hipLaunchKernelGGL(fin_text_scan1_kernel, dim3(nc), dim3(FIN_TPB), 0, stream, d_blk_sum, nb, d_blk_off, d_blk_off + nb);
    hipLaunchKernelGGL(fin_text_scan2_kernel, dim3(1), dim3(FIN_TPB), 0, stream, d_blk_off + nb, nc, d_total);
    return (int)hipGetLastError();
}
extern "C" int fin_launch_text_write(const void* pairs, uint64_t n_pairs, const uint64_t* d_blk_off, const uint32_t* d_last_bits, char* d_text, hipStream_t stream) {
    if (n_pairs == 0) return 0;
    hipLaunchKernelGGL(fin_text_write_kernel, dim3(fin_text_blocks(n_pairs)), dim3(FIN_TPB), 0, stream, (const int2*)pairs, n_pairs, d_blk_off, fin_text_blocks(n_pairs), d_last_bits, d_text);
    return (int)hipGetLastError();
}
