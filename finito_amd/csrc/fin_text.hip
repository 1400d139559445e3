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
// (unpack != 0: a sum's bits from `unpack` upwards are a count of found pairs -- fin_text3_len_kernel --, added up per chunk into *n_found)
__global__ __launch_bounds__(FIN_TPB) void fin_text_scan1_kernel(const uint32_t* blk_sum, uint32_t n_blocks, uint64_t* blk_off, uint64_t* chunk_sum, uint32_t unpack, unsigned long long* n_found) {
    __shared__ uint32_t lds_wave[FIN_TPB / 64];
    __shared__ uint32_t lds_found;
    if (threadIdx.x == 0) lds_found = 0;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * FIN_TEXT_SCAN_CHUNK + threadIdx.x * FIN_TEXT_SCAN_PER;
    uint32_t v[FIN_TEXT_SCAN_PER];
    uint32_t s = 0, found = 0;
#pragma unroll
    for (int i = 0; i < FIN_TEXT_SCAN_PER; i++) {
        v[i] = b0 + i < n_blocks ? blk_sum[b0 + i] : 0u;
        if (unpack) { found += v[i] >> unpack; v[i] &= (1u << unpack) - 1u; }
        s += v[i];
    }
    if (found) atomicAdd(&lds_found, found);   // (a chunk's found pairs: 4096 segments of at most FIN_TEXT_SEG)
    uint32_t total;   // (a chunk's sums stay below 2^32: 4096 blocks of at most FIN_TEXT_PAIRS * 24 bytes, or as many segments of at most FIN_TEXT_SEG * 24)
    uint32_t run = block_exclusive_scan(s, lds_wave, total);
#pragma unroll
    for (int i = 0; i < FIN_TEXT_SCAN_PER; i++) { if (b0 + i < n_blocks) blk_off[b0 + i] = run; run += v[i]; }
    if (threadIdx.x == 0) { chunk_sum[blockIdx.x] = total; if (n_found && lds_found) atomicAdd(n_found, (unsigned long long)lds_found); }
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

// ---- round 4: the text of a batch that ran with fast-path records (fin_prepass.hip; fin_batch_text_mode 1 / 2) -------------------------------
// A read the fast path finished is a 32-byte record (FinFastRec), and its pairs are a function of that record: the kernels below make them
// again instead of reading them back -- 90 % of chr1's pairs -- and in text-only mode they were never written.  The unit of work is a SEGMENT:
// a read, or (reads of more than FIN_TEXT_SEG pairs: a table made by the host) FIN_TEXT_SEG consecutive pairs of one.  A wave takes 64
// consecutive segments -- a lane loads one segment's facts, the wave then works through them one by one -- and there is no block-wide step.
//   fin_text3_len_kernel    per segment: bytes of its text, and (bits 17..) its found pairs -- the count fin_batch_download reports.  A finished
//                           read in closed form from its record (a lane each: the runs between the ruled-out slots have consecutive offsets);
//                           any other segment from its pairs in memory, the wave together
//   fin_text_scan1/2        as above, over segments (the unpacking form)
//   fin_text3_write_kernel  128 pairs at a time: two per lane, wave prefix sum of their lengths (DPP), formatted into the wave's LDS staging at
//                           the same offset modulo 16 as in the output, copied out in aligned 16-byte pieces.  A finished read's pairs share
//                           the unitig's digits and the leading digits of the offsets: made once, a lane computes three digits of its own
// Measured on chr1 (1.16e9 pairs, 13.5 GB of text): lengths 0.72 ms (round 3's pass over the pairs: 2.3), write 5.5 (5.7) -- the write kernel is
// bound by its instruction count and LDS stores (230 vector + 160 scalar instructions per read after the specialisation, 320 + 220 before), not
// by memory: forms tried and dropped are named where they were (a wave per segment, stores of 2 / 4 / 8 bytes off their alignment, plain stores).
#define FIN_TEXT_SEG 4096u
#define FIN_TEXT3_GROUP 128u
struct Text3Src {
    const int2* pairs; const uint64_t* out_offs; const FinFastRec* frec;   // frec: zeroed before the run, so meta >> 16 != 0 marks a finished read's record (null: none)
    const uint2* seg;                                                     // {read, first pair inside it} per segment; null: segment i = read i
    uint32_t n_seg, k1;
};
namespace {
// A wave takes 64 consecutive segments: every lane loads one segment's facts -- the read's bounds, its record, its place in the text -- in ONE
// round of independent loads, then the wave works through them one by one, the facts broadcast from their lane into scalar registers.  (A wave
// per segment spent its time waiting for that round: 10^7 waves of 120 pairs each, 3.7 ms of latency whatever the work.)
struct Text3Seg { uint64_t first_pair; uint32_t first, n, nk; bool done; uint4 ra, rb; };
__device__ __forceinline__ Text3Seg text3_segment_of_lane(const Text3Src& S, uint32_t sg) {
    Text3Seg g;
    g.first_pair = 0; g.first = 0; g.n = 0; g.nk = 0; g.done = false; g.ra = make_uint4(0, 0, 0, 0); g.rb = g.ra;
    if (sg >= S.n_seg) return g;
    uint32_t r = sg;
    if (S.seg) { const uint2 e = S.seg[sg]; r = e.x; g.first = e.y; }
    const uint64_t a = S.out_offs[r], e = S.out_offs[r + 1];
    if (S.frec) { const uint4* q = (const uint4*)(S.frec + r); g.ra = q[0]; g.rb = q[1]; }
    g.nk = (uint32_t)(e - a);
    g.first_pair = a + g.first;
    const uint32_t left = g.nk - g.first;
    g.n = S.seg ? (left < FIN_TEXT_SEG ? left : FIN_TEXT_SEG) : left;
    return g;
}
__device__ __forceinline__ uint32_t from_lane(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ Text3Seg text3_broadcast(const Text3Seg& l, int src) {
    Text3Seg g;
    g.first_pair = ((uint64_t)from_lane((uint32_t)(l.first_pair >> 32), src) << 32) | from_lane((uint32_t)l.first_pair, src);
    g.first = from_lane(l.first, src); g.n = from_lane(l.n, src); g.nk = from_lane(l.nk, src);
    g.ra = make_uint4(from_lane(l.ra.x, src), from_lane(l.ra.y, src), from_lane(l.ra.z, src), from_lane(l.ra.w, src));
    g.rb = make_uint4(from_lane(l.rb.x, src), from_lane(l.rb.y, src), from_lane(l.rb.z, src), from_lane(l.rb.w, src));
    g.done = (g.ra.z >> 16) != 0u;
    return g;
}
// inclusive prefix sum over the wave's 64 lanes: four shifts inside the rows of 16, then the rows' totals handed on (DPP: six adds, no LDS)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// A finished read's text is regular: one unitig number for all its pairs, consecutive offsets.  Per wave: the unitig's digits once; per group of
// FIN_TEXT3_GROUP pairs: the offsets are B + d with B a multiple of 1000 and d < 1000 + FIN_TEXT3_GROUP, so their digits in front of the last three
// are those of B / 1000 or of B / 1000 + 1 -- two strings made once per group -- and a lane computes three digits of its own.  A character costs a
// lane one or two instructions instead of the eight of put_number's digit loop; the formatting kernels are bound by their instruction count.
struct DecStr { uint64_t lo; uint32_t hi, n; };   // a number's characters, the most significant in byte 0 of lo; hi: the ninth and tenth
__device__ __forceinline__ DecStr dec_str(uint32_t v) {
    DecStr s; s.n = ndigits(v); s.lo = 0; s.hi = 0;
    for (uint32_t j = 0; j < s.n; j++) { s.hi = (s.hi << 8) | (uint32_t)(s.lo >> 56); s.lo = (s.lo << 8) | (uint64_t)('0' + v % 10u); v /= 10u; }
    return s;
}
struct DoneGroup { uint32_t B; DecStr H0, H1; };
__device__ __forceinline__ DoneGroup done_group(const Text3Seg& g, uint32_t i0) {
    // pairs first + i0 .. of the read, at most FIN_TEXT3_GROUP of them: their slots of strand A, the smallest one's offset
    const uint32_t ia = g.first + i0, ib = g.first + (i0 + FIN_TEXT3_GROUP < g.n ? i0 + FIN_TEXT3_GROUP : g.n);
    const uint32_t min_sl = (g.ra.z & 0x100u) ? g.ra.w - ib : ia, min_off = g.ra.y + min_sl;
    DoneGroup G; G.B = min_off - min_off % 1000u;
    const uint32_t H = min_off / 1000u;
    G.H0 = dec_str(H); if (H == 0u) G.H0.n = 0u;   // (offsets below 1000 have no digits in front of their last three, and not always three)
    G.H1 = dec_str(H + 1u);
    return G;
}
// sum of the digit counts of a, a + 1, ..., b (a <= b)
__device__ __forceinline__ uint32_t digit_sum(uint32_t a, uint32_t b) {
    uint32_t s = b - a + 1u, p = 10u;
#pragma unroll
    for (int t = 1; t < 10; t++) { const uint32_t lo = a > p ? a : p; if (b >= lo) s += b - lo + 1u; p *= 10u; }   // (the values with more than t digits)
    return s;
}
// bytes of a finished read's whole text and (bits 17..) its found pairs, from its record alone: the slots a disagreeing position E rules out are
// [E - k + 1, E] (positions ascending, fin_prepass.hip), the runs between them have consecutive offsets -- no pair is looked at
__device__ __forceinline__ uint32_t done_read_len(const uint4 a, const uint4 b, uint32_t k1) {
    const uint32_t nk = a.w, nE = a.z & 0xFFu;
    if ((a.z >> 16) == 2u) return 8u * nk;
    const uint32_t per = ndigits(a.x) + 4u;
    uint32_t bytes = 0, found = 0, cur = 0;
    for (uint32_t j = 0; j < nE; j++) {
        const uint32_t w = j < 2u ? b.x : j < 4u ? b.y : j < 6u ? b.z : b.w, E = (j & 1u) ? w >> 16 : w & 0xFFFFu;
        uint32_t lo = E > k1 ? E - k1 : 0u; if (lo < cur) lo = cur;
        const uint32_t hi = E < nk - 1u ? E : nk - 1u;
        if (lo > nk) lo = nk;
        if (lo > cur) { bytes += (lo - cur) * per + digit_sum(a.y + cur, a.y + lo - 1u); found += lo - cur; }
        if (hi + 1u > lo) { bytes += 8u * (hi + 1u - lo); if (hi + 1u > cur) cur = hi + 1u; }
        else if (lo > cur) cur = lo;
    }
    if (cur < nk) { bytes += (nk - cur) * per + digit_sum(a.y + cur, a.y + nk - 1u); found += nk - cur; }
    return bytes + (found << 17);
}
struct DonePair { bool gap, carry; uint32_t len, d, n_low, n_hi; };
// a read's disagreeing positions as the slot test wants them (per read, wave-uniform): the first four, 0xFFFFFFFF where there is none (E - sl <= k - 1
// is then false for every slot); more than four: the uniform branch in done_pair
struct DoneE { uint32_t e0, e1, e2, e3; bool all_gap, more; };
__device__ __forceinline__ DoneE done_positions(const Text3Seg& g) {
    const uint32_t nE = g.ra.z & 0xFFu;
    DoneE x;
    x.e0 = nE > 0u ? g.rb.x & 0xFFFFu : 0xFFFFFFFFu; x.e1 = nE > 1u ? g.rb.x >> 16 : 0xFFFFFFFFu;
    x.e2 = nE > 2u ? g.rb.y & 0xFFFFu : 0xFFFFFFFFu; x.e3 = nE > 3u ? g.rb.y >> 16 : 0xFFFFFFFFu;
    x.all_gap = (g.ra.z >> 16) == 2u; x.more = nE > 4u;
    return x;
}
__device__ __forceinline__ DonePair done_pair(const Text3Seg& g, const DoneE& x, const DoneGroup& G, uint32_t i, uint32_t k1, uint32_t nU) {
    const uint4 a = g.ra, b = g.rb;
    const uint32_t sl = (a.z & 0x100u) ? a.w - 1u - i : i;
    DonePair q;
    uint32_t m = x.e0 - sl;   // (unsigned: a position in front of the slot wraps to something large)
    m = min(m, x.e1 - sl); m = min(m, x.e2 - sl); m = min(m, x.e3 - sl);
    if (x.more) {
        const uint32_t nE = a.z & 0xFFu;
        m = min(m, (b.z & 0xFFFFu) - sl);
        if (nE > 5u) m = min(m, (b.z >> 16) - sl);
        if (nE > 6u) m = min(m, (b.w & 0xFFFFu) - sl);
        if (nE > 7u) m = min(m, (b.w >> 16) - sl);
    }
    q.gap = x.all_gap || m <= k1;
    q.d = a.y + sl - G.B;
    q.carry = q.d >= 1000u;
    if (q.carry) q.d -= 1000u;
    q.n_hi = q.carry ? G.H1.n : G.H0.n;
    q.n_low = q.n_hi ? 3u : 1u + (q.d >= 10u ? 1u : 0u) + (q.d >= 100u ? 1u : 0u);
    q.len = q.gap ? 8u : nU + q.n_hi + q.n_low + 4u;
    return q;
}
}  // namespace

__global__ __launch_bounds__(FIN_TPB) void fin_text3_len_kernel(Text3Src S, uint32_t* seg_sum) {
    const uint32_t lane = threadIdx.x & 63u, sg0 = (blockIdx.x * (FIN_TPB / 64) + (threadIdx.x >> 6)) * 64u;
    if (sg0 >= S.n_seg) return;
    const Text3Seg mine = text3_segment_of_lane(S, sg0 + lane);
    // a finished read's segment (always the whole read: the fast path takes reads of at most 256 bases): closed form, a lane each
    const bool closed = (mine.ra.z >> 16) != 0u && mine.first == 0u && mine.n == mine.nk && mine.n == mine.ra.w;
    uint32_t my_sum = closed ? done_read_len(mine.ra, mine.rb, S.k1) : 0u;   // (bytes below 2^17 -- FIN_TEXT_SEG pairs of at most 24 --, found pairs above them)
    // every other segment: its pairs from memory (or, part of a finished read -- not produced today --, from the record), the wave together
    uint64_t rest = __ballot(!closed && mine.n != 0u);
    while (rest) {
        const int src = __ffsll((long long)rest) - 1;
        rest &= rest - 1ull;
        const Text3Seg g = text3_broadcast(mine, src);
        uint32_t s = 0;
        if (g.done) {
            const uint32_t nU = ndigits(g.ra.x);
            const DoneE X = done_positions(g);
            for (uint32_t i0 = 0; i0 < g.n; i0 += FIN_TEXT3_GROUP) {
                const DoneGroup G = done_group(g, i0);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const uint32_t i = i0 + 2u * lane + (uint32_t)j;
                    if (i < g.n) { const DonePair q = done_pair(g, X, G, g.first + i, S.k1, nU); s += q.len + (q.gap ? 0u : 1u << 17); }
                }
            }
        } else
            for (uint32_t i = lane; i < g.n; i += 64u) {
                const int2 pr = S.pairs[g.first_pair + i];
                s += pair_len(pr) + (pr.x >= 0 ? 1u << 17 : 0u);
            }
        s = wave_inclusive_scan(s, lane);
        const uint32_t tot = from_lane(s, 63);
        if ((int)lane == src) my_sum = tot;
    }
    if (sg0 + lane < S.n_seg) seg_sum[sg0 + lane] = my_sum;
}

namespace {
// One group of a finished read into the staging, specialised by the number of digits of the unitig (NU; 0: any, the count in U.n): the per-pair code
// is straight-line -- with the digit count a run-time value every character cost a scalar compare and branch, and the scalar unit became the bound
template <int NU>
__device__ __forceinline__ uint32_t text3_done_group(const Text3Seg& g, const DoneE& X, const DoneGroup& G, const DecStr& U, uint32_t i0, uint32_t k1, uint32_t lane, char* stage) {
    const uint32_t nU = NU ? (uint32_t)NU : U.n;
    DonePair q[2]; uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uint32_t i = i0 + 2u * lane + (uint32_t)j;
        q[j].len = 0;
        if (i < g.n) { q[j] = done_pair(g, X, G, g.first + i, k1, nU); s += q[j].len; }
    }
    const uint32_t incl = wave_inclusive_scan(s, lane);
    const uint32_t total = from_lane(incl, 63);
    char* p = stage + (incl - s);
    const uint32_t n_hi_max = G.H0.n > G.H1.n ? G.H0.n : G.H1.n;
    const uint32_t u_lo = (uint32_t)U.lo, u_hi = (uint32_t)(U.lo >> 32);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uint32_t i = i0 + 2u * lane + (uint32_t)j;
        if (i >= g.n) break;
        const uint32_t sep = (g.first + i + 1u == g.nk) ? (uint32_t)'\n' : (uint32_t)' ';
        // (a character per store: measured against stores of 4 and 2 bytes at whatever address they fall on -- fewer instructions, but gfx950
        //  replays every LDS store that is off its alignment, SQ_LDS_UNALIGNED_STALL became the kernel's bound)
        p[0] = '(';
        if (q[j].gap) { p[1] = '-'; p[2] = '1'; p[3] = ','; p[4] = '-'; p[5] = '1'; p[6] = ')'; p[7] = (char)sep; }
        else {
            if (nU > 0u) p[1] = (char)u_lo;
            if (nU > 1u) p[2] = (char)(u_lo >> 8);
            if (nU > 2u) p[3] = (char)(u_lo >> 16);
            if (nU > 3u) p[4] = (char)(u_lo >> 24);
            if (nU > 4u) p[5] = (char)u_hi;
            if (nU > 5u) p[6] = (char)(u_hi >> 8);
            if (nU > 6u) p[7] = (char)(u_hi >> 16);
            if (nU > 7u) p[8] = (char)(u_hi >> 24);
            if (nU > 8u) p[9] = (char)U.hi;
            if (nU > 9u) p[10] = (char)(U.hi >> 8);
            char* p2 = p + 1u + nU;
            p2[0] = ',';
            if (n_hi_max > 0u) {   // the offset's leading digits: one of the group's two strings (at most seven characters)
                const uint64_t hs = q[j].carry ? G.H1.lo : G.H0.lo;
                const uint32_t hs_lo = (uint32_t)hs, hs_hi = (uint32_t)(hs >> 32);
                if (q[j].n_hi > 0u) p2[1] = (char)hs_lo;
                if (n_hi_max > 1u) {
                    if (q[j].n_hi > 1u) p2[2] = (char)(hs_lo >> 8);
                    if (n_hi_max > 2u) {
                        if (q[j].n_hi > 2u) p2[3] = (char)(hs_lo >> 16);
                        if (q[j].n_hi > 3u) p2[4] = (char)(hs_lo >> 24);
                        if (q[j].n_hi > 4u) p2[5] = (char)hs_hi;
                        if (q[j].n_hi > 5u) p2[6] = (char)(hs_hi >> 8);
                        if (q[j].n_hi > 6u) p2[7] = (char)(hs_hi >> 16);
                    }
                }
            }
            char* p3 = p2 + 1u + q[j].n_hi;
            const uint32_t d = q[j].d, a = (d * 5243u) >> 19, r = d - 100u * a, b = (r * 103u) >> 10, c0 = r - 10u * b;   // d < 1000: its three digits
            uint64_t t = (uint64_t)(0x29303030u + (a | (b << 8) | (c0 << 16))) | ((uint64_t)sep << 32);   // "abc)" and the separator ...
            t >>= 8u * (3u - q[j].n_low);                                                                  // ... without the digits a short offset does not have
            const uint32_t t_lo = (uint32_t)t;
            p3[0] = (char)t_lo; p3[1] = (char)(t_lo >> 8); p3[2] = (char)(t_lo >> 16);
            if (q[j].n_low > 1u) p3[3] = (char)(t_lo >> 24);
            if (q[j].n_low > 2u) p3[4] = (char)(uint32_t)(t >> 32);
        }
        p += q[j].len;
    }
    return total;
}
// the staged bytes of a group to their place: up to the first 16-byte boundary of the output singly, then 16 at a time (both sides aligned), then the rest
// (the staging is the wave's own, and a wave's LDS instructions execute in the order they were issued: what the lanes wrote is what they read
//  here, and the next group's writes come after these reads -- no fence, which would also wait for the global stores)
__device__ __forceinline__ void text3_copy_out(char* dst, const char* stage, uint32_t mis, uint32_t total, uint32_t lane) {
    __builtin_amdgcn_wave_barrier();
    const uint32_t head = mis == 0u ? 0u : (16u - mis < total ? 16u - mis : total);
    if (lane < head) dst[lane] = stage[lane];
    const uint32_t chunks = (total - head) >> 4;
    for (uint32_t c = lane; c < chunks; c += 64u) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(*(const u4*)(stage + head + 16u * c), (u4*)(dst + head + 16u * c));   // (plain stores: 6.7 instead of 5.5 ms per chr1 batch)
    }
    const uint32_t tail0 = head + 16u * chunks;
    if (lane < total - tail0) dst[tail0 + lane] = stage[tail0 + lane];
    __builtin_amdgcn_wave_barrier();
}
}  // namespace

__global__ __launch_bounds__(FIN_TPB) void fin_text3_write_kernel(Text3Src S, const uint64_t* seg_off, char* text) {
    __shared__ __attribute__((aligned(16))) char stage_all[FIN_TPB / 64][FIN_TEXT3_GROUP * FIN_TEXT_MAX_PAIR + 16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, sg0 = (blockIdx.x * (FIN_TPB / 64) + (threadIdx.x >> 6)) * 64u;
    if (sg0 >= S.n_seg) return;
    const Text3Seg mine = text3_segment_of_lane(S, sg0 + lane);
    uint64_t my_at = 0;
    if (sg0 + lane < S.n_seg) my_at = seg_off[sg0 + lane] + seg_off[S.n_seg + (sg0 + lane) / FIN_TEXT_SCAN_CHUNK];
    char* const stage_raw = stage_all[wave];
    const bool my_done = (mine.ra.z >> 16) != 0u;
    // ---- the finished reads' segments.  What their formatting needs once -- the unitig's digits, the first group's offset strings -- is made by
    //      the lanes side by side, each for its own segment (made inside the loop it was most of the loop) ----
    {
        DecStr myU = {0ull, 0u, 0u}; DoneGroup myG = {0u, {0ull, 0u, 0u}, {0ull, 0u, 0u}};
        if (my_done) { myU = dec_str(mine.ra.x); myG = done_group(mine, 0u); }
        uint64_t todo = __ballot(my_done && mine.n != 0u);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const Text3Seg g = text3_broadcast(mine, src);
            char* dst = text + (((uint64_t)from_lane((uint32_t)(my_at >> 32), src) << 32) | from_lane((uint32_t)my_at, src));
            DecStr U; DoneGroup G0;
            U.lo = ((uint64_t)from_lane((uint32_t)(myU.lo >> 32), src) << 32) | from_lane((uint32_t)myU.lo, src); U.hi = from_lane(myU.hi, src); U.n = from_lane(myU.n, src);
            G0.B = from_lane(myG.B, src);
            G0.H0.lo = ((uint64_t)from_lane((uint32_t)(myG.H0.lo >> 32), src) << 32) | from_lane((uint32_t)myG.H0.lo, src); G0.H0.n = from_lane(myG.H0.n, src); G0.H0.hi = 0;
            G0.H1.lo = ((uint64_t)from_lane((uint32_t)(myG.H1.lo >> 32), src) << 32) | from_lane((uint32_t)myG.H1.lo, src); G0.H1.n = from_lane(myG.H1.n, src); G0.H1.hi = 0;
            const DoneE X = done_positions(g);
            for (uint32_t i0 = 0; i0 < g.n; i0 += FIN_TEXT3_GROUP) {
                const uint32_t mis = (uint32_t)((uintptr_t)dst & 15u);
                char* const stage = stage_raw + mis;
                const DoneGroup G = i0 == 0u ? G0 : done_group(g, i0);
                uint32_t total;
                switch (U.n) {   // (wave-uniform)
                    case 4: total = text3_done_group<4>(g, X, G, U, i0, S.k1, lane, stage); break;
                    case 5: total = text3_done_group<5>(g, X, G, U, i0, S.k1, lane, stage); break;
                    case 6: total = text3_done_group<6>(g, X, G, U, i0, S.k1, lane, stage); break;
                    case 7: total = text3_done_group<7>(g, X, G, U, i0, S.k1, lane, stage); break;
                    default: total = text3_done_group<0>(g, X, G, U, i0, S.k1, lane, stage); break;
                }
                text3_copy_out(dst, stage, mis, total, lane);
                dst += total;
            }
        }
    }
    // ---- every other segment: its pairs from memory ----
    uint64_t todo = __ballot(!my_done && mine.n != 0u);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t g_n = from_lane(mine.n, src), g_first = from_lane(mine.first, src), g_nk = from_lane(mine.nk, src);
        const uint64_t g_first_pair = ((uint64_t)from_lane((uint32_t)(mine.first_pair >> 32), src) << 32) | from_lane((uint32_t)mine.first_pair, src);
        char* dst = text + (((uint64_t)from_lane((uint32_t)(my_at >> 32), src) << 32) | from_lane((uint32_t)my_at, src));
        for (uint32_t i0 = 0; i0 < g_n; i0 += FIN_TEXT3_GROUP) {
            const uint32_t mis = (uint32_t)((uintptr_t)dst & 15u);
            char* const stage = stage_raw + mis;
            int2 pr[2]; uint32_t nd[2]; uint32_t s = 0;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t i = i0 + 2u * lane + (uint32_t)j;
                pr[j] = make_int2(-1, -1); nd[j] = 0;
                if (i < g_n) {
                    pr[j] = S.pairs[g_first_pair + i];
                    const uint32_t nu = pr[j].x < 0 ? 2u : ndigits((uint32_t)pr[j].x), np = pr[j].x < 0 ? 2u : ndigits((uint32_t)pr[j].y);
                    nd[j] = nu | (np << 4);
                    s += nu + np + 4u;
                }
            }
            const uint32_t incl = wave_inclusive_scan(s, lane);
            const uint32_t total = from_lane(incl, 63);
            char* p = stage + (incl - s);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t i = i0 + 2u * lane + (uint32_t)j;
                if (i >= g_n) break;
                *p++ = '(';
                if (pr[j].x < 0) { *p++ = '-'; *p++ = '1'; *p++ = ','; *p++ = '-'; *p++ = '1'; }
                else { p = put_number(p, (uint32_t)pr[j].x, nd[j] & 15u); *p++ = ','; p = put_number(p, (uint32_t)pr[j].y, nd[j] >> 4); }
                *p++ = ')';
                *p++ = (g_first + i + 1u == g_nk) ? '\n' : ' ';
            }
            text3_copy_out(dst, stage, mis, total, lane);
            dst += total;
        }
    }
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
    hipLaunchKernelGGL(fin_text_scan1_kernel, dim3(nc), dim3(FIN_TPB), 0, stream, d_blk_sum, nb, d_blk_off, d_blk_off + nb, 0u, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(fin_text_scan2_kernel, dim3(1), dim3(FIN_TPB), 0, stream, d_blk_off + nb, nc, d_total);
    return (int)hipGetLastError();
}
// The same two steps for a batch that ran with fast-path records (frec may be null: then every pair comes from memory).  seg / n_seg: see
// Text3Src (null: a segment per read).  d_seg_sum: n_seg u32; d_seg_off: fin_text3_off_words(n_seg) u64; d_found (may be null): the number of
// found pairs is ADDED to it.
extern "C" uint64_t fin_text3_off_words(uint64_t n_seg) { return n_seg + (n_seg + FIN_TEXT_SCAN_CHUNK - 1) / FIN_TEXT_SCAN_CHUNK + 1; }
extern "C" uint32_t fin_text3_seg_pairs(void) { return FIN_TEXT_SEG; }
extern "C" int fin_launch_text3_lengths(const void* pairs, const uint64_t* out_offs, const void* frec, const void* seg, uint32_t n_seg, uint32_t k,
                                        uint32_t* d_seg_sum, uint64_t* d_seg_off, uint64_t* d_total, unsigned long long* d_found, hipStream_t stream) {
    if (n_seg == 0) return (int)hipMemsetAsync(d_total, 0, 8, stream);
    const Text3Src S = {(const int2*)pairs, out_offs, (const FinFastRec*)frec, (const uint2*)seg, n_seg, k - 1u};
    const uint32_t wpb = FIN_TPB;   // (64 segments per wave)
    hipLaunchKernelGGL(fin_text3_len_kernel, dim3((n_seg + wpb - 1) / wpb), dim3(FIN_TPB), 0, stream, S, d_seg_sum);
    const uint32_t nc = (n_seg + FIN_TEXT_SCAN_CHUNK - 1) / FIN_TEXT_SCAN_CHUNK;
    hipLaunchKernelGGL(fin_text_scan1_kernel, dim3(nc), dim3(FIN_TPB), 0, stream, d_seg_sum, n_seg, d_seg_off, d_seg_off + n_seg, 17u, d_found);
    hipLaunchKernelGGL(fin_text_scan2_kernel, dim3(1), dim3(FIN_TPB), 0, stream, d_seg_off + n_seg, nc, d_total);
    return (int)hipGetLastError();
}
extern "C" int fin_launch_text3_write(const void* pairs, const uint64_t* out_offs, const void* frec, const void* seg, uint32_t n_seg, uint32_t k,
                                      const uint64_t* d_seg_off, char* d_text, hipStream_t stream) {
    if (n_seg == 0) return 0;
    const Text3Src S = {(const int2*)pairs, out_offs, (const FinFastRec*)frec, (const uint2*)seg, n_seg, k - 1u};
    const uint32_t wpb = FIN_TPB;   // (64 segments per wave)
    hipLaunchKernelGGL(fin_text3_write_kernel, dim3((n_seg + wpb - 1) / wpb), dim3(FIN_TPB), 0, stream, S, d_seg_off, d_text);
    return (int)hipGetLastError();
}
extern "C" int fin_launch_text_write(const void* pairs, uint64_t n_pairs, const uint64_t* d_blk_off, const uint32_t* d_last_bits, char* d_text, hipStream_t stream) {
    if (n_pairs == 0) return 0;
    hipLaunchKernelGGL(fin_text_write_kernel, dim3(fin_text_blocks(n_pairs)), dim3(FIN_TPB), 0, stream, (const int2*)pairs, n_pairs, d_blk_off, fin_text_blocks(n_pairs), d_last_bits, d_text);
    return (int)hipGetLastError();
}
