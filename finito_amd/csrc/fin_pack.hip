// fin_pack.hip -- ingest kernel of a step: ASCII reads -> 2-bit chunks of both strands.
//
// The reference's streaming loop makes the reverse complement of every read inside its timed region (sbwt::get_rc,
// search_fmin.hh:50) and both searches decode ASCII base by base (common.hh:106-112).  Here that work is ONE streaming pass at
// the head of every step (fin_batch_run): per read `[forward chunks | reverse-complement chunks]`, a chunk = 32 bases as
// {u64 2-bit codes (A0 C1 G2 T3, base j at bits 2j), u32 validity bits, u32 0} -- 16 bytes, one load per 32 bases for the search
// kernels, which never see ASCII or reverse-complement anything.  Case-insensitive; any other byte is an invalid base.
//
// HBM-streaming bound: 2 x 16-byte loads in, 1 x 16-byte store out per chunk (about 2 B in + 1.07 B out per base of the batch).
// A wave owns FIN_PACK_SPAN consecutive output chunks: one binary search over the reads' first-chunk table for the span's first
// chunk, after that it walks the table forward -- per 64 chunks one coalesced load of the next 64 descriptors and a 6-step
// search among them with ds_bpermute.  (Round 1 ran a 24-step global binary search per chunk: 3.6 ms per 10 M reads.)
#include <cstdlib>

#include "fin_device.h"
#include "fin_kernels.h"

#define FIN_PACK_SPAN 4096u   // at most; a small batch takes shorter spans so that the chip still has waves enough (fin_launch_pack_reads)

namespace {
// 0x80 in every byte of v that is zero (exact: no borrow crosses a byte)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; }

// four ASCII bases (byte j = position j) -> 8 code bits (2 per base) and 4 validity bits
__device__ __forceinline__ void pack4(uint32_t w, bool comp, uint32_t& codes, uint32_t& valid) {
    const uint32_t x = w & 0xDFDFDFDFu;                                   // upper case
    const uint32_t r = (x >> 1) & 0x03030303u;                            // A 0, C 1, T 2, G 3
    // the one letter a byte with these two bits can be: 'A' + {0, 2, 0x13, 6}[r], byte-wise (0/1 bytes times small constants: no carry)
    const uint32_t b0 = r & 0x01010101u, b1 = (r >> 1) & 0x01010101u;
    const uint32_t expect = 0x41414141u + (b0 << 1) + b1 * 0x13u - (b0 & b1) * 0x0Fu;
    const uint32_t ok = zero_bytes(x ^ expect) >> 7;                      // 1 in every byte that is a base
    uint32_t y = r ^ b1;                                                  // A 0, C 1, G 2, T 3
    if (comp) y ^= 0x03030303u;
    y &= ok * 3u;                                                         // an invalid base has code 0
    codes = (y * 0x01041040u) >> 24;                                      // byte j's two bits -> bits 2j
    valid = ((ok * 0x01020408u) >> 24) & 0xFu;                            // byte j's flag -> bit j
}
}  // namespace

__global__ __launch_bounds__(FIN_TPB) void fin_pack_reads_kernel(const uint8_t* bases, const uint64_t* offs, const FinReadDesc* desc,
                                                                  uint4* packed, uint32_t n_reads, uint64_t n_chunks, uint32_t span) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * FIN_TPB + threadIdx.x) >> 6;
    const uint64_t c0 = wave * span;
    if (c0 >= n_chunks) return;
    const uint64_t c1 = c0 + span < n_chunks ? c0 + span : n_chunks;
    // last read whose first chunk is <= c0 (desc[r].off = first chunk of read r; reads without chunks share their successor's)
    uint32_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (desc[mid].off <= c0) lo = mid; else hi = mid; }
    uint32_t rcur = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    for (uint64_t c = c0; c < c1; c += 64) {
        const uint32_t nact = (uint32_t)(c1 - c < 64 ? c1 - c : 64);   // chunks of this round (wave-uniform)
        bool todo = lane < nact;
        // rounds: a window of 64 descriptors from rcur on covers the 64 chunks unless more than 63 reads begin inside them
        // (only with empty reads); lanes whose owner may lie beyond the window go again with the window moved on
        for (;;) {
            const uint32_t ri = rcur + lane < n_reads ? rcur + lane : n_reads;   // desc[n_reads].off = n_chunks (sentinel)
            const FinReadDesc d = desc[ri];
            const uint64_t bo = offs[ri];
            const uint32_t rel = d.off <= c ? 0u : (d.off - c > 64u ? 64u : (uint32_t)(d.off - c));   // first chunk relative to c
            uint32_t i = 0;   // largest window slot whose read starts at or before this lane's chunk
#pragma unroll
            for (uint32_t step = 32; step >= 1; step >>= 1) {
                const uint32_t v = (uint32_t)__shfl((int)rel, (int)(i + step));
                if (v <= lane) i += step;
            }
            const uint32_t o_lo = (uint32_t)__shfl((int)(uint32_t)d.off, (int)i), o_hi = (uint32_t)__shfl((int)(uint32_t)(d.off >> 32), (int)i);
            const uint32_t b_lo = (uint32_t)__shfl((int)(uint32_t)bo, (int)i), b_hi = (uint32_t)__shfl((int)(uint32_t)(bo >> 32), (int)i);
            const uint32_t len = (uint32_t)__shfl((int)d.len, (int)i);
            const bool inside = i < 63u || rcur + 63u >= n_reads;   // slot 63 may hide later reads that start at the same chunk or before this lane's
            if (todo && inside) {
                const uint64_t first = o_lo | ((uint64_t)o_hi << 32), o = b_lo | ((uint64_t)b_hi << 32);
                const uint32_t nch = (len + 31u) >> 5;
                const uint32_t w = (uint32_t)(c + lane - first);
                const bool s = w >= nch;                            // reverse-complement half
                const uint32_t ci = s ? w - nch : w;
                const uint32_t p0 = ci * 32u;                       // first position of the chunk in strand coordinates
                const uint32_t cnt = len - p0 < 32u ? len - p0 : 32u;
                // forward: bytes o+p0 ..; reverse: window [o+len-p0-32, o+len-p0) read backwards (64 guard bytes around the buffer)
                const uint8_t* src = s ? bases + o + len - p0 - 32 : bases + o + p0;
                uint4 va, vb;
                __builtin_memcpy(&va, src, 16); __builtin_memcpy(&vb, src + 16, 16);
                const uint32_t wds[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
                uint64_t codes = 0; uint32_t valid = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    // strand positions 4q..4q+3: forward = dword q; reverse = dword 7-q with its bytes reversed, complemented
                    const uint32_t wd = s ? __builtin_bswap32(wds[7 - q]) : wds[q];
                    uint32_t c8, v4;
                    pack4(wd, s, c8, v4);
                    codes |= (uint64_t)c8 << (8 * q);
                    valid |= v4 << (4 * q);
                }
                const uint32_t keep = cnt >= 32u ? 0xFFFFFFFFu : ((1u << cnt) - 1u);
                valid &= keep;
                // codes of invalid positions are 0: spread the 32 validity bits to 2 bits each
                uint64_t m = valid;
                m = (m | (m << 16)) & 0x0000FFFF0000FFFFull; m = (m | (m << 8)) & 0x00FF00FF00FF00FFull;
                m = (m | (m << 4)) & 0x0F0F0F0F0F0F0F0Full; m = (m | (m << 2)) & 0x3333333333333333ull;
                m = (m | (m << 1)) & 0x5555555555555555ull;
                codes &= m | (m << 1);
                {   // (written once, read by later kernels from HBM anyway: a 1.6 GB stream does not stay in any cache)
                    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                    const u4 t = {(uint32_t)codes, (uint32_t)(codes >> 32), valid, 0u};
                    __builtin_nontemporal_store(t, (u4*)&packed[c + lane]);
                }
                todo = false;
            }
            // owner of the round's last chunk = where the next window starts (wave-uniform); it is final when that lane was served
            const uint32_t i_last = (uint32_t)__shfl((int)i, (int)(nact - 1u));
            const bool more = __any(todo);
            rcur += more ? 63u : i_last;
            if (!more) break;
        }
    }
}

extern "C" int fin_launch_pack_reads(const uint8_t* bases, const uint64_t* offs, const FinReadDesc* desc, void* packed, uint32_t n_reads,
                                     uint64_t n_chunks, hipStream_t stream) {
    if (n_reads == 0 || n_chunks == 0) return 0;
    // a wave's span: FIN_PACK_SPAN chunks, or -- a batch too small to give the chip two rounds of waves at that -- as few as 256 (multiples of 64).
    // (configs[1], 1 M reads: 2 441 waves of 4096 chunks took 178 us, a quarter of what the chip holds at once)
    uint32_t span = FIN_PACK_SPAN;
    static const uint32_t env_span = [] { const char* e = getenv("FINITO_PACK_SPAN"); return e ? (uint32_t)atoi(e) : 0u; }();   // (experiments; read once, not per launch)
    if (env_span) span = env_span;
    else {
        const uint64_t want_waves = 16384;
        const uint64_t s = (n_chunks / want_waves + 63) / 64 * 64;
        span = (uint32_t)(s < 256 ? 256 : s > FIN_PACK_SPAN ? FIN_PACK_SPAN : s);
    }
    if (span < 64u || span % 64u) span = FIN_PACK_SPAN;
    const uint64_t waves = (n_chunks + span - 1) / span;
    const uint64_t blocks = (waves * 64 + FIN_TPB - 1) / FIN_TPB;
    hipLaunchKernelGGL(fin_pack_reads_kernel, dim3((uint32_t)blocks), dim3(FIN_TPB), 0, stream, bases, offs, desc, (uint4*)packed, n_reads, n_chunks, span);
    return (int)hipGetLastError();
}
