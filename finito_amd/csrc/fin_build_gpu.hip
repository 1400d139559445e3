// fin_build_gpu.hip -- index construction ON THE DEVICE ("then accelerate"; every k the reference has: k <= 255).
//
// The same construction as fin_build.cpp (which it must equal bit for bit: tests/test_build_gpu.py compares the containers), as kernels:
// the reference builds this chain on the CPU -- `sbwt build`, lcs_basic_parallel_algorithm (lcs_basic_parallel_algorithm.hpp:52-120),
// permute_unitigs (PackedStrings.hh:105-135), FinimizerIndexBuilder::add_sequence (FinimizerIndex.hh:321-389) -- and the host builder of
// this package needs 7.5 s for 250 Mbp.  Here:
//   1  k-mers of every unitig as 2k-bit keys (base j at bits 2j: colex order = integer order), one lane per 256 text positions
//   2  rocPRIM radix sort + unique                                   -> the k-mer nodes in SBWT order
//   3  dummy nodes: every unitig whose first k-mer has no predecessor gives its k-1 proper prefixes; two stable radix sorts
//      (length, then key) + unique; their places between the k-mers by binary search
//   4  node bytes: LCS of neighbouring nodes = xor + clz of their keys (a lane per 64-node block)
//   5  planes: every node's one marked in-edge (binary searches in the sorted k-mers, atomicOr), C array, rank bases (scan)
//   6  permute_unitigs: stable radix sort of (first k-mer, input number); Ustart marks; ends; 2-bit text
//   7  finimizers: the unitig text streamed through the plain search (fin_kernel_b.hip's machinery, a lane per 512 positions with 2k of
//      run-up), the reference's overwrite rule (FinimizerIndex.hh:370-378) as an atomic max per node -- as in fin_build.cpp
//   8  dictionaries' masks and ranks (scan), global offsets compacted in rank order, thermometer planes, sampling
// and the result is copied back into the host-side fin_index (everything downstream -- save, export, upload -- is unchanged).
// Keys are 64-bit integers for k <= 32 and 128-bit ones (two words; rocPRIM sorts them as they are) for 33 <= k <= 64: every kernel that
// touches a key is a template over the key type; above 64 the key is a struct of four (k <= 128) or eight 64-bit words (GbWide) that rocPRIM sorts
// through a decomposer, and above 128 the exact LCS bytes are made beside the 7-bit ones (FinDevIndex::lcs8).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "fin_device.h"
#include "fin_index.hpp"

#define GB_SEG 256u    // text positions per lane (k-mer extraction)
#define GB_FSEG 512u   // text positions per lane (finimizer pass)

namespace {

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t b) { if (p) { (void)hipFree(p); p = nullptr; } bytes = b; return hipMalloc(&p, b ? b : 16); }
    template <typename T> T* as() const { return (T*)p; }
};

typedef __uint128_t gb_key128;
// keys of W 64-bit words for k > 64 (W = 4: k <= 128, W = 8: k <= 255), w[0] the least significant: the operators the kernels below use on a key, so that
// every one of them is the same template for 64-bit, 128-bit and wide keys (rocPRIM sorts a wide key through a decomposer, sort_keys_of / sort_pairs)
template <int W>
struct GbWide {
    uint64_t w[W];
    __host__ __device__ __forceinline__ GbWide() {}
    __host__ __device__ __forceinline__ GbWide(uint64_t v) { w[0] = v; for (int i = 1; i < W; i++) w[i] = 0; }
    __host__ __device__ __forceinline__ explicit operator uint64_t() const { return w[0]; }
    __host__ __device__ __forceinline__ GbWide operator<<(int s) const {   // 0 <= s < 64 W
        GbWide r; const int ws = s >> 6, bs = s & 63;
        for (int i = W - 1; i >= 0; i--) {
            const int j = i - ws;
            uint64_t v = 0;
            if (j >= 0) { v = w[j] << bs; if (bs && j >= 1) v |= w[j - 1] >> (64 - bs); }
            r.w[i] = v;
        }
        return r;
    }
    __host__ __device__ __forceinline__ GbWide operator>>(int s) const {
        GbWide r; const int ws = s >> 6, bs = s & 63;
        for (int i = 0; i < W; i++) {
            const int j = i + ws;
            uint64_t v = 0;
            if (j < W) { v = w[j] >> bs; if (bs && j + 1 < W) v |= w[j + 1] << (64 - bs); }
            r.w[i] = v;
        }
        return r;
    }
    __host__ __device__ __forceinline__ GbWide operator|(const GbWide& o) const { GbWide r; for (int i = 0; i < W; i++) r.w[i] = w[i] | o.w[i]; return r; }
    __host__ __device__ __forceinline__ GbWide operator&(const GbWide& o) const { GbWide r; for (int i = 0; i < W; i++) r.w[i] = w[i] & o.w[i]; return r; }
    __host__ __device__ __forceinline__ GbWide operator^(const GbWide& o) const { GbWide r; for (int i = 0; i < W; i++) r.w[i] = w[i] ^ o.w[i]; return r; }
    __host__ __device__ __forceinline__ GbWide operator~() const { GbWide r; for (int i = 0; i < W; i++) r.w[i] = ~w[i]; return r; }
    __host__ __device__ __forceinline__ GbWide& operator|=(const GbWide& o) { for (int i = 0; i < W; i++) w[i] |= o.w[i]; return *this; }
    __host__ __device__ __forceinline__ bool operator==(const GbWide& o) const { bool e = true; for (int i = 0; i < W; i++) e = e && w[i] == o.w[i]; return e; }
    __host__ __device__ __forceinline__ bool operator!=(const GbWide& o) const { return !(*this == o); }
    __host__ __device__ __forceinline__ bool operator<(const GbWide& o) const {
        for (int i = W - 1; i >= 0; i--) if (w[i] != o.w[i]) return w[i] < o.w[i];
        return false;
    }
    __host__ __device__ static __forceinline__ GbWide low_bits(int bits) {
        GbWide r;
        for (int i = 0; i < W; i++) { const int lo = 64 * i; r.w[i] = bits >= lo + 64 ? ~0ull : (bits <= lo ? 0ull : ((1ull << (bits - lo)) - 1ull)); }
        return r;
    }
};
template <typename Key> struct GbIsWide { static constexpr bool value = false; };
template <int W> struct GbIsWide<GbWide<W>> { static constexpr bool value = true; };
template <int W> struct GbWideDecomposer;   // the key's words as a tuple of references, the most significant first (rocPRIM's order)
template <> struct GbWideDecomposer<4> {
    __host__ __device__ rocprim::tuple<uint64_t&, uint64_t&, uint64_t&, uint64_t&> operator()(GbWide<4>& k) const {
        return rocprim::tuple<uint64_t&, uint64_t&, uint64_t&, uint64_t&>(k.w[3], k.w[2], k.w[1], k.w[0]);
    }
};
template <> struct GbWideDecomposer<8> {
    __host__ __device__ rocprim::tuple<uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&> operator()(GbWide<8>& k) const {
        return rocprim::tuple<uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&, uint64_t&>(k.w[7], k.w[6], k.w[5], k.w[4], k.w[3], k.w[2], k.w[1], k.w[0]);
    }
};
template <typename Key> struct GbKeyBits { static constexpr int value = (int)sizeof(Key) * 8; };
template <typename Key> __host__ __device__ __forceinline__ Key gb_mask(int bits) {
    if constexpr (GbIsWide<Key>::value) return Key::low_bits(bits);
    else return bits >= GbKeyBits<Key>::value ? ~(Key)0 : (((Key)1 << bits) - (Key)1);
}
__device__ __forceinline__ int gb_clz(uint64_t x) { return __clzll((long long)x); }
__device__ __forceinline__ int gb_clz(gb_key128 x) { const uint64_t hi = (uint64_t)(x >> 64); return hi ? __clzll((long long)hi) : 64 + __clzll((long long)(uint64_t)x); }
template <int W> __device__ __forceinline__ int gb_clz(const GbWide<W>& x) {
    for (int i = W - 1; i >= 0; i--) if (x.w[i]) return 64 * (W - 1 - i) + __clzll((long long)x.w[i]);
    return 64 * W;
}
template <typename Key>
struct GbKmers {   // the sorted distinct k-mers with a bucket index over their top bits, and the dummies between them
    const Key* kmers; uint64_t m;
    const uint32_t* bk; uint32_t shift; uint32_t B;   // bk[b] = first k-mer whose key >> shift >= b
    const Key* dk; const uint32_t* dl; const uint32_t* dpos; uint32_t D;   // dummies (pkey, len), dpos[d] = k-mers before dummy d
};
template <typename Key>
__device__ __forceinline__ uint64_t gb_lower_bound(const GbKmers<Key>& g, Key key) {
    uint64_t lo, hi;
    if (g.B) { const uint64_t b = (uint64_t)(key >> g.shift); lo = g.bk[b]; hi = g.bk[b + 1]; } else { lo = 0; hi = g.m; }
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (g.kmers[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// node index of k-mer rank r: r plus the dummies placed at or before it (dpos[d] <= r)
template <typename Key>
__device__ __forceinline__ uint64_t gb_node_of_rank(const GbKmers<Key>& g, uint64_t r) {
    uint32_t lo = 0, hi = g.D;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (g.dpos[mid] <= r) lo = mid + 1; else hi = mid; }
    return r + lo;
}
template <typename Key>
__device__ __forceinline__ int64_t gb_find_dummy(const GbKmers<Key>& g, Key pkey, uint32_t len) {
    uint32_t lo = 0, hi = g.D;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const bool less = g.dk[mid] != pkey ? g.dk[mid] < pkey : g.dl[mid] < len;
        if (less) lo = mid + 1; else hi = mid;
    }
    return (lo < g.D && g.dk[lo] == pkey && g.dl[lo] == len) ? (int64_t)lo : -1;
}

__global__ __launch_bounds__(256) void gb_encode_kernel(const char* ascii, uint64_t total, uint8_t* codes, uint32_t* bad) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint8_t c = (uint8_t)ascii[i] & (uint8_t)~32u;
    uint8_t v = 0;
    if (c == 'A') v = 0; else if (c == 'C') v = 1; else if (c == 'G') v = 2; else if (c == 'T') v = 3; else atomicOr(bad, 1u);
    codes[i] = v;
}
__device__ __forceinline__ uint32_t gb_unitig_of(const uint64_t* offs, uint32_t nu, uint64_t g) {   // u with offs[u] <= g < offs[u+1]
    uint32_t lo = 0, hi = nu;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (offs[mid + 1] <= g) lo = mid + 1; else hi = mid; }
    return lo;
}
// raw[p - u(k-1)] = key of the k-mer that starts at input position p of unitig u
template <typename Key>
__global__ __launch_bounds__(256) void gb_kmers_kernel(const uint8_t* codes, const uint64_t* offs, uint32_t nu, int k, uint64_t total, Key* raw) {
    const uint64_t s0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * GB_SEG;   // k-mer END positions [s0, s1)
    if (s0 >= total) return;
    const uint64_t s1 = s0 + GB_SEG < total ? s0 + GB_SEG : total;
    const int kb = 2 * k;
    const Key mask = gb_mask<Key>(kb);
    uint32_t u = gb_unitig_of(offs, nu, s0);
    uint64_t ustart = offs[u], uend = offs[u + 1];
    uint64_t g = ustart;
    if (s0 >= (uint64_t)(k - 1) && s0 - (uint64_t)(k - 1) > g) g = s0 - (uint64_t)(k - 1);
    Key key = 0; uint32_t depth = 0;
    for (; g < s1; g++) {
        while (g >= uend) { u++; ustart = uend; uend = offs[u + 1]; depth = 0; }
        key = ((key >> 2) | ((Key)codes[g] << (kb - 2))) & mask;
        depth++;
        if (depth >= (uint32_t)k && g >= s0) raw[(g - (uint64_t)(k - 1)) - (uint64_t)u * (uint64_t)(k - 1)] = key;
    }
}
template <typename Key>
__global__ __launch_bounds__(256) void gb_heads_kernel(const Key* sorted, uint64_t n, uint32_t* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
}
template <typename Key>
__global__ __launch_bounds__(256) void gb_scatter64_kernel(const Key* in, const uint32_t* flag, const uint32_t* pos, uint64_t n, Key* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = in[i];
}
template <typename Key>
__global__ __launch_bounds__(256) void gb_buckets_kernel(const Key* kmers, uint64_t m, uint32_t shift, uint32_t nb, uint32_t* bk) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b > nb) return;
    if (b == nb) { bk[b] = (uint32_t)m; return; }
    const Key key = (Key)b << shift;
    uint64_t lo = 0, hi = m;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (kmers[mid] < key) lo = mid + 1; else hi = mid; }
    bk[b] = (uint32_t)lo;
}
// first k-mer of every unitig; dummy slots of the unitigs whose first k-mer has no predecessor (slot 0 = the root)
template <typename Key>
__global__ __launch_bounds__(256) void gb_first_kernel(GbKmers<Key> g, const uint8_t* codes, const uint64_t* offs, uint32_t nu, int k, Key* fk, uint32_t* iota,
                                                       Key* dk, uint32_t* dl) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= nu) return;
    const int kb = 2 * k;
    const Key mask_k = gb_mask<Key>(kb), mask_p = gb_mask<Key>(kb - 2);
    Key X = 0;
    for (int j = 0; j < k; j++) X |= (Key)codes[offs[u] + (uint64_t)j] << (2 * j);
    fk[u] = X; iota[u] = u;
    const Key P = X & mask_p, q = P << 2;
    const uint64_t r = gb_lower_bound(g, q);
    const bool has_pred = r < g.m && (g.kmers[r] >> 2) == P;
    const size_t base = 1 + (size_t)u * (size_t)(k - 1);
    for (int j = 1; j < k; j++) {
        dk[base + (size_t)(j - 1)] = has_pred ? ~(Key)0 : ((X << (2 * (k - j))) & mask_k);
        dl[base + (size_t)(j - 1)] = has_pred ? 0xFFFFFFFFu : (uint32_t)j;
    }
    if (u == 0) { dk[0] = (Key)0; dl[0] = 0u; }
}
template <typename Key>
__global__ __launch_bounds__(256) void gb_dheads_kernel(const Key* dk, const uint32_t* dl, uint64_t n, uint32_t* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (dk[i] != ~(Key)0 && (i == 0 || dk[i] != dk[i - 1] || dl[i] != dl[i - 1])) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void gb_scatter32_kernel(const uint32_t* in, const uint32_t* flag, const uint32_t* pos, uint64_t n, uint32_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = in[i];
}
template <typename Key>
__global__ __launch_bounds__(256) void gb_dpos_kernel(GbKmers<Key> g, uint32_t* dpos) {
    const uint32_t d = blockIdx.x * 256 + threadIdx.x;
    if (d < g.D) dpos[d] = (uint32_t)gb_lower_bound(g, g.dk[d]);
}
// node bytes: LCS[i] = common suffix length of node i and node i-1 ('$' never extends a match); a lane per block
template <typename Key>
__global__ __launch_bounds__(256) void gb_lcs_kernel(GbKmers<Key> g, int k, uint64_t n, FinNodeBlock* blocks, uint8_t* lcs8 /* k > FIN_FAST_K: the exact values, a byte per node; else null */) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t s = b * 64;
    if (s >= n) return;
    const uint64_t e = s + 64 < n ? s + 64 : n, from = s == 0 ? 0 : s - 1;
    const int kb = 2 * k;
    uint32_t lo = 0, hi = g.D;   // dummies among the first `from` nodes: smallest d with d + dpos[d] >= from
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)mid + g.dpos[mid] < from) lo = mid + 1; else hi = mid; }
    uint32_t d = lo; uint64_t r = from - d;
    Key prev_key = 0; uint32_t prev_len = 0;
    uint8_t bytes[64];
    for (int j = 0; j < 64; j++) bytes[j] = 0;
    for (uint64_t i = from; i < e; i++) {
        Key key; uint32_t len;
        if (d < g.D && (uint64_t)d + g.dpos[d] == i) { key = g.dk[d]; len = g.dl[d]; d++; }
        else { key = g.kmers[r]; len = (uint32_t)k; r++; }
        if (i >= s) {
            uint32_t lcs = 0;
            if (i > 0) {
                const Key x = key ^ prev_key;
                const uint32_t match = x == 0 ? (uint32_t)k : (uint32_t)((gb_clz(x) - (GbKeyBits<Key>::value - kb)) / 2);
                lcs = min(match, min(len, prev_len));
            }
            bytes[i - s] = (uint8_t)min(lcs, (uint32_t)FIN_LCS_MASK);
            if (lcs8) lcs8[i] = (uint8_t)lcs;
        }
        prev_key = key; prev_len = len;
    }
    uint64_t* dst = (uint64_t*)blocks[b].node;
    for (int w = 0; w < 8; w++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v |= (uint64_t)bytes[8 * w + j] << (8 * j);
        dst[w] = v;
    }
}
__device__ __forceinline__ void gb_set_plane(FinNodeBlock* blocks, int c, uint64_t u) {
    const uint32_t o = (uint32_t)(u & 63);
    atomicOr(o < 32 ? &blocks[u >> 6].rec[c].plane_lo : &blocks[u >> 6].rec[c].plane_hi, 1u << (o & 31));
}
// every k-mer node's marked in-edge: labelled with its last char, leaving the first node of the group whose (k-1)-suffix equals its (k-1)-prefix
template <typename Key>
__global__ __launch_bounds__(256) void gb_planes_kmers_kernel(GbKmers<Key> g, int k, FinNodeBlock* blocks) {
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= g.m) return;
    const int kb = 2 * k;
    const Key mask_k = gb_mask<Key>(kb), mask_p = gb_mask<Key>(kb - 2);
    const Key X = g.kmers[v];
    const int c = (int)(uint64_t)(X >> (kb - 2)) & 3;
    const Key P = X & mask_p, q = P << 2;
    const uint64_t r = gb_lower_bound(g, q);
    uint64_t u;
    if (r < g.m && (g.kmers[r] >> 2) == P) u = gb_node_of_rank(g, r);
    else {
        int64_t d = gb_find_dummy(g, (X << 2) & mask_k, (uint32_t)(k - 1));
        if (d < 0) d = 0;   // cannot happen: step 3 created it
        u = (uint64_t)d + g.dpos[d];
    }
    gb_set_plane(blocks, c, u);
}
template <typename Key>
__global__ __launch_bounds__(256) void gb_planes_dummies_kernel(GbKmers<Key> g, int k, FinNodeBlock* blocks) {
    const uint32_t d = blockIdx.x * 256 + threadIdx.x;
    if (d == 0 || d >= g.D) return;
    const int kb = 2 * k;
    const Key mask_k = gb_mask<Key>(kb);
    const Key pk = g.dk[d]; const uint32_t j = g.dl[d];
    const int c = (int)(uint64_t)(pk >> (kb - 2)) & 3;
    int64_t p = gb_find_dummy(g, (pk << 2) & mask_k, j - 1);
    if (p < 0) p = 0;
    gb_set_plane(blocks, c, (uint64_t)p + g.dpos[p]);
}
__global__ __launch_bounds__(256) void gb_popc_kernel(const FinNodeBlock* blocks, uint64_t nblk, uint32_t* pop) {   // pop[c * nblk + b]
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    for (int c = 0; c < 4; c++) pop[(size_t)c * nblk + b] = (uint32_t)__popcll(fin_plane(blocks[b].rec[c]));
}
__global__ __launch_bounds__(256) void gb_bases_kernel(FinNodeBlock* blocks, uint64_t nblk, const uint32_t* ex, uint32_t C0, uint32_t C1, uint32_t C2, uint32_t C3) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    const uint32_t C[4] = {C0, C1, C2, C3};
    for (int c = 0; c < 4; c++) blocks[b].rec[c].base = C[c] + ex[(size_t)c * nblk + b];
}
// permuted unitig r = input unitig perm[r]: Ustart mark on the node of its first k-mer, its length
template <typename Key>
__global__ __launch_bounds__(256) void gb_ustart_kernel(GbKmers<Key> g, const Key* fk_sorted, const uint32_t* perm, const uint64_t* offs, uint32_t nu, FinNodeBlock* blocks,
                                                        uint32_t* len_perm, uint32_t* bad) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nu) return;
    const Key X = fk_sorted[r];
    const uint64_t i = gb_lower_bound(g, X);
    if (!(i < g.m && g.kmers[i] == X)) { atomicOr(bad, 2u); return; }
    const uint64_t node = gb_node_of_rank(g, i);
    uint8_t* byte = &blocks[node >> 6].node[node & 63];
    uint32_t* word = (uint32_t*)((uintptr_t)byte & ~(uintptr_t)3);
    atomicOr(word, (uint32_t)FIN_USTART_BIT << (8 * ((uintptr_t)byte & 3)));
    const uint32_t u = perm[r];
    len_perm[r] = (uint32_t)(offs[u + 1] - offs[u]);
}
__global__ __launch_bounds__(256) void gb_ends_kernel(const uint32_t* ustart_ex, const uint32_t* len_perm, uint32_t nu, uint32_t* ends) {   // ends_p layout
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) ends[0] = 0;
    if (r < nu) ends[r + 1] = ustart_ex[r] + len_perm[r];
    if (r < 8) ends[nu + 1 + r] = 0xFFFFFFFFu;
}
// 2-bit text of the permuted unitigs, a lane per output word (16 bases)
__global__ __launch_bounds__(256) void gb_concat_kernel(const uint8_t* codes, const uint64_t* offs, const uint32_t* perm, const uint32_t* ends, uint32_t nu, uint64_t total_len,
                                                        uint32_t* concat, uint64_t n_words) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const uint64_t g0 = w * 16;
    if (g0 >= total_len) { concat[w] = 0; return; }
    uint32_t lo = 0, hi = nu;   // r with ends[r] <= g0 < ends[r+1]
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ends[mid + 1] <= g0) lo = mid + 1; else hi = mid; }
    uint32_t r = lo;
    uint32_t val = 0;
    for (uint64_t g = g0; g < g0 + 16 && g < total_len; g++) {
        while (ends[r + 1] <= g) r++;
        val |= (uint32_t)codes[offs[perm[r]] + (g - ends[r])] << (2 * (g & 15));
    }
    concat[w] = val;
}

// ---- finimizers: add_sequence over the permuted text (FinimizerIndex.hh:321-389) -- the plain streaming search of fin_kernel_b.hip with
//      the overwrite rule as an atomic max: per node the event with the largest in-unitig end, the first in text order on ties, an end
//      of 0 reading as "unset" (fin_build.cpp step 8) ----
struct FLdsDeque {
    static constexpr uint32_t CAP = 16;
    uint64_t* base; uint32_t limit;
    __device__ __forceinline__ uint64_t get(uint32_t i) const { return base[(i & (CAP - 1)) * FIN_TPB]; }
    __device__ __forceinline__ void set(uint32_t i, uint64_t v) { base[(i & (CAP - 1)) * FIN_TPB] = v; }
};
struct FGlobalDeque {
    static constexpr uint32_t CAP = 256;
    uint64_t* base; uint64_t stride; uint32_t limit;
    __device__ __forceinline__ uint64_t get(uint32_t i) const { return base[(uint64_t)(i & (CAP - 1)) * stride]; }
    __device__ __forceinline__ void set(uint32_t i, uint64_t v) { base[(uint64_t)(i & (CAP - 1)) * stride] = v; }
};
template <typename DQ>
__device__ bool fmin_segment(const FinDevIndex& ix, uint32_t s0, uint32_t s1, unsigned long long* best, DQ dq) {
    const uint32_t n = ix.n_nodes;
    const int k = (int)ix.k;
    uint32_t u = ix.samp[s0 >> ix.samp_shift];
    while (ix.ends[u + 1] <= s0) u++;
    uint32_t ustart = ix.ends[u], uend = ix.ends[u + 1];
    uint32_t g = ustart;
    if (s0 >= (uint32_t)(2 * k) && s0 - (uint32_t)(2 * k) > g) g = s0 - (uint32_t)(2 * k);
    uint32_t il = 0, ir = n - 1;
    uint32_t start = g, kstart = g;   // (the builder's search keeps no k-mer interval: a k-mer ends wherever k bases of a unitig are behind)
    uint32_t dq_head = 0, dq_cnt = 0;
    for (; g < s1; g++) {
        if (g >= uend) {
            do { u++; ustart = uend; uend = ix.ends[u + 1]; } while (g >= uend);
            il = 0; ir = n - 1; start = g; kstart = g; dq_head = 0; dq_cnt = 0;
        }
        const uint32_t c = d_concat(ix, g);
        uint32_t nl, nr;
        bool ok = d_extend(ix, c, il, ir, nl, nr);
        while (!ok) {   // (unreachable on a consistent index: every substring of a unitig is in the SBWT)
            ++start;
            if (start > g) { nl = 0; nr = n - 1; break; }
            d_drop(ix, (int)(g - start), il, ir);
            ok = d_extend(ix, c, il, ir, nl, nr);
        }
        il = nl; ir = nr;
        if (g - kstart + 1 > (uint32_t)k) kstart = g + 1 - (uint32_t)k;   // the k-mer window
        while (dq_cnt) {
            const uint64_t f = dq.get(dq_head);
            if (dq_end(f, g) - dq_len(f) + 1 < kstart) { dq_head++; dq_cnt--; } else break;
        }
        if (il == ir) {
            uint32_t cl = 0, cc = 0;
            do {
                cl = g - start + 1; cc = il;
                start++;
                d_drop(ix, (int)(g - start + 1), il, ir);
            } while (il == ir);
            const uint64_t cand = dq_pack(cl, cc, g);
            if (dq_cnt && (dq.get(dq_head) >> 24) > (cand >> 24)) dq_cnt = 0;
            else { while (dq_cnt && (dq.get(dq_head + dq_cnt - 1) >> 24) > (cand >> 24)) dq_cnt--; }
            if (dq_cnt >= dq.limit) return false;
            dq.set(dq_head + dq_cnt, cand); dq_cnt++;
        }
        if (g - ustart + 1 >= (uint32_t)k && g >= s0 && dq_cnt) {
            const uint64_t w = dq.get(dq_head);
            const uint32_t fin_end = dq_end(w, g), fin_colex = dq_colex(w);
            const uint32_t end_in = fin_end - ustart;
            const unsigned long long key = end_in > 0 ? (((unsigned long long)end_in << 32) | (0xFFFFFFFFull - fin_end)) : (unsigned long long)fin_end + 1ull;
            atomicMax(&best[fin_colex], key);
        }
    }
    return true;
}
__global__ __launch_bounds__(FIN_TPB) void gb_fmin_kernel(FinDevIndex ix, unsigned long long* best, uint32_t n_seg, uint32_t* ovf_list, uint32_t* ovf_count) {
    __shared__ uint64_t lds_dq[FLdsDeque::CAP * FIN_TPB];
    const uint32_t seg = blockIdx.x * FIN_TPB + threadIdx.x;
    if (seg >= n_seg) return;
    const uint64_t s0 = (uint64_t)seg * GB_FSEG;
    const uint32_t s1 = (uint32_t)(s0 + GB_FSEG < ix.total_len ? s0 + GB_FSEG : ix.total_len);
    FLdsDeque dq{lds_dq + threadIdx.x, FLdsDeque::CAP};
    if (!fmin_segment<FLdsDeque>(ix, (uint32_t)s0, s1, best, dq)) ovf_list[atomicAdd(ovf_count, 1u)] = seg;
}
__global__ __launch_bounds__(FIN_TPB) void gb_fmin_overflow_kernel(FinDevIndex ix, unsigned long long* best, const uint32_t* ovf_list, const uint32_t* ovf_count, uint64_t* scratch) {
    const uint32_t nthreads = gridDim.x * FIN_TPB, tid = blockIdx.x * FIN_TPB + threadIdx.x;
    const uint32_t cnt = *ovf_count;
    FGlobalDeque dq{scratch + tid, nthreads, FGlobalDeque::CAP};
    for (uint32_t i = tid; i < cnt; i += nthreads) {
        const uint64_t s0 = (uint64_t)ovf_list[i] * GB_FSEG;
        const uint32_t s1 = (uint32_t)(s0 + GB_FSEG < ix.total_len ? s0 + GB_FSEG : ix.total_len);
        (void)fmin_segment<FGlobalDeque>(ix, (uint32_t)s0, s1, best, dq);
    }
}
// per block: the dictionaries' masks and how many bits each holds
__global__ __launch_bounds__(256) void gb_masks_kernel(const FinNodeBlock* blocks, const unsigned long long* best, uint64_t n, uint64_t nblk, FinBlockInfo* info, uint32_t* cnt_f, uint32_t* cnt_u) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    const uint64_t lim = n - b * 64 < 64 ? n - b * 64 : 64;
    uint64_t fm = 0, um = 0;
    for (uint64_t j = 0; j < lim; j++) {
        if (best[b * 64 + j]) fm |= 1ull << j;
        if (blocks[b].node[j] & FIN_USTART_BIT) um |= 1ull << j;
    }
    info[b].fmin_mask_lo = (uint32_t)fm; info[b].fmin_mask_hi = (uint32_t)(fm >> 32);
    info[b].ustart_mask_lo = (uint32_t)um; info[b].ustart_mask_hi = (uint32_t)(um >> 32);
    cnt_f[b] = (uint32_t)__popcll(fm); cnt_u[b] = (uint32_t)__popcll(um);
}
__global__ __launch_bounds__(256) void gb_ranks_kernel(const unsigned long long* best, uint64_t nblk, FinBlockInfo* info, const uint32_t* ex_f, const uint32_t* ex_u, uint32_t tot_f, uint32_t tot_u, uint32_t* goff) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk + 2) return;
    if (b >= nblk) { info[b] = FinBlockInfo{tot_f, 0, 0, 0, 0, tot_u}; return; }
    info[b].fmin_rank = ex_f[b]; info[b].ustart_rank = ex_u[b];
    uint64_t fm = info[b].fmin_mask_lo | ((uint64_t)info[b].fmin_mask_hi << 32);
    uint32_t at = ex_f[b];
    while (fm) {
        const int j = __ffsll((long long)fm) - 1;
        fm &= fm - 1;
        const unsigned long long key = best[b * 64 + (uint64_t)j];
        goff[at++] = (key >> 32) ? (uint32_t)(0xFFFFFFFFull - (key & 0xFFFFFFFFull)) : (uint32_t)(key - 1);
    }
}
__global__ __launch_bounds__(256) void gb_hist_kernel(const FinNodeBlock* blocks, uint64_t n, uint64_t nblk, unsigned long long* hist) {
    __shared__ uint32_t h[128];
    if (threadIdx.x < 128) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < nblk; b += (uint64_t)gridDim.x * 256) {
        const uint64_t lim = n - b * 64 < 64 ? n - b * 64 : 64;
        for (uint64_t j = 0; j < lim; j++) atomicAdd(&h[blocks[b].node[j] & FIN_LCS_MASK], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 128 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void gb_thermo_kernel(FinNodeBlock* blocks, uint64_t nblk, int t0) {
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    uint64_t p0 = 0, p1 = 0;
    for (int j = 0; j < 64; j++) {
        int c = (int)(blocks[b].node[j] & FIN_LCS_MASK) - t0;
        c = c < 0 ? 0 : (c > 3 ? 3 : c);
        p0 |= (uint64_t)(c & 1) << j; p1 |= (uint64_t)(c >> 1) << j;
    }
    blocks[b].th0 = p0; blocks[b].th1 = p1;
}

#define GBCHK(call)                                                                                      \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return -3; }    \
    } while (0)
static inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + 255) / 256)); }

// stable radix sort of (keys, values) in place through scratch copies
template <typename K, typename V>
static hipError_t radix_pairs(void* t, size_t& need, K* keys, K* k2, V* vals, V* v2, uint64_t n, unsigned bits) {
    if constexpr (GbIsWide<K>::value) return rocprim::radix_sort_pairs(t, need, keys, k2, vals, v2, n, GbWideDecomposer<(int)(sizeof(K) / 8)>{}, 0u, bits);
    else return rocprim::radix_sort_pairs(t, need, keys, k2, vals, v2, n, 0, bits);
}
template <typename K>
static hipError_t radix_keys(void* t, size_t& need, K* in, K* out, uint64_t n, unsigned bits) {
    if constexpr (GbIsWide<K>::value) return rocprim::radix_sort_keys(t, need, in, out, n, GbWideDecomposer<(int)(sizeof(K) / 8)>{}, 0u, bits);
    else return rocprim::radix_sort_keys(t, need, in, out, n, 0, bits);
}
template <typename K, typename V>
static hipError_t sort_pairs(K* keys, V* vals, uint64_t n, unsigned bits, DevBuf& tmp) {
    DevBuf k2, v2;
    hipError_t e;
    if ((e = k2.alloc(n * sizeof(K))) != hipSuccess || (e = v2.alloc(n * sizeof(V))) != hipSuccess) return e;
    size_t need = 0;
    if ((e = radix_pairs<K, V>(nullptr, need, keys, k2.as<K>(), vals, v2.as<V>(), n, bits)) != hipSuccess) return e;
    if (need > tmp.bytes && (e = tmp.alloc(need)) != hipSuccess) return e;
    if ((e = radix_pairs<K, V>(tmp.p, need, keys, k2.as<K>(), vals, v2.as<V>(), n, bits)) != hipSuccess) return e;
    if ((e = hipMemcpy(keys, k2.p, n * sizeof(K), hipMemcpyDeviceToDevice)) != hipSuccess) return e;
    return hipMemcpy(vals, v2.p, n * sizeof(V), hipMemcpyDeviceToDevice);
}
static hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, DevBuf& tmp) {
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, in, out, 0u, n, rocprim::plus<uint32_t>());
    if (e != hipSuccess) return e;
    if (need > tmp.bytes && (e = tmp.alloc(need)) != hipSuccess) return e;
    return rocprim::exclusive_scan(tmp.p, need, in, out, 0u, n, rocprim::plus<uint32_t>());
}
static hipError_t last_u32(const uint32_t* d, uint64_t n, uint32_t& v) { v = 0; return n ? hipMemcpy(&v, d + (n - 1), 4, hipMemcpyDeviceToHost) : hipSuccess; }

}  // namespace

// Builds the index of the unitigs on `device` and leaves it in `out` (host side, like fin_build_index).  0, or a negative error with a
// message: -1 bad input, -3 device error, -5 size limits.  phase_ms (may be null, 8 doubles): milliseconds per stage.
template <typename Key>
static int build_index_gpu_impl(const char* bases, const uint64_t* offsets, uint64_t n_unitigs, int k, int device, fin_index& out, std::string& err, double* phase_ms) {
    if (n_unitigs == 0) { err = "no unitigs"; return -1; }
    if (n_unitigs >= 0x7FFFFFFFull) { err = "too many unitigs for this build"; return -5; }
    const uint64_t base0 = offsets[0], total = offsets[n_unitigs] - base0;
    if (total >= 0xFFFFFFF0ull) { err = "index too large for one index: total unitig length >= 2^32 (use a partitioned index, fin_pindex_build_device)"; return -5; }
    const uint32_t nu = (uint32_t)n_unitigs;
    std::vector<uint64_t> offs(n_unitigs + 1);
    for (uint64_t u = 0; u <= n_unitigs; u++) offs[u] = offsets[u] - base0;
    for (uint64_t u = 0; u < n_unitigs; u++)
        if (offs[u + 1] - offs[u] < (uint64_t)k) { err = "unitig " + std::to_string(u) + " is shorter than k"; return -1; }
    GBCHK(hipSetDevice(device));
    hipEvent_t ev[10];
    for (auto& e : ev) GBCHK(hipEventCreate(&e));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < 10; i++) (void)hipEventDestroy(e[i]); } } evg{ev};
    int evn = 0;
    auto mark = [&]() { if (evn < 10) (void)hipEventRecord(ev[evn++], nullptr); };
    const int kb = 2 * k;
    DevBuf tmp, d_ascii, d_codes, d_offs, d_bad;
    mark();
    // ---- 1. upload, encode, k-mers ----
    GBCHK(d_ascii.alloc(total)); GBCHK(d_codes.alloc(total + 16)); GBCHK(d_offs.alloc((n_unitigs + 1) * 8)); GBCHK(d_bad.alloc(16));
    GBCHK(hipMemcpy(d_ascii.p, bases + base0, total, hipMemcpyHostToDevice));
    GBCHK(hipMemcpy(d_offs.p, offs.data(), (n_unitigs + 1) * 8, hipMemcpyHostToDevice));
    GBCHK(hipMemset(d_bad.p, 0, 16));
    hipLaunchKernelGGL(gb_encode_kernel, grid_for(total), dim3(256), 0, nullptr, d_ascii.as<char>(), total, d_codes.as<uint8_t>(), d_bad.as<uint32_t>());
    const uint64_t T = total - (uint64_t)nu * (uint64_t)(k - 1);
    DevBuf d_raw, d_sorted;
    GBCHK(d_raw.alloc(T * sizeof(Key))); GBCHK(d_sorted.alloc(T * sizeof(Key)));
    hipLaunchKernelGGL(gb_kmers_kernel<Key>, grid_for((total + GB_SEG - 1) / GB_SEG), dim3(256), 0, nullptr, d_codes.as<uint8_t>(), d_offs.as<uint64_t>(), nu, k, total, d_raw.as<Key>());
    GBCHK(hipGetLastError());
    { uint32_t bad = 0; GBCHK(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost)); if (bad) { err = "unitigs contain a base outside ACGT (the reference's PackedStrings throws here, PackedStrings.hh:57)"; return -1; } }
    (void)d_ascii.alloc(0);
    mark();
    // ---- 2. sort + unique ----
    {
        size_t need = 0;
        GBCHK(radix_keys<Key>(nullptr, need, d_raw.as<Key>(), d_sorted.as<Key>(), T, (unsigned)kb));
        GBCHK(tmp.alloc(need));
        GBCHK(radix_keys<Key>(tmp.p, need, d_raw.as<Key>(), d_sorted.as<Key>(), T, (unsigned)kb));
    }
    uint64_t m = 0;
    DevBuf d_kmers;
    {
        DevBuf d_flag, d_pos;
        GBCHK(d_flag.alloc(T * 4)); GBCHK(d_pos.alloc(T * 4));
        hipLaunchKernelGGL(gb_heads_kernel<Key>, grid_for(T), dim3(256), 0, nullptr, d_sorted.as<Key>(), T, d_flag.as<uint32_t>());
        GBCHK(exclusive_scan_u32(d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), T, tmp));
        uint32_t lp = 0, lf = 0;
        GBCHK(last_u32(d_pos.as<uint32_t>(), T, lp)); GBCHK(last_u32(d_flag.as<uint32_t>(), T, lf));
        m = (uint64_t)lp + lf;
        GBCHK(d_kmers.alloc(m * sizeof(Key) + 16));
        hipLaunchKernelGGL(gb_scatter64_kernel<Key>, grid_for(T), dim3(256), 0, nullptr, d_sorted.as<Key>(), d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), T, d_kmers.as<Key>());
        GBCHK(hipGetLastError());
        GBCHK(hipDeviceSynchronize());
    }
    (void)d_raw.alloc(0); (void)d_sorted.alloc(0);
    GbKmers<Key> g{};
    g.kmers = d_kmers.as<Key>(); g.m = m;
    int B = kb - 2; if (B > 20) B = 20;
    while (B > 0 && (m >> B) < 32) B--;
    g.B = (uint32_t)B; g.shift = (uint32_t)(kb - B);
    DevBuf d_bk;
    GBCHK(d_bk.alloc(((1ull << B) + 2) * 4));
    if (B) hipLaunchKernelGGL(gb_buckets_kernel<Key>, grid_for((1ull << B) + 1), dim3(256), 0, nullptr, g.kmers, m, g.shift, 1u << B, d_bk.as<uint32_t>());
    g.bk = d_bk.as<uint32_t>();
    mark();
    // ---- 3. first k-mers, dummies ----
    const uint64_t nd_slots = 1 + (uint64_t)nu * (uint64_t)(k - 1);
    DevBuf d_fk, d_iota, d_dk, d_dl;
    GBCHK(d_fk.alloc((uint64_t)nu * sizeof(Key))); GBCHK(d_iota.alloc((uint64_t)nu * 4)); GBCHK(d_dk.alloc(nd_slots * sizeof(Key))); GBCHK(d_dl.alloc(nd_slots * 4));
    g.dk = nullptr; g.dl = nullptr; g.dpos = nullptr; g.D = 0;
    hipLaunchKernelGGL(gb_first_kernel<Key>, grid_for(nu), dim3(256), 0, nullptr, g, d_codes.as<uint8_t>(), d_offs.as<uint64_t>(), nu, k, d_fk.as<Key>(), d_iota.as<uint32_t>(),
                       d_dk.as<Key>(), d_dl.as<uint32_t>());
    GBCHK(hipGetLastError());
    // order by (pkey, len): stable sort by len, then by pkey
    GBCHK((sort_pairs<uint32_t, Key>(d_dl.as<uint32_t>(), d_dk.as<Key>(), nd_slots, 32, tmp)));
    GBCHK((sort_pairs<Key, uint32_t>(d_dk.as<Key>(), d_dl.as<uint32_t>(), nd_slots, (unsigned)GbKeyBits<Key>::value, tmp)));   // (every bit: the slots that stay empty hold all ones)
    uint64_t D = 0;
    DevBuf d_udk, d_udl, d_dpos;
    {
        DevBuf d_flag, d_pos;
        GBCHK(d_flag.alloc(nd_slots * 4)); GBCHK(d_pos.alloc(nd_slots * 4));
        hipLaunchKernelGGL(gb_dheads_kernel<Key>, grid_for(nd_slots), dim3(256), 0, nullptr, d_dk.as<Key>(), d_dl.as<uint32_t>(), nd_slots, d_flag.as<uint32_t>());
        GBCHK(exclusive_scan_u32(d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), nd_slots, tmp));
        uint32_t lp = 0, lf = 0;
        GBCHK(last_u32(d_pos.as<uint32_t>(), nd_slots, lp)); GBCHK(last_u32(d_flag.as<uint32_t>(), nd_slots, lf));
        D = (uint64_t)lp + lf;
        GBCHK(d_udk.alloc(D * sizeof(Key) + 16)); GBCHK(d_udl.alloc(D * 4 + 16)); GBCHK(d_dpos.alloc(D * 4 + 16));
        hipLaunchKernelGGL(gb_scatter64_kernel<Key>, grid_for(nd_slots), dim3(256), 0, nullptr, d_dk.as<Key>(), d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), nd_slots, d_udk.as<Key>());
        hipLaunchKernelGGL(gb_scatter32_kernel, grid_for(nd_slots), dim3(256), 0, nullptr, d_dl.as<uint32_t>(), d_flag.as<uint32_t>(), d_pos.as<uint32_t>(), nd_slots, d_udl.as<uint32_t>());
        GBCHK(hipGetLastError());
        GBCHK(hipDeviceSynchronize());
    }
    (void)d_dk.alloc(0); (void)d_dl.alloc(0);
    g.dk = d_udk.as<Key>(); g.dl = d_udl.as<uint32_t>(); g.D = (uint32_t)D;
    hipLaunchKernelGGL(gb_dpos_kernel<Key>, grid_for(D), dim3(256), 0, nullptr, g, d_dpos.as<uint32_t>());
    g.dpos = d_dpos.as<uint32_t>();
    const uint64_t n = m + D;
    if (n >= 0xFFFFFFC0ull) { err = "index too large for one index: n_nodes >= 2^32 (use a partitioned index, fin_pindex_build_device)"; return -5; }
    const uint64_t nblk = (n + 63) / 64;
    mark();
    // ---- 4./5. node bytes, planes, C array, bases ----
    DevBuf d_blocks;
    GBCHK(d_blocks.alloc(nblk * sizeof(FinNodeBlock)));
    GBCHK(hipMemset(d_blocks.p, 0, nblk * sizeof(FinNodeBlock)));
    FinNodeBlock* blocks = d_blocks.as<FinNodeBlock>();
    DevBuf d_lcs8;   // k > FIN_FAST_K: the exact LCS values (the node bytes hold min(LCS, 127)) -- for the finimizer pass below and the container
    if (k > FIN_FAST_K) { GBCHK(d_lcs8.alloc(n + 64)); GBCHK(hipMemset(d_lcs8.p, 0, n + 64)); }
    hipLaunchKernelGGL(gb_lcs_kernel<Key>, grid_for(nblk), dim3(256), 0, nullptr, g, k, n, blocks, k > FIN_FAST_K ? d_lcs8.as<uint8_t>() : (uint8_t*)nullptr);
    hipLaunchKernelGGL(gb_planes_kmers_kernel<Key>, grid_for(m), dim3(256), 0, nullptr, g, k, blocks);
    hipLaunchKernelGGL(gb_planes_dummies_kernel<Key>, grid_for(D), dim3(256), 0, nullptr, g, k, blocks);
    GBCHK(hipGetLastError());
    uint64_t C[4] = {1, 0, 0, 0};
    {
        DevBuf d_pop, d_ex;
        GBCHK(d_pop.alloc(4 * nblk * 4)); GBCHK(d_ex.alloc(4 * nblk * 4));
        hipLaunchKernelGGL(gb_popc_kernel, grid_for(nblk), dim3(256), 0, nullptr, blocks, nblk, d_pop.as<uint32_t>());
        uint64_t tot[4];
        for (int c = 0; c < 4; c++) {
            GBCHK(exclusive_scan_u32(d_pop.as<uint32_t>() + (size_t)c * nblk, d_ex.as<uint32_t>() + (size_t)c * nblk, nblk, tmp));
            uint32_t lp = 0, lf = 0;
            GBCHK(last_u32(d_ex.as<uint32_t>() + (size_t)c * nblk, nblk, lp)); GBCHK(last_u32(d_pop.as<uint32_t>() + (size_t)c * nblk, nblk, lf));
            tot[c] = (uint64_t)lp + lf;
        }
        for (int c = 0; c < 3; c++) C[c + 1] = C[c] + tot[c];
        if (C[3] + tot[3] != n) { err = "internal error: SBWT edge count does not match node count"; return -1; }
        hipLaunchKernelGGL(gb_bases_kernel, grid_for(nblk), dim3(256), 0, nullptr, blocks, nblk, d_ex.as<uint32_t>(), (uint32_t)C[0], (uint32_t)C[1], (uint32_t)C[2], (uint32_t)C[3]);
        GBCHK(hipGetLastError());
        GBCHK(hipDeviceSynchronize());
    }
    mark();
    // ---- 6. permute_unitigs, Ustart, ends, text ----
    GBCHK((sort_pairs<Key, uint32_t>(d_fk.as<Key>(), d_iota.as<uint32_t>(), nu, (unsigned)kb, tmp)));   // stable: ties keep the input order
    DevBuf d_lenp, d_ustart_ex, d_ends, d_concat;
    const uint64_t n_cwords = total / 16 + 8;
    GBCHK(d_lenp.alloc((uint64_t)nu * 4)); GBCHK(d_ustart_ex.alloc((uint64_t)nu * 4)); GBCHK(d_ends.alloc(((uint64_t)nu + 1 + 8) * 4)); GBCHK(d_concat.alloc(n_cwords * 4));
    hipLaunchKernelGGL(gb_ustart_kernel<Key>, grid_for(nu), dim3(256), 0, nullptr, g, d_fk.as<Key>(), d_iota.as<uint32_t>(), d_offs.as<uint64_t>(), nu, blocks, d_lenp.as<uint32_t>(), d_bad.as<uint32_t>());
    GBCHK(exclusive_scan_u32(d_lenp.as<uint32_t>(), d_ustart_ex.as<uint32_t>(), nu, tmp));
    hipLaunchKernelGGL(gb_ends_kernel, grid_for((uint64_t)nu + 8), dim3(256), 0, nullptr, d_ustart_ex.as<uint32_t>(), d_lenp.as<uint32_t>(), nu, d_ends.as<uint32_t>());
    hipLaunchKernelGGL(gb_concat_kernel, grid_for(n_cwords), dim3(256), 0, nullptr, d_codes.as<uint8_t>(), d_offs.as<uint64_t>(), d_iota.as<uint32_t>(), d_ends.as<uint32_t>(), nu, total,
                       d_concat.as<uint32_t>(), n_cwords);
    GBCHK(hipGetLastError());
    { uint32_t bad = 0; GBCHK(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost)); if (bad & 2u) { err = "internal error: first k-mer of a unitig missing from the SBWT"; return -1; } }
    out.k = (uint32_t)k; out.n_nodes = n; out.n_kmers = m; out.n_unitigs = nu; out.total_len = total;
    for (int c = 0; c < 4; c++) out.C[c] = C[c];
    out.lcs8.clear();
    if (k > FIN_FAST_K) { out.lcs8.assign(n, 0); GBCHK(hipMemcpy(out.lcs8.data(), d_lcs8.p, n, hipMemcpyDeviceToHost)); }
    out.ends.assign((size_t)nu + 1 + 8, 0);
    GBCHK(hipMemcpy(out.ends.data(), d_ends.p, out.ends.size() * 4, hipMemcpyDeviceToHost));
    fin_finish_sampling(out);
    DevBuf d_samp;
    GBCHK(d_samp.alloc(out.samp.size() * 4));
    GBCHK(hipMemcpy(d_samp.p, out.samp.data(), out.samp.size() * 4, hipMemcpyHostToDevice));
    mark();
    // ---- 7. finimizers ----
    DevBuf d_best;
    GBCHK(d_best.alloc((n + 64) * 8));
    GBCHK(hipMemset(d_best.p, 0, (n + 64) * 8));
    {
        FinDevIndex ix{};
        ix.blocks = blocks; ix.ends = d_ends.as<uint32_t>(); ix.samp = d_samp.as<uint32_t>(); ix.concat = d_concat.as<uint32_t>();
        ix.n_nodes = (uint32_t)n; ix.n_unitigs = nu; ix.total_len = (uint32_t)total; ix.k = (uint32_t)k; ix.samp_shift = out.samp_shift; ix.n_samp = (uint32_t)out.samp.size();
        for (int c = 0; c < 4; c++) ix.C[c] = (uint32_t)C[c];
        ix.C[4] = (uint32_t)n; ix.lcs8 = k > FIN_FAST_K ? d_lcs8.as<uint8_t>() : nullptr;
        const uint64_t n_seg = (total + GB_FSEG - 1) / GB_FSEG;
        DevBuf d_list, d_cnt, d_scratch;
        GBCHK(d_list.alloc((n_seg + 4) * 4)); GBCHK(d_cnt.alloc(16)); GBCHK(d_scratch.alloc(64ull * FIN_TPB * FGlobalDeque::CAP * 8));
        GBCHK(hipMemset(d_cnt.p, 0, 16));
        hipLaunchKernelGGL(gb_fmin_kernel, dim3((uint32_t)((n_seg + FIN_TPB - 1) / FIN_TPB)), dim3(FIN_TPB), 0, nullptr, ix, d_best.as<unsigned long long>(), (uint32_t)n_seg, d_list.as<uint32_t>(), d_cnt.as<uint32_t>());
        hipLaunchKernelGGL(gb_fmin_overflow_kernel, dim3(64), dim3(FIN_TPB), 0, nullptr, ix, d_best.as<unsigned long long>(), d_list.as<uint32_t>(), d_cnt.as<uint32_t>(), d_scratch.as<uint64_t>());
        GBCHK(hipGetLastError());
        GBCHK(hipDeviceSynchronize());
    }
    mark();
    // ---- 8. dictionaries, thermometer ----
    DevBuf d_info, d_goff;
    uint64_t nf = 0, nus = 0;
    {
        DevBuf d_cf, d_cu, d_ef, d_eu;
        GBCHK(d_info.alloc((nblk + 2) * sizeof(FinBlockInfo))); GBCHK(d_cf.alloc(nblk * 4)); GBCHK(d_cu.alloc(nblk * 4)); GBCHK(d_ef.alloc(nblk * 4)); GBCHK(d_eu.alloc(nblk * 4));
        hipLaunchKernelGGL(gb_masks_kernel, grid_for(nblk), dim3(256), 0, nullptr, blocks, d_best.as<unsigned long long>(), n, nblk, d_info.as<FinBlockInfo>(), d_cf.as<uint32_t>(), d_cu.as<uint32_t>());
        GBCHK(exclusive_scan_u32(d_cf.as<uint32_t>(), d_ef.as<uint32_t>(), nblk, tmp));
        GBCHK(exclusive_scan_u32(d_cu.as<uint32_t>(), d_eu.as<uint32_t>(), nblk, tmp));
        uint32_t a = 0, b = 0;
        GBCHK(last_u32(d_ef.as<uint32_t>(), nblk, a)); GBCHK(last_u32(d_cf.as<uint32_t>(), nblk, b)); nf = (uint64_t)a + b;
        GBCHK(last_u32(d_eu.as<uint32_t>(), nblk, a)); GBCHK(last_u32(d_cu.as<uint32_t>(), nblk, b)); nus = (uint64_t)a + b;
        GBCHK(d_goff.alloc((nf + 8) * 4));
        GBCHK(hipMemset(d_goff.p, 0, (nf + 8) * 4));
        hipLaunchKernelGGL(gb_ranks_kernel, grid_for(nblk + 2), dim3(256), 0, nullptr, d_best.as<unsigned long long>(), nblk, d_info.as<FinBlockInfo>(), d_ef.as<uint32_t>(), d_eu.as<uint32_t>(),
                           (uint32_t)nf, (uint32_t)nus, d_goff.as<uint32_t>());
        GBCHK(hipGetLastError());
    }
    (void)nus;
    {
        DevBuf d_hist;
        GBCHK(d_hist.alloc(128 * 8));
        GBCHK(hipMemset(d_hist.p, 0, 128 * 8));
        hipLaunchKernelGGL(gb_hist_kernel, dim3(1024), dim3(256), 0, nullptr, blocks, n, nblk, d_hist.as<unsigned long long>());
        unsigned long long hist[128];
        GBCHK(hipMemcpy(hist, d_hist.p, sizeof hist, hipMemcpyDeviceToHost));
        int t0 = 0; unsigned long long bestsum = 0;   // as fin_finish_thermometer
        for (int t = 0; t + 3 < 128; t++) { const unsigned long long sum = hist[t + 1] + hist[t + 2] + hist[t + 3]; if (sum > bestsum) { bestsum = sum; t0 = t; } }
        if (const char* e = getenv("FINITO_LCS_T0")) { const int v = atoi(e); if (v >= 0 && v < 124) t0 = v; }
        out.lcs_t0 = (uint32_t)t0;
        hipLaunchKernelGGL(gb_thermo_kernel, grid_for(nblk), dim3(256), 0, nullptr, blocks, nblk, t0);
        GBCHK(hipGetLastError());
    }
    mark();
    // ---- copy back ----
    if (!out.blocks.resize(nblk)) { err = "out of memory (blocks)"; return -4; }
    GBCHK(hipMemcpy(out.blocks.p, d_blocks.p, nblk * sizeof(FinNodeBlock), hipMemcpyDeviceToHost));
    out.blkinfo.assign(nblk + 2, FinBlockInfo{0, 0, 0, 0, 0, 0});
    GBCHK(hipMemcpy(out.blkinfo.data(), d_info.p, (nblk + 2) * sizeof(FinBlockInfo), hipMemcpyDeviceToHost));
    out.goff.assign(nf + 8, 0);
    GBCHK(hipMemcpy(out.goff.data(), d_goff.p, (nf + 8) * 4, hipMemcpyDeviceToHost));
    out.n_fmin = nf;
    out.concat.assign(n_cwords, 0);
    GBCHK(hipMemcpy(out.concat.data(), d_concat.p, n_cwords * 4, hipMemcpyDeviceToHost));
    mark();
    GBCHK(hipDeviceSynchronize());
    if (phase_ms) for (int i = 0; i + 1 < evn && i < 8; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); phase_ms[i] = ms; }
    return 0;
}

int fin_build_index_gpu(const char* bases, const uint64_t* offsets, uint64_t n_unitigs, int k, int device, fin_index& out, std::string& err, double* phase_ms) {
    if (k < 2 || k > 255) { err = "the device builder handles k in [2, 255]"; return -5; }
    if (k <= 32) return build_index_gpu_impl<uint64_t>(bases, offsets, n_unitigs, k, device, out, err, phase_ms);
    if (k <= 64) return build_index_gpu_impl<gb_key128>(bases, offsets, n_unitigs, k, device, out, err, phase_ms);
    if (k <= 128) return build_index_gpu_impl<GbWide<4>>(bases, offsets, n_unitigs, k, device, out, err, phase_ms);
    return build_index_gpu_impl<GbWide<8>>(bases, offsets, n_unitigs, k, device, out, err, phase_ms);
}
