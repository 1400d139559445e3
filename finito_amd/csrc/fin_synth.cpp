// fin_synth.cpp -- seeded synthetic inputs for the benchmark configurations and a ground-truth
// checker that works at full size.  Tooling around the path, not part of it.
//
//   genome : iid uniform ACGT (splitmix64-seeded xoshiro256**)
//   unitigs: the genome cut into pieces of uniform length [k, max_len] overlapping by k-1 (every genome k-mer in
//            exactly one piece), each piece reverse-complemented with p = 1/2, order shuffled
//   reads  : start uniform, fixed length, strand p = 1/2, iid substitutions, a fraction of fully random reads
//   check  : every error-free k-mer of a genome-derived read must localize to the piece that holds it
#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (int i = 0; i < 4; i++) s[i] = splitmix(seed); }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    inline uint64_t next() {
        const uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return result;
    }
    inline uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
    inline double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
const char ACGT[5] = "ACGT";
inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
inline int code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }
}  // namespace

extern "C" {

void fin_synth_genome(uint64_t n, uint64_t seed, char* out) {
    const uint64_t CH = 1 << 20;
    const uint64_t nch = (n + CH - 1) / CH;
#pragma omp parallel for schedule(static)
    for (uint64_t c = 0; c < nch; c++) {
        Rng r(seed * 0x100000001b3ull + c);
        uint64_t lo = c * CH, hi = std::min(n, lo + CH);
        uint64_t i = lo;
        for (; i + 32 <= hi; i += 32) { uint64_t x = r.next(); for (int j = 0; j < 32; j++) out[i + j] = ACGT[(x >> (2 * j)) & 3]; }
        for (; i < hi; i++) out[i] = ACGT[r.next() & 3];
    }
}

// Returns the number of pieces, or -(needed) if a capacity is too small.  Pieces are emitted in shuffled order;
// piece_gstart/piece_glen/piece_rc describe them in that order.  out_bases needs n + pieces*(k-1) bytes.
int64_t fin_synth_unitigs(const char* genome, uint64_t n, int k, uint32_t max_len, uint64_t seed, char* out_bases,
                          uint64_t out_cap, uint64_t* out_offsets, uint64_t* piece_gstart, uint32_t* piece_glen,
                          uint8_t* piece_rc, int64_t cap_pieces) {
    Rng r(seed);
    std::vector<uint64_t> gs; std::vector<uint32_t> gl;
    uint64_t s = 0;
    if (n < (uint64_t)k) return 0;
    while (true) {
        uint64_t L = (uint64_t)k + r.below((uint64_t)max_len - (uint64_t)k + 1);
        uint64_t e = std::min(n, s + L);
        gs.push_back(s); gl.push_back((uint32_t)(e - s));
        if (e == n) break;
        s = e - (uint64_t)(k - 1);
    }
    const int64_t np = (int64_t)gs.size();
    if (np > cap_pieces) return -np;
    std::vector<uint32_t> order((size_t)np);
    for (int64_t i = 0; i < np; i++) order[(size_t)i] = (uint32_t)i;
    for (int64_t i = np - 1; i > 0; i--) { uint64_t j = r.below((uint64_t)i + 1); std::swap(order[(size_t)i], order[(size_t)j]); }
    uint64_t tot = 0;
    out_offsets[0] = 0;
    for (int64_t i = 0; i < np; i++) {
        uint32_t p = order[(size_t)i];
        piece_gstart[i] = gs[p]; piece_glen[i] = gl[p]; piece_rc[i] = (uint8_t)(r.next() & 1);
        tot += gl[p];
        out_offsets[i + 1] = tot;
    }
    if (tot > out_cap) return -np;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < np; i++) {
        const char* src = genome + piece_gstart[i];
        char* dst = out_bases + out_offsets[i];
        uint32_t L = piece_glen[i];
        if (!piece_rc[i]) memcpy(dst, src, L);
        else for (uint32_t j = 0; j < L; j++) dst[j] = comp(src[L - 1 - j]);
    }
    return np;
}

// read_gstart[r] = genome start or -1 for a fully random read; read_rc[r] = 1 if the read is the reverse complement
// of the genome window; err_mask (optional, one byte per base) marks substituted bases in read coordinates.
void fin_synth_reads(const char* genome, uint64_t n, uint64_t n_reads, uint32_t read_len, double err_rate,
                     double random_frac, uint64_t seed, char* out_bases, uint64_t* out_offsets, int64_t* read_gstart,
                     uint8_t* read_rc, uint8_t* err_mask) {
    for (uint64_t r = 0; r <= n_reads; r++) out_offsets[r] = r * (uint64_t)read_len;
    const uint64_t thr = (uint64_t)(err_rate * 18446744073709551615.0);
#pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < n_reads; r++) {
        Rng g(seed ^ (0x9e3779b97f4a7c15ull * (r + 1)));
        char* dst = out_bases + r * (uint64_t)read_len;
        uint8_t* em = err_mask ? err_mask + r * (uint64_t)read_len : nullptr;
        if (g.unit() < random_frac || n < read_len) {
            for (uint32_t j = 0; j < read_len; j++) dst[j] = ACGT[g.next() & 3];
            if (em) memset(em, 1, read_len);
            read_gstart[r] = -1; read_rc[r] = 0;
            continue;
        }
        uint64_t a = g.below(n - read_len + 1);
        uint8_t strand = (uint8_t)(g.next() & 1);
        read_gstart[r] = (int64_t)a; read_rc[r] = strand;
        for (uint32_t j = 0; j < read_len; j++) {
            char c = strand ? comp(genome[a + read_len - 1 - j]) : genome[a + j];
            uint8_t e = 0;
            if (g.next() < thr) { c = ACGT[(code(c) + 1 + (int)g.below(3)) & 3]; e = 1; }
            dst[j] = c;
            if (em) em[j] = e;
        }
    }
}

// Ground truth at any size.  pieces are given in the order the unitigs were handed to the builder; unitig_id[i] is
// the id the index assigned to piece i (colex order of first k-mers).  pairs: merged int32 output, reads back to
// back.  Returns the number of error-free k-mers whose result differs from the piece that holds them, and stores
// how many were checked.
int64_t fin_synth_check(uint64_t n_pieces, const uint64_t* piece_gstart, const uint32_t* piece_glen, const uint8_t* piece_rc,
                        const uint32_t* unitig_id, int k, uint64_t n_reads, uint32_t read_len, const int64_t* read_gstart,
                        const uint8_t* read_rc, const uint8_t* err_mask, const int32_t* pairs, uint64_t* n_checked,
                        int64_t* first_bad_read) {
    std::vector<uint32_t> by_start((size_t)n_pieces);
    for (uint64_t i = 0; i < n_pieces; i++) by_start[(size_t)i] = (uint32_t)i;
    std::sort(by_start.begin(), by_start.end(), [&](uint32_t a, uint32_t b) { return piece_gstart[a] < piece_gstart[b]; });
    std::vector<uint64_t> starts((size_t)n_pieces);
    for (uint64_t i = 0; i < n_pieces; i++) starts[(size_t)i] = piece_gstart[by_start[(size_t)i]];
    const int64_t nk = (int64_t)read_len - k + 1;
    int64_t bad = 0; uint64_t checked = 0; int64_t first = -1;
    if (nk <= 0) { if (n_checked) *n_checked = 0; return 0; }
#pragma omp parallel for schedule(static) reduction(+ : bad, checked)
    for (uint64_t r = 0; r < n_reads; r++) {
        if (read_gstart[r] < 0) continue;
        const uint8_t* em = err_mask + r * (uint64_t)read_len;
        const int32_t* pr = pairs + 2 * r * (uint64_t)nk;
        int errs_in_window = 0;
        for (int j = 0; j < k - 1; j++) errs_in_window += em[j];
        for (int64_t j = 0; j < nk; j++) {
            errs_in_window += em[j + k - 1];
            if (errs_in_window == 0) {
                uint64_t p = read_rc[r] ? (uint64_t)read_gstart[r] + (uint64_t)(nk - 1 - j) : (uint64_t)read_gstart[r] + (uint64_t)j;
                // the piece holding the whole k-mer [p, p+k): last piece with start <= p whose end covers p+k
                size_t idx = (size_t)(std::upper_bound(starts.begin(), starts.end(), p) - starts.begin()) - 1;
                const uint32_t pi = by_start[idx];   // s_{i+1} = e_i - (k-1), so [p, p+k) fits piece i iff p < s_{i+1}
                uint64_t off = piece_rc[pi] ? (piece_gstart[pi] + piece_glen[pi]) - (p + (uint64_t)k) : p - piece_gstart[pi];
                checked++;
                if (pr[2 * j] != (int32_t)unitig_id[pi] || pr[2 * j + 1] != (int32_t)off) {
                    bad++;
#pragma omp critical
                    { if (first < 0 || (int64_t)r < first) first = (int64_t)r; }
                }
            }
            errs_in_window -= em[j];
        }
    }
    if (n_checked) *n_checked = checked;
    if (first_bad_read) *first_bad_read = first;
    return bad;
}

}  // extern "C"
