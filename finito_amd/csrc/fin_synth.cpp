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
void fin_synth_reads_at(const char* genome, uint64_t n, uint64_t first_record, uint64_t n_reads, uint32_t read_len, double err_rate,
                        double random_frac, uint64_t seed, char* out_bases, uint64_t* out_offsets, int64_t* read_gstart,
                        uint8_t* read_rc, uint8_t* err_mask);
void fin_synth_reads(const char* genome, uint64_t n, uint64_t n_reads, uint32_t read_len, double err_rate,
                     double random_frac, uint64_t seed, char* out_bases, uint64_t* out_offsets, int64_t* read_gstart,
                     uint8_t* read_rc, uint8_t* err_mask) {
    fin_synth_reads_at(genome, n, 0, n_reads, read_len, err_rate, random_frac, seed, out_bases, out_offsets, read_gstart, read_rc, err_mask);
}
// records [first_record, first_record + n_reads) of the read set `seed` names: a record's content depends on its number alone, so a set
// can be made in pieces -- by rank, by batch -- and is the same set however it is cut (bench.py's strong-scaling mode)
void fin_synth_reads_at(const char* genome, uint64_t n, uint64_t first_record, uint64_t n_reads, uint32_t read_len, double err_rate,
                        double random_frac, uint64_t seed, char* out_bases, uint64_t* out_offsets, int64_t* read_gstart,
                        uint8_t* read_rc, uint8_t* err_mask) {
    for (uint64_t r = 0; r <= n_reads; r++) out_offsets[r] = r * (uint64_t)read_len;
    const uint64_t thr = (uint64_t)(err_rate * 18446744073709551615.0);
#pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < n_reads; r++) {
        Rng g(seed ^ (0x9e3779b97f4a7c15ull * (first_record + r + 1)));
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

// ---- repeat-rich inputs (round 3) ------------------------------------------------------------------------------------------
// The iid genome above is the easiest input the path can meet: every 19-mer is unique, a read lies in one long unitig.  Real genomes
// are 40-50 % repeats: interspersed families whose copies diverged by 1-10 %, tandem arrays, a few large recent duplications.  Their
// k-mers are NOT all distinct, and a unitig set -- which holds every k-mer once, whichever strand -- is short and branchy there.
//   fin_synth_repeat_genome : an iid background with such repeats written over it (seeded)
//   fin_synth_spss          : a DISJOINT spectrum-preserving string set of a genome: every canonical k-mer (a k-mer and its reverse
//                             complement are one) kept at its first occurrence only -- the pieces break wherever a k-mer was seen
//                             before, as the unitigs of a de Bruijn graph do -- cut to max_len, flipped, shuffled like
//                             fin_synth_unitigs.  Also returns, for every k-mer position that is not a first occurrence, where the
//                             first one is, so that the ground truth holds for repeats too (fin_synth_check2).
#include <parallel/algorithm>

namespace {
template <typename K> struct KPt { K key; uint32_t pos; };
// canonical 2-bit keys of every k-mer start of g[0, n) (k <= 32 in 64-bit keys, k <= 64 in 128-bit ones), sorted by (key, pos)
template <typename K>
void sorted_canonical_kmers(const char* g, uint64_t n, int k, std::vector<KPt<K>>& v) {
    typedef KPt<K> KP;
    const uint64_t nk = n >= (uint64_t)k ? n - (uint64_t)k + 1 : 0;
    v.resize((size_t)nk);
    const K mask = 2 * k == (int)sizeof(K) * 8 ? ~(K)0 : (((K)1 << (2 * k)) - (K)1);
    const uint64_t CH = 1 << 20, nch = (nk + CH - 1) / CH;
#pragma omp parallel for schedule(static)
    for (uint64_t c = 0; c < nch; c++) {
        const uint64_t lo = c * CH, hi = std::min(nk, lo + CH);
        K f = 0, r = 0;
        for (int j = 0; j < k - 1; j++) { const K x = (K)code(g[lo + (uint64_t)j]); f = (f << 2) | x; r = (r >> 2) | ((3 - x) << (2 * (k - 1))); }
        for (uint64_t p = lo; p < hi; p++) {
            const K x = (K)code(g[p + (uint64_t)(k - 1)]);
            f = ((f << 2) | x) & mask; r = (r >> 2) | ((3 - x) << (2 * (k - 1)));
            v[(size_t)p] = KP{f < r ? f : r, (uint32_t)p};
        }
    }
    __gnu_parallel::sort(v.begin(), v.end(), [](const KP& a, const KP& b) { return a.key != b.key ? a.key < b.key : a.pos < b.pos; });
}
}  // namespace

extern "C" {

// n bases; about `repeat_frac` of them inside repeats.  Families: short interspersed (300 bp, many copies), long interspersed (up to
// 6 kb, copies truncated at their 5' end), a handful of others; every copy in a random orientation with its own divergence drawn from
// [div_lo, div_hi] (substitutions); tandem arrays (unit 2..50, 2 % per-unit mutation); a few segmental duplications (20..100 kb, 1-2 %).
void fin_synth_repeat_genome(uint64_t n, uint64_t seed, double repeat_frac, double div_lo, double div_hi, char* out) {
    fin_synth_genome(n, seed, out);
    if (n < 20000 || repeat_frac <= 0) return;
    Rng r(seed ^ 0x5eedbeefcafef00dull);
    auto rnd_seq = [&](std::vector<char>& s, uint64_t len) { s.resize((size_t)len); for (auto& c : s) c = ACGT[r.next() & 3]; };
    auto mutate = [&](std::vector<char>& s, double d) { for (auto& c : s) if (r.unit() < d) c = ACGT[(code(c) + 1 + (int)r.below(3)) & 3]; };
    auto place = [&](const std::vector<char>& s, uint64_t at, bool flip) {
        const uint64_t L = s.size();
        if (at + L > n) return;
        if (!flip) memcpy(out + at, s.data(), (size_t)L);
        else for (uint64_t j = 0; j < L; j++) out[at + j] = comp(s[(size_t)(L - 1 - j)]);
    };
    struct Fam { std::vector<char> cons; double share; bool truncate; };
    std::vector<Fam> fams;
    const double shares[] = {0.25, 0.40, 0.05, 0.05, 0.04, 0.04, 0.03, 0.03};   // of the interspersed part (0.89 of all repeats)
    const uint64_t lens[] = {300, 6000, 200, 450, 800, 1200, 2000, 3500};
    for (int f = 0; f < 8; f++) {
        // a family = a few subfamily consensuses 3-6 % apart
        std::vector<char> master; rnd_seq(master, std::min<uint64_t>(lens[f], n / 50));
        const int nsub = f < 2 ? 6 : 2;
        for (int s = 0; s < nsub; s++) { Fam fm; fm.cons = master; mutate(fm.cons, 0.03 + 0.03 * r.unit()); fm.share = shares[f] / nsub; fm.truncate = f == 1; fams.push_back(fm); }
    }
    const double inter = 0.89 * repeat_frac * (double)n;
    for (const Fam& fm : fams) {
        uint64_t budget = (uint64_t)(fm.share * inter), used = 0;
        while (used < budget) {
            std::vector<char> c = fm.cons;
            if (fm.truncate) { const uint64_t keep = 300 + r.below(c.size() - 300 + 1); c.erase(c.begin(), c.begin() + (long)(c.size() - keep)); }
            mutate(c, div_lo + (div_hi - div_lo) * r.unit());
            place(c, r.below(n - c.size()), (r.next() & 1) != 0);
            used += c.size();
        }
    }
    {   // tandem arrays: 0.04 of the repeats
        uint64_t budget = (uint64_t)(0.04 * repeat_frac * (double)n), used = 0;
        while (used < budget) {
            std::vector<char> unit; rnd_seq(unit, 2 + r.below(49));
            const uint64_t total = 60 + r.below(1500);
            std::vector<char> arr;
            while (arr.size() < total) { std::vector<char> u2 = unit; mutate(u2, 0.02); arr.insert(arr.end(), u2.begin(), u2.end()); }
            place(arr, r.below(n - arr.size()), false);
            used += arr.size();
        }
    }
    {   // segmental duplications: 0.07 of the repeats, copied from the genome as it is now (repeats inside them included)
        uint64_t budget = (uint64_t)(0.07 * repeat_frac * (double)n), used = 0;
        while (used < budget) {
            const uint64_t L = std::min<uint64_t>(20000 + r.below(80001), n / 20);
            const uint64_t src = r.below(n - L), dst = r.below(n - L);
            if (src + L > dst && dst + L > src) continue;
            std::vector<char> c(out + src, out + src + L);
            mutate(c, 0.01 + 0.01 * r.unit());
            place(c, dst, (r.next() & 1) != 0);
            used += L;
        }
    }
}

// Pieces as fin_synth_unitigs returns them (shuffled order; start, length and flip of each).  dup_pos / dup_first (capacity cap_dups):
// the k-mer starts that are not the first occurrence of their canonical k-mer, ascending, each with that first occurrence;
// *n_dups = their number.  multi (n bytes, may be null): 1 for every k-mer start whose canonical k-mer occurs more than once (first
// occurrences included).  Returns the number of pieces, or -(needed) if a capacity is too small (then *n_dups says how many dups).
}  // extern "C"
template <typename K>
static int64_t spss_impl(const char* genome, uint64_t n, int k, uint32_t max_len, uint64_t seed, char* out_bases, uint64_t out_cap,
                         uint64_t* out_offsets, uint64_t* piece_gstart, uint32_t* piece_glen, uint8_t* piece_rc, int64_t cap_pieces,
                         uint32_t* dup_pos, uint32_t* dup_first, uint64_t cap_dups, uint64_t* n_dups, uint8_t* multi) {
    typedef KPt<K> KP;
    std::vector<KP> v;
    sorted_canonical_kmers<K>(genome, n, k, v);
    const uint64_t nk = v.size();
    std::vector<uint8_t> dup((size_t)nk, 0);
    std::vector<uint32_t> first;   // for dup positions only, filled below in position order
    uint64_t nd = 0;
    if (multi) memset(multi, 0, (size_t)n);
    // groups of equal keys: positions ascending inside a group
    std::vector<uint32_t> first_of((size_t)0);
    {
        // pass 1: mark, count
#pragma omp parallel for schedule(static) reduction(+ : nd)
        for (uint64_t i = 1; i < nk; i++)
            if (v[(size_t)i].key == v[(size_t)i - 1].key) { dup[v[(size_t)i].pos] = 1; nd++; }
        if (multi) {
#pragma omp parallel for schedule(static)
            for (uint64_t i = 0; i < nk; i++) {
                const bool m = (i > 0 && v[(size_t)i].key == v[(size_t)i - 1].key) || (i + 1 < nk && v[(size_t)i].key == v[(size_t)i + 1].key);
                if (m) multi[v[(size_t)i].pos] = 1;
            }
        }
    }
    *n_dups = nd;
    if (nd > cap_dups) return -1;
    {   // pass 2: first occurrence of every dup, written at the dup's rank among dup positions
        std::vector<uint64_t> rank_at;   // prefix count of dup[] per 64k block
        const uint64_t B = 1 << 16, nb = (nk + B - 1) / B;
        rank_at.assign((size_t)nb + 1, 0);
#pragma omp parallel for schedule(static)
        for (uint64_t b = 0; b < nb; b++) { uint64_t c = 0; for (uint64_t p = b * B; p < std::min(nk, (b + 1) * B); p++) c += dup[(size_t)p]; rank_at[(size_t)b + 1] = c; }
        for (uint64_t b = 0; b < nb; b++) rank_at[(size_t)b + 1] += rank_at[(size_t)b];
        std::vector<uint32_t> rank_of((size_t)nk);   // rank of position p among dups (only meaningful where dup[p])
#pragma omp parallel for schedule(static)
        for (uint64_t b = 0; b < nb; b++) { uint64_t c = rank_at[(size_t)b]; for (uint64_t p = b * B; p < std::min(nk, (b + 1) * B); p++) { rank_of[(size_t)p] = (uint32_t)c; c += dup[(size_t)p]; } }
#pragma omp parallel for schedule(static)
        for (uint64_t p = 0; p < nk; p++) if (dup[(size_t)p]) dup_pos[rank_of[(size_t)p]] = (uint32_t)p;
        // group heads: one sequential walk over the sorted array (a group of a tandem array's k-mer can hold 10^5 positions)
        uint64_t h = 0;
        for (uint64_t i = 1; i < nk; i++) {
            if (v[(size_t)i].key != v[(size_t)i - 1].key) h = i;
            else dup_first[rank_of[v[(size_t)i].pos]] = v[(size_t)h].pos;
        }
    }
    std::vector<KP>().swap(v);
    // runs of first-occurrence k-mer starts -> pieces of at most max_len bases overlapping by k-1 inside a run
    Rng r(seed);
    std::vector<uint64_t> gs; std::vector<uint32_t> gl;
    uint64_t p = 0;
    while (p < nk) {
        if (dup[(size_t)p]) { p++; continue; }
        uint64_t e = p;
        while (e + 1 < nk && !dup[(size_t)e + 1]) e++;
        // k-mer starts p..e -> bases [p, e + k)
        uint64_t s = p; const uint64_t end = e + (uint64_t)k;
        while (true) {
            const uint64_t L = (uint64_t)k + r.below((uint64_t)max_len - (uint64_t)k + 1);
            const uint64_t pe = std::min(end, s + L);
            gs.push_back(s); gl.push_back((uint32_t)(pe - s));
            if (pe == end) break;
            s = pe - (uint64_t)(k - 1);
        }
        p = e + 1;
    }
    const int64_t np = (int64_t)gs.size();
    uint64_t tot = 0;
    for (int64_t i = 0; i < np; i++) tot += gl[(size_t)i];
    if (np > cap_pieces || tot > out_cap) { out_offsets[0] = tot; return -np; }
    std::vector<uint32_t> order((size_t)np);
    for (int64_t i = 0; i < np; i++) order[(size_t)i] = (uint32_t)i;
    for (int64_t i = np - 1; i > 0; i--) { uint64_t j = r.below((uint64_t)i + 1); std::swap(order[(size_t)i], order[(size_t)j]); }
    tot = 0;
    out_offsets[0] = 0;
    for (int64_t i = 0; i < np; i++) {
        const uint32_t q = order[(size_t)i];
        piece_gstart[i] = gs[q]; piece_glen[i] = gl[q]; piece_rc[i] = (uint8_t)(r.next() & 1);
        tot += gl[q];
        out_offsets[i + 1] = tot;
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < np; i++) {
        const char* src = genome + piece_gstart[i];
        char* dst = out_bases + out_offsets[i];
        const uint32_t L = piece_glen[i];
        if (!piece_rc[i]) memcpy(dst, src, L);
        else for (uint32_t j = 0; j < L; j++) dst[j] = comp(src[L - 1 - j]);
    }
    return np;
}
extern "C" {
int64_t fin_synth_spss(const char* genome, uint64_t n, int k, uint32_t max_len, uint64_t seed, char* out_bases, uint64_t out_cap,
                       uint64_t* out_offsets, uint64_t* piece_gstart, uint32_t* piece_glen, uint8_t* piece_rc, int64_t cap_pieces,
                       uint32_t* dup_pos, uint32_t* dup_first, uint64_t cap_dups, uint64_t* n_dups, uint8_t* multi) {
    if (k > 64 || n < (uint64_t)k || n >= 0xFFFFFFFFull) return 0;
    if (k <= 32) return spss_impl<uint64_t>(genome, n, k, max_len, seed, out_bases, out_cap, out_offsets, piece_gstart, piece_glen, piece_rc, cap_pieces, dup_pos, dup_first, cap_dups, n_dups, multi);
    return spss_impl<unsigned __int128>(genome, n, k, max_len, seed, out_bases, out_cap, out_offsets, piece_gstart, piece_glen, piece_rc, cap_pieces, dup_pos, dup_first, cap_dups, n_dups, multi);
}

// fin_synth_check for repeat-rich inputs.  dup_pos/dup_first (n_dups entries, ascending; may be empty): a k-mer that starts at a listed
// position is expected at the place of its first occurrence (fin_synth_spss).  skip (n bytes, may be null): k-mer starts that are not
// checked (a set that is not disjoint: k-mers with several places).
int64_t fin_synth_check2(uint64_t n_pieces, const uint64_t* piece_gstart, const uint32_t* piece_glen, const uint8_t* piece_rc,
                         const uint32_t* unitig_id, int k, uint64_t n_reads, uint32_t read_len, const int64_t* read_gstart,
                         const uint8_t* read_rc, const uint8_t* err_mask, const int32_t* pairs, const uint32_t* dup_pos,
                         const uint32_t* dup_first, uint64_t n_dups, const uint8_t* skip, uint64_t* n_checked, int64_t* first_bad_read) {
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
                if (!(skip && skip[p])) {
                    if (n_dups) {
                        const uint32_t* it = std::lower_bound(dup_pos, dup_pos + n_dups, (uint32_t)p);
                        if (it != dup_pos + n_dups && *it == (uint32_t)p) p = dup_first[it - dup_pos];
                    }
                    size_t idx = (size_t)(std::upper_bound(starts.begin(), starts.end(), p) - starts.begin()) - 1;
                    // (pieces of different runs may overlap by fewer than k-1 bases: take the one that holds the whole k-mer)
                    while (idx > 0 && p + (uint64_t)k > starts[idx] + piece_glen[by_start[idx]]) idx--;
                    const uint32_t pi = by_start[idx];
                    uint64_t off = piece_rc[pi] ? (piece_gstart[pi] + piece_glen[pi]) - (p + (uint64_t)k) : p - piece_gstart[pi];
                    checked++;
                    if (p + (uint64_t)k > piece_gstart[pi] + piece_glen[pi] || pr[2 * j] != (int32_t)unitig_id[pi] || pr[2 * j + 1] != (int32_t)off) {
                        bad++;
#pragma omp critical
                        { if (first < 0 || (int64_t)r < first) first = (int64_t)r; }
                    }
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
