// fin_stats.cpp -- the statistics-only modes of `build-fmin` (--type shortest | verify, frequency threshold t):
// build_shortest_streaming_search (build_fmin.hh:134-200), verify_shortest_streaming_search (:95-132), remove_ns (:216-242),
// print_finimizer_stats (common.hh:188-206).  Host code over the block layout, like the builder: these modes only count
// finimizers, they have no query path.
#include <algorithm>
#include <array>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/finito_amd.h"
#include "fin_index.hpp"

namespace {
struct Tup { int64_t len, f, colex, end; };   // std::tuple order of the reference: {length, frequency, interval start, end}
inline bool gt(const Tup& a, const Tup& b) {
    if (a.len != b.len) return a.len > b.len;
    if (a.f != b.f) return a.f > b.f;
    if (a.colex != b.colex) return a.colex > b.colex;
    return a.end > b.end;
}
inline int code(char ch) {
    switch (ch & ~32) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}
using Triple = std::array<int64_t, 3>;

// one sequence of --type shortest; the ring buffer reproduces BoundedDeque (BoundedDeque.hh:5-75), reads of stale slots included
bool shortest_streaming(const fin_index& x, const char* in, int64_t n_in, int64_t t, std::vector<uint8_t>& found, std::vector<Triple>& out) {
    const FinNodeBlock* B = x.blocks.p;
    const uint8_t* const lcs8 = x.lcs8_or_null();
    const int64_t n = (int64_t)x.n_nodes, k = x.k;
    const int64_t dsize = n_in > 0 ? n_in : 1;
    std::vector<Tup> buf((size_t)dsize, Tup{0, 0, 0, 0});
    int64_t front = dsize - 1, back = 0, n_el = 0;
    auto inc = [&](int64_t i) { return (int64_t)(((uint64_t)(i + 1)) % (uint64_t)dsize); };
    auto dec = [&](int64_t i) { return (int64_t)(((uint64_t)(i - 1 + dsize)) % (uint64_t)dsize); };
    Tup w{k + 2, n, n, n_in}, cur{0, 0, 0, 0};
    int64_t kmer = 0, start = 0;
    FinIval I{0, n - 1};
    for (int64_t end = 0; end < n_in; end++) {
        const int c = code(in[end]);
        if (c < 0) return false;
        I = fin_host_extend(B, c, I);
        if (I.first < 0) return false;   // (the reference would spin in its while loop here)
        int64_t freq = I.second - I.first + 1, ist = I.first;
        if (freq <= t) {
            while (freq <= t) {
                cur = Tup{end - start + 1, freq, ist, end};
                start++;
                I = fin_host_drop(B, lcs8, n, end - start + 1, I);
                freq = I.second - I.first + 1; ist = I.first;
            }
            if (gt(w, cur)) { n_el = 0; front = dsize - 1; back = 0; w = cur; }
            else while (gt(buf[(size_t)dec(back)], cur)) { back = dec(back); n_el--; }
            buf[(size_t)back] = cur; back = inc(back); n_el++;
        }
        if (end >= k - 1) {
            if (!found[(size_t)w.colex]) {
                out.push_back(Triple{w.len, w.f, w.colex});
                if (w.end >= k - 1) found[(size_t)w.colex] = 1;
            }
            kmer++;
            while (w.end - w.len + 1 < kmer) {
                front = inc(front); n_el--;
                w = n_el == 0 ? Tup{k + 1, n, n, kmer + k} : buf[(size_t)inc(front)];
            }
        }
    }
    return true;
}

// --type verify on one ACGT stretch: every substring of every k-window from the full interval
bool verify_windows(const fin_index& x, const char* in, int64_t n_in, int64_t t, std::vector<Triple>& out) {
    const FinNodeBlock* B = x.blocks.p;
    const int64_t n = (int64_t)x.n_nodes, k = x.k;
    for (int64_t i = 0; i + k <= n_in; i++) {
        Tup w{k + 1, n, n, n_in};
        for (int64_t start = i; start < k + i; start++) {
            FinIval I{0, n - 1};
            for (int64_t end = start; end < k + i; end++) {
                const int c = code(in[end]);
                if (c < 0) return false;
                I = fin_host_extend(B, c, I);
                const int64_t freq = I.second - I.first + 1;   // (-1,-1) counts as frequency 1, as in the reference
                if (freq <= t) { const Tup nf{end - start + 1, freq, I.first, end}; if (gt(w, nf)) w = nf; }
            }
        }
        out.push_back(Triple{w.len, w.f, w.colex});
    }
    return true;
}
}  // namespace

extern "C" int fin_index_finimizer_stats(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_seqs, int type, int64_t t,
                                         int64_t* n_finimizers, int64_t* sum_freq, int64_t* sum_len, char* err, size_t errlen) {
    auto fail = [&](int rc, const char* msg) { if (err && errlen) snprintf(err, errlen, "%s", msg); return rc; };
    if (!idx || !offsets || (n_seqs && !bases) || (type != FIN_STATS_SHORTEST && type != FIN_STATS_VERIFY) || t < 1) return fail(FIN_EINVAL, "bad argument");
    std::vector<Triple> v;
    std::vector<uint8_t> found((size_t)idx->n_nodes + 2, 0);   // (+1: the reference indexes fmin_found[n_nodes] when a window has no candidate)
    for (uint64_t s = 0; s < n_seqs; s++) {
        const char* seq = bases + offsets[s];
        const int64_t len = (int64_t)(offsets[s + 1] - offsets[s]);
        if (type == FIN_STATS_SHORTEST) {
            if (!shortest_streaming(*idx, seq, len, t, found, v)) return fail(FIN_EINVAL, "a sequence leaves the index (these modes expect the indexed unitigs)");
        } else {
            int64_t st = 0;   // remove_ns: maximal ACGT stretches of at least k bases
            for (int64_t i = 0; i <= len; i++) {
                if (i == len || code(seq[i]) < 0) {
                    if (i - st >= (int64_t)idx->k && !verify_windows(*idx, seq + st, i - st, t, v)) return fail(FIN_EINVAL, "internal error");
                    st = i + 1;
                }
            }
        }
    }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    int64_t sf = 0, sl = 0;
    for (const Triple& tr : v) { sl += tr[0]; sf += tr[1]; }
    if (n_finimizers) *n_finimizers = (int64_t)v.size();
    if (sum_freq) *sum_freq = sf;
    if (sum_len) *sum_len = sl;
    return FIN_OK;
}
