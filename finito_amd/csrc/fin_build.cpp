// fin_build.cpp -- index construction of the product (host, C++/OpenMP).
//
// Replaces, for type "rarest" / t = 1, the reference's build chain:
//   external `sbwt build` (README.md:33-35; semantics SURVEY.md 8a-5, node order pinned by tests.cpp:110-123),
//   lcs_basic_parallel_algorithm (lcs_basic_parallel_algorithm.hpp:52-120),
//   permute_unitigs (PackedStrings.hh:105-135) and
//   FinimizerIndexBuilder (FinimizerIndex.hh:273-389),
// and writes the result straight into the 128-byte node-block layout of fin_format.h.
//
// Not a translation: the reference propagates labels for k rounds to get the LCS and walks the unitigs
// sequentially; here the colex-sorted k-mer set itself gives node order, LCS (xor + clz of adjacent keys) and
// edges (each node's marked in-edge comes from the first node of its (k-1)-prefix group), and the finimizer
// pass runs over unitigs in parallel with the reference's overwrite rule (FinimizerIndex.hh:370-378) expressed
// as an atomic max.  Equality with the literal restatement (oracle/) is tested in tests/test_builder_parity.py.
#include <omp.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "fin_index.hpp"

typedef unsigned __int128 u128;

// k-mer keys wider than 128 bits (k > 64): N little-endian 64-bit words with the integer operations the builder uses.  The colex
// order of k-mers is the integer order of their keys (base j at bits 2j), whatever the width.
template <int N>
struct BigKey {
    uint64_t w[N];
    BigKey() {}
    BigKey(uint64_t x) { w[0] = x; for (int i = 1; i < N; i++) w[i] = 0; }
    BigKey(int x) : BigKey((uint64_t)(int64_t)x) { if (x < 0) for (int i = 1; i < N; i++) w[i] = ~0ull; }
    explicit operator uint64_t() const { return w[0]; }
    explicit operator int() const { return (int)w[0]; }
    BigKey operator<<(int s) const {
        BigKey r; const int ws = s >> 6, bs = s & 63;
        for (int i = N - 1; i >= 0; i--) {
            uint64_t v = i - ws >= 0 ? w[i - ws] << bs : 0;
            if (bs && i - ws - 1 >= 0) v |= w[i - ws - 1] >> (64 - bs);
            r.w[i] = v;
        }
        return r;
    }
    BigKey operator>>(int s) const {
        BigKey r; const int ws = s >> 6, bs = s & 63;
        for (int i = 0; i < N; i++) {
            uint64_t v = i + ws < N ? w[i + ws] >> bs : 0;
            if (bs && i + ws + 1 < N) v |= w[i + ws + 1] << (64 - bs);
            r.w[i] = v;
        }
        return r;
    }
    BigKey operator&(const BigKey& o) const { BigKey r; for (int i = 0; i < N; i++) r.w[i] = w[i] & o.w[i]; return r; }
    BigKey operator|(const BigKey& o) const { BigKey r; for (int i = 0; i < N; i++) r.w[i] = w[i] | o.w[i]; return r; }
    BigKey operator^(const BigKey& o) const { BigKey r; for (int i = 0; i < N; i++) r.w[i] = w[i] ^ o.w[i]; return r; }
    BigKey operator~() const { BigKey r; for (int i = 0; i < N; i++) r.w[i] = ~w[i]; return r; }
    BigKey& operator|=(const BigKey& o) { for (int i = 0; i < N; i++) w[i] |= o.w[i]; return *this; }
    BigKey operator-(const BigKey& o) const {
        BigKey r; unsigned __int128 borrow = 0;
        for (int i = 0; i < N; i++) { const unsigned __int128 d = (unsigned __int128)w[i] - o.w[i] - borrow; r.w[i] = (uint64_t)d; borrow = (d >> 64) & 1; }
        return r;
    }
    bool operator==(const BigKey& o) const { for (int i = 0; i < N; i++) if (w[i] != o.w[i]) return false; return true; }
    bool operator!=(const BigKey& o) const { return !(*this == o); }
    bool operator<(const BigKey& o) const { for (int i = N - 1; i >= 0; i--) if (w[i] != o.w[i]) return w[i] < o.w[i]; return false; }
};
template <int N>
static inline int clz_key(const BigKey<N>& x) {
    for (int i = N - 1; i >= 0; i--) if (x.w[i]) return 64 * (N - 1 - i) + __builtin_clzll(x.w[i]);
    return 64 * N;
}

static inline int clz_key(uint64_t x) { return __builtin_clzll(x); }
static inline int clz_key(u128 x) {
    uint64_t hi = (uint64_t)(x >> 64);
    return hi ? __builtin_clzll(hi) : 64 + __builtin_clzll((uint64_t)x);
}

template <typename K>
struct Dummy {
    K pkey;          // real chars moved to the top positions, '$' positions zero
    uint32_t len;    // number of real chars (0 = root)
    bool operator<(const Dummy& o) const { return pkey != o.pkey ? pkey < o.pkey : len < o.len; }
    bool operator==(const Dummy& o) const { return pkey == o.pkey && len == o.len; }
};

template <typename K>
struct Builder {
    static constexpr int KBITS = (int)sizeof(K) * 8;
    int k, kb;
    K mask_k, mask_p;
    const uint8_t* codes; const uint64_t* offs; uint64_t nu;
    std::vector<K> kmers; uint64_t m = 0;
    int B = 0, shift = 0;
    std::vector<uint64_t> bk;      // bucket start in kmers, 2^B + 1 entries
    std::vector<Dummy<K>> dum; std::vector<uint64_t> dpos; uint64_t D = 0;
    std::vector<uint64_t> dlo;     // per bucket: #dummies with dpos < bk[b]

    inline uint64_t bucket_of(K key) const { return B == 0 ? 0 : (uint64_t)(key >> shift); }
    inline uint64_t lower_bound_kmer(K key) const {
        uint64_t b = bucket_of(key);
        return (uint64_t)(std::lower_bound(kmers.begin() + bk[b], kmers.begin() + bk[b + 1], key) - kmers.begin());
    }
    inline int64_t find_kmer(K key) const {
        uint64_t r = lower_bound_kmer(key);
        return (r < m && kmers[r] == key) ? (int64_t)r : -1;
    }
    inline uint64_t node_of_dummy(uint64_t d) const { return d + dpos[d]; }
    // node index of k-mer rank r, which lies in bucket b (bk[b] <= r <= bk[b+1])
    inline uint64_t node_of_rank(uint64_t r, uint64_t b) const {
        uint64_t lo = dlo[b], hi = dlo[b + 1];
        uint64_t c = (uint64_t)(std::upper_bound(dpos.begin() + lo, dpos.begin() + hi, r) - dpos.begin());
        return r + c;
    }
    inline int64_t find_dummy(K pkey, uint32_t len) const {
        Dummy<K> q{pkey, len};
        auto it = std::lower_bound(dum.begin(), dum.end(), q);
        return (it != dum.end() && *it == q) ? (int64_t)(it - dum.begin()) : -1;
    }
    inline K key_at(uint64_t s) const {   // k-mer starting at codes[s]
        K key = 0;
        for (int j = 0; j < k; j++) key |= (K)codes[s + j] << (2 * j);
        return key;
    }

    int run(fin_index& out, std::string& err);
};

template <typename K>
int Builder<K>::run(fin_index& out, std::string& err) {
    kb = 2 * k;
    mask_k = (kb == KBITS) ? ~(K)0 : (((K)1 << kb) - 1);
    mask_p = ((K)1 << (kb - 2)) - 1;
    const int nt = omp_get_max_threads();

    // ---- 1. all k-mer occurrences ----
    std::vector<uint64_t> koff(nu + 1, 0);
    for (uint64_t u = 0; u < nu; u++) koff[u + 1] = koff[u] + (offs[u + 1] - offs[u] - (uint64_t)k + 1);
    const uint64_t T = koff[nu];
    std::vector<K> raw(T);
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t u = 0; u < nu; u++) {
        uint64_t s = offs[u], len = offs[u + 1] - offs[u];
        K key = key_at(s);
        K* dst = raw.data() + koff[u];
        dst[0] = key;
        for (uint64_t p = 1; p + k <= len; p++) {
            key = (key >> 2) | ((K)codes[s + p + k - 1] << (kb - 2));
            dst[p] = key;
        }
    }

    // ---- 2. colex sort = integer sort of the keys: one MSD partition pass, then per-bucket sorts ----
    B = kb - 2; if (B > 20) B = 20;
    while (B > 0 && (T >> B) < 48) B--;
    shift = kb - B;
    const uint64_t NB = (uint64_t)1 << B;
    {
        std::vector<uint64_t> hist((size_t)nt * NB, 0);
#pragma omp parallel num_threads(nt)
        {
            int tid = omp_get_thread_num();
            uint64_t lo = T * tid / nt, hi = T * (tid + 1) / nt;
            uint64_t* h = hist.data() + (size_t)tid * NB;
            for (uint64_t i = lo; i < hi; i++) h[bucket_of(raw[i])]++;
        }
        std::vector<uint64_t> bstart(NB + 1);
        uint64_t run_ = 0;
        for (uint64_t b = 0; b < NB; b++) {
            bstart[b] = run_;
            for (int t = 0; t < nt; t++) { uint64_t c = hist[(size_t)t * NB + b]; hist[(size_t)t * NB + b] = run_; run_ += c; }
        }
        bstart[NB] = run_;
        std::vector<K> sorted(T);
#pragma omp parallel num_threads(nt)
        {
            int tid = omp_get_thread_num();
            uint64_t lo = T * tid / nt, hi = T * (tid + 1) / nt;
            uint64_t* h = hist.data() + (size_t)tid * NB;
            for (uint64_t i = lo; i < hi; i++) sorted[h[bucket_of(raw[i])]++] = raw[i];
        }
        std::vector<K>().swap(raw);
        std::vector<uint64_t>().swap(hist);
        std::vector<uint64_t> ucnt(NB + 1, 0);
#pragma omp parallel for schedule(dynamic, 64)
        for (uint64_t b = 0; b < NB; b++) {
            K* s = sorted.data() + bstart[b]; K* e = sorted.data() + bstart[b + 1];
            std::sort(s, e);
            ucnt[b] = (uint64_t)(std::unique(s, e) - s);
        }
        bk.assign(NB + 1, 0);
        for (uint64_t b = 0; b < NB; b++) bk[b + 1] = bk[b] + ucnt[b];
        m = bk[NB];
        kmers.resize(m);
#pragma omp parallel for schedule(dynamic, 256)
        for (uint64_t b = 0; b < NB; b++)
            std::copy(sorted.begin() + bstart[b], sorted.begin() + bstart[b] + ucnt[b], kmers.begin() + bk[b]);
    }

    // ---- 3. dummy nodes: $-padded proper prefixes of every k-mer without a predecessor, plus the root ----
    // Only a unitig's first k-mer can lack a predecessor (every other k-mer follows one in its own unitig).
    {
        std::vector<std::vector<Dummy<K>>> loc(nt);
#pragma omp parallel for schedule(dynamic, 256)
        for (uint64_t u = 0; u < nu; u++) {
            K X = key_at(offs[u]);
            K P = X & mask_p;
            K q = (K)(P << 2);
            uint64_t g = lower_bound_kmer(q);
            bool has_pred = g < m && (K)(kmers[g] >> 2) == P;
            if (!has_pred) {
                auto& v = loc[omp_get_thread_num()];
                for (int j = 1; j < k; j++) v.push_back(Dummy<K>{(K)((X << (2 * (k - j))) & mask_k), (uint32_t)j});
            }
        }
        dum.push_back(Dummy<K>{0, 0});
        for (auto& v : loc) { dum.insert(dum.end(), v.begin(), v.end()); std::vector<Dummy<K>>().swap(v); }
        std::sort(dum.begin(), dum.end());
        dum.erase(std::unique(dum.begin(), dum.end()), dum.end());
        D = dum.size();
        dpos.resize(D);
#pragma omp parallel for schedule(static)
        for (uint64_t d = 0; d < D; d++) dpos[d] = lower_bound_kmer(dum[d].pkey);
        dlo.resize(NB + 1);
#pragma omp parallel for schedule(static)
        for (uint64_t b = 0; b <= NB; b++) dlo[b] = (uint64_t)(std::lower_bound(dpos.begin(), dpos.end(), bk[b]) - dpos.begin());
    }
    const uint64_t n = m + D;
    if (n >= 0xFFFFFFC0ull) { err = "index too large for one index: n_nodes >= 2^32 (use a partitioned index, fin_pindex_build_device)"; return -5; }
    const uint64_t nblk = (n + 63) / 64;
    if (!out.blocks.resize(nblk)) { err = "out of memory (blocks)"; return -4; }
    FinNodeBlock* Bk = out.blocks.p;
    out.lcs8.clear();
    if (k > FIN_FAST_K) out.lcs8.assign(n, 0);   // exact values beside the 7-bit ones of the node bytes
    uint8_t* const lcs8 = out.lcs8.empty() ? nullptr : out.lcs8.data();

    // ---- 4. node bytes: LCS[i] = common suffix length of node i and node i-1 ('$' never extends a match) ----
    {
        const uint64_t CH = 512;   // blocks per task
        const uint64_t ntask = (nblk + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 4)
        for (uint64_t t = 0; t < ntask; t++) {
            uint64_t s = t * CH * 64, e = std::min(n, (t + 1) * CH * 64);
            uint64_t from = s == 0 ? 0 : s - 1;
            // number of dummies among the first `from` nodes: smallest d with d + dpos[d] >= from
            uint64_t lo = 0, hi = D;
            while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (mid + dpos[mid] < from) lo = mid + 1; else hi = mid; }
            uint64_t d = lo, r = from - d;
            K prev_key = 0; uint32_t prev_len = 0;
            for (uint64_t i = from; i < e; i++) {
                K key; uint32_t len;
                if (d < D && d + dpos[d] == i) { key = dum[d].pkey; len = dum[d].len; d++; }
                else { key = kmers[r]; len = (uint32_t)k; r++; }
                if (i >= s) {
                    uint32_t lcs = 0;
                    if (i > 0) {
                        K x = key ^ prev_key;
                        uint32_t match = x == 0 ? (uint32_t)k : (uint32_t)((clz_key(x) - (KBITS - kb)) / 2);
                        lcs = std::min(match, std::min(len, prev_len));
                    }
                    Bk[i >> 6].node[i & 63] = (uint8_t)std::min<uint32_t>(lcs, FIN_LCS_MASK);
                    if (lcs8) lcs8[i] = (uint8_t)lcs;
                }
                prev_key = key; prev_len = len;
            }
        }
    }

    // ---- 5. planes: every non-root node v has one marked in-edge, labelled with v's last char, leaving the
    //         first node of the group whose (k-1)-suffix equals v's (k-1)-prefix ----
    auto set_plane = [&](int c, uint64_t u) {
        const uint32_t o = (uint32_t)(u & 63);
        __atomic_fetch_or(o < 32 ? &Bk[u >> 6].rec[c].plane_lo : &Bk[u >> 6].rec[c].plane_hi, 1u << (o & 31), __ATOMIC_RELAXED);
    };
    {
        const uint64_t NBk = (uint64_t)1 << B;
#pragma omp parallel for schedule(dynamic, 64)
        for (uint64_t b = 0; b < NBk; b++) {
            for (uint64_t v = bk[b]; v < bk[b + 1]; v++) {
                K X = kmers[v];
                int c = (int)(X >> (kb - 2)) & 3;
                K P = X & mask_p;
                K q = (K)(P << 2);
                uint64_t gb = bucket_of(q);
                uint64_t g = (uint64_t)(std::lower_bound(kmers.begin() + bk[gb], kmers.begin() + bk[gb + 1], q) - kmers.begin());
                uint64_t u;
                if (g < m && (K)(kmers[g] >> 2) == P) u = node_of_rank(g, gb);
                else {
                    int64_t d = find_dummy((K)((X << 2) & mask_k), (uint32_t)(k - 1));
                    if (d < 0) { d = 0; }   // cannot happen: step 3 created it
                    u = node_of_dummy((uint64_t)d);
                }
                set_plane(c, u);
            }
        }
#pragma omp parallel for schedule(static)
        for (uint64_t d = 1; d < D; d++) {
            K pk = dum[d].pkey; uint32_t j = dum[d].len;
            int c = (int)(pk >> (kb - 2)) & 3;
            int64_t p = find_dummy((K)((pk << 2) & mask_k), j - 1);
            if (p < 0) p = 0;
            set_plane(c, node_of_dummy((uint64_t)p));
        }
    }

    // ---- 6. C array and per-block bases (C[c] + rank_c(64 b)) ----
    {
        uint64_t tot[4] = {0, 0, 0, 0};
        for (uint64_t b = 0; b < nblk; b++) for (int c = 0; c < 4; c++) tot[c] += (uint64_t)__builtin_popcountll(fin_plane(Bk[b].rec[c]));
        out.C[0] = 1;
        for (int c = 0; c < 3; c++) out.C[c + 1] = out.C[c] + tot[c];
        if (out.C[3] + tot[3] != n) { err = "internal error: SBWT edge count does not match node count"; return -1; }
        uint64_t run_[4] = {out.C[0], out.C[1], out.C[2], out.C[3]};
        for (uint64_t b = 0; b < nblk; b++)
            for (int c = 0; c < 4; c++) { Bk[b].rec[c].base = (uint32_t)run_[c]; run_[c] += (uint64_t)__builtin_popcountll(fin_plane(Bk[b].rec[c])); }
    }

    // ---- 7. permute_unitigs: order by colex of the first k-mer (ties by input order), Ustart marks ----
    std::vector<uint64_t> perm(nu);
    {
        std::vector<std::pair<K, uint64_t>> fk(nu);
#pragma omp parallel for schedule(static)
        for (uint64_t u = 0; u < nu; u++) fk[u] = {key_at(offs[u]), u};
        std::sort(fk.begin(), fk.end());
        for (uint64_t r = 0; r < nu; r++) perm[r] = fk[r].second;
        for (uint64_t u = 0; u < nu; u++) {
            K X = fk[u].first;
            uint64_t b = bucket_of(X);
            int64_t r = find_kmer(X);
            if (r < 0) { err = "internal error: first k-mer of a unitig missing from the SBWT"; return -1; }
            uint64_t node = node_of_rank((uint64_t)r, b);
            Bk[node >> 6].node[node & 63] |= FIN_USTART_BIT;
        }
    }
    uint64_t total_len = offs[nu] - offs[0];
    if (total_len >= 0xFFFFFFF0ull) { err = "index too large for one index: total unitig length >= 2^32 (use a partitioned index, fin_pindex_build_device)"; return -5; }
    out.k = (uint32_t)k; out.n_nodes = n; out.n_kmers = m; out.n_unitigs = nu; out.total_len = total_len;
    out.ends.assign(nu + 1 + 8, 0xFFFFFFFFu);
    out.ends[0] = 0;
    std::vector<uint64_t> ustart_of(nu + 1, 0);   // global start of permuted unitig r
    for (uint64_t r = 0; r < nu; r++) {
        uint64_t u = perm[r];
        ustart_of[r + 1] = ustart_of[r] + (offs[u + 1] - offs[u]);
        out.ends[r + 1] = (uint32_t)ustart_of[r + 1];
    }
    out.concat.assign(total_len / 16 + 8, 0);
    {
        // each permuted unitig writes its own bit range; words shared by two unitigs are merged with atomics
#pragma omp parallel for schedule(dynamic, 64)
        for (uint64_t r = 0; r < nu; r++) {
            uint64_t u = perm[r], len = offs[u + 1] - offs[u], g0 = ustart_of[r];
            const uint8_t* s = codes + offs[u];
            uint64_t j = 0;
            while (j < len) {
                uint64_t g = g0 + j, w = g >> 4;
                uint32_t val = 0; uint64_t g_end = std::min(g0 + len, (w + 1) << 4);
                for (uint64_t gg = g; gg < g_end; gg++, j++) val |= (uint32_t)s[j] << (2 * (gg & 15));
                __atomic_fetch_or(&out.concat[w], val, __ATOMIC_RELAXED);
            }
        }
    }

    // ---- 8. finimizers: FinimizerIndexBuilder::add_sequence over every unitig (FinimizerIndex.hh:321-389).
    // The sequential overwrite rule `fmin_found == 0 || fmin_found < end` (:370) keeps, per colex, the event with
    // the largest in-unitig end, the first unitig in processing order on ties -- except that an end of 0 reads as
    // "unset", so if every event has end 0 the last one stays.  As a max-reduction over keys:
    //   end > 0 : (end << 32) | (0xFFFFFFFF - global_offset)      end == 0 : global_offset + 1
    std::vector<uint64_t> best(n, 0);
    {
        struct T4 { int64_t f, len, colex, end; };
        auto gt = [](const T4& a, const T4& b) {
            if (a.f != b.f) return a.f > b.f;
            if (a.len != b.len) return a.len > b.len;
            if (a.colex != b.colex) return a.colex > b.colex;
            return a.end > b.end;
        };
        const int64_t nn = (int64_t)n;
        uint32_t cap = 1; while (cap < (uint32_t)(k + 4)) cap <<= 1;
#pragma omp parallel
        {
            std::vector<T4> ring(cap);
#pragma omp for schedule(dynamic, 16)
            for (uint64_t r = 0; r < nu; r++) {
                uint64_t u = perm[r];
                const uint8_t* s = codes + offs[u];
                const int64_t str_len = (int64_t)(offs[u + 1] - offs[u]);
                const uint64_t unitig_start = ustart_of[r];
                uint32_t head = 0, cnt = 0;   // deque = ring[head .. head+cnt)
                T4 w{nn, k + 1, nn, str_len};
                T4 curr{0, 0, 0, 0};
                int64_t kmer = 0, start = 0;
                FinIval I{0, nn - 1};
                for (int64_t end = 0; end < str_len; end++) {
                    I = fin_host_extend(Bk, s[end], I);
                    int64_t freq = I.second - I.first + 1;
                    int64_t I_start = I.first;
                    if (freq == 1) {
                        while (freq == 1) {
                            curr = T4{freq, end - start + 1, I_start, end};
                            start++;
                            I = fin_host_drop(Bk, lcs8, nn, end - start + 1, I);
                            freq = I.second - I.first + 1;
                            I_start = I.first;
                        }
                        if (gt(w, curr)) { head = 0; cnt = 0; w = curr; }
                        else { while (cnt > 0 && gt(ring[(head + cnt - 1) & (cap - 1)], curr)) cnt--; }
                        ring[(head + cnt) & (cap - 1)] = curr; cnt++;
                    }
                    if (end >= k - 1) {
                        uint64_t off = unitig_start + (uint64_t)w.end;
                        uint64_t key = w.end > 0 ? (((uint64_t)w.end << 32) | (0xFFFFFFFFull - off)) : off + 1;
                        if (w.colex >= 0 && w.colex < nn) {
                            uint64_t cur = __atomic_load_n(&best[w.colex], __ATOMIC_RELAXED);
                            while (cur < key && !__atomic_compare_exchange_n(&best[w.colex], &cur, key, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                        }
                        kmer++;
                        while (w.end - w.len + 1 < kmer) {
                            head = (head + 1) & (cap - 1); if (cnt > 0) cnt--;
                            if (cnt == 0) w = T4{nn, k + 1, kmer + 1, kmer + k};
                            else w = ring[head];
                        }
                    }
                }
            }
        }
    }
    {
        uint64_t nf = 0, nus = 0;
        out.blkinfo.assign(nblk + 2, FinBlockInfo{0, 0, 0, 0, 0, 0});
        for (uint64_t b = 0; b < nblk; b++) {
            out.blkinfo[b].ustart_rank = (uint32_t)nus; out.blkinfo[b].fmin_rank = (uint32_t)nf;
            uint64_t lim = std::min<uint64_t>(64, n - b * 64);
            uint64_t fm = 0, um = 0;
            for (uint64_t j = 0; j < lim; j++) {
                uint64_t key = best[b * 64 + j];
                if (key) {
                    fm |= 1ull << j; nf++;
                    out.goff.push_back((key >> 32) ? (uint32_t)(0xFFFFFFFFull - (key & 0xFFFFFFFFull)) : (uint32_t)(key - 1));
                }
                if (Bk[b].node[j] & FIN_USTART_BIT) { um |= 1ull << j; nus++; }
            }
            out.blkinfo[b].fmin_mask_lo = (uint32_t)fm; out.blkinfo[b].fmin_mask_hi = (uint32_t)(fm >> 32);
            out.blkinfo[b].ustart_mask_lo = (uint32_t)um; out.blkinfo[b].ustart_mask_hi = (uint32_t)(um >> 32);
        }
        out.blkinfo[nblk].ustart_rank = out.blkinfo[nblk + 1].ustart_rank = (uint32_t)nus;
        out.blkinfo[nblk].fmin_rank = out.blkinfo[nblk + 1].fmin_rank = (uint32_t)nf;
        out.goff.resize(out.goff.size() + 8, 0);   // padding so that 16-byte reads of any element stay inside
        out.n_fmin = nf;
    }
    fin_finish_sampling(out);
    fin_finish_thermometer(out, -1);
    return 0;
}

// Thermometer planes th0/th1 (fin_format.h): pick the three consecutive LCS thresholds that cover the most nodes (the drop
// thresholds cluster around log4(n), like the LCS values themselves; any choice is correct, only speed depends on it).
void fin_finish_thermometer(fin_index& x, int forced_t0) {
    FinNodeBlock* Bk = x.blocks.p;
    const uint64_t n = x.n_nodes, nblk = x.blocks.n;
    uint64_t hist[128] = {0};
    for (uint64_t i = 0; i < n; i++) hist[Bk[i >> 6].node[i & 63] & FIN_LCS_MASK]++;
    int t0 = 0; uint64_t best = 0;
    for (int t = 0; t + 3 < 128; t++) { uint64_t sum = hist[t + 1] + hist[t + 2] + hist[t + 3]; if (sum > best) { best = sum; t0 = t; } }
    if (const char* e = getenv("FINITO_LCS_T0")) { int v = atoi(e); if (v >= 0 && v < 124) t0 = v; }
    if (forced_t0 >= 0) t0 = forced_t0;
    x.lcs_t0 = (uint32_t)t0;
#pragma omp parallel for schedule(static)
    for (uint64_t b = 0; b < nblk; b++) {
        uint64_t p0 = 0, p1 = 0;
        for (int j = 0; j < 64; j++) {
            int c = (int)(Bk[b].node[j] & FIN_LCS_MASK) - t0;
            c = c < 0 ? 0 : (c > 3 ? 3 : c);
            p0 |= (uint64_t)(c & 1) << j; p1 |= (uint64_t)(c >> 1) << j;
        }
        Bk[b].th0 = p0; Bk[b].th1 = p1;
    }
}

// samp[j] = number of unitig ends <= (j << samp_shift): turns PackedStrings::global_offset_to_local_offset's
// upper_bound (PackedStrings.hh:95) into one table read plus a short forward scan with the same result.
void fin_finish_sampling(fin_index& x) {
    uint64_t avg = x.n_unitigs ? x.total_len / x.n_unitigs : 1;
    uint32_t sh = 4;
    while ((1ull << (sh + 1)) <= avg && sh < 20) sh++;
    x.samp_shift = sh;
    uint64_t ns = (x.total_len >> sh) + 2;
    x.samp.assign(ns + 8, 0);
    uint64_t u = 0;
    for (uint64_t j = 0; j < ns + 8; j++) {
        uint64_t g = j << sh;
        while (u < x.n_unitigs && x.ends[u + 1] <= g) u++;
        x.samp[j] = (uint32_t)u;
    }
}

int fin_build_index(const char* bases, const uint64_t* offsets, uint64_t n_unitigs, int k, int n_threads,
                    fin_index& out, std::string& err) {
    if (k < 2 || k > FIN_MAX_K) {
        err = "k must be in [2, " + std::to_string(FIN_MAX_K) + "] (got " + std::to_string(k) + "): LCS values are kept in a byte, as in the reference";
        return k > FIN_MAX_K ? -5 : -1;
    }
    if (n_unitigs == 0) { err = "no unitigs"; return -1; }
    if (n_threads > 0) omp_set_num_threads(n_threads);
    const uint64_t base0 = offsets[0], total = offsets[n_unitigs] - base0;
    std::vector<uint8_t> codes(total + 1);
    std::vector<uint64_t> offs(n_unitigs + 1);
    for (uint64_t u = 0; u <= n_unitigs; u++) offs[u] = offsets[u] - base0;
    for (uint64_t u = 0; u < n_unitigs; u++)
        if (offs[u + 1] - offs[u] < (uint64_t)k) { err = "unitig " + std::to_string(u) + " is shorter than k"; return -1; }
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (uint64_t i = 0; i < total; i++) {
        uint8_t c = (uint8_t)bases[base0 + i] & (uint8_t)~32u;
        uint8_t v;
        switch (c) { case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break; default: v = 0; bad = 1; }
        codes[i] = v;
    }
    if (bad) { err = "unitigs contain a base outside ACGT (the reference's PackedStrings throws here, PackedStrings.hh:57)"; return -1; }
    if (k <= 32) {
        Builder<uint64_t> b; b.k = k; b.codes = codes.data(); b.offs = offs.data(); b.nu = n_unitigs;
        return b.run(out, err);
    } else if (k <= 64) {
        Builder<u128> b; b.k = k; b.codes = codes.data(); b.offs = offs.data(); b.nu = n_unitigs;
        return b.run(out, err);
    } else if (k <= 96) {
        Builder<BigKey<3>> b; b.k = k; b.codes = codes.data(); b.offs = offs.data(); b.nu = n_unitigs;
        return b.run(out, err);
    } else if (k <= 128) {
        Builder<BigKey<4>> b; b.k = k; b.codes = codes.data(); b.offs = offs.data(); b.nu = n_unitigs;
        return b.run(out, err);
    } else if (k <= 192) {
        Builder<BigKey<6>> b; b.k = k; b.codes = codes.data(); b.offs = offs.data(); b.nu = n_unitigs;
        return b.run(out, err);
    } else {
        Builder<BigKey<8>> b; b.k = k; b.codes = codes.data(); b.offs = offs.data(); b.nu = n_unitigs;
        return b.run(out, err);
    }
}

// ---- container file -----------------------------------------------------------------------------------------
template <typename T>
static bool wr(FILE* f, const T* p, uint64_t n) { return n == 0 || fwrite(p, sizeof(T), n, f) == n; }
template <typename T>
static bool rd(FILE* f, T* p, uint64_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }

int fin_save_index(const fin_index& x, const std::string& prefix, std::string& err) {
    std::string path = prefix + ".finamd";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open " + path + " for writing"; return -2; }
    FinFileHeader h; memset(&h, 0, sizeof h);
    h.magic = FIN_MAGIC; h.version = x.lcs8.empty() ? 4 : 5; h.lcs_t0 = x.lcs_t0; h.k = x.k;
    h.n_nodes = x.n_nodes; h.n_kmers = x.n_kmers; h.n_unitigs = x.n_unitigs; h.total_len = x.total_len; h.n_fmin = x.n_fmin;
    for (int c = 0; c < 4; c++) h.C[c] = x.C[c];
    h.samp_shift = x.samp_shift; h.n_samp = (uint32_t)x.samp.size();
    h.n_blocks = x.blocks.n; h.n_concat_words = x.concat.size();
    bool ok = wr(f, &h, 1) && wr(f, x.blocks.p, x.blocks.n) && wr(f, x.blkinfo.data(), x.blkinfo.size()) && wr(f, x.goff.data(), x.goff.size()) &&
              wr(f, x.ends.data(), x.ends.size()) && wr(f, x.samp.data(), x.samp.size()) && wr(f, x.concat.data(), x.concat.size()) &&
              wr(f, x.lcs8.data(), x.lcs8.size());
    ok = (fclose(f) == 0) && ok;
    if (!ok) { err = "write error on " + path; return -2; }
    return 0;
}

int fin_load_index(const std::string& prefix, fin_index& x, std::string& err) {
    std::string path = prefix + ".finamd";
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return -2; }
    FinFileHeader h;
    if (!rd(f, &h, 1) || h.magic != FIN_MAGIC || (h.version != 4 && h.version != 5)) { fclose(f); err = path + " is not a finito-amd index container (version 4 or 5)"; return -2; }
    if (h.k < 2 || h.k > FIN_MAX_K || (h.version == 5) != (h.k > FIN_FAST_K) || h.n_blocks != (h.n_nodes + 63) / 64 || h.n_nodes >= 0xFFFFFFC0ull) { fclose(f); err = path + ": inconsistent header"; return -2; }
    x.k = h.k; x.n_nodes = h.n_nodes; x.n_kmers = h.n_kmers; x.n_unitigs = h.n_unitigs; x.total_len = h.total_len; x.n_fmin = h.n_fmin;
    for (int c = 0; c < 4; c++) x.C[c] = h.C[c];
    x.samp_shift = h.samp_shift; x.lcs_t0 = (uint32_t)h.lcs_t0;
    if (!x.blocks.resize(h.n_blocks)) { fclose(f); err = "out of memory"; return -4; }
    x.blkinfo.resize(h.n_blocks + 2); x.goff.resize(h.n_fmin + 8); x.ends.resize(h.n_unitigs + 9); x.samp.resize(h.n_samp); x.concat.resize(h.n_concat_words);
    bool ok = rd(f, x.blocks.p, x.blocks.n) && rd(f, x.blkinfo.data(), x.blkinfo.size()) && rd(f, x.goff.data(), x.goff.size()) && rd(f, x.ends.data(), x.ends.size()) &&
              rd(f, x.samp.data(), x.samp.size()) && rd(f, x.concat.data(), x.concat.size());
    x.lcs8.clear();
    if (ok && h.version == 5) { x.lcs8.resize(h.n_nodes); ok = rd(f, x.lcs8.data(), x.lcs8.size()); }
    fclose(f);
    if (!ok) { err = path + ": truncated"; return -2; }
    // endpoints: increasing, inside the text, every unitig at least k long (the device code walks them without further checks)
    if (x.ends.empty() || x.ends[0] != 0) { err = path + ": unitig endpoints damaged"; return -2; }
    for (uint64_t u = 0; u < x.n_unitigs; u++)
        if (x.ends[u + 1] < x.ends[u] || x.ends[u + 1] > x.total_len || (uint64_t)x.ends[u + 1] - x.ends[u] < x.k) { err = path + ": unitig endpoints damaged (not increasing, beyond the text, or a unitig shorter than k)"; return -2; }
    return 0;
}
