// fin_sdsl.cpp -- the reference's on-disk index layout: reader and writer.
//
// FinimizerIndex::serialize / load (FinimizerIndex.hh:187-241) keep an index as seven files:
//   <p>.O.sdsl                 sdsl::int_vector<>   global_offsets, one entry per finimizer in fmin-rank order
//   <p>.FBV.sdsl               sdsl::bit_vector     fmin, one bit per SBWT node
//   <p>.packed_unitigs.sdsl    sdsl::int_vector<2>  the unitigs concatenated in colex order of their first k-mer, A0 C1 G2 T3
//   <p>.unitig_endpoints.sdsl  sdsl::int_vector<>   exclusive end of every unitig in that concatenation
//   <p>.Ustart.sdsl            sdsl::bit_vector     Ustart, one bit per SBWT node
//   <p>.LCS.sdsl               sdsl::int_vector<>   LCS, one entry per SBWT node
//   <p>.sbwt                   sbwt::plain_matrix_sbwt_t::serialize (no variant string; the file `sbwt build` writes and
//                              build-fmin -i reads has the string "plain-matrix" in front, build_fmin.hh:353-364)
//
// PARITY UNPINNED.  The reference tree holds no index file, no serialization test, and neither sdsl-lite nor algbio/SBWT (both live
// in an empty, un-pinned submodule).  The byte layouts below are those libraries' published formats, restated:
//   sdsl::int_vector<w>::serialize (sdsl-lite v2, int_vector.hpp):  u64 size in BITS; for w = 0 (runtime width) one u8 width;
//       then ceil(bits / 64) little-endian u64 words, element i at bits [i*width, (i+1)*width)
//   sdsl::rank_support_v5<>::serialize:  its int_vector<64> of superblock counts (2 words per 2048 bits); never needed here, the
//       reader skips it and the writer fills it by rank_support_v5's construction rule
//   sbwt::SBWT<SubsetMatrixRank>::serialize (SBWT.hh):  string version; A,C,G,T bit_vectors; their four rank supports;
//       bit_vector suffix_group_starts (may be empty); vector<int64> C; vector<pair<int64,int64>> kmer_prefix_precalc;
//       int64 precalc_k, n_nodes, n_kmers, k.   strings = int64 length + bytes; std::vectors = int64 BYTE count + raw data.
// What is tested: writer -> reader round trip bit for bit, the layout above byte by byte on a small index, and that an index loaded
// from the seven files answers exactly like the index it was written from (tests/test_sdsl_format.py).  Whether a file written by
// the real tools loads is unverified until somebody supplies one.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "fin_index.hpp"

namespace {

struct File {
    FILE* f = nullptr;
    std::string path;
    File(const std::string& p, const char* mode) : path(p) { f = fopen(p.c_str(), mode); }
    ~File() { if (f) fclose(f); }
    bool rd(void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
    bool wr(const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
};

struct IntVec { uint64_t bits = 0; uint8_t width = 0; std::vector<uint64_t> w; };   // width 0 never occurs after a successful read

inline uint64_t iv_get(const IntVec& v, uint64_t i) {
    const uint64_t bit = i * v.width, wi = bit >> 6; const unsigned off = (unsigned)(bit & 63);
    uint64_t x = v.w[wi] >> off;
    if (off + v.width > 64) x |= v.w[wi + 1] << (64 - off);
    return v.width == 64 ? x : (x & ((1ull << v.width) - 1));
}
inline void iv_set(IntVec& v, uint64_t i, uint64_t x) {
    const uint64_t bit = i * v.width, wi = bit >> 6; const unsigned off = (unsigned)(bit & 63);
    const uint64_t mask = v.width == 64 ? ~0ull : ((1ull << v.width) - 1);
    v.w[wi] = (v.w[wi] & ~(mask << off)) | ((x & mask) << off);
    if (off + v.width > 64) { const unsigned hi = off + v.width - 64; v.w[wi + 1] = (v.w[wi + 1] & ~((1ull << hi) - 1)) | ((x & mask) >> (64 - off)); }
}
void iv_init(IntVec& v, uint64_t n, uint8_t width) { v.width = width; v.bits = n * width; v.w.assign((v.bits + 63) / 64 + 1, 0); }
inline int bits_needed(uint64_t x) { return x == 0 ? 1 : 64 - __builtin_clzll(x); }

// sdsl::int_vector<fixed>::load; fixed = 0: runtime width (the width byte is in the file)
bool read_int_vector(File& f, uint8_t fixed, IntVec& v, std::string& err, uint64_t max_bits = 1ull << 40) {
    if (!f.rd(&v.bits, 8)) { err = f.path + ": truncated (size)"; return false; }
    v.width = fixed;
    if (fixed == 0) {
        if (!f.rd(&v.width, 1)) { err = f.path + ": truncated (width)"; return false; }
        if (v.width == 0 || v.width > 64) { err = f.path + ": bad int_vector width " + std::to_string(v.width); return false; }
    }
    if (v.bits > max_bits || v.bits % v.width != 0) { err = f.path + ": implausible int_vector size " + std::to_string(v.bits) + " bits"; return false; }
    const uint64_t nw = (v.bits + 63) / 64;
    v.w.assign(nw + 1, 0);   // one spare word: iv_get may look at w[wi+1]
    if (!f.rd(v.w.data(), nw * 8)) { err = f.path + ": truncated (data)"; return false; }
    return true;
}
bool write_int_vector(File& f, uint8_t fixed, const IntVec& v) {
    const uint64_t nw = (v.bits + 63) / 64;
    return f.wr(&v.bits, 8) && (fixed != 0 || f.wr(&v.width, 1)) && f.wr(v.w.data(), nw * 8);
}
bool read_file_vec(const std::string& path, uint8_t fixed, IntVec& v, std::string& err) {
    File f(path, "rb");
    if (!f.f) { err = "cannot open " + path; return false; }
    return read_int_vector(f, fixed, v, err);
}
bool write_file_vec(const std::string& path, uint8_t fixed, const IntVec& v, std::string& err) {
    File f(path, "wb");
    if (!f.f || !write_int_vector(f, fixed, v)) { err = "cannot write " + path; return false; }
    return true;
}

bool read_string(File& f, std::string& s, std::string& err) {
    int64_t n = 0;
    if (!f.rd(&n, 8) || n < 0 || n > 4096) { err = f.path + ": bad string length"; return false; }
    s.resize((size_t)n);
    if (!f.rd(&s[0], (size_t)n)) { err = f.path + ": truncated (string)"; return false; }
    return true;
}
bool write_string(File& f, const std::string& s) { const int64_t n = (int64_t)s.size(); return f.wr(&n, 8) && f.wr(s.data(), s.size()); }

// rank_support_v5<1>(v): its int_vector<64> m_basic_block, by the constructor's rule (sdsl-lite v2, rank_support_v5.hpp): per
// 2048-bit superblock one absolute count and one word of five 12-bit counts of the ones before each 384-bit block
IntVec rank_v5_blocks(const IntVec& bv) {
    IntVec r; r.width = 64;
    const uint64_t cap_words = (bv.bits + 63) / 64;
    if (bv.bits == 0) { r.bits = 2 * 64; r.w.assign(3, 0); return r; }
    const uint64_t nbb = (((cap_words * 64) >> 11) + 1) << 1;
    r.bits = nbb * 64; r.w.assign(nbb + 1, 0);
    uint64_t j = 0, sum = (uint64_t)__builtin_popcountll(bv.w[0]), second = 0, cnt_words = 1;
    for (uint64_t i = 1; i < cap_words; ++i, ++cnt_words) {
        if (cnt_words == 32) { j += 2; r.w[j - 1] = second; r.w[j] = r.w[j - 2] + sum; second = sum = cnt_words = 0; }
        else if (cnt_words % 6 == 0) second |= sum << (60 - 12 * (cnt_words / 6));
        sum += (uint64_t)__builtin_popcountll(bv.w[i]);
    }
    if (cnt_words % 6 == 0) second |= sum << (60 - 12 * (cnt_words / 6));
    if (cnt_words == 32) { j += 2; r.w[j - 1] = second; r.w[j] = r.w[j - 2] + sum; r.w[j + 1] = 0; }
    else r.w[j + 1] = second;
    return r;
}

struct SbwtFile {
    std::string version;
    IntVec plane[4];
    std::vector<int64_t> C;
    int64_t precalc_k = 0, n_nodes = 0, n_kmers = 0, k = 0;
};

bool read_sbwt(const std::string& path, bool with_variant, SbwtFile& s, std::string& err) {
    File f(path, "rb");
    if (!f.f) { err = "cannot open " + path; return false; }
    if (with_variant) {
        std::string variant;
        if (!read_string(f, variant, err)) return false;
        if (variant != "plain-matrix") { err = path + ": SBWT variant '" + variant + "' (only plain-matrix is supported, as in build_fmin.hh:353-361)"; return false; }
    }
    if (!read_string(f, s.version, err)) return false;
    for (int c = 0; c < 4; c++) if (!read_int_vector(f, 1, s.plane[c], err)) return false;
    for (int c = 0; c < 4; c++) { IntVec skip; if (!read_int_vector(f, 64, skip, err)) return false; }   // rank supports: rebuilt, not read
    { IntVec sgs; if (!read_int_vector(f, 1, sgs, err)) return false; }                                     // suffix_group_starts: not used on this path
    int64_t nbytes = 0;
    if (!f.rd(&nbytes, 8) || nbytes != 32) { err = path + ": C array is not 4 x int64"; return false; }
    s.C.resize(4);
    if (!f.rd(s.C.data(), 32)) { err = path + ": truncated (C)"; return false; }
    if (!f.rd(&nbytes, 8) || nbytes < 0 || nbytes % 16 != 0 || nbytes > (16ll << 32)) { err = path + ": bad k-mer prefix table size"; return false; }
    if (fseek(f.f, (long)nbytes, SEEK_CUR) != 0) { err = path + ": truncated (prefix table)"; return false; }
    if (!f.rd(&s.precalc_k, 8) || !f.rd(&s.n_nodes, 8) || !f.rd(&s.n_kmers, 8) || !f.rd(&s.k, 8)) { err = path + ": truncated (trailer)"; return false; }
    for (int c = 0; c < 4; c++)
        if ((int64_t)s.plane[c].bits != s.n_nodes) { err = path + ": bit-plane length " + std::to_string(s.plane[c].bits) + " != number of nodes " + std::to_string(s.n_nodes); return false; }
    if (s.k < 1 || s.k > 255 || s.n_kmers < 0 || s.n_kmers > s.n_nodes) { err = path + ": implausible k / k-mer count"; return false; }
    return true;
}

bool write_sbwt(const std::string& path, bool with_variant, const SbwtFile& s, std::string& err) {
    File f(path, "wb");
    if (!f.f) { err = "cannot write " + path; return false; }
    bool ok = true;
    if (with_variant) ok = ok && write_string(f, "plain-matrix");
    ok = ok && write_string(f, s.version);
    for (int c = 0; c < 4; c++) ok = ok && write_int_vector(f, 1, s.plane[c]);
    for (int c = 0; c < 4; c++) { const IntVec r = rank_v5_blocks(s.plane[c]); ok = ok && write_int_vector(f, 64, r); }
    { IntVec empty; empty.width = 1; empty.bits = 0; empty.w.assign(1, 0); ok = ok && write_int_vector(f, 1, empty); }
    int64_t nbytes = 32;
    ok = ok && f.wr(&nbytes, 8) && f.wr(s.C.data(), 32);
    const int64_t pre[2] = {0, s.n_nodes - 1};   // precalc_k = 0: one entry, the interval of the empty string
    nbytes = 16;
    ok = ok && f.wr(&nbytes, 8) && f.wr(pre, 16);
    const int64_t zero = 0;
    ok = ok && f.wr(&zero, 8) && f.wr(&s.n_nodes, 8) && f.wr(&s.n_kmers, 8) && f.wr(&s.k, 8);
    if (!ok) err = "write error on " + path;
    return ok;
}

}  // namespace

// ---- fin_index <-> the seven files ----------------------------------------------------------------------------------------
int fin_save_reference_layout(const fin_index& x, const std::string& prefix, std::string& err) {
    const uint64_t n = x.n_nodes, nblk = x.blocks.n;
    const FinNodeBlock* B = x.blocks.p;
    SbwtFile s;
    s.version = "v0.1"; s.k = x.k; s.n_nodes = (int64_t)n; s.n_kmers = (int64_t)x.n_kmers;
    s.C = {(int64_t)x.C[0], (int64_t)x.C[1], (int64_t)x.C[2], (int64_t)x.C[3]};
    IntVec fmin, ustart;
    for (int c = 0; c < 4; c++) iv_init(s.plane[c], n, 1);
    iv_init(fmin, n, 1); iv_init(ustart, n, 1);
    for (uint64_t b = 0; b < nblk; b++) {
        for (int c = 0; c < 4; c++) s.plane[c].w[b] = fin_plane(B[b].rec[c]);
        fmin.w[b] = x.blkinfo[b].fmin_mask_lo | ((uint64_t)x.blkinfo[b].fmin_mask_hi << 32);
        ustart.w[b] = x.blkinfo[b].ustart_mask_lo | ((uint64_t)x.blkinfo[b].ustart_mask_hi << 32);
    }
    IntVec lcs; iv_init(lcs, n, (uint8_t)bits_needed((uint64_t)x.k - 1));   // packed to bits(k-1) like lcs_basic_parallel_algorithm.hpp:115
    for (uint64_t i = 0; i < n; i++) iv_set(lcs, i, fin_host_lcs(B, x.lcs8_or_null(), (int64_t)i));
    uint64_t max_off = 0;
    for (uint64_t i = 0; i < x.n_fmin; i++) if (x.goff[i] > max_off) max_off = x.goff[i];
    IntVec goff; iv_init(goff, x.n_fmin, (uint8_t)bits_needed(max_off));          // FinimizerIndex.hh:301-306
    for (uint64_t i = 0; i < x.n_fmin; i++) iv_set(goff, i, x.goff[i]);
    // (PackedStrings.hh:44 is int_vector<>(n, 64 - clz(total_length)): the TWO-argument constructor, whose second argument is the default
    //  VALUE -- the width stays int_vector<>'s default, 64)
    IntVec ends; iv_init(ends, x.n_unitigs, 64);
    for (uint64_t u = 0; u < x.n_unitigs; u++) iv_set(ends, u, x.ends[u + 1]);
    IntVec concat; iv_init(concat, x.total_len, 2);
    for (uint64_t wi = 0; wi < (x.total_len + 31) / 32; wi++) {   // 16 bases per u32 here, 32 per u64 there: same bit order
        const uint64_t lo = 2 * wi < x.concat.size() ? x.concat[2 * wi] : 0, hi = 2 * wi + 1 < x.concat.size() ? x.concat[2 * wi + 1] : 0;
        concat.w[wi] = lo | (hi << 32);
    }
    if (x.total_len % 32) concat.w[x.total_len / 32] &= (1ull << (2 * (x.total_len % 32))) - 1;
    if (!write_file_vec(prefix + ".O.sdsl", 0, goff, err) || !write_file_vec(prefix + ".FBV.sdsl", 1, fmin, err) ||
        !write_file_vec(prefix + ".packed_unitigs.sdsl", 2, concat, err) || !write_file_vec(prefix + ".unitig_endpoints.sdsl", 0, ends, err) ||
        !write_file_vec(prefix + ".Ustart.sdsl", 1, ustart, err) || !write_file_vec(prefix + ".LCS.sdsl", 0, lcs, err) ||
        !write_sbwt(prefix + ".sbwt", false, s, err))
        return -2;
    return 0;
}

// the SBWT alone in the form `sbwt build` writes and build-fmin -i reads (variant string in front)
int fin_save_sbwt_file(const fin_index& x, const std::string& path, std::string& err) {
    SbwtFile s;
    s.version = "v0.1"; s.k = x.k; s.n_nodes = (int64_t)x.n_nodes; s.n_kmers = (int64_t)x.n_kmers;
    s.C = {(int64_t)x.C[0], (int64_t)x.C[1], (int64_t)x.C[2], (int64_t)x.C[3]};
    for (int c = 0; c < 4; c++) { iv_init(s.plane[c], x.n_nodes, 1); for (uint64_t b = 0; b < x.blocks.n; b++) s.plane[c].w[b] = fin_plane(x.blocks.p[b].rec[c]); }
    return write_sbwt(path, true, s, err) ? 0 : -2;
}

// k, node and k-mer counts and C array of an SBWT file; planes are compared with `expect` when given (build-fmin -i: the SBWT is a
// pure function of the unitigs and k, so the file can only confirm what the builder computes)
int fin_read_sbwt_file(const std::string& path, bool with_variant, int64_t& k, int64_t& n_nodes, int64_t& n_kmers, const fin_index* expect, std::string& err) {
    SbwtFile s;
    if (!read_sbwt(path, with_variant, s, err)) return -2;
    k = s.k; n_nodes = s.n_nodes; n_kmers = s.n_kmers;
    if (expect) {
        if ((uint64_t)s.n_nodes != expect->n_nodes || (uint64_t)s.n_kmers != expect->n_kmers || (uint32_t)s.k != expect->k) {
            err = path + ": SBWT has k=" + std::to_string(s.k) + ", " + std::to_string(s.n_nodes) + " nodes, " + std::to_string(s.n_kmers) + " k-mers; the unitigs give k=" +
                  std::to_string(expect->k) + ", " + std::to_string(expect->n_nodes) + " nodes, " + std::to_string(expect->n_kmers) + " k-mers: not the SBWT of these unitigs";
            return -1;
        }
        for (int c = 0; c < 4; c++)
            for (uint64_t b = 0; b < expect->blocks.n; b++)
                if (s.plane[c].w[b] != fin_plane(expect->blocks.p[b].rec[c])) { err = path + ": bit-planes differ from the SBWT of these unitigs"; return -1; }
    }
    return 0;
}

// LCS file (--lcs, build_fmin.hh:373-383) against the LCS the builder computed
int fin_check_lcs_file(const std::string& path, const fin_index& x, std::string& err) {
    IntVec lcs;
    if (!read_file_vec(path, 0, lcs, err)) return -2;
    if (lcs.bits / lcs.width != x.n_nodes) { err = path + ": " + std::to_string(lcs.bits / lcs.width) + " LCS entries for " + std::to_string(x.n_nodes) + " nodes"; return -1; }
    for (uint64_t i = 0; i < x.n_nodes; i++)
        if (iv_get(lcs, i) != (uint64_t)fin_host_lcs(x.blocks.p, x.lcs8_or_null(), (int64_t)i)) { err = path + ": LCS[" + std::to_string(i) + "] differs from the LCS of this SBWT"; return -1; }
    return 0;
}

int fin_load_reference_layout(const std::string& prefix, fin_index& x, std::string& err) {
    SbwtFile s;
    if (!read_sbwt(prefix + ".sbwt", false, s, err)) return -2;
    IntVec goff, fmin, concat, ends, ustart, lcs;
    if (!read_file_vec(prefix + ".LCS.sdsl", 0, lcs, err) || !read_file_vec(prefix + ".FBV.sdsl", 1, fmin, err) || !read_file_vec(prefix + ".O.sdsl", 0, goff, err) ||
        !read_file_vec(prefix + ".packed_unitigs.sdsl", 2, concat, err) || !read_file_vec(prefix + ".unitig_endpoints.sdsl", 0, ends, err) ||
        !read_file_vec(prefix + ".Ustart.sdsl", 1, ustart, err))
        return -2;
    const uint64_t n = (uint64_t)s.n_nodes;
    if (s.k < 2 || s.k > FIN_MAX_K) { err = "k = " + std::to_string(s.k) + " is outside [2, " + std::to_string(FIN_MAX_K) + "]"; return -5; }
    if (n >= 0xFFFFFFC0ull) { err = "index too large for this build: n_nodes >= 2^32"; return -5; }
    if (lcs.bits / lcs.width != n || fmin.bits != n || ustart.bits != n) { err = prefix + ": LCS / fmin / Ustart lengths do not match the SBWT's " + std::to_string(n) + " nodes"; return -2; }
    const uint64_t total_len = concat.bits / 2, nu = ends.bits / ends.width, nf = goff.bits / goff.width;
    if (total_len >= 0xFFFFFFF0ull) { err = "index too large for this build: total unitig length >= 2^32"; return -5; }
    if (nu == 0 || iv_get(ends, nu - 1) != total_len) { err = prefix + ": unitig endpoints do not end at the length of the packed unitigs"; return -2; }
    const uint64_t nblk = (n + 63) / 64;
    if (!x.blocks.resize(nblk)) { err = "out of memory (blocks)"; return -4; }
    x.k = (uint32_t)s.k; x.n_nodes = n; x.n_kmers = (uint64_t)s.n_kmers; x.n_unitigs = nu; x.total_len = total_len; x.n_fmin = nf;
    x.lcs8.clear();
    if (s.k > FIN_FAST_K) x.lcs8.assign(n, 0);
    FinNodeBlock* B = x.blocks.p;
    x.blkinfo.assign(nblk + 2, FinBlockInfo{0, 0, 0, 0, 0, 0});
    uint64_t tot[4] = {0, 0, 0, 0}, nfm = 0, nus = 0;
    for (uint64_t b = 0; b < nblk; b++) {
        for (int c = 0; c < 4; c++) {
            const uint64_t w = s.plane[c].w[b];
            B[b].rec[c].plane_lo = (uint32_t)w; B[b].rec[c].plane_hi = (uint32_t)(w >> 32);
            tot[c] += (uint64_t)__builtin_popcountll(w);
        }
        const uint64_t fm = fmin.w[b], um = ustart.w[b];
        const uint64_t lim = n - b * 64 < 64 ? n - b * 64 : 64;
        for (uint64_t j = 0; j < lim; j++) {
            const uint64_t v = iv_get(lcs, b * 64 + j);
            if (v > 254 || (v > FIN_LCS_MASK && x.lcs8.empty())) { err = prefix + ": LCS value " + std::to_string(v) + " is not below k"; return -2; }
            B[b].node[j] = (uint8_t)(v < FIN_LCS_MASK ? v : FIN_LCS_MASK) | (uint8_t)(((um >> j) & 1) ? FIN_USTART_BIT : 0);
            if (!x.lcs8.empty()) x.lcs8[b * 64 + j] = (uint8_t)v;
        }
        x.blkinfo[b].fmin_rank = (uint32_t)nfm; x.blkinfo[b].ustart_rank = (uint32_t)nus;
        x.blkinfo[b].fmin_mask_lo = (uint32_t)fm; x.blkinfo[b].fmin_mask_hi = (uint32_t)(fm >> 32);
        x.blkinfo[b].ustart_mask_lo = (uint32_t)um; x.blkinfo[b].ustart_mask_hi = (uint32_t)(um >> 32);
        nfm += (uint64_t)__builtin_popcountll(fm); nus += (uint64_t)__builtin_popcountll(um);
    }
    x.blkinfo[nblk].fmin_rank = x.blkinfo[nblk + 1].fmin_rank = (uint32_t)nfm;
    x.blkinfo[nblk].ustart_rank = x.blkinfo[nblk + 1].ustart_rank = (uint32_t)nus;
    if (nfm != nf) { err = prefix + ": " + std::to_string(nfm) + " finimizer marks but " + std::to_string(nf) + " offsets"; return -2; }
    if (nus > nu) { err = prefix + ": " + std::to_string(nus) + " unitig-start marks but " + std::to_string(nu) + " unitigs"; return -2; }   // (fewer: unitigs sharing a first k-mer)
    // C array and per-block rank bases (C[c] + rank_c(64 b)); C[0] = 1 (node 0 is the root)
    x.C[0] = 1;
    for (int c = 0; c < 3; c++) x.C[c + 1] = x.C[c] + tot[c];
    if (x.C[3] + tot[3] != n) { err = prefix + ".sbwt: edge marks (" + std::to_string(tot[0] + tot[1] + tot[2] + tot[3]) + ") + 1 != nodes (" + std::to_string(n) + ")"; return -2; }
    for (int c = 0; c < 4; c++)
        if ((uint64_t)s.C[c] != x.C[c]) { err = prefix + ".sbwt: stored C array differs from the bit-planes' counts"; return -2; }
    {
        uint64_t run[4] = {x.C[0], x.C[1], x.C[2], x.C[3]};
        for (uint64_t b = 0; b < nblk; b++)
            for (int c = 0; c < 4; c++) { B[b].rec[c].base = (uint32_t)run[c]; run[c] += (uint64_t)__builtin_popcountll(fin_plane(B[b].rec[c])); }
    }
    x.goff.assign(nf + 8, 0);
    for (uint64_t i = 0; i < nf; i++) {
        const uint64_t v = iv_get(goff, i);
        if (v >= total_len) { err = prefix + ".O.sdsl: offset beyond the packed unitigs"; return -2; }
        x.goff[i] = (uint32_t)v;
    }
    x.ends.assign(nu + 1 + 8, 0xFFFFFFFFu);
    x.ends[0] = 0;
    for (uint64_t u = 0; u < nu; u++) {
        const uint64_t e = iv_get(ends, u);
        if (e > total_len || e < x.ends[u]) { err = prefix + ".unitig_endpoints.sdsl: endpoints not increasing"; return -2; }
        if (e - x.ends[u] < (uint64_t)x.k) { err = prefix + ".unitig_endpoints.sdsl: a unitig shorter than k"; return -2; }
        x.ends[u + 1] = (uint32_t)e;
    }
    x.concat.assign(total_len / 16 + 8, 0);
    for (uint64_t wi = 0; wi < (total_len + 31) / 32; wi++) { x.concat[2 * wi] = (uint32_t)concat.w[wi]; x.concat[2 * wi + 1] = (uint32_t)(concat.w[wi] >> 32); }
    fin_finish_sampling(x);
    fin_finish_thermometer(x, -1);
    return 0;
}
