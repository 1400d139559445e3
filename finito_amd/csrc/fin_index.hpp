// fin_index.hpp -- host-side index object of the product (no oracle code here).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <atomic>

#include "fin_format.h"

#define FIN_N_OPTIONS 32

struct FinBlockArray {   // 128-B aligned array of FinNodeBlock
    FinNodeBlock* p = nullptr;
    uint64_t n = 0;
    FinBlockArray() {}
    FinBlockArray(const FinBlockArray&) = delete;
    FinBlockArray& operator=(const FinBlockArray&) = delete;
    ~FinBlockArray() { free(p); }
    bool resize(uint64_t nb) {
        free(p); p = nullptr; n = 0;
        if (nb == 0) return true;
        void* q = nullptr;
        if (posix_memalign(&q, 128, nb * sizeof(FinNodeBlock)) != 0) return false;
        memset(q, 0, nb * sizeof(FinNodeBlock));
        p = (FinNodeBlock*)q; n = nb;
        return true;
    }
};

struct fin_index {
    uint32_t k = 0;
    uint64_t n_nodes = 0, n_kmers = 0, n_unitigs = 0, total_len = 0, n_fmin = 0;
    uint64_t C[4] = {0, 0, 0, 0};
    uint32_t samp_shift = 0;
    uint32_t lcs_t0 = 0;
    FinBlockArray blocks;
    std::vector<FinBlockInfo> blkinfo;
    std::vector<uint32_t> goff, ends, samp, concat;   // ends = ends_p layout (see fin_format.h)
    std::vector<uint8_t> lcs8;                        // k > FIN_FAST_K: the exact LCS array (the node bytes hold min(LCS, 127)); else empty
    const uint8_t* lcs8_or_null() const { return lcs8.empty() ? nullptr : lcs8.data(); }

    // HBM replicas ("loads into HBM once"): one per device the index was sent to; replicas[0] is the default
    struct Replica {
        int device = -1;
        void* d_blocks = nullptr; void* d_blkinfo = nullptr; void* d_goff = nullptr; void* d_ends = nullptr; void* d_samp = nullptr; void* d_concat = nullptr; void* d_ptab = nullptr; void* d_jtab = nullptr; void* d_pos = nullptr; void* d_filt = nullptr; void* d_lcs8 = nullptr; void* d_safe = nullptr; void* d_kt3 = nullptr; void* d_ktx = nullptr; void* d_rcwin = nullptr; void* d_cbf = nullptr; void* d_fbf = nullptr;
        bool lean = false;             // uploaded with option "lean_tables": no prefix table, no anchor table; probes through the directional string filter
        uint64_t table_bytes = 0;      // HBM the upload allocated beyond the index arrays: prefix / jump / anchor / k-mer tables, filters, bitmaps
        bool anchors_built = false;    // the upload ran fin_launch_build_anchors: dev.safe (null = every place safe) and n_unsafe are known
        uint64_t n_rc_pairs = 0;       // k-mers of the text whose reverse complement is in the index too (counted with the anchor pass)
        uint64_t n_unsafe = 0;         // k-mer positions of the text that are not the place the reference reports for their k-mer
        uint64_t n_unverified = 0;     // ... of them, places whose k-mer's answer is a place that does not spell it (the exact side table of the k-mer table holds these)
        double anchors_ms = 0;         // time that build took
        FinDevIndex dev{};
    };
    std::vector<Replica> replicas;
    const Replica* replica_on(int device) const {
        for (const auto& r : replicas) if (r.device == device) return &r;
        return nullptr;
    }

    // per-index overrides of the process-wide options (fin_index_set_option; fin_capi.cpp knows the ids)
    std::atomic<int64_t> opt_val[FIN_N_OPTIONS]; std::atomic<bool> opt_set[FIN_N_OPTIONS];

    // device batches the host pipeline (fin_search_batch*) keeps between calls: their HBM buffers only grow, so a caller that
    // streams chunks of similar size through the pipeline allocates once.  (device, batch) pairs; freed with the index.
    mutable std::mutex pool_mu;
    mutable std::vector<std::pair<int, struct fin_batch*>> batch_pool;

    fin_index() { for (int i = 0; i < FIN_N_OPTIONS; i++) { opt_val[i].store(0); opt_set[i].store(false); } }
    fin_index(const fin_index&) = delete;
    fin_index& operator=(const fin_index&) = delete;
};

// ---- host-side SBWT primitives over the block layout (used by the builder only; search runs on the GPU) ----
struct FinIval { int64_t first, second; };

static inline uint64_t fin_mask_below(unsigned o) { return o == 0 ? 0ull : (~0ull >> (64 - o)); }        // bits [0,o)
static inline uint64_t fin_mask_incl(unsigned o) { return ~0ull >> (63 - o); }                            // bits [0,o]

// update_sbwt_interval for one character; formula restated by the reference at common.hh:26-36
static inline FinIval fin_host_extend(const FinNodeBlock* B, int c, FinIval I) {
    if (I.first < 0) return I;
    const FinNodeBlock& bl = B[I.first >> 6];
    const FinNodeBlock& br = B[I.second >> 6];
    int64_t l = (int64_t)bl.rec[c].base + __builtin_popcountll(fin_plane(bl.rec[c]) & fin_mask_below((unsigned)(I.first & 63)));
    int64_t r = (int64_t)br.rec[c].base + __builtin_popcountll(fin_plane(br.rec[c]) & fin_mask_incl((unsigned)(I.second & 63))) - 1;
    if (l > r) return FinIval{-1, -1};
    return FinIval{l, r};
}
// (lcs8: the exact LCS array of an index with k > 128, else null -- the node bytes hold min(LCS, 127))
static inline unsigned fin_host_lcs(const FinNodeBlock* B, const uint8_t* lcs8, int64_t i) { return lcs8 ? lcs8[i] : (B[i >> 6].node[i & 63] & FIN_LCS_MASK); }
// drop_first_char, common.hh:38-48
static inline FinIval fin_host_drop(const FinNodeBlock* B, const uint8_t* lcs8, int64_t n_nodes, int64_t new_len, FinIval I) {
    if (I.first < 0) return I;
    if (new_len <= 0) return FinIval{0, n_nodes - 1};
    while (I.first > 0 && (int64_t)fin_host_lcs(B, lcs8, I.first) >= new_len) I.first--;
    while (I.second < n_nodes - 1 && (int64_t)fin_host_lcs(B, lcs8, I.second + 1) >= new_len) I.second++;
    return I;
}

int fin_build_index(const char* bases, const uint64_t* offsets, uint64_t n_unitigs, int k, int n_threads,
                    fin_index& out, std::string& err);
// the same index built on a HIP device (fin_build_gpu.hip; k <= 32): bit-identical to fin_build_index's; phase_ms: 8 doubles or null
int fin_build_index_gpu(const char* bases, const uint64_t* offsets, uint64_t n_unitigs, int k, int device, fin_index& out, std::string& err, double* phase_ms);
void fin_finish_sampling(fin_index& x);
void fin_finish_thermometer(fin_index& x, int forced_t0);
int fin_save_index(const fin_index& x, const std::string& prefix, std::string& err);
int fin_load_index(const std::string& prefix, fin_index& x, std::string& err);

// the reference's on-disk layout: seven files <prefix>.{O,FBV,packed_unitigs,unitig_endpoints,Ustart,LCS}.sdsl + <prefix>.sbwt (fin_sdsl.cpp)
int fin_save_reference_layout(const fin_index& x, const std::string& prefix, std::string& err);
int fin_load_reference_layout(const std::string& prefix, fin_index& x, std::string& err);
int fin_save_sbwt_file(const fin_index& x, const std::string& path, std::string& err);
int fin_read_sbwt_file(const std::string& path, bool with_variant, int64_t& k, int64_t& n_nodes, int64_t& n_kmers, const fin_index* expect, std::string& err);
int fin_check_lcs_file(const std::string& path, const fin_index& x, std::string& err);
