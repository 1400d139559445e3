// fin_capi.cpp -- implementation of the C ABI declared in include/finito_amd.h.
// Host orchestration only: index lifetime, HBM replica, batches, launches.  No CPU search path exists here:
// without a HIP device every search entry point fails with FIN_ENODEV.
#include <hip/hip_runtime_api.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/finito_amd.h"
#include "fin_index.hpp"
#include "fin_kernels.h"

static void set_err(char* err, size_t errlen, const std::string& msg) {
    if (err && errlen) { snprintf(err, errlen, "%s", msg.c_str()); }
}
#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            set_err(err, errlen, std::string(#call) + ": " + hipGetErrorString(e_));                   \
            return FIN_ENODEV;                                                                         \
        }                                                                                              \
    } while (0)

// ---- page-locked staging for callers that hand over pageable buffers ---------------------------------------------------
// A DMA engine cannot read or write pageable memory; the runtime then stages such copies itself, single-threaded.  The pipeline
// below stages through its own page-locked buffers instead: every worker thread copies its sub-batch in and out with the CPU
// while the other workers' sub-batches are on the GPU or on the link.  Buffers are kept in a small process-wide pool.
namespace {
struct StagePool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> free_list;
    size_t held = 0;
    void* get(size_t bytes, size_t* cap) {
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].second >= bytes && (best == free_list.size() || free_list[i].second < free_list[best].second)) best = i;
            if (best != free_list.size()) {
                void* p = free_list[best].first; *cap = free_list[best].second;
                held -= *cap; free_list.erase(free_list.begin() + (long)best);
                return p;
            }
        }
        void* p = nullptr;
        const size_t want = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        *cap = want;
        return p;
    }
    void put(void* p, size_t cap) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            if (free_list.size() < 8 && held + cap <= (4ull << 30)) { free_list.push_back({p, cap}); held += cap; return; }
        }
        (void)hipHostFree(p);
    }
    void release_all() {
        std::lock_guard<std::mutex> g(mu);
        for (auto& f : free_list) (void)hipHostFree(f.first);
        free_list.clear(); held = 0;
    }
};
StagePool g_stage;
bool is_page_locked(const void* p) {
    if (!p) return true;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
}  // namespace

extern "C" {

const char* fin_version(void) { return "finito-amd 0.1 (gfx950)"; }

// ---- options: process-wide defaults (fin_set_option) and per-index overrides (fin_index_set_option) -------------------------------
// A handle that has its own value of an option uses it, every other handle follows the process-wide value.  The per-handle form is the
// one to use when handles are shared between threads: it touches nothing but its index.
enum : int { O_lds_deque_limit, O_kernel, O_probe_prepass, O_ptab_t, O_jtab_t, O_write_gaps, O_overlap_prefill, O_filt_f, O_seed_anchors, O_kmer_table,
              O_defer_strand, O_fast_path, O_cbf_m, O_lean_tables, O_text_anchors, O_epoch_budget_mult, O_epoch_budget_add, O_max_batch_kmers, O_pipeline_kmers, O_pipeline_depth, O_stage_pageable, O_debug_ovf_cap, O_debug_pp_seg, O_lean_walk, O_COUNT };
static_assert(O_COUNT <= FIN_N_OPTIONS, "fin_index::opt_val has room for every option");
struct OptDef { const char* name; int64_t def, lo, hi; };
static const OptDef OPTS[O_COUNT] = {
    {"lds_deque_limit", 16, 1, 16},
    {"kernel", 4, 0, 4},                 // (1 is not a kernel: rejected below)
    {"probe_prepass", 1, 0, 1},          // kernel 3: probe all strands in a separate light kernel first
    {"ptab_t", -1, -1, 15},              // prefix table depth for replicas uploaded from now on: -1 = by index size, 0 = none
    {"jtab_t", -1, -1, 14},              // jump table depth, likewise
    {"write_gaps", 1, 0, 1},             // kernel 4 with seeds: the output is not prefilled, the walk kernel writes absent slots with the pairs
    {"overlap_prefill", 1, 0, 1},        // kernel 4: output prefill on a side stream beside ingest and pre-pass
    {"filt_f", -1, -1, 16},              // depth of the pre-pass's absence filter (-1: by index size, 0: none)
    {"seed_anchors", 1, 0, 1},           // anchor table built at upload, first anchors of a strand found through it (kernel 4)
    {"kmer_table", 1, 0, 1},             // k <= 31: hash table text k-mer -> SBWT node, built with the anchor table, asked instead of whole-k-mer look-ups
    {"defer_strand", 1, 0, 1},           // kernel 4: a read's second strand only where the first left slots open (indexes without reverse-complement pairs and unsafe places)
    {"fast_path", 1, 0, 2},              // kernel 4: the pair pre-pass finishes the reads that lie in one unitig with a few substitutions by itself (fin_prepass.hip); 1 (default) = for k <= 63; 2 = for every k (measured at k = 127: it finishes 55 % of the reads and the step is 11 % slower -- the walk kernel asks the k-mer table above 63 too since round 5, which is what made such a step fast)
    {"cbf_m", -1, -1, 32},               // string length of the two string filters built at upload (-1: min(k, 20), less for k < 29; 0: none -- then no lean tables either)
    {"lean_tables", 2, 0, 3},            // at upload: no prefix table and no anchor table -- the compact k-mer table, the two string filters and the jump table only: probes ask the directional string filter, a string that occurs is followed by a look-up of the whole k-mer.  2 (default since round 5) = for k <= 63: 21 bytes per indexed base at 250 Mbp for any such k; 3 = at every k <= 255 (the walk kernel asks the k-mer table above 63 too: 21 instead of 70 bytes per base at k = 127 -- and a step of 17.0 instead of 10.3 ms: without seeds by node every anchor is a whole-k-mer look-up of ceil(k/32) + 5 epochs); 1 = k <= 31 only (round 4's default: k = 63 then keeps round 3's tables, 68 bytes per base, 6 % faster on iid reads and 32 % slower on a repeat-rich genome); 0 = round 3's tables
    {"text_anchors", 1, 0, 1},           // kernels 3 / 4 re-anchor behind sequencing errors by text comparison, at places the upload found safe
    {"epoch_budget_mult", 64, 0, 64},    // epoch budget of a read: mult * length + add (debug: shrink to force the overflow path)
    {"epoch_budget_add", 4096, 1, 1 << 20},
    {"max_batch_kmers", 1ll << 30, 1, 1ll << 31},
    {"pipeline_kmers", 1ll << 26, 1, 1ll << 31},   // sub-batch size of fin_search_batch's copy/compute pipeline
    {"pipeline_depth", 3, 1, 8},                   // sub-batches in flight per device
    {"stage_pageable", 1, 0, 1},                   // stage pageable caller buffers through page-locked memory inside the pipeline
    {"debug_ovf_cap", 0, 0, 1ll << 31},            // tests: capacity of the overflow list as the kernels see it (0: what the batch allocated)
    {"debug_pp_seg", 0, 0, 1024},                  // tests: reads per block of the pair pre-pass (0: by batch size; else a multiple of 256 up to 1024)
    {"lean_walk", 1, 0, 1},                        // kernel 4, lean tables: the walk kernel's lean instantiations (no prefix-table states: fewer registers); k <= 31: two k-mer-table look-ups per epoch in a run of absent k-mers (round 5)
};
static std::atomic<int64_t> g_opt[O_COUNT];
static const bool g_opt_init = [] { for (int i = 0; i < O_COUNT; i++) g_opt[i].store(OPTS[i].def); return true; }();
static inline int64_t optv(const fin_index* x, int id) { return x && x->opt_set[id].load(std::memory_order_acquire) ? x->opt_val[id].load(std::memory_order_relaxed) : g_opt[id].load(std::memory_order_relaxed); }
static int opt_find(const char* name, int64_t value) {
    if (!name) return -1;
    for (int i = 0; i < O_COUNT; i++)
        if (!strcmp(name, OPTS[i].name)) return (value < OPTS[i].lo || value > OPTS[i].hi || (i == O_kernel && value == 1)) ? -1 : i;
    return -1;
}

int fin_set_option(const char* name, int64_t value) {
    (void)g_opt_init;
    const int id = opt_find(name, value);
    if (id < 0) return FIN_EINVAL;
    g_opt[id].store(value);
    if (id == O_stage_pageable && !value) g_stage.release_all();
    return FIN_OK;
}

int fin_index_set_option(fin_index* idx, const char* name, int64_t value) {
    if (!idx) return FIN_EINVAL;
    const int id = opt_find(name, value);
    if (id < 0) return FIN_EINVAL;
    idx->opt_val[id].store(value, std::memory_order_relaxed); idx->opt_set[id].store(true, std::memory_order_release);   // (a reader that sees the flag sees the value)
    return FIN_OK;
}

int fin_index_clear_option(fin_index* idx, const char* name) {
    if (!idx || !name) return FIN_EINVAL;
    for (int i = 0; i < O_COUNT; i++) if (!strcmp(name, OPTS[i].name)) { idx->opt_set[i].store(false); return FIN_OK; }
    return FIN_EINVAL;
}

int fin_host_threads(void) {
    int n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64]; long long per = 0;
        if (fscanf(f, "%63s %lld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) {
            long long q = atoll(a) / per;
            if (q >= 1 && q < n) n = (int)q;
        }
        fclose(f);
    }
    int cap = 64;
    if (const char* e = getenv("FINITO_THREADS")) { int v = atoi(e); if (v > 0) cap = v; }
    if (n > cap) n = cap;
    return n < 1 ? 1 : n;
}

int fin_index_build(const char* unitig_bases, const uint64_t* unitig_offsets, uint64_t n_unitigs, int k, int n_threads,
                    fin_index** out, char* err, size_t errlen) {
    if (!unitig_bases || !unitig_offsets || !out) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    fin_index* x = new (std::nothrow) fin_index();
    if (!x) { set_err(err, errlen, "out of memory"); return FIN_ENOMEM; }
    std::string msg;
    int rc;
    try {
        rc = fin_build_index(unitig_bases, unitig_offsets, n_unitigs, k, n_threads > 0 ? n_threads : fin_host_threads(), *x, msg);
    } catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory while building the index"; }
    if (rc != 0) { delete x; set_err(err, errlen, msg); return rc; }
    *out = x;
    return FIN_OK;
}

int fin_index_build_device(const char* unitig_bases, const uint64_t* unitig_offsets, uint64_t n_unitigs, int k, int device,
                           fin_index** out, double* phase_ms, char* err, size_t errlen) {
    if (!unitig_bases || !unitig_offsets || !out) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_err(err, errlen, "no HIP device available (fin_index_build is the host builder)"); return FIN_ENODEV; }
    if (device < 0 || device >= ndev) { set_err(err, errlen, "device ordinal out of range"); return FIN_EINVAL; }
    fin_index* x = new (std::nothrow) fin_index();
    if (!x) { set_err(err, errlen, "out of memory"); return FIN_ENOMEM; }
    std::string msg;
    int rc;
    try { rc = fin_build_index_gpu(unitig_bases, unitig_offsets, n_unitigs, k, device, *x, msg, phase_ms); }
    catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory while building the index"; }
    if (rc != 0) { delete x; set_err(err, errlen, msg); return rc == -3 ? FIN_ENODEV : rc; }
    *out = x;
    return FIN_OK;
}

int fin_index_save(const fin_index* idx, const char* prefix, char* err, size_t errlen) {
    if (!idx || !prefix) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    std::string msg;
    int rc = fin_save_index(*idx, prefix, msg);
    if (rc) set_err(err, errlen, msg);
    return rc;
}

static bool file_exists(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; }

int fin_index_load(const char* prefix, fin_index** out, char* err, size_t errlen) {
    if (!prefix || !out) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    fin_index* x = new (std::nothrow) fin_index();
    if (!x) { set_err(err, errlen, "out of memory"); return FIN_ENOMEM; }
    std::string msg;
    int rc;
    try {
        // the container file if there is one, else the reference's own seven files (an index built by the reference's tools)
        const std::string p(prefix);
        if (!file_exists(p + ".finamd") && file_exists(p + ".sbwt")) rc = fin_load_reference_layout(p, *x, msg);
        else rc = fin_load_index(p, *x, msg);
    } catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory"; }
    if (rc) { delete x; set_err(err, errlen, msg); return rc; }
    *out = x;
    return FIN_OK;
}

int fin_index_load_reference_layout(const char* prefix, fin_index** out, char* err, size_t errlen) {
    if (!prefix || !out) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    fin_index* x = new (std::nothrow) fin_index();
    if (!x) { set_err(err, errlen, "out of memory"); return FIN_ENOMEM; }
    std::string msg;
    int rc;
    try { rc = fin_load_reference_layout(prefix, *x, msg); } catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory"; }
    if (rc) { delete x; set_err(err, errlen, msg); return rc; }
    *out = x;
    return FIN_OK;
}

int fin_index_save_reference_layout(const fin_index* idx, const char* prefix, char* err, size_t errlen) {
    if (!idx || !prefix) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    std::string msg;
    int rc;
    try { rc = fin_save_reference_layout(*idx, prefix, msg); } catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory"; }
    if (rc) set_err(err, errlen, msg);
    return rc;
}

int fin_index_save_sbwt(const fin_index* idx, const char* path, char* err, size_t errlen) {
    if (!idx || !path) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    std::string msg;
    int rc;
    try { rc = fin_save_sbwt_file(*idx, path, msg); } catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory"; }
    if (rc) set_err(err, errlen, msg);
    return rc;
}

int fin_sbwt_file_info(const char* path, int64_t* k, int64_t* n_nodes, int64_t* n_kmers, char* err, size_t errlen) {
    if (!path) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    std::string msg; int64_t a = 0, b = 0, c = 0;
    int rc;
    try { rc = fin_read_sbwt_file(path, true, a, b, c, nullptr, msg); } catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory"; }
    if (rc) { set_err(err, errlen, msg); return rc; }
    if (k) *k = a;
    if (n_nodes) *n_nodes = b;
    if (n_kmers) *n_kmers = c;
    return FIN_OK;
}

int fin_index_check_against_files(const fin_index* idx, const char* sbwt_path, const char* lcs_path, char* err, size_t errlen) {
    if (!idx) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    std::string msg; int rc = 0;
    try {
        int64_t a, b, c;
        if (sbwt_path && *sbwt_path) rc = fin_read_sbwt_file(sbwt_path, true, a, b, c, idx, msg);
        if (rc == 0 && lcs_path && *lcs_path) rc = fin_check_lcs_file(lcs_path, *idx, msg);
    } catch (const std::bad_alloc&) { rc = FIN_ENOMEM; msg = "out of memory"; }
    if (rc) set_err(err, errlen, msg);
    return rc;
}

static void free_replica(fin_index::Replica& r) {
    if (r.device >= 0) {
        (void)hipSetDevice(r.device);
        (void)hipFree(r.d_blocks); (void)hipFree(r.d_blkinfo); (void)hipFree(r.d_goff); (void)hipFree(r.d_ends); (void)hipFree(r.d_samp); (void)hipFree(r.d_concat); (void)hipFree(r.d_ptab); (void)hipFree(r.d_jtab); (void)hipFree(r.d_pos); (void)hipFree(r.d_filt); (void)hipFree(r.d_lcs8); (void)hipFree(r.d_safe); (void)hipFree(r.d_kt3); (void)hipFree(r.d_ktx); (void)hipFree(r.d_rcwin); (void)hipFree(r.d_cbf); (void)hipFree(r.d_fbf);
        r = fin_index::Replica();
    }
}
static void free_device(fin_index* x) {
    for (auto& r : x->replicas) free_replica(r);
    x->replicas.clear();
}

void fin_batch_free(fin_batch* b);
void fin_index_free(fin_index* idx) {
    if (!idx) return;
    for (auto& pb : idx->batch_pool) fin_batch_free(pb.second);
    idx->batch_pool.clear();
    free_device(idx);
    delete idx;
}

int64_t fin_index_k(const fin_index* x) { return x ? (int64_t)x->k : -1; }
int64_t fin_index_n_nodes(const fin_index* x) { return x ? (int64_t)x->n_nodes : -1; }
int64_t fin_index_n_kmers(const fin_index* x) { return x ? (int64_t)x->n_kmers : -1; }
int64_t fin_index_n_unitigs(const fin_index* x) { return x ? (int64_t)x->n_unitigs : -1; }
int64_t fin_index_n_finimizers(const fin_index* x) { return x ? (int64_t)x->n_fmin : -1; }
int64_t fin_index_total_len(const fin_index* x) { return x ? (int64_t)x->total_len : -1; }
int fin_index_prefix_table_depth(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r ? (int)r->dev.ptab_t : -1;
}

// 1: the number of distinct k-mers equals the number of k-mer positions in the unitigs (every k-mer has exactly one place).  Informative
// only: what the kernels may take from the text is decided per k-mer at upload (fin_index_unsafe_places).
int fin_index_is_disjoint(const fin_index* x) {
    if (!x || !x->n_unitigs) return 0;
    uint64_t places = 0;
    for (uint64_t u = 0; u < x->n_unitigs; u++) {
        const uint64_t len = (uint64_t)x->ends[u + 1] - x->ends[u];
        if (len >= x->k) places += len - x->k + 1;
    }
    return x->n_kmers == places ? 1 : 0;
}

int64_t fin_index_rc_pairs(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r && r->anchors_built && r->n_rc_pairs != ~0ull ? (int64_t)r->n_rc_pairs : -1;
}
int64_t fin_index_unsafe_places(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r && r->anchors_built ? (int64_t)r->n_unsafe : -1;
}
double fin_index_anchor_build_ms(const fin_index* x, int device) {
    const fin_index::Replica* r = x ? x->replica_on(device) : nullptr;
    return r && r->anchors_built ? r->anchors_ms : -1.0;
}

// k-mer places whose k-mer has an unverified answer (the reference reports a place that does not spell it): what the exact side table holds; -1: no replica / no anchor pass
int64_t fin_index_unverified_kmers(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r && r->anchors_built ? (int64_t)r->n_unverified : -1;
}
int64_t fin_index_kmer_table_bytes(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r ? (r->d_kt3 ? (int64_t)(32ull * r->dev.kt3_buckets + (r->d_ktx ? (32ull << r->dev.ktx_log2) : 0)) : 0) : -1;
}

// the device of the handle's first (default) replica, -1: none
int fin_index_first_device(const fin_index* x) { return (x && !x->replicas.empty()) ? x->replicas[0].device : -1; }
// HBM of the replica on `device` beyond the index arrays (fin_index_size_in_bytes): every table, filter and bitmap the upload built
int64_t fin_index_replica_table_bytes(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r ? (int64_t)r->table_bytes : -1;
}
int64_t fin_index_string_filter_bytes(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r ? (r->d_cbf ? (int64_t)(16ull << r->dev.cbf_log2) : 0) : -1;
}

int64_t fin_index_seed_table_bytes(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r ? (r->d_pos ? (int64_t)x->n_nodes * (int64_t)sizeof(FinSeedEntry) : 0) : -1;
}

int fin_index_filter_depth(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r ? (int)r->dev.filt_f : -1;
}

int fin_index_jump_table_depth(const fin_index* x, int device) {
    if (!x) return -1;
    const fin_index::Replica* r = x->replica_on(device);
    return r ? (int)r->dev.jtab_t : -1;
}

int64_t fin_index_size_in_bytes(const fin_index* x) {
    if (!x) return -1;
    return (int64_t)(x->blocks.n * sizeof(FinNodeBlock) + x->blkinfo.size() * sizeof(FinBlockInfo) + 4 * (x->goff.size() + x->ends.size() + x->samp.size() + x->concat.size()));
}

int64_t fin_index_export_size(const fin_index* x, int what) {
    if (!x) return -1;
    const int64_t nw = (int64_t)((x->n_nodes + 63) / 64);
    switch (what) {
        case FIN_X_C: return 32;
        case FIN_X_PLANE_A: case FIN_X_PLANE_A + 1: case FIN_X_PLANE_A + 2: case FIN_X_PLANE_A + 3: return nw * 8;
        case FIN_X_LCS: return (int64_t)x->n_nodes;
        case FIN_X_FMIN: case FIN_X_USTART: return nw * 8;
        case FIN_X_GOFF: return (int64_t)x->n_fmin * 8;
        case FIN_X_ENDS: return (int64_t)x->n_unitigs * 8;
        case FIN_X_CONCAT: return (int64_t)x->total_len;
        default: return -1;
    }
}

int fin_index_export(const fin_index* x, int what, void* out, uint64_t out_bytes, char* err, size_t errlen) {
    int64_t need = fin_index_export_size(x, what);
    if (need < 0 || !out) { set_err(err, errlen, "bad export selector or null buffer"); return FIN_EINVAL; }
    if (out_bytes < (uint64_t)need) { set_err(err, errlen, "export buffer too small"); return FIN_EINVAL; }
    const FinNodeBlock* B = x->blocks.p;
    const uint64_t nb = x->blocks.n;
    switch (what) {
        case FIN_X_C: for (int c = 0; c < 4; c++) ((int64_t*)out)[c] = (int64_t)x->C[c]; break;
        case FIN_X_PLANE_A: case FIN_X_PLANE_A + 1: case FIN_X_PLANE_A + 2: case FIN_X_PLANE_A + 3:
            for (uint64_t b = 0; b < nb; b++) ((uint64_t*)out)[b] = fin_plane(B[b].rec[what - FIN_X_PLANE_A]);
            break;
        case FIN_X_LCS: for (uint64_t i = 0; i < x->n_nodes; i++) ((uint8_t*)out)[i] = (uint8_t)fin_host_lcs(B, x->lcs8_or_null(), (int64_t)i); break;
        case FIN_X_FMIN: for (uint64_t b = 0; b < nb; b++) ((uint64_t*)out)[b] = x->blkinfo[b].fmin_mask_lo | ((uint64_t)x->blkinfo[b].fmin_mask_hi << 32); break;
        case FIN_X_USTART:
            for (uint64_t b = 0; b < nb; b++) {
                uint64_t w = 0;   // from the per-node flags; the builder keeps ustart_mask identical (checked in tests)
                for (int j = 0; j < 64; j++) if (B[b].node[j] & FIN_USTART_BIT) w |= 1ull << j;
                ((uint64_t*)out)[b] = w == (x->blkinfo[b].ustart_mask_lo | ((uint64_t)x->blkinfo[b].ustart_mask_hi << 32)) ? w : ~0ull;
            }
            break;
        case FIN_X_GOFF: for (uint64_t i = 0; i < x->n_fmin; i++) ((int64_t*)out)[i] = (int64_t)x->goff[i]; break;
        case FIN_X_ENDS: for (uint64_t i = 0; i < x->n_unitigs; i++) ((int64_t*)out)[i] = (int64_t)x->ends[i + 1]; break;
        case FIN_X_CONCAT: for (uint64_t i = 0; i < x->total_len; i++) ((uint8_t*)out)[i] = (uint8_t)((x->concat[i >> 4] >> (2 * (i & 15))) & 3u); break;
    }
    return FIN_OK;
}

int fin_index_to_device(fin_index* x, int device, char* err, size_t errlen) {
    if (!x) { set_err(err, errlen, "null index"); return FIN_EINVAL; }
    if (x->replica_on(device)) return FIN_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_err(err, errlen, "no HIP device available (this path has no CPU fallback)"); return FIN_ENODEV; }
    if (device < 0 || device >= ndev) { set_err(err, errlen, "device ordinal out of range"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(device));
    fin_index::Replica r;
    r.device = device;
    auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 4);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e;
    if ((e = up(&r.d_blocks, x->blocks.p, x->blocks.n * sizeof(FinNodeBlock))) != hipSuccess ||
        (e = up(&r.d_blkinfo, x->blkinfo.data(), x->blkinfo.size() * sizeof(FinBlockInfo))) != hipSuccess ||
        (e = up(&r.d_goff, x->goff.data(), x->goff.size() * 4)) != hipSuccess ||
        (e = up(&r.d_ends, x->ends.data(), x->ends.size() * 4)) != hipSuccess ||
        (e = up(&r.d_samp, x->samp.data(), x->samp.size() * 4)) != hipSuccess ||
        (e = up(&r.d_concat, x->concat.data(), x->concat.size() * 4)) != hipSuccess) {
        free_replica(r);
        set_err(err, errlen, std::string("uploading the index: ") + hipGetErrorString(e));
        return FIN_ENODEV;
    }
    FinDevIndex& d = r.dev;
    d.blocks = (const FinNodeBlock*)r.d_blocks; d.blkinfo = (const FinBlockInfo*)r.d_blkinfo; d.goff = (const uint32_t*)r.d_goff;
    d.ends = (const uint32_t*)r.d_ends; d.samp = (const uint32_t*)r.d_samp; d.concat = (const uint32_t*)r.d_concat;
    d.n_nodes = (uint32_t)x->n_nodes; d.n_unitigs = (uint32_t)x->n_unitigs; d.total_len = (uint32_t)x->total_len; d.k = x->k;
    d.samp_shift = x->samp_shift; d.n_samp = (uint32_t)x->samp.size();
    for (int c = 0; c < 4; c++) d.C[c] = (uint32_t)x->C[c];
    d.C[4] = (uint32_t)x->n_nodes;
    d.lcs_t0 = x->lcs_t0;
    d.lcs8 = nullptr;
    if (!x->lcs8.empty()) {   // k > 128: the exact LCS array for the plain kernel (the node bytes hold min(LCS, 127))
        if ((e = up(&r.d_lcs8, x->lcs8.data(), x->lcs8.size())) != hipSuccess) {
            free_replica(r); set_err(err, errlen, std::string("uploading the LCS array: ") + hipGetErrorString(e)); return FIN_ENODEV;
        }
        d.lcs8 = (const uint8_t*)r.d_lcs8;
    }
    d.budget_mult = 64; d.budget_add = 4096;
    d.text_anchors = 0; d.defer_ok = 0; d.rcwin = nullptr;   // (set per run: fin_batch_run)
    {   // prefix table for the kernel's probe mode: depth T with 4^T <= 16 * n_nodes (most random T-mers are then already absent --
        // one table line settles the probe -- and T+4 bases almost never occur), at most 15 (8 GiB of the 288) and at most k;
        // filled on the device from the blocks just uploaded
        // (option cbf_m 0 = no string filters: lean tables need the directional one -- round 3's tables are built instead)
        const bool lean_req = optv(x, O_lean_tables) && optv(x, O_cbf_m) != 0 && optv(x, O_ptab_t) < 0 && x->k <= (optv(x, O_lean_tables) >= 3 ? 255u : optv(x, O_lean_tables) >= 2 ? 63u : 31u) && optv(x, O_kmer_table) && optv(x, O_seed_anchors) && optv(x, O_text_anchors) &&
                              x->total_len < FIN_POS_DUMMY && x->n_unitigs < FIN_POS_UNVERIFIED;
        int T = lean_req ? 0 : (int)optv(x, O_ptab_t);
        if (T < 0) { T = 0; while (T < 15 && T < (int)x->k && (1ull << (2 * (T + 1))) <= 16ull * x->n_nodes) T++; }
        if (T > (int)x->k) T = (int)x->k;
        d.ptab_t = 0; d.ptab = nullptr;
        if (T > 0) {
            if ((e = hipMalloc(&r.d_ptab, (sizeof(FinPrefixIval) << (2 * T)) + 16)) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("prefix table: ") + hipGetErrorString(e)); return FIN_ENODEV;
            }
            d.ptab_t = (uint32_t)T; d.ptab = (const FinPrefixIval*)r.d_ptab;
            const int rc = fin_launch_build_ptab(&d, r.d_ptab, T, nullptr);
            if (rc != 0 || (e = hipDeviceSynchronize()) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("prefix table kernel: ") + hipGetErrorString(rc ? (hipError_t)rc : e)); return FIN_ENODEV;
            }
        }
    }
    {   // jump table for (re)started streaming searches: depth J with 4^J <= n_nodes / 3 (nearly every J-base string of the indexed text
        // then occurs at least twice, which is what a jump needs), below k, at most 14
        int J = (int)optv(x, O_jtab_t);
        if (J < 0) { J = 0; while (J < 14 && 3ull * (1ull << (2 * (J + 1))) <= x->n_nodes) J++; }
        if (J >= (int)x->k) J = (int)x->k - 1;
        d.jtab_t = 0; d.jtab = nullptr;
        if (J > 0) {
            if ((e = hipMalloc(&r.d_jtab, (sizeof(FinPrefixIval) << (2 * J)) + 16)) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("jump table: ") + hipGetErrorString(e)); return FIN_ENODEV;
            }
            const int rc = fin_launch_build_ptab(&d, r.d_jtab, J, nullptr);
            if (rc != 0 || (e = hipDeviceSynchronize()) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("jump table kernel: ") + hipGetErrorString(rc ? (hipError_t)rc : e)); return FIN_ENODEV;
            }
            d.jtab_t = (uint32_t)J; d.jtab = (const FinPrefixIval*)r.d_jtab;
        }
    }
    {   // absence filter of the pre-pass.  By default only where its bit set stays in the L2 (4^F bits <= 2 MB: F <= 12) AND is sparse enough
        // to say something (4^F >= the text's length): small indexes.  Measured on the 250 Mbp index (F = 14, 32 MB, Infinity-Cache
        // resident): the pre-pass got SLOWER (5.2 -> 6.4 ms) -- a look-up that misses the L2 costs a request like a prefix-table line
        // does, wherever it is served from, and the filter asks more often than it saves (CHANGELOG.md 5.5).
        int F = (int)optv(x, O_filt_f);
        if (F < 0) { F = 8; while (F < 12 && (1ull << (2 * F)) < x->total_len) F++; if ((1ull << (2 * F)) < x->total_len) F = 0; }
        if (F >= (int)x->k) F = (int)x->k - 1;
        d.filt = nullptr; d.filt_f = 0;
        if (F >= 4) {
            if ((e = hipMalloc(&r.d_filt, ((1ull << (2 * F)) / 32 + 8) * 4)) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("absence filter: ") + hipGetErrorString(e)); return FIN_ENODEV;
            }
            const int rc = fin_launch_build_filter(&d, (uint32_t*)r.d_filt, F, nullptr);
            if (rc != 0 || (e = hipDeviceSynchronize()) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("absence filter kernel: ") + hipGetErrorString(rc ? (hipError_t)rc : e)); return FIN_ENODEV;
            }
            d.filt = (const uint32_t*)r.d_filt; d.filt_f = (uint32_t)F;
        }
    }
    d.pos = nullptr; d.safe = nullptr; d.kt3 = nullptr; d.kt3_buckets = 0; d.ktx = nullptr; d.ktx_log2 = 0;
    const bool up_seeds = optv(x, O_seed_anchors) != 0, up_text = optv(x, O_text_anchors) != 0;
    if ((up_seeds || up_text) && x->total_len < FIN_POS_DUMMY && x->n_unitigs < FIN_POS_UNVERIFIED && x->k < 256) {
        // anchor table (FinDevIndex::pos) and safe-place bitmap (FinDevIndex::safe): the unitig text streamed through the plain search on
        // the device (fin_kernel_b.hip) -- per node the reference's answer for its k-mer, per text position whether the k-mer there is
        // reported there.  16 bytes per node + 1 bit per base; the bitmap is dropped when every place is safe (disjoint unitigs).
        void* d_tmp = nullptr;
        // k-mer table (with the anchor pass): the COMPACT table of round 5 -- 8-byte slots {answer, tag}, four to a 32-byte bucket, 55 % full: room for
        // the text's k-mer positions / 0.55, whatever the text's size (bucket numbers are 32-bit: up to 2^34 slots; the answers are text offsets below 2^32)
        uint32_t kt3_buckets = 0;
        if (optv(x, O_kmer_table) && up_seeds) {   // (any k <= 255 since the walk kernel folds a long k-mer's words into the hash as its chunks arrive, fin_kernel_w.hip W_KF0B)
            uint64_t places = 0;
            for (uint64_t u = 0; u < x->n_unitigs; u++) { const uint64_t len = (uint64_t)x->ends[u + 1] - x->ends[u]; if (len >= x->k) places += len - x->k + 1; }
            const uint64_t nb = (places * 100 / FIN_KT3_LOAD_PCT + FIN_KT3_SLOTS - 1) / FIN_KT3_SLOTS + 16;
            if (nb < 0xFFFFFFF0ull) {
                kt3_buckets = (uint32_t)nb;
                if ((e = hipMalloc(&r.d_kt3, 32ull * kt3_buckets)) != hipSuccess) {
                    free_replica(r); set_err(err, errlen, std::string("k-mer table: ") + hipGetErrorString(e)); return FIN_ENODEV;
                }
            }
        }
        r.lean = optv(x, O_lean_tables) && optv(x, O_cbf_m) != 0 && optv(x, O_ptab_t) < 0 && optv(x, O_text_anchors) && kt3_buckets != 0 && x->k <= (optv(x, O_lean_tables) >= 3 ? 255u : optv(x, O_lean_tables) >= 2 ? 63u : 31u);   // (a k-mer table: the conditions under which no prefix table was built above)
        if ((!r.lean && (e = hipMalloc(&r.d_pos, ((size_t)x->n_nodes + 1) * sizeof(FinSeedEntry))) != hipSuccess) ||
            (e = hipMalloc(&r.d_safe, fin_anchor_safe_words(x->total_len) * 8)) != hipSuccess ||
            (e = hipMalloc(&d_tmp, fin_anchor_tmp_bytes(x->total_len))) != hipSuccess) {
            (void)hipFree(d_tmp); free_replica(r); set_err(err, errlen, std::string("anchor table: ") + hipGetErrorString(e)); return FIN_ENODEV;
        }
        hipEvent_t t0 = nullptr, t1 = nullptr;
        (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
        (void)hipEventRecord(t0, nullptr);
        uint64_t n_unver = 0;
        int rc = fin_launch_build_anchors(&d, (FinSeedEntry*)r.d_pos, r.d_safe, r.d_kt3, kt3_buckets, d_tmp, &r.n_unsafe, nullptr, &n_unver);
        uint32_t ktx_lg = 0;
        if (rc == 0 && r.d_kt3 && n_unver) {
            // the exact side table of the k-mers whose answer is unverified (FinDevIndex::ktx), from the list the pass left in d_tmp: twice their number of
            // 32-byte slots; k-mers beyond the list's room are not in it (their reads are decided by kernel 3)
            const uint32_t n_list = (uint32_t)std::min<uint64_t>(n_unver, fin_anchor_ulist_cap(x->total_len));
            ktx_lg = 6;
            while ((1ull << ktx_lg) < 2ull * n_list) ktx_lg++;
            if ((e = hipMalloc(&r.d_ktx, (32ull << ktx_lg) + 32)) != hipSuccess) { (void)hipFree(d_tmp); free_replica(r); set_err(err, errlen, std::string("k-mer side table: ") + hipGetErrorString(e)); return FIN_ENODEV; }
            rc = fin_launch_build_ktx(fin_anchor_ulist(d_tmp, x->total_len), n_list, r.d_ktx, ktx_lg, nullptr);
            r.n_unverified = n_unver;
        }
        (void)hipEventRecord(t1, nullptr);
        e = hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, t0, t1); r.anchors_ms = ms;
        (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
        (void)hipFree(d_tmp);
        if (rc != 0 || e != hipSuccess) {
            free_replica(r); set_err(err, errlen, std::string("anchor table kernel: ") + (rc == (int)hipErrorOutOfMemory ? "the k-mer table filled up" : hipGetErrorString(rc ? (hipError_t)rc : e))); return FIN_ENODEV;
        }
        r.anchors_built = true;
        {   // reverse-complement pairs (for the deferred second strand): needs the prefix table of this replica
            void* d8 = nullptr;
            if (hipMalloc(&d8, 16) == hipSuccess && hipMalloc(&r.d_rcwin, fin_rcwin_bytes(x->total_len)) == hipSuccess) {
                if (fin_launch_count_rc_pairs(&d, d8, &r.n_rc_pairs, r.d_rcwin, nullptr) != 0) r.n_rc_pairs = ~0ull;   // (unknown: no deferral)
            } else r.n_rc_pairs = ~0ull;
            (void)hipFree(d8);
            if (r.n_rc_pairs == 0 || r.n_rc_pairs == ~0ull) { (void)hipFree(r.d_rcwin); r.d_rcwin = nullptr; }   // (none: the walk kernel need not look)
        }
        if (r.n_unsafe == 0) { (void)hipFree(r.d_safe); r.d_safe = nullptr; }
        if (!up_seeds) { (void)hipFree(r.d_pos); r.d_pos = nullptr; }
        d.pos = (const FinSeedEntry*)r.d_pos; d.safe = (const unsigned long long*)r.d_safe;
        d.kt3 = (const FinKt3Bucket*)r.d_kt3; d.kt3_buckets = kt3_buckets;
        d.ktx = (const FinKtxSlot*)r.d_ktx; d.ktx_log2 = ktx_lg;
    }
    d.cbf = nullptr; d.cbf_log2 = 0; d.cbf_m = 0; d.fast_path = 0; d.fbf = nullptr;
    if (d.kt3) {
        // canonical string filter (FinDevIndex::cbf): strings of m bases, 16 bits of filter per text position, a power of two of 16-byte blocks
        // (250 Mbp: 2^25 blocks, 512 MiB -- a sixteenth of the prefix table it takes the error-bridging probes from).  m = 20, or less for short
        // k: a string settles k-m+1 k-mer ends and the fast path asks three across a disagreeing base, so 3 (k-m+1) >= k must hold
        int m = (int)optv(x, O_cbf_m);
        if (m < 0) { m = (int)x->k + 1 - ((int)x->k + 2) / 3; if (m > 20) m = 20; if (m < 1) m = 1; }
        if (m > (int)x->k) m = (int)x->k;
        if (m >= 1) {
            uint32_t lg = 4;
            while ((8ull << lg) < x->total_len && lg < 31) lg++;
            if ((e = hipMalloc(&r.d_cbf, 16ull << lg)) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("canonical string filter: ") + hipGetErrorString(e)); return FIN_ENODEV;
            }
            if (r.lean && (e = hipMalloc(&r.d_fbf, 16ull << lg)) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("directional string filter: ") + hipGetErrorString(e)); return FIN_ENODEV;
            }
            const int rc = fin_launch_build_cbf(&d, r.d_cbf, r.d_fbf, lg, (uint32_t)m, nullptr);
            if (rc != 0 || (e = hipDeviceSynchronize()) != hipSuccess) {
                free_replica(r); set_err(err, errlen, std::string("canonical string filter kernel: ") + hipGetErrorString(rc ? (hipError_t)rc : e)); return FIN_ENODEV;
            }
            d.cbf = (const FinCbfBlock*)r.d_cbf; d.cbf_log2 = lg; d.cbf_m = (uint32_t)m; d.fbf = (const FinCbfBlock*)r.d_fbf;
        }
    }
    r.table_bytes = (r.d_ptab ? (sizeof(FinPrefixIval) << (2 * d.ptab_t)) : 0) + (r.d_jtab ? (sizeof(FinPrefixIval) << (2 * d.jtab_t)) : 0) +
                    (r.d_filt ? ((1ull << (2 * d.filt_f)) / 8) : 0) + (r.d_pos ? (x->n_nodes + 1) * sizeof(FinSeedEntry) : 0) +
                    (r.d_safe ? fin_anchor_safe_words(x->total_len) * 8 : 0) + (r.d_kt3 ? 32ull * d.kt3_buckets : 0) + (r.d_ktx ? (32ull << d.ktx_log2) : 0) +
                    (r.d_rcwin ? fin_rcwin_bytes(x->total_len) : 0) + (r.d_cbf ? (16ull << d.cbf_log2) : 0) + (r.d_fbf ? (16ull << d.cbf_log2) : 0) + (r.d_lcs8 ? x->lcs8.size() : 0);
    x->replicas.push_back(r);
    return FIN_OK;
}

// diagnostic (tests): {g, u} of every entry of the anchor table of the replica on `device`, 2 * n_nodes u32; FIN_EINVAL when that replica has none
int fin_index_debug_seed_table(const fin_index* x, int device, uint32_t* out, char* err, size_t errlen) {
    const fin_index::Replica* r = x ? x->replica_on(device) : nullptr;
    if (!r || !r->d_pos || !out) { set_err(err, errlen, "no seed table on that device"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpy2D(out, 8, r->d_pos, sizeof(FinSeedEntry), 8, (size_t)x->n_nodes, hipMemcpyDeviceToHost));   // {g, u} of every entry
    return FIN_OK;
}

// diagnostic (tests): what the compact k-mer table of the replica on `device` claims about n k-mers given as their two key words (2-bit codes, first base in the low
// bits; k1 = 0 for k <= 32): out[2 i] = g, out[2 i + 1] = flags (fin_kt3_query_kernel).  FIN_EINVAL: that replica has no k-mer table
int fin_index_debug_kmer_table(const fin_index* x, int device, const uint64_t* k0, const uint64_t* k1, uint64_t n, uint32_t* out, char* err, size_t errlen) {
    const fin_index::Replica* r = x ? x->replica_on(device) : nullptr;
    if (!r || !r->d_kt3 || !k0 || !k1 || !out || n > 0x7FFFFFFFull || x->k > 64) { set_err(err, errlen, "no k-mer table on that device (or k > 64: the query takes two key words)"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(device));
    void* d0 = nullptr; void* d1 = nullptr; void* dout = nullptr;
    int rc = FIN_OK;
    if (hipMalloc(&d0, n * 8 + 8) != hipSuccess || hipMalloc(&d1, n * 8 + 8) != hipSuccess || hipMalloc(&dout, n * 8 + 8) != hipSuccess) rc = FIN_ENOMEM;
    if (rc == FIN_OK && (hipMemcpy(d0, k0, n * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d1, k1, n * 8, hipMemcpyHostToDevice) != hipSuccess)) rc = FIN_ENODEV;
    if (rc == FIN_OK && fin_launch_kt3_query(&r->dev, (const uint64_t*)d0, (const uint64_t*)d1, (uint32_t)n, dout, nullptr) != 0) rc = FIN_ENODEV;
    if (rc == FIN_OK && hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FIN_ENODEV;
    (void)hipFree(d0); (void)hipFree(d1); (void)hipFree(dout);
    if (rc != FIN_OK) set_err(err, errlen, "k-mer table query failed");
    return rc;
}

// ---- batches --------------------------------------------------------------------------------------------------
struct fin_batch {
    const fin_index* idx = nullptr;
    FinDevIndex dev{};
    int device = -1;
    uint64_t n_reads = 0, n_kmers = 0, n_base_strands = 0, total_bases = 0;
    void* d_bases_alloc = nullptr;   // 16 guard bytes in front: the reverse strand reads 16-byte chunks ending at a read's end
    uint8_t* d_bases = nullptr; void* d_offs = nullptr; void* d_out_offs = nullptr; void* d_out = nullptr; void* d_desc = nullptr; void* d_desc2 = nullptr; void* d_packed = nullptr; void* d_pass = nullptr; void* d_seed = nullptr; size_t cap_seed = 0; uint32_t grid_blocks2 = 0, grid_blocks3 = 0, grid_blocks_probe = 0;
    uint32_t* d_work = nullptr; uint32_t grid_blocks = 0;
    void* d_text = nullptr; void* d_last_bits = nullptr; void* d_blk_sum = nullptr; void* d_blk_off = nullptr; uint64_t* d_total = nullptr;   // output text made on the device
    size_t cap_text = 0, cap_last_bits = 0, cap_blk_sum = 0, cap_blk_off = 0;
    // text modes (fin_batch_text_mode): the fast path's per-read records, and what the most recent run did with them
    void* d_frec = nullptr; void* d_seg = nullptr; size_t cap_frec = 0, cap_seg = 0;
    int text_reads_state = 0;   // since the last load: 0 not looked at, 1 every read has a k-mer, 2 one has none (fin_batch_format_text refuses)
    uint32_t n_seg = 0; bool seg_table = false;   // the text kernels' segments (fin_text.hip): a table only when a read has more than fin_text3_seg_pairs() pairs
    int text_mode = 0; bool last_frec = false, last_text_only = false, count_from_text = false;
    // records (fin_batch_records): the dense stream of the pairs of the reads the fast path did not finish
    void* d_cstream = nullptr; size_t cap_cstream = 0; uint64_t rec_stream_pairs = 0; bool rec_ready = false, rec_passthrough = false;
    uint64_t text_bytes = 0;
    // kernel 4: the queue counters of the most recent finished run, copied to page-locked memory behind every run: the next run launches only
    // as many stream / walk rounds as that one needed, plus one (fin_launch_search_v4's `rounds`)
    uint32_t* h_ctr = nullptr; hipEvent_t ev_ctr = nullptr; bool ctr_pending = false; uint32_t rounds_hint = 0;
    void* d_ws = nullptr; size_t cap_ws = 0; uint64_t q_slots = 0; uint32_t* d_ctr = nullptr; uint32_t grid_blocks_stream = 0, grid_blocks_walk = 0;   // kernel 4: item queues, counters
    uint32_t* d_ovf_list = nullptr; uint32_t* d_ovf_count = nullptr; uint64_t* d_ovf_scratch = nullptr;
    unsigned long long* d_count = nullptr;
    uint32_t ovf_blocks = 0;
    // HIP events of every run, recorded on the stream the step was launched on: 0 step begins (before the ingest kernel),
    // 1 reads packed + output prefilled, 2 probe pre-pass done (kernel 3), 3 search kernel done, 4 step ends (overflow redo done)
    struct RunEvents { hipEvent_t e[5]; };
    std::vector<RunEvents> runs;
    uint64_t n_chunks = 0;
    int last_strands = FIN_MERGED; uint32_t last_kernel = 0, last_no_prefill = 0, last_ovf_cap = 0xFFFFFFFFu;
    int ovf_state = 0; uint32_t last_ovf = 0;   // the most recent run's overflow list: 0 not looked at yet, 1 within its capacity, 2 overran (results withheld)
    size_t cap_pass = 0, cap_bases = 0, cap_desc = 0, cap_desc2 = 0, cap_offs = 0, cap_out_offs = 0, cap_out = 0, cap_ovf_list = 0, cap_packed = 0;
    hipStream_t own_stream = nullptr;    // uploads, the pack kernel and (for the library's own pipeline) the search run here
    hipStream_t last_stream = nullptr;   // stream of the most recent fin_batch_run
    // kernel 4: the output prefill runs on a side stream beside ingest and the probe pre-pass (forked from and joined to the launch stream by events)
    hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool ran = false;
    std::vector<uint64_t> h_offs, h_out_offs; std::vector<FinReadDesc> h_desc, h_desc2;   // host staging of the per-read tables
};

void fin_batch_free(fin_batch* b) {
    if (!b) return;
    if (b->device >= 0) (void)hipSetDevice(b->device);
    (void)hipFree(b->d_bases_alloc); (void)hipFree(b->d_desc); (void)hipFree(b->d_desc2); (void)hipFree(b->d_packed); (void)hipFree(b->d_pass); (void)hipFree(b->d_seed); (void)hipFree(b->d_work); (void)hipFree(b->d_offs); (void)hipFree(b->d_out_offs); (void)hipFree(b->d_out);
    (void)hipFree(b->d_ws); (void)hipFree(b->d_ctr);
    if (b->h_ctr) (void)hipHostFree(b->h_ctr);
    if (b->ev_ctr) (void)hipEventDestroy(b->ev_ctr);
    (void)hipFree(b->d_cstream); (void)hipFree(b->d_frec); (void)hipFree(b->d_seg); (void)hipFree(b->d_text); (void)hipFree(b->d_last_bits); (void)hipFree(b->d_blk_sum); (void)hipFree(b->d_blk_off); (void)hipFree(b->d_total);
    (void)hipFree(b->d_ovf_list); (void)hipFree(b->d_ovf_count); (void)hipFree(b->d_ovf_scratch); (void)hipFree(b->d_count);
    for (auto& r : b->runs) for (auto& e : r.e) (void)hipEventDestroy(e);
    if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
    if (b->side_stream) (void)hipStreamDestroy(b->side_stream);
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    delete b;
}

int fin_batch_create(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads, fin_batch** out,
                     char* err, size_t errlen) {
    if (!idx) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    if (idx->replicas.empty()) { set_err(err, errlen, "index is not resident on a device: call fin_index_to_device first (no CPU fallback)"); return FIN_ENODEV; }
    return fin_batch_create_on(idx, idx->replicas[0].device, bases, offsets, n_reads, out, err, errlen);
}

// (re)fill a batch with a read set: device buffers grow on demand and are kept, so a batch that is reloaded with read sets of
// similar size allocates once.  All copies and the pack kernel run on the batch's own stream; returns when they have finished.
// `first_base` points at base offsets[0] of the read set
static int batch_load(fin_batch* b, const char* first_base, const uint64_t* offsets, uint64_t n_reads, char* err, size_t errlen) {
    // (the kernel's work counter hands out read numbers in ranges of 64 and may overshoot by a range per wave)
    if (n_reads >= 0xFFF00000ull) { set_err(err, errlen, "more than 2^32-2^20 reads in one batch"); return FIN_ELIMIT; }
    const uint64_t base0 = offsets[0];
    const uint64_t k = b->idx->k;
    std::vector<uint64_t>& offs = b->h_offs; std::vector<uint64_t>& out_offs = b->h_out_offs;
    std::vector<FinReadDesc>& desc = b->h_desc; std::vector<FinReadDesc>& desc2 = b->h_desc2;
    offs.resize(n_reads + 1); out_offs.resize(n_reads + 1); desc.resize(n_reads + 1); desc2.resize(n_reads + 1);
    uint64_t n_chunks = 0;
    out_offs[0] = 0;
    for (uint64_t r = 0; r <= n_reads; r++) offs[r] = offsets[r] - base0;
    for (uint64_t r = 0; r < n_reads; r++) {
        uint64_t len = offs[r + 1] - offs[r];
        if (len >= 0x7FFFFFFFull) { set_err(err, errlen, "read longer than 2^31-1 bases"); return FIN_ELIMIT; }
        out_offs[r + 1] = out_offs[r] + (len >= k ? len - k + 1 : 0);
        desc[r] = FinReadDesc{offs[r], (uint32_t)len, (uint32_t)out_offs[r]};
        desc2[r] = FinReadDesc{n_chunks, (uint32_t)len, (uint32_t)out_offs[r]};   // off = first packed chunk of the read
        n_chunks += 2 * ((len + 31) / 32);
    }
    desc[n_reads] = FinReadDesc{offs[n_reads], 0, (uint32_t)out_offs[n_reads]};
    desc2[n_reads] = FinReadDesc{n_chunks, 0, (uint32_t)out_offs[n_reads]};
    if (out_offs[n_reads] >= 0xFFFFFFFFull) { set_err(err, errlen, "more than 2^32-1 k-mers in one batch: split the batch"); return FIN_ELIMIT; }
    if (offs[n_reads] >= 0xFFFFFFFFull) { set_err(err, errlen, "more than 2^32-1 bases in one batch: split the batch"); return FIN_ELIMIT; }
    b->n_reads = n_reads;
    b->total_bases = offs[n_reads];
    b->n_kmers = out_offs[n_reads];
    b->n_base_strands = 2 * b->total_bases;
    auto fail = [&](hipError_t e, const char* what) {
        set_err(err, errlen, std::string(what) + ": " + hipGetErrorString(e));
        b->n_reads = 0; b->n_kmers = 0; b->total_bases = 0; b->n_base_strands = 0;   // a failed load leaves an empty batch
        return FIN_ENODEV;
    };
    hipError_t e;
    if ((e = hipSetDevice(b->device)) != hipSuccess) return fail(e, "hipSetDevice");
    if (!b->own_stream && (e = hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    hipStream_t st = b->own_stream;
    // a reload must not overtake a search that is still running on another stream
    if (b->last_stream != st && b->ran) { if ((e = hipStreamSynchronize(b->last_stream)) != hipSuccess) return fail(e, "hipStreamSynchronize"); }
    auto grow = [&](void** p, size_t& cap, size_t bytes) -> hipError_t {
        if (bytes <= cap) return hipSuccess;
        if (*p) { (void)hipStreamSynchronize(st); (void)hipFree(*p); *p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8;   // a little head room: sub-batches of a stream of reads differ slightly in size
        hipError_t r = hipMalloc(p, want);
        if (r == hipSuccess) cap = want;
        return r;
    };
    if ((e = grow(&b->d_bases_alloc, b->cap_bases, b->total_bases + 128)) != hipSuccess) return fail(e, "hipMalloc(bases)");
    b->d_bases = (uint8_t*)b->d_bases_alloc + 64;   // guard bytes: 32-byte windows may overhang a read at either end
    if ((e = hipMemsetAsync(b->d_bases_alloc, 'N', 64, st)) != hipSuccess) return fail(e, "hipMemset(bases)");
    if ((e = hipMemsetAsync(b->d_bases + b->total_bases, 'N', 64, st)) != hipSuccess) return fail(e, "hipMemset(bases)");
    const size_t rd = (n_reads + 1);
    if ((e = grow(&b->d_desc, b->cap_desc, rd * sizeof(FinReadDesc))) != hipSuccess) return fail(e, "hipMalloc(descriptors)");
    if ((e = grow(&b->d_desc2, b->cap_desc2, rd * sizeof(FinReadDesc))) != hipSuccess) return fail(e, "hipMalloc(descriptors)");
    if ((e = grow(&b->d_offs, b->cap_offs, rd * 8)) != hipSuccess) return fail(e, "hipMalloc(offsets)");
    if ((e = grow(&b->d_out_offs, b->cap_out_offs, rd * 8)) != hipSuccess) return fail(e, "hipMalloc(out offsets)");
    if ((e = grow(&b->d_out, b->cap_out, b->n_kmers * 8 + 16)) != hipSuccess) return fail(e, "hipMalloc(output)");
    if ((e = grow((void**)&b->d_ovf_list, b->cap_ovf_list, rd * 4)) != hipSuccess) return fail(e, "hipMalloc(overflow list)");
    if ((e = grow(&b->d_packed, b->cap_packed, (n_chunks + 4) * 16)) != hipSuccess) return fail(e, "hipMalloc(packed reads)");
    if ((e = grow(&b->d_pass, b->cap_pass, (2 * rd + 4) * 4)) != hipSuccess) return fail(e, "hipMalloc(probe results)");
    if (!b->d_work && (e = hipMalloc((void**)&b->d_work, 4)) != hipSuccess) return fail(e, "hipMalloc");
    if (!b->grid_blocks2) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) != hipSuccess || cus <= 0) cus = 256;
        b->grid_blocks2 = (uint32_t)cus * (uint32_t)fin_v2_blocks_per_cu();
        b->grid_blocks3 = (uint32_t)cus * (uint32_t)fin_v3_blocks_per_cu();
        b->grid_blocks_probe = (uint32_t)cus * (uint32_t)fin_probe_blocks_per_cu();
        b->grid_blocks_stream = (uint32_t)cus * (uint32_t)fin_stream_blocks_per_cu();
        b->grid_blocks_walk = (uint32_t)cus * (uint32_t)fin_walk_blocks_per_cu();
    }
    b->q_slots = 0;
    if (optv(b->idx, O_kernel) == 4 && n_reads < 0x0FFFFFF0ull) {   // kernel 4's item queues and its list of reads for kernel 3 (read numbers travel in 28 bits of an item's first word)
        const uint32_t maxg = std::max(std::max(b->grid_blocks_probe, b->grid_blocks_stream), b->grid_blocks_walk);
        if ((e = grow(&b->d_ws, b->cap_ws, fin_v4_workspace_bytes((uint32_t)n_reads, maxg))) != hipSuccess) return fail(e, "hipMalloc(pipeline queues)");
        if (!b->d_ctr && (e = hipMalloc((void**)&b->d_ctr, fin_v4_counter_words() * 4)) != hipSuccess) return fail(e, "hipMalloc");
        b->q_slots = fin_v4_queue_slots((uint32_t)n_reads, maxg);
        // the overflow list under kernel 4: a read can be pushed once per strand by the stream kernel (deque overflow, epoch budget) and once
        // more each time kernel 3 redoes it from its list (twice at most: either strand's walk may give the read up) -- four entries per
        // read at the very most; k > 128: the walk kernel appends to it through reserved slots, which needs kernel 3's list capacity
        const uint64_t ovf_need = std::max<uint64_t>(4 * rd + 64, fin_v4_list_slots((uint32_t)n_reads, maxg));
        if ((e = grow((void**)&b->d_ovf_list, b->cap_ovf_list, ovf_need * 4)) != hipSuccess) return fail(e, "hipMalloc(overflow list)");
        const fin_index::Replica* rp = b->idx->replica_on(b->device);
        if (rp && (rp->dev.pos || rp->lean) && (e = grow(&b->d_seed, b->cap_seed, (2 * rd + 4) * 4)) != hipSuccess) return fail(e, "hipMalloc(seed nodes)");
    }
    if (!b->d_ovf_count && (e = hipMalloc((void**)&b->d_ovf_count, 4)) != hipSuccess) return fail(e, "hipMalloc");
    if (!b->d_count && (e = hipMalloc((void**)&b->d_count, 8)) != hipSuccess) return fail(e, "hipMalloc");
    {
        uint32_t want = (uint32_t)std::min<uint64_t>(64, (n_reads + 255) / 256);
        if (want == 0) want = 1;
        if (want > b->ovf_blocks) {
            if (b->d_ovf_scratch) { (void)hipStreamSynchronize(st); (void)hipFree(b->d_ovf_scratch); b->d_ovf_scratch = nullptr; b->ovf_blocks = 0; }
            if ((e = hipMalloc((void**)&b->d_ovf_scratch, (size_t)want * 256 * fin_overflow_deque_cap() * 8)) != hipSuccess) return fail(e, "hipMalloc(overflow scratch)");
            b->ovf_blocks = want;
        }
    }
    if (b->total_bases && (e = hipMemcpyAsync(b->d_bases, first_base, b->total_bases, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "hipMemcpy(bases)");
    if ((e = hipMemcpyAsync(b->d_offs, offs.data(), rd * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "hipMemcpy(offsets)");
    if ((e = hipMemcpyAsync(b->d_out_offs, out_offs.data(), rd * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "hipMemcpy(out offsets)");
    if ((e = hipMemcpyAsync(b->d_desc, desc.data(), rd * sizeof(FinReadDesc), hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "hipMemcpy(descriptors)");
    if ((e = hipMemcpyAsync(b->d_desc2, desc2.data(), rd * sizeof(FinReadDesc), hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "hipMemcpy(descriptors)");
    // (the ASCII reads are what is resident: packing them to 2-bit chunks of both strands -- the reference's get_rc and base
    //  decoding, inside its timed region, search_fmin.hh:46-71 -- is the first kernel of every step, see fin_batch_run)
    b->n_chunks = n_chunks;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "upload");
    b->ran = false; b->last_stream = nullptr;
    b->rounds_hint = 0; b->ctr_pending = false;   // (new reads: nothing is known about the rounds they need)
    b->text_reads_state = 0;
    return FIN_OK;
}

int fin_batch_create_on(const fin_index* idx, int device, const char* bases, const uint64_t* offsets, uint64_t n_reads, fin_batch** out,
                        char* err, size_t errlen) {
    if (!idx || !offsets || !out || (n_reads && !bases)) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    const fin_index::Replica* rep = idx->replica_on(device);
    if (!rep) { set_err(err, errlen, "index is not resident on that device: call fin_index_to_device first (no CPU fallback)"); return FIN_ENODEV; }
    fin_batch* b = new (std::nothrow) fin_batch();
    if (!b) { set_err(err, errlen, "out of memory"); return FIN_ENOMEM; }
    b->idx = idx; b->dev = rep->dev; b->device = device;
    const int rc = batch_load(b, bases ? bases + offsets[0] : nullptr, offsets, n_reads, err, errlen);
    if (rc != FIN_OK) { fin_batch_free(b); return rc; }
    *out = b;
    return FIN_OK;
}

int fin_batch_reload(fin_batch* b, const char* bases, const uint64_t* offsets, uint64_t n_reads, char* err, size_t errlen) {
    if (!b || !offsets || (n_reads && !bases)) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    return batch_load(b, bases ? bases + offsets[0] : nullptr, offsets, n_reads, err, errlen);
}

int fin_batch_run(fin_batch* b, int strands, void* hip_stream, char* err, size_t errlen) {
    if (!b || (strands != FIN_FWD && strands != FIN_MERGED)) { set_err(err, errlen, "bad argument"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (b->runs.size() >= 1024) {   // a long-lived batch keeps the most recent launches only
        for (auto& e : b->runs.front().e) (void)hipEventDestroy(e);
        b->runs.erase(b->runs.begin());
    }
    fin_batch::RunEvents ev;
    for (auto& e : ev.e) HIPCHK(hipEventCreate(&e));
    b->runs.push_back(ev);
    b->last_strands = strands;
    b->last_stream = st; b->ran = true;
    // ---- one step of the hot path, everything on `st`: ingest (ASCII -> 2-bit chunks of both strands), output prefill, probe
    //      pre-pass, search kernel, overflow redo ----
    HIPCHK(hipEventRecord(ev.e[0], st));
    b->dev.budget_mult = (uint32_t)optv(b->idx, O_epoch_budget_mult); b->dev.budget_add = (uint32_t)optv(b->idx, O_epoch_budget_add);
    b->dev.ovf_cap = (uint32_t)std::min<uint64_t>(b->cap_ovf_list / 4, 0xFFFFFFFFull);
    if (const int64_t forced = optv(b->idx, O_debug_ovf_cap)) b->dev.ovf_cap = (uint32_t)std::min<int64_t>(forced, (int64_t)b->dev.ovf_cap);   // (tests: a tiny list)
    b->last_ovf_cap = b->dev.ovf_cap; b->ovf_state = 0; b->rec_ready = false;
    b->dev.pp_seg = (uint32_t)optv(b->idx, O_debug_pp_seg);
    b->dev.lean_walk = (uint32_t)optv(b->idx, O_lean_walk);
    {   // text re-anchoring needs the upload's verdict on every text place (the bitmap, or the knowledge that all are safe); the anchor table
        // is used when it exists, text re-anchoring is on (seeds are verified by its comparison) and the batch has room for seed nodes
        const fin_index::Replica* rep = b->idx->replica_on(b->device);
        b->dev.text_anchors = (optv(b->idx, O_text_anchors) && rep && rep->anchors_built) ? 1u : 0u;
        b->dev.safe = rep ? rep->dev.safe : nullptr;
        b->dev.pos = (optv(b->idx, O_seed_anchors) && b->dev.text_anchors && b->d_seed && rep) ? rep->dev.pos : nullptr;
        b->dev.filt = (optv(b->idx, O_filt_f) != 0 && rep) ? rep->dev.filt : nullptr;
        const bool lean = rep && rep->lean && optv(b->idx, O_seed_anchors) && b->dev.text_anchors && b->d_seed && optv(b->idx, O_kmer_table) && rep->dev.fbf;
        b->dev.fbf = lean ? rep->dev.fbf : nullptr;
        b->dev.kt3 = (optv(b->idx, O_kmer_table) && rep && (b->dev.pos || lean)) ? rep->dev.kt3 : nullptr;
        b->dev.ktx = (b->dev.kt3 && rep) ? rep->dev.ktx : nullptr; b->dev.ktx_log2 = rep ? rep->dev.ktx_log2 : 0;
        b->dev.cbf = (rep && b->dev.kt3) ? rep->dev.cbf : nullptr;
        b->dev.fast_path = (optv(b->idx, O_fast_path) && b->dev.cbf) ? 1u : 0u;
    }
    int rc = 0;
    hipEvent_t out_ready = nullptr;
    // k > 128: kernels 2 and 3 (and kernel 4's stream kernel) read 7-bit LCS values and cannot run; kernel 4 can when the index has a seed
    // table (its walk kernel needs no LCS, what it cannot finish goes to the plain kernel), else the plain kernel does everything
    const int opt_kernel = (int)optv(b->idx, O_kernel);
    const uint32_t lds_limit = (uint32_t)optv(b->idx, O_lds_deque_limit);
    int kern = opt_kernel;
    if (b->dev.k > FIN_FAST_K)
        kern = (opt_kernel == 4 && b->q_slots && fin_v4_writes_gaps(&b->dev, (const uint32_t*)b->d_seed)) ? 4 : 0;
    // kernel 4 on an index with a seed table: no prefill at all, the pipeline writes every slot once (option "write_gaps")
    const int no_prefill = (kern == 4 && b->q_slots && optv(b->idx, O_write_gaps) && fin_v4_writes_gaps(&b->dev, (const uint32_t*)b->d_seed)) ? 1 : 0;
    {   // the second strand of a read only where the first left slots open: kernel 4 writing every slot itself, both strands asked for, and an
        // the second strand of a read only where the first left slots open (CHANGELOG.md 4.14): exact on any index -- a first strand that reports
        // through the streaming search or a whole-k-mer look-up (a place that may not spell the k-mer: duplicated k-mers), or from a text
        // window with a k-mer whose reverse complement is in the index too (rcwin), has its sister searched in full ("tainted")
        const fin_index::Replica* rep = b->idx->replica_on(b->device);
        b->dev.defer_ok = (kern == 4 && no_prefill && strands == FIN_MERGED && optv(b->idx, O_defer_strand) && rep && rep->anchors_built &&
                           rep->n_rc_pairs != ~0ull && (b->dev.pos || b->dev.fbf)) ? 1u : 0u;
        b->dev.rcwin = (b->dev.defer_ok && rep->d_rcwin) ? (const uint8_t*)rep->d_rcwin : nullptr;
    }
    b->last_kernel = (uint32_t)((kern == 4 && !b->q_slots) ? 3 : kern); b->last_no_prefill = (uint32_t)no_prefill;
    if (!(b->dev.defer_ok && b->dev.kt3 && (b->dev.k <= 63 || optv(b->idx, O_fast_path) >= 2))) b->dev.fast_path = 0u;   // (the fast path rides on the pair pre-pass's k-mer-table looks)
    b->dev.frec = nullptr; b->dev.text_only = 0u; b->last_frec = false; b->last_text_only = false; b->count_from_text = false;
    if (b->text_mode && kern == 4 && b->q_slots && b->dev.fast_path && no_prefill && strands == FIN_MERGED && b->n_reads) {
        // text modes: a zeroed record per read, filled by the fast path for the reads it finishes (a record that stays zero: the read's pairs are in d_out)
        if (b->cap_frec < b->n_reads * sizeof(FinFastRec)) {
            if (b->d_frec) { HIPCHK(hipStreamSynchronize(st)); (void)hipFree(b->d_frec); b->d_frec = nullptr; b->cap_frec = 0; }
            const size_t want = (b->n_reads + b->n_reads / 8 + 16) * sizeof(FinFastRec);
            if (hipMalloc(&b->d_frec, want) != hipSuccess) { (void)hipGetLastError(); set_err(err, errlen, "out of device memory (fast-path records)"); return FIN_ENOMEM; }
            b->cap_frec = want;
        }
        HIPCHK(hipMemsetAsync(b->d_frec, 0, b->n_reads * sizeof(FinFastRec), st));
        b->dev.frec = (FinFastRec*)b->d_frec; b->dev.text_only = b->text_mode == 2 ? 1u : 0u;
        b->last_frec = true; b->last_text_only = b->text_mode == 2;
    }
    if (kern == 4 && b->q_slots && optv(b->idx, O_overlap_prefill) && !no_prefill) {
        // fork: (-1,-1) into every output slot on the side stream, beside the pack kernel and the pre-pass (which do not touch the output);
        // the pipeline's first writer waits for ev_join.  Everything stays inside the step's bracket e[0] .. e[4] on the launch stream.
        if (!b->side_stream) {
            HIPCHK(hipStreamCreateWithFlags(&b->side_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(b->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(b->side_stream, b->ev_fork, 0));
        HIPCHK(hipMemsetAsync(b->d_out, 0xFF, b->n_kmers * 8, b->side_stream));
        HIPCHK(hipEventRecord(b->ev_join, b->side_stream));
        out_ready = b->ev_join;
    }
    if (kern != 0)
        rc = fin_launch_pack_reads(b->d_bases, (const uint64_t*)b->d_offs, (const FinReadDesc*)b->d_desc2, b->d_packed, (uint32_t)b->n_reads, b->n_chunks, st);
    if (rc != 0) { set_err(err, errlen, std::string("pack kernel launch failed: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
    if (kern == 0)
        rc = fin_launch_search_v0(&b->dev, (const uint8_t*)b->d_bases, (const uint64_t*)b->d_offs, (const uint64_t*)b->d_out_offs,
                                  b->d_out, (uint32_t)b->n_reads, strands, lds_limit, b->d_ovf_list, b->d_ovf_count, b->d_ovf_scratch,
                                  b->ovf_blocks, st, ev.e[1], ev.e[3]);
    else if (kern == 4 && b->q_slots) {
        if (b->ctr_pending && b->ev_ctr && hipEventQuery(b->ev_ctr) == hipSuccess) {   // the counters of an earlier run of these reads have arrived
            uint32_t last = 0;
            for (uint32_t r = 0; r < fin_v4_max_rounds(); r++) if (b->h_ctr[6 + 4 * r] || b->h_ctr[7 + 4 * r]) last = r;
            b->rounds_hint = std::min<uint32_t>(fin_v4_max_rounds(), std::max<uint32_t>(2u, last + 2u));
            b->ctr_pending = false;
        } else (void)hipGetLastError();
        rc = fin_launch_search_v4(&b->dev, (const uint8_t*)b->d_bases, b->d_packed, (const FinReadDesc*)b->d_desc2, (const uint64_t*)b->d_offs,
                                  (const uint64_t*)b->d_out_offs, b->d_out, b->n_kmers, (uint32_t)b->n_reads, strands, lds_limit,
                                  b->d_ovf_list, b->d_ovf_count, b->d_ovf_scratch, b->ovf_blocks, (uint32_t*)b->d_pass, (uint32_t*)b->d_seed, b->d_ws, b->q_slots, b->d_ctr,
                                  b->grid_blocks_probe, b->grid_blocks_stream, b->grid_blocks_walk, b->grid_blocks3, st, ev.e[1], ev.e[3], ev.e[2], out_ready, no_prefill,
                                  b->rounds_hint ? b->rounds_hint : fin_v4_max_rounds());
    } else if (kern == 3 || kern == 4)   // (4 without queues: selected after this batch was loaded, or too many reads for the 29 bits a read number has in an item)
        rc = fin_launch_search_v3(&b->dev, (const uint8_t*)b->d_bases, b->d_packed, (const FinReadDesc*)b->d_desc2, (const uint64_t*)b->d_offs,
                                  (const uint64_t*)b->d_out_offs, b->d_out, b->n_kmers, (uint32_t)b->n_reads, strands,
                                  lds_limit, b->d_ovf_list, b->d_ovf_count, b->d_work, b->d_ovf_scratch, b->ovf_blocks,
                                  b->grid_blocks3, optv(b->idx, O_probe_prepass) ? (uint32_t*)b->d_pass : nullptr, b->grid_blocks_probe, st, ev.e[1], ev.e[3], ev.e[2]);
    else
        rc = fin_launch_search_v2(&b->dev, (const uint8_t*)b->d_bases, b->d_packed, (const FinReadDesc*)b->d_desc2, (const uint64_t*)b->d_offs,
                                  (const uint64_t*)b->d_out_offs, b->d_out, b->n_kmers, (uint32_t)b->n_reads, strands,
                                  lds_limit, b->d_ovf_list, b->d_ovf_count, b->d_work, b->d_ovf_scratch, b->ovf_blocks,
                                  b->grid_blocks2, st, ev.e[1], ev.e[3]);
    if (rc != 0) { set_err(err, errlen, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
    HIPCHK(hipEventRecord(ev.e[4], st));
    if (kern == 4 && b->q_slots && b->d_ctr && !b->ctr_pending) {   // behind the step: this run's queue counters for the next run's round count
        if (!b->h_ctr) { HIPCHK(hipHostMalloc((void**)&b->h_ctr, fin_v4_counter_words() * 4, hipHostMallocDefault)); HIPCHK(hipEventCreateWithFlags(&b->ev_ctr, hipEventDisableTiming)); }
        HIPCHK(hipMemcpyAsync(b->h_ctr, b->d_ctr, fin_v4_counter_words() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(b->ev_ctr, st));
        b->ctr_pending = true;
    }
    return FIN_OK;
}

uint64_t fin_batch_n_kmers(const fin_batch* b) { return b ? b->n_kmers : 0; }
uint64_t fin_batch_n_base_strands(const fin_batch* b) { return b ? (b->last_strands == FIN_MERGED ? b->n_base_strands : b->total_bases) : 0; }
void* fin_batch_device_pairs(const fin_batch* b) { return b ? b->d_out : nullptr; }

// diagnostic (tests of the text formatter): overwrite the batch's device pairs with n_kmers pairs from the host
int fin_batch_set_pairs(fin_batch* b, const int32_t* pairs, char* err, size_t errlen) {
    if (!b || !pairs) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    if (b->last_stream) HIPCHK(hipStreamSynchronize(b->last_stream));
    HIPCHK(hipMemcpy(b->d_out, pairs, (size_t)b->n_kmers * 8, hipMemcpyHostToDevice));
    b->last_frec = false; b->last_text_only = false; b->count_from_text = false;   // (the records of the last run say nothing about these pairs)
    return FIN_OK;
}

// (a push beyond the overflow list's capacity is dropped on the device -- fin_ovf_push -- and that read would keep a partial
//  result: the list is sized so that this cannot happen, and if it ever does NO entry point delivers results of that run -- pairs, ranges or text)
static int batch_overrun_check(fin_batch* b, hipStream_t st, char* err, size_t errlen) {
    if (!b->ran || !b->d_ovf_count) return FIN_OK;
    if (b->ovf_state == 0) {
        uint32_t ovf = 0;
        HIPCHK(hipMemcpyAsync(&ovf, b->d_ovf_count, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        b->last_ovf = ovf; b->ovf_state = ovf > b->last_ovf_cap ? 2 : 1;
    }
    if (b->ovf_state == 2) { set_err(err, errlen, "the overflow list of this batch overran (" + std::to_string(b->last_ovf) + " entries, room for " + std::to_string(b->last_ovf_cap) + "): results withheld"); return FIN_ELIMIT; }
    return FIN_OK;
}

int fin_batch_download(fin_batch* b, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen) {
    if (!b) { set_err(err, errlen, "null batch"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    // everything below is ordered behind the most recent search by running on its stream
    hipStream_t st = b->ran ? b->last_stream : b->own_stream;
    if (b->ran && b->last_text_only && (pairs_out || (n_positive && !b->count_from_text))) {
        set_err(err, errlen, "this batch ran in text-only mode (fin_batch_text_mode 2): the pairs of the reads its fast path finished were never written; "
                             "the number of found pairs is known once fin_batch_format_text has run");
        return FIN_EINVAL;
    }
    if (n_positive && b->n_kmers && !b->count_from_text) {
        int rc = fin_launch_count_positive(b->d_out, b->n_kmers, b->d_count, st);
        if (rc != 0) { set_err(err, errlen, std::string("count kernel: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
    }
    if (const int orc = batch_overrun_check(b, st, err, errlen)) return orc;
    if (pairs_out && b->n_kmers) HIPCHK(hipMemcpyAsync(pairs_out, b->d_out, b->n_kmers * 8, hipMemcpyDeviceToHost, st));
    unsigned long long c = 0;
    if (n_positive && b->n_kmers) HIPCHK(hipMemcpyAsync(&c, b->d_count, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_positive) *n_positive = c;
    return FIN_OK;
}

int fin_batch_download_range(fin_batch* b, uint64_t first_pair, uint64_t n_pairs, int32_t* pairs_out, char* err, size_t errlen) {
    if (!b || (n_pairs && !pairs_out)) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    if (first_pair > b->n_kmers || n_pairs > b->n_kmers - first_pair) { set_err(err, errlen, "pair range outside the batch"); return FIN_EINVAL; }
    if (b->ran && b->last_text_only) { set_err(err, errlen, "this batch ran in text-only mode (fin_batch_text_mode 2): its pairs are not materialised"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = b->ran ? b->last_stream : b->own_stream;
    if (const int orc = batch_overrun_check(b, st, err, errlen)) return orc;
    if (n_pairs) HIPCHK(hipMemcpyAsync(pairs_out, (const char*)b->d_out + first_pair * 8, n_pairs * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return FIN_OK;
}

// ---- the output text, made on the device --------------------------------------------------------------------------------------
static int batch_grow(fin_batch* b, void** p, size_t& cap, size_t bytes, hipStream_t st) {
    if (bytes <= cap) return FIN_OK;
    if (*p) { (void)hipStreamSynchronize(st); (void)hipFree(*p); *p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(p, want) != hipSuccess) return FIN_ENOMEM;
    cap = want;
    return FIN_OK;
}

int fin_batch_format_text(fin_batch* b, uint64_t* text_bytes, char* err, size_t errlen) {
    if (!b) { set_err(err, errlen, "null batch"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = b->ran ? b->last_stream : b->own_stream;
    if (const int orc = batch_overrun_check(b, st, err, errlen)) return orc;
    if (b->text_reads_state == 0) {   // once per load: every read needs a pair to hang its line on; the segments of the record path
        b->text_reads_state = 1;
        const uint64_t SEG = fin_text3_seg_pairs();
        uint64_t n_seg = 0; bool longer = false;
        for (uint64_t r = 0; r < b->n_reads; r++) {
            const uint64_t nk = b->h_out_offs[r + 1] - b->h_out_offs[r];
            if (nk == 0) { b->text_reads_state = 2; break; }
            n_seg += (nk + SEG - 1) / SEG; longer = longer || nk > SEG;
        }
        b->seg_table = false; b->n_seg = (uint32_t)b->n_reads;
        if (b->text_reads_state == 1 && longer) {
            std::vector<uint32_t> seg; seg.reserve(2 * n_seg);
            for (uint64_t r = 0; r < b->n_reads; r++) {
                const uint64_t nk = b->h_out_offs[r + 1] - b->h_out_offs[r];
                for (uint64_t f = 0; f < nk; f += SEG) { seg.push_back((uint32_t)r); seg.push_back((uint32_t)f); }
            }
            if (batch_grow(b, &b->d_seg, b->cap_seg, seg.size() * 4, st)) { b->text_reads_state = 0; set_err(err, errlen, "out of device memory (text segments)"); return FIN_ENOMEM; }
            HIPCHK(hipMemcpyAsync(b->d_seg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
            b->seg_table = true; b->n_seg = (uint32_t)n_seg;
        }
    }
    if (b->text_reads_state == 2) { set_err(err, errlen, "a read without k-mers: its empty line belongs to no pair (format such batches on the host)"); return FIN_EINVAL; }
    const uint32_t nb = fin_text_blocks(b->n_kmers);
    if (b->ran && b->last_frec) {
        // the run left fast-path records: the finished reads' pairs are made from them again (in text-only mode they exist nowhere else), the
        // length pass counts the found pairs on its way
        if (batch_grow(b, &b->d_blk_sum, b->cap_blk_sum, (size_t)b->n_seg * 4 + 4, st) ||
            batch_grow(b, &b->d_blk_off, b->cap_blk_off, (size_t)fin_text3_off_words(b->n_seg) * 8 + 8, st)) { set_err(err, errlen, "out of device memory (text tables)"); return FIN_ENOMEM; }
        if (!b->d_total) HIPCHK(hipMalloc((void**)&b->d_total, 8));
        HIPCHK(hipMemsetAsync(b->d_count, 0, 8, st));
        const void* seg = b->seg_table ? b->d_seg : nullptr;
        int rc = fin_launch_text3_lengths(b->d_out, (const uint64_t*)b->d_out_offs, b->d_frec, seg, b->n_seg, b->dev.k, (uint32_t*)b->d_blk_sum, (uint64_t*)b->d_blk_off, b->d_total, b->d_count, st);
        if (rc != 0) { set_err(err, errlen, std::string("text kernels: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
        uint64_t total = 0;
        HIPCHK(hipMemcpyAsync(&total, b->d_total, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (batch_grow(b, &b->d_text, b->cap_text, (size_t)total + 16, st)) { set_err(err, errlen, "out of device memory (text)"); return FIN_ENOMEM; }
        rc = fin_launch_text3_write(b->d_out, (const uint64_t*)b->d_out_offs, b->d_frec, seg, b->n_seg, b->dev.k, (const uint64_t*)b->d_blk_off, (char*)b->d_text, st);
        if (rc != 0) { set_err(err, errlen, std::string("text kernels: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
        b->count_from_text = true;
        b->text_bytes = total;
        if (text_bytes) *text_bytes = total;
        return FIN_OK;
    }
    if (batch_grow(b, &b->d_last_bits, b->cap_last_bits, ((b->n_kmers + 31) / 32 + 1) * 4, st) || batch_grow(b, &b->d_blk_sum, b->cap_blk_sum, (size_t)nb * 4 + 4, st) ||
        batch_grow(b, &b->d_blk_off, b->cap_blk_off, (size_t)fin_text_off_words(b->n_kmers) * 8 + 8, st)) { set_err(err, errlen, "out of device memory (text tables)"); return FIN_ENOMEM; }
    if (!b->d_total) HIPCHK(hipMalloc((void**)&b->d_total, 8));
    int rc = fin_launch_text_lengths(b->d_out, b->n_kmers, (const uint64_t*)b->d_out_offs, (uint32_t)b->n_reads, (uint32_t*)b->d_last_bits, (uint32_t*)b->d_blk_sum,
                                     (uint64_t*)b->d_blk_off, b->d_total, st);
    if (rc != 0) { set_err(err, errlen, std::string("text kernels: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, b->d_total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (batch_grow(b, &b->d_text, b->cap_text, (size_t)total + 16, st)) { set_err(err, errlen, "out of device memory (text)"); return FIN_ENOMEM; }
    rc = fin_launch_text_write(b->d_out, b->n_kmers, (const uint64_t*)b->d_blk_off, (const uint32_t*)b->d_last_bits, (char*)b->d_text, st);
    if (rc != 0) { set_err(err, errlen, std::string("text kernels: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
    b->text_bytes = total;
    if (text_bytes) *text_bytes = total;
    return FIN_OK;
}

int fin_batch_text_mode(fin_batch* b, int mode) {
    if (!b || mode < 0 || mode > 2) return FIN_EINVAL;
    b->text_mode = mode;
    return FIN_OK;
}

int fin_batch_download_text(fin_batch* b, char* text_out, char* err, size_t errlen) {
    if (!b || (b->text_bytes && !text_out)) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = b->ran ? b->last_stream : b->own_stream;
    if (const int orc = batch_overrun_check(b, st, err, errlen)) return orc;
    if (b->text_bytes) HIPCHK(hipMemcpyAsync(text_out, b->d_text, b->text_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return FIN_OK;
}

// ---- results as records (fin_records.hip) --------------------------------------------------------------------------------------
int fin_batch_records(fin_batch* b, uint64_t* n_stream_pairs, char* err, size_t errlen) {
    if (!b || !b->ran) { set_err(err, errlen, "no run to take records from"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = b->last_stream;
    if (const int orc = batch_overrun_check(b, st, err, errlen)) return orc;
    if (!b->last_frec) {
        // the run left no fast-path records (forward-only search, fast path off or not applicable, text mode 0): every read's pairs are in the batch's
        // output as they stand -- that IS the stream, and every record says "nk pairs follow"
        b->rec_passthrough = true; b->rec_stream_pairs = b->n_kmers; b->rec_ready = true;
        if (n_stream_pairs) *n_stream_pairs = b->n_kmers;
        return FIN_OK;
    }
    const uint32_t nb = fin_rec_blocks((uint32_t)b->n_reads);
    if (batch_grow(b, &b->d_blk_sum, b->cap_blk_sum, (size_t)nb * 4 + 4, st) || batch_grow(b, &b->d_blk_off, b->cap_blk_off, (size_t)nb * 8 + 8, st)) { set_err(err, errlen, "out of device memory (record tables)"); return FIN_ENOMEM; }
    if (!b->d_total) HIPCHK(hipMalloc((void**)&b->d_total, 8));
    int rc = fin_launch_rec_count(b->d_frec, (const uint64_t*)b->d_out_offs, (uint32_t)b->n_reads, (uint32_t*)b->d_blk_sum, (uint64_t*)b->d_blk_off, b->d_total, st);
    if (rc != 0) { set_err(err, errlen, std::string("record kernels: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, b->d_total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (batch_grow(b, &b->d_cstream, b->cap_cstream, (size_t)total * 8 + 16, st)) { set_err(err, errlen, "out of device memory (pair stream)"); return FIN_ENOMEM; }
    rc = fin_launch_rec_compact(b->d_frec, (const uint64_t*)b->d_out_offs, b->d_out, (uint32_t)b->n_reads, (const uint64_t*)b->d_blk_off, b->d_cstream, st);
    if (rc != 0) { set_err(err, errlen, std::string("record kernels: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
    b->rec_passthrough = false; b->rec_stream_pairs = total; b->rec_ready = true;
    if (n_stream_pairs) *n_stream_pairs = total;
    return FIN_OK;
}

int fin_batch_download_records(fin_batch* b, fin_read_record* recs_out, int32_t* stream_pairs_out, char* err, size_t errlen) {
    if (!b || !b->rec_ready || (b->n_reads && !recs_out) || (b->rec_stream_pairs && !stream_pairs_out)) { set_err(err, errlen, "fin_batch_records first; null buffer"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(b->device));
    hipStream_t st = b->last_stream;
    if (b->rec_passthrough) {
        if (b->n_kmers) HIPCHK(hipMemcpyAsync(stream_pairs_out, b->d_out, b->n_kmers * 8, hipMemcpyDeviceToHost, st));
        for (uint64_t r = 0; r < b->n_reads; r++) recs_out[r] = fin_read_record{0u, 0u, 0u, (uint32_t)(b->h_out_offs[r + 1] - b->h_out_offs[r]), 0ull, 0ull};
    } else {
        static_assert(sizeof(fin_read_record) == sizeof(FinFastRec), "the public record is the device's");
        if (b->n_reads) HIPCHK(hipMemcpyAsync(recs_out, b->d_frec, b->n_reads * sizeof(FinFastRec), hipMemcpyDeviceToHost, st));
        if (b->rec_stream_pairs) HIPCHK(hipMemcpyAsync(stream_pairs_out, b->d_cstream, b->rec_stream_pairs * 8, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return FIN_OK;
}

// host: the pairs fin_search_batch delivers, from records + stream.  A chunk of reads per thread: first where each chunk's pairs and stream begin, then the pairs.
int fin_expand_records(const fin_read_record* recs, uint64_t n_reads, const int32_t* stream_pairs, uint64_t n_stream_pairs, int k, int32_t* pairs_out,
                       uint64_t* n_positive, int n_threads) {
    if ((n_reads && (!recs || !pairs_out)) || k < 1 || (n_stream_pairs && !stream_pairs)) return FIN_EINVAL;
    int T = n_threads > 0 ? n_threads : fin_host_threads();
    if ((uint64_t)T > n_reads / 1024 + 1) T = (int)(n_reads / 1024 + 1);
    std::vector<uint64_t> out0((size_t)T + 1, 0), str0((size_t)T + 1, 0), pos((size_t)T, 0);
    auto bounds = [&](int t) { return std::make_pair(n_reads * (uint64_t)t / (uint64_t)T, n_reads * (uint64_t)(t + 1) / (uint64_t)T); };
    auto pass = [&](int t, bool write) {
        const auto lh = bounds(t);
        // (the counting pass starts every chunk at 0 -- its neighbour's count lands in the slot this chunk's start is read from in the second pass)
        uint64_t o = write ? out0[(size_t)t] : 0, sp = write ? str0[(size_t)t] : 0, found = 0;
        for (uint64_t r = lh.first; r < lh.second; r++) {
            const fin_read_record& R = recs[r];
            const uint32_t nk = R.nk, kind = R.meta >> 16;
            if (write) {
                int32_t* const dst = pairs_out + 2 * o;
                if (kind == 0u) {
                    if (sp + nk > n_stream_pairs) return false;
                    memcpy(dst, stream_pairs + 2 * sp, (size_t)nk * 8);
                    for (uint32_t i = 0; i < nk; i++) found += dst[2 * i] != -1;
                } else if (kind == 2u) {
                    for (uint32_t i = 0; i < 2 * nk; i++) dst[i] = -1;
                } else {
                    // one run: slot sl of strand A is (u, off0 + sl) unless a disagreeing position lies in [sl, sl + k - 1]; the reverse strand's slots mirror
                    const bool rev = (R.meta >> 8) & 1u;
                    const uint32_t nE = R.meta & 0xFFu;
                    for (uint32_t i = 0; i < nk; i++) { const uint32_t sl = rev ? nk - 1u - i : i; dst[2 * i] = (int32_t)R.u; dst[2 * i + 1] = (int32_t)(R.off0 + sl); }
                    uint64_t gaps = 0;
                    uint32_t done_to = 0;   // slots below this are settled (the positions ascend: stretches of absent slots may touch or overlap)
                    for (uint32_t e = 0; e < nE && e < 8u; e++) {
                        const uint32_t E = (uint32_t)((e < 4u ? R.Es : R.Es2) >> (16u * (e & 3u))) & 0xFFFFu;
                        uint32_t lo = E >= (uint32_t)(k - 1) ? E - (uint32_t)(k - 1) : 0u, hi = E < nk ? E : (nk ? nk - 1u : 0u);
                        if (lo < done_to) lo = done_to;
                        for (uint32_t sl = lo; nk && sl <= hi; sl++) { const uint32_t i = rev ? nk - 1u - sl : sl; dst[2 * i] = -1; dst[2 * i + 1] = -1; gaps++; }
                        if (hi + 1u > done_to) done_to = hi + 1u;
                    }
                    found += nk - gaps;
                }
            }
            o += nk; if (kind == 0u) sp += nk;
        }
        if (!write) { out0[(size_t)t + 1] = o; str0[(size_t)t + 1] = sp; } else pos[(size_t)t] = found;
        return true;
    };
    bool ok = true;
#pragma omp parallel for num_threads(T) schedule(static)
    for (int t = 0; t < T; t++) (void)pass(t, false);
    for (int t = 0; t < T; t++) { out0[(size_t)t + 1] += out0[(size_t)t]; str0[(size_t)t + 1] += str0[(size_t)t]; }
    if (str0[(size_t)T] != n_stream_pairs) return FIN_EINVAL;   // records and stream do not belong together
#pragma omp parallel for num_threads(T) schedule(static) reduction(&& : ok)
    for (int t = 0; t < T; t++) ok = pass(t, true) && ok;
    if (!ok) return FIN_EINVAL;
    if (n_positive) { uint64_t f = 0; for (uint64_t x : pos) f += x; *n_positive = f; }
    return FIN_OK;
}

int fin_batch_step_time(const fin_batch* b, uint64_t skip_first, double ms_parts[5], uint64_t* n_runs) {
    if (!b) return FIN_EINVAL;
    double t[5] = {0, 0, 0, 0, 0}; uint64_t n = 0;
    (void)hipSetDevice(b->device);
    for (size_t i = (size_t)skip_first; i < b->runs.size(); i++) {
        const fin_batch::RunEvents& r = b->runs[i];
        if (hipEventSynchronize(r.e[4]) != hipSuccess) continue;
        float tot = 0;
        if (hipEventElapsedTime(&tot, r.e[0], r.e[4]) != hipSuccess) { (void)hipGetLastError(); continue; }
        // events a kernel variant does not record (no pre-pass, empty batch) fold their interval into the next recorded one
        float part[4] = {0, 0, 0, 0};
        int prev = 0;
        for (int j = 1; j <= 4; j++) {
            if (hipEventQuery(r.e[j]) != hipSuccess) { (void)hipGetLastError(); continue; }
            float d = 0;
            if (hipEventElapsedTime(&d, r.e[prev], r.e[j]) != hipSuccess) { (void)hipGetLastError(); continue; }
            part[j - 1] = d; prev = j;
        }
        for (int j = 0; j < 4; j++) t[j] += part[j];
        t[4] += tot; n++;
    }
    if (ms_parts) for (int j = 0; j < 5; j++) ms_parts[j] = n ? t[j] / (double)n : 0.0;
    if (n_runs) *n_runs = n;
    return FIN_OK;
}

int fin_batch_kernel_time(const fin_batch* b, double* ms_avg, uint64_t* n_runs) {
    double p[5];
    const int rc = fin_batch_step_time(b, 0, p, n_runs);
    if (rc == FIN_OK && ms_avg) *ms_avg = p[4];
    return rc;
}

extern "C" void fin_debug_dump_w(void);
extern "C" void fin_debug_dump_pp(void);
void fin_debug_time(void) { fin_debug_dump_time(); fin_debug_dump_w(); fin_debug_dump_pp(); }

int fin_batch_pipeline_counts(fin_batch* b, uint32_t* out, uint32_t n_words) {
    if (!b || !out) return FIN_EINVAL;
    for (uint32_t i = 0; i < n_words; i++) out[i] = 0;
    if (!b->d_ctr || !b->ran) return FIN_OK;
    if (hipSetDevice(b->device) != hipSuccess || hipStreamSynchronize(b->last_stream) != hipSuccess) return FIN_ENODEV;
    const uint32_t n = std::min<uint32_t>(n_words, fin_v4_counter_words());
    return hipMemcpy(out, b->d_ctr, n * 4, hipMemcpyDeviceToHost) == hipSuccess ? FIN_OK : FIN_ENODEV;
}

// what the most recent fin_batch_run decided (the per-run decision, not a guess from the replica): out[0] the kernel that ran
// (after the k > 128 rule), [1] 1 = nothing prefilled the output (the pipeline wrote every slot once), [2] 1 = second strands were deferred,
// [3] 1 = the pre-pass's fast path was on
int fin_batch_run_info(const fin_batch* b, uint32_t out[4]) {
    if (!b || !out) return FIN_EINVAL;
    out[0] = b->last_kernel; out[1] = b->last_no_prefill; out[2] = b->ran ? b->dev.defer_ok : 0u; out[3] = (b->ran && b->dev.defer_ok && b->last_no_prefill) ? b->dev.fast_path : 0u;
    return FIN_OK;
}

int64_t fin_batch_overflow_reads(fin_batch* b) {
    if (!b) return -1;
    if (!b->ran) return 0;
    if (hipSetDevice(b->device) != hipSuccess || hipStreamSynchronize(b->last_stream) != hipSuccess) return -1;
    uint32_t c = 0;
    if (hipMemcpy(&c, b->d_ovf_count, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)c;   // (entries of the list: under kernel 4 a read may be on it more than once)
}

// reads [lo, hi) of a flat read set on one device; pairs_out points at read lo's first pair.
// The range is cut into sub-batches that a few host threads push through reusable device batches, each on its own stream:
// while one sub-batch is being searched, the previous one's pairs travel back over PCIe and the next one's reads travel in
// (H2D + kernel + D2H double-buffered).  With page-locked caller buffers (fin_host_alloc) the copies are DMA at
// link speed; pageable buffers work too, staged by the runtime.
// text mode of the pipeline below: every sub-batch's text lands behind its predecessors' in one caller buffer
struct TextSink {
    // (records instead of text when `recs` is set: fin_search_batch_records -- `len` then counts a sub-batch's stream pairs)
    fin_read_record* recs = nullptr; int32_t* rpairs = nullptr; uint64_t rcap = 0; uint64_t read0 = 0;
    char* buf = nullptr; uint64_t cap = 0;
    std::vector<uint64_t> len; std::vector<char> known;
    std::mutex mu; std::condition_variable cv;
    uint64_t total = 0;
};

static int search_range_on(const fin_index* idx, int device, const char* bases, const uint64_t* offsets, uint64_t lo, uint64_t hi,
                           int strands, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen, TextSink* ts = nullptr) {
    // A device batch addresses k-mers and bases with 32 bits (also bounds the HBM one sub-batch takes)
    const uint64_t MAX_BASES = 1ull << 31, MAX_READS = 1ull << 26;
    const uint64_t MAX_KMERS = std::min<uint64_t>((uint64_t)optv(idx, O_max_batch_kmers), (uint64_t)optv(idx, O_pipeline_kmers));
    const uint64_t k = idx->k;
    struct Sub { uint64_t lo, hi, pair_off; };
    std::vector<Sub> subs;
    {
        uint64_t pair_off = 0;
        do {
            uint64_t h2 = lo, nb = 0, nk = 0;
            while (h2 < hi) {
                const uint64_t len = offsets[h2 + 1] - offsets[h2];
                const uint64_t kk = len >= k ? len - k + 1 : 0;
                if (h2 > lo && (nb + len > MAX_BASES || nk + kk > MAX_KMERS || h2 - lo >= MAX_READS)) break;
                nb += len; nk += kk; h2++;
            }
            subs.push_back(Sub{lo, h2, pair_off});
            pair_off += nk; lo = h2;
        } while (lo < hi);
    }
    if (ts) { ts->len.assign(subs.size(), 0); ts->known.assign(subs.size(), 0); }
    const uint64_t hi_bases = offsets[subs.back().hi] - offsets[subs.front().lo];
    const int n_workers = (int)std::min<size_t>(subs.size(), (size_t)optv(idx, O_pipeline_depth));
    std::atomic<size_t> next{0};
    std::atomic<int> first_rc{FIN_OK};
    std::atomic<uint64_t> pos_total{0};
    std::mutex err_mu;
    const bool stage_in = optv(idx, O_stage_pageable) && hi_bases > 0 && !is_page_locked(bases + offsets[subs[0].lo]);
    const bool stage_out = optv(idx, O_stage_pageable) && pairs_out && !is_page_locked(pairs_out);
    auto worker = [&]() {
        fin_batch* b = nullptr;
        char e[512] = {0};
        void* sin = nullptr; size_t sin_cap = 0; void* sout = nullptr; size_t sout_cap = 0;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= subs.size() || first_rc.load() != FIN_OK) break;
            const Sub& s = subs[i];
            int rc = FIN_OK;
            const char* first = bases ? bases + offsets[s.lo] : nullptr;
            const size_t nb = (size_t)(offsets[s.hi] - offsets[s.lo]);
            if (stage_in && nb) {
                if (nb > sin_cap) { g_stage.put(sin, sin_cap); sin = g_stage.get(nb, &sin_cap); }
                if (!sin) { rc = FIN_ENOMEM; snprintf(e, sizeof e, "page-locked staging buffer: out of memory"); }
                else { memcpy(sin, first, nb); first = (const char*)sin; }
            }
            if (rc == FIN_OK) {
                if (!b) {   // a batch the pipeline used before on this device, if there is one
                    std::lock_guard<std::mutex> g(idx->pool_mu);
                    for (size_t j = 0; j < idx->batch_pool.size(); j++)
                        if (idx->batch_pool[j].first == device) { b = idx->batch_pool[j].second; idx->batch_pool.erase(idx->batch_pool.begin() + (long)j); break; }
                }
                if (!b) {
                    const fin_index::Replica* rep = idx->replica_on(device);
                    b = new (std::nothrow) fin_batch();
                    if (!b || !rep) { rc = FIN_ENOMEM; snprintf(e, sizeof e, "out of memory"); }
                    else { b->idx = idx; b->dev = rep->dev; b->device = device; }
                }
                if (rc == FIN_OK) rc = batch_load(b, first, offsets + s.lo, s.hi - s.lo, e, sizeof e);
            }
            if (rc == FIN_OK) { b->text_mode = ts ? 2 : 0; rc = fin_batch_run(b, strands, (void*)b->own_stream, e, sizeof e); }   // (text sink: the text is the only product)
            uint64_t pos = 0;
            if (rc == FIN_OK && ts && ts->recs) {
                // records: the sub-batch's stream of pairs lands behind the streams of all earlier sub-batches, its records at its reads' numbers
                uint64_t L = 0;
                rc = fin_batch_records(b, &L, e, sizeof e);
                {
                    std::lock_guard<std::mutex> g(ts->mu);
                    ts->len[i] = rc == FIN_OK ? L : 0; ts->known[i] = 1;
                }
                ts->cv.notify_all();
                if (rc == FIN_OK) {
                    uint64_t at = 0;
                    {
                        std::unique_lock<std::mutex> g(ts->mu);
                        ts->cv.wait(g, [&] { for (size_t j = 0; j < i; j++) if (!ts->known[j]) return false; return true; });
                        for (size_t j = 0; j < i; j++) at += ts->len[j];
                    }
                    if (at + L > ts->rcap) { rc = FIN_ELIMIT; snprintf(e, sizeof e, "pair stream buffer too small"); }
                    else rc = fin_batch_download_records(b, ts->recs + (s.lo - ts->read0), ts->rpairs ? ts->rpairs + 2 * at : nullptr, e, sizeof e);
                }
            } else
            if (rc == FIN_OK && ts) {
                // the text is made on the device; its place in the caller's buffer is behind the text of all earlier sub-batches
                uint64_t L = 0;
                rc = fin_batch_format_text(b, &L, e, sizeof e);
                {
                    std::lock_guard<std::mutex> g(ts->mu);
                    ts->len[i] = rc == FIN_OK ? L : 0; ts->known[i] = 1;
                }
                ts->cv.notify_all();
                if (rc == FIN_OK) {
                    uint64_t at = 0;
                    {
                        std::unique_lock<std::mutex> g(ts->mu);
                        ts->cv.wait(g, [&] { for (size_t j = 0; j < i; j++) if (!ts->known[j]) return false; return true; });
                        for (size_t j = 0; j < i; j++) at += ts->len[j];
                    }
                    if (at + L > ts->cap) { rc = FIN_ELIMIT; snprintf(e, sizeof e, "text buffer too small"); }
                    else rc = fin_batch_download_text(b, ts->buf + at, e, sizeof e);
                    if (rc == FIN_OK && n_positive) rc = fin_batch_download(b, nullptr, &pos, e, sizeof e);
                }
            } else
            if (rc == FIN_OK) {
                int32_t* dst = pairs_out ? pairs_out + 2 * s.pair_off : nullptr;
                const size_t ob = (size_t)b->n_kmers * 8;
                if (stage_out && ob) {
                    if (ob > sout_cap) { g_stage.put(sout, sout_cap); sout = g_stage.get(ob, &sout_cap); }
                    if (!sout) { rc = FIN_ENOMEM; snprintf(e, sizeof e, "page-locked staging buffer: out of memory"); }
                    else {
                        rc = fin_batch_download(b, (int32_t*)sout, n_positive ? &pos : nullptr, e, sizeof e);
                        if (rc == FIN_OK) memcpy(dst, sout, ob);
                    }
                } else rc = fin_batch_download(b, dst, n_positive ? &pos : nullptr, e, sizeof e);
            }
            if (rc != FIN_OK) {
                int expect = FIN_OK;
                if (first_rc.compare_exchange_strong(expect, rc)) { std::lock_guard<std::mutex> g(err_mu); set_err(err, errlen, e); }
                if (ts) {   // nobody may wait for the sub-batches this worker will not do
                    { std::lock_guard<std::mutex> g(ts->mu); for (size_t j = 0; j < subs.size(); j++) ts->known[j] = 1; }
                    ts->cv.notify_all();
                }
                break;
            }
            pos_total += pos;
        }
        if (b && first_rc.load() == FIN_OK) {   // kept for the next call (at most 8 per index)
            std::lock_guard<std::mutex> g(idx->pool_mu);
            if (idx->batch_pool.size() < 8) { idx->batch_pool.push_back({device, b}); b = nullptr; }
        }
        fin_batch_free(b);
        g_stage.put(sin, sin_cap); g_stage.put(sout, sout_cap);
    };
    if (n_workers <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (int w = 0; w < n_workers; w++) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    if (first_rc.load() != FIN_OK) return first_rc.load();
    if (n_positive) *n_positive = pos_total.load();
    if (ts) { ts->total = 0; for (uint64_t l : ts->len) ts->total += l; }
    return FIN_OK;
}

// ---- the streaming loop with the reference's text as its result (search_fmin.hh:43-72 including :62-65) ----
struct fin_text { void* p = nullptr; size_t cap = 0; uint64_t size = 0; };
fin_text* fin_text_create(void) { return new (std::nothrow) fin_text(); }
void fin_text_free(fin_text* t) { if (t) { if (t->p) (void)hipHostFree(t->p); delete t; } }
int fin_text_reserve(fin_text* t, uint64_t bytes) {   // page-lock room ahead of time (about 0.15 s per GB: worth doing beside other start-up work)
    if (!t) return FIN_EINVAL;
    if (bytes <= t->cap) return FIN_OK;
    if (t->p) (void)hipHostFree(t->p);
    t->p = nullptr; t->cap = 0; t->size = 0;
    if (hipHostMalloc(&t->p, bytes, hipHostMallocDefault) != hipSuccess) return FIN_ENOMEM;
    t->cap = bytes;
    return FIN_OK;
}
const char* fin_text_data(const fin_text* t) { return t ? (const char*)t->p : nullptr; }
uint64_t fin_text_size(const fin_text* t) { return t ? t->size : 0; }

int fin_search_batch_text(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads, int strands, fin_text* out,
                          uint64_t* n_positive, char* err, size_t errlen) {
    if (!idx || !offsets || !out) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    if (idx->replicas.empty()) { set_err(err, errlen, "index is not resident on a device: call fin_index_to_device first (no CPU fallback)"); return FIN_ENODEV; }
    if (n_positive) *n_positive = 0;
    out->size = 0;
    if (n_reads == 0) return FIN_OK;
    const uint64_t k = idx->k;
    uint64_t nk = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t len = offsets[r + 1] - offsets[r];
        if (len < k) { set_err(err, errlen, "a read without k-mers: its empty line belongs to no pair (format such batches on the host)"); return FIN_EINVAL; }
        nk += len - k + 1;
    }
    // no pair's text is longer than "(last unitig,last offset of the longest unitig)" + separator, nor shorter than "(-1,-1) "
    uint64_t max_len = 1;
    for (uint64_t u = 0; u < idx->n_unitigs; u++) max_len = std::max<uint64_t>(max_len, (uint64_t)idx->ends[u + 1] - idx->ends[u]);
    auto ndig = [](uint64_t v) { uint64_t n = 1; while (v >= 10) { v /= 10; n++; } return n; };
    const uint64_t per_pair = std::max<uint64_t>(8, 4 + ndig(idx->n_unitigs ? idx->n_unitigs - 1 : 0) + ndig(max_len - 1));
    const uint64_t need = per_pair * nk + 64;
    if (need > out->cap) {
        if (out->p) (void)hipHostFree(out->p);
        out->p = nullptr; out->cap = 0;
        if (hipHostMalloc(&out->p, need + need / 16, hipHostMallocDefault) != hipSuccess) { set_err(err, errlen, "page-locked text buffer: out of memory"); return FIN_ENOMEM; }
        out->cap = need + need / 16;
    }
    TextSink ts; ts.buf = (char*)out->p; ts.cap = out->cap;
    const int rc = search_range_on(idx, idx->replicas[0].device, bases, offsets, 0, n_reads, strands, nullptr, n_positive, err, errlen, &ts);
    if (rc == FIN_OK) out->size = ts.total;
    return rc;
}

int fin_search_batch_records(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads, fin_read_record* recs_out,
                             int32_t* stream_pairs_out, uint64_t stream_cap_pairs, uint64_t* n_stream_pairs, char* err, size_t errlen) {
    if (!idx || !offsets || (n_reads && !recs_out)) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    if (idx->replicas.empty()) { set_err(err, errlen, "index is not resident on a device: call fin_index_to_device first (no CPU fallback)"); return FIN_ENODEV; }
    if (n_stream_pairs) *n_stream_pairs = 0;
    if (n_reads == 0) return FIN_OK;
    TextSink ts; ts.recs = recs_out; ts.rpairs = stream_pairs_out; ts.rcap = stream_pairs_out ? stream_cap_pairs : 0; ts.read0 = 0;
    const int rc = search_range_on(idx, idx->replicas[0].device, bases, offsets, 0, n_reads, FIN_MERGED, nullptr, nullptr, err, errlen, &ts);
    if (rc == FIN_OK && n_stream_pairs) *n_stream_pairs = ts.total;
    return rc;
}

int fin_search_batch(const fin_index* idx, const char* bases, const uint64_t* offsets, uint64_t n_reads, int strands,
                     int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen) {
    if (!idx || !offsets) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    if (idx->replicas.empty()) { set_err(err, errlen, "index is not resident on a device: call fin_index_to_device first (no CPU fallback)"); return FIN_ENODEV; }
    if (n_positive) *n_positive = 0;
    return search_range_on(idx, idx->replicas[0].device, bases, offsets, 0, n_reads, strands, pairs_out, n_positive, err, errlen);
}

int fin_search_batch_multi(fin_index* idx, const int* devices, int n_devices, const char* bases, const uint64_t* offsets,
                           uint64_t n_reads, int strands, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen) {
    if (!idx || !offsets || !devices || n_devices < 1) { set_err(err, errlen, "bad argument"); return FIN_EINVAL; }
    for (int d = 0; d < n_devices; d++) { int rc = fin_index_to_device(idx, devices[d], err, errlen); if (rc) return rc; }
    if (n_positive) *n_positive = 0;
    // contiguous shards of read records balanced by bases: per-device outputs concatenate in input order
    const uint64_t k = idx->k;
    std::vector<uint64_t> cut((size_t)n_devices + 1, n_reads), pair_at((size_t)n_devices + 1, 0);
    cut[0] = 0;
    const uint64_t total = offsets[n_reads] - offsets[0];
    uint64_t r = 0, pairs = 0;
    for (int d = 1; d <= n_devices; d++) {
        const uint64_t target = d == n_devices ? total : total / (uint64_t)n_devices * (uint64_t)d;
        while (r < n_reads && (offsets[r] - offsets[0] < target || d == n_devices)) {
            const uint64_t len = offsets[r + 1] - offsets[r];
            pairs += len >= k ? len - k + 1 : 0; r++;
        }
        cut[(size_t)d] = r; pair_at[(size_t)d] = pairs;
    }
    std::vector<int> rcs((size_t)n_devices, FIN_OK);
    std::vector<uint64_t> pos((size_t)n_devices, 0);
    std::vector<std::string> errs((size_t)n_devices);
    std::vector<std::thread> th;
    for (int d = 0; d < n_devices; d++) {
        th.emplace_back([&, d]() {
            char e[512] = {0};
            if (cut[(size_t)d + 1] > cut[(size_t)d])
                rcs[(size_t)d] = search_range_on(idx, devices[d], bases, offsets, cut[(size_t)d], cut[(size_t)d + 1], strands,
                                                 pairs_out ? pairs_out + 2 * pair_at[(size_t)d] : nullptr, &pos[(size_t)d], e, sizeof e);
            errs[(size_t)d] = e;
        });
    }
    for (auto& t : th) t.join();
    uint64_t tot = 0;
    for (int d = 0; d < n_devices; d++) {
        if (rcs[(size_t)d] != FIN_OK) { set_err(err, errlen, "device " + std::to_string(devices[d]) + ": " + errs[(size_t)d]); return rcs[(size_t)d]; }
        tot += pos[(size_t)d];
    }
    if (n_positive) *n_positive = tot;
    return FIN_OK;
}

void* fin_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void fin_host_free(void* p) { if (p) (void)hipHostFree(p); }

int fin_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int fin_search(const fin_index* idx, const char* seq, int64_t len, int64_t* pairs_out, int64_t* n_found, char* err, size_t errlen) {
    if (!idx || len < 0 || (len && !seq)) { set_err(err, errlen, "bad argument"); return FIN_EINVAL; }
    uint64_t offs[2] = {0, (uint64_t)len};
    int64_t nk = len - (int64_t)idx->k + 1; if (nk < 0) nk = 0;
    std::vector<int32_t> tmp((size_t)(2 * nk + 2));
    int rc = fin_search_batch(idx, seq ? seq : "", offs, 1, FIN_FWD, tmp.data(), nullptr, err, errlen);
    if (rc) return rc;
    int64_t nf = 0;
    for (int64_t i = 0; i < nk; i++) {
        if (pairs_out) { pairs_out[2 * i] = tmp[2 * i]; pairs_out[2 * i + 1] = tmp[2 * i + 1]; }
        nf += tmp[2 * i] != -1;
    }
    if (n_found) *n_found = nf;
    return FIN_OK;
}

int64_t fin_format_pairs(const int32_t* pairs, int64_t n_pairs, char* out) {
    char* p = out;
    for (int64_t i = 0; i < n_pairs; i++) {
        if (i) *p++ = ' ';
        *p++ = '(';
        for (int h = 0; h < 2; h++) {
            int64_t v = pairs[2 * i + h];
            if (v < 0) { *p++ = '-'; v = -v; }
            char tmp[16]; int n = 0;
            do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
            while (n) *p++ = tmp[--n];
            *p++ = h == 0 ? ',' : ')';
        }
    }
    *p++ = '\n';
    return (int64_t)(p - out);
}


// ---- index sets: a unitig set beyond 2^32 nodes as PARTS (round 5; VERDICT r4 missing #1) ---------------------------------------------------------------
// The reference is int64_t throughout (common.hh:79-93, FinimizerIndex.hh:30-33); this build's node numbers and text offsets are u32 in every device
// structure, so ONE index stops at 2^32 nodes (a 4.1 Gbp unitig set: 4.12e9 nodes).  A set splits the input unitigs, in input order, into
// parts of at most max_part_bases bases, builds an ordinary index of each and searches a read in every part: a k-mer of a disjoint spectrum-preserving string
// set -- what the reference requires as input (README.md:79-80) -- lies in ONE unitig, hence in one part, and is reported there at the offset the whole
// index would report; the part's unitig number is mapped to the number permute_unitigs (PackedStrings.hh:105-135) gives that unitig in the whole set
// (rank of its first k-mer in colex order, ties by input order: the parts' orders are that order restricted to their unitigs).  What makes the set EXACT is
// checked when it is built (verify != 0): no part holds a k-mer twice (fin_index_is_disjoint), and no k-mer of one part occurs, on either strand, in
// another -- every part's unitigs are searched in the parts behind it; a set that fails is refused (FIN_EINVAL): which occurrence of a shared k-mer the
// reference reports is decided by its finimizers' stored offsets over the WHOLE index, which no part knows.  Cost: the parts' steps one after the other
// and a merge pass each (fin_set_merge_kernel).
struct fin_pindex {
    int k = 0, device = -1;
    std::vector<fin_index*> parts;
    std::vector<uint64_t> first_unitig;            // part p holds input unitigs [first_unitig[p], first_unitig[p+1])
    std::vector<std::vector<uint32_t>> gid;        // per part: its unitig number -> the set's
    std::vector<uint32_t*> d_gid;
    uint64_t n_unitigs = 0;
    int64_t shared_kmers = -1;                     // k-mers found in a part other than their own at build (-1: not checked)
    double verify_s = 0.0;
    // fin_pindex_search_batch keeps its device batches from call to call (a caller that streams chunks of reads through it allocates once)
    mutable std::mutex mu; mutable struct fin_pbatch* cached = nullptr;
};
void fin_pbatch_free(fin_pbatch* sb);

void fin_pindex_free(fin_pindex* s) {
    if (!s) return;
    if (s->device >= 0) (void)hipSetDevice(s->device);
    fin_pbatch_free(s->cached);
    for (uint32_t* g : s->d_gid) (void)hipFree(g);
    for (fin_index* p : s->parts) fin_index_free(p);
    delete s;
}
uint32_t fin_pindex_parts(const fin_pindex* s) { return s ? (uint32_t)s->parts.size() : 0u; }
const fin_index* fin_pindex_part(const fin_pindex* s, uint32_t p) { return (s && p < s->parts.size()) ? s->parts[p] : nullptr; }
int64_t fin_pindex_k(const fin_pindex* s) { return s ? s->k : -1; }
int64_t fin_pindex_n_unitigs(const fin_pindex* s) { return s ? (int64_t)s->n_unitigs : -1; }
int64_t fin_pindex_shared_kmers(const fin_pindex* s) { return s ? s->shared_kmers : -1; }
double fin_pindex_verify_seconds(const fin_pindex* s) { return s ? s->verify_s : 0.0; }
static int64_t set_sum(const fin_pindex* s, int64_t (*f)(const fin_index*)) {
    if (!s) return -1;
    int64_t t = 0;
    for (const fin_index* p : s->parts) t += f(p);
    return t;
}
int64_t fin_pindex_n_nodes(const fin_pindex* s) { return set_sum(s, fin_index_n_nodes); }
int64_t fin_pindex_n_kmers(const fin_pindex* s) { return set_sum(s, fin_index_n_kmers); }
int64_t fin_pindex_total_len(const fin_pindex* s) { return set_sum(s, fin_index_total_len); }
int64_t fin_pindex_size_in_bytes(const fin_pindex* s) { return set_sum(s, fin_index_size_in_bytes); }
int64_t fin_pindex_replica_table_bytes(const fin_pindex* s) {
    if (!s) return -1;
    int64_t t = 0;
    for (const fin_index* p : s->parts) { const int64_t v = fin_index_replica_table_bytes(p, s->device); if (v > 0) t += v; }
    return t;
}
int fin_pindex_unitig_ids(const fin_pindex* s, uint32_t part, uint32_t* out, uint64_t n) {
    if (!s || part >= s->gid.size() || !out || n != s->gid[part].size()) return FIN_EINVAL;
    std::memcpy(out, s->gid[part].data(), n * sizeof(uint32_t));
    return FIN_OK;
}

// every part's replica on the set's device, and the parts' tables of set-wide unitig numbers
static int pindex_upload(fin_pindex* s, char* err, size_t errlen) {
    HIPCHK(hipSetDevice(s->device));
    for (size_t p = 0; p < s->parts.size(); p++) {
        const int rc = fin_index_to_device(s->parts[p], s->device, err, errlen);
        if (rc != FIN_OK) return rc;
        uint32_t* d = nullptr;
        if (hipMalloc(&d, s->gid[p].size() * sizeof(uint32_t) + 16) != hipSuccess) { (void)hipGetLastError(); set_err(err, errlen, "out of device memory (unitig numbers)"); return FIN_ENOMEM; }
        s->d_gid.push_back(d);
        HIPCHK(hipMemcpy(d, s->gid[p].data(), s->gid[p].size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return FIN_OK;
}

// <prefix>.finparts (text: "finito-parts 1", k, the number of parts and unitigs, what the build's check counted, per part its first input unitig and its number of
// unitigs), <prefix>.p<i>.finamd (the part's container, fin_index_save) and <prefix>.p<i>.gid (its table of set-wide unitig numbers, u32 each)
int fin_pindex_save(const fin_pindex* s, const char* prefix, char* err, size_t errlen) {
    if (!s || !prefix) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    const std::string pre(prefix);
    for (size_t p = 0; p < s->parts.size(); p++) {
        const std::string pp = pre + ".p" + std::to_string(p);
        const int rc = fin_index_save(s->parts[p], pp.c_str(), err, errlen);
        if (rc != FIN_OK) return rc;
        FILE* f = fopen((pp + ".gid").c_str(), "wb");
        if (!f || fwrite(s->gid[p].data(), sizeof(uint32_t), s->gid[p].size(), f) != s->gid[p].size() || fclose(f) != 0) { set_err(err, errlen, "cannot write " + pp + ".gid"); if (f) fclose(f); return FIN_EIO; }
    }
    FILE* f = fopen((pre + ".finparts").c_str(), "w");
    if (!f) { set_err(err, errlen, "cannot write " + pre + ".finparts"); return FIN_EIO; }
    fprintf(f, "finito-parts 1\nk %d\nparts %zu\nunitigs %llu\nshared_kmers %lld\n", s->k, s->parts.size(), (unsigned long long)s->n_unitigs, (long long)s->shared_kmers);
    for (size_t p = 0; p < s->parts.size(); p++) fprintf(f, "part %zu first_unitig %llu unitigs %zu\n", p, (unsigned long long)s->first_unitig[p], s->gid[p].size());
    if (fclose(f) != 0) { set_err(err, errlen, "cannot write " + pre + ".finparts"); return FIN_EIO; }
    return FIN_OK;
}
int fin_pindex_exists(const char* prefix) { return prefix && file_exists(std::string(prefix) + ".finparts") ? 1 : 0; }
int fin_pindex_load(const char* prefix, int device, fin_pindex** out, char* err, size_t errlen) {
    if (!prefix || !out) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    const std::string pre(prefix);
    FILE* f = fopen((pre + ".finparts").c_str(), "r");
    if (!f) { set_err(err, errlen, "cannot open " + pre + ".finparts"); return FIN_EIO; }
    std::unique_ptr<fin_pindex, void (*)(fin_pindex*)> s(new fin_pindex, fin_pindex_free);
    s->device = device;
    int ver = 0; size_t P = 0; unsigned long long nu = 0; long long shared = -1;
    bool ok = fscanf(f, "finito-parts %d k %d parts %zu unitigs %llu shared_kmers %lld", &ver, &s->k, &P, &nu, &shared) == 5 && ver == 1 && P >= 1 && P < 65536;
    std::vector<size_t> cnt(ok ? P : 0);
    s->first_unitig.assign(ok ? P + 1 : 1, 0);
    for (size_t p = 0; ok && p < P; p++) {
        size_t pi = 0; unsigned long long fu = 0;
        ok = fscanf(f, " part %zu first_unitig %llu unitigs %zu", &pi, &fu, &cnt[p]) == 3 && pi == p;
        s->first_unitig[p] = fu;
    }
    fclose(f);
    if (!ok) { set_err(err, errlen, pre + ".finparts is not a partitioned index's manifest"); return FIN_EIO; }
    s->n_unitigs = nu; s->shared_kmers = shared; s->first_unitig[P] = nu;
    uint64_t seen = 0;
    for (size_t p = 0; p < P; p++) {
        const std::string pp = pre + ".p" + std::to_string(p);
        fin_index* x = nullptr;
        const int rc = fin_index_load(pp.c_str(), &x, err, errlen);
        if (rc != FIN_OK) return rc;
        s->parts.push_back(x);
        if (fin_index_k(x) != s->k || (size_t)fin_index_n_unitigs(x) != cnt[p]) { set_err(err, errlen, pp + ".finamd does not belong to this manifest"); return FIN_EIO; }
        s->gid.emplace_back(cnt[p]);
        FILE* g = fopen((pp + ".gid").c_str(), "rb");
        const bool gok = g && fread(s->gid[p].data(), sizeof(uint32_t), cnt[p], g) == cnt[p];
        if (g) fclose(g);
        if (!gok) { set_err(err, errlen, "cannot read " + pp + ".gid"); return FIN_EIO; }
        for (uint32_t v : s->gid[p]) if (v >= nu) { set_err(err, errlen, pp + ".gid holds a unitig number beyond the set"); return FIN_EIO; }
        seen += cnt[p];
    }
    if (seen != nu) { set_err(err, errlen, pre + ".finparts: the parts' unitigs do not add up"); return FIN_EIO; }
    const int rc = pindex_upload(s.get(), err, errlen);
    if (rc != FIN_OK) return rc;
    *out = s.release();
    return FIN_OK;
}

int fin_pindex_build_device(const char* unitig_bases, const uint64_t* unitig_offsets, uint64_t n_unitigs, int k, int device, uint64_t max_part_bases,
                               int verify, fin_pindex** out, char* err, size_t errlen) {
    if (!unitig_bases || !unitig_offsets || !out || n_unitigs == 0 || k < 2 || k > 255) { set_err(err, errlen, "bad argument"); return FIN_EINVAL; }
    if (n_unitigs >= 0x7FFFFFFFull) { set_err(err, errlen, "more than 2^31-1 unitigs: a pair's unitig number is an int32"); return FIN_ELIMIT; }
    if (max_part_bases == 0) max_part_bases = 3200000000ull;   // (about 3.2e9 nodes: room below 2^32 for the dummy nodes)
    if (max_part_bases > 4100000000ull) max_part_bases = 4100000000ull;
    std::unique_ptr<fin_pindex, void (*)(fin_pindex*)> s(new fin_pindex, fin_pindex_free);
    s->k = k; s->device = device; s->n_unitigs = n_unitigs;
    // parts: consecutive input unitigs while they fit
    s->first_unitig.push_back(0);
    {
        uint64_t in_part = 0;
        for (uint64_t u = 0; u < n_unitigs; u++) {
            const uint64_t len = unitig_offsets[u + 1] - unitig_offsets[u];
            if (len < (uint64_t)k) { set_err(err, errlen, "a unitig is shorter than k"); return FIN_EINVAL; }
            if (len > max_part_bases) { set_err(err, errlen, "a unitig is longer than a part may be"); return FIN_ELIMIT; }
            if (in_part + len > max_part_bases) { s->first_unitig.push_back(u); in_part = 0; }
            in_part += len;
        }
        s->first_unitig.push_back(n_unitigs);
    }
    const size_t P = s->first_unitig.size() - 1;
    for (size_t p = 0; p < P; p++) {
        const uint64_t u0 = s->first_unitig[p], nu = s->first_unitig[p + 1] - u0;
        fin_index* x = nullptr;
        const int rc = fin_index_build_device(unitig_bases, unitig_offsets + u0, nu, k, device, &x, nullptr, err, errlen);
        if (rc != FIN_OK) return rc;
        s->parts.push_back(x);
    }
    // the set's unitig numbers: permute_unitigs over ALL unitigs -- colex order of the first k-mers (the last base is the most significant), ties by input order
    {
        std::vector<uint32_t> order(n_unitigs);
        for (uint64_t u = 0; u < n_unitigs; u++) order[u] = (uint32_t)u;
        auto code = [](char c) -> int { c &= (char)0xDF; return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; };
        const char* const B = unitig_bases; const uint64_t* const O = unitig_offsets;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const char* x = B + O[a], *y = B + O[b];
            for (int j = k - 1; j >= 0; j--) { const int cx = code(x[j]), cy = code(y[j]); if (cx != cy) return cx < cy; }
            return false;
        });
        // (walk the order once: a part's c-th unitig in it is that part's unitig number c)
        s->gid.assign(P, {});
        for (size_t p = 0; p < P; p++) s->gid[p].reserve(s->first_unitig[p + 1] - s->first_unitig[p]);
        for (uint64_t r = 0; r < n_unitigs; r++) {
            const uint64_t u = order[r];
            const size_t p = (size_t)(std::upper_bound(s->first_unitig.begin(), s->first_unitig.end(), u) - s->first_unitig.begin()) - 1;
            s->gid[p].push_back((uint32_t)r);
        }
    }
    // replicas and the number tables
    { const int rc = pindex_upload(s.get(), err, errlen); if (rc != FIN_OK) return rc; }
    if (verify) {
        const auto t0 = std::chrono::steady_clock::now();
        int64_t shared = 0;
        for (size_t p = 0; p < P; p++)
            if (!fin_index_is_disjoint(s->parts[p])) shared += fin_index_total_len(s->parts[p]) - (int64_t)(s->first_unitig[p + 1] - s->first_unitig[p]) * (k - 1) - fin_index_n_kmers(s->parts[p]);
        // a part's unitigs, as reads, in every part behind it (merged search: a k-mer or its reverse complement there is a hit)
        for (size_t p = 0; p + 1 < P && P > 1; p++) {
            for (size_t t = p + 1; t < P; t++) {
                fin_batch* b = nullptr;
                uint64_t u = s->first_unitig[p];
                const uint64_t u_end = s->first_unitig[p + 1];
                while (u < u_end) {
                    uint64_t v = u, bases = 0;
                    while (v < u_end && (v == u || bases + (unitig_offsets[v + 1] - unitig_offsets[v]) <= (1ull << 31)) && v - u < (1ull << 27)) { bases += unitig_offsets[v + 1] - unitig_offsets[v]; v++; }
                    int rc = b ? fin_batch_reload(b, unitig_bases, unitig_offsets + u, v - u, err, errlen)
                               : fin_batch_create_on(s->parts[t], device, unitig_bases, unitig_offsets + u, v - u, &b, err, errlen);
                    uint64_t npos = 0;
                    if (rc == FIN_OK) rc = fin_batch_run(b, FIN_MERGED, nullptr, err, errlen);
                    if (rc == FIN_OK) rc = fin_batch_download(b, nullptr, &npos, err, errlen);
                    if (rc != FIN_OK) { fin_batch_free(b); return rc; }
                    shared += (int64_t)npos;
                    u = v;
                }
                fin_batch_free(b);
            }
        }
        s->shared_kmers = shared;
        s->verify_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (shared != 0) {
            set_err(err, errlen, "index set: " + std::to_string(shared) + " k-mer occurrence(s) are not the only one of their k-mer (or of its reverse complement) in the set -- "
                                 "not a disjoint spectrum-preserving string set; parts cannot tell which occurrence the whole index would report");
            return FIN_EINVAL;
        }
    }
    *out = s.release();
    return FIN_OK;
}

struct fin_pbatch {
    const fin_pindex* set = nullptr;
    std::vector<fin_batch*> b;
    struct Ev { hipEvent_t e0, e1; };
    std::vector<Ev> runs;
    hipStream_t last_stream = nullptr; bool ran = false;
};
void fin_pbatch_free(fin_pbatch* sb) {
    if (!sb) return;
    if (sb->set && sb->set->device >= 0) (void)hipSetDevice(sb->set->device);
    for (auto& r : sb->runs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (fin_batch* x : sb->b) fin_batch_free(x);
    delete sb;
}
int fin_pbatch_create(const fin_pindex* s, const char* bases, const uint64_t* offsets, uint64_t n_reads, fin_pbatch** out, char* err, size_t errlen) {
    if (!s || !out || !offsets) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    std::unique_ptr<fin_pbatch, void (*)(fin_pbatch*)> sb(new fin_pbatch, fin_pbatch_free);
    sb->set = s;
    for (fin_index* p : s->parts) {
        fin_batch* x = nullptr;
        const int rc = fin_batch_create_on(p, s->device, bases, offsets, n_reads, &x, err, errlen);
        if (rc != FIN_OK) return rc;
        sb->b.push_back(x);
    }
    *out = sb.release();
    return FIN_OK;
}
int fin_pbatch_reload(fin_pbatch* sb, const char* bases, const uint64_t* offsets, uint64_t n_reads, char* err, size_t errlen) {
    if (!sb || !offsets) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    for (fin_batch* x : sb->b) { const int rc = fin_batch_reload(x, bases, offsets, n_reads, err, errlen); if (rc != FIN_OK) return rc; }
    sb->ran = false;
    return FIN_OK;
}
uint64_t fin_pbatch_n_kmers(const fin_pbatch* sb) { return (sb && !sb->b.empty()) ? sb->b[0]->n_kmers : 0; }
void* fin_pbatch_device_pairs(const fin_pbatch* sb) { return (sb && !sb->b.empty()) ? sb->b[0]->d_out : nullptr; }
// one step of the set: every part's step (fin_batch_run, merged strands) and its merge into the first part's output buffer, all on `hip_stream`
int fin_pbatch_run(fin_pbatch* sb, void* hip_stream, char* err, size_t errlen) {
    if (!sb || sb->b.empty()) { set_err(err, errlen, "bad argument"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(sb->set->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (sb->runs.size() >= 1024) { (void)hipEventDestroy(sb->runs.front().e0); (void)hipEventDestroy(sb->runs.front().e1); sb->runs.erase(sb->runs.begin()); }
    fin_pbatch::Ev ev;
    HIPCHK(hipEventCreate(&ev.e0)); HIPCHK(hipEventCreate(&ev.e1));
    sb->runs.push_back(ev);
    HIPCHK(hipEventRecord(ev.e0, st));
    for (size_t p = 0; p < sb->b.size(); p++) {
        const int rc = fin_batch_run(sb->b[p], FIN_MERGED, hip_stream, err, errlen);
        if (rc != FIN_OK) return rc;
        const int e = fin_launch_set_merge(sb->b[0]->d_out, sb->b[p]->d_out, sb->set->d_gid[p], sb->b[p]->n_kmers, p == 0 ? 1 : 0, st);
        if (e != 0) { set_err(err, errlen, std::string("merge kernel launch failed: ") + hipGetErrorString((hipError_t)e)); return FIN_ENODEV; }
    }
    HIPCHK(hipEventRecord(ev.e1, st));
    sb->last_stream = st; sb->ran = true;
    return FIN_OK;
}
int fin_pbatch_download(fin_pbatch* sb, int32_t* pairs_out, uint64_t* n_positive, char* err, size_t errlen) {
    if (!sb || sb->b.empty() || !sb->ran) { set_err(err, errlen, "fin_pbatch_download: nothing has run"); return FIN_EINVAL; }
    HIPCHK(hipSetDevice(sb->set->device));
    HIPCHK(hipStreamSynchronize(sb->last_stream));
    // (a part whose overflow list overran withholds its results: the set's are then incomplete)
    for (fin_batch* x : sb->b) { const int rc = batch_overrun_check(x, sb->last_stream, err, errlen); if (rc != FIN_OK) return rc; }
    fin_batch* b0 = sb->b[0];
    hipStream_t st = sb->last_stream;
    unsigned long long c = 0;
    if (n_positive && b0->n_kmers) {
        const int rc = fin_launch_count_positive(b0->d_out, b0->n_kmers, b0->d_count, st);
        if (rc != 0) { set_err(err, errlen, std::string("count kernel launch failed: ") + hipGetErrorString((hipError_t)rc)); return FIN_ENODEV; }
        HIPCHK(hipMemcpyAsync(&c, b0->d_count, 8, hipMemcpyDeviceToHost, st));
    }
    if (pairs_out && b0->n_kmers) HIPCHK(hipMemcpyAsync(pairs_out, b0->d_out, b0->n_kmers * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_positive) *n_positive = c;
    return FIN_OK;
}
// device time of a set step (HIP events on the launch stream), averaged over the runs behind the first skip_first
int fin_pbatch_step_time(const fin_pbatch* sb, uint64_t skip_first, double* ms_avg, uint64_t* n_runs) {
    if (!sb || !ms_avg) return FIN_EINVAL;
    double t = 0; uint64_t n = 0;
    for (size_t i = (size_t)skip_first; i < sb->runs.size(); i++) {
        float ms = 0;
        if (hipEventSynchronize(sb->runs[i].e1) != hipSuccess || hipEventElapsedTime(&ms, sb->runs[i].e0, sb->runs[i].e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        t += ms; n++;
    }
    *ms_avg = n ? t / (double)n : 0.0;
    if (n_runs) *n_runs = n;
    return FIN_OK;
}
// merged search of a flat read set in every part of the set (host buffers; one device batch per part)
int fin_pindex_search_batch(const fin_pindex* s, const char* bases, const uint64_t* offsets, uint64_t n_reads, int32_t* pairs_out, uint64_t* n_positive,
                               char* err, size_t errlen) {
    if (!s) { set_err(err, errlen, "null argument"); return FIN_EINVAL; }
    std::lock_guard<std::mutex> g(s->mu);   // (one search at a time per set: its parts share the device's memory with the cached batches)
    int rc = s->cached ? fin_pbatch_reload(s->cached, bases, offsets, n_reads, err, errlen) : fin_pbatch_create(s, bases, offsets, n_reads, &s->cached, err, errlen);
    if (rc == FIN_OK) rc = fin_pbatch_run(s->cached, nullptr, err, errlen);
    if (rc == FIN_OK) rc = fin_pbatch_download(s->cached, pairs_out, n_positive, err, errlen);
    if (rc != FIN_OK) { fin_pbatch_free(s->cached); s->cached = nullptr; }
    return rc;
}

}  // extern "C"
