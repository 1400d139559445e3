// FinimizerIndex.hh -- C++ host mirror of the reference's FinimizerIndex (include/FinimizerIndex.hh:26-259)
// over the C ABI of include/finito_amd.h.  Same member names and error behaviour (std::runtime_error, caught by
// main exactly like src/main.cpp:51-57), so the reference's call sites (search_fmin.hh:47,51; tests.cpp:95...)
// read the same.  Header-only; links against libfinito_amd.so.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/finito_amd.h"

class FinimizerIndex {
public:
    struct QueryResult {
        std::vector<std::pair<int64_t, int64_t>> local_offsets;   // unitig id, offset in the unitig
        int64_t n_found = 0;
    };

private:
    FinimizerIndex(const FinimizerIndex&) = delete;              // as the reference (:36-38)
    FinimizerIndex& operator=(const FinimizerIndex&) = delete;
    fin_index* h = nullptr;
    // a unitig set beyond 2^32 nodes: the index is a PARTITIONED one (fin_pindex_*; PartitionedFinimizerIndex below is the plain wrapper) -- load() finds it by
    // its manifest, build_partitioned() makes it; searches, counts and serialize() go to it, the reference's public members are not materialised
    fin_pindex* ph = nullptr;
    int device = 0;
    std::vector<int> devices;   // more than one: batches are sharded by record over these GPUs

    static void check(int rc, const char* err) {
        if (rc != FIN_OK) throw std::runtime_error(err[0] ? err : "finito_amd call failed");
    }

public:
    FinimizerIndex() { wire(); }
    explicit FinimizerIndex(int device_) : device(device_) { wire(); }
    // use GPUs first .. first+n-1 of this node for batches (reads sharded by record, index replicated)
    void use_devices(int first, int n) { device = first; devices.clear(); for (int i = 0; i < n; i++) devices.push_back(first + i); }
    ~FinimizerIndex() { fin_index_free(h); fin_pindex_free(ph); }
    bool partitioned() const { return ph != nullptr; }
    int64_t number_of_parts() const { return ph ? (int64_t)fin_pindex_parts(ph) : 1; }

    // FinimizerIndexBuilder (FinimizerIndex.hh:262-395): unitigs as one buffer + n+1 offsets
    void build(const std::string& bases, const std::vector<uint64_t>& offsets, int k, int n_threads = 0) {
        char err[512] = {0};
        fin_index_free(h); h = nullptr; forget();
        check(fin_index_build(bases.data(), offsets.data(), offsets.size() - 1, k, n_threads, &h, err, sizeof err), err);
    }
    // the same on a HIP device (fin_index_build_device: any k <= 255; bit-identical index)
    void build_on_device(const std::string& bases, const std::vector<uint64_t>& offsets, int k, int dev = 0) {
        char err[512] = {0};
        fin_index_free(h); h = nullptr; forget();
        check(fin_index_build_device(bases.data(), offsets.data(), offsets.size() - 1, k, dev, &h, nullptr, err, sizeof err), err);
    }
    // the unitigs as parts of at most max_part_bases bases (0: 3.2e9), each an ordinary index built and resident on `dev`; throws for a set that is not a
    // disjoint spectrum-preserving string set (fin_pindex_build_device with verify)
    void build_partitioned(const std::string& bases, const std::vector<uint64_t>& offsets, int k, int dev = 0, uint64_t max_part_bases = 0) {
        char err[1024] = {0};
        fin_index_free(h); h = nullptr; forget(); fin_pindex_free(ph); ph = nullptr;
        check(fin_pindex_build_device(bases.data(), offsets.data(), offsets.size() - 1, k, dev, max_part_bases, 1, &ph, err, sizeof err), err);
    }
    void serialize(const std::string& index_prefix) const {
        char err[512] = {0};
        if (ph) check(fin_pindex_save(ph, index_prefix.c_str(), err, sizeof err), err);
        else check(fin_index_save(h, index_prefix.c_str(), err, sizeof err), err);
    }
    void load(const std::string& index_prefix) {
        char err[512] = {0};
        fin_index_free(h); h = nullptr; forget(); fin_pindex_free(ph); ph = nullptr;
        if (fin_pindex_exists(index_prefix.c_str())) check(fin_pindex_load(index_prefix.c_str(), device, &ph, err, sizeof err), err);   // (its parts' replicas are uploaded here)
        else check(fin_index_load(index_prefix.c_str(), &h, err, sizeof err), err);
    }
    void to_device() {
        if (ph) return;   // (a partitioned index lives on its one device since it was built or loaded)
        char err[512] = {0};
        check(fin_index_to_device(h, device, err, sizeof err), err);
        for (int d : devices) check(fin_index_to_device(h, d, err, sizeof err), err);
    }
    int64_t size_in_bytes() const { return ph ? fin_pindex_size_in_bytes(ph) : fin_index_size_in_bytes(h); }
    // HBM of the first replica beyond the index arrays: every table, filter and bitmap the upload derived (-1: not on a device)
    int64_t replica_table_bytes() const { if (ph) return fin_pindex_replica_table_bytes(ph); const int d = fin_index_first_device(h); return d < 0 ? -1 : fin_index_replica_table_bytes(h, d); }
    // the statistics-only modes of build-fmin (build_fmin.hh:95-214): {distinct finimizers, sum of frequencies, sum of lengths}
    void finimizer_stats(const std::string& bases, const std::vector<uint64_t>& offsets, int type, int64_t t, int64_t& n, int64_t& sum_freq, int64_t& sum_len) const {
        char err[512] = {0};
        check(fin_index_finimizer_stats(h, bases.data(), offsets.data(), offsets.size() - 1, type, t, &n, &sum_freq, &sum_len, err, sizeof err), err);
    }
    int64_t get_k() const { return ph ? fin_pindex_k(ph) : fin_index_k(h); }
    int64_t number_of_subsets() const { return ph ? fin_pindex_n_nodes(ph) : fin_index_n_nodes(h); }   // (a partitioned index: summed over its parts -- may pass 2^32)
    int64_t number_of_kmers() const { return ph ? fin_pindex_n_kmers(ph) : fin_index_n_kmers(h); }
    int64_t number_of_unitigs() const { return ph ? fin_pindex_n_unitigs(ph) : fin_index_n_unitigs(h); }
    int64_t number_of_finimizers() const {
        if (!ph) return fin_index_n_finimizers(h);
        int64_t t = 0;
        for (uint32_t p = 0; p < fin_pindex_parts(ph); p++) t += fin_index_n_finimizers(fin_pindex_part(ph, p));
        return t;
    }
    const fin_index* handle() const { return h; }

    // The members the reference class exposes publicly (FinimizerIndex.hh:108-115) and its tests read (tests.cpp:66-83,125-141,195),
    // as MEMBERS with the reference's own access syntax:
    //     *index->LCS          index->unitigs.concat   index->unitigs.ends   index->fmin   index->global_offsets (+ .width())   index->Ustart
    // Each is a read-only vector decoded from the HBM layout the first time it is touched (and again after build() / load()).
    template <typename T>
    class Member {
        friend class FinimizerIndex;
        const FinimizerIndex* owner = nullptr; int what = 0; bool as_bits = false;
        mutable std::vector<T> v; mutable bool have = false;
        const std::vector<T>& get() const { if (!have) { v = owner->template materialise<T>(what, as_bits); have = true; } return v; }
    public:
        typedef T value_type;
        const std::vector<T>& operator*() const { return get(); }      // *index->LCS (the reference holds LCS by unique_ptr)
        const std::vector<T>* operator->() const { return &get(); }
        operator const std::vector<T>&() const { return get(); }
        size_t size() const { return get().size(); }
        T operator[](size_t i) const { return get()[i]; }
        typename std::vector<T>::const_iterator begin() const { return get().begin(); }
        typename std::vector<T>::const_iterator end() const { return get().end(); }
        // sdsl::int_vector<>::width() of a bit-compressed vector (tests.cpp:135): bits of the largest value, at least 1
        int width() const { uint64_t m = 0; for (const T& x : get()) if ((uint64_t)x > m) m = (uint64_t)x; int w = 1; while (m >>= 1) w++; return w; }
    };
    struct PackedStringsMember {             // PackedStrings (PackedStrings.hh:26-29)
        Member<uint8_t> concat;              // one 0..3 code per base (A C G T), the unitigs in index order
        Member<int64_t> ends;                // exclusive end of every unitig in concat
        int64_t number_of_strings() const { return (int64_t)ends.size(); }
    };
    Member<int64_t> LCS;
    PackedStringsMember unitigs;
    Member<uint8_t> fmin;                    // one 0/1 per node
    Member<int64_t> global_offsets;
    Member<uint8_t> Ustart;
    std::vector<int64_t> C_array() const { return export_as<int64_t, int64_t>(FIN_X_C); }                 // sbwt->get_C_array()
    std::vector<bool> plane(int c) const { return bits(FIN_X_PLANE_A + c); }                              // sbwt subset rank structure, A_bits .. T_bits
    // the reference's own seven-file layout (FinimizerIndex::serialize, :187-207); parity unpinned, see fin_sdsl.cpp
    void serialize_reference_layout(const std::string& index_prefix) const {
        char err[512] = {0};
        check(fin_index_save_reference_layout(h, index_prefix.c_str(), err, sizeof err), err);
    }

private:
    void wire() {
        auto set = [this](auto& m, int what, bool as_bits) { m.owner = this; m.what = what; m.as_bits = as_bits; };
        set(LCS, FIN_X_LCS, false); set(unitigs.concat, FIN_X_CONCAT, false); set(unitigs.ends, FIN_X_ENDS, false);
        set(fmin, FIN_X_FMIN, true); set(global_offsets, FIN_X_GOFF, false); set(Ustart, FIN_X_USTART, true);
    }
    void forget() { LCS.have = unitigs.concat.have = unitigs.ends.have = fmin.have = global_offsets.have = Ustart.have = false; }
    template <typename T>
    std::vector<T> materialise(int what, bool as_bits) const {
        if (as_bits) { const std::vector<bool> b = bits(what); return std::vector<T>(b.begin(), b.end()); }
        switch (what) {
            case FIN_X_LCS: case FIN_X_CONCAT: return export_as<uint8_t, T>(what);
            default: return export_as<int64_t, T>(what);
        }
    }
    template <typename S, typename T>
    std::vector<T> export_as(int what) const {
        char err[512] = {0};
        const int64_t nbytes = fin_index_export_size(h, what);
        if (nbytes < 0) throw std::runtime_error("no index");
        std::vector<S> raw((size_t)nbytes / sizeof(S) + 1);
        check(fin_index_export(h, what, raw.data(), (uint64_t)nbytes, err, sizeof err), err);
        raw.resize((size_t)nbytes / sizeof(S));
        return std::vector<T>(raw.begin(), raw.end());
    }
    std::vector<bool> bits(int what) const {
        const std::vector<uint64_t> w = export_as<uint64_t, uint64_t>(what);
        const int64_t n = number_of_subsets();
        std::vector<bool> v((size_t)n);
        for (int64_t i = 0; i < n; i++) v[(size_t)i] = (w[(size_t)(i >> 6)] >> (i & 63)) & 1;
        return v;
    }

public:

    // FinimizerIndex::search(const std::string&) const (:119)
    QueryResult search(const std::string& query) const {
        char err[512] = {0};
        int64_t k = get_k();
        int64_t nk = (int64_t)query.size() - k + 1; if (nk < 0) nk = 0;
        std::vector<int64_t> pairs((size_t)(2 * nk + 2));
        QueryResult ans;
        if (fin_index_to_device(const_cast<fin_index*>(h), device, err, sizeof err) != FIN_OK) throw std::runtime_error(err);
        check(fin_search(h, query.data(), (int64_t)query.size(), pairs.data(), &ans.n_found, err, sizeof err), err);
        ans.local_offsets.reserve((size_t)nk);
        for (int64_t i = 0; i < nk; i++) ans.local_offsets.push_back({pairs[2 * i], pairs[2 * i + 1]});
        return ans;
    }

    // the streaming loop of search_fmin.hh:43-72 for a batch of reads (both strands merged); int32 pairs back to back
    void search_batch(const char* bases, const uint64_t* offsets, uint64_t n_reads, std::vector<int32_t>& pairs,
                      uint64_t& total_positive) const {
        char err[512] = {0};
        uint64_t nk = 0;
        const uint64_t k = (uint64_t)get_k();
        for (uint64_t r = 0; r < n_reads; r++) { uint64_t len = offsets[r + 1] - offsets[r]; if (len >= k) nk += len - k + 1; }
        pairs.resize((size_t)(2 * nk + 2));
        if (devices.size() > 1)
            check(fin_search_batch_multi(h, devices.data(), (int)devices.size(), bases, offsets, n_reads, FIN_MERGED, pairs.data(), &total_positive, err, sizeof err), err);
        else
            check(fin_search_batch(h, bases, offsets, n_reads, FIN_MERGED, pairs.data(), &total_positive, err, sizeof err), err);
        pairs.resize((size_t)(2 * nk));
    }
    // the same loop with its printed text as the result, formatted on the GPU (fin_search_batch_text); false = this batch has to be
    // formatted on the host (several GPUs in use, or a read without k-mers)
    bool search_batch_text(const char* bases, const uint64_t* offsets, uint64_t n_reads, fin_text* out, uint64_t& total_positive) const {
        if (devices.size() > 1 || ph) return false;   // (a partitioned index delivers pairs: the host formats them)
        const uint64_t k = (uint64_t)get_k();
        for (uint64_t r = 0; r < n_reads; r++) if (offsets[r + 1] - offsets[r] < k) return false;
        char err[512] = {0};
        check(fin_search_batch_text(h, bases, offsets, n_reads, FIN_MERGED, out, &total_positive, err, sizeof err), err);
        return true;
    }
    // QueryResult::n_found summed over a batch of reads searched on ONE strand as given (FinimizerIndex::search on each read): what
    // search_fmin.hh:66-67 accumulates as kmers_count (reads) and kmers_count_rev (their reverse complements).  Pairs stay on the device.
    uint64_t count_found_one_strand(const char* bases, const uint64_t* offsets, uint64_t n_reads) const {
        char err[512] = {0};
        uint64_t pos = 0;
        if (ph) throw std::runtime_error("one-strand counts are not available on a partitioned index");
        if (devices.size() > 1)
            check(fin_search_batch_multi(h, devices.data(), (int)devices.size(), bases, offsets, n_reads, FIN_FWD, nullptr, &pos, err, sizeof err), err);
        else
            check(fin_search_batch(h, bases, offsets, n_reads, FIN_FWD, nullptr, &pos, err, sizeof err), err);
        return pos;
    }
    // same, into a caller buffer of 2*(number of k-mers)+2 int32 (page-locked memory from fin_host_alloc makes the copies DMA)
    void search_batch_into(const char* bases, const uint64_t* offsets, uint64_t n_reads, int32_t* pairs, uint64_t& total_positive) const {
        char err[512] = {0};
        if (ph) check(fin_pindex_search_batch(ph, bases, offsets, n_reads, pairs, &total_positive, err, sizeof err), err);
        else if (devices.size() > 1)
            check(fin_search_batch_multi(h, devices.data(), (int)devices.size(), bases, offsets, n_reads, FIN_MERGED, pairs, &total_positive, err, sizeof err), err);
        else
            check(fin_search_batch(h, bases, offsets, n_reads, FIN_MERGED, pairs, &total_positive, err, sizeof err), err);
    }
};

// A unitig set beyond 2^32 nodes as parts (include/finito_amd.h: fin_pindex_*): the answers of ONE FinimizerIndex of all the unitigs -- unitig numbers are
// permute_unitigs' over the whole set (PackedStrings.hh:105-135) -- for the input the reference requires, a disjoint spectrum-preserving string set
// (README.md:79-80); `verify` checks that on the device and the constructor throws std::runtime_error for a set that is not.
class PartitionedFinimizerIndex {
    PartitionedFinimizerIndex(const PartitionedFinimizerIndex&) = delete;
    PartitionedFinimizerIndex& operator=(const PartitionedFinimizerIndex&) = delete;
    fin_pindex* h = nullptr;

public:
    PartitionedFinimizerIndex(const std::string& bases, const std::vector<uint64_t>& offsets, int k, int device = 0, uint64_t max_part_bases = 0, bool verify = true) {
        char err[1024] = {0};
        const int rc = fin_pindex_build_device(bases.data(), offsets.data(), offsets.size() - 1, k, device, max_part_bases, verify ? 1 : 0, &h, err, sizeof err);
        if (rc != FIN_OK) throw std::runtime_error(err[0] ? err : "fin_pindex_build_device failed");
    }
    ~PartitionedFinimizerIndex() { fin_pindex_free(h); }
    int64_t get_k() const { return fin_pindex_k(h); }
    int64_t number_of_parts() const { return (int64_t)fin_pindex_parts(h); }
    int64_t number_of_subsets() const { return fin_pindex_n_nodes(h); }     // (summed over the parts: may pass 2^32)
    int64_t number_of_kmers() const { return fin_pindex_n_kmers(h); }
    int64_t number_of_unitigs() const { return fin_pindex_n_unitigs(h); }
    int64_t size_in_bytes() const { return fin_pindex_size_in_bytes(h); }
    // run_fmin_queries_streaming's searches and merge (search_fmin.hh:46-60) over a flat read set: pairs[2 * number of k-mers], total_positive
    void search_batch_into(const char* bases, const uint64_t* offsets, uint64_t n_reads, int32_t* pairs, uint64_t& total_positive) const {
        char err[512] = {0};
        if (fin_pindex_search_batch(h, bases, offsets, n_reads, pairs, &total_positive, err, sizeof err) != FIN_OK) throw std::runtime_error(err[0] ? err : "fin_pindex_search_batch failed");
    }
};
