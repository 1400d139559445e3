// Runs the reference's known-answer tests (src/tests.cpp:62-317, fed as text by tests/test_cpp_mirror.py from
// tests/golden/reference_kat.json) through the C++ mirror finito_amd/csrc/FinimizerIndex.hh -- the same calls the reference's own
// tests make: build, the public members with the reference's own syntax (*index->LCS, index->unitigs.concat / .ends, index->fmin,
// index->global_offsets (+ .width()), index->Ustart: tests.cpp:78-83,135-141), search().
// usage: kat_mirror [--no-search] < cases.txt     (--no-search: structure checks only, needs no GPU)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../finito_amd/csrc/FinimizerIndex.hh"

static int n_checks = 0;
template <typename A, typename B>
static void assert_equal(const A& a, const B& b, const std::string& what) {   // tests.cpp:24-32
    n_checks++;
    bool same = a.size() == b.size();
    for (size_t i = 0; same && i < a.size(); i++) same = (long long)a[i] == (long long)b[i];
    if (!same) {
        std::cerr << "MISMATCH " << what << "\n  got     :";
        for (size_t i = 0; i < a.size(); i++) std::cerr << ' ' << (long long)a[i];
        std::cerr << "\n  expected:";
        for (size_t i = 0; i < b.size(); i++) std::cerr << ' ' << (long long)b[i];
        std::cerr << std::endl;
        exit(1);
    }
}
static std::vector<long long> nums(std::istringstream& in) { std::vector<long long> v; long long x; while (in >> x) v.push_back(x); return v; }

int main(int argc, char** argv) {
    const bool do_search = !(argc > 1 && !strcmp(argv[1], "--no-search"));
    std::string line, name;
    FinimizerIndex* index = nullptr;
    std::vector<std::string> unitigs; int k = 0;
    auto build = [&]() {   // build_index of tests.cpp:42-56
        if (index) return;
        std::string bases; std::vector<uint64_t> offsets{0};
        for (auto& u : unitigs) { bases += u; offsets.push_back(bases.size()); }
        index = new FinimizerIndex(0);
        index->build(bases, offsets, k, 2);
    };
    try {
        while (std::getline(std::cin, line)) {
            std::istringstream in(line);
            std::string tag; in >> tag;
            if (tag == "CASE") { in >> name >> k; unitigs.clear(); delete index; index = nullptr; }
            else if (tag == "U") { std::string u; in >> u; unitigs.push_back(u); }
            else if (tag == "LCS") { build(); assert_equal(*index->LCS, nums(in), name + ": LCS"); }
            else if (tag == "CONCAT") { build(); assert_equal(index->unitigs.concat, nums(in), name + ": unitigs.concat"); }
            else if (tag == "ENDS") { build(); assert_equal(index->unitigs.ends, nums(in), name + ": unitigs.ends"); }
            else if (tag == "FMIN") { build(); assert_equal(index->fmin, nums(in), name + ": fmin"); }
            else if (tag == "GOFF") {
                build();
                const std::vector<long long> exp = nums(in);
                assert_equal(index->global_offsets, exp, name + ": global_offsets");
                long long m = 0; for (long long x : exp) if (x > m) m = x;   // tests.cpp:135-136: the width of the bit-compressed vector
                int w = 1; while (m >>= 1) w++;
                assert_equal(std::vector<long long>{index->global_offsets.width()}, std::vector<long long>{w}, name + ": global_offsets.width()");
            }
            else if (tag == "USTART") { build(); assert_equal(index->Ustart, nums(in), name + ": Ustart"); }
            else if (tag == "CARRAY") { build(); assert_equal(index->C_array(), nums(in), name + ": C array"); }
            else if (tag == "NODES") { build(); long long n; in >> n; assert_equal(std::vector<long long>{index->number_of_subsets()}, std::vector<long long>{n}, name + ": number_of_subsets"); }
            else if (tag == "Q" && do_search) {   // query, n_found (-1: not given), pairs flat
                build();
                std::string q; long long nf; in >> q >> nf;
                const std::vector<long long> exp = nums(in);
                FinimizerIndex::QueryResult res = index->search(q);
                std::vector<long long> got;
                for (auto& p : res.local_offsets) { got.push_back(p.first); got.push_back(p.second); }
                assert_equal(got, exp, name + ": search(" + q + ")");
                if (nf >= 0) assert_equal(std::vector<long long>{res.n_found}, std::vector<long long>{nf}, name + ": n_found(" + q + ")");
            } else if (tag == "M" && do_search) {   // the streaming loop's strand merge (search_fmin.hh:47-60; tests.cpp:259-288)
                build();
                std::string q; in >> q;
                const std::vector<long long> exp = nums(in);
                std::vector<int32_t> pairs; uint64_t pos = 0;
                const uint64_t offs[2] = {0, q.size()};
                index->to_device();
                index->search_batch(q.data(), offs, 1, pairs, pos);
                assert_equal(pairs, exp, name + ": merged(" + q + ")");
            }
        }
    } catch (const std::exception& e) { std::cerr << "Runtime error: " << e.what() << std::endl; return 2; }
    delete index;
    printf("ok %d checks%s\n", n_checks, do_search ? "" : " (no search)");
    return 0;
}
