// main.cpp -- the `finito` command: same sub-commands, flags, help/exit behaviour, log lines and output text as the
// reference's `benchmark` binary (src/main.cpp:21-59, include/build_fmin.hh:302-402, include/search_fmin.hh:130-213),
// with the search running on the MI355X through the C ABI.
//
// Differences, all on the build side: the reference reads a plain-matrix SBWT produced by the external `sbwt build`
// tool (-i) and takes k from it; here the SBWT is a pure function of the unitigs and k and is rebuilt, so -i is optional:
// when given, k is taken from the file and the file (and an --lcs file) must match what was rebuilt; without it k comes
// from -k (default 31).  --type rarest (-t 1) builds an index; --type shortest / verify print the reference's finimizer
// statistics for threshold -t (build_fmin.hh:252-268).  The index is one container file <prefix>.finamd; --sdsl 1 also writes
// the reference's seven files, and search-fmin -i <prefix> loads those when there is no container (fin_sdsl.cpp).
#include <omp.h>
#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <iostream>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "FinimizerIndex.hh"

using namespace std;

static int64_t cur_time_micros() {
    return chrono::duration_cast<chrono::microseconds>(chrono::steady_clock::now().time_since_epoch()).count();
}
static void write_log(const string& msg) { cerr << msg << endl; }

// ---- FASTA / FASTQ reader, plain or gzipped (SBWT's SeqIO::Reader in the reference) ------------------------------
class SeqReader {
    gzFile f;
    vector<char> buf; size_t pos = 0, lim = 0;
    bool eof = false;
    int getc_() {
        if (pos == lim) {
            if (eof) return -1;
            int n = gzread(f, buf.data(), (unsigned)buf.size());
            if (n <= 0) { eof = true; return -1; }
            pos = 0; lim = (size_t)n;
        }
        return (unsigned char)buf[pos++];
    }
    int peek_() { int c = getc_(); if (c >= 0) pos--; return c; }
    void skip_line() {
        for (;;) {
            if (pos == lim && peek_() < 0) return;
            const char* nl = (const char*)memchr(buf.data() + pos, '\n', lim - pos);
            if (nl) { pos = (size_t)(nl - buf.data()) + 1; return; }
            pos = lim;
        }
    }
    void read_line(string& out) {
        for (;;) {
            if (pos == lim && peek_() < 0) return;
            const char* b = buf.data() + pos;
            const char* nl = (const char*)memchr(b, '\n', lim - pos);
            size_t n = nl ? (size_t)(nl - b) : lim - pos;
            size_t m = n;
            if (m && b[m - 1] == '\r') m--;
            out.append(b, m);
            pos += n + (nl ? 1 : 0);
            if (nl) return;
        }
    }

public:
    string read_buf;
    explicit SeqReader(const string& path) : buf(1 << 20) {
        f = gzopen(path.c_str(), "rb");
        if (!f) throw runtime_error("Error opening file " + path);
    }
    ~SeqReader() { if (f) gzclose(f); }
    // returns the length of the next sequence, 0 at end of file (get_next_read_to_buffer of the reference)
    int64_t get_next_read_to_buffer() {
        read_buf.clear();
        int c;
        while ((c = peek_()) >= 0 && (c == '\n' || c == '\r')) getc_();
        if (c < 0) return 0;
        if (c == '>') {
            skip_line();
            while ((c = peek_()) >= 0 && c != '>') read_line(read_buf);
        } else if (c == '@') {
            skip_line();
            read_line(read_buf);
            skip_line();   // '+'
            skip_line();   // qualities (multi-line FASTQ is not supported, as in the reference)
        } else {
            throw runtime_error("Error: input is neither FASTA nor FASTQ");
        }
        return (int64_t)read_buf.size();
    }
};

// ---- a team of host threads that SLEEP between jobs ---------------------------------------------------------------------------
// The stages of search-fmin work side by side, each with all host threads.  OpenMP teams would spin at the end of every parallel
// region (libgomp's default wait policy, fixed when the runtime loads): under a container's CPU quota the idle spinning of three
// teams gets every stage throttled (measured on a 16-core share: parser 3.1 s instead of 0.2 s).  These threads wait on a
// condition variable instead.
class Team {
    vector<thread> th; mutex mu; condition_variable cv, done_cv;
    const function<void(int, int)>* job = nullptr; uint64_t gen = 0; int pending = 0; bool quit = false;
    void worker(int t) {
        uint64_t seen = 0;
        for (;;) {
            const function<void(int, int)>* j;
            { unique_lock<mutex> g(mu); cv.wait(g, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; j = job; }
            (*j)(t, n);
            { lock_guard<mutex> g(mu); if (--pending == 0) done_cv.notify_one(); }
        }
    }
public:
    const int n;
    explicit Team(int n_) : n(n_ < 1 ? 1 : n_) { for (int t = 1; t < n; t++) th.emplace_back([this, t] { worker(t); }); }
    ~Team() { { lock_guard<mutex> g(mu); quit = true; } cv.notify_all(); for (auto& t : th) t.join(); }
    // f(t, n) on every member t of the team; returns when all are done.  An exception in a member is rethrown here.
    void run(const function<void(int, int)>& f) {
        exception_ptr err; mutex emu;
        const function<void(int, int)> safe = [&](int t, int nn) { try { f(t, nn); } catch (...) { lock_guard<mutex> g(emu); if (!err) err = current_exception(); } };
        { lock_guard<mutex> g(mu); job = &safe; pending = n - 1; gen++; }
        cv.notify_all();
        safe(0, n);
        { unique_lock<mutex> g(mu); done_cv.wait(g, [&] { return pending == 0; }); }
        if (err) rethrow_exception(err);
    }
    // [lo, hi) of member t when N items are dealt out evenly
    static pair<size_t, size_t> share(size_t N, int t, int n) { return {N * (size_t)t / (size_t)n, N * (size_t)(t + 1) / (size_t)n}; }
};

// ---- block-parallel reader for uncompressed files (f-3: the reference parses one read at a time, search_fmin.hh:43-45) -----------
// The file is read in blocks of a few hundred MB; in a block all threads look for line starts, then for records (FASTQ: groups of
// four lines, checked for their '@' and '+'; FASTA: from one '>' line to the next, sequence lines concatenated), then copy the
// sequences side by side into the caller's buffer.  A block that does not look regular (blank lines inside a FASTQ, a stray
// character) is parsed by the sequential rules of SeqReader instead, so both readers accept the same files and give the same reads.
class BlockReader {
    int fd = -1;
    const char* map = nullptr; size_t size = 0, pos = 0;   // the file, mapped; everything before pos has been parsed
    vector<vector<uint32_t>> nl; vector<uint32_t> ends;    // (kept between blocks: no allocation per block)
    Team team{fin_host_threads()};
    static bool is_gzip(const string& path) {
        unsigned char m[2] = {0, 0};
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) throw runtime_error("Error opening file " + path);
        const size_t n = fread(m, 1, 2, f); fclose(f);
        return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
    }
    // the sequential rules (SeqReader::get_next_read_to_buffer) over memory; returns bytes consumed (whole records only unless `last`)
    static size_t parse_sequential(const char* p, size_t n, bool last, char* dst, size_t& n_bases, vector<uint64_t>& offsets) {
        size_t i = 0, done = 0;
        auto line_end = [&](size_t from) -> size_t { const char* nl = (const char*)memchr(p + from, '\n', n - from); return nl ? (size_t)(nl - p) : n; };
        for (;;) {
            while (i < n && (p[i] == '\n' || p[i] == '\r')) i++;
            if (i >= n) { done = n; break; }
            const size_t rec = i;
            if (p[i] == '>') {
                size_t e = line_end(i); if (e == n && !last) { done = rec; break; }
                i = e + 1;
                const size_t b0 = n_bases;
                bool complete = false;
                for (;;) {
                    if (i >= n) { complete = last; break; }
                    if (p[i] == '>') { complete = true; break; }
                    e = line_end(i);
                    if (e == n && !last) break;
                    size_t m = e - i; if (m && p[i + m - 1] == '\r') m--;
                    memcpy(dst + n_bases, p + i, m); n_bases += m;
                    i = e + 1;
                }
                if (!complete) { n_bases = b0; done = rec; break; }
                if (n_bases > b0) offsets.push_back(n_bases);   // (an empty sequence ends the reference's loop, search_fmin.hh:44; here it is skipped)
                done = i < n ? i : n;
            } else if (p[i] == '@') {
                // header, sequence, '+', qualities; at the end of the file a truncated record still yields its sequence line
                size_t ls[4] = {0, 0, 0, 0}, le[4] = {0, 0, 0, 0}; size_t j = i; int got = 0;
                for (int l = 0; l < 4 && j <= n; l++) {
                    ls[l] = j; le[l] = j < n ? line_end(j) : n;
                    const bool has_nl = le[l] < n;
                    if (!has_nl && !last) break;   // an incomplete line: wait for more data
                    got++;
                    j = le[l] + 1;
                    if (!has_nl) break;            // the file's last line, without '\n'
                }
                if (got < 4 && !last) { done = rec; break; }
                if (got >= 2) {
                    size_t m = le[1] - ls[1]; if (m && p[ls[1] + m - 1] == '\r') m--;
                    if (m) { memcpy(dst + n_bases, p + ls[1], m); n_bases += m; offsets.push_back(n_bases); }
                }
                if (got < 4 || le[3] >= n) { done = n; break; }
                i = le[3] + 1;
                done = i;
            } else throw runtime_error("Error: input is neither FASTA nor FASTQ");
        }
        return done;
    }

public:
    // a regular, uncompressed file that can be mapped
    static bool usable(const string& path) {
        if (is_gzip(path)) return false;
        struct stat st;
        return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
    }
    explicit BlockReader(const string& path) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw runtime_error("Error opening file " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) throw runtime_error("Error opening file " + path);
        size = (size_t)st.st_size;
        if (size) {
            void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) throw runtime_error("Error mapping file " + path);
            map = (const char*)m;
            (void)madvise(m, size, MADV_SEQUENTIAL);
        }
    }
    ~BlockReader() { if (map) munmap((void*)map, size); if (fd >= 0) close(fd); }
    // Next block of reads: bases side by side into dst (room for `block_bytes` bases), offsets = {0, end of read 0, ...}.
    // Returns false at the end of the file.  dst_grow(n) must return a buffer of at least n bytes, keeping nothing.
    template <class Grow>
    bool next(size_t block_bytes, Grow dst_grow, size_t& n_bases, vector<uint64_t>& offsets) {
        n_bases = 0; offsets.assign(1, 0);
        size_t want = block_bytes;
        for (;;) {
            if (pos >= size) return false;
            const size_t n = min(want, size - pos);
            const bool eof = pos + n == size;
            char* dst = dst_grow(n);
            const char* p = map + pos;
            size_t consumed = 0;
            // ---- line starts, all threads ----
            const int nt = team.n;
            if (nl.size() < (size_t)nt) nl.resize((size_t)nt);
            for (auto& v : nl) v.clear();
            bool regular = n < 0xFFFFFFF0ull;
            if (regular) {
                team.run([&](int t, int tn) {
                    const size_t lo = n * (size_t)t / (size_t)tn, hi = n * (size_t)(t + 1) / (size_t)tn;
                    vector<uint32_t>& v = nl[(size_t)t];
                    for (const char* q = p + lo; q < p + hi;) {
                        const char* x = (const char*)memchr(q, '\n', (size_t)(p + hi - q));
                        if (!x) break;
                        v.push_back((uint32_t)(x - p)); q = x + 1;
                    }
                });
            }
            vector<size_t> base((size_t)nt + 1, 0);
            for (int t = 0; t < nt; t++) base[(size_t)t + 1] = base[(size_t)t] + nl[(size_t)t].size();
            const size_t n_nl = base[(size_t)nt];
            if (ends.size() < n_nl) ends.resize(n_nl + n_nl / 8);   // position of every '\n'
            team.run([&](int t, int) { if (!nl[(size_t)t].empty()) memcpy(ends.data() + base[(size_t)t], nl[(size_t)t].data(), nl[(size_t)t].size() * 4); });
            auto line_start = [&](size_t l) -> size_t { return l == 0 ? 0 : (size_t)ends[l - 1] + 1; };
            const size_t n_lines = n_nl + ((eof && (n_nl == 0 ? n > 0 : (size_t)ends[n_nl - 1] + 1 < n)) ? 1 : 0);   // a last line without '\n' counts at the end of the file
            auto line_len = [&](size_t l) -> size_t {
                const size_t s0 = line_start(l), e = l < n_nl ? (size_t)ends[l] : n;
                size_t m = e - s0; if (m && p[s0 + m - 1] == '\r') m--;
                return m;
            };
            bool parsed = false;
            if (regular && n_lines >= 1 && p[0] == '@') {
                // FASTQ: groups of four lines (before the end of the file only lines that end in '\n' count)
                const size_t full = n_lines / 4, rem = n_lines % 4;
                const size_t n_rec = full + ((eof && rem >= 2) ? 1 : 0);   // a truncated last record still has its sequence line
                bool ok = n_rec > 0;
                {
                    vector<char> okt((size_t)nt, 1);
                    team.run([&](int t, int tn) {
                        const auto sh = Team::share(full, t, tn);
                        bool o = true;
                        for (size_t r = sh.first; r < sh.second && o; r++) o = p[line_start(4 * r)] == '@' && p[line_start(4 * r + 2)] == '+' && line_len(4 * r + 1) > 0;
                        okt[(size_t)t] = o;
                    });
                    for (char o : okt) ok = ok && o;
                }
                if (ok && n_rec > full) ok = p[line_start(4 * full)] == '@' && line_len(4 * full + 1) > 0;
                if (ok) {
                    offsets.resize(n_rec + 1);
                    team.run([&](int t, int tn) { const auto sh = Team::share(n_rec, t, tn); for (size_t r = sh.first; r < sh.second; r++) offsets[r + 1] = line_len(4 * r + 1); });
                    for (size_t r = 0; r < n_rec; r++) offsets[r + 1] += offsets[r];
                    team.run([&](int t, int tn) {
                        const auto sh = Team::share(n_rec, t, tn);
                        for (size_t r = sh.first; r < sh.second; r++) memcpy(dst + offsets[r], p + line_start(4 * r + 1), (size_t)(offsets[r + 1] - offsets[r]));
                    });
                    n_bases = (size_t)offsets[n_rec];
                    consumed = eof ? n : line_start(4 * full);
                    parsed = true;
                }
            } else if (regular && n_lines >= 1 && p[0] == '>') {
                // FASTA: a record runs from its '>' line to the next one
                vector<vector<uint32_t>> hd((size_t)nt);
                team.run([&](int t, int tn) {
                    const auto sh = Team::share(n_lines, t, tn);
                    for (size_t l = sh.first; l < sh.second; l++) { const size_t s0 = line_start(l); if (s0 < n && p[s0] == '>') hd[(size_t)t].push_back((uint32_t)l); }
                });
                vector<uint32_t> heads;
                for (auto& v : hd) heads.insert(heads.end(), v.begin(), v.end());
                const size_t n_rec = eof ? heads.size() : (heads.empty() ? 0 : heads.size() - 1);   // the last record may go on in the next block
                if (n_rec > 0) {
                    vector<uint64_t> len(n_rec + 1, 0);
                    // (records are dealt out by their first line, so that members get about the same number of lines whatever the record sizes)
                    auto rec_share = [&](int t, int tn) -> pair<size_t, size_t> {
                        const auto ls = Team::share(n_lines, t, tn);
                        const size_t a = (size_t)(lower_bound(heads.begin(), heads.begin() + (long)n_rec, (uint32_t)ls.first) - heads.begin());
                        const size_t b = t + 1 == tn ? n_rec : (size_t)(lower_bound(heads.begin(), heads.begin() + (long)n_rec, (uint32_t)ls.second) - heads.begin());
                        return {a, b};
                    };
                    team.run([&](int t, int tn) {
                        const auto sh = rec_share(t, tn);
                        for (size_t r = sh.first; r < sh.second; r++) {
                            const size_t l1 = r + 1 < heads.size() ? heads[r + 1] : n_lines;
                            uint64_t m = 0;
                            for (size_t l = (size_t)heads[r] + 1; l < l1; l++) m += line_len(l);
                            len[r + 1] = m;
                        }
                    });
                    for (size_t r = 0; r < n_rec; r++) len[r + 1] += len[r];
                    team.run([&](int t, int tn) {
                        const auto sh = rec_share(t, tn);
                        for (size_t r = sh.first; r < sh.second; r++) {
                            const size_t l1 = r + 1 < heads.size() ? heads[r + 1] : n_lines;
                            char* d = dst + len[r];
                            for (size_t l = (size_t)heads[r] + 1; l < l1; l++) { const size_t m = line_len(l); memcpy(d, p + line_start(l), m); d += m; }
                        }
                    });
                    // reads without bases are skipped, as by the sequential rules
                    offsets.assign(1, 0);
                    bool any_empty = false;
                    for (size_t r = 0; r < n_rec; r++) any_empty |= len[r + 1] == len[r];
                    if (!any_empty) { offsets.resize(n_rec + 1); for (size_t r = 0; r <= n_rec; r++) offsets[r] = len[r]; }
                    else for (size_t r = 0; r < n_rec; r++) if (len[r + 1] > len[r]) offsets.push_back(len[r + 1]);
                    n_bases = (size_t)len[n_rec];
                    consumed = eof ? n : line_start(heads[n_rec]);
                    parsed = true;
                }
            }
            if (!parsed) consumed = parse_sequential(p, n, eof, dst, n_bases, offsets);
            pos += consumed;
            if (offsets.size() > 1) return true;
            if (eof) return false;                       // nothing but blank lines / an unfinished tail was left
            want = want + (want >> 1) + block_bytes / 4; // no complete record yet (one record larger than the block): look further
        }
    }
};

static vector<string> readlines(const string& path) {
    ifstream in(path);
    if (!in.good()) throw runtime_error("Error opening file " + path);
    vector<string> v; string line;
    while (getline(in, line)) if (!line.empty()) v.push_back(line);
    return v;
}
static void check_readable(const string& path) {
    ifstream in(path);
    if (!in.good()) throw runtime_error("Error reading file: " + path);
}
static void check_writable(const string& path) {
    ofstream out(path, ios::app);
    if (!out.good()) throw runtime_error("Error writing to file: " + path);
}

// ---- tiny option parser with cxxopts' surface for the flags the two commands use --------------------------------
struct Opts {
    map<string, string> val; bool help = false;
    bool has(const string& k) const { return val.count(k) > 0; }
    string get(const string& k, const string& def = "") const { auto it = val.find(k); return it == val.end() ? def : it->second; }
};
static Opts parse(int argc, char** argv, const map<string, string>& short_to_long, const vector<string>& longs) {
    Opts o;
    for (int i = 1; i < argc; i++) {
        string a = argv[i], key, value; bool have_value = false;
        if (a == "-h" || a == "--help") { o.help = true; continue; }
        if (a.rfind("--", 0) == 0) {
            key = a.substr(2);
            size_t eq = key.find('=');
            if (eq != string::npos) { value = key.substr(eq + 1); key = key.substr(0, eq); have_value = true; }
        } else if (a.size() >= 2 && a[0] == '-') {
            string s = a.substr(1, 1);
            auto it = short_to_long.find(s);
            if (it == short_to_long.end()) throw runtime_error("Option '" + a + "' does not exist");
            key = it->second;
            if (a.size() > 2) { value = a.substr(2); have_value = true; }
        } else throw runtime_error("Unexpected argument: " + a);
        bool known = false;
        for (auto& l : longs) known |= l == key;
        if (!known) throw runtime_error("Option '" + a + "' does not exist");
        if (!have_value) {
            if (i + 1 >= argc) throw runtime_error("Option '" + a + "' is missing an argument");
            value = argv[++i];
        }
        o.val[key] = value;
    }
    return o;
}

static const char* BUILD_HELP =
    "Find all Finimizers of all input reads.\nUsage:\n  build-fmin [OPTION...]\n\n"
    "  -o, --out-file arg    Output index filename prefix.\n"
    "  -i, --index-file arg  SBWT file of the unitigs (plain-matrix, from `sbwt build`). Optional here: the SBWT is rebuilt\n"
    "                        from the unitigs; when given, k is taken from it and it is checked against the rebuilt one.\n"
    "  -u, --in-file arg     The SPSS in FASTA or FASTQ format, possibly gzipped. Multi-line FASTQ is not\n"
    "                        supported. If the file extension is .txt, this is interpreted as a list of\n"
    "                        files, one per line.\n"
    "  -k arg                k-mer length when no SBWT file is given (default: 31)\n"
    "      --type arg        Decide which streaming search type you prefer. Available types:  rarest shortest verify. The latter two only provide some stats. (default: rarest)\n"
    "  -t arg                Maximum finimizer frequency (default: 1)\n"
    "      --lcs arg         LCS file of the SBWT; checked against the recomputed LCS. (default: \"\")\n"
    "      --sdsl arg        1: also write the reference's own index files <prefix>.*.sdsl + <prefix>.sbwt\n"
    "      --parts-max-bases N  build a PARTITIONED index: parts of at most N bases of unitigs, each an ordinary index (automatic above 4e9 bases:\n"
    "                          one index holds fewer than 2^32 nodes); needs a GPU and a disjoint spectrum-preserving string set\n"
    "      --device-build arg  0: build on the host; 1: build on the GPU or fail; default: GPU when there is one\n"
    "      --device arg      HIP device ordinal of the device build (default: 0)\n"
    "      --threads arg     Host threads for construction (default: all)\n"
    "  -h, --help            Print usage\n";

static const char* SEARCH_HELP =
    "Query all Finimizers of all input reads.\nUsage:\n  search-fmin [OPTION...]\n\n"
    "  -o, --out-file arg    Output filename, or stdout if not given.\n"
    "  -i, --index-file arg  Index filename prefix.\n"
    "  -q, --query-file arg  The query in FASTA or FASTQ format, possibly gzipped. Multi-line FASTQ is not\n"
    "                        supported. If the file extension is .txt, this is interpreted as a list of\n"
    "                        query files, one per line. In this case, --out-file is also interpreted as a\n"
    "                        list of output files in the same manner, one line for each input file.\n"
    "      --device arg      first HIP device ordinal (default: 0)\n"
    "      --gpus arg        number of GPUs to shard the reads over, index replicated (default: all visible)\n"
    "      --strand-counts arg  1: also count the k-mers found on each strand by itself, as the reference logs them (\"Found kmers\",\n"
    "                        \"Found kmers reverse\") and sums them into <index>.stats: two more search passes per read (default: 0 --\n"
    "                        the stats field is then the merged count)\n"
    "  -h, --help            Print usage\n";

static int build_fmin(int argc, char** argv) {
    Opts o = parse(argc, argv, {{"o", "out-file"}, {"i", "index-file"}, {"u", "in-file"}, {"t", "t"}, {"k", "k"}},
                   {"out-file", "index-file", "in-file", "type", "t", "lcs", "k", "threads", "sdsl", "device-build", "device", "parts-max-bases"});
    if (argc == 1 || o.help) { cerr << BUILD_HELP << endl; exit(1); }
    if (!o.has("in-file")) throw runtime_error("Option 'in-file' has no value");
    if (!o.has("out-file")) throw runtime_error("Option 'out-file' has no value");
    int64_t t = stoll(o.get("t", "1"));
    string type = o.get("type", "rarest");
    if (type != "rarest" && type != "shortest" && type != "verify") {   // build_fmin.hh:363-367
        cerr << "Error: unknown type: " << type << endl << "Available types are: rarest shortest verify" << endl; return 1;
    }
    if (type == "rarest" && t != 1) throw runtime_error("t != 1 does not make sense with rarest type");   // build_fmin.hh:245-247
    if (t < 1) throw runtime_error("t must be at least 1");
    // k: the reference takes it from the SBWT file given with -i (build_fmin.hh:363-364).  Here the SBWT is rebuilt from the unitigs, so
    // -i is optional; when it is given, k comes from it (a -k that disagrees is an error) and the file is checked against what was built.
    int k = stoi(o.get("k", "31"));
    const string sbwt_file = o.get("index-file"), lcs_file = o.get("lcs");
    if (!sbwt_file.empty()) {
        int64_t kf = 0, nn = 0, nk = 0; char err[512] = {0};
        if (fin_sbwt_file_info(sbwt_file.c_str(), &kf, &nn, &nk, err, sizeof err) != FIN_OK) throw runtime_error(string("Error loading index from file: ") + err);
        if (o.has("k") && (int64_t)k != kf) throw runtime_error("-k " + to_string(k) + " does not match the SBWT file (k = " + to_string(kf) + ")");
        k = (int)kf;
        write_log("Loading the index variant plain-matrix");
    }
    string in_file = o.get("in-file");
    vector<string> input_files;
    if (in_file.size() >= 4 && in_file.substr(in_file.size() - 4) == ".txt") input_files = readlines(in_file);
    else input_files = {in_file};
    for (auto& f : input_files) check_readable(f);
    string out_prefix = o.get("out-file");

    string bases; vector<uint64_t> offsets{0};
    for (auto& f : input_files) {
        write_log("Searching Finimizers from input file " + f + " to index prefix " + out_prefix);
        SeqReader reader(f);
        while (reader.get_next_read_to_buffer() > 0) { bases += reader.read_buf; offsets.push_back(bases.size()); }
    }
    FinimizerIndex index;
    // A unitig set beyond one index's 2^32 nodes (or --parts-max-bases N): a PARTITIONED index -- parts of at most N bases, each an ordinary index built and kept
    // on the GPU; search-fmin finds it by its manifest <prefix>.finparts.  Needs a GPU; the set must be what the reference requires, a disjoint spectrum-
    // preserving string set (README.md:79-80), which the build checks; only type rarest.
    const uint64_t parts_max = o.has("parts-max-bases") ? stoull(o.get("parts-max-bases")) : 0;
    if (parts_max || bases.size() >= 4000000000ull) {
        if (type != "rarest") throw runtime_error("a partitioned index (more than 4e9 bases of unitigs, or --parts-max-bases) is built for type rarest only");
        if (!sbwt_file.empty() || !lcs_file.empty()) throw runtime_error("-i / --lcs cannot be checked against a partitioned index");
        index.build_partitioned(bases, offsets, k, stoi(o.get("device", "0")), parts_max);
        write_log("Partitioned index built on the GPU: " + to_string(index.number_of_parts()) + " parts");
        write_log("#SBWT nodes: " + to_string(index.number_of_subsets()));
        write_log("#Distinct finimizers: " + to_string(index.number_of_finimizers()));
        index.serialize(out_prefix);
        ofstream pstats(out_prefix + "_stats.txt", ios::app);   // build_fmin.hh:386-399
        if (pstats.is_open()) {
            pstats << to_string(t) + "," << index.number_of_finimizers() << "," << index.number_of_finimizers() << ",1.000000,," << index.number_of_kmers() << "\n";
            cout << "String appended to the file successfully." << endl;
        } else cerr << "Error: Unable to open file." << endl;
        return 0;
    }
    // the device builder when there is a GPU and k <= 64 (the same index, bit for bit, about 30 times sooner: fin_build_gpu.hip), unless
    // --device-build 0; else the host builder
    const string want_dev = o.get("device-build", "auto");
    bool on_device = false;
    if (want_dev != "0" && want_dev != "false" && fin_device_count() > 0) {
        try { index.build_on_device(bases, offsets, k, stoi(o.get("device", "0"))); on_device = true; }
        catch (const exception& e) { if (want_dev != "auto") throw; write_log(string("device build failed (") + e.what() + "), using the host builder"); }
    }
    if (!on_device) index.build(bases, offsets, k, stoi(o.get("threads", "0")));
    write_log(on_device ? "Index built on the GPU" : "Index built on the host");
    if (!sbwt_file.empty() || !lcs_file.empty()) {   // -i / --lcs: must be the SBWT / LCS of these unitigs
        char err[512] = {0};
        if (fin_index_check_against_files(index.handle(), sbwt_file.c_str(), lcs_file.c_str(), err, sizeof err) != FIN_OK) throw runtime_error(err);
        if (!lcs_file.empty()) cerr << "LCS_file loaded" << endl;
    }
    if (type != "rarest") {
        // statistics only, no index is written (build_fmin.hh:252-268); the log and the stats line are print_finimizer_stats' (common.hh:188-206)
        int64_t nf = 0, sum_freq = 0, sum_len = 0;
        index.finimizer_stats(bases, offsets, type == "shortest" ? FIN_STATS_SHORTEST : FIN_STATS_VERIFY, t, nf, sum_freq, sum_len);
        const string result = to_string(nf) + "," + to_string(sum_freq) + "," + to_string((float)sum_freq / (float)nf) + "," +
                              to_string((float)sum_len / (float)nf) + "," + to_string(index.number_of_kmers());
        write_log(to_string(t) + "," + result);
        write_log("#SBWT nodes: " + to_string(index.number_of_subsets()));
        write_log("#Distinct finimizers: " + to_string(nf));
        write_log("Sum of frequencies: " + to_string(sum_freq));
        write_log("Avg frequency: " + to_string((float)sum_freq / (float)nf));
        write_log("Avg length: " + to_string((float)sum_len / (float)nf));
        ofstream stats(out_prefix + "_stats.txt", ios::app);   // build_fmin.hh:386-399
        if (stats.is_open()) { stats << to_string(t) + "," << result << "\n"; cout << "String appended to the file successfully." << endl; }
        else cerr << "Error: Unable to open file." << endl;
        return 0;
    }
    write_log("#SBWT nodes: " + to_string(index.number_of_subsets()));
    write_log("#Distinct finimizers: " + to_string(index.number_of_finimizers()));
    index.serialize(out_prefix);
    if (o.has("sdsl") && o.get("sdsl") != "0" && o.get("sdsl") != "false") index.serialize_reference_layout(out_prefix);   // + the reference's own seven files (FinimizerIndex.hh:187-207)
    ofstream stats(out_prefix + "_stats.txt", ios::app);   // build_fmin.hh:386-399
    if (stats.is_open()) {
        stats << to_string(t) + "," << index.number_of_finimizers() << "," << index.number_of_finimizers() << ",1.000000,,"
              << index.number_of_kmers() << "\n";
        cout << "String appended to the file successfully." << endl;
    } else cerr << "Error: Unable to open file." << endl;
    return 0;
}

// ---- the streaming query loop (search_fmin.hh:33-84) as a three-stage host pipeline ------------------------------------
// parse (one thread) -> search (GPU, fin_search_batch on page-locked buffers) -> format + write (all host threads), chunks of
// ~256 MB of bases cycling through the stages, so the file parser, the PCIe/GPU work and the text formatter overlap.
struct PinnedBuf {   // page-locked host memory that only grows
    void* p = nullptr; size_t cap = 0;
    char* get(size_t n) {
        if (n > cap) {
            fin_host_free(p); p = nullptr; cap = 0;
            const size_t want = n + n / 8 + 4096;
            p = fin_host_alloc(want);
            if (!p) throw runtime_error("page-locked host allocation failed (no HIP device?)");
            cap = want;
        }
        return (char*)p;
    }
    ~PinnedBuf() { fin_host_free(p); }
};
// page-locked buffers made ready while the index loads (page-locking costs about 0.15 s per GB): the chunks of the first file pick
// them up instead of allocating in the pipeline
struct Prewarmed { vector<fin_text*> texts; vector<pair<void*, size_t>> bases; mutex mu; } g_prewarmed;
struct Chunk {
    PinnedBuf bases, pairs;
    fin_text* text = nullptr; bool as_text = false;   // the chunk's output text when the GPU formatted it
    ~Chunk() { fin_text_free(text); }
    size_t n_bases = 0;
    vector<uint64_t> offsets, pair_off;
    uint64_t positive = 0, positive_fwd = 0, positive_rev = 0;   // merged; --strand-counts: search(read) / search(rc(read)) each by itself
    vector<char> rc;                                             // --strand-counts: the chunk's reads reverse-complemented
    bool failed = false;
};
template <class T>
class BlockingQueue {
    mutex mu; condition_variable cv; deque<T> q;
public:
    void push(T v) { { lock_guard<mutex> g(mu); q.push_back(v); } cv.notify_one(); }
    T pop() { unique_lock<mutex> g(mu); cv.wait(g, [&] { return !q.empty(); }); T v = q.front(); q.pop_front(); return v; }
};
struct OutSink {   // regular files are written by all threads at once, anything else sequentially
    int fd = 1; bool seekable = false; uint64_t pos = 0; bool own = false;
    explicit OutSink(const string* path) {
        if (path) {
            fd = open(path->c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) throw runtime_error("Error writing to file: " + *path);
            own = true;
        }
        struct stat st;
        seekable = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && own;
    }
    ~OutSink() { if (own) close(fd); }
    static void write_all(int fd, const char* p, size_t n, int64_t at) {
        while (n) {
            ssize_t w = at >= 0 ? pwrite(fd, p, n, (off_t)at) : write(fd, p, n);
            if (w < 0) { if (errno == EINTR) continue; throw runtime_error(string("write failed: ") + strerror(errno)); }
            p += w; n -= (size_t)w; if (at >= 0) at += w;
        }
    }
};

// --strand-counts 1: kmers_count / kmers_count_rev of search_fmin.hh:66-67 -- the hits of search(read) and of search(rc(read)) each by
// itself, which the merged pairs do not show (a forward hit hides the reverse strand's) -- from two forward-only passes per chunk
static bool g_strand_counts = false;

static int64_t run_fmin_queries_streaming(SeqReader* reader, BlockReader* breader, OutSink& out, const FinimizerIndex& index, const string& stats_filename) {
    const int64_t k = index.get_k();
    const bool gpu_text = getenv("FINITO_HOST_FORMAT") == nullptr;   // (FINITO_HOST_FORMAT=1: format the text on the host as round 1 did)
    // chunks of 48 MB of bases (about 4*10^7 k-mers, half a GB of text), four in flight: page-locking host memory costs about
    // 0.15 s per GB, so the stages' buffers are kept small and reused
    const size_t BATCH_BASES = 48u << 20;
    constexpr int N_CHUNKS = 4;
    Chunk chunks[N_CHUNKS];
    BlockingQueue<Chunk*> free_q, search_q, format_q;
    for (auto& c : chunks) free_q.push(&c);
    exception_ptr first_error; mutex err_mu;
    auto note_error = [&]() { lock_guard<mutex> g(err_mu); if (!first_error) first_error = current_exception(); };
    atomic<bool> stop{false};
    atomic<int64_t> t_parse{0}, t_search{0}, t_write{0};   // busy time of the three stages (FINITO_TIMING=1 prints them)

    // stage 1: parse reads into page-locked chunks
    thread parser([&]() {
        bool more = true;
        try {
            while (breader && more && !stop.load()) {   // uncompressed input: whole blocks, parsed by all threads
                Chunk* c = free_q.pop();
                c->failed = false; c->n_bases = 0; c->positive = 0; c->positive_fwd = 0; c->positive_rev = 0;
                if (!c->bases.p) {   // a page-locked buffer made ready beside the index load, if one is there by now
                    lock_guard<mutex> g(g_prewarmed.mu);
                    if (!g_prewarmed.bases.empty()) { c->bases.p = g_prewarmed.bases.back().first; c->bases.cap = g_prewarmed.bases.back().second; g_prewarmed.bases.pop_back(); }
                }
                const int64_t tp0 = cur_time_micros();
                more = breader->next(BATCH_BASES, [&](size_t nbytes) { return c->bases.get(nbytes); }, c->n_bases, c->offsets);
                t_parse += cur_time_micros() - tp0;
                if (more) search_q.push(c); else free_q.push(c);
            }
            while (!breader && more && !stop.load()) {
                Chunk* c = free_q.pop();
                c->failed = false; c->n_bases = 0; c->offsets.assign(1, 0); c->positive = 0; c->positive_fwd = 0; c->positive_rev = 0;
                char* dst = c->bases.get(BATCH_BASES);
                for (;;) {
                    const int64_t len = reader->get_next_read_to_buffer();
                    if (len == 0) { more = false; break; }
                    if (c->n_bases + (size_t)len > c->bases.cap) {   // a read longer than the room that is left: enlarge, keeping the content
                        PinnedBuf bigger; char* nd = bigger.get(c->n_bases + (size_t)len + BATCH_BASES / 4);
                        memcpy(nd, dst, c->n_bases);
                        swap(bigger.p, c->bases.p); swap(bigger.cap, c->bases.cap);
                        dst = nd;
                    }
                    memcpy(dst + c->n_bases, reader->read_buf.data(), (size_t)len);
                    c->n_bases += (size_t)len; c->offsets.push_back(c->n_bases);
                    if (c->n_bases >= BATCH_BASES) break;
                }
                if (c->offsets.size() > 1) search_q.push(c); else free_q.push(c);
            }
        } catch (...) { note_error(); stop = true; }
        search_q.push(nullptr);
    });

    // stage 2: the GPU search; results land in the chunk's page-locked pair buffer
    int64_t search_wait_micros = 0, t_first = -1;
    thread searcher([&]() {
        for (;;) {
            const int64_t tw = cur_time_micros();
            Chunk* c = search_q.pop();
            if (t_first >= 0) search_wait_micros += cur_time_micros() - tw;
            if (!c) break;
            if (t_first < 0) t_first = cur_time_micros();
            const int64_t ts0 = cur_time_micros();
            try {
                if (!stop.load()) {
                    const uint64_t n_reads = c->offsets.size() - 1;
                    c->pair_off.resize(n_reads + 1); c->pair_off[0] = 0;
                    for (uint64_t r = 0; r < n_reads; r++) {
                        const int64_t len = (int64_t)(c->offsets[r + 1] - c->offsets[r]);
                        c->pair_off[r + 1] = c->pair_off[r] + (uint64_t)(len >= k ? len - k + 1 : 0);
                    }
                    // the text comes from the GPU when it can (one device, every read has a k-mer), else the pairs do
                    if (!c->text) {
                        lock_guard<mutex> g(g_prewarmed.mu);
                        if (!g_prewarmed.texts.empty()) { c->text = g_prewarmed.texts.back(); g_prewarmed.texts.pop_back(); }
                    }
                    if (!c->text) c->text = fin_text_create();
                    c->as_text = gpu_text && c->text && index.search_batch_text(c->bases.get(0), c->offsets.data(), n_reads, c->text, c->positive);
                    if (!c->as_text) {
                        int32_t* pairs = (int32_t*)c->pairs.get((size_t)(2 * c->pair_off[n_reads] + 2) * sizeof(int32_t));
                        index.search_batch_into(c->bases.get(0), c->offsets.data(), n_reads, pairs, c->positive);
                    }
                    if (g_strand_counts) {
                        c->positive_fwd = index.count_found_one_strand(c->bases.get(0), c->offsets.data(), n_reads);
                        // rc(read) for every read (sbwt::get_rc, search_fmin.hh:50), same offsets
                        c->rc.resize(c->n_bases + 1);
                        const char* src = c->bases.get(0);
#pragma omp parallel for schedule(static)
                        for (int64_t r = 0; r < (int64_t)n_reads; r++) {
                            const uint64_t a = c->offsets[(size_t)r], b = c->offsets[(size_t)r + 1];
                            for (uint64_t i = a; i < b; i++) {
                                const char ch = src[b - 1 - (i - a)];
                                char o2 = ch;
                                switch (ch) { case 'A': o2 = 'T'; break; case 'C': o2 = 'G'; break; case 'G': o2 = 'C'; break; case 'T': o2 = 'A'; break;
                                              case 'a': o2 = 't'; break; case 'c': o2 = 'g'; break; case 'g': o2 = 'c'; break; case 't': o2 = 'a'; break; default: break; }
                                c->rc[(size_t)i] = o2;
                            }
                        }
                        c->positive_rev = index.count_found_one_strand(c->rc.data(), c->offsets.data(), n_reads);
                    }
                } else c->failed = true;
            } catch (...) { note_error(); stop = true; c->failed = true; }
            t_search += cur_time_micros() - ts0;
            format_q.push(c);
        }
        format_q.push(nullptr);
    });

    // stage 3 (this thread + OpenMP team): text "(u,p) (u,p)...\n" per read, search_fmin.hh:62-65, written in input order
    int64_t number_of_queries = 0; uint64_t total_positive = 0, kmers_count = 0, kmers_count_rev = 0;
    Team team(fin_host_threads());
    const int nt = team.n;
    vector<unique_ptr<char[]>> part(nt); vector<size_t> part_cap(nt, 0), part_len(nt, 0);
    int64_t t_last = -1;
    for (;;) {
        Chunk* c = format_q.pop();
        if (!c) break;
        const int64_t tw0 = cur_time_micros();
        if (!c->failed && !stop.load()) {
            try {
                const uint64_t n_reads = c->offsets.size() - 1;
                const vector<uint64_t>& pair_off = c->pair_off;
                number_of_queries += (int64_t)pair_off[n_reads];
                total_positive += c->positive; kmers_count += c->positive_fwd; kmers_count_rev += c->positive_rev;
                if (c->as_text) {   // already text: all threads write their slice of it
                    const char* tp = fin_text_data(c->text); const uint64_t tn = fin_text_size(c->text);
                    // (pwrite by all threads; appending through a shared mapping was tried and is slower: 1.1 s against 0.9 s for 5.2 GB
                    //  on tmpfs, 2.3 s against 0.7 s on a disk-backed file)
                    if (out.seekable) {
                        team.run([&](int t, int tt) { const auto sh = Team::share((size_t)tn, t, tt); OutSink::write_all(out.fd, tp + sh.first, sh.second - sh.first, (int64_t)(out.pos + sh.first)); });
                        out.pos += tn;
                    } else OutSink::write_all(out.fd, tp, (size_t)tn, -1);
                    t_last = cur_time_micros();
                    t_write += t_last - tw0;
                    free_q.push(c);
                    continue;
                }
                const int32_t* pairs = (const int32_t*)c->pairs.get(0);
                vector<uint64_t> cut(nt + 1, n_reads);
                cut[0] = 0;
                for (int t = 1; t < nt; t++)
                    cut[t] = (uint64_t)(lower_bound(pair_off.begin(), pair_off.end(), pair_off[n_reads] * (uint64_t)t / (uint64_t)nt) - pair_off.begin());
                for (int t = 1; t <= nt; t++) cut[t] = min<uint64_t>(max(cut[t], cut[t - 1]), n_reads);
                team.run([&](int t, int) {
                    const uint64_t lo = cut[t], hi = cut[t + 1];
                    const size_t need = (size_t)(pair_off[hi] - pair_off[lo]) * 24 + 2 * (size_t)(hi - lo) + 16;
                    if (need > part_cap[t]) { part[t].reset(new char[need + need / 8]); part_cap[t] = need + need / 8; }
                    char* q = part[t].get();
                    for (uint64_t r = lo; r < hi; r++) q += fin_format_pairs(pairs + 2 * pair_off[r], (int64_t)(pair_off[r + 1] - pair_off[r]), q);
                    part_len[t] = (size_t)(q - part[t].get());
                });
                if (out.seekable)
                    team.run([&](int t, int) {
                        uint64_t at = out.pos;
                        for (int i = 0; i < t; i++) at += part_len[i];
                        OutSink::write_all(out.fd, part[t].get(), part_len[t], (int64_t)at);
                    });
                if (out.seekable) { for (int t = 0; t < nt; t++) out.pos += part_len[t]; }
                else for (int t = 0; t < nt; t++) OutSink::write_all(out.fd, part[t].get(), part_len[t], -1);
            } catch (...) { note_error(); stop = true; }
        }
        t_last = cur_time_micros();
        t_write += t_last - tw0;
        free_q.push(c);
    }
    parser.join(); searcher.join();
    if (getenv("FINITO_TIMING"))
        cerr << "[timing] stage busy seconds: parse " << t_parse.load() * 1e-6 << "  search(+PCIe" << (gpu_text ? "+GPU text" : "") << ") " << t_search.load() * 1e-6
             << "  " << (gpu_text ? "write " : "format+write ") << t_write.load() * 1e-6 << endl;
    if (first_error) rethrow_exception(first_error);
    // the reference's timed region is search + formatting + printing per read; here: first chunk entering the search until the last
    // chunk is written, minus the time the search stage sat waiting for the parser
    int64_t total_micros = t_first >= 0 && t_last >= 0 ? t_last - t_first - search_wait_micros : 0;
    if (total_micros < 0) total_micros = 0;
    write_log("k " + to_string(k));
    write_log("us/query: " + to_string(number_of_queries ? (double)total_micros / (double)number_of_queries : 0.0) + " (excluding I/O etc)");
    if (g_strand_counts) {   // search_fmin.hh:75-76
        write_log("Found kmers: " + to_string(kmers_count));
        write_log("Found kmers reverse : " + to_string(kmers_count_rev));
    }
    write_log("Total found kmers: " + to_string(total_positive));
    ofstream statsfile(stats_filename, ios::app);
    // (the reference's second field is kmers_count + kmers_count_rev, search_fmin.hh:81; without --strand-counts: the merged count)
    statsfile << to_string(k) + "," + to_string(g_strand_counts ? kmers_count + kmers_count_rev : total_positive) + "," + to_string(number_of_queries);
    return number_of_queries;
}

static int search_fmin(int argc, char** argv) {
    int64_t micros_start = cur_time_micros();
    Opts o = parse(argc, argv, {{"o", "out-file"}, {"i", "index-file"}, {"q", "query-file"}}, {"out-file", "index-file", "query-file", "device", "gpus", "strand-counts"});
    if (argc == 1 || o.help) { cerr << SEARCH_HELP << endl; exit(1); }
    g_strand_counts = o.has("strand-counts") && o.get("strand-counts") != "0" && o.get("strand-counts") != "false";
    if (!o.has("query-file")) throw runtime_error("Option 'query-file' has no value");
    if (!o.has("index-file")) throw runtime_error("Option 'index-file' has no value");
    string queryfile = o.get("query-file");
    vector<string> query_files;
    bool multi_file = queryfile.size() >= 4 && queryfile.substr(queryfile.size() - 4) == ".txt";
    if (multi_file) query_files = readlines(queryfile); else query_files = {queryfile};
    for (auto& f : query_files) check_readable(f);
    optional<vector<string>> output_files;
    if (o.has("out-file")) {
        string outfile = o.get("out-file");
        if (multi_file) output_files = readlines(outfile); else output_files = vector<string>{outfile};
        for (auto& f : output_files.value()) check_writable(f);
    } else write_log("No output file given, writing to stdout");
    if (output_files.has_value() && output_files.value().size() != query_files.size())
        throw runtime_error("Number of input and output files does not match (" + to_string(query_files.size()) + " vs " +
                            to_string(output_files.value().size()) + ")");
    string index_prefix = o.get("index-file");
    cerr << "Loading index..." << endl;
    const int first_dev = stoi(o.get("device", "0"));
    // beside the index load: page-lock the pipeline's buffers (four chunks of 48 MB of bases and of up to 16 bytes of text per k-mer)
    // -- as many as the query files can fill (a run on a few reads must not pin 3 GB), none for a multi-GPU run (its text is formatted on
    // the host), and no more once the searches are done
    atomic<bool> prewarm_stop{false};
    int prewarm_chunks = 4;
    {
        uint64_t qbytes = 0;
        for (auto& f : query_files) { struct stat sb; if (stat(f.c_str(), &sb) == 0) qbytes += (uint64_t)sb.st_size; else qbytes += 1ull << 32; }
        const bool gz = !query_files.empty() && query_files[0].size() > 3 && query_files[0].substr(query_files[0].size() - 3) == ".gz";
        const uint64_t est = gz ? qbytes * 4 : qbytes;   // (bases are at most the file's size; a gzip file inflates about fourfold)
        prewarm_chunks = (int)min<uint64_t>(4, (est + (48u << 20) - 1) / (48u << 20));
        const int want_gpus = o.has("gpus") ? stoi(o.get("gpus")) : fin_device_count() - first_dev;
        if (want_gpus > 1) prewarm_chunks = 0;
    }
    thread prewarm([&]() {
        if (getenv("FINITO_HOST_FORMAT")) return;
        for (int i = 0; i < prewarm_chunks && !prewarm_stop.load(); i++) {
            const size_t nb = (48u << 20) + (48u << 20) / 8 + 4096;
            void* p = fin_host_alloc(nb);
            fin_text* t = fin_text_create();
            if (t && fin_text_reserve(t, 16ull * (48u << 20)) != FIN_OK) { fin_text_free(t); t = nullptr; }
            lock_guard<mutex> g(g_prewarmed.mu);
            if (p) g_prewarmed.bases.push_back({p, nb});
            if (t) g_prewarmed.texts.push_back(t);
        }
    });
    struct Joiner { thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{prewarm};   // (an exception below must not leave the thread running)
    FinimizerIndex index(first_dev);
    int ngpus = o.has("gpus") ? stoi(o.get("gpus")) : fin_device_count() - first_dev;
    if (ngpus > 1) index.use_devices(first_dev, ngpus);
    const int64_t t_l0 = cur_time_micros();
    index.load(index_prefix);
    const int64_t t_l1 = cur_time_micros();
    index.to_device();
    if (getenv("FINITO_TIMING"))
        cerr << "[timing] startup seconds: until load " << (t_l0 - micros_start) * 1e-6 << "  index load " << (t_l1 - t_l0) * 1e-6 << "  upload + tables (first HIP call) "
             << (cur_time_micros() - t_l1) * 1e-6 << endl;
    (void)fin_set_option("pipeline_kmers", 1 << 24);   // three sub-batches per chunk: upload, search and download overlap inside a chunk too
    if (ngpus > 1) cerr << "Reads sharded by record over " << ngpus << " GPUs (index replicated)" << endl;
    cerr << "Index loaded" << endl;
    const int64_t k = index.get_k();
    cerr << "k = " << to_string(k) << " SBWT nodes: " << to_string(index.number_of_subsets()) << " kmers: " << to_string(index.number_of_kmers()) << endl;
    int64_t number_of_queries = 0;
    for (size_t i = 0; i < query_files.size(); i++) {
        write_log("Running streaming queries from input file " + query_files[i]);
        OutSink out(output_files.has_value() ? &output_files.value()[i] : nullptr);
        if (BlockReader::usable(query_files[i])) {
            BlockReader breader(query_files[i]);
            number_of_queries += run_fmin_queries_streaming(nullptr, &breader, out, index, index_prefix + ".stats");
        } else {
            SeqReader reader(query_files[i]);
            number_of_queries += run_fmin_queries_streaming(&reader, nullptr, out, index, index_prefix + ".stats");
        }
    }
    prewarm_stop = true;
    if (prewarm.joinable()) prewarm.join();
    for (fin_text* t : g_prewarmed.texts) fin_text_free(t);
    for (auto& b : g_prewarmed.bases) fin_host_free(b.first);
    g_prewarmed.texts.clear(); g_prewarmed.bases.clear();
    int64_t new_total_micros = cur_time_micros() - micros_start;
    write_log("us/query end-to-end: " + to_string((double)new_total_micros / (double)number_of_queries));
    write_log("total number of queries: " + to_string(number_of_queries));
    ofstream statsfile2(index_prefix + "stats.txt", ios::app);   // sic: no dot, search_fmin.hh:197
    statsfile2 << "," + to_string((double)new_total_micros / (double)number_of_queries);
    int64_t bytes = index.size_in_bytes();
    write_log("bytes: " + to_string(bytes));
    {   // (the reference's two figures are the INDEX's -- size_in_bytes(), FinimizerIndex.hh:244-258 -- and stay so in the stats file;
        //  what a replica of it occupies in HBM with every table the upload derives is logged beside them)
        const int64_t tables = index.replica_table_bytes();
        if (tables >= 0) {
            write_log("bytes in HBM per replica (index + derived tables): " + to_string(bytes + tables));
            write_log("bits per k-mer in HBM per replica: " + to_string(static_cast<double>((bytes + tables) * 8) / (double)index.number_of_kmers()));
        }
    }
    statsfile2 << "," + to_string(bytes);
    statsfile2 << "," + to_string(static_cast<double>(bytes * 8) / (double)index.number_of_kmers()) + "\n";
    statsfile2 << "," + to_string(index.number_of_kmers()) + "\n";
    return 0;
}

// diagnostic (not a reference command): the reads a file yields, one per line -- `finito parse-reads <file> [seq|block] [block bytes]`;
// tests compare the block-parallel reader with the sequential one on irregular files
static int parse_reads(int argc, char** argv) {
    if (argc < 2) { cerr << "usage: parse-reads <file> [seq|block] [block bytes]" << endl; return 1; }
    const string path = argv[1], which = argc > 2 ? argv[2] : "block";
    const size_t block = argc > 3 ? (size_t)stoull(argv[3]) : (256u << 20);
    if (which == "seq" || !BlockReader::usable(path)) {
        SeqReader r(path);
        while (r.get_next_read_to_buffer() > 0) { fwrite(r.read_buf.data(), 1, r.read_buf.size(), stdout); fputc('\n', stdout); }
    } else {
        BlockReader r(path);
        vector<char> dst; size_t nb = 0; vector<uint64_t> offs;
        const bool quiet = getenv("FINITO_PARSE_QUIET") != nullptr;   // timing only
        const int64_t t0 = cur_time_micros(); uint64_t n_reads = 0, n_bases = 0;
        while (r.next(block, [&](size_t n) { if (dst.size() < n) dst.resize(n); return dst.data(); }, nb, offs)) {
            n_reads += offs.size() - 1; n_bases += nb;
            if (!quiet) for (size_t i = 0; i + 1 < offs.size(); i++) { fwrite(dst.data() + offs[i], 1, (size_t)(offs[i + 1] - offs[i]), stdout); fputc('\n', stdout); }
        }
        if (quiet) cerr << n_reads << " reads, " << n_bases << " bases in " << (cur_time_micros() - t0) * 1e-6 << " s" << endl;
    }
    return 0;
}

static vector<string> commands = {"build-fmin", "search-fmin"};
static void print_help(char** argv) {
    cerr << "Available commands: " << endl;
    for (auto& S : commands) cerr << "   " << argv[0] << " " << S << endl;
    cerr << "Running a command without arguments prints the usage instructions for the command." << endl;
}

int main(int argc, char** argv) {
    omp_set_num_threads(fin_host_threads());
    if (argc == 1) { print_help(argv); return 1; }
    string command = argv[1];
    if (command == "--help" || command == "-h") { print_help(argv); return 1; }
    for (int i = 1; i < argc; i++) argv[i - 1] = argv[i];
    argc--;
    try {
        if (command == "build-fmin") return build_fmin(argc, argv);
        else if (command == "search-fmin") return search_fmin(argc, argv);
        else if (command == "parse-reads") return parse_reads(argc, argv);
        else throw runtime_error("Invalid command: " + command);
    } catch (const runtime_error& e) {
        cerr << "Runtime error: " << e.what() << '\n';
        return 1;
    } catch (const exception& e) {
        cerr << "Error: " << e.what() << '\n';
        return 1;
    }
}
