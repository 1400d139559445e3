// tools/kt3_collisions.cpp -- how many k-mers of a unitig set share bucket AND tag in the compact k-mer table (fin_format.h: fin_kt3_hash)?  CPU only.
// usage: python3 -c "from finito_amd import synth; import numpy as np; g = synth.repeat_genome(20_000_000); u = synth.spss(g, 63);
//                    np.asarray(u.bases).tofile('ub.bin'); np.asarray(u.offsets).astype(np.uint64).tofile('uo.bin')"
//        g++ -O2 -o kt3_collisions tools/kt3_collisions.cpp && ./kt3_collisions        (reads ub.bin / uo.bin from the working directory; k = 63)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>
#include "/root/repo/finito_amd/csrc/fin_format.h"
static std::vector<uint8_t> rd(const char* p) { FILE* f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint8_t> v(n); fread(v.data(), 1, n, f); fclose(f); return v; }
int main() {
    auto ub = rd("ub.bin"); auto uo8 = rd("uo.bin"); const uint64_t* uo = (const uint64_t*)uo8.data(); const size_t nu = uo8.size() / 8 - 1;
    const int k = 63; uint64_t places = 0;
    for (size_t u = 0; u < nu; u++) if (uo[u + 1] - uo[u] >= (uint64_t)k) places += uo[u + 1] - uo[u] - k + 1;
    const uint32_t nb = (uint32_t)((places * 100 / 60 + 3) / 4 + 16);
    std::unordered_map<uint64_t, std::pair<uint64_t,uint64_t>> seen; seen.reserve(places * 2); uint64_t coll = 0, same = 0, fewbits = 0;
    for (size_t u = 0; u < nu; u++) {
        uint64_t k0 = 0, k1 = 0; 
        for (uint64_t i = uo[u]; i < uo[u + 1]; i++) {
            const char ch = ub[i]; const uint64_t c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
            k0 = (k0 >> 2) | (k1 << 62); k1 = (k1 >> 2) | (c << (2 * ((k - 33) & 31)));
            if (i - uo[u] + 1 < (uint64_t)k) continue;
            const uint64_t h = fin_kt3_hash(k0, k1);
            const uint64_t id = ((uint64_t)fin_kt3_bucket(h, nb) << 32) | ((uint32_t)h & FIN_KT3_TAGMASK);
            auto it = seen.find(id);
            if (it == seen.end()) seen[id] = {k0, k1};
            else if (it->second.first == k0 && it->second.second == k1) same++;
            else { coll++; if (__builtin_popcountll(it->second.first ^ k0) <= 4) fewbits++; }
        }
    }
    printf("places %llu distinct ids %zu same %llu collisions %llu (few-bit %llu)\n", (unsigned long long)places, seen.size(), (unsigned long long)same, (unsigned long long)coll, (unsigned long long)fewbits);
}
