#!/usr/bin/env python3
"""Deep fuzz of round 3's deferred second strand on the GPU (many seeds): tests/test_search_gpu.py's deferral, non-disjoint-family and
small-index tests with other random streams, kernel 4.  usage: tools/fuzz_defer.py [n_seeds] | mixed N [seed] | family N [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tests.test_search_gpu as T
import finito_amd as fa

def main(n_seeds=8):
    fa.lib().fin_set_option(b"kernel", 4)
    orig = np.random.default_rng
    for seed in range(n_seeds):
        t0 = time.time()
        np.random.default_rng = lambda s=0, _o=orig, _seed=seed: _o(50021 + 104729 * _seed + (int(s) if isinstance(s, (int, np.integer)) else 0))
        try:
            T.test_deferred_second_strand(4)
            T.test_non_disjoint_families(4, 4711)
            T.test_fuzz_many_small_indexes()
            print("  family:", T.defer_family_cases(150, 9000 + seed), flush=True)   # (the widened generator, tests/util.py)
        finally:
            np.random.default_rng = orig
        print("seed", seed, "ok", "%.0f s" % (time.time() - t0), flush=True)

def mixed_indexes(n_cases=200, seed=1):
    print("mixed indexes ok:", T.mixed_index_cases(n_cases, seed), flush=True)

def family(n_cases=1000, seed=1):
    """the deferred strand's hard family (identical / near-duplicate / reverse-complement unitigs; reads, rc(reads), the unitigs themselves,
    reads past a unitig's end) on the device against the faithful oracle, defer_strand 1 and 0"""
    fa.lib().fin_set_option(b"kernel", 4)
    done = 0
    while done < n_cases:
        n = min(250, n_cases - done)
        print("family ok:", T.defer_family_cases(n, seed * 1000003 + done, ks=(7, 9, 12, 16, 21, 31, 32, 40, 63)), flush=True)
        done += n

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "family":
        family(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1); sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "mixed":
        mixed_indexes(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1); sys.exit(0)
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
