#!/usr/bin/env python3
"""Deep fuzz of round 3's deferred second strand on the GPU (many seeds): tests/test_search_gpu.py's deferral, non-disjoint-family and
small-index tests with other random streams, kernel 4.  usage: tools/fuzz_defer.py [n_seeds]"""
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
        finally:
            np.random.default_rng = orig
        print("seed", seed, "ok", "%.0f s" % (time.time() - t0), flush=True)

def mixed_indexes(n_cases=200, seed=1):
    """indexes with duplicated stretches and reverse-complement copies at every k, reads of both strands with errors, chimeras and junk:
    the device (defaults: second strands deferred) against the faithful oracle"""
    from oracle.oracle import OracleIndex
    from tests.util import cut_unitigs, mosaic_read, random_genome, rc, sample_reads
    from tests.test_oracle_lazy import non_disjoint_sets
    rng = np.random.default_rng(seed)
    stats = {"cases": 0, "rc_pairs": 0, "unsafe": 0}
    for case in range(n_cases):
        k = int(rng.choice([7, 12, 16, 21, 31, 32, 40, 63]))
        if case % 3 == 0:
            g, unitigs = non_disjoint_sets(rng, case, k)
        else:
            g = random_genome(rng, int(rng.integers(1500, 12000)))
            for _ in range(int(rng.integers(0, 5))):
                a = int(rng.integers(0, len(g) - 300)); n = int(rng.integers(k + 2, 300)); at = int(rng.integers(0, len(g)))
                g = g[:at] + g[a:a + n] + g[at:]
            unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(2 * k, 12 * k)), flip=bool(case % 2))
            for _ in range(int(rng.integers(0, 6))):
                a = int(rng.integers(0, len(g) - 300)); unitigs.append(rc(g[a:a + int(rng.integers(k, 300))]))
        unitigs = [u for u in unitigs if len(u) >= k]
        o = OracleIndex.build(unitigs, k)
        p = fa.FinimizerIndex.build(unitigs, k).to_device(0)
        L = min(len(g), int(rng.integers(k, 400)))
        reads = [mosaic_read(rng, g, k, 400) for _ in range(60)] + sample_reads(rng, g, 150, L, err=float(rng.choice([0.0, 0.01, 0.03])), random_frac=0.05) + [g[:min(len(g), 3000)], rc(g[-min(len(g), 1200):])]
        reads += [rc(r) for r in reads[:50]]
        exp, _, _ = o.search_batch(reads)
        got, _ = p.search_reads(reads, fa.FIN_MERGED)
        assert np.array_equal(got.astype(np.int64), exp), "mixed case %d (seed %d, k=%d)" % (case, seed, k)
        stats["cases"] += 1; stats["rc_pairs"] += p.rc_pairs() > 0; stats["unsafe"] += p.unsafe_places() > 0
        p.close()
    print("mixed indexes ok:", stats, flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "mixed":
        mixed_indexes(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1); sys.exit(0)
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
