#!/usr/bin/env python3
"""CPU fuzz: the lazy restatement (oracle/finito_lazy.c, the algorithm the default kernels run) against the faithful restatement on the
deferred strand's hard family (tests/util.py: defer_family_case -- identical / near-duplicate / reverse-complement unitigs; reads, their reverse
complements, the unitigs themselves, reads that run past a unitig's end).  usage: tools/fuzz_lazy.py [n_sets] [first_seed] [n_procs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(args):
    seed0, n = args
    from oracle.oracle import OracleIndex
    from tests.util import defer_family_case
    bad = []
    for seed in range(seed0, seed0 + n):
        rng = np.random.default_rng(seed)
        k = int(rng.choice([7, 9, 12, 16, 21, 31, 32, 40, 47, 63]))
        g, unitigs, reads = defer_family_case(rng, seed, k)
        o = OracleIndex.build(unitigs, k)
        exp, _, _ = o.search_batch(reads)
        T = int(rng.choice([0, 2, 4, 6])); J = int(rng.choice([0, 1, 2, 3]))
        for defer in (True, False):
            for kt, lean in ((True, False), (True, True), (False, False)):   # (lean tables: the device's default for k <= 31)
                got = o.search_batch_lazy(reads, ptab_t=T, jump_t=J, seeds=True, kmer_table=kt, defer=defer, lean=lean)
                if not np.array_equal(got, exp):
                    bad.append((seed, k, T, J, defer, kt, lean))
    return bad


if __name__ == "__main__":
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    import multiprocessing as mp
    chunk = 25
    jobs = [(first + i, min(chunk, n_sets - i)) for i in range(0, n_sets, chunk)]
    t0 = time.time(); bad = []; done = 0
    with mp.Pool(procs) as pool:
        for b in pool.imap_unordered(run, jobs):
            bad += b; done += 1
            if done % 40 == 0:
                print("%d / %d sets, %d bad, %.0f s" % (done * chunk, n_sets, len(bad), time.time() - t0), flush=True)
    print("sets %d (seeds %d..%d): %d differing runs %s" % (n_sets, first, first + n_sets - 1, len(bad), bad[:20]))
    sys.exit(1 if bad else 0)
