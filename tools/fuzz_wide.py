"""Deep fuzz of the walk kernel's wide-key look-ups (64 <= k <= 255; fin_kernel_w.hip W_KF0B / W_REANCH): random index sets -- half of them with duplicated
stretches and reverse-complement copies (unsafe places, unverified answers, flagged windows) -- and read mixes that cross unitig ends, carry errors and N's, on
kernel 4 with the fast path off / on (every third set under option lean_tables 3), against the faithful oracle.   usage: python tools/fuzz_wide.py [n_cases] [seed]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import finito_amd as fa
from oracle.oracle import OracleIndex
from tests.util import cut_unitigs, mosaic_read, random_genome, rc, sample_reads

import os
for kv in filter(None, os.environ.get("FINITO_OPTS", "").split(",")):   # any process-wide option, "name=value,..." (as bench.py)
    name, val = kv.split("=")
    assert fa.lib().fin_set_option(name.encode(), int(val)) == 0, kv
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
bad = 0
for case in range(n_cases):
    k = int(rng.integers(64, 256))
    g = random_genome(rng, int(rng.integers(4 * k + 2000, 30000)))
    dup = case % 2 == 1
    if dup:
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, len(g) - 3 * k)); n = int(rng.integers(k + 1, 3 * k)); at = int(rng.integers(0, len(g)))
            g = g[:at] + g[a:a + n] + g[at:]
    unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(k + 2, 6 * k + 500)), flip=bool(rng.integers(0, 2)))
    if dup:
        unitigs += [rc(g[a:a + int(rng.integers(k, 2 * k + 50))]) for a in rng.integers(0, len(g) - 3 * k, int(rng.integers(1, 4)))]
    L = int(rng.integers(k, 2 * k + 300))
    reads = sample_reads(rng, g, 300, L, err=float(rng.choice([0.0, 0.003, 0.01, 0.03])), random_frac=0.05)
    reads += [mosaic_read(rng, g, k, 3 * k + 200) for _ in range(80)]
    reads += [u for u in unitigs[:15]] + [rc(u) for u in unitigs[:15]] + [g[:k], rc(g[-k:]), "", "N" * k, g[5:5 + k - 1]]
    for i in range(20):   # N's and lower case
        r = list(reads[i]);
        if len(r) > 3: r[int(rng.integers(0, len(r)))] = "N"
        reads.append("".join(r).lower() if i % 2 else "".join(r))
    o = OracleIndex.build(unitigs, k)
    exp, _, _ = o.search_batch(reads, n_threads=8)
    p = fa.FinimizerIndex.build_on_device(unitigs, k, 0)
    if case % 3 == 2:
        p.set_option("lean_tables", 3)   # every third set: lean tables above 63 (no prefix table, no anchor table)
    p.to_device(0)
    for fp in (1, 2):
        p.set_option("fast_path", fp)
        got, _ = p.search_reads(reads)
        if not np.array_equal(got.astype(np.int64), exp):
            bad += 1
            d = np.nonzero((got.astype(np.int64) != exp).any(axis=1))[0]
            print("MISMATCH case %d k=%d dup=%d fast_path=%d: %d pairs differ, first at %d: got %s want %s" % (case, k, dup, fp, len(d), d[0], got[d[0]], exp[d[0]]), flush=True)
    p.close()
    if case % 10 == 9:
        print("case %d done (k=%d)" % (case, k), flush=True)
print("fuzz_wide: %d cases, %d mismatches, unitig sets with duplicates: %d" % (n_cases, bad, n_cases // 2))
sys.exit(1 if bad else 0)
