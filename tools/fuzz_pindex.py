"""Deep fuzz of partitioned indexes (fin_pindex_*): random disjoint unitig sets cut into 2 .. 12 parts, any k in [9, 120], read mixes with errors, N's and reads that
cross unitig ends, against the faithful oracle's ONE index of all the unitigs; every fourth case a set that is not disjoint, which the build must refuse -- or, where the check lets it through (a reverse-complemented stretch inside one part), answer exactly.
usage: python tools/fuzz_pindex.py [n_cases] [seed]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import finito_amd as fa
from oracle.oracle import OracleIndex
from tests.util import cut_unitigs, mosaic_read, random_genome, rc, sample_reads

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 424242)
bad = refused = kept = 0
for case in range(n_cases):
    k = int(rng.choice([9, 13, 21, 31, 32, 33, 47, 63, 64, 65, 100, 120]))
    g = random_genome(rng, int(rng.integers(20000, 70000)))
    unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(k + 2, 5 * k + 600)), flip=bool(rng.integers(0, 2)))
    total = sum(len(u) for u in unitigs)
    max_part = max(max(len(u) for u in unitigs), total // int(rng.integers(2, 13)))
    nondis = case % 4 == 3
    if nondis:   # not disjoint: a stretch again, as it is or reverse-complemented, somewhere else
        a = int(rng.integers(0, len(g) - 2 * k)); piece = g[a:a + int(rng.integers(k, 2 * k))]
        unitigs.insert(int(rng.integers(0, len(unitigs) + 1)), piece if rng.integers(0, 2) else rc(piece))
        max_part = max(max_part, len(piece))
    if k < 21:
        continue   # (a random genome of this size repeats 9- and 13-mers by itself: such a set is refused, rightly)
    L = int(rng.integers(k, 2 * k + 200))
    reads = sample_reads(rng, g, 250, L, err=float(rng.choice([0.0, 0.01, 0.03])), random_frac=0.05) + [mosaic_read(rng, g, k, 2 * k + 300) for _ in range(60)]
    reads += unitigs[:10] + [rc(u) for u in unitigs[:10]] + ["", "N" * k, g[:k], g[7:7 + k - 1]]
    if nondis:
        reads += [piece, rc(piece), g[max(0, a - 60):a + len(piece) + 60], rc(g[max(0, a - 60):a + len(piece) + 60])]
    exp, _, _ = OracleIndex.build(unitigs, k).search_batch(reads, n_threads=8)
    try:
        p = fa.PartitionedIndex(unitigs, k, max_part_bases=max_part)
    except fa.FinitoError:
        if not nondis:
            bad += 1; print("REFUSED a disjoint set: case %d k=%d" % (case, k), flush=True)
        refused += 1
        continue
    if nondis:
        kept += 1   # (a set the check lets through -- a reverse-complemented stretch inside ONE part is that part's own business -- must still answer as one index)
    got, npos = p.search_reads(reads)
    if not np.array_equal(got.astype(np.int64), exp) or npos != int((exp[:, 0] != -1).sum()):
        bad += 1
        d = np.nonzero((got.astype(np.int64) != exp).any(axis=1))[0]
        print("MISMATCH case %d k=%d parts=%d: %d pairs differ" % (case, k, p.n_parts, len(d)), flush=True)
    p.close()
print("fuzz_pindex: %d cases, %d mismatches, %d non-disjoint sets refused, %d let through (and exact)" % (n_cases, bad, refused, kept))
sys.exit(1 if bad else 0)
