#!/usr/bin/env python3
"""Diagnostic: run one batch several times and report the reads whose pairs differ between runs (a path that depends on which lanes share a wave).
usage: tools/debug_nondet.py [genome_bases] [n_reads] [k] [kind]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finito_amd as fa
from finito_amd import synth

n_g = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n_r = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
kind = sys.argv[4] if len(sys.argv) > 4 else "iid"
g = synth.repeat_genome(n_g, seed=9) if kind == "repeats" else synth.genome(n_g)
u = synth.spss(g, k, max_len=4000) if kind == "repeats" else synth.unitigs(g, k)
idx = fa.FinimizerIndex.build_on_device(u.as_tuple(), k, 0)
idx.to_device(0)
r = synth.reads(g, n_r, read_len=150)
b = idx.batch(r.as_tuple())
runs = []
for i in range(4):
    b.run(fa.FIN_MERGED)
    got, npos = b.download()
    runs.append(got.copy())
    bad, checked, first = synth.check_ground_truth(idx, u, r, got)
    print("run", i, "found", npos, "ground truth: bad", bad, "of", checked, "first bad read", first, flush=True)
nk = 150 - k + 1
for i in range(1, 4):
    d = np.nonzero((runs[i] != runs[0]).any(axis=1))[0]
    print("run", i, "differs from run 0 in", len(d), "pairs; reads", sorted(set((d // nk).tolist()))[:10])
    for j in d[:6]:
        print("   pair", int(j), "read", int(j // nk), "slot", int(j % nk), "run0", runs[0][j].tolist(), "run%d" % i, runs[i][j].tolist())
