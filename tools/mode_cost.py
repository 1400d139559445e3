#!/usr/bin/env python3
"""Kernel time per base for pure read populations on the chr1 index (forward strand only):
M0 = error-free windows of the unitig text (every k-mer present, long walks), M1 = the same with 1 % substitutions,
R = random reads (nothing present).  Tells what a matching and a non-matching strand cost."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finito_amd as fa
from finito_amd import synth
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
L = 150
g = synth.genome(250_000_000); u = synth.unitigs(g, 31)
idx = fa.FinimizerIndex.build(u.as_tuple(), 31).to_device(0)
rng = np.random.default_rng(5)
# windows inside unitigs: pick unitigs long enough, uniform start
ulen = (u.offsets[1:] - u.offsets[:-1]).astype(np.int64)
ok = np.nonzero(ulen >= L)[0]
pick = ok[rng.integers(0, len(ok), n_reads)]
st = u.offsets[pick].astype(np.int64) + (rng.random(n_reads) * (ulen[pick] - L + 1)).astype(np.int64)
ix = (st[:, None] + np.arange(L)[None, :]).reshape(-1)
m0 = u.bases[ix].copy()
offs = (np.arange(n_reads + 1, dtype=np.uint64) * L)
m1 = m0.copy()
e = np.nonzero(rng.random(m1.size) < 0.01)[0]
m1[e] = np.frombuffer(b"ACGT", dtype=np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), m1[e]) + rng.integers(1, 4, e.size)) % 4]
r = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_reads * L)]
if "KERNEL" in os.environ:
    fa.lib().fin_set_option(b"kernel", int(os.environ["KERNEL"]))
only = os.environ.get("ONLY")
for name, bases in (("M0 (error-free)", m0), ("M1 (1% errors)", m1), ("R (random)", r)):
    if only and not name.startswith(only):
        continue
    for strands, sn in ((fa.FIN_FWD, "fwd"),) if only else ((fa.FIN_FWD, "fwd"), (fa.FIN_MERGED, "merged")):
        b = idx.batch((bases, offs))
        b.run(strands); b.download(want_pairs=False)
        for _ in range(0 if only else 2): b.run(strands)
        _, npos = b.download(want_pairs=False)
        ms, n = b.kernel_time_ms()
        nb = n_reads * L * (2 if strands == fa.FIN_MERGED else 1)
        print("%-16s %-6s kernel %.2f ms  = %.1f ps per base-strand, found %.1f%%" % (name, sn, ms, ms * 1e9 / nb, 100.0 * npos / b.n_kmers), flush=True)
        b.close()
