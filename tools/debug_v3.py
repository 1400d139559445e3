#!/usr/bin/env python3
"""First differences between kernel 3 and the oracle on a small random case: tools/debug_v3.py k [seed] [genome] [n_reads] [read_len]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finito_amd as fa
from oracle.oracle import OracleIndex
from tests.util import cut_unitigs, random_genome, sample_reads
k = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100 + k
gl = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
nr = int(sys.argv[4]) if len(sys.argv) > 4 else 400
rl = int(sys.argv[5]) if len(sys.argv) > 5 else 150
rng = np.random.default_rng(seed)
g = random_genome(rng, gl)
unitigs = cut_unitigs(rng, g, k)
reads = sample_reads(rng, g, nr, rl)
fa.lib().fin_set_option(b"kernel", int(os.environ.get("KERNEL", "3")))
if "PTAB" in os.environ: fa.lib().fin_set_option(b"ptab_t", int(os.environ["PTAB"]))
p = fa.FinimizerIndex.build(unitigs, k).to_device(0); o = OracleIndex.build(unitigs, k)
for mode, name in ((fa.FIN_FWD, "fwd"), (fa.FIN_MERGED, "merged")):
    got, _ = p.search_reads(reads, mode)
    nbad = 0; off = 0
    for ri, r in enumerate(reads):
        nk = max(0, len(r) - k + 1)
        exp = np.array(o.search(r)[0] if mode == fa.FIN_FWD else o.search_merged(r), dtype=np.int64).reshape(-1, 2)
        gg = got[off:off + nk].astype(np.int64); off += nk
        if not np.array_equal(gg, exp):
            bad = np.nonzero((gg != exp).any(axis=1))[0]
            if nbad < 4:
                print(name, "read", ri, "len", len(r), "bad positions", bad[:12].tolist(), "n_bad", len(bad))
                for b in bad[:4]: print("   i=%d got=%s exp=%s" % (b, gg[b].tolist(), exp[b].tolist()))
            nbad += 1
    print(name, "reads with differences:", nbad, "of", len(reads))
