import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
import finito_amd as fa
from finito_amd import synth
from oracle.oracle import OracleIndex
g = synth.genome(2_000_000)
u = synth.unitigs(g, 31)
r = synth.reads(g, 50_000)
idx = fa.FinimizerIndex.build(u.as_tuple(), 31).to_device(0)
print("tables", idx.replica_table_bytes(), "ktab", idx.kmer_table_bytes(), "cbf", idx.string_filter_bytes(), "ptab T", idx.prefix_table_depth())
o = OracleIndex.build(u.as_tuple(), 31)
exp, _, _ = o.search_batch(r.as_tuple(), n_threads=8)
for fast in (1, 0):
    fa.lib().fin_set_option(b"fast_path", fast)
    b = idx.batch(r.as_tuple()); b.run(fa.FIN_MERGED); got, npos = b.download()
    pc = b.pipeline_counts(48)
    ok = np.array_equal(got.astype(np.int64), exp)
    print("fast_path", fast, "equal", ok, "fast reads", pc[41], "of", len(r), "sisters", pc[40])
    if not ok:
        bad = np.nonzero((got.astype(np.int64) != exp).any(axis=1))[0]
        print(len(bad), bad[:10], got[bad[:5]].tolist(), exp[bad[:5]].tolist())
    for _ in range(5): b.run(fa.FIN_MERGED)
    print("step ms", b.step_time_ms(2))
    b.close()
