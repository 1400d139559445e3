#!/bin/bash
# Where the wall time of `finito search-fmin` goes on the 250 Mbp index with few reads (the case bench.py's end_to_end.cli_* times):
# container load, upload + tables, page-locked buffers, pipeline, output.   usage: tools/cli_startup.sh [n_reads]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
N=${1:-2000000}
T=/dev/shm/fin_startup; mkdir -p $T
python - <<PY
import numpy as np, sys, time
sys.path.insert(0, "$ROOT")
import finito_amd as fa
from finito_amd import synth
g = synth.genome(250_000_000); u = synth.unitigs(g, 31); r = synth.reads(g, $N)
t = time.time(); idx = fa.FinimizerIndex.build(u.as_tuple(), 31); print("build %.1f s" % (time.time() - t))
idx.serialize("$T/idx")
L = r.read_len; b = r.bases[: $N * L].reshape($N, L)
rec = np.empty(($N, 2 * L + 7), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8); rec[:, 3:3 + L] = b; rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 6 + L:6 + 2 * L] = ord("I"); rec[:, 6 + 2 * L] = ord("\n"); rec.tofile("$T/r.fq")
PY
now() { date +%s.%N; }
for SINK in $T/out.txt /dev/null; do
  S=$(now); FINITO_TIMING=1 finito_amd/finito search-fmin -i $T/idx -q $T/r.fq -o $SINK 2>&1 | grep -E "timing|us/query|startup"; python3 -c "print('== sink $SINK: wall %.2f s' % ($(now) - $S))"
done
rm -rf $T
