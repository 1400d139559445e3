#!/bin/bash
# End-to-end `finito search-fmin` on FASTQ files (plain and gzipped) against a 50 Mbp index; prints wall times.
# usage: tools/cli_e2e.sh [n_reads]   (default 4000000: several 256 MB chunks, so the host pipeline's stages overlap)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
N=${1:-4000000}
T=/tmp/fin_e2e; mkdir -p $T
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$ROOT")
from finito_amd import synth
g = synth.genome(50_000_000); u = synth.unitigs(g, 31); r = synth.reads(g, $N)
with open("$T/u.fna", "wb") as f:
    b = u.bases.tobytes()
    for i in range(len(u)):
        f.write(b">%d\n" % i); f.write(b[int(u.offsets[i]):int(u.offsets[i+1])]); f.write(b"\n")
L = r.read_len; b = r.bases.tobytes(); q = b"I" * L
with open("$T/r.fq", "wb") as f:
    for i in range(len(r)):
        f.write(b"@r%d\n" % i); f.write(b[i*L:(i+1)*L]); f.write(b"\n+\n"); f.write(q); f.write(b"\n")
PY
now() { date +%s.%N; }
el() { python3 -c "print('%.2f' % ($(now) - $1))"; }
head -c 1200000000 $T/r.fq | head -n 4000000 > $T/r1m.fq   # the first 1 M reads: output checksum comparable across rounds
gzip -1 -k -f $T/r.fq
S=$(now); finito_amd/finito build-fmin -o $T/idx -u $T/u.fna -k 31 2>&1 | tail -2; echo "build-fmin wall $(el $S) s"
S=$(now); finito_amd/finito search-fmin -i $T/idx -q $T/r.fq -o $T/out.txt 2>&1 | grep -E "us/query|Total found"; echo "search-fmin (plain fastq, $N reads) wall $(el $S) s"
S=$(now); finito_amd/finito search-fmin -i $T/idx -q $T/r.fq.gz -o $T/out2.txt 2>&1 | grep -E "us/query"; echo "search-fmin (gzip fastq) wall $(el $S) s"
S=$(now); finito_amd/finito search-fmin -i $T/idx -q $T/r.fq 2>/dev/null | md5sum; echo "search-fmin (plain fastq, stdout to a pipe) wall $(el $S) s"
cmp $T/out.txt $T/out2.txt && ls -la $T/out.txt && md5sum $T/out.txt
finito_amd/finito search-fmin -i $T/idx -q $T/r1m.fq -o $T/out1m.txt 2>&1 | grep -E "us/query end"; md5sum $T/out1m.txt
