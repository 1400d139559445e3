#!/bin/bash
# On the GPU box: per-kernel times (rocprofv3 kernel trace) of one bench workload under two settings of FINITO_OPTS.
# usage: tools/trace_ab.sh <tag> "<optsA>" "<optsB>" [bench args...]   -> gpurun_out/trace_<tag>/{A,B}.txt
set -o pipefail
TAG=$1; A=$2; B=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in A B; do
  if [ $V = A ]; then export FINITO_OPTS="$A"; else export FINITO_OPTS="$B"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$V -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e "$@" > $OUT/$V.json 2> $OUT/$V.log || exit 1
  F=$(find $OUT/$V -name "*kernel_stats.csv" | head -1)
  { echo "FINITO_OPTS=$FINITO_OPTS"; python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r["Name"].startswith("fin_") or float(r["Percentage"]) > 1:
        print("%-60s calls %6s  total %10.3f ms  avg %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  } > $OUT/$V.txt
  find $OUT/$V -name "*kernel_trace.csv" -size +2M -delete; find $OUT/$V -name "*.db" -delete
done
cat $OUT/A.txt $OUT/B.txt
