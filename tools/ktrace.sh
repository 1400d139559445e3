#!/bin/bash
# rocprofv3 kernel trace of a short bench run: per-kernel average durations.  usage: tools/ktrace.sh <tag> [bench args]
TAG=${1:-kt}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e "$@" > $OUT/trace.json 2> $OUT/trace.log || exit 1
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kernel_stats.csv
python3 - $F <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"].split("(")[0]
    if n.startswith("fin_") and not n.startswith("fin_build") and "gb_" not in n:
        print("%-34s calls %4s  avg %10.1f us  max %10.1f us" % (n[:34], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
