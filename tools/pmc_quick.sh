#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass of bench.py and per-kernel sums.  usage: tools/pmc_quick.sh <tag> "<counters>" [bench args]
set -o pipefail
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e "$@" > $OUT/bench.json 2> $OUT/bench.log || { tail -5 $OUT/bench.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
sums = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "?").split("(")[0][:40]
        sums[kn][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[kn][row["Counter_Name"]] += 1
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for kn in sorted(sums):
        if not kn.startswith("fin_"): continue
        for c in sorted(sums[kn]):
            line = "pmc %-28s %-26s launches=%-4d sum=%.6g per_launch=%.6g" % (kn, c, cnt[kn][c], sums[kn][c], sums[kn][c] / cnt[kn][c])
            print(line); fo.write(line + "\n")
PY
rm -rf $OUT/raw
