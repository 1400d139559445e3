#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of bench.py, per-kernel totals per step and the launches of the pipeline's rounds.
# usage: tools/trace_quick.sh <tag> [bench args]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e "$@" > $OUT/bench.json 2> $OUT/bench.log || { tail -5 $OUT/bench.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
f = glob.glob(os.path.join(out, "raw", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
from collections import defaultdict
tot = defaultdict(float); per = defaultdict(list)
for r in rows:
    kn = r["Kernel_Name"].split("(")[0][:36]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[kn] += d; per[kn].append(d)
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for kn in sorted(tot, key=lambda k: -tot[k]):
        line = "%-38s launches=%-4d total_ms=%9.3f  longest: %s" % (kn, len(per[kn]), tot[kn], " ".join("%.2f" % x for x in sorted(per[kn], reverse=True)[:9]))
        print(line); fo.write(line + "\n")
PY
rm -rf $OUT/raw
