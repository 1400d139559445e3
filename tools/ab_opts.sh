#!/bin/bash
# A/B on one box of process-wide options (FINITO_OPTS) over a list of workloads: step parts per setting.
# usage: tools/ab_opts.sh <tag> "<workloads>" "<optsA>" "<optsB>" ... [-- bench args]      ("-" = no option)
TAG=$1; WLS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for W in $WLS; do
  for S in "${SETS[@]}"; do
    O=$S; [ "$S" == "-" ] && O=""
    N=$(echo "$S" | tr '=,' '__')
    FINITO_OPTS=$O timeout -k 10 400 python3 $ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu --no-e2e --no-text "$@" > $OUT/${W}_$N.json 2> $OUT/${W}_$N.err || { echo "FAILED $W $S"; tail -3 $OUT/${W}_$N.err; exit 1; }
    python3 - "$OUT/${W}_$N.json" "$W" "$S" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d["roofline"]["kernel_ms_parts"]
print("%-14s %-28s step %.3f ms = ingest %.3f + pre-pass %.3f + search %.3f   %.4g k-mers/s   fast %s  k3list %s" % (sys.argv[2], sys.argv[3], p["step"], p["ingest_prefill"], p["probe_prepass"], p["search"], d["value"],
      d["roofline"].get("reads_finished_by_the_fast_path"), d["roofline"].get("pipeline_queue_slots", {}).get("kernel3_list")))
PY
  done
done
