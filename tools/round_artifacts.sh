#!/bin/bash
# On the GPU box: everything profiles/rNN holds for a round, from ONE box -- the five workloads' bench lines, the rocprofv3 trace + PMC passes
# of chr1 (traffic entry), the stress mix, the CLI runs, the builders' timing.   usage: tools/round_artifacts.sh <tag>   -> gpurun_out/<tag>/
set -o pipefail
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for W in chr1 ecoli k63 chr1_repeats chr1_dups k63_repeats; do
  python bench.py --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { echo "bench $W failed"; tail -5 $OUT/bench_$W.err; exit 1; }
  python - $OUT/bench_$W.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(d["config"]["workload"].split(" = ")[0], "%.4g k-mers/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % r["frac"], "B/k-mer %.2f" % r["algorithmic_bytes_per_kmer"],
      {k: round(v, 3) for k, v in r["kernel_ms_parts"].items()}, "text %.2f ms" % r["stages"]["text"]["ms"] if "text" in r.get("stages", {}) else "")
PY
done
tools/profile_gpu.sh $TAG > $OUT/profile.log 2>&1 || { echo "profile failed"; tail -5 $OUT/profile.log; exit 1; }
cp gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_chr1_summary.txt; cp gpurun_out/prof_$TAG/kernel_stats.csv $OUT/kernel_stats.csv; cp gpurun_out/prof_$TAG/traffic_entry.json $OUT/traffic_entry.json
grep "traffic entry" $OUT/rocprofv3_chr1_summary.txt | cut -c1-400
python tools/stress_mix.py > $OUT/stress_mix.txt 2> $OUT/stress_mix.err || { echo "stress mix failed"; tail -5 $OUT/stress_mix.err; exit 1; }
tail -25 $OUT/stress_mix.txt
tools/cli_e2e.sh > $OUT/cli_end_to_end.txt 2>&1 || echo "cli_e2e failed"
tools/cli_stages.sh > $OUT/cli_stages.txt 2>&1 || echo "cli_stages failed"
python tools/build_timing.py > $OUT/build_timing_chr1.txt 2>&1 || echo "build timing failed"
tail -n 4 $OUT/cli_end_to_end.txt; tail -n 4 $OUT/build_timing_chr1.txt
