#!/bin/bash
# On the GPU box: everything profiles/rNN holds for a round, from ONE box -- the five workloads' bench lines, the rocprofv3 trace + PMC passes
# of chr1 (traffic entry), the stress mix, the CLI runs, the builders' timing.   usage: tools/round_artifacts.sh <tag> [benches|rest]   -> gpurun_out/<tag>/
# (a gpurun call lasts twenty minutes at most: "benches" = the workloads' bench lines, "rest" = profile, stress mix, CLI, builders; no second argument: both)
set -o pipefail
TAG=${1:-round}; PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
if [ "$PART" != "rest" ]; then
for W in chr1 ecoli k63 chr1_repeats chr1_dups k63_repeats k127; do
  python bench.py --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { echo "bench $W failed"; tail -5 $OUT/bench_$W.err; exit 1; }
  python - $OUT/bench_$W.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(d["config"]["workload"].split(" = ")[0], "%.4g k-mers/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % r["frac"], "B/k-mer %.2f" % r["algorithmic_bytes_per_kmer"],
      {k: round(v, 3) for k, v in r["kernel_ms_parts"].items()}, "text %.2f ms" % r["stages"]["text"]["ms"] if "text" in r.get("stages", {}) else "")
PY
done
# round 5: the strong-scaling mode's N = 1 point (configs[3]: ONE set of 100 M reads, here in four device batches on one GPU) and indexes beyond 2^30 bases (1.2 Gbp; 3 Gbp = a human genome; 4.1 Gbp = 96 % of the 2^32-node limit)
python bench.py --workload chr1x8 --gpus 1 --steps 2 --warmup 1 --no-cpu --no-e2e --no-legs > $OUT/bench_chr1x8_n1.json 2> $OUT/bench_chr1x8_n1.err || { echo "bench chr1x8 (N=1) failed"; tail -5 $OUT/bench_chr1x8_n1.err; }
python bench.py --workload chr1 --genome 1200000000 --steps 3 --warmup 1 --no-cpu --no-e2e --no-legs > $OUT/bench_1200Mbp.json 2> $OUT/bench_1200Mbp.err || { echo "bench 1.2 Gbp failed"; tail -5 $OUT/bench_1200Mbp.err; }
python bench.py --workload chr1 --genome 3000000000 --steps 3 --warmup 1 --no-cpu --no-e2e --no-legs --no-text > $OUT/bench_3000Mbp.json 2> $OUT/bench_3000Mbp.err || { echo "bench 3 Gbp failed"; tail -5 $OUT/bench_3000Mbp.err; }
python bench.py --workload chr1 --genome 4100000000 --steps 3 --warmup 1 --no-cpu --no-e2e --no-legs --no-text > $OUT/bench_4100Mbp.json 2> $OUT/bench_4100Mbp.err || { echo "bench 4.1 Gbp failed"; tail -5 $OUT/bench_4100Mbp.err; }
# beyond 2^32 nodes: 6.2 Gbp of unitigs as a partitioned index of two parts (k = 63: an iid genome of that size repeats 31-mers by chance and is refused)
python bench.py --workload k63 --genome 6000000000 --parts-max-bases 3200000000 --steps 3 --warmup 1 > $OUT/bench_pindex_k63_6Gbp.json 2> $OUT/bench_pindex_k63_6Gbp.err || { echo "bench partitioned 6 Gbp failed"; tail -5 $OUT/bench_pindex_k63_6Gbp.err; }
grep -h "partitioned index built\|ground truth" $OUT/bench_pindex_k63_6Gbp.err
for F in bench_chr1x8_n1 bench_1200Mbp bench_3000Mbp bench_4100Mbp; do python - $OUT/$F.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
    print(sys.argv[1].split("/")[-1], "%.4g k-mers/s" % d["value"], "%.3f ms per step" % d["ms_per_step"], d["scaling"], "reads", c.get("reads_total"), "batches", c.get("batches_per_gpu"),
          "index bases", c["index_bases"], "tables B/base", c["derived_tables_bytes_per_indexed_base"], {k: round(v, 3) for k, v in d["roofline"]["kernel_ms_parts"].items()})
except Exception as e:
    print("no line:", e)
PY
done
fi
[ "$PART" == "benches" ] && exit 0
tools/profile_gpu.sh $TAG > $OUT/profile.log 2>&1 || { echo "profile failed"; tail -5 $OUT/profile.log; exit 1; }
cp gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_chr1_summary.txt; cp gpurun_out/prof_$TAG/kernel_stats.csv $OUT/kernel_stats.csv; cp gpurun_out/prof_$TAG/traffic_entry.json $OUT/traffic_entry.json
grep "traffic entry" $OUT/rocprofv3_chr1_summary.txt | cut -c1-400
python tools/stress_mix.py > $OUT/stress_mix.txt 2> $OUT/stress_mix.err || { echo "stress mix failed"; tail -5 $OUT/stress_mix.err; exit 1; }
tail -25 $OUT/stress_mix.txt
tools/cli_e2e.sh > $OUT/cli_end_to_end.txt 2>&1 || echo "cli_e2e failed"
tools/cli_stages.sh > $OUT/cli_stages.txt 2>&1 || echo "cli_stages failed"
python tools/build_timing.py > $OUT/build_timing_chr1.txt 2>&1 || echo "build timing failed"
tail -n 4 $OUT/cli_end_to_end.txt; tail -n 4 $OUT/build_timing_chr1.txt
