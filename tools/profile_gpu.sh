#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of the default bench workload.
# usage: tools/profile_gpu.sh <tag> [bench args...]        -> gpurun_out/prof_<tag>/{summary.txt,kernel_stats.csv,traffic_entry.json}
# (bench.py runs without its end-to-end leg: that leg starts the CLI as a child process, which must not happen under the profiler)
set -o pipefail
TAG=${1:-run}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--no-text: every launch of the profiled run belongs to a default step -- the step_with_text passes ran the pre-pass in text-only mode, which writes no pairs
#  for the reads it finishes, and pulled its per-launch average down: VERDICT r4 weak #3)
ARGS="--steps 2 --warmup 1 --no-cpu --no-e2e --no-text $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.log || exit 1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$N.json 2> $OUT/pmc_$N.log || echo "pmc pass $C failed"
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
