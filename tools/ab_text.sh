#!/bin/bash
# A/B of a compile-time switch of fin_text.hip on one box: tools/ab_text.sh <MACRO> [values...]  -> text-stage ms of the chr1 batch
set -o pipefail
M=$1; shift
VALS=${@:-0 1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$M; mkdir -p $OUT
cd $ROOT/finito_amd/csrc
for REP in 1 2; do for V in $VALS; do
  touch fin_text.hip
  make -s HIPFLAGS_EXTRA="-D$M=$V" all > $OUT/make_$V.log 2>&1 || exit 1
  (cd $ROOT && timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-e2e --no-cpu 2> /dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$M=$V', 'text ms', round(d['step_with_text']['ms_text'],3), 'step ms', round(d['ms_per_step'],3))") | tee -a $OUT/result.txt
done; done
touch fin_text.hip; make -s all > /dev/null 2>&1
