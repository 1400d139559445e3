#!/bin/bash
# A/B of a compile-time switch on ONE box: tools/ab_macro.sh <file.hip> <MACRO> <workload> [values...]   (each value twice, interleaved)
set -o pipefail
F=$1; M=$2; W=$3; shift 3
VALS=${@:-0 1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$M; mkdir -p $OUT
cd $ROOT/finito_amd/csrc
for REP in 1 2; do for V in $VALS; do
  touch $F
  make -s HIPFLAGS_EXTRA="-D$M=$V" all > $OUT/make_$V.log 2>&1 || exit 1
  (cd $ROOT && timeout -k 10 300 python bench.py --workload $W --steps 10 --warmup 2 --no-e2e --no-cpu 2> /dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$M=$V', '$W', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms_parts'].items()})") | tee -a $OUT/result.txt
done; done
touch $F; make -s all > /dev/null 2>&1
