#!/bin/bash
# A/B of a compile-time switch on ONE box over several workloads: tools/ab_macro2.sh <file.hip> <MACRO> "<workloads>" values...
F=$1; M=$2; WS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT/finito_amd/csrc
for V in $@; do
  touch $F; make -s HIPFLAGS_EXTRA="-D$M=$V" all > /dev/null 2>&1 || exit 1
  for W in $WS; do (cd $ROOT && timeout -k 10 300 python bench.py --workload $W --steps 10 --warmup 3 --no-e2e --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$M=$V', '$W', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms_parts'].items()})"); done
done
touch $F; make -s all > /dev/null 2>&1
