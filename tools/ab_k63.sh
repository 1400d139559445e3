#!/bin/bash
# A/B on ONE box: the k = 63 step with and without the start-at-E rule of the bridging strings for k > 32
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_k63; mkdir -p $OUT
cd $ROOT/finito_amd/csrc
for V in 0 1 0 1; do
  touch fin_kernel_w.hip
  make -s HIPFLAGS_EXTRA="-DFIN_W_LONGK_ESTART=$V" all > $OUT/make_$V.log 2>&1 || exit 1
  (cd $ROOT && timeout -k 10 300 python bench.py --workload k63 --steps 10 --warmup 2 --no-e2e --no-cpu 2> /dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('ESTART=$V', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms_parts'].items()})") | tee -a $OUT/result.txt
done
touch fin_kernel_w.hip; make -s all > /dev/null 2>&1
