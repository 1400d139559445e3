#!/bin/bash
# A/B on one box: the k-mer-table look-ups of the walk kernel with and without the second slot fetched along.  usage: tools/ab_kf_pair.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
run() {
  for W in chr1 ecoli chr1_repeats chr1_dups k63 k63_repeats; do
    python bench.py --workload $W --steps 8 --warmup 2 --no-e2e --no-cpu --no-legs 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1 $W', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms_parts'].items()})"
  done
}
for V in "1 1" "0 0" "1 1" "0 0"; do
  set -- $V
  touch finito_amd/csrc/fin_kernel_w.hip
  make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_W_KT_PAIR=$1 -DFIN_W_KT2_PAIR=$2" ../libfinito_amd.so 2>&1 | grep -E " error"
  run "pair=$1/$2"
done
touch finito_amd/csrc/fin_kernel_w.hip; make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
