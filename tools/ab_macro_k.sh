#!/bin/bash
# A/B on one box: builds of the kernels with different macro settings.  usage: tools/ab_macro_w.sh "<flags A>" "<flags B>" ...   (workloads: $AB_WORKLOADS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
WL=${AB_WORKLOADS:-"chr1 ecoli chr1_repeats chr1_dups"}
for V in "$@"; do
  touch finito_amd/csrc/fin_kernel_w.hip finito_amd/csrc/fin_prepass.hip
  make -s -C finito_amd/csrc HIPFLAGS_EXTRA="$V" ../libfinito_amd.so 2>&1 | grep -E " error"
  for W in $WL; do
    python bench.py --workload $W --steps 8 --warmup 2 --no-e2e --no-cpu --no-legs --no-text 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('[$V] $W', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms_parts'].items()})"
  done
done
touch finito_amd/csrc/fin_kernel_w.hip finito_amd/csrc/fin_prepass.hip; make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
