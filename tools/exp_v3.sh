#!/bin/bash
# usage: tools/exp_v3.sh [--kernel N] "<HIPFLAGS_EXTRA variant 1>" ...   (kernel 3/4 build variants: chr1 bench without the CPU leg per variant)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
KARG=""
if [ "$1" == "--kernel" ]; then KARG="--kernel $2"; shift 2; fi
for V in "$@"; do
  touch finito_amd/csrc/fin_kernel_v3.hip finito_amd/csrc/fin_kernel_w.hip
  make -s -C finito_amd/csrc HIPFLAGS_EXTRA="$V" ../libfinito_amd.so 2>&1 | grep -E " error"
  python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu --no-e2e $KARG 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT [$V]', 'k-mers/s %.4g' % d['value'], 'step_ms %.2f' % d['roofline']['kernel_ms'], {k: round(v, 2) for k, v in d['roofline'].get('kernel_ms_parts').items()})"
done
touch finito_amd/csrc/fin_kernel_v3.hip finito_amd/csrc/fin_kernel_w.hip; make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
