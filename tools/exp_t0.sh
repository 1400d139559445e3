#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_STATS" ../libfinito_amd.so 2>&1 | grep -E "error"
for T in auto 9 10 11 12 13; do
  if [ $T = auto ]; then unset FINITO_LCS_T0; else export FINITO_LCS_T0=$T; fi
  echo "T0=$T $(python bench.py --workload chr1 --steps 1 --warmup 0 --no-cpu --reads 2000000 2>&1 | grep -E 'fin_stats' | tail -1 | sed 's/.*epoch=\([0-9]*\).*win_exti=\([0-9]*\).*/epochs=\1 bdrop=\2/')"
done
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E "error"
for T in 10 11 12; do
  export FINITO_LCS_T0=$T
  python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T0=$T BENCH k-mers/s %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])"
done
