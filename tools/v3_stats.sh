#!/bin/bash
# Lane-epochs by mode of kernel 3 (diagnostic -DFIN_V3_STATS build), then restores the product build.  usage: tools/v3_stats.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
touch finito_amd/csrc/fin_kernel_v3.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_V3_STATS" ../libfinito_amd.so 2>&1 | grep -E " error"
python bench.py --workload chr1 --steps 1 --warmup 0 --no-cpu --no-e2e --reads 2000000 "$@" 2>&1 | grep -E "fin_v3_stats" | tail -2
touch finito_amd/csrc/fin_kernel_v3.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu --no-e2e "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH k-mers/s %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])"
exit 0
