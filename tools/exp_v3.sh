#!/bin/bash
# usage: tools/exp_v3.sh "<HIPFLAGS_EXTRA variant 1>" ...   (kernel 3 variants: chr1 bench without the CPU leg per variant)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in "$@"; do
  touch finito_amd/csrc/fin_kernel_v3.hip
  make -s -C finito_amd/csrc HIPFLAGS_EXTRA="$V" ../libfinito_amd.so 2>&1 | grep -E " error|v3.hip:9[0-9]:.*(VGPRs:|Scratch)"
  python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT [$V]', 'k-mers/s %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_parts'))"
done
touch finito_amd/csrc/fin_kernel_v3.hip; make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
