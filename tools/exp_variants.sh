#!/bin/bash
# usage: tools/exp_variants.sh "<HIPFLAGS_EXTRA variant 1>" "<variant 2>" ...   (runs the chr1 bench without the CPU leg per variant)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in "$@"; do
  touch finito_amd/csrc/fin_kernel_v2.hip
  make -s -C finito_amd/csrc HIPFLAGS_EXTRA="$V" ../libfinito_amd.so 2>&1 | grep -E "error|v2.hip:102.*(VGPRs:|Spill:|Occupancy|Scratch)"
  python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT [$V]', 'k-mers/s %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])"
done
