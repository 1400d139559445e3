#!/bin/bash
# One GPU iteration: v2 parity tests, epoch statistics (diagnostic build), then the chr1 bench on the product build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout -k 10 300 python -m pytest tests/test_search_gpu.py -m gpu -x -q --timeout 100 -k "v2" 2>&1 | tail -3 || exit 1
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_STATS" ../libfinito_amd.so 2>&1 | grep -E "error"
python bench.py --workload chr1 --steps 1 --warmup 0 --no-cpu --reads 2000000 2>&1 | grep fin_stats | tail -1
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E "error"
python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH k-mers/s %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])"
