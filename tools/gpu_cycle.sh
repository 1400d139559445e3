#!/bin/bash
# One GPU iteration for kernel $1 (default 2): parity tests, epoch statistics (diagnostic build), then the chr1 bench on the product build.
K=${1:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout -k 10 300 python -m pytest tests/test_search_gpu.py -m gpu -x -q --timeout 200 -k "v$K and not full_size" 2>&1 | tail -3 || exit 1
touch finito_amd/csrc/fin_kernel_v$K.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_STATS" ../libfinito_amd.so 2>&1 | grep -E "error"
python bench.py --workload chr1 --steps 1 --warmup 0 --no-cpu --reads 2000000 --kernel $K 2>&1 | grep -E "fin_stats|fin_time" | tail -2
touch finito_amd/csrc/fin_kernel_v$K.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E "error"
python bench.py --workload chr1 --steps 3 --warmup 1 --no-cpu --kernel $K 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH k-mers/s %.4g' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])"
