#!/bin/bash
# Per-block execution statistics of the search kernel (diagnostic -DFIN_BLOCKS build), then restores the product build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_BLOCKS" ../libfinito_amd.so 2>&1 | grep -E " error"
python bench.py --workload chr1 --steps 1 --warmup 0 --no-cpu --reads 2000000 2>&1 | grep -E "fin_blocks"
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
