#!/bin/bash
# Per-block execution statistics of kernel 2 on a purely matching population (forward strand only, 1 % errors): what the streaming
# blocks cost when every lane is in matching mode, as in kernel 3's search kernel.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_BLOCKS" ../libfinito_amd.so 2>&1 | grep -E " error"
KERNEL=2 ONLY=M1 python tools/mode_cost.py 2000000 2>&1 | grep -E "fin_blocks" | tail -42
touch finito_amd/csrc/fin_kernel_v2.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
