#!/bin/bash
# Wave-cycle shares per segment of the search kernels' epoch (diagnostic -DFIN_V3_TIME build), then restores the product build.
# usage: tools/time_stats.sh [bench args, e.g. --kernel 4]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
touch finito_amd/csrc/fin_kernel_v3.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_V3_TIME" ../libfinito_amd.so 2>&1 | grep -E " error"
python - "$@" <<'PY'
import ctypes, json, subprocess, sys, os
sys.argv = ["bench.py", "--workload", "chr1", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e", "--reads", "4000000"] + sys.argv[1:]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    import finito_amd as fa
    fa.lib().fin_debug_time()
PY
touch finito_amd/csrc/fin_kernel_v3.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
