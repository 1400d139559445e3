#!/bin/bash
# Why items leave the walk kernel for the streaming search (diagnostic -DFIN_W_DEBUG build), then restores the product build.
# usage: tools/w_stats.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
touch finito_amd/csrc/fin_kernel_w.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_W_DEBUG" ../libfinito_amd.so 2>&1 | grep -E " error"
python - "$@" <<'PY'
import sys
sys.argv = ["bench.py", "--workload", "chr1", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e"] + sys.argv[1:]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    import finito_amd as fa
    fa.lib().fin_debug_time()
PY
touch finito_amd/csrc/fin_kernel_w.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
