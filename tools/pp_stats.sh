#!/bin/bash
# Why reads leave the pre-pass's fast path (diagnostic -DFIN_PP_STATS build), then restores the product build.
# usage: tools/pp_stats.sh [bench args, e.g. --workload ecoli]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
touch finito_amd/csrc/fin_prepass.hip
make -s -C finito_amd/csrc HIPFLAGS_EXTRA="-DFIN_PP_STATS" ../libfinito_amd.so 2>&1 | grep -E " error"
python - "$@" <<'PY'
import sys
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e"] + sys.argv[1:]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    import finito_amd as fa
    fa.lib().fin_debug_time()
PY
touch finito_amd/csrc/fin_prepass.hip
make -s -C finito_amd/csrc ../libfinito_amd.so 2>&1 | grep -E " error"
exit 0
