"""Runs bench.py on a library that was built with -DFIN_W_DEBUG (tools/w_stats.sh builds one on the spot; or build it before the GPU call:
make -C finito_amd/csrc HIPFLAGS_EXTRA=-DFIN_W_DEBUG after touching fin_kernel_w.hip) and dumps the walk kernel's counters.  usage: python tools/w_stats_run.py [bench args]"""
import runpy
import sys

sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e", "--no-legs", "--no-text"] + sys.argv[1:]
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    import finito_amd as fa
    fa.lib().fin_debug_time()
