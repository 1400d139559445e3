#!/usr/bin/env python3
"""host builder vs device builder on the benchmark's index (tools/build_timing.py [genome bases] [repeats]): wall times, the device
builder's stages, container equality"""
import sys, time, hashlib, os, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finito_amd as fa
from finito_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
rep = len(sys.argv) > 2 and sys.argv[2] == "repeats"
g = synth.repeat_genome(n) if rep else synth.genome(n)
u = synth.spss(g, 31) if rep else synth.unitigs(g, 31)
print("unitigs: %d, %d bases" % (len(u), int(u.offsets[-1])), flush=True)
fa.FinimizerIndex.build_on_device((u.bases[:4000000], u.offsets[:1] if False else np.array([0, 4000000], dtype=np.uint64)), 31, 0).close()   # warm-up: HIP context, rocPRIM kernels
t = time.time(); d = fa.FinimizerIndex.build_on_device(u.as_tuple(), 31, 0); td = time.time() - t
print("device build: %.2f s wall; stages (ms): %s; sum %.0f ms" % (td, {k: round(v, 1) for k, v in d.build_phase_ms.items()}, sum(d.build_phase_ms.values())), flush=True)
t = time.time(); h = fa.FinimizerIndex.build(u.as_tuple(), 31); th = time.time() - t
print("host build: %.2f s wall (%d threads)" % (th, fa.host_threads()), flush=True)
tmp = tempfile.mkdtemp(dir="/dev/shm")
h.serialize(tmp + "/h"); d.serialize(tmp + "/d")
ha = hashlib.md5(open(tmp + "/h.finamd", "rb").read()).hexdigest(); hb = hashlib.md5(open(tmp + "/d.finamd", "rb").read()).hexdigest()
print("containers identical:", ha == hb, ha, "nodes", d.n_nodes, "speed-up %.1fx" % (th / td))
import shutil; shutil.rmtree(tmp)
