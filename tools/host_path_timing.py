#!/usr/bin/env python3
"""Times the host-buffer path fin_search_batch (H2D + pack + search + D2H) on the chr1 index; prints k-mers/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finito_amd as fa
from finito_amd import synth
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
g = synth.genome(250_000_000); u = synth.unitigs(g, 31)
idx = fa.FinimizerIndex.build(u.as_tuple(), 31).to_device(0)
r = synth.reads(g, n_reads)
idx.search_reads(r.subset(0, 1000).as_tuple())
for rep in range(3):
    t = time.perf_counter(); pairs, npos = idx.search_reads(r.as_tuple()); dt = time.perf_counter() - t
    print("fin_search_batch: %d reads, %d k-mers in %.3f s = %.3g k-mers/s (PCIe-inclusive)" % (n_reads, pairs.shape[0], dt, pairs.shape[0] / dt), flush=True)

out = np.zeros_like(pairs)   # a pageable output buffer whose pages exist already (a fresh one pays first-touch page faults on top)
for rep in range(3):
    t = time.perf_counter(); p1, _ = idx.search_reads(r.as_tuple(), out=out); dt = time.perf_counter() - t
    print("fin_search_batch, pageable buffers reused: %.3f s = %.3g k-mers/s" % (dt, p1.shape[0] / dt), flush=True)
assert np.array_equal(p1, pairs)
del out, p1

pin_out = fa.PinnedArray((pairs.shape[0], 2), np.int32)
pin_in = fa.PinnedArray((r.bases.size,), np.uint8)
pin_in.array[:] = r.bases
for rep in range(3):
    t = time.perf_counter(); p2, npos2 = idx.search_reads((pin_in.array, r.offsets), out=pin_out.array); dt = time.perf_counter() - t
    print("fin_search_batch, pinned buffers: %.3f s = %.3g k-mers/s" % (dt, p2.shape[0] / dt), flush=True)
assert np.array_equal(p2, pairs)
