"""Multi-GPU layout of the path: reads sharded by record, index replicated, no collective on the data path.

The reference is single-process; SURVEY.md section 8(e) fixes the sharding: contiguous blocks of read records balanced by
cumulative bases (so per-rank outputs concatenate in input order), one full index replica per GPU.  torch.distributed is
used for the barrier and for moving the index container's *path* only.
"""
import os

import numpy as np


def shard_bounds(offsets, world):
    """Record ranges [lo, hi) per rank: contiguous, covering, balanced by cumulative bases (not by read count)."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    total = int(offsets[-1] - offsets[0])
    cuts = [0]
    for r in range(1, world):
        target = int(offsets[0]) + total * r // world
        cuts.append(int(np.searchsorted(offsets, target, side="left")))
    cuts.append(n)
    for i in range(1, len(cuts)):
        cuts[i] = min(max(cuts[i], cuts[i - 1]), n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_reads(bases, offsets, rank, world):
    """This rank's slice of a flat read set, re-based to offset 0."""
    lo, hi = shard_bounds(offsets, world)[rank]
    offsets = np.asarray(offsets, dtype=np.uint64)
    b0, b1 = int(offsets[lo]), int(offsets[hi])
    return np.ascontiguousarray(bases[b0:b1]), (offsets[lo:hi + 1] - offsets[lo]).astype(np.uint64), (lo, hi)


def replicate_index(build_fn, prefix, rank, dist=None):
    """Rank 0 builds the index and writes the container; every other rank loads it (FinimizerIndex::serialize/load,
    FinimizerIndex.hh:187-241).  Returns this rank's FinimizerIndex (not yet on a device)."""
    import finito_amd as fa
    idx = None
    if rank == 0:
        idx = build_fn()
        if dist is not None:
            idx.serialize(prefix)
    if dist is not None:
        dist.barrier()
        if rank != 0:
            idx = fa.FinimizerIndex().load(prefix)
        dist.barrier()
        if rank == 0:
            try:
                os.unlink(prefix + ".finamd")
            except OSError:
                pass
    return idx


def pair_offsets(offsets, k):
    """First output pair of every read (reads back to back, max(0, len-k+1) pairs each)."""
    lens = (np.asarray(offsets[1:], dtype=np.int64) - np.asarray(offsets[:-1], dtype=np.int64))
    nk = np.maximum(lens - k + 1, 0)
    out = np.zeros(len(nk) + 1, dtype=np.int64)
    np.cumsum(nk, out=out[1:])
    return out
