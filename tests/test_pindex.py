"""Partitioned indexes (round 5; include/finito_amd.h: fin_pindex_*): a unitig set held as parts, each an ordinary index below 2^32 nodes, must answer
pair for pair what ONE index of all the unitigs answers -- the reference's FinimizerIndex over the whole set (FinimizerIndex.hh:26-259, unitig numbers by
permute_unitigs, PackedStrings.hh:105-135) -- which the oracle computes here; and a set that is not a disjoint spectrum-preserving string set
(README.md:79-80) must be refused.  The sizes here force parts of a few thousand bases; bench.py --parts runs a set beyond 2^32 nodes."""
import numpy as np
import pytest

import finito_amd as fa


def _reads(rng, g, k, unitigs):
    from tests.test_search_gpu import _fast_path_reads
    from tests.util import sample_reads
    return _fast_path_reads(rng, g, k, unitigs) + sample_reads(rng, g, 300, 150 if k < 100 else 400, err=0.02) + ["", "ACGT", g[100:100 + k - 1], g[50:50 + k]]


@pytest.mark.gpu
@pytest.mark.parametrize("k,max_part", [(31, 9000), (21, 25000), (63, 12000), (47, 100000), (100, 15000)])
def test_parts_answer_as_one_index(k, max_part):
    from oracle.oracle import OracleIndex
    from tests.util import cut_unitigs, random_genome
    rng = np.random.default_rng(1000 + k)
    g = random_genome(rng, 60000)
    unitigs = cut_unitigs(rng, g, k, max_len=900)
    reads = _reads(rng, g, k, unitigs)
    o = OracleIndex.build(unitigs, k)
    exp, _, _ = o.search_batch(reads)
    one = fa.FinimizerIndex.build(unitigs, k).to_device(0)
    want, want_pos = one.search_reads(reads)
    assert np.array_equal(want.astype(np.int64), exp)
    p = fa.PartitionedIndex(unitigs, k, device=0, max_part_bases=max_part, verify=True)
    try:
        total = sum(len(u) for u in unitigs)
        assert (p.n_parts == 1) if max_part >= total else (p.n_parts >= 3 and p.n_parts == len(p.part_nodes()))
        assert p.shared_kmers == 0 and p.n_unitigs == len(unitigs) and p.total_len == total and p.n_kmers == one.n_kmers
        assert p.n_nodes >= one.n_nodes   # (every part brings its own dummy nodes)
        # the set's unitig numbers are permute_unitigs' over ALL unitigs: the parts' tables together are a permutation, each ascending
        ids = np.concatenate([p.unitig_ids(i) for i in range(p.n_parts)])
        assert np.array_equal(np.sort(ids), np.arange(len(unitigs), dtype=np.uint32))
        assert all((np.diff(p.unitig_ids(i).astype(np.int64)) > 0).all() for i in range(p.n_parts))
        got, npos = p.search_reads(reads)
        assert np.array_equal(got.astype(np.int64), exp), "k=%d: parts and the whole index disagree" % k
        assert npos == want_pos == int((exp[:, 0] != -1).sum())
        # the device-resident form, run twice (the first part's buffer is the set's result: a second run must not see renumbered pairs)
        b = p.batch(reads)
        b.run(); b.run()
        got2, npos2 = b.download()
        assert np.array_equal(got2, got) and npos2 == npos
        ms, n = b.step_time_ms()
        assert n == 2 and ms > 0
        b.close()
    finally:
        p.close(); one.close()


@pytest.mark.gpu
def test_a_set_that_is_not_disjoint_is_refused():
    from tests.util import cut_unitigs, random_genome, rc
    rng = np.random.default_rng(4242)
    k = 31
    g = random_genome(rng, 40000)
    unitigs = cut_unitigs(rng, g, k, max_len=700)
    fa.PartitionedIndex(unitigs, k, max_part_bases=8000).close()   # the set itself is fine
    far = len(unitigs) - 1
    long3 = next(u for u in unitigs[:8] if len(u) >= k + 40)
    for extra, where in ((unitigs[0][:200], far), (rc(unitigs[1][:150]), far), (long3[5:5 + k], far), (unitigs[2], 3)):
        # a stretch of an early unitig again -- as it is, reverse-complemented, a single k-mer -- in a part far behind; a whole unitig twice inside one part
        bad = unitigs[:where] + [extra] + unitigs[where:]
        with pytest.raises(fa.FinitoError) as e:
            fa.PartitionedIndex(bad, k, max_part_bases=8000, verify=True)
        assert e.value.code == fa.FIN_EINVAL and "disjoint" in str(e.value)
        q = fa.PartitionedIndex(bad, k, max_part_bases=8000, verify=False)   # unchecked: builds (and is the caller's risk)
        assert q.shared_kmers == -1
        q.close()
    with pytest.raises(fa.FinitoError):
        fa.PartitionedIndex(unitigs + ["ACGT"], k, max_part_bases=8000)   # a unitig shorter than k
    with pytest.raises(fa.FinitoError):
        fa.PartitionedIndex(unitigs, k, max_part_bases=300)               # a unitig longer than a part may be
