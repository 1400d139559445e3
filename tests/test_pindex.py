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


@pytest.mark.gpu
def test_partitioned_index_on_disk_and_through_the_command(tmp_path):
    """serialize / load (the manifest, every part's container, the tables of set-wide unitig numbers) and the command: `finito build-fmin --parts-max-bases`
    writes a partitioned index, `finito search-fmin` finds it by its manifest and prints the text the one-index run prints, byte for byte
    (search_fmin.hh:62-65 through the host formatter)"""
    import os
    import subprocess
    from tests.util import cut_unitigs, random_genome
    rng = np.random.default_rng(909)
    k = 31
    g = random_genome(rng, 50000)
    unitigs = cut_unitigs(rng, g, k, max_len=800)
    reads = _reads(rng, g, k, unitigs)
    reads = [r for r in reads if len(r) > 0]   # (FASTA records)
    p = fa.PartitionedIndex(unitigs, k, max_part_bases=12000)
    want, want_pos = p.search_reads(reads)
    p.serialize(str(tmp_path / "px"))
    assert fa.PartitionedIndex.exists(str(tmp_path / "px")) and not fa.PartitionedIndex.exists(str(tmp_path / "nothing"))
    n_parts = p.n_parts
    p.close()
    q = fa.PartitionedIndex.load(str(tmp_path / "px"))
    assert q.n_parts == n_parts and q.k == k and q.n_unitigs == len(unitigs) and q.shared_kmers == 0
    got, got_pos = q.search_reads(reads)
    assert np.array_equal(got, want) and got_pos == want_pos
    got2, _ = q.search_reads(reads[:100])   # (the set's cached device batches, reloaded with another read set)
    assert np.array_equal(got2, want[: len(got2)])
    q.close()
    with open(tmp_path / "px.p1.gid", "r+b") as f:   # a damaged table is refused
        f.truncate(8)
    with pytest.raises(fa.FinitoError):
        fa.PartitionedIndex.load(str(tmp_path / "px"))
    # the command
    cli = os.path.join(os.path.dirname(fa.__file__), "finito")
    ufa, rfa = tmp_path / "u.fna", tmp_path / "r.fna"
    ufa.write_text("".join(">u%d\n%s\n" % (i, u) for i, u in enumerate(unitigs)))
    rfa.write_text("".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads)))
    for tag, extra in (("one", []), ("parts", ["--parts-max-bases", "12000"])):
        b = subprocess.run([cli, "build-fmin", "-u", str(ufa), "-o", str(tmp_path / tag), "-k", str(k)] + extra, capture_output=True, text=True, timeout=300)
        assert b.returncode == 0, b.stderr[-800:]
        sr = subprocess.run([cli, "search-fmin", "-i", str(tmp_path / tag), "-q", str(rfa), "-o", str(tmp_path / (tag + ".txt")), "--gpus", "1"], capture_output=True, text=True, timeout=300)
        assert sr.returncode == 0, sr.stderr[-800:]
    assert os.path.exists(tmp_path / "parts.finparts") and not os.path.exists(tmp_path / "one.finparts")
    a, b_ = open(tmp_path / "one.txt", "rb").read(), open(tmp_path / "parts.txt", "rb").read()
    assert a == b_ and len(a) > 10000
    exp = fa.format_pairs  # (the text of the pairs the API delivered, read by read)
    nk = [max(0, len(r) - k + 1) for r in reads]
    off = np.concatenate([[0], np.cumsum(nk)])
    assert b_ == "".join(exp(want[off[i]:off[i + 1]]) for i in range(len(reads))).encode()
