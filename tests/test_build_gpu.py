"""SURVEY 8 f-1, "then accelerate": the index built ON THE DEVICE (finito_amd/csrc/fin_build_gpu.hip: k-mer extraction, rocPRIM radix sort,
dummy nodes, LCS from neighbouring keys, edge marks, permute_unitigs, the finimizer pass as kernels) must be the host builder's index bit
for bit -- the two container files are compared byte by byte -- and the host builder equals the oracle's literal construction
(tests/test_builder_parity.py).  Matches lcs_basic_parallel_algorithm.hpp:52-120, PackedStrings.hh:105-135, FinimizerIndex.hh:321-389."""
import numpy as np
import pytest

import finito_amd as fa
from finito_amd import synth
from tests.util import cut_unitigs, random_genome, rc

pytestmark = pytest.mark.gpu


def same_container(unitigs, k, tmp_path, tag):
    h = fa.FinimizerIndex.build(unitigs, k)
    d = fa.FinimizerIndex.build_on_device(unitigs, k, 0)
    assert (d.n_nodes, d.n_kmers, d.n_unitigs, d.n_finimizers, d.total_len) == (h.n_nodes, h.n_kmers, h.n_unitigs, h.n_finimizers, h.total_len), tag
    for what in (fa.X_C, fa.X_PLANE_A, fa.X_PLANE_A + 1, fa.X_PLANE_A + 2, fa.X_PLANE_A + 3, fa.X_LCS, fa.X_FMIN, fa.X_USTART, fa.X_GOFF, fa.X_ENDS, fa.X_CONCAT):
        assert np.array_equal(d.export(what), h.export(what)), "%s: component %d" % (tag, what)
    h.serialize(str(tmp_path / "h")); d.serialize(str(tmp_path / "d"))
    a = open(tmp_path / "h.finamd", "rb").read(); b = open(tmp_path / "d.finamd", "rb").read()
    assert a == b, "%s: container files differ (%d vs %d bytes)" % (tag, len(a), len(b))
    return d


def test_reference_cases_on_device(kat, tmp_path):
    for c in kat:
        same_container(c["unitigs"], c["k"], tmp_path, c["name"])


@pytest.mark.parametrize("k", [2, 3, 5, 8, 13, 16, 21, 31, 32, 33, 47, 63, 64, 65, 100, 127, 128, 129, 200, 250, 255])   # (k > 32: two-word keys; > 64: four words; > 128: eight, and exact LCS bytes beside the 7-bit ones)
def test_random_sets_every_k(k, tmp_path):
    rng = np.random.default_rng(400 + k)
    for case in range(6):
        g = random_genome(rng, int(rng.integers(60, 20000)))
        if case % 3 == 0:
            unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(k + 1, 4 * k + 200)))
        elif case % 3 == 1:   # not disjoint: repeated pieces, shared first k-mers, low complexity
            base = random_genome(rng, int(rng.integers(3, 40)))
            gg = base * 30
            unitigs = [gg[a:a + L] for a, L in ((int(rng.integers(0, 60)), int(rng.integers(k, k + 80))) for _ in range(int(rng.integers(1, 40))))]
            unitigs += ["A" * (k + 7), "AC" * k, "T" * (k + 1)]
        else:                 # unrelated strings: many dummy nodes
            unitigs = [random_genome(rng, int(rng.integers(k, k + 30))) for _ in range(int(rng.integers(1, 80)))]
        unitigs = [u for u in unitigs if len(u) >= k]
        same_container(unitigs, k, tmp_path, "k=%d case %d" % (k, case))


def test_repeat_rich_and_config2_scale(tmp_path):
    """a repeat-rich disjoint set (short pieces, many dummy nodes) and BASELINE config 2's index (5 Mbp, k = 31), with the searches of
    the device-built index checked against the ground truth"""
    g = synth.repeat_genome(600_000, seed=9)
    u = synth.spss(g, 31, max_len=1500)
    same_container(u.as_tuple(), 31, tmp_path, "repeats")
    g = synth.genome(5_000_000)
    u = synth.unitigs(g, 31)
    d = same_container(u.as_tuple(), 31, tmp_path, "config2")
    assert sum(d.build_phase_ms.values()) > 0
    d.to_device(0)
    r = synth.reads(g, 50_000)
    got, _ = d.search_reads(r.as_tuple(), fa.FIN_MERGED)
    bad, checked, first = synth.check_ground_truth(d, u, r, got)
    assert bad == 0 and checked > 0, (bad, checked, first)
    # k = 63 (two-word keys) at the same scale, and its build time
    u = synth.unitigs(g, 63)
    d = same_container(u.as_tuple(), 63, tmp_path, "config2 at k = 63")
    assert sum(d.build_phase_ms.values()) < 1000.0, d.build_phase_ms


def test_wide_keys_at_config2_scale(tmp_path):
    """k = 127 (four-word keys) and k = 200 (eight words, exact LCS bytes beside the 7-bit ones): config 2's 5 Mbp, container = the host builder's, searches of the
    device-built index against the ground truth -- VERDICT r4 next #10 (lcs_basic_parallel_algorithm.hpp:52-120 takes any k <= 255)"""
    g = synth.genome(5_000_000)
    for k, rl in ((127, 250), (200, 400)):
        u = synth.unitigs(g, k)
        d = same_container(u.as_tuple(), k, tmp_path, "config2 at k = %d" % k)
        assert sum(d.build_phase_ms.values()) < 3000.0, d.build_phase_ms
        d.to_device(0)
        r = synth.reads(g, 20_000, read_len=rl)
        got, _ = d.search_reads(r.as_tuple(), fa.FIN_MERGED)
        bad, checked, first = synth.check_ground_truth(d, u, r, got)
        assert bad == 0 and checked > 0, (k, bad, checked, first)
        d.close()


def test_device_builder_errors():
    with pytest.raises(fa.FinitoError) as e:
        fa.FinimizerIndex.build_on_device(["ACGTACGTAA" * 40], 256, 0)
    assert e.value.code == -5
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex.build_on_device(["ACGNACGT"], 4, 0)
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex.build_on_device(["ACG"], 4, 0)
