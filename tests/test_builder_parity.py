"""The product's index builder (finito_amd/csrc/fin_build.cpp) must produce exactly the structures the literal
restatement of the reference's construction produces (oracle: SBWT node order per tests.cpp:110-123, LCS by
label propagation, permute_unitigs, add_sequence).  Runs on CPU."""
import numpy as np
import pytest

import finito_amd as fa
from oracle.oracle import OracleIndex
from tests.util import cut_unitigs, random_genome, unpack_bits


def assert_same_index(unitigs, k):
    o = OracleIndex.build(unitigs, k)
    p = fa.FinimizerIndex.build(unitigs, k, n_threads=4)
    n = o.n_nodes
    assert p.k == k and p.n_nodes == n
    assert p.n_kmers == o.n_kmers
    assert p.n_unitigs == o.n_unitigs and p.total_len == o.total_len
    assert p.export(fa.X_C).tolist() == o.C_array().tolist()
    for c in range(4):
        assert np.array_equal(unpack_bits(p.export(fa.X_PLANE_A + c), n), o.plane(c)), "plane %d" % c
    assert np.array_equal(p.export(fa.X_LCS), o.lcs())
    assert np.array_equal(unpack_bits(p.export(fa.X_USTART), n), o.ustart())
    assert np.array_equal(unpack_bits(p.export(fa.X_FMIN), n), o.fmin())
    assert np.array_equal(p.export(fa.X_ENDS), o.ends())
    assert np.array_equal(p.export(fa.X_CONCAT), o.concat())
    # the reference sizes global_offsets by the number of distinct finimizers (FinimizerIndex.hh:301)
    assert p.n_finimizers == o.n_fmin == int(o.fmin().sum())
    assert np.array_equal(p.export(fa.X_GOFF), o.global_offsets())
    return p, o


@pytest.mark.parametrize("name", ["test_shortest_unique_construction", "test_finimizer_branch", "test_reverse_complement_branch",
                                  "test_leftmost", "test_finimizer_selection", "test_incoming_rc_branch", "test_walk", "example_fna_k4"])
def test_reference_cases(kat, name):
    c = next(x for x in kat if x["name"] == name)
    p, _ = assert_same_index(c["unitigs"], c["k"])
    for key, what in (("lcs", fa.X_LCS), ("ends", fa.X_ENDS), ("global_offsets", fa.X_GOFF), ("concat", fa.X_CONCAT)):
        if key in c:
            assert p.export(what).tolist() == c[key]
    for key, what in (("fmin", fa.X_FMIN), ("ustart", fa.X_USTART)):
        if key in c:
            assert unpack_bits(p.export(what), p.n_nodes).tolist() == c[key]


@pytest.mark.parametrize("k", [2, 3, 4, 5, 9, 16, 31, 32, 33, 47, 63, 64, 65, 80, 96, 97, 100, 127, 128, 129, 160, 192, 193, 250, 255])
def test_random_dspss(k):
    rng = np.random.default_rng(100 + k)
    g = random_genome(rng, 4000 if k > 8 else 300)
    assert_same_index(cut_unitigs(rng, g, k, max_len=max(2 * k, 120)), k)


@pytest.mark.parametrize("k", [4, 6, 12])
def test_repetitive_non_disjoint(k):
    """Low-complexity, non-disjoint string sets: duplicate k-mers across unitigs, shared first k-mers,
    many dummy nodes -- the reference does not require a disjoint SPSS (test_walk has a duplicate k-mer)."""
    rng = np.random.default_rng(7 + k)
    base = random_genome(rng, 60)
    unitigs = []
    for _ in range(25):
        a = int(rng.integers(0, 40)); L = int(rng.integers(k, 30))
        s = (base + base)[a:a + L]
        if len(s) >= k:
            unitigs.append(s)
    unitigs += ["A" * (k + 3), "AC" * k, "T" * k, unitigs[0]]
    assert_same_index(unitigs, k)


def test_medium_k31():
    rng = np.random.default_rng(31)
    g = random_genome(rng, 60000)
    assert_same_index(cut_unitigs(rng, g, 31, max_len=2000), 31)


def test_rejects_bad_input():
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex.build(["ACGT", "AC"], 4)          # unitig shorter than k
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex.build(["ACGTNACGT"], 4)           # PackedStrings.hh:57 throws in the reference
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex.build(["ACGT" * 80], 256)         # k limit: LCS values are kept in a byte, as in the reference


@pytest.mark.parametrize("k", [15, 200])   # (k > 128: container version 5, with the exact LCS array beside the 7-bit node bytes)
def test_save_load_roundtrip(tmp_path, k):
    rng = np.random.default_rng(5)
    g = random_genome(rng, 5000) + "ACGT" * 100 + random_genome(rng, 300) + "ACGT" * 100   # (a long repeat: LCS values above 127 at k = 200)
    p = fa.FinimizerIndex.build(cut_unitigs(rng, g, k), k)
    if k > 128:
        assert int(p.export(fa.X_LCS).max()) > 127
    p.serialize(tmp_path / "idx")
    q = fa.FinimizerIndex().load(tmp_path / "idx")
    assert (q.k, q.n_nodes, q.n_kmers, q.n_unitigs, q.n_finimizers) == (p.k, p.n_nodes, p.n_kmers, p.n_unitigs, p.n_finimizers)
    for what in (fa.X_C, fa.X_PLANE_A, fa.X_PLANE_A + 3, fa.X_LCS, fa.X_FMIN, fa.X_USTART, fa.X_GOFF, fa.X_ENDS, fa.X_CONCAT):
        assert np.array_equal(p.export(what), q.export(what))
    assert q.size_in_bytes() == p.size_in_bytes() > 0
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex().load(tmp_path / "missing")


def test_property_random_string_sets():
    """hypothesis: ANY set of ACGT strings of length >= k gives the same index in the product builder and in the literal
    restatement (duplicates, repeats, shared prefixes, single unitigs, k at the word-size boundaries included)."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @st.composite
    def case(draw):
        k = draw(st.sampled_from([2, 3, 4, 5, 7, 8, 16, 31, 32, 33]))
        alpha = draw(st.sampled_from(["ACGT", "AC", "A", "ACG"]))
        n = draw(st.integers(1, 12))
        strs = [draw(st.text(alphabet=alpha, min_size=k, max_size=k + draw(st.integers(0, 40)))) for _ in range(n)]
        return k, strs

    @settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck))
    @given(case())
    def run(c):
        k, strs = c
        assert_same_index(strs, k)

    run()


@pytest.mark.parametrize("k", [5, 9, 15, 31])
def test_finimizer_statistics_modes(k):
    """build-fmin --type shortest / verify: the product's host code against the oracle's restatement; the streaming
    mode against the brute-force mode (the reference's own cross-check); and, at t = 1, against the number of finimizers of the
    index itself (the rarest type)."""
    rng = np.random.default_rng(500 + k)
    g = random_genome(rng, 4000)
    unitigs = cut_unitigs(rng, g, k, max_len=max(2 * k, 120))
    p = fa.FinimizerIndex.build(unitigs, k)
    o = OracleIndex.build(unitigs, k)
    for t in (1, 2, 4):
        ps, pv = p.finimizer_stats(unitigs, "shortest", t), p.finimizer_stats(unitigs, "verify", t)
        assert ps == o.finimizer_stats(unitigs, "shortest", t) and pv == o.finimizer_stats(unitigs, "verify", t)
        assert ps == pv, (k, t)
        if t == 1:
            assert ps[0] == p.n_finimizers and ps[1] == ps[0]
    # verify skips non-ACGT characters by cutting the sequence (remove_ns); lower case is accepted
    broken = [unitigs[0][:k + 3] + "N" + unitigs[0][k + 3:], unitigs[1].lower()]
    assert p.finimizer_stats(broken, "verify", 1) == o.finimizer_stats(broken, "verify", 1)
    with pytest.raises(Exception):
        p.finimizer_stats(["ACGT" * 40 + "N"], "shortest", 1)


def test_config2_index_every_component_vs_independent_oracle_build():
    """BASELINE config 2's index (5 Mbp unitigs, k=31): every exported component of the product's sort-based builder -- C array, the four
    planes, LCS, fmin, Ustart, global offsets, unitig ends, packed text -- equals the oracle's literal construction from the same
    unitigs (the big-index parity no longer rests on components the product exported itself)."""
    from finito_amd import synth
    g = synth.genome(5_000_000)
    u = synth.unitigs(g, 31)
    p, o = assert_same_index(u.as_tuple(), 31)
    assert p.n_nodes > 5_000_000 and p.is_disjoint()
