"""Pins the CPU oracle against every known-answer vector the reference's own tests hold
(/root/reference/src/tests.cpp:62-317, committed as tests/golden/reference_kat.json)."""
import numpy as np
import pytest

from oracle.oracle import OracleIndex, format_pairs


def _case(kat, name):
    return next(c for c in kat if c["name"] == name)


def _colex_rank_of_unitig(unitigs, k, q):
    # tests.cpp:216-237 get_unitig_ranks: rank by reversed first k-mer
    order = sorted(unitigs, key=lambda s: s[:k][::-1])
    return order.index(q)


@pytest.mark.parametrize("name", [
    "test_shortest_unique_construction", "test_shortest_unique_queries", "test_finimizer_branch",
    "test_reverse_complement_branch", "test_leftmost", "test_finimizer_selection", "test_incoming_rc_branch",
    "test_reverse_complement_query", "test_walk", "example_fna_k4"])
def test_reference_vectors(kat, name):
    c = _case(kat, name)
    idx = OracleIndex.build(c["unitigs"], c["k"])
    if "labels" in c:
        assert idx.labels() == c["labels"]
    if "n_nodes" in c:
        assert idx.n_nodes == c["n_nodes"]
    if "C" in c:
        assert idx.C_array().tolist() == c["C"]
    if "lcs" in c:
        assert idx.lcs().tolist() == c["lcs"]
    if "concat" in c:
        assert idx.concat().tolist() == c["concat"]
    if "ends" in c:
        assert idx.ends().tolist() == c["ends"]
    if "fmin" in c:
        assert idx.fmin().tolist() == c["fmin"]
    if "global_offsets" in c:
        assert idx.global_offsets().tolist() == c["global_offsets"]
    if "ustart" in c:
        assert idx.ustart().tolist() == c["ustart"]
    for q in c.get("queries", []):
        pairs, n_found = idx.search(q["q"])
        if q.get("pairs_rank_of_query_unitig"):
            assert pairs == [(_colex_rank_of_unitig(c["unitigs"], c["k"], q["q"]), 0)]
        else:
            assert pairs == [tuple(p) for p in q["pairs"]]
        if "n_found" in q:
            assert n_found == q["n_found"]
    for q in c.get("merged_queries", []):
        assert idx.search_merged(q["q"]) == [tuple(p) for p in q["pairs"]]


def test_walk_is_load_bearing(kat):
    """in test_walk the unitig holds CCGT twice; (0,3) at index 6 only comes from walking."""
    c = _case(kat, "test_walk")
    idx = OracleIndex.build(c["unitigs"], c["k"])
    pairs, _ = idx.search(c["queries"][0]["q"])
    assert pairs[6] == (0, 3)


def test_output_text_format():
    # search_fmin.hh:62-65
    assert format_pairs([(0, 2), (-1, -1), (0, 0)]) == "(0,2) (-1,-1) (0,0)\n"
    assert format_pairs([]) == "\n"


def test_short_and_empty_reads(kat):
    c = _case(kat, "test_leftmost")
    idx = OracleIndex.build(c["unitigs"], c["k"])
    assert idx.search("") == ([], 0)
    assert idx.search("CGG") == ([], 0)
    assert idx.search_merged("AC") == []


def test_invalid_base_defined_behaviour(kat):
    """Reference: UB (common.hh:108-111 + FinimizerIndex.hh:150). Defined here: every k-mer overlapping the bad base is (-1,-1)."""
    c = _case(kat, "test_leftmost")
    idx = OracleIndex.build(c["unitigs"], c["k"])
    good, _ = idx.search("CGGTTACCC")
    bad, _ = idx.search("CGGTNACCC")
    assert bad[0] == good[0]
    assert bad[1:5] == [(-1, -1)] * 4
    assert len(bad) == len(good)
    low, _ = idx.search("cggttaccc")
    assert low == good


def test_brute_force_cross_check_random_dspss():
    """Independent check (idea of ref_implementation/src/minimizer_index.rs:465-479): hash map k-mer -> (unitig, offset)."""
    rng = np.random.default_rng(7)
    for k in (5, 9, 15):
        N = 3000
        genome = "".join("ACGT"[x] for x in rng.integers(0, 4, N))
        # distinct k-mers only: cut at first repeated k-mer occurrence boundaries
        seen, pieces, start = set(), [], 0
        i = 0
        while i + k <= N:
            km = genome[i:i + k]
            if km in seen:
                if i + k - 1 - start >= k:
                    pieces.append(genome[start:i + k - 1])
                start = i + 1
            else:
                seen.add(km)
            i += 1
        if N - start >= k:
            pieces.append(genome[start:])
        # verify disjointness and rebuild the set that is really present
        table = {}
        rc = lambda s: s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        pieces = [p if rng.random() < 0.5 else rc(p) for p in pieces]
        flat = [p[j:j + k] for p in pieces for j in range(len(p) - k + 1)]
        if len(set(flat)) != len(flat):
            # rc flips may introduce collisions; keep only a disjoint prefix
            keep, present = [], set()
            for p in pieces:
                kms = [p[j:j + k] for j in range(len(p) - k + 1)]
                if len(set(kms)) == len(kms) and not (set(kms) & present):
                    keep.append(p); present |= set(kms)
            pieces = keep
        idx = OracleIndex.build(pieces, k)
        order = sorted(range(len(pieces)), key=lambda u: (pieces[u][:k][::-1], u))
        for newid, u in enumerate(order):
            for j in range(len(pieces[u]) - k + 1):
                table[pieces[u][j:j + k]] = (newid, j)
        assert idx.n_kmers == len(table)
        for _ in range(60):
            a = int(rng.integers(0, N - 60))
            read = list(genome[a:a + 60])
            for _ in range(int(rng.integers(0, 3))):
                read[int(rng.integers(0, 60))] = "ACGT"[int(rng.integers(0, 4))]
            read = "".join(read)
            got = idx.search_merged(read)
            exp = []
            for j in range(len(read) - k + 1):
                f = table.get(read[j:j + k], (-1, -1))
                if f[0] == -1:
                    f = table.get(rc(read[j:j + k]), (-1, -1))
                    if f[0] != -1:
                        pass
                exp.append(f)
            # forward hit wins; otherwise the rc strand's hit, reported in the unitig's own coordinates
            assert got == exp
