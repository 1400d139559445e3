"""The oracle's SECOND restatement (oracle/finito_lazy.c): the lazy algorithm the product's default kernels run -- absence proofs
by probes, walk along the unitig text, verified restarts of the streaming search -- stated on the CPU independently of the device
code.  It must return exactly the pairs of the faithful restatement (which follows the reference line by line and is pinned by the
reference's own vectors, tests/test_oracle_golden.py) for every prefix-table depth; its counters feed bench.py's roofline."""
import numpy as np
import pytest

from finito_amd import synth
from oracle.oracle import Counters, LazyCounters, OracleIndex
from tests.util import DEFER_KAT, cut_unitigs, defer_family_case, mosaic_read, random_genome, rc, sample_reads

DEPTHS = (0, 1, 3, 6, 9)


def _same(o, reads, depths=DEPTHS, tag=""):
    exp, _, _ = o.search_batch(reads)
    # text re-anchoring behind sequencing errors and seeds (anchors from unique probe strings and the reference's answer for their
    # node's k-mer -- kernel 4's way) on EVERY index, disjoint or not: a place found by text comparison is only used when it is the
    # place the reference reports for that k-mer (round 3; lz_text_safe / lz_node_pos)
    modes = ((False, False), (True, False), (True, True), (False, True))
    for T in depths:
        for J in (0, 1, 2, max(1, T - 2), T + 3):   # jump-table depths below, around and above the probe table's
            for dj, sd in modes:
                F = min(o.k - 1, (0, 2, 5, 7)[(T + J) % 4])   # pre-pass absence filter off / at several depths
                got = o.search_batch_lazy(reads, ptab_t=T, jump_t=J, disjoint=dj, seeds=sd, filt_f=F)
                assert np.array_equal(got, exp), "%s lazy(T=%d, J=%d, disjoint=%s, seeds=%s, F=%d) != faithful" % (tag, T, J, dj, sd, F)
                if sd and dj and o.k <= 31 and T == depths[-1]:   # lean tables (the device's default for k <= 31): probes = exact occurrences of m-base strings, seeds = places
                    got = o.search_batch_lazy(reads, ptab_t=T, jump_t=J, disjoint=dj, seeds=sd, lean=True)
                    assert np.array_equal(got, exp), "%s lazy(lean tables, J=%d) != faithful" % (tag, J)
    return o.is_disjoint()


def test_reference_vectors_lazy(kat):
    """every query of the reference's tests (src/tests.cpp), both strands merged, through the lazy restatement"""
    for c in kat:
        o = OracleIndex.build(c["unitigs"], c["k"])
        qs = [q["q"] for q in c.get("queries", [])] + [q["q"] for q in c.get("merged_queries", [])] + list(c["unitigs"])
        _same(o, qs, tag=c["name"])
        for q in c.get("merged_queries", []):
            for T in DEPTHS:
                for J in (0, 1, 2, 3):
                    got = o.search_batch_lazy([q["q"]], ptab_t=T, jump_t=J)
                    assert got.tolist() == [list(p) for p in q["pairs"]]


def test_lazy_equals_faithful_small_indexes():
    """the fuzz families of the GPU suite (random pieces, periodic text, low complexity, unrelated strings; ragged, empty, lower-case
    and non-ACGT reads)"""
    rng = np.random.default_rng(99)
    for case in range(120):
        k = int(rng.integers(2, 14))
        mode = case % 4
        if mode == 0:
            g = random_genome(rng, int(rng.integers(60, 3000)))
            unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(k + 1, 4 * k + 40)))
        elif mode == 1:
            g = random_genome(rng, int(rng.integers(3, 25))) * 40
            unitigs = [g[a:a + L] for a, L in ((int(rng.integers(0, 100)), int(rng.integers(k, k + 60))) for _ in range(int(rng.integers(1, 30))))]
        elif mode == 2:
            g = "".join("AC"[x] for x in rng.integers(0, 2, int(rng.integers(50, 800)))) + random_genome(rng, 100)
            unitigs = cut_unitigs(rng, g, k, max_len=k + 30, flip=False)
        else:
            g = random_genome(rng, 2000)
            unitigs = [random_genome(rng, int(rng.integers(k, k + 25))) for _ in range(int(rng.integers(1, 60)))]
        unitigs = [u for u in unitigs if len(u) >= k]
        if not unitigs:
            continue
        o = OracleIndex.build(unitigs, k)
        reads = [mosaic_read(rng, g, k, 220) for _ in range(25)] + ["", "A", unitigs[0], rc(unitigs[-1]), unitigs[0].lower()]
        _same(o, reads, depths=(0, 2, 5), tag="case %d (k=%d, mode %d)" % (case, k, mode))


def test_lazy_equals_faithful_walks_restarts_probes():
    """longer k, matching stretches of every length, errors at every spacing, repeats (duplicate k-mers: the walk must follow the
    copy the reference follows), junk and N's"""
    rng = np.random.default_rng(5151)
    n_disjoint = 0
    for case in range(30):
        k = int(rng.integers(6, 41))
        g = random_genome(rng, int(rng.integers(400, 12000)))
        if case % 5 == 4:
            g = g[:len(g) // 3] * 3 + random_genome(rng, 200)
            unitigs = [g[a:a + n] for a, n in ((int(rng.integers(0, len(g) - k)), int(rng.integers(k, 5 * k + 50))) for _ in range(60))]
        else:
            unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(k + 1, 6 * k + 200)), flip=bool(case % 2))
        unitigs = [u for u in unitigs if len(u) >= k]
        o = OracleIndex.build(unitigs, k)
        reads = [mosaic_read(rng, g, k, 500) for _ in range(40)] + [g[:min(len(g), 1200)], rc(g[-700:])]
        n_disjoint += _same(o, reads, depths=(0, 3, 7), tag="case %d (k=%d)" % (case, k))
    assert n_disjoint >= 10   # (most of the random-genome cases: the text re-anchoring path was exercised)


def non_disjoint_sets(rng, case, k):
    """string sets whose k-mers do NOT all have one place: (0) a disjoint set plus a few extra pieces that repeat stretches of it
    (near-disjoint: a handful of duplicated k-mers), (1) matchtig-like pieces that overlap by more than k-1, (2) a genome with diverged
    copies of a block (interspersed repeats) cut into overlapping windows, (3) tandem repeats.  Returns (genome, unitigs)."""
    fam = case % 4
    if fam == 0:
        g = random_genome(rng, int(rng.integers(800, 6000)))
        unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(2 * k, 6 * k + 100)))
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, len(g) - 2 * k)); n = int(rng.integers(k, 3 * k))
            piece = g[a:a + n]
            unitigs.insert(int(rng.integers(0, len(unitigs) + 1)), piece if rng.random() < 0.5 else rc(piece))
    elif fam == 1:
        g = random_genome(rng, int(rng.integers(800, 5000)))
        unitigs, a = [], 0
        while a + k <= len(g):
            n = int(rng.integers(k, 4 * k + 60))
            unitigs.append(g[a:a + n])
            a += max(1, n - int(rng.integers(k - 1, 2 * k)))   # overlap of k-1 .. 2k-1 bases
    elif fam == 2:
        block = random_genome(rng, int(rng.integers(3 * k, 12 * k)))
        parts = []
        for _ in range(int(rng.integers(3, 9))):
            copy = list(block)
            div = float(rng.choice([0.0, 0.01, 0.05, 0.1]))
            for i in range(len(copy)):
                if rng.random() < div:
                    copy[i] = "ACGT"[int(rng.integers(0, 4))]
            parts.append(random_genome(rng, int(rng.integers(k, 8 * k))) + "".join(copy))
        g = "".join(parts) + random_genome(rng, 3 * k)
        unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(2 * k, 8 * k)))
    else:
        unit = random_genome(rng, int(rng.integers(2, 3 * k)))
        g = random_genome(rng, 5 * k) + unit * int(rng.integers(3, 30)) + random_genome(rng, 5 * k) + unit * 4 + random_genome(rng, 3 * k)
        unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(2 * k, 6 * k)))
    return g, [u for u in unitigs if len(u) >= k]


def test_lazy_equals_faithful_non_disjoint_families():
    """round 3: text re-anchoring and seeds on indexes with duplicated k-mers -- a place found by comparing the read with the text is
    used iff it is the place the reference reports for that k-mer; everything else goes the reference's way"""
    rng = np.random.default_rng(20261004)
    n_nd = unsafe = seeded = texted = 0
    for case in range(48):
        k = int(rng.choice([5, 8, 12, 16, 21, 31]))
        g, unitigs = non_disjoint_sets(rng, case, k)
        o = OracleIndex.build(unitigs, k)
        reads = [mosaic_read(rng, g, k, 400) for _ in range(30)] + sample_reads(rng, g, 20, min(len(g), 150), err=0.02) + [g[:min(len(g), 1500)], rc(g[-600:])]
        n_nd += not _same(o, reads, depths=(0, 4, 7), tag="case %d (k=%d)" % (case, k))
        lc = LazyCounters()
        o.search_batch_lazy(reads, ptab_t=6, jump_t=4, counters=lc)
        unsafe += lc.unsafe_places; seeded += lc.seed_anchors; texted += lc.text_anchors
    assert n_nd >= 36 and unsafe > 0 and seeded > 1000 and texted > 100   # the new paths ran, on indexes that are not disjoint


@pytest.mark.parametrize("k,read_len", [(31, 150), (63, 250)])
def test_lazy_counters_on_benchmark_shaped_input(k, read_len):
    """seeded benchmark-shaped input (SURVEY 8d generator): same pairs, and the lazy algorithm's byte count is well below the
    reference algorithm's -- which is why a roofline fraction built on the reference's bytes exceeded 1"""
    g = synth.genome(300_000)
    u = synth.unitigs(g, k)
    r = synth.reads(g, 1500, read_len=read_len)
    o = OracleIndex.build(u.as_tuple(), k)
    ctr, lc, ls = Counters(), LazyCounters(), LazyCounters()
    exp, _, _ = o.search_batch(r.as_tuple(), counters=ctr)
    # kernel 4's algorithm (seeds): nearly every anchor comes from a unique probe string, hardly a base is streamed
    assert np.array_equal(o.search_batch_lazy(r.as_tuple(), ptab_t=9, jump_t=7, counters=ls, n_threads=2, fast=False), exp)
    assert ls.seed_anchors > 0.4 * ls.strands_searched and ls.seed_anchors + ls.text_anchors + ls.anchors >= ls.strands_searched - 20 and ls.seed_anchors <= ls.seed_lookups and ls.seed_verdicts == ls.strands_searched
    assert ls.anchors < 0.1 * ls.seed_anchors and sum(ls.stage_bytes().values()) == ls.algorithmic_bytes()
    # ... with the second strand deferred (this index has no reverse-complement pairs): most reads carry an error, their other strand
    # is searched only in the slots the first left open -- and without deferral the same pairs cost more bytes
    assert ls.deferred_strands > 0.3 * ls.reads and 0 < ls.deferred_slots < 0.7 * ls.kmers
    lnd = LazyCounters()
    assert np.array_equal(o.search_batch_lazy(r.as_tuple(), ptab_t=9, jump_t=7, counters=lnd, defer=False), exp)
    assert lnd.deferred_strands == 0 and ls.algorithmic_bytes() < lnd.algorithmic_bytes() and lnd.fast_reads == 0
    # ... and with the pre-pass's FAST PATH (round 4; k <= 31): most reads are finished by one comparison with the text and a few
    # string-filter blocks -- fewer bytes again, nothing left for the walk on those reads, the 5 % of reads from nowhere proven absent whole
    lf = LazyCounters()
    assert np.array_equal(o.search_batch_lazy(r.as_tuple(), ptab_t=9, jump_t=7, counters=lf, n_threads=2), exp)
    if k <= 31:   # lean tables: the same pairs on fewer bytes -- no prefix-table entry, no node block, no anchor-table entry
        ll = LazyCounters()
        assert np.array_equal(o.search_batch_lazy(r.as_tuple(), ptab_t=9, jump_t=7, counters=ll, n_threads=2, lean=True), exp)
        assert ll.fbf_lookups > 0 and ll.table_entries == 0 and ll.probe_lines == 0 and ll.seed_lookups == 0 and ll.place_anchors > 0
        assert ll.algorithmic_bytes() < lf.algorithmic_bytes() and sum(ll.stage_bytes(output_in_search=True).values()) == ll.algorithmic_bytes()
    # (k <= 31: the looks are the k-mer table's; 32 <= k <= 63: the fast path's own two-word anchor table, 32-byte slots)
    assert lf.fast_reads > 0.75 * lf.reads and 0.03 * lf.reads < lf.fast_absent_reads < 0.08 * lf.reads
    assert lf.algorithmic_bytes() < 0.95 * ls.algorithmic_bytes() and lf.fast_bytes() > 0 and lf.fast_cbf > lf.fast_reads
    assert sum(lf.stage_bytes(output_in_search=True).values()) == lf.algorithmic_bytes() == sum(lf.parts().values())
    assert lf.strands_searched < 0.3 * ls.strands_searched and (lf.fast_looks2 > 0) == (k > 31)
    # kernel 3's algorithm (no seeds)
    got = o.search_batch_lazy(r.as_tuple(), ptab_t=9, jump_t=7, seeds=False, counters=lc, n_threads=2)
    assert np.array_equal(got, exp)
    assert ls.stream_steps < 0.25 * lc.stream_steps and ls.algorithmic_bytes() < lc.algorithmic_bytes() and lc.seed_lookups == 0
    assert o.is_disjoint() and lc.text_anchors > 0.3 * lc.anchors   # (the generator's unitigs hold every k-mer once: errors are bridged by text comparison)
    lc0 = LazyCounters()
    assert np.array_equal(o.search_batch_lazy(r.as_tuple(), ptab_t=9, jump_t=0, disjoint=True, seeds=False, counters=lc0), exp)
    lcn = LazyCounters()
    assert np.array_equal(o.search_batch_lazy(r.as_tuple(), ptab_t=9, jump_t=7, disjoint=False, counters=lcn), exp)
    assert lcn.text_anchors == 0 and lc.stream_steps < 0.85 * lcn.stream_steps
    # the jump table saves streamed bases one for one
    assert lc.jumped_bases > 0 and lc.stream_steps + lc.jumped_bases == lc0.stream_steps and lc.stream_steps < 0.95 * lc0.stream_steps
    assert lc.kmers == ctr.kmers == exp.shape[0] and lc.found == ctr.found == int((exp[:, 0] != -1).sum())
    assert lc.reads == 1500 and lc.strands == 3000 and 0 < lc.strands_searched < lc.strands
    # most hits come from walks, anchors are rare, and far fewer bases are streamed than the reference streams
    assert lc.walk_bases > 0.9 * (lc.found - lc.anchors) and lc.stream_steps < 0.6 * ctr.base_strands
    assert sum(lc.parts().values()) == lc.algorithmic_bytes() < 0.6 * ctr.algorithmic_bytes()
    assert lc.parts()["output"] == 8 * lc.kmers


@pytest.mark.parametrize("k", [19, 21, 31, 40])
def test_lazy_deferred_second_strand_every_pre_pass_route(k):
    """the pair pre-pass (deferred second strand) takes a read down one of four routes -- forward first k-mer present, reverse first k-mer
    present, forward probed to a verdict, forward absent altogether -- with the k-mer table or single probes as its look: all of them give
    the faithful restatement's pairs on indexes without reverse-complement pairs"""
    rng = np.random.default_rng(900 + k)
    g = random_genome(rng, 40000)
    o = OracleIndex.build(cut_unitigs(rng, g, k, max_len=600, flip=False), k)
    if not (o.L.fo_index_rc_free(o.h) and o.is_disjoint()):
        pytest.skip("this seed's set has a reverse-complement pair")
    reads = []
    for i in range(400):
        n = int(rng.integers(k, 260)) if i % 7 else int(rng.integers(1, k + 2))
        a = int(rng.integers(0, len(g) - n))
        r = list(g[a:a + n])
        for _ in range(int(rng.integers(0, 4))):          # sequencing errors and Ns, often inside the first or the last k-mer
            where = int(rng.integers(0, n)) if rng.random() < 0.5 else (int(rng.integers(0, min(n, k))) if rng.random() < 0.5 else n - 1 - int(rng.integers(0, min(n, k))))
            r[where] = "ACGTN"[int(rng.integers(0, 5))]
        r = "".join(r)
        if i % 11 == 0:
            r = random_genome(rng, n)                     # a read from nowhere: both strands absent altogether
        reads.append(r if rng.random() < 0.5 else rc(r))
    exp, _, _ = o.search_batch(reads)
    routes = LazyCounters()
    for T in (0, 3, 6, 9):
        for kt in (False, True):
            for F in (0, 5):
                lc = LazyCounters()
                got = o.search_batch_lazy(reads, ptab_t=T, jump_t=max(0, T - 2), seeds=True, filt_f=min(F, k - 1), kmer_table=kt, defer=True, counters=lc)
                assert np.array_equal(got, exp), "k=%d T=%d kmer_table=%s F=%d" % (k, T, kt, F)
                assert lc.deferred_strands > 100 and (lc.prepass_ktab > 0) == (kt and k <= 31)
                routes = lc
    assert routes.strands == 2 * routes.reads


def test_lazy_deferred_strand_walks_into_the_other_strands_slots():
    """a deferred FORWARD strand's walk runs past the end of its stretch into slots its reverse sister filled, and wins them
    (search_fmin.hh:54-60) -- round 3's restatement searched the deferred strand as a sub-read that ended with the stretch and differed from
    the faithful one on about 1 index set in 100 of this family (identical / near-duplicate / reverse-complement unitigs; reads, rc(reads),
    the unitigs themselves, reads that run past a unitig's end).  tools/fuzz_lazy.py runs the same generator over tens of thousands of sets."""
    for k, unitigs, read, last in DEFER_KAT:
        o = OracleIndex.build(unitigs, k)
        exp, _, _ = o.search_batch([read, rc(read)])
        assert tuple(exp[len(read) - k].tolist()) == last
        for T, J in ((0, 0), (4, 2), (3, 1)):
            for kt in (False, True):
                for defer in (True, False):
                    got = o.search_batch_lazy([read, rc(read)], ptab_t=T, jump_t=J, seeds=True, kmer_table=kt, defer=defer)
                    assert np.array_equal(got, exp), (k, T, J, kt, defer)
    # seeds on which round 3's restatement differed (found with tools/fuzz_lazy.py before the fix) + a stretch of fresh ones
    for seed in [31, 151, 181, 227, 241, 289, 435, 531, 900, 980, 991, 1321, 1370, 1413, 1481, 1793, 1795, 1871] + list(range(5000, 5150)):
        rng = np.random.default_rng(seed)
        k = int(rng.choice([7, 9, 12, 16, 21, 31, 32, 40]))
        g, unitigs, reads = defer_family_case(rng, seed, k)
        o = OracleIndex.build(unitigs, k)
        exp, _, _ = o.search_batch(reads)
        T = int(rng.choice([0, 2, 4, 6])); J = int(rng.choice([0, 1, 2, 3]))
        for defer in (True, False):
            for kt in ((True, False) if k <= 31 else (False,)):
                got = o.search_batch_lazy(reads, ptab_t=T, jump_t=J, seeds=True, kmer_table=kt, defer=defer)
                assert np.array_equal(got, exp), (seed, k, T, J, defer, kt)
    # 32 <= k <= 63: the two-word k-mer table serves the whole-k-mer look-ups and the fast path whatever the other tables are (round 4's last form)
    for seed in range(7000, 7040):
        rng = np.random.default_rng(seed)
        k = int(rng.choice([32, 40, 47, 63]))
        g, unitigs, reads = defer_family_case(rng, seed, k)
        o = OracleIndex.build(unitigs, k)
        exp, _, _ = o.search_batch(reads)
        T = int(rng.choice([0, 2, 4, 6])); J = int(rng.choice([0, 1, 2, 3]))
        for defer in (True, False):
            for lean in (False, True):
                got = o.search_batch_lazy(reads, ptab_t=T, jump_t=J, seeds=True, kmer_table=True, defer=defer, lean=lean)
                assert np.array_equal(got, exp), (seed, k, T, J, defer, lean)
