"""Results as records (round 5; include/finito_amd.h: fin_read_record): the host-side expander against a brute-force statement of the record's meaning
(CPU), and -- on the GPU -- records + stream of the fast path's read mix expanded to exactly the pairs fin_search_batch delivers and the oracle computes."""
import numpy as np
import pytest

import finito_amd as fa


def brute_expand(recs, stream, k):
    out, sp = [], 0
    for r in recs:
        nk, kind = int(r["nk"]), int(r["meta"]) >> 16
        if kind == 0:
            out += [tuple(x) for x in stream[sp:sp + nk]]; sp += nk
        elif kind == 2:
            out += [(-1, -1)] * nk
        else:
            rev, nE = (int(r["meta"]) >> 8) & 1, int(r["meta"]) & 0xFF
            Es = [(int(r["Es"] if e < 4 else r["Es2"]) >> (16 * (e & 3))) & 0xFFFF for e in range(nE)]
            for i in range(nk):
                sl = nk - 1 - i if rev else i
                out.append((-1, -1) if any(sl <= E <= sl + k - 1 for E in Es) else (int(r["u"]), int(r["off0"]) + sl))
    assert sp == len(stream)
    return np.array(out, dtype=np.int32).reshape(-1, 2)


def test_expander_against_the_records_meaning():
    rng = np.random.default_rng(5)
    for k in (4, 21, 31, 63):
        recs = np.zeros(3000, dtype=fa.RECORD_DTYPE)
        stream = []
        for r in recs:
            nk = int(rng.integers(0, 260)); kind = int(rng.integers(0, 3))
            r["nk"] = nk
            if kind == 0:
                stream += [(int(rng.integers(-1, 50)),) * 2 for _ in range(nk)]
                continue
            nE = int(rng.integers(0, 9)) if kind == 1 else 0
            Es = sorted(int(x) for x in rng.integers(0, nk + k - 1, nE)) if nk else []
            nE = len(Es)
            r["u"], r["off0"], r["meta"] = int(rng.integers(0, 1000)), int(rng.integers(0, 5000)), nE | (int(rng.integers(0, 2)) << 8) | (kind << 16)
            r["Es"] = sum(E << (16 * e) for e, E in enumerate(Es[:4])); r["Es2"] = sum(E << (16 * e) for e, E in enumerate(Es[4:]))
        stream = np.array(stream, dtype=np.int32).reshape(-1, 2)
        want = brute_expand(recs, stream, k)
        for threads in (1, 3, 0):
            got, npos = fa.expand_records(recs, stream, k, n_threads=threads)
            assert np.array_equal(got, want) and npos == int((want[:, 0] != -1).sum())
        with pytest.raises(fa.FinitoError):   # a stream that is not this record set's
            fa.expand_records(recs, stream[:-1] if len(stream) else np.zeros((1, 2), np.int32), k)


@pytest.mark.gpu
def test_records_of_a_batch_expand_to_the_pairs():
    from oracle.oracle import OracleIndex
    from tests.test_search_gpu import _fast_path_reads
    from tests.util import cut_unitigs, random_genome, rc
    L = fa.lib()
    rng = np.random.default_rng(77)
    for case, k in enumerate((31, 21, 47, 63, 31)):
        g = random_genome(rng, 40000)
        if case == 4:     # duplicated stretches and reverse-complement copies: unsafe places, flagged windows
            for _ in range(5):
                a = int(rng.integers(0, len(g) - 300)); n = int(rng.integers(k + 3, 300)); at = int(rng.integers(0, len(g)))
                g = g[:at] + g[a:a + n] + g[at:]
        unitigs = cut_unitigs(rng, g, k, max_len=900, flip=bool(case % 2))
        if case == 4:
            unitigs += [rc(g[a:a + 200]) for a in (1000, 7000)]
        p = fa.FinimizerIndex.build(unitigs, k).to_device(0)
        o = OracleIndex.build(unitigs, k)
        reads = _fast_path_reads(rng, g, k, unitigs) + ["", "ACGT", g[100:100 + k - 1]]
        exp, _, _ = o.search_batch(reads)
        for sub in (1 << 26, 20000):   # one device batch; several sub-batches whose streams concatenate
            assert L.fin_set_option(b"pipeline_kmers", sub) == 0
            try:
                recs, stream = p.search_reads_records(reads)
            finally:
                L.fin_set_option(b"pipeline_kmers", 1 << 26)
            kinds = recs["meta"] >> 16
            assert (kinds == 1).sum() > 0.4 * len(reads) and (kinds == 0).sum() > 0 and len(stream) < 0.6 * len(exp)   # most reads travel as 32 bytes
            got, npos = fa.expand_records(recs, stream, k)
            assert np.array_equal(got.astype(np.int64), exp), "case %d k=%d sub-batch %d" % (case, k, sub)
            assert npos == int((exp[:, 0] != -1).sum())
        # without the fast path every read is kind 0 and the stream is the pairs
        L.fin_set_option(b"fast_path", 0)
        try:
            recs, stream = p.search_reads_records(reads[:400])
        finally:
            L.fin_set_option(b"fast_path", 1)
        assert (recs["meta"] >> 16 == 0).all()
        assert np.array_equal(fa.expand_records(recs, stream, k)[0].astype(np.int64), o.search_batch(reads[:400])[0])
        p.close()
