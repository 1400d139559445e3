"""the reference's on-disk index layout -- seven files <prefix>.{O,FBV,packed_unitigs,unitig_endpoints,Ustart,LCS}.sdsl
+ <prefix>.sbwt (FinimizerIndex::serialize / load, FinimizerIndex.hh:187-241).

PARITY UNPINNED: the reference tree holds no index file and no serialization test, and sdsl-lite / algbio-SBWT are absent, so
nothing here can be compared with bytes the real tools wrote.  What is checked: the byte layout the writer produces, spelled out
below field by field (sdsl-lite v2 int_vector::serialize, rank_support_v5, SBWT::serialize -- see finito_amd/csrc/fin_sdsl.cpp),
and that writer -> reader is lossless: every component of the index and every search-relevant table comes back identical."""
import os
import struct

import numpy as np
import pytest

import finito_amd as fa
from tests.util import cut_unitigs, random_genome, unpack_bits


def _u64(b, o):
    return struct.unpack_from("<Q", b, o)[0]


def _read_int_vector(b, o, fixed):
    """(values, next offset): u64 size in bits; u8 width when the width is a run-time property; ceil(bits/64) u64 words"""
    bits = _u64(b, o); o += 8
    w = fixed
    if fixed == 0:
        w = b[o]; o += 1
    nw = (bits + 63) // 64
    words = np.frombuffer(b, dtype="<u8", count=nw, offset=o)
    o += 8 * nw
    allbits = np.unpackbits(words.view(np.uint8), bitorder="little")[:bits]
    vals = allbits.reshape(-1, w).astype(np.uint64) @ (1 << np.arange(w, dtype=np.uint64)) if bits else np.zeros(0, dtype=np.uint64)
    return vals, w, o


def _components_equal(a, b):
    ca, cb = a.components(), b.components()
    assert ca["n_nodes"] == cb["n_nodes"] and ca["n_fmin"] == cb["n_fmin"]
    for key in ("lcs", "fmin", "ustart", "goff", "concat", "ends"):
        assert np.array_equal(ca[key], cb[key]), key
    for c in range(4):
        assert np.array_equal(ca["planes"][c], cb["planes"][c])
    assert np.array_equal(a.export(fa.X_C), b.export(fa.X_C))
    assert (a.k, a.n_kmers, a.n_unitigs, a.total_len) == (b.k, b.n_kmers, b.n_unitigs, b.total_len)


@pytest.mark.parametrize("k,glen", [(4, 300), (11, 5000), (31, 40000), (64, 20000), (150, 8000), (255, 6000)])
def test_reference_layout_round_trip(tmp_path, k, glen):
    rng = np.random.default_rng(k)
    g = random_genome(rng, glen)
    idx = fa.FinimizerIndex.build(cut_unitigs(rng, g, k, max_len=4 * k + 50), k)
    prefix = str(tmp_path / "ref")
    idx.serialize_reference_layout(prefix)
    for ext in (".O.sdsl", ".FBV.sdsl", ".packed_unitigs.sdsl", ".unitig_endpoints.sdsl", ".Ustart.sdsl", ".LCS.sdsl", ".sbwt"):
        assert os.path.getsize(prefix + ext) > 0
    back = fa.FinimizerIndex().load_reference_layout(prefix)
    _components_equal(idx, back)
    # FinimizerIndex::load picks the seven files up when there is no container file
    again = fa.FinimizerIndex().load(prefix)
    _components_equal(idx, again)
    # and the container written from the loaded index is byte-identical to the one written from the built index
    idx.serialize(str(tmp_path / "a")); back.serialize(str(tmp_path / "b"))
    assert open(str(tmp_path / "a.finamd"), "rb").read() == open(str(tmp_path / "b.finamd"), "rb").read()


def test_reference_layout_bytes(tmp_path, kat):
    """the layout, field by field, on the paper example (tests.cpp:16, k = 4): values from the reference's own vectors"""
    c = next(x for x in kat if x["name"] == "test_shortest_unique_construction")
    idx = fa.FinimizerIndex.build(c["unitigs"], c["k"])
    prefix = str(tmp_path / "p")
    idx.serialize_reference_layout(prefix)
    n = idx.n_nodes
    # LCS: int_vector<> of n entries packed to bits(k-1) = 2 bits (lcs_basic_parallel_algorithm.hpp:115)
    b = open(prefix + ".LCS.sdsl", "rb").read()
    vals, w, o = _read_int_vector(b, 0, 0)
    assert w == 2 and o == len(b) == 8 + 1 + 8 * ((2 * n + 63) // 64) and vals.tolist() == c["lcs"]
    # fmin / Ustart: bit_vector = u64 bit count + words
    for ext, key in ((".FBV.sdsl", "fmin"), (".Ustart.sdsl", "ustart")):
        b = open(prefix + ext, "rb").read()
        vals, w, o = _read_int_vector(b, 0, 1)
        assert o == len(b) == 8 + 8 * ((n + 63) // 64) and vals.tolist() == c[key]
    # global_offsets: width = bits(max offset) (FinimizerIndex.hh:301-306)
    b = open(prefix + ".O.sdsl", "rb").read()
    vals, w, o = _read_int_vector(b, 0, 0)
    assert vals.tolist() == c["global_offsets"] and w == max(c["global_offsets"]).bit_length() and o == len(b)
    # unitig endpoints: width 64 -- PackedStrings.hh:44 calls int_vector<>(n, 64 - clz(total length)), the two-argument constructor, whose
    # second argument is the default value; packed unitigs: int_vector<2>, A0 C1 G2 T3
    b = open(prefix + ".unitig_endpoints.sdsl", "rb").read()
    vals, w, o = _read_int_vector(b, 0, 0)
    assert vals.tolist() == c["ends"] and w == 64 and o == len(b)
    b = open(prefix + ".packed_unitigs.sdsl", "rb").read()
    vals, w, o = _read_int_vector(b, 0, 2)
    assert vals.tolist() == c["concat"] and _u64(b, 0) == 2 * len(c["concat"]) and o == len(b)
    # .sbwt: string version; 4 bit_vectors; 4 rank supports (int_vector<64>); bit_vector suffix_group_starts; vector<int64> C;
    #        vector<pair<int64,int64>> k-mer prefix table; int64 precalc_k, n_nodes, n_kmers, k
    b = open(prefix + ".sbwt", "rb").read()
    ln = struct.unpack_from("<q", b, 0)[0]; o = 8
    assert b[o:o + ln] == b"v0.1"; o += ln
    planes = []
    for _ in range(4):
        vals, w, o = _read_int_vector(b, o, 1); planes.append(vals); assert len(vals) == n
    ref_planes = [unpack_bits(idx.export(fa.X_PLANE_A + ch), n) for ch in range(4)]
    assert all(np.array_equal(planes[ch], ref_planes[ch]) for ch in range(4))
    for ch in range(4):   # rank_support_v5: 2 words per 2048 bits (+2): word 0 of a superblock = ones before it
        vals, w, o = _read_int_vector(b, o, 64)
        assert len(vals) == 2 * ((((n + 63) // 64 * 64) >> 11) + 1) and vals[0] == 0
    vals, w, o = _read_int_vector(b, o, 1); assert len(vals) == 0
    assert struct.unpack_from("<q", b, o)[0] == 32; o += 8
    assert list(struct.unpack_from("<4q", b, o)) == idx.export(fa.X_C).tolist() == [1] + [1 + int(sum(p.sum() for p in planes[:i + 1])) for i in range(3)]; o += 32
    nb = struct.unpack_from("<q", b, o)[0]; o += 8 + nb
    assert struct.unpack_from("<4q", b, o) == (0, n, idx.n_kmers, c["k"]) and o + 32 == len(b)


def test_rank_support_v5_words(tmp_path):
    """the rank-support words the writer emits follow rank_support_v5's construction: absolute count per 2048-bit superblock, then
    five 12-bit counts of the ones before each further 384-bit block"""
    rng = np.random.default_rng(7)
    g = random_genome(rng, 30000)
    idx = fa.FinimizerIndex.build(cut_unitigs(rng, g, 15, max_len=300), 15)
    prefix = str(tmp_path / "p")
    idx.serialize_reference_layout(prefix)
    b = open(prefix + ".sbwt", "rb").read()
    o = 8 + struct.unpack_from("<q", b, 0)[0]
    planes = []
    for _ in range(4):
        vals, _, o = _read_int_vector(b, o, 1); planes.append(vals.astype(np.int64))
    for ch in range(4):
        vals, _, o = _read_int_vector(b, o, 64)
        cum = np.concatenate([[0], np.cumsum(planes[ch])])
        nsb = len(planes[ch]) // 2048 + 1
        for sb in range(nsb):
            assert int(vals[2 * sb]) == int(cum[min(sb * 2048, len(planes[ch]))])
            for j in range(1, 6):
                pos = sb * 2048 + 384 * j
                if pos <= ((len(planes[ch]) + 63) // 64) * 64 - 64:   # counts are written as the scan passes a block boundary
                    rel = (int(vals[2 * sb + 1]) >> (60 - 12 * j)) & 0xFFF
                    assert rel == int(cum[min(pos, len(planes[ch]))] - cum[sb * 2048]), (ch, sb, j)


def test_sbwt_and_lcs_files_are_checked(tmp_path):
    """build-fmin -i x.sbwt --lcs f (build_fmin.hh:346-383): files that belong to these unitigs pass, others are refused"""
    rng = np.random.default_rng(3)
    g = random_genome(rng, 8000)
    u = cut_unitigs(rng, g, 21, max_len=300)
    idx = fa.FinimizerIndex.build(u, 21)
    sb = str(tmp_path / "x.sbwt")
    idx.save_sbwt(sb)
    assert fa.sbwt_file_info(sb) == (21, idx.n_nodes, idx.n_kmers)
    idx.serialize_reference_layout(str(tmp_path / "p"))
    idx.check_against_files(sb, str(tmp_path / "p.LCS.sdsl"))
    other = fa.FinimizerIndex.build(cut_unitigs(rng, random_genome(rng, 8000), 21, max_len=300), 21)
    with pytest.raises(fa.FinitoError):
        other.check_against_files(sb, None)
    with pytest.raises(fa.FinitoError):
        other.check_against_files(None, str(tmp_path / "p.LCS.sdsl"))
    with pytest.raises(fa.FinitoError):   # the index's own .sbwt has no variant string in front: not what -i expects
        fa.sbwt_file_info(str(tmp_path / "p.sbwt"))


def test_damaged_files_are_refused(tmp_path):
    rng = np.random.default_rng(5)
    idx = fa.FinimizerIndex.build(cut_unitigs(rng, random_genome(rng, 3000), 9, max_len=100), 9)
    prefix = str(tmp_path / "p")
    idx.serialize_reference_layout(prefix)
    good = open(prefix + ".LCS.sdsl", "rb").read()
    open(prefix + ".LCS.sdsl", "wb").write(good[:-8])
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex().load_reference_layout(prefix)
    open(prefix + ".LCS.sdsl", "wb").write(good)
    fa.FinimizerIndex().load_reference_layout(prefix)
    os.unlink(prefix + ".Ustart.sdsl")
    with pytest.raises(fa.FinitoError):
        fa.FinimizerIndex().load_reference_layout(prefix)
