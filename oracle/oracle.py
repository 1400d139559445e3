"""ctypes binding of the CPU ORACLE (oracle/finito_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (finito_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "finito_oracle.c")
    hdr = os.path.join(_HERE, "finito_oracle.h")
    lazy = os.path.join(_HERE, "finito_lazy.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(lazy)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return so


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "base_strands", "kmers", "found", "extends", "rank_lines", "drops", "lcs_entries", "lcs_lines",
        "anchors", "walked", "max_deque", "max_deque_eager")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    def algorithmic_bytes(self):
        """64*(rank_lines + lcs_lines + 4*anchors + walked/256) + in + 8*out."""
        return (64 * (self.rank_lines + self.lcs_lines + 4 * self.anchors + (self.walked + 255) // 256)
                + self.base_strands + 8 * self.kmers)


class LazyCounters(C.Structure):
    """fo_lazy_counters: work of the lazy algorithm (finito_lazy.c); algorithmic_bytes() is the model stated in finito_oracle.h"""
    _fields_ = [(n, C.c_int64) for n in (
        "reads", "strands", "strands_searched", "bases", "kmers", "found", "chunks_packed",
        "table_entries", "probe_extends", "probe_lines", "chunks_probe",
        "stream_steps", "stream_lines", "chunks_search", "anchors", "walk_bases", "text_windows",
        "restarts_short", "restarts_failed_check", "restarts_k1", "restarts_full_margin", "restarts_margin",
        "jump_entries", "jumped_bases", "text_anchors", "prepass_entries", "prepass_lines", "filter_checks", "full_anchors", "seed_lookups", "seed_anchors", "seed_verdicts", "unsafe_places", "safe_checks", "ktab_lookups", "deferred_strands", "deferred_slots", "full_lookups", "full_lines", "full_entries", "bridge_lines", "bridge_entries", "uend_lines", "uend_entries", "uend_probes", "prepass_ktab",
        "fast_reads", "fast_absent_reads", "fast_tries", "fast_looks", "fast_chunks", "fast_text_words", "fast_cbf", "fast_redesc", "fast_looks2",
        "fbf_lookups", "prepass_fbf", "place_anchors")]
    # (round 5: a look-up of the compact k-mer table is one 32-byte bucket whatever k is; an anchor it claims is compared with the text -- 16 bytes -- behind its locate)
    MODEL = ("128*(probe_lines+stream_lines) + 8*(table_entries+jump_entries) + 40*anchors + 16*seed_lookups + 16*text_windows + 8*safe_checks + 32*ktab_lookups "
             "+ 16*(chunks_probe+chunks_search) + 8*filter_checks + 8*strands + 8*seed_verdicts + 16*reads + bases + 16*chunks_packed + 8*kmers "
             "+ 16*(fast_chunks+fast_cbf+fast_redesc) + 32*(fast_looks+fast_looks2) + 8*fast_text_words + 20*fast_tries + 16*fbf_lookups + 36*place_anchors  [oracle/finito_oracle.h, fo_lazy_counters]")

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    def parts(self):
        return {"node_lines": 128 * (self.probe_lines + self.stream_lines), "prefix_table": 8 * (self.table_entries + self.jump_entries),
                "dictionaries": 40 * self.anchors + 16 * self.seed_lookups, "kmer_table": 32 * self.ktab_lookups, "unitig_text": 16 * self.text_windows + 8 * self.safe_checks,
                "read_chunks": 16 * (self.chunks_probe + self.chunks_search), "per_read": 8 * self.strands + 8 * self.seed_verdicts + 16 * self.reads, "absence_filter": 8 * self.filter_checks,
                "ingest": self.bases + 16 * self.chunks_packed, "output": 8 * self.kmers, "fast_path": self.fast_bytes(),
                "string_filter_probes": 16 * self.fbf_lookups, "locates": 36 * self.place_anchors}

    def fast_bytes(self):
        """the pre-pass's fast path (round 4): later looks, chunks, text words, string-filter blocks, locates"""
        return 16 * (self.fast_chunks + self.fast_cbf + self.fast_redesc) + 32 * (self.fast_looks + self.fast_looks2) + 8 * self.fast_text_words + 20 * self.fast_tries

    def algorithmic_bytes(self):
        return sum(self.parts().values())

    def stage_bytes(self, output_in_search=False):
        """the same bytes split by the part of the step that moves them (sums to algorithmic_bytes()).  output_in_search: the output
        is not prefilled -- the search stage writes every slot once (kernel 4 with seeds); else the prefill in front writes them"""
        out_a, out_b = (0, 8 * self.kmers) if output_in_search else (8 * self.kmers, 0)
        # (the fast path writes its reads' pairs from the pre-pass: their share of the output moves with them)
        out_f = 0
        if output_in_search and self.fast_reads and self.reads:
            out_f = int(out_b * (self.fast_reads / self.reads)); out_b -= out_f
        return {"ingest_prefill": self.bases + 16 * self.chunks_packed + 16 * self.reads + out_a,
                "probe_prepass": 128 * self.prepass_lines + 8 * self.prepass_entries + 32 * self.prepass_ktab + 16 * self.chunks_probe + 8 * self.filter_checks + 8 * self.strands + 8 * self.seed_verdicts + self.fast_bytes() + out_f + 16 * self.prepass_fbf,
                "search": 128 * (self.probe_lines - self.prepass_lines + self.stream_lines) + 8 * (self.table_entries - self.prepass_entries + self.jump_entries)
                          + 40 * self.anchors + 16 * self.seed_lookups + 16 * self.text_windows + 8 * self.safe_checks + 32 * (self.ktab_lookups - self.prepass_ktab) + 16 * self.chunks_search + out_b
                          + 16 * (self.fbf_lookups - self.prepass_fbf) + 36 * self.place_anchors}


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i64, u64p, i64p, u8p, cp = C.c_void_p, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.c_char_p
        L.fo_build.restype = vp
        L.fo_build.argtypes = [cp, u64p, i64, C.c_int]
        L.fo_from_components.restype = vp
        L.fo_from_components.argtypes = [C.c_int, i64, C.POINTER(u64p), u8p, u64p, u64p, i64p, i64, u8p, i64, i64p, i64]
        L.fo_free.argtypes = [vp]
        for f in ("fo_k", "fo_n_nodes", "fo_n_kmers", "fo_n_unitigs", "fo_n_fmin", "fo_total_len", "fo_size_in_bytes"):
            getattr(L, f).restype = i64
            getattr(L, f).argtypes = [vp]
        L.fo_get_C.argtypes = [vp, i64p]
        L.fo_get_plane.argtypes = [vp, C.c_int, u8p]
        for f in ("fo_get_lcs", "fo_get_fmin", "fo_get_ustart", "fo_get_concat"):
            getattr(L, f).argtypes = [vp, u8p]
        L.fo_get_goff.argtypes = [vp, i64p]
        L.fo_get_ends.argtypes = [vp, i64p]
        L.fo_get_label.argtypes = [vp, i64, cp]
        L.fo_search.restype = i64
        L.fo_search.argtypes = [vp, cp, i64, i64p, i64p, C.POINTER(Counters)]
        L.fo_search_merged.restype = i64
        L.fo_search_merged.argtypes = [vp, cp, i64, i64p, C.POINTER(Counters)]
        L.fo_finimizer_stats.argtypes = [vp, cp, u64p, i64, C.c_int, i64, i64p]
        L.fo_search_batch.restype = C.c_double
        L.fo_search_batch.argtypes = [vp, cp, u64p, i64, i64p, C.c_int, C.c_int, C.POINTER(Counters), u64p]
        L.fo_search_batch_lazy.restype = i64
        L.fo_search_batch_lazy.argtypes = [vp, cp, u64p, i64, i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(LazyCounters)]
        L.fo_index_is_disjoint.argtypes = [vp]
        L.fo_index_rc_free.argtypes = [vp]
        L.fo_index_all_verified.argtypes = [vp]
        L.fo_format_pairs.restype = i64
        L.fo_format_pairs.argtypes = [i64p, i64, cp]
        _LIB = L
    return _LIB


def _flatten(seqs):
    if isinstance(seqs, tuple) and len(seqs) == 2 and isinstance(seqs[0], np.ndarray):
        bases, offsets = seqs
        return np.ascontiguousarray(bases, dtype=np.uint8), np.ascontiguousarray(offsets, dtype=np.uint64)
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    for i, b in enumerate(bs):
        offsets[i + 1] = offsets[i] + len(b)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    if bases.size == 0:
        bases = np.zeros(1, dtype=np.uint8)
    return bases, offsets


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class OracleIndex:
    """Mirror of the reference's FinimizerIndex (FinimizerIndex.hh:26-259) on the CPU oracle."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("oracle: index construction failed (k out of range, unitig shorter than k or non-ACGT base)")
        self.h = C.c_void_p(handle)
        self.L = lib()

    @classmethod
    def build(cls, unitigs, k):
        bases, offsets = _flatten(unitigs)
        L = lib()
        return cls(L.fo_build(bases.ctypes.data_as(C.c_char_p), _p(offsets, C.c_uint64), len(offsets) - 1, k))

    @classmethod
    def from_components(cls, k, comp):
        """comp: dict with planes (4 x uint64 words), lcs (u8), fmin/ustart (uint64 words), goff (i64), concat (u8 codes), ends (i64)."""
        L = lib()
        n = int(comp["n_nodes"])
        planes = [np.ascontiguousarray(comp["planes"][c], dtype=np.uint64) for c in range(4)]
        arr = (C.POINTER(C.c_uint64) * 4)(*[_p(p, C.c_uint64) for p in planes])
        lcs = np.ascontiguousarray(comp["lcs"], dtype=np.uint8)
        fmin = np.ascontiguousarray(comp["fmin"], dtype=np.uint64)
        ust = np.ascontiguousarray(comp["ustart"], dtype=np.uint64)
        goff = np.ascontiguousarray(comp["goff"], dtype=np.int64)
        concat = np.ascontiguousarray(comp["concat"], dtype=np.uint8)
        ends = np.ascontiguousarray(comp["ends"], dtype=np.int64)
        if goff.size == 0:
            goff = np.zeros(1, dtype=np.int64)
        h = L.fo_from_components(k, n, arr, _p(lcs, C.c_uint8), _p(fmin, C.c_uint64), _p(ust, C.c_uint64),
                                 _p(goff, C.c_int64), int(comp["n_fmin"]), _p(concat, C.c_uint8), concat.size,
                                 _p(ends, C.c_int64), ends.size)
        return cls(h)

    def __del__(self):
        try:
            if self.h:
                self.L.fo_free(self.h)
                self.h = None
        except Exception:
            pass

    # --- scalar properties
    @property
    def k(self): return self.L.fo_k(self.h)
    @property
    def n_nodes(self): return self.L.fo_n_nodes(self.h)
    @property
    def n_kmers(self): return self.L.fo_n_kmers(self.h)
    @property
    def n_unitigs(self): return self.L.fo_n_unitigs(self.h)
    @property
    def n_fmin(self): return self.L.fo_n_fmin(self.h)
    @property
    def total_len(self): return self.L.fo_total_len(self.h)
    def size_in_bytes(self): return self.L.fo_size_in_bytes(self.h)

    def is_disjoint(self):
        """every k-mer of the index has exactly one place in the unitigs (only known for indexes made by build())"""
        return bool(self.L.fo_index_is_disjoint(self.h))

    # --- components
    def C_array(self):
        a = np.zeros(4, dtype=np.int64); self.L.fo_get_C(self.h, _p(a, C.c_int64)); return a
    def plane(self, c):
        a = np.zeros(self.n_nodes, dtype=np.uint8); self.L.fo_get_plane(self.h, c, _p(a, C.c_uint8)); return a
    def _u8(self, fn, n):
        a = np.zeros(max(n, 1), dtype=np.uint8); getattr(self.L, fn)(self.h, _p(a, C.c_uint8)); return a[:n]
    def lcs(self): return self._u8("fo_get_lcs", self.n_nodes)
    def fmin(self): return self._u8("fo_get_fmin", self.n_nodes)
    def ustart(self): return self._u8("fo_get_ustart", self.n_nodes)
    def concat(self): return self._u8("fo_get_concat", self.total_len)
    def global_offsets(self):
        a = np.zeros(max(self.n_fmin, 1), dtype=np.int64); self.L.fo_get_goff(self.h, _p(a, C.c_int64)); return a[:self.n_fmin]
    def ends(self):
        a = np.zeros(max(self.n_unitigs, 1), dtype=np.int64); self.L.fo_get_ends(self.h, _p(a, C.c_int64)); return a[:self.n_unitigs]
    def labels(self):
        out = []
        buf = C.create_string_buffer(int(self.k) + 1)
        for i in range(self.n_nodes):
            if self.L.fo_get_label(self.h, i, buf) != 0:
                return None
            out.append(buf.raw[:self.k].decode())
        return out

    # --- queries
    def search(self, q, counters=None):
        """FinimizerIndex::search: returns (list of (unitig, offset), n_found)."""
        qb = q.encode() if isinstance(q, str) else bytes(q)
        nk = max(0, len(qb) - self.k + 1)
        out = np.zeros(2 * nk + 2, dtype=np.int64)
        nf = C.c_int64(0)
        n = self.L.fo_search(self.h, qb, len(qb), _p(out, C.c_int64), C.byref(nf), C.byref(counters) if counters is not None else None)
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)], int(nf.value)

    def search_merged(self, q, counters=None):
        qb = q.encode() if isinstance(q, str) else bytes(q)
        nk = max(0, len(qb) - self.k + 1)
        out = np.zeros(2 * nk + 2, dtype=np.int64)
        n = self.L.fo_search_merged(self.h, qb, len(qb), _p(out, C.c_int64), C.byref(counters) if counters is not None else None)
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]

    def finimizer_stats(self, seqs, kind="shortest", t=1):
        """build-fmin --type shortest / verify (build_fmin.hh:95-214): (number of distinct finimizers, sum of frequencies, sum of lengths)"""
        bases, offsets = _flatten(seqs)
        out = np.zeros(3, dtype=np.int64)
        rc = self.L.fo_finimizer_stats(self.h, bases.ctypes.data_as(C.c_char_p), _p(offsets, C.c_uint64), len(offsets) - 1,
                                       1 if kind == "shortest" else 2, t, _p(out, C.c_int64))
        if rc != 0:
            raise ValueError("a sequence leaves the index")
        return tuple(int(v) for v in out)

    def search_batch(self, reads, want_pairs=True, format_text=False, n_threads=1, counters=None):
        """run_fmin_queries_streaming over a batch: returns (pairs[int64, n_kmers x 2] or None, seconds, text_checksum)."""
        bases, offsets = _flatten(reads)
        k = self.k
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        nk = int(np.maximum(lens - k + 1, 0).sum())
        out = np.zeros((max(nk, 1), 2), dtype=np.int64) if want_pairs else None
        cs = C.c_uint64(0)
        secs = self.L.fo_search_batch(self.h, bases.ctypes.data_as(C.c_char_p), _p(offsets, C.c_uint64), len(lens),
                                      _p(out, C.c_int64) if want_pairs else None, int(format_text), int(n_threads),
                                      C.byref(counters) if counters is not None else None, C.byref(cs))
        return (out[:nk] if want_pairs else None), float(secs), int(cs.value)


def _lazy(self, reads, ptab_t=0, jump_t=0, disjoint=True, counters=None, n_threads=1, seeds=None, filt_f=0, count_safe_checks=False, kmer_table=None, defer=None, rc_pairs=None, fast=None, lean=False):
    """The lazy algorithm of the product's kernels restated on the CPU (finito_lazy.c): merged pairs [n_kmers, 2] int64.
    disjoint: text re-anchoring (the name is round 2's, when it needed a disjoint index; since round 3 the place of every k-mer found
    by text comparison is checked against the reference's answer, so any index may use it); seeds: anchors through unique probe
    strings (default: as disjoint -- what kernel 4 does; kernel 3 has no seeds); filt_f: depth of the pre-pass's absence filter
    (0: none; below k); count_safe_checks: the index has unsafe places, i.e. the device carries the bitmap and reads it."""
    bases, offsets = _flatten(reads)
    lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
    nk = int(np.maximum(lens - self.k + 1, 0).sum())
    out = np.zeros((max(nk, 1), 2), dtype=np.int64)
    if seeds is None:
        seeds = disjoint
    if defer is None:   # the second strand of a read only where the first left slots open (kernel 4 on any index with an anchor table; CHANGELOG.md 4.14)
        defer = bool(seeds)
    if rc_pairs is None:   # does the index hold a k-mer and its reverse complement?  (the device counts them at upload: fin_index_rc_pairs)
        rc_pairs = bool(defer) and not bool(self.L.fo_index_rc_free(self.h))
    if kmer_table is None:   # what the device does: the compact table exists at every k on replicas with an anchor pass (round 5: the walk kernel asks it above 63 too)
        kmer_table = bool(seeds)
    if fast is None:   # the pre-pass's fast path (round 4): what the device does wherever it has the k-mer table and defers second strands
        fast = bool(kmer_table) and bool(defer) and self.k <= 63
    flags = int(bool(disjoint)) | (2 if seeds else 0) | (4 if count_safe_checks else 0) | (8 if kmer_table else 0) | (16 if defer else 0) | (32 if (defer and rc_pairs) else 0) | (64 if fast else 0) | (128 if (lean and seeds and kmer_table) else 0) | ((int(filt_f) & 0xFF) << 8)
    n = self.L.fo_search_batch_lazy(self.h, bases.ctypes.data_as(C.c_char_p), _p(offsets, C.c_uint64), len(lens), _p(out, C.c_int64),
                                    int(ptab_t), int(jump_t), flags, int(n_threads), C.byref(counters) if counters is not None else None)
    assert n == nk
    return out[:nk]


OracleIndex.search_batch_lazy = _lazy


def format_pairs(pairs):
    """The reference's output line for one read (search_fmin.hh:62-65)."""
    return " ".join("(%d,%d)" % (int(u), int(p)) for u, p in pairs) + "\n"
