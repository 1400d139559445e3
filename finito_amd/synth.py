"""Seeded synthetic inputs of the benchmark configurations -- tooling around the path.

Genome iid uniform ACGT; unitigs = pieces overlapping by k-1, each reverse-complemented with p=1/2, shuffled;
reads = 150/250 bp windows, random strand, 1 % substitutions, 5 % fully random reads.  Seeds are fixed here.
"""
import ctypes as C

import numpy as np

from . import lib

SEED_GENOME, SEED_UNITIGS, SEED_READS = 1, 2, 3


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Unitigs:
    def __init__(self, bases, offsets, gstart, glen, rcflag, k):
        self.bases, self.offsets, self.gstart, self.glen, self.rc, self.k = bases, offsets, gstart, glen, rcflag, k

    def __len__(self):
        return len(self.gstart)

    def as_tuple(self):
        return (self.bases, self.offsets)

    def strings(self):
        b = self.bases.tobytes()
        return [b[int(self.offsets[i]):int(self.offsets[i + 1])].decode() for i in range(len(self))]


class Reads:
    def __init__(self, bases, offsets, gstart, rcflag, err_mask, read_len):
        self.bases, self.offsets, self.gstart, self.rc, self.err_mask, self.read_len = bases, offsets, gstart, rcflag, err_mask, read_len

    def __len__(self):
        return len(self.gstart)

    def as_tuple(self):
        return (self.bases, self.offsets)

    def subset(self, lo, hi):
        L = self.read_len
        return Reads(self.bases[lo * L:hi * L], (self.offsets[lo:hi + 1] - self.offsets[lo]).astype(np.uint64), self.gstart[lo:hi],
                     self.rc[lo:hi], self.err_mask[lo * L:hi * L], L)

    def take(self, idx):
        """the reads with the given numbers (a sample spread over the batch: bench.py's CPU legs)"""
        L = self.read_len
        idx = np.asarray(idx, dtype=np.int64)
        return Reads(np.ascontiguousarray(self.bases.reshape(-1, L)[idx]).reshape(-1), (np.arange(len(idx) + 1, dtype=np.uint64) * np.uint64(L)), self.gstart[idx],
                     self.rc[idx], np.ascontiguousarray(self.err_mask.reshape(-1, L)[idx]).reshape(-1), L)

    def strings(self):
        b = self.bases.tobytes()
        L = self.read_len
        return [b[i * L:(i + 1) * L].decode() for i in range(len(self))]


def genome(n, seed=SEED_GENOME):
    L = lib()
    out = np.empty(n, dtype=np.uint8)
    L.fin_synth_genome.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    L.fin_synth_genome(n, seed, out.ctypes.data_as(C.c_void_p))
    return out


def unitigs(g, k, max_len=4000, seed=SEED_UNITIGS):
    L = lib()
    n = len(g)
    cap = int(n // max(1, (k + max_len) // 2 - (k - 1)) * 2 + 1024)
    L.fin_synth_unitigs.restype = C.c_int64
    L.fin_synth_unitigs.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    while True:
        out_cap = n + cap * (k - 1) + 64
        bases = np.empty(out_cap, dtype=np.uint8)
        offsets = np.zeros(cap + 1, dtype=np.uint64)
        gstart = np.zeros(cap, dtype=np.uint64); glen = np.zeros(cap, dtype=np.uint32); rcf = np.zeros(cap, dtype=np.uint8)
        np_ = L.fin_synth_unitigs(g.ctypes.data_as(C.c_void_p), n, k, max_len, seed, bases.ctypes.data_as(C.c_void_p), out_cap,
                                  offsets.ctypes.data_as(C.c_void_p), gstart.ctypes.data_as(C.c_void_p),
                                  glen.ctypes.data_as(C.c_void_p), rcf.ctypes.data_as(C.c_void_p), cap)
        if np_ >= 0:
            break
        cap = int(-np_) + 16
    tot = int(offsets[np_])
    return Unitigs(bases[:tot], offsets[:np_ + 1], gstart[:np_], glen[:np_], rcf[:np_], k)


def repeat_genome(n, seed=SEED_GENOME, repeat_frac=0.45, div=(0.01, 0.10)):
    """an iid background with repeats written over it (fin_synth_repeat_genome): interspersed families whose copies diverged by
    div[0]..div[1], tandem arrays, segmental duplications -- about repeat_frac of the bases"""
    L = lib()
    out = np.empty(n, dtype=np.uint8)
    L.fin_synth_repeat_genome.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_void_p]
    L.fin_synth_repeat_genome(n, seed, float(repeat_frac), float(div[0]), float(div[1]), out.ctypes.data_as(C.c_void_p))
    return out


def spss(g, k, max_len=4000, seed=SEED_UNITIGS):
    """a DISJOINT spectrum-preserving string set of g (fin_synth_spss): every canonical k-mer kept at its first occurrence only, the
    pieces break wherever a k-mer was seen before (as the unitigs of a de Bruijn graph do).  The returned Unitigs carry dup_pos /
    dup_first (k-mer starts that are not a first occurrence -> that first occurrence) for check_ground_truth, and `multi` (per k-mer
    start: its canonical k-mer occurs more than once)."""
    L = lib()
    n = len(g)
    assert k <= 64 and n < 2 ** 32 - 1
    L.fin_synth_spss.restype = C.c_int64
    L.fin_synth_spss.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    cap = int(n // max(1, (k + max_len) // 2 - (k - 1)) * 2 + 1024)
    cap_d = 1024
    out_cap = n + cap * (k - 1) + 64
    multi = np.zeros(n, dtype=np.uint8)
    while True:
        bases = np.empty(out_cap, dtype=np.uint8)
        offsets = np.zeros(cap + 1, dtype=np.uint64)
        gstart = np.zeros(cap, dtype=np.uint64); glen = np.zeros(cap, dtype=np.uint32); rcf = np.zeros(cap, dtype=np.uint8)
        dpos = np.zeros(cap_d, dtype=np.uint32); dfirst = np.zeros(cap_d, dtype=np.uint32)
        nd = C.c_uint64(0)
        np_ = L.fin_synth_spss(g.ctypes.data_as(C.c_void_p), n, k, max_len, seed, bases.ctypes.data_as(C.c_void_p), out_cap,
                               offsets.ctypes.data_as(C.c_void_p), gstart.ctypes.data_as(C.c_void_p), glen.ctypes.data_as(C.c_void_p),
                               rcf.ctypes.data_as(C.c_void_p), cap, dpos.ctypes.data_as(C.c_void_p), dfirst.ctypes.data_as(C.c_void_p), cap_d,
                               C.byref(nd), multi.ctypes.data_as(C.c_void_p))
        if np_ >= 0:
            break
        if int(nd.value) > cap_d:
            cap_d = int(nd.value) + 16
        else:
            cap = int(-np_) + 16
            out_cap = max(out_cap, int(offsets[0]) + 64)
    tot = int(offsets[np_])
    u = Unitigs(bases[:tot], offsets[:np_ + 1], gstart[:np_], glen[:np_], rcf[:np_], k)
    u.dup_pos, u.dup_first, u.multi = dpos[:int(nd.value)], dfirst[:int(nd.value)], multi
    return u


def kmer_multiplicity(g, k):
    """per k-mer start of g: 1 if its canonical k-mer occurs more than once (used to skip those in the ground truth of a set that is
    not disjoint)"""
    return spss(g, k).multi


def reads(g, n_reads, read_len=150, err_rate=0.01, random_frac=0.05, seed=SEED_READS, first=0):
    """records [first, first + n_reads) of the read set `seed` names (a record depends on its number only)"""
    L = lib()
    bases = np.empty(n_reads * read_len, dtype=np.uint8)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    gstart = np.zeros(n_reads, dtype=np.int64); rcf = np.zeros(n_reads, dtype=np.uint8)
    em = np.zeros(n_reads * read_len, dtype=np.uint8)
    L.fin_synth_reads_at.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fin_synth_reads_at(g.ctypes.data_as(C.c_void_p), len(g), first, n_reads, read_len, err_rate, random_frac, seed,
                         bases.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), gstart.ctypes.data_as(C.c_void_p),
                         rcf.ctypes.data_as(C.c_void_p), em.ctypes.data_as(C.c_void_p))
    return Reads(bases, offsets, gstart, rcf, em, read_len)


def unitig_ids(index, u):
    """Id the index gave each generated piece: rank of its first k-mer in colex order, ties by input order
    (permute_unitigs, PackedStrings.hh:105-135)."""
    k = u.k
    n = len(u)
    starts = u.offsets[:-1].astype(np.int64)
    first = u.bases[(starts[:, None] + np.arange(k - 1, -1, -1)[None, :])]      # reversed first k-mers
    keys = np.ascontiguousarray(first).view("S%d" % k).ravel()
    order = np.argsort(keys, kind="stable")
    ids = np.empty(n, dtype=np.uint32)
    ids[order] = np.arange(n, dtype=np.uint32)
    return ids


def check_ground_truth(index, u, r, pairs, skip=None):
    """(mismatches, checked, first_bad_read): every error-free k-mer of a genome-derived read must come back as the
    piece that holds it -- a size-independent property usable at full benchmark size.  Unitigs made by spss(): a k-mer that is not
    the first occurrence of its canonical k-mer must come back as the piece that holds the first one.  skip (bytes per k-mer start
    of the genome): positions that are not checked (a set that is not disjoint)."""
    L = lib()
    ids = unitig_ids(index, u)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    checked = C.c_uint64(0); first = C.c_int64(-1)
    dpos = getattr(u, "dup_pos", None)
    if dpos is None and skip is None:
        L.fin_synth_check.restype = C.c_int64
        L.fin_synth_check.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
        bad = L.fin_synth_check(len(u), u.gstart.ctypes.data_as(C.c_void_p), u.glen.ctypes.data_as(C.c_void_p), u.rc.ctypes.data_as(C.c_void_p),
                                ids.ctypes.data_as(C.c_void_p), index.k, len(r), r.read_len, r.gstart.ctypes.data_as(C.c_void_p),
                                r.rc.ctypes.data_as(C.c_void_p), r.err_mask.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p),
                                C.byref(checked), C.byref(first))
        return int(bad), int(checked.value), int(first.value)
    if dpos is None:
        dpos = np.zeros(0, dtype=np.uint32)
    dfirst = getattr(u, "dup_first", np.zeros(0, dtype=np.uint32))
    L.fin_synth_check2.restype = C.c_int64
    L.fin_synth_check2.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
    bad = L.fin_synth_check2(len(u), u.gstart.ctypes.data_as(C.c_void_p), u.glen.ctypes.data_as(C.c_void_p), u.rc.ctypes.data_as(C.c_void_p),
                             ids.ctypes.data_as(C.c_void_p), index.k, len(r), r.read_len, r.gstart.ctypes.data_as(C.c_void_p),
                             r.rc.ctypes.data_as(C.c_void_p), r.err_mask.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p),
                             dpos.ctypes.data_as(C.c_void_p), dfirst.ctypes.data_as(C.c_void_p), len(dpos),
                             skip.ctypes.data_as(C.c_void_p) if skip is not None else None, C.byref(checked), C.byref(first))
    return int(bad), int(checked.value), int(first.value)
