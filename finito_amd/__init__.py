"""finito_amd -- MI355X-native search-fmin k-mer localization (host binding over the C ABI).

Thin ctypes plumbing over libfinito_amd.so (include/finito_amd.h).  The class and function names mirror the
reference's interface for this path: FinimizerIndex.{search, load, serialize, size_in_bytes}
(include/FinimizerIndex.hh:26-259) and run_fmin_queries_streaming (include/search_fmin.hh:33-84).

There is no CPU search path in this package: if the HIP library is missing or no device is present the query
calls raise.  The CPU oracle under oracle/ is test infrastructure and is never imported from here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libfinito_amd.so")
_LIB = None

FIN_FWD, FIN_MERGED = 0, 1
FIN_OK, FIN_EINVAL, FIN_EIO, FIN_ENODEV, FIN_ENOMEM, FIN_ELIMIT = 0, -1, -2, -3, -4, -5   # include/finito_amd.h
X_C, X_PLANE_A, X_LCS, X_FMIN, X_USTART, X_GOFF, X_ENDS, X_CONCAT = 0, 1, 5, 6, 7, 8, 9, 10


class FinitoError(RuntimeError):
    """std::runtime_error of the reference (caught in src/main.cpp:51-57)."""

    def __init__(self, code, msg):
        super().__init__("%s (code %d)" % (msg, code))
        self.code = code


def host_threads():
    """Usable host cores: affinity mask capped by the cgroup CPU quota (a GPU box hands each GPU a share of its cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("FINITO_THREADS", "64"))))


def build_native(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    cmd = ["make", "-s", "-C", src, "all"]
    if force:
        cmd.insert(1, "-B")
    subprocess.check_call(cmd)
    return _LIBPATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIBPATH):
            raise FinitoError(-3, "libfinito_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                                  "there is no fallback path")
        L = C.CDLL(_LIBPATH)
        vp, i64, u64, cp = C.c_void_p, C.c_int64, C.c_uint64, C.c_char_p
        u64p, i64p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        L.fin_version.restype = cp
        L.fin_host_threads.restype = C.c_int
        L.fin_index_build.argtypes = [cp, u64p, u64, C.c_int, C.c_int, C.POINTER(vp), cp, C.c_size_t]
        L.fin_index_save.argtypes = [vp, cp, cp, C.c_size_t]
        L.fin_index_load.argtypes = [cp, C.POINTER(vp), cp, C.c_size_t]
        L.fin_index_free.argtypes = [vp]
        L.fin_index_save_reference_layout.argtypes = [vp, cp, cp, C.c_size_t]
        L.fin_index_load_reference_layout.argtypes = [cp, C.POINTER(vp), cp, C.c_size_t]
        L.fin_index_save_sbwt.argtypes = [vp, cp, cp, C.c_size_t]
        L.fin_sbwt_file_info.argtypes = [cp, i64p, i64p, i64p, cp, C.c_size_t]
        L.fin_index_check_against_files.argtypes = [vp, cp, cp, cp, C.c_size_t]
        for f in ("fin_index_k", "fin_index_n_nodes", "fin_index_n_kmers", "fin_index_n_unitigs", "fin_index_n_finimizers",
                  "fin_index_total_len", "fin_index_size_in_bytes"):
            getattr(L, f).restype = i64
            getattr(L, f).argtypes = [vp]
        L.fin_index_export_size.restype = i64
        L.fin_index_export_size.argtypes = [vp, C.c_int]
        L.fin_index_export.argtypes = [vp, C.c_int, vp, u64, cp, C.c_size_t]
        L.fin_index_to_device.argtypes = [vp, C.c_int, cp, C.c_size_t]
        L.fin_index_prefix_table_depth.argtypes = [vp, C.c_int]
        L.fin_index_jump_table_depth.argtypes = [vp, C.c_int]
        L.fin_index_filter_depth.argtypes = [vp, C.c_int]
        L.fin_index_seed_table_bytes.argtypes = [vp, C.c_int]
        L.fin_index_seed_table_bytes.restype = C.c_int64
        L.fin_index_is_disjoint.argtypes = [vp]
        L.fin_index_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.fin_index_clear_option.argtypes = [vp, C.c_char_p]
        L.fin_index_kmer_table_bytes.argtypes = [vp, C.c_int]
        L.fin_index_kmer_table_bytes.restype = C.c_int64
        L.fin_index_string_filter_bytes.argtypes = [vp, C.c_int]
        L.fin_index_string_filter_bytes.restype = C.c_int64
        L.fin_index_replica_table_bytes.argtypes = [vp, C.c_int]
        L.fin_index_replica_table_bytes.restype = C.c_int64
        L.fin_index_rc_pairs.argtypes = [vp, C.c_int]
        L.fin_index_rc_pairs.restype = C.c_int64
        L.fin_index_unsafe_places.argtypes = [vp, C.c_int]
        L.fin_index_unsafe_places.restype = C.c_int64
        L.fin_index_anchor_build_ms.argtypes = [vp, C.c_int]
        L.fin_index_anchor_build_ms.restype = C.c_double
        L.fin_index_debug_seed_table.argtypes = [vp, C.c_int, vp, cp, C.c_size_t]
        L.fin_index_finimizer_stats.argtypes = [vp, cp, u64p, u64, C.c_int, i64, i64p, i64p, i64p, cp, C.c_size_t]
        L.fin_search.argtypes = [vp, cp, i64, i64p, i64p, cp, C.c_size_t]
        L.fin_search_batch.argtypes = [vp, cp, u64p, u64, C.c_int, i32p, u64p, cp, C.c_size_t]
        L.fin_batch_create.argtypes = [vp, cp, u64p, u64, C.POINTER(vp), cp, C.c_size_t]
        L.fin_batch_create_on.argtypes = [vp, C.c_int, cp, u64p, u64, C.POINTER(vp), cp, C.c_size_t]
        L.fin_search_batch_multi.argtypes = [vp, C.POINTER(C.c_int), C.c_int, cp, u64p, u64, C.c_int, i32p, u64p, cp, C.c_size_t]
        L.fin_device_count.restype = C.c_int
        L.fin_host_alloc.restype = vp
        L.fin_host_alloc.argtypes = [C.c_size_t]
        L.fin_host_free.argtypes = [vp]
        L.fin_batch_run.argtypes = [vp, C.c_int, vp, cp, C.c_size_t]
        L.fin_batch_reload.argtypes = [vp, cp, u64p, u64, cp, C.c_size_t]
        L.fin_batch_n_kmers.restype = u64
        L.fin_batch_n_kmers.argtypes = [vp]
        L.fin_batch_n_base_strands.restype = u64
        L.fin_batch_n_base_strands.argtypes = [vp]
        L.fin_batch_set_pairs.argtypes = [vp, C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]
        L.fin_batch_device_pairs.restype = vp
        L.fin_batch_device_pairs.argtypes = [vp]
        L.fin_batch_download.argtypes = [vp, i32p, u64p, cp, C.c_size_t]
        L.fin_batch_kernel_time.argtypes = [vp, C.POINTER(C.c_double), u64p]
        L.fin_batch_format_text.argtypes = [vp, u64p, cp, C.c_size_t]
        L.fin_batch_download_text.argtypes = [vp, vp, cp, C.c_size_t]
        L.fin_batch_text_mode.argtypes = [vp, C.c_int]
        L.fin_text_create.restype = vp
        L.fin_text_free.argtypes = [vp]
        L.fin_text_data.restype = vp
        L.fin_text_data.argtypes = [vp]
        L.fin_text_size.restype = u64
        L.fin_text_size.argtypes = [vp]
        L.fin_search_batch_text.argtypes = [vp, cp, u64p, u64, C.c_int, vp, u64p, cp, C.c_size_t]
        L.fin_batch_download_range.argtypes = [vp, u64, u64, i32p, cp, C.c_size_t]
        L.fin_batch_step_time.argtypes = [vp, u64, C.POINTER(C.c_double), u64p]
        L.fin_batch_free.argtypes = [vp]
        L.fin_batch_overflow_reads.restype = i64
        L.fin_batch_overflow_reads.argtypes = [vp]
        L.fin_format_pairs.restype = i64
        L.fin_format_pairs.argtypes = [i32p, i64, cp]
        _LIB = L
    return _LIB


def sbwt_file_info(path):
    """(k, nodes, k-mers) of an SBWT file written by `sbwt build` (plain-matrix variant)"""
    k, n, m = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    err = C.create_string_buffer(512)
    _check(lib().fin_sbwt_file_info(str(path).encode(), C.byref(k), C.byref(n), C.byref(m), err, 512), err)
    return int(k.value), int(n.value), int(m.value)


def _check(rc, errbuf):
    if rc != 0:
        raise FinitoError(rc, errbuf.value.decode(errors="replace") or "finito_amd call failed")


def flatten(seqs):
    """list of str/bytes -> (uint8 bases, uint64 offsets[n+1]); (bases, offsets) arrays pass through."""
    if isinstance(seqs, tuple) and len(seqs) == 2 and isinstance(seqs[0], np.ndarray):
        return np.ascontiguousarray(seqs[0], dtype=np.uint8), np.ascontiguousarray(seqs[1], dtype=np.uint64)
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    joined = b"".join(bs)
    bases = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(1, dtype=np.uint8)
    return bases, offsets


class PinnedArray:
    """numpy view of page-locked host memory (fin_host_alloc): PCIe copies to/from it run at link speed."""

    def __init__(self, shape, dtype):
        self.L = lib()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self.L.fin_host_alloc(max(n, 1))
        if not self.ptr:
            raise FinitoError(-4, "fin_host_alloc failed (no HIP device or out of pinned memory)")
        buf = (C.c_char * max(n, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.L.fin_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QueryResult:
    """FinimizerIndex::QueryResult (FinimizerIndex.hh:30-33)."""

    def __init__(self, local_offsets, n_found):
        self.local_offsets = local_offsets
        self.n_found = n_found


class Batch:
    """Reads resident in HBM with their output buffer (fin_batch_* of the C ABI)."""

    def __init__(self, index, reads):
        self.index = index
        self.L = lib()
        bases, offsets = flatten(reads)
        self._keep = (bases, offsets)
        self.n_reads = len(offsets) - 1
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_create(index.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       self.n_reads, C.byref(h), err, 512), err)
        self.h = h
        self._keep = None   # the reads live in HBM now

    def reload(self, reads):
        """Replace the reads of this batch, keeping its device buffers (fin_batch_reload)."""
        bases, offsets = flatten(reads)
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_reload(self.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       len(offsets) - 1, err, 512), err)
        self.n_reads = len(offsets) - 1

    @property
    def n_kmers(self):
        return int(self.L.fin_batch_n_kmers(self.h))

    @property
    def n_base_strands(self):
        return int(self.L.fin_batch_n_base_strands(self.h))

    def run(self, strands=FIN_MERGED, stream=None):
        """Enqueue the search on a HIP stream (int handle, e.g. torch.cuda.current_stream().cuda_stream); no sync."""
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_run(self.h, strands, C.c_void_p(stream or 0), err, 512), err)

    def download(self, want_pairs=True, want_positive=True):
        n = self.n_kmers
        out = np.empty((max(n, 1), 2), dtype=np.int32) if want_pairs else None
        npos = C.c_uint64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_download(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)) if want_pairs else None,
                                         C.byref(npos) if want_positive else None, err, 512), err)
        return (out[:n] if want_pairs else None), int(npos.value)

    def text(self):
        """the reference's output text of this batch's pairs, formatted on the device (fin_batch_format_text): bytes"""
        n = C.c_uint64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_format_text(self.h, C.byref(n), err, 512), err)
        buf = C.create_string_buffer(max(int(n.value), 1))
        _check(self.L.fin_batch_download_text(self.h, buf, err, 512), err)
        return buf.raw[:int(n.value)]

    def text_mode(self, mode):
        """0: pairs (default); 1: pairs + the fast path's per-read records (the text is made from them); 2: text only (fin_batch_text_mode)"""
        if self.L.fin_batch_text_mode(self.h, int(mode)) != 0:
            raise FinitoError("fin_batch_text_mode(%r)" % (mode,))

    def format_text(self):
        """format the reference's output text of this batch's pairs on the device and leave it there (fin_batch_format_text): bytes
        of text.  Runs on the stream of the last run(); returns after the text is complete."""
        n = C.c_uint64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_format_text(self.h, C.byref(n), err, 512), err)
        return int(n.value)

    def download_range(self, first_pair, n_pairs):
        """int32 pairs [first_pair, first_pair + n_pairs) of the output (fin_batch_download_range)"""
        out = np.empty((max(n_pairs, 1), 2), dtype=np.int32)
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_download_range(self.h, int(first_pair), int(n_pairs), out.ctypes.data_as(C.POINTER(C.c_int32)), err, 512), err)
        return out[:n_pairs]

    def set_pairs(self, pairs):
        """diagnostic: overwrite the batch's pairs in HBM (fin_batch_set_pairs) -- for tests of the text formatter"""
        a = np.ascontiguousarray(pairs, dtype=np.int32)
        err = C.create_string_buffer(512)
        _check(self.L.fin_batch_set_pairs(self.h, a.ctypes.data_as(C.POINTER(C.c_int32)), err, 512), err)

    def device_pairs_ptr(self):
        return int(self.L.fin_batch_device_pairs(self.h) or 0)

    def pipeline_counts(self, n=64):
        """kernel 4's queue counters of the last run (fin_batch_pipeline_counts)"""
        out = (C.c_uint32 * n)()
        self.L.fin_batch_pipeline_counts(self.h, out, n)
        return list(out)

    def overflow_reads(self):
        return int(self.L.fin_batch_overflow_reads(self.h))

    def run_info(self):
        """what the most recent run decided (fin_batch_run_info): {'kernel', 'no_prefill', 'deferred', 'fast_path'}"""
        out = (C.c_uint32 * 4)()
        self.L.fin_batch_run_info(self.h, out)
        return {"kernel": int(out[0]), "no_prefill": bool(out[1]), "deferred": bool(out[2]), "fast_path": bool(out[3])}

    def kernel_time_ms(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        self.L.fin_batch_kernel_time(self.h, C.byref(ms), C.byref(n))
        return float(ms.value), int(n.value)

    def step_time_ms(self, skip_first=0):
        """({'ingest_prefill', 'probe_prepass', 'search', 'overflow_tail', 'step'} -> ms averaged over the runs after the first
        `skip_first`, number of runs): device time of a step from HIP events on the launch stream (fin_batch_step_time)"""
        p, n = (C.c_double * 5)(), C.c_uint64(0)
        self.L.fin_batch_step_time(self.h, int(skip_first), p, C.byref(n))
        return dict(zip(("ingest_prefill", "probe_prepass", "search", "overflow_tail", "step"), (float(x) for x in p))), int(n.value)

    def close(self):
        if getattr(self, "h", None):
            self.L.fin_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FinimizerIndex:
    """Mirror of the reference class (FinimizerIndex.hh:26-259) backed by the HIP path."""

    def __init__(self, handle=None):
        self.L = lib()
        self.h = handle

    # -- construction / persistence ---------------------------------------------------------------------------
    @classmethod
    def build(cls, unitigs, k, n_threads=0):
        """FinimizerIndexBuilder (FinimizerIndex.hh:262-395) + `sbwt build` + LCS, in one call."""
        L = lib()
        bases, offsets = flatten(unitigs)
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        _check(L.fin_index_build(bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 len(offsets) - 1, int(k), int(n_threads) if n_threads > 0 else host_threads(), C.byref(h), err, 512), err)
        return cls(h)

    @classmethod
    def build_on_device(cls, unitigs, k, device=0):
        """the same index built on a HIP device (fin_index_build_device; every k <= 255): bit-identical to build()'s.  The stages' device
        times in milliseconds are left in .build_phase_ms"""
        L = lib()
        bases, offsets = flatten(unitigs)
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        ph = (C.c_double * 8)()
        L.fin_index_build_device.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
        _check(L.fin_index_build_device(bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1, int(k), int(device),
                                        C.byref(h), ph, err, 512), err)
        x = cls(h)
        x.build_phase_ms = dict(zip(("upload_kmers", "sort_unique", "dummies", "sbwt", "unitigs", "finimizers", "dictionaries", "copy_back"), (float(v) for v in ph)))
        return x

    def load(self, index_prefix):
        """FinimizerIndex::load (FinimizerIndex.hh:209-241)."""
        self.close()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_load(str(index_prefix).encode(), C.byref(h), err, 512), err)
        self.h = h
        return self

    def serialize(self, index_prefix):
        """FinimizerIndex::serialize (FinimizerIndex.hh:187-207)."""
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_save(self.h, str(index_prefix).encode(), err, 512), err)

    def serialize_reference_layout(self, index_prefix):
        """FinimizerIndex::serialize in the reference's own seven-file layout (FinimizerIndex.hh:187-207); parity unpinned."""
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_save_reference_layout(self.h, str(index_prefix).encode(), err, 512), err)

    def load_reference_layout(self, index_prefix):
        self.close()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_load_reference_layout(str(index_prefix).encode(), C.byref(h), err, 512), err)
        self.h = h
        return self

    def save_sbwt(self, path):
        """the SBWT as `sbwt build` writes it (what build-fmin -i reads)"""
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_save_sbwt(self.h, str(path).encode(), err, 512), err)

    def check_against_files(self, sbwt_path=None, lcs_path=None):
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_check_against_files(self.h, str(sbwt_path).encode() if sbwt_path else None,
                                                    str(lcs_path).encode() if lcs_path else None, err, 512), err)

    def size_in_bytes(self):
        return int(self.L.fin_index_size_in_bytes(self.h))

    def finimizer_stats(self, seqs, kind="shortest", t=1):
        """build-fmin --type shortest / verify (build_fmin.hh:95-214): (distinct finimizers, sum of frequencies, sum of lengths)."""
        bases, offsets = flatten(seqs)
        n, sf, sl = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_finimizer_stats(self.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                len(offsets) - 1, 1 if kind == "shortest" else 2, int(t), C.byref(n), C.byref(sf), C.byref(sl), err, 512), err)
        return int(n.value), int(sf.value), int(sl.value)

    def prefix_table_depth(self, device=0):
        """T of the 4^T-entry prefix table the device replica carries for the kernel's probe mode (0: none)."""
        return int(self.L.fin_index_prefix_table_depth(self.h, int(device)))

    def is_disjoint(self):
        """every k-mer of the index has exactly one place in the unitigs (fin_index_is_disjoint)"""
        return bool(self.L.fin_index_is_disjoint(self.h))

    def seed_table(self, device=0):
        """the device replica's anchor table as a numpy array [n_nodes, 2] of u32: [:, 0] = the reference's answer for the node's k-mer
        (offset of its last base in the concatenated unitigs; 0xFFFFFFFF: none; 0xFFFFFF00 | d: a dummy node), [:, 1] = the entry's
        unitig with the top bit set when the text at that place does not spell the k-mer (unverified).  None when that replica has no
        table (option seed_anchors 0 at upload)"""
        import numpy as np
        out = np.empty((self.n_nodes, 2), dtype=np.uint32)
        err = C.create_string_buffer(512)
        rc = self.L.fin_index_debug_seed_table(self.h, int(device), out.ctypes.data_as(C.c_void_p), err, 512)
        return out if rc == 0 else None

    def kmer_table_query(self, kmers, device=0):
        """fin_index_debug_kmer_table: what the compact k-mer table claims about each k-mer (strings over ACGT of length k): (g, flags) arrays"""
        k = self.k
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        k0 = np.zeros(len(kmers), dtype=np.uint64); k1 = np.zeros(len(kmers), dtype=np.uint64)
        for i, s in enumerate(kmers):
            a = b = 0
            for j, ch in enumerate(s):
                if j < 32: a |= code[ch] << (2 * j)
                else: b |= code[ch] << (2 * (j - 32))
            k0[i] = a; k1[i] = b
        out = np.zeros((len(kmers), 2), dtype=np.uint32)
        err = C.create_string_buffer(512)
        self.L.fin_index_debug_kmer_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_char_p, C.c_size_t]
        _check(self.L.fin_index_debug_kmer_table(self.h, int(device), k0.ctypes.data_as(C.c_void_p), k1.ctypes.data_as(C.c_void_p), len(kmers), out.ctypes.data_as(C.c_void_p), err, 512), err)
        return out[:, 0].copy(), out[:, 1].copy()

    def seed_table_bytes(self, device=0):
        """bytes of the anchor table the device replica carries (0: none -- option seed_anchors 0 at upload)"""
        return int(self.L.fin_index_seed_table_bytes(self.h, int(device)))

    def set_option(self, name, value):
        """fin_index_set_option: this handle's own value of a tuning switch (None: follow the process-wide value again)"""
        nm = name.encode() if isinstance(name, str) else name
        rc = self.L.fin_index_clear_option(self.h, nm) if value is None else self.L.fin_index_set_option(self.h, nm, int(value))
        if rc != 0:
            raise FinitoError(rc, "bad option %r = %r" % (name, value))
        return self

    def kmer_table_bytes(self, device=0):
        """bytes of the k-mer table (text k-mer -> SBWT node) the device replica carries (0: none -- k > 31, or option kmer_table 0 at upload)"""
        return int(self.L.fin_index_kmer_table_bytes(self.h, int(device)))

    def string_filter_bytes(self, device=0):
        """bytes of the canonical string filter the device replica carries (round 4: the fast path's absence proofs; 0: none)"""
        return int(self.L.fin_index_string_filter_bytes(self.h, int(device)))

    def replica_table_bytes(self, device=0):
        """HBM the device replica occupies beyond the index arrays: every derived table, filter and bitmap (fin_index_replica_table_bytes)"""
        return int(self.L.fin_index_replica_table_bytes(self.h, int(device)))

    def unsafe_places(self, device=0):
        """k-mer positions of the unitig text that are not the place the reference reports for their k-mer (0 on disjoint unitigs;
        -1: not computed) -- fin_index_unsafe_places"""
        return int(self.L.fin_index_unsafe_places(self.h, int(device)))

    def unverified_kmers(self, device=0):
        """k-mer places whose k-mer's answer is a place that does not spell it (kept whole in the k-mer table's exact side table); -1: no anchor pass"""
        self.L.fin_index_unverified_kmers.restype = C.c_int64
        self.L.fin_index_unverified_kmers.argtypes = [C.c_void_p, C.c_int]
        return int(self.L.fin_index_unverified_kmers(self.h, device))

    def rc_pairs(self, device=0):
        """k-mers of the unitig text whose reverse complement is in the index too (fin_index_rc_pairs; -1: not counted)"""
        return int(self.L.fin_index_rc_pairs(self.h, int(device)))

    def defers_second_strand(self, device=0):
        """kernel 4 may search a read's second strand only where the first left slots open on this replica (option defer_strand aside)"""
        return self.rc_pairs(device) >= 0 and (self.seed_table_bytes(device) > 0 or self.lean_tables(device))

    def lean_tables(self, device=0):
        """the replica was uploaded with "lean_tables" (k <= 31, the default): k-mer table + string filters, no prefix table, no anchor table"""
        return self.seed_table_bytes(device) == 0 and self.kmer_table_bytes(device) > 0 and self.string_filter_bytes(device) > 0 and self.prefix_table_depth(device) == 0

    def anchor_build_ms(self, device=0):
        return float(self.L.fin_index_anchor_build_ms(self.h, int(device)))

    def filter_depth(self, device=0):
        """F of the 4^F-bit absence filter the device replica carries for the pre-pass (0: none)."""
        return int(self.L.fin_index_filter_depth(self.h, int(device)))

    def jump_table_depth(self, device=0):
        """J of the 4^J-entry jump table the device replica carries for (re)started streaming searches (0: none)."""
        return int(self.L.fin_index_jump_table_depth(self.h, int(device)))

    def to_device(self, device=0):
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_to_device(self.h, int(device), err, 512), err)
        return self

    # -- scalar members -----------------------------------------------------------------------------------------
    @property
    def k(self): return int(self.L.fin_index_k(self.h))
    @property
    def n_nodes(self): return int(self.L.fin_index_n_nodes(self.h))
    @property
    def n_kmers(self): return int(self.L.fin_index_n_kmers(self.h))
    @property
    def n_unitigs(self): return int(self.L.fin_index_n_unitigs(self.h))
    @property
    def n_finimizers(self): return int(self.L.fin_index_n_finimizers(self.h))
    @property
    def total_len(self): return int(self.L.fin_index_total_len(self.h))

    def export(self, what):
        """Decoded view of a public member of the reference class (FinimizerIndex.hh:108-115)."""
        nbytes = int(self.L.fin_index_export_size(self.h, what))
        dt = {X_C: np.int64, X_LCS: np.uint8, X_GOFF: np.int64, X_ENDS: np.int64, X_CONCAT: np.uint8}.get(what, np.uint64)
        out = np.zeros(max(nbytes // np.dtype(dt).itemsize, 1), dtype=dt)
        err = C.create_string_buffer(512)
        _check(self.L.fin_index_export(self.h, what, out.ctypes.data_as(C.c_void_p), out.nbytes, err, 512), err)
        return out[: nbytes // np.dtype(dt).itemsize]

    def components(self):
        """Everything the oracle needs to assemble the same index (tests / cpu_baseline at large sizes)."""
        return {"n_nodes": self.n_nodes, "planes": [self.export(X_PLANE_A + c) for c in range(4)], "lcs": self.export(X_LCS),
                "fmin": self.export(X_FMIN), "ustart": self.export(X_USTART), "goff": self.export(X_GOFF),
                "n_fmin": self.n_finimizers, "concat": self.export(X_CONCAT), "ends": self.export(X_ENDS)}

    # -- queries ------------------------------------------------------------------------------------------------
    def search(self, query):
        """FinimizerIndex::search(const std::string&) (FinimizerIndex.hh:119): one strand of one read."""
        qb = query.encode() if isinstance(query, str) else bytes(query)
        nk = max(0, len(qb) - self.k + 1)
        out = np.zeros(2 * nk + 2, dtype=np.int64)
        nf = C.c_int64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_search(self.h, qb, len(qb), out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nf), err, 512), err)
        return QueryResult([(int(out[2 * i]), int(out[2 * i + 1])) for i in range(nk)], int(nf.value))

    def search_reads(self, reads, strands=FIN_MERGED, out=None):
        """run_fmin_queries_streaming (search_fmin.hh:33-84) over host buffers (fin_search_batch): (int32 pairs [n_kmers, 2],
        total_positive).  `out` may be a preallocated (e.g. PinnedArray(...).array) int32 [>= n_kmers, 2] buffer."""
        bases, offsets = flatten(reads)
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        nk = int(np.maximum(lens - self.k + 1, 0).sum())
        if out is None:
            out = np.empty((max(nk, 1), 2), dtype=np.int32)
        assert out.dtype == np.int32 and out.flags["C_CONTIGUOUS"] and out.shape[0] >= max(nk, 1)
        npos = C.c_uint64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_search_batch(self.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       len(lens), int(strands), out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(npos), err, 512), err)
        return out[:nk], int(npos.value)

    def search_reads_records(self, reads):
        """fin_search_batch_records: (records, stream) -- a 32-byte record per read (structured array: u, off0, meta, nk, Es, Es2) and the pairs of the
        reads whose record says "nk pairs follow" (kind 0), back to back"""
        bases, offsets = flatten(reads)
        n = len(offsets) - 1
        nk = int(np.maximum(np.diff(offsets.astype(np.int64)) - self.k + 1, 0).sum())
        recs = np.zeros(n, dtype=RECORD_DTYPE)
        stream = np.empty((max(nk, 1), 2), dtype=np.int32)
        got = C.c_uint64(0)
        err = C.create_string_buffer(512)
        self.L.fin_search_batch_records.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        _check(self.L.fin_search_batch_records(self.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                               recs.ctypes.data_as(C.c_void_p), stream.ctypes.data_as(C.c_void_p), nk, C.byref(got), err, 512), err)
        return recs, stream[: int(got.value)]

    def search_reads_text(self, reads, strands=FIN_MERGED):
        """run_fmin_queries_streaming with its printed text as the result (fin_search_batch_text): (bytes, total_positive)"""
        bases, offsets = flatten(reads)
        t = self.L.fin_text_create()
        try:
            npos = C.c_uint64(0)
            err = C.create_string_buffer(512)
            _check(self.L.fin_search_batch_text(self.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                len(offsets) - 1, int(strands), t, C.byref(npos), err, 512), err)
            n = int(self.L.fin_text_size(t))
            return C.string_at(self.L.fin_text_data(t), n) if n else b"", int(npos.value)
        finally:
            self.L.fin_text_free(t)

    def search_reads_multi(self, reads, devices, strands=FIN_MERGED):
        """fin_search_batch_multi: the same loop with the reads sharded by record over several GPUs (index replicated)."""
        bases, offsets = flatten(reads)
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        nk = int(np.maximum(lens - self.k + 1, 0).sum())
        out = np.empty((max(nk, 1), 2), dtype=np.int32)
        npos = C.c_uint64(0)
        err = C.create_string_buffer(512)
        devs = (C.c_int * len(devices))(*devices)
        _check(self.L.fin_search_batch_multi(self.h, devs, len(devices), bases.ctypes.data_as(C.c_char_p),
                                             offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(lens), int(strands),
                                             out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(npos), err, 512), err)
        return out[:nk], int(npos.value)

    def batch(self, reads):
        return Batch(self, reads)

    def close(self):
        if getattr(self, "h", None):
            self.L.fin_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


RECORD_DTYPE = np.dtype([("u", np.uint32), ("off0", np.uint32), ("meta", np.uint32), ("nk", np.uint32), ("Es", np.uint64), ("Es2", np.uint64)])


class PartitionedBatch:
    """A read set resident on the device of a PartitionedIndex (fin_pbatch_*): run() = every part's step and its merge."""

    def __init__(self, pindex, reads):
        self.L = lib(); self.h = C.c_void_p(); self.pindex = pindex
        bases, offsets = flatten(reads)
        err = C.create_string_buffer(512)
        _check(self.L.fin_pbatch_create(pindex.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1,
                                        C.byref(self.h), err, 512), err)
        self.n_kmers = int(self.L.fin_pbatch_n_kmers(self.h))

    def run(self, stream=None):
        err = C.create_string_buffer(512)
        _check(self.L.fin_pbatch_run(self.h, C.c_void_p(stream or 0), err, 512), err)

    def download(self, want_pairs=True):
        out = np.empty((max(self.n_kmers, 1), 2), dtype=np.int32) if want_pairs else None
        npos = C.c_uint64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_pbatch_download(self.h, out.ctypes.data_as(C.c_void_p) if want_pairs else None, C.byref(npos), err, 512), err)
        return (out[: self.n_kmers] if want_pairs else None), int(npos.value)

    def step_time_ms(self, skip_first=0):
        ms = C.c_double(0); n = C.c_uint64(0)
        self.L.fin_pbatch_step_time(self.h, int(skip_first), C.byref(ms), C.byref(n))
        return float(ms.value), int(n.value)

    def close(self):
        if self.h:
            self.L.fin_pbatch_free(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PartitionedIndex:
    """A unitig set beyond 2^32 nodes as parts of at most max_part_bases bases (fin_pindex_*; include/finito_amd.h): an ordinary index and device
    replica of each part, every read searched in every part, results those of ONE FinimizerIndex (FinimizerIndex.hh:26-259) of all the unitigs --
    for the input the reference requires, a disjoint spectrum-preserving string set (README.md:79-80), which verify=True checks on the device."""

    def __init__(self, unitigs, k, device=0, max_part_bases=0, verify=True, _load_prefix=None):
        self.L = L = lib(); self.h = C.c_void_p()
        vp, cp = C.c_void_p, C.c_char_p
        L.fin_pindex_save.argtypes = [vp, cp, cp, C.c_size_t]
        L.fin_pindex_load.argtypes = [cp, C.c_int, C.POINTER(vp), cp, C.c_size_t]
        L.fin_pindex_exists.argtypes = [cp]
        L.fin_pbatch_reload.argtypes = [vp, cp, C.POINTER(C.c_uint64), C.c_uint64, cp, C.c_size_t]
        L.fin_pindex_build_device.argtypes = [cp, C.POINTER(C.c_uint64), C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(vp), cp, C.c_size_t]
        L.fin_pindex_free.argtypes = [vp]
        L.fin_pindex_parts.argtypes = [vp]; L.fin_pindex_parts.restype = C.c_uint32
        for f in ("k", "n_nodes", "n_kmers", "n_unitigs", "total_len", "size_in_bytes", "replica_table_bytes", "shared_kmers"):
            getattr(L, "fin_pindex_" + f).argtypes = [vp]; getattr(L, "fin_pindex_" + f).restype = C.c_int64
        L.fin_pindex_verify_seconds.argtypes = [vp]; L.fin_pindex_verify_seconds.restype = C.c_double
        L.fin_pindex_part.argtypes = [vp, C.c_uint32]; L.fin_pindex_part.restype = vp
        L.fin_pindex_unitig_ids.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
        L.fin_pindex_search_batch.argtypes = [vp, cp, C.POINTER(C.c_uint64), C.c_uint64, vp, C.POINTER(C.c_uint64), cp, C.c_size_t]
        L.fin_pbatch_create.argtypes = [vp, cp, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(vp), cp, C.c_size_t]
        L.fin_pbatch_run.argtypes = [vp, vp, cp, C.c_size_t]
        L.fin_pbatch_n_kmers.argtypes = [vp]; L.fin_pbatch_n_kmers.restype = C.c_uint64
        L.fin_pbatch_device_pairs.argtypes = [vp]; L.fin_pbatch_device_pairs.restype = vp
        L.fin_pbatch_download.argtypes = [vp, vp, C.POINTER(C.c_uint64), cp, C.c_size_t]
        L.fin_pbatch_step_time.argtypes = [vp, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.fin_pbatch_free.argtypes = [vp]
        err = C.create_string_buffer(1024)
        self.device = int(device)
        if _load_prefix is not None:
            _check(L.fin_pindex_load(str(_load_prefix).encode(), int(device), C.byref(self.h), err, 1024), err)
            return
        bases, offsets = flatten(unitigs)
        _check(L.fin_pindex_build_device(bases.ctypes.data_as(cp), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1, int(k), int(device),
                                         int(max_part_bases), 1 if verify else 0, C.byref(self.h), err, 1024), err)

    @classmethod
    def load(cls, index_prefix, device=0):
        """fin_pindex_load: the parts written by serialize() (<prefix>.finparts, .p<i>.finamd, .p<i>.gid), their replicas uploaded to `device`"""
        return cls(None, 0, device=device, _load_prefix=index_prefix)

    @staticmethod
    def exists(index_prefix):
        lib().fin_pindex_exists.argtypes = [C.c_char_p]
        return bool(lib().fin_pindex_exists(str(index_prefix).encode()))

    def serialize(self, index_prefix):
        """fin_pindex_save (FinimizerIndex::serialize of every part + the manifest)"""
        err = C.create_string_buffer(512)
        _check(self.L.fin_pindex_save(self.h, str(index_prefix).encode(), err, 512), err)

    n_parts = property(lambda self: int(self.L.fin_pindex_parts(self.h)))
    k = property(lambda self: int(self.L.fin_pindex_k(self.h)))
    n_nodes = property(lambda self: int(self.L.fin_pindex_n_nodes(self.h)))
    n_kmers = property(lambda self: int(self.L.fin_pindex_n_kmers(self.h)))
    n_unitigs = property(lambda self: int(self.L.fin_pindex_n_unitigs(self.h)))
    total_len = property(lambda self: int(self.L.fin_pindex_total_len(self.h)))
    shared_kmers = property(lambda self: int(self.L.fin_pindex_shared_kmers(self.h)))
    verify_seconds = property(lambda self: float(self.L.fin_pindex_verify_seconds(self.h)))

    def size_in_bytes(self):
        return int(self.L.fin_pindex_size_in_bytes(self.h))

    def replica_table_bytes(self):
        return int(self.L.fin_pindex_replica_table_bytes(self.h))

    def part_nodes(self):
        """n_nodes of every part (each below 2^32)"""
        self.L.fin_index_n_nodes.argtypes = [C.c_void_p]
        return [int(self.L.fin_index_n_nodes(self.L.fin_pindex_part(self.h, p))) for p in range(self.n_parts)]

    def unitig_ids(self, part):
        """the set's number of each of the part's unitigs (permute_unitigs over the whole set, PackedStrings.hh:105-135)"""
        self.L.fin_index_n_unitigs.argtypes = [C.c_void_p]
        n = int(self.L.fin_index_n_unitigs(self.L.fin_pindex_part(self.h, part)))
        out = np.empty(n, dtype=np.uint32)
        if self.L.fin_pindex_unitig_ids(self.h, part, out.ctypes.data_as(C.c_void_p), n) != 0:
            raise FinitoError(FIN_EINVAL, "fin_pindex_unitig_ids")
        return out

    def search_reads(self, reads):
        """merged search of a read set in every part (fin_pindex_search_batch): (int32 pairs [n_kmers, 2], total_positive)"""
        bases, offsets = flatten(reads)
        k = self.k
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        nk = int(np.maximum(lens - k + 1, 0).sum())
        out = np.empty((max(nk, 1), 2), dtype=np.int32)
        npos = C.c_uint64(0)
        err = C.create_string_buffer(512)
        _check(self.L.fin_pindex_search_batch(self.h, bases.ctypes.data_as(C.c_char_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1,
                                              out.ctypes.data_as(C.c_void_p), C.byref(npos), err, 512), err)
        return out[:nk], int(npos.value)

    def batch(self, reads):
        return PartitionedBatch(self, reads)

    def close(self):
        if self.h:
            self.L.fin_pindex_free(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def expand_records(recs, stream, k, n_threads=0):
    """fin_expand_records (host): the pairs fin_search_batch delivers, from records + stream; returns (pairs, n_positive)"""
    L = lib()
    recs = np.ascontiguousarray(recs, dtype=RECORD_DTYPE); stream = np.ascontiguousarray(stream, dtype=np.int32)
    nk = int(recs["nk"].astype(np.int64).sum())
    out = np.empty((max(nk, 1), 2), dtype=np.int32)
    pos = C.c_uint64(0)
    L.fin_expand_records.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    rc = L.fin_expand_records(recs.ctypes.data_as(C.c_void_p), len(recs), stream.ctypes.data_as(C.c_void_p), len(stream.reshape(-1, 2)), int(k),
                              out.ctypes.data_as(C.c_void_p), C.byref(pos), int(n_threads))
    if rc != 0:
        raise FinitoError(rc, "fin_expand_records: records and stream do not belong together")
    return out[:nk], int(pos.value)


def format_pairs(pairs):
    """The reference's output line for one read: '(u,p) (u,p) ...\\n' (search_fmin.hh:62-65)."""
    p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    buf = C.create_string_buffer(24 * len(p) + 2)
    n = lib().fin_format_pairs(p.ctypes.data_as(C.POINTER(C.c_int32)), len(p), buf)
    return buf.raw[:n].decode()
