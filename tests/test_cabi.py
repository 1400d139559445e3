"""The C-ABI library loads and exports every symbol include/finito_amd.h declares; device-less behaviour is loud."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import finito_amd as fa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="finito_amd.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fin_[a-z_0-9]+)\s*\(", src)))


def test_every_exported_symbol_is_declared_in_a_header():
    """the library exports nothing under fin_* that no header under include/ declares (boundary: finito_amd.h; generator and checker
    tooling: finito_synth.h); kernel launchers and the like are internal (fin_launch_*, fin_v4_*, ... live in csrc/fin_kernels.h)"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "finito_amd", "libfinito_amd.so")], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("fin_")}
    internal = set(re.findall(r"\b(fin_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "finito_amd", "csrc", "fin_kernels.h")).read()))
    internal |= {"fin_debug_dump_w", "fin_debug_dump_pp", "fin_debug_time", "fin_debug_dump_time"}
    declared = set(declared_symbols()) | set(declared_symbols("finito_synth.h"))
    assert set(declared_symbols("finito_synth.h")) <= exported
    extra = sorted(n for n in exported - declared - internal if not n.startswith(("fin_build_", "fin_save_", "fin_load_", "fin_read_", "fin_check_", "fin_finish_", "fin_host_")))
    assert not extra, "exported but declared nowhere: %s" % extra


def test_all_declared_symbols_exported():
    L = fa.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libfinito_amd.so does not export %s" % n
    assert b"gfx950" in L.fin_version()


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C with no C++ or torch types."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "finito_amd.h"\nint main(void){ return fin_version() == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_format_pairs_matches_reference_text():
    # search_fmin.hh:62-65
    assert fa.format_pairs([(0, 2), (-1, -1), (0, 0)]) == "(0,2) (-1,-1) (0,0)\n"
    assert fa.format_pairs(np.zeros((0, 2), dtype=np.int32)) == "\n"
    assert fa.format_pairs([(123456, 7890123)]) == "(123456,7890123)\n"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_search_fails_loudly_without_device():
    """No CPU fallback: on a box without a HIP device the query entry points must raise, not compute."""
    idx = fa.FinimizerIndex.build(["ACGGT", "CGGTA"], 4)
    with pytest.raises(fa.FinitoError) as e:
        idx.to_device(0)
    assert e.value.code == -3
    with pytest.raises(fa.FinitoError) as e:
        idx.search("ACGGT")
    assert e.value.code == -3
    with pytest.raises(fa.FinitoError):
        idx.search_reads(["ACGGTA"])


def test_options_process_wide_and_per_handle():
    """fin_set_option / fin_index_set_option: names and ranges are checked; a handle's own value does not touch the process-wide one"""
    L = fa.lib()
    assert L.fin_set_option(b"kernel", 1) != 0 and L.fin_set_option(b"kernel", 7) != 0 and L.fin_set_option(b"no_such_option", 1) != 0
    assert L.fin_set_option(b"lds_deque_limit", 17) != 0 and L.fin_set_option(b"lds_deque_limit", 16) == 0
    idx = fa.FinimizerIndex.build(["ACGGT", "CGGTA"], 4)
    idx.set_option("kernel", 2).set_option("pipeline_depth", 1).set_option("kernel", None)
    for bad in (("kernel", 1), ("kmer_table", 2), ("nonsense", 0)):
        with pytest.raises(fa.FinitoError):
            idx.set_option(*bad)
    assert L.fin_set_option(b"kernel", 4) == 0


def test_partitioned_index_entry_points_refuse_bad_input_without_a_device(tmp_path):
    """fin_pindex_*: argument checks and file errors come before any HIP call (include/finito_amd.h; the searches themselves are GPU tests, tests/test_pindex.py)"""
    L = fa.lib()
    vp, cp = C.c_void_p, C.c_char_p
    L.fin_pindex_build_device.argtypes = [cp, C.POINTER(C.c_uint64), C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(vp), cp, C.c_size_t]
    L.fin_pindex_load.argtypes = [cp, C.c_int, C.POINTER(vp), cp, C.c_size_t]
    L.fin_pindex_exists.argtypes = [cp]
    L.fin_pindex_parts.argtypes = [vp]; L.fin_pindex_parts.restype = C.c_uint32
    L.fin_pindex_n_nodes.argtypes = [vp]; L.fin_pindex_n_nodes.restype = C.c_int64
    L.fin_pindex_free.argtypes = [vp]
    err = C.create_string_buffer(512)
    h = vp()
    bases = b"ACGTACGTACGTACGTAAAC"
    offs = (C.c_uint64 * 3)(0, 12, 20)
    assert L.fin_pindex_build_device(bases, offs, 2, 1, 0, 0, 1, C.byref(h), err, 512) == fa.FIN_EINVAL            # k < 2
    assert L.fin_pindex_build_device(bases, offs, 2, 256, 0, 0, 1, C.byref(h), err, 512) == fa.FIN_EINVAL          # k > 255
    assert L.fin_pindex_build_device(bases, offs, 0, 5, 0, 0, 1, C.byref(h), err, 512) == fa.FIN_EINVAL            # no unitigs
    assert L.fin_pindex_build_device(bases, offs, 2, 9, 0, 0, 1, C.byref(h), err, 512) == fa.FIN_EINVAL and b"shorter than k" in err.value
    assert L.fin_pindex_build_device(bases, offs, 2, 5, 0, 10, 1, C.byref(h), err, 512) == fa.FIN_ELIMIT and b"longer than a part" in err.value
    assert L.fin_pindex_exists(str(tmp_path / "nothing").encode()) == 0
    assert L.fin_pindex_load(str(tmp_path / "nothing").encode(), 0, C.byref(h), err, 512) == fa.FIN_EIO
    (tmp_path / "bad.finparts").write_text("not a manifest\n")
    assert L.fin_pindex_exists(str(tmp_path / "bad").encode()) == 1
    assert L.fin_pindex_load(str(tmp_path / "bad").encode(), 0, C.byref(h), err, 512) == fa.FIN_EIO and b"manifest" in err.value
    (tmp_path / "half.finparts").write_text("finito-parts 1\nk 31\nparts 2\nunitigs 10\nshared_kmers 0\npart 0 first_unitig 0 unitigs 5\npart 1 first_unitig 5 unitigs 5\n")
    assert L.fin_pindex_load(str(tmp_path / "half").encode(), 0, C.byref(h), err, 512) != 0   # its parts' containers are not there
    assert not h.value
    assert L.fin_pindex_parts(None) == 0 and L.fin_pindex_n_nodes(None) == -1
    L.fin_pindex_free(None)
