"""The reference's known-answer tests through the C++ mirror of its class (finito_amd/csrc/FinimizerIndex.hh), not through ctypes:
a C++ program (tests/cpp/kat_mirror.cpp) makes the calls src/tests.cpp makes -- build, the public members, search() -- on the cases
of tests/golden/reference_kat.json."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "finito_amd")


def _binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "kat_mirror")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", out, os.path.join(ROOT, "tests", "cpp", "kat_mirror.cpp"),
                           "-L" + LIBDIR, "-lfinito_amd", "-Wl,-rpath," + LIBDIR])
    return out


def _cases_text(kat):
    lines = []
    for c in kat:
        lines.append("CASE %s %d" % (c["name"], c["k"]))
        lines += ["U " + u for u in c["unitigs"]]
        for key, tag in (("lcs", "LCS"), ("concat", "CONCAT"), ("ends", "ENDS"), ("fmin", "FMIN"), ("global_offsets", "GOFF"), ("ustart", "USTART"), ("C", "CARRAY")):
            if key in c:
                lines.append(tag + " " + " ".join(str(v) for v in c[key]))
        if "n_nodes" in c:
            lines.append("NODES %d" % c["n_nodes"])
        for q in c.get("queries", []):
            if q.get("pairs_rank_of_query_unitig"):
                order = sorted(c["unitigs"], key=lambda s: s[:c["k"]][::-1])
                pairs = [[order.index(q["q"]), 0]]
            else:
                pairs = q["pairs"]
            lines.append("Q %s %d %s" % (q["q"], q.get("n_found", -1), " ".join("%d %d" % tuple(p) for p in pairs)))
        for q in c.get("merged_queries", []):
            lines.append("M %s %s" % (q["q"], " ".join("%d %d" % tuple(p) for p in q["pairs"])))
    return "\n".join(lines) + "\n"


def _run(binary, text, *args):
    p = subprocess.run([binary, *args], input=text, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.startswith("ok "), p.stdout
    return int(p.stdout.split()[1])


def test_public_members_through_the_cpp_mirror(kat, tmp_path_factory):
    """LCS, unitigs.concat / ends, fmin, global_offsets, Ustart, C array of every reference case (no GPU needed)"""
    n = _run(_binary(tmp_path_factory), _cases_text(kat), "--no-search")
    assert n >= 15


@pytest.mark.gpu
def test_reference_kats_through_the_cpp_mirror(kat, tmp_path_factory):
    """all nine reference tests, queries included, through FinimizerIndex::search of the mirror"""
    text = _cases_text(kat)
    n = _run(_binary(tmp_path_factory), text)
    assert n >= 15 + text.count("\nQ ") + text.count("\nM ")
