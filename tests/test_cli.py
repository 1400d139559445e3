"""The `finito` command keeps the reference's command-line surface (src/main.cpp:21-59, build_fmin.hh:302, search_fmin.hh:130)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import finito_amd as fa
from oracle.oracle import OracleIndex, format_pairs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "finito_amd", "finito")
EXAMPLE = [("2", "ACAGGTA"), ("3", "GTAGGAAA"), ("1", "GTAAGTCT")]   # ref_implementation/example.fna (k=4 example of the paper)


def run(*args):
    return subprocess.run([BIN, *args], capture_output=True, text=True)


def write_fasta(path, recs):
    with open(path, "w") as f:
        for n, s in recs:
            f.write(">%s\n%s\n" % (n, s))


def test_no_arguments_prints_help_and_exits_1():
    r = run()
    assert r.returncode == 1 and "build-fmin" in r.stderr and "search-fmin" in r.stderr      # main.cpp:30-33
    for cmd in ("build-fmin", "search-fmin"):
        r = run(cmd)
        assert r.returncode == 1 and "Usage" in r.stderr                                      # build_fmin.hh:329-332
    r = run("frobnicate")
    assert r.returncode == 1 and "Runtime error: Invalid command: frobnicate" in r.stderr     # main.cpp:48,52
    r = run("build-fmin", "-o", "/tmp/x", "-u", "/nonexistent.fna")
    assert r.returncode == 1 and "Runtime error" in r.stderr


def test_build_fmin_example_matches_oracle(tmp_path):
    fna = tmp_path / "example.fna"
    write_fasta(fna, EXAMPLE)
    r = run("build-fmin", "-o", str(tmp_path / "idx"), "-u", str(fna), "-k", "4")
    assert r.returncode == 0, r.stderr
    assert os.path.exists(tmp_path / "idx.finamd") and os.path.exists(str(tmp_path / "idx") + "_stats.txt")
    p = fa.FinimizerIndex().load(tmp_path / "idx")
    o = OracleIndex.build([s for _, s in EXAMPLE], 4)
    assert p.n_nodes == o.n_nodes == 18 and p.export(fa.X_C).tolist() == [1, 8, 10, 15]
    assert np.array_equal(p.export(fa.X_LCS), o.lcs()) and np.array_equal(p.export(fa.X_GOFF), o.global_offsets())
    # t != 1 is rejected like the reference (build_fmin.hh:245-247); gzip and FASTQ inputs are read
    r = run("build-fmin", "-o", str(tmp_path / "idx2"), "-u", str(fna), "-k", "4", "-t", "2")
    assert r.returncode == 1 and "t != 1 does not make sense with rarest type" in r.stderr
    fq = tmp_path / "u.fq.gz"
    with gzip.open(fq, "wt") as f:
        for n, s in EXAMPLE:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    r = run("build-fmin", "-o", str(tmp_path / "idx3"), "-u", str(fq), "-k", "4")
    assert r.returncode == 0, r.stderr
    q = fa.FinimizerIndex().load(tmp_path / "idx3")
    assert np.array_equal(q.export(fa.X_LCS), p.export(fa.X_LCS))


def test_build_fmin_statistics_types(tmp_path):
    """--type shortest / verify print print_finimizer_stats' lines and append "t,count,sum_freq,avg_freq,avg_len,n_kmers" to
    <out>_stats.txt without writing an index (build_fmin.hh:252-268, 386-399; common.hh:188-206)."""
    fna = tmp_path / "example.fna"
    write_fasta(fna, EXAMPLE)
    o = OracleIndex.build([s for _, s in EXAMPLE], 4)
    for kind in ("shortest", "verify"):
        n, sf, sl = o.finimizer_stats([s for _, s in EXAMPLE], kind, 2)
        out = tmp_path / ("st_" + kind)
        r = run("build-fmin", "-o", str(out), "-u", str(fna), "-k", "4", "--type", kind, "-t", "2")
        assert r.returncode == 0, r.stderr
        assert "#Distinct finimizers: %d" % n in r.stderr and "Sum of frequencies: %d" % sf in r.stderr and "Avg length: " in r.stderr
        assert not os.path.exists(str(out) + ".finamd")
        line = open(str(out) + "_stats.txt").read().strip().split(",")
        assert line[0] == "2" and int(line[1]) == n and int(line[2]) == sf and abs(float(line[4]) - sl / n) < 1e-5 and int(line[5]) == o.n_kmers
    r = run("build-fmin", "-o", str(tmp_path / "x"), "-u", str(fna), "-k", "4", "--type", "fastest")
    assert r.returncode == 1 and "unknown type" in r.stderr


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_search_fmin_without_device_is_a_runtime_error(tmp_path):
    fna = tmp_path / "example.fna"
    write_fasta(fna, EXAMPLE)
    assert run("build-fmin", "-o", str(tmp_path / "idx"), "-u", str(fna), "-k", "4").returncode == 0
    r = run("search-fmin", "-i", str(tmp_path / "idx"), "-q", str(fna))
    assert r.returncode == 1 and "Runtime error" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_search_fmin_text_output(tmp_path):
    """BASELINE config 1: example.fna unitigs, queries = unitigs (k=4: the file's sequences are 7-8 bp)."""
    fna = tmp_path / "example.fna"
    write_fasta(fna, EXAMPLE)
    assert run("build-fmin", "-o", str(tmp_path / "idx"), "-u", str(fna), "-k", "4").returncode == 0
    r = run("search-fmin", "-i", str(tmp_path / "idx"), "-q", str(fna), "-o", str(tmp_path / "out.txt"))
    assert r.returncode == 0, r.stderr
    text = open(tmp_path / "out.txt").read()
    assert text == "(1,0) (1,1) (1,2) (1,3)\n(2,0) (2,1) (2,2) (2,3) (2,4)\n(0,0) (0,1) (0,2) (0,3) (0,4)\n"
    assert "us/query" in r.stderr and "Total found kmers: 14" in r.stderr
    # larger, gzipped FASTQ queries, stdout output: identical to the oracle's text
    rng = np.random.default_rng(4)
    from tests.util import cut_unitigs, random_genome, sample_reads
    g = random_genome(rng, 30000)
    unitigs = cut_unitigs(rng, g, 31, max_len=900)
    write_fasta(tmp_path / "u.fna", [(str(i), s) for i, s in enumerate(unitigs)])
    reads = sample_reads(rng, g, 300, 150)
    with gzip.open(tmp_path / "r.fq.gz", "wt") as f:
        for i, s in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    assert run("build-fmin", "-o", str(tmp_path / "big"), "-u", str(tmp_path / "u.fna")).returncode == 0
    r = run("search-fmin", "-i", str(tmp_path / "big"), "-q", str(tmp_path / "r.fq.gz"))
    assert r.returncode == 0, r.stderr
    o = OracleIndex.build(unitigs, 31)
    assert r.stdout == "".join(format_pairs(o.search_merged(s)) for s in reads)
    # --strand-counts 1: the reference's per-strand log lines and stats field (search_fmin.hh:66-67, 75-76, 81): n_found of
    # search(read) and of search(rc(read)), each by itself
    from tests.util import rc
    r = run("search-fmin", "-i", str(tmp_path / "big"), "-q", str(tmp_path / "r.fq.gz"), "-o", str(tmp_path / "o2.txt"), "--strand-counts", "1")
    assert r.returncode == 0, r.stderr
    nf = sum(o.search(s)[1] for s in reads); nr = sum(o.search(rc(s))[1] for s in reads)
    assert "Found kmers: %d\n" % nf in r.stderr and "Found kmers reverse : %d\n" % nr in r.stderr
    assert open(tmp_path / "o2.txt").read() == "".join(format_pairs(o.search_merged(s)) for s in reads)
    last = open(str(tmp_path / "big") + ".stats").read().split("31,")[-1]
    assert last.split(",")[0] == str(nf + nr)


def _parse(path, which, block=None):
    args = ["parse-reads", str(path), which] + ([str(block)] if block else [])
    r = run(*args)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_block_parallel_reader_equals_sequential_reader(tmp_path):
    """f-3: uncompressed FASTA/FASTQ is parsed a block at a time by all host threads; the sequential reader (used for gzip) defines
    what a file yields.  Irregular files -- multi-line FASTA, CRLF, blank lines, no final newline, a truncated last record, records
    larger than a block, blocks that end anywhere -- must give the same reads through both."""
    rng = np.random.default_rng(12)

    def seq(n):
        return "".join("ACGT"[x] for x in rng.integers(0, 4, n))

    files = {}
    recs = [seq(int(rng.integers(1, 400))) for _ in range(500)]
    files["plain.fq"] = "".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(recs))
    files["qual_at.fq"] = "".join("@r%d\n%s\n+\n%s\n" % (i, s, "@" * len(s)) for i, s in enumerate(recs))       # qualities that look like headers
    files["crlf.fq"] = files["plain.fq"].replace("\n", "\r\n")
    files["no_final_newline.fq"] = files["plain.fq"][:-1]
    files["truncated.fq"] = files["plain.fq"][:files["plain.fq"].rfind("+")]                                       # last record: header + sequence only
    files["blank_lines.fq"] = files["plain.fq"].replace("@r100\n", "\n\n@r100\n").replace("@r300\n", "\n@r300\n")
    files["single_line.fa"] = "".join(">u%d\n%s\n" % (i, s) for i, s in enumerate(recs))
    files["multi_line.fa"] = "".join(">u%d some text\n%s\n" % (i, "\n".join(s[j:j + 60] for j in range(0, len(s), 60))) for i, s in enumerate(recs))
    files["multi_line_crlf.fa"] = files["multi_line.fa"].replace("\n", "\r\n")
    files["blank.fa"] = files["multi_line.fa"].replace(">u7 ", "\n\n>u7 ") + "\n\n"
    files["no_final_newline.fa"] = files["multi_line.fa"][:-1]
    files["one_big_record.fa"] = ">big\n" + "\n".join(seq(70) for _ in range(300)) + "\n>small\nACGT\n"
    for name, text in files.items():
        p = tmp_path / name
        p.write_bytes(text.encode())
        want = _parse(p, "seq")
        assert want.count("\n") >= 2, name
        for block in (None, 1 << 16, 4096, 1000, 333):
            assert _parse(p, "block", block) == want, (name, block)
    # gzip input goes through the sequential reader whatever is asked for
    with gzip.open(tmp_path / "z.fq.gz", "wt") as f:
        f.write(files["plain.fq"])
    assert _parse(tmp_path / "z.fq.gz", "block") == _parse(tmp_path / "plain.fq", "seq")


def test_build_fmin_honours_sbwt_and_lcs_files(tmp_path):
    """build_fmin.hh:346-383: -i <x.sbwt> gives k and must be the SBWT of the unitigs, --lcs must be its LCS; --sdsl 1 writes the
    reference's seven files beside the container, and an index is loadable from those alone."""
    rng = np.random.default_rng(21)
    from tests.util import cut_unitigs, random_genome
    unitigs = cut_unitigs(rng, random_genome(rng, 6000), 17, max_len=300)
    write_fasta(tmp_path / "u.fna", [(str(i), s) for i, s in enumerate(unitigs)])
    idx = fa.FinimizerIndex.build(unitigs, 17)
    idx.save_sbwt(tmp_path / "u.sbwt")
    r = run("build-fmin", "-o", str(tmp_path / "a"), "-u", str(tmp_path / "u.fna"), "-i", str(tmp_path / "u.sbwt"), "--sdsl", "1")
    assert r.returncode == 0, r.stderr                                   # k = 17 came from the file, not from the default 31
    a = fa.FinimizerIndex().load(tmp_path / "a")
    assert a.k == 17 and a.n_kmers == idx.n_kmers
    for ext in (".O.sdsl", ".FBV.sdsl", ".packed_unitigs.sdsl", ".unitig_endpoints.sdsl", ".Ustart.sdsl", ".LCS.sdsl", ".sbwt", ".finamd"):
        assert os.path.exists(str(tmp_path / "a") + ext)
    os.unlink(str(tmp_path / "a") + ".finamd")
    b = fa.FinimizerIndex().load(tmp_path / "a")                          # the seven files alone
    assert np.array_equal(b.export(fa.X_LCS), idx.export(fa.X_LCS)) and np.array_equal(b.export(fa.X_GOFF), idx.export(fa.X_GOFF))
    r = run("build-fmin", "-o", str(tmp_path / "b"), "-u", str(tmp_path / "u.fna"), "-i", str(tmp_path / "u.sbwt"), "--lcs", str(tmp_path / "a.LCS.sdsl"))
    assert r.returncode == 0 and "LCS_file loaded" in r.stderr
    r = run("build-fmin", "-o", str(tmp_path / "c"), "-u", str(tmp_path / "u.fna"), "-i", str(tmp_path / "u.sbwt"), "-k", "19")
    assert r.returncode == 1 and "does not match" in r.stderr
    other = cut_unitigs(rng, random_genome(rng, 6000), 17, max_len=300)
    write_fasta(tmp_path / "v.fna", [(str(i), s) for i, s in enumerate(other)])
    r = run("build-fmin", "-o", str(tmp_path / "d"), "-u", str(tmp_path / "v.fna"), "-i", str(tmp_path / "u.sbwt"))
    assert r.returncode == 1 and "not the SBWT of these unitigs" in r.stderr
