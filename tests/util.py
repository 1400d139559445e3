"""Shared helpers for the tests (small pure-numpy synthetic inputs; the big seeded generator is finito_amd.synth)."""
import numpy as np

_RC = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s[::-1].translate(_RC)


def random_genome(rng, n):
    return "".join("ACGT"[x] for x in rng.integers(0, 4, n))


def cut_unitigs(rng, genome, k, max_len=200, flip=True, shuffle=True):
    """Pieces overlapping by k-1 (every genome k-mer in exactly one piece), randomly reverse-complemented."""
    pieces, s, n = [], 0, len(genome)
    while s + k <= n:
        L = int(rng.integers(k, max_len + 1))
        e = min(n, s + L)
        if n - e < 1:
            e = n
        pieces.append(genome[s:e])
        if e == n:
            break
        s = e - (k - 1)
    if flip:
        pieces = [p if rng.random() < 0.5 else rc(p) for p in pieces]
    if shuffle:
        order = rng.permutation(len(pieces))
        pieces = [pieces[i] for i in order]
    return pieces


def sample_reads(rng, genome, n_reads, read_len, err=0.01, random_frac=0.05):
    reads = []
    n = len(genome)
    for _ in range(n_reads):
        if rng.random() < random_frac:
            reads.append(random_genome(rng, read_len))
            continue
        a = int(rng.integers(0, n - read_len + 1))
        r = list(genome[a:a + read_len])
        for i in range(read_len):
            if rng.random() < err:
                r[i] = "ACGT"[(("ACGT".index(r[i])) + int(rng.integers(1, 4))) % 4]
        r = "".join(r)
        reads.append(r if rng.random() < 0.5 else rc(r))
    return reads


def unpack_bits(words, n):
    return np.unpackbits(np.ascontiguousarray(words, dtype=np.uint64).view(np.uint8), bitorder="little")[:n]


def mosaic_read(rng, g, k, max_len):
    """pieces of the genome (either strand) with substitutions at a per-read rate, junk in between, the odd non-ACGT base"""
    out = []
    err = [0.0, 0.0, 0.005, 0.02, 0.1][int(rng.integers(0, 5))]
    L = int(rng.integers(0, max_len))
    while sum(len(x) for x in out) < L:
        t = int(rng.integers(0, 10))
        if t < 7:
            a = int(rng.integers(0, len(g) - 1)); n = int(rng.integers(1, max(2, min(L + 1, 6 * k))))
            piece = g[a:a + n]
            if rng.random() < 0.5:
                piece = rc(piece)
            piece = list(piece)
            for i in range(len(piece)):
                if rng.random() < err:
                    piece[i] = "ACGT"[int(rng.integers(0, 4))]
            out.append("".join(piece))
        elif t < 9:
            out.append(random_genome(rng, int(rng.integers(1, 3 * k))))
        else:
            out.append("N" if rng.random() < 0.7 else "n")
    return "".join(out)[:L]


# ---- the deferred second strand's hard family -------------------------------------------------------------------------
# The reference's forward walk (FinimizerIndex.hh:47-102) compares ONE new base per step with the unitig text behind its anchor; with
# duplicated unitigs the branch dictionary's rank names the wrong copy (common.hh:61-67) and the walk follows a text that does not spell
# the read's k-mers -- into slots where the reverse strand found the true place; the forward pair wins the merge (search_fmin.hh:54-60).
DEFER_KAT = [   # (k, unitigs, read): the judge's three minimised counter-examples of round 3's CPU restatement (faithful pair of the last slot)
    (12, ["AAAGGACCGGACTTG", "GACAAGTCCGGTC", "GACTTTTCCCGG", "GACTTTTCCCGG", "CCCAGAGACGGTTC"], "GGACAAGTCCGGTCC", (3, 2)),
    (21, ["CGTGACCCATTATTATCTCGCAAGTGGTTAAAAACGCGCGGTCGGCCCTAGTTAAGGCGGCTCGGCGTTTAGTC",
          "GTGACCCATTATTATCTCGCAAGTGGTTAAAAACGCGCGGTCGGCCCTAGTTAAGGCGGCTCGGCGTTTAA", "GGTTGCTGATTAAACGCCGAGCCGCCTT"],
     "TCCCATTATTATCTCGCAAGTGGTTAAAAACGCGCGGTCGGCCCTAGTTAAGGCGGCTCGGCGTTTAATC", (1, 53)),
    (31, ["CCACTACAGTCCTCAAACTGAGGACTGCAAGAACCCAATTCA", "GTATTGATACTAGAGATGATTGAGAGTAAAGCCTCTTCGACCC", "GCCTAGCTTCTTGTTTGCACTCTCATTTCAC",
          "GCCTAGCTTCTTGTTTGCACTCTCATTTCAC", "TGTACCATTTCGACAGAGGGTGTGTGAATTGGGTTCTTGCAGTCCTCAGTTTGA"],
     "TCCACTACAGTCCTCAAACTGAGGACTGCAAGAACCCAATTCAC", (2, 12)),
]


def defer_family_case(rng, case, k, n_reads=60):
    """One index set + reads of the family on which a deferred strand's walk matters: identical unitigs, near-duplicates that differ in their
    last (or first) bases, reverse-complement copies, duplicated stretches; reads = mosaics and samples AND their reverse complements, the
    unitigs and their reverse complements themselves, reads that run one or a few bases past a unitig's end (either end, either strand).
    Returns (genome, unitigs, reads)."""
    from tests.test_oracle_lazy import non_disjoint_sets
    fam = case % 5
    if fam == 0:
        g, unitigs = non_disjoint_sets(rng, int(rng.integers(0, 4)), k)
    else:
        g = random_genome(rng, int(rng.integers(6 * k + 50, 40 * k + 400)))
        if fam >= 3:   # duplicated stretches inside the genome
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, max(1, len(g) - 3 * k))); n = int(rng.integers(k + 1, 5 * k)); at = int(rng.integers(0, len(g)))
                g = g[:at] + g[a:a + n] + g[at:]
        unitigs = cut_unitigs(rng, g, k, max_len=int(rng.integers(k + 2, 6 * k + 40)), flip=bool(case & 1))
    unitigs = [u for u in unitigs if len(u) >= k]
    base = list(unitigs)
    for _ in range(int(rng.integers(1, 5))):          # identical copies
        unitigs.insert(int(rng.integers(0, len(unitigs) + 1)), base[int(rng.integers(0, len(base)))])
    for _ in range(int(rng.integers(1, 5))):          # near-duplicates: last / first bases changed, cut or grown
        u = base[int(rng.integers(0, len(base)))]
        t = int(rng.integers(0, 4)); m = int(rng.integers(1, 4))
        if t == 0: v = u[:-m] + random_genome(rng, m)
        elif t == 1: v = random_genome(rng, m) + u[m:]
        elif t == 2: v = u + random_genome(rng, m)
        else: v = u[:max(k, len(u) - m)]
        if len(v) >= k:
            unitigs.insert(int(rng.integers(0, len(unitigs) + 1)), v)
    for _ in range(int(rng.integers(0, 4))):          # reverse-complement copies (whole or part)
        u = base[int(rng.integers(0, len(base)))]
        a = int(rng.integers(0, len(u) - k + 1))
        unitigs.insert(int(rng.integers(0, len(unitigs) + 1)), rc(u[a:a + int(rng.integers(k, len(u) - a + 1))]))
    L = min(len(g), int(rng.integers(k, 6 * k + 60)))
    reads = [mosaic_read(rng, g, k, 5 * k + 100) for _ in range(n_reads // 3)]
    reads += sample_reads(rng, g, n_reads // 3, L, err=float(rng.choice([0.0, 0.0, 0.01, 0.03])), random_frac=0.03)
    for u in unitigs[:40]:                             # the unitigs themselves, and reads that run past their ends
        reads.append(u)
        for _ in range(2):
            t = int(rng.integers(0, 5)); m = int(rng.integers(1, 4)); j = random_genome(rng, m)
            if t == 0: reads.append(u + j)
            elif t == 1: reads.append(j + u)
            elif t == 2: reads.append(j + u + random_genome(rng, 1))
            elif t == 3 and len(u) > k + 2:
                a = int(rng.integers(0, len(u) - k)); reads.append(u[a:] + j)
            else:
                w = list(u); w[int(rng.integers(0, len(w)))] = "ACGTN"[int(rng.integers(0, 5))]; reads.append("".join(w))
    reads += [rc(r) for r in reads]
    return g, unitigs, reads
