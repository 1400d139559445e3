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
