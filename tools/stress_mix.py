#!/usr/bin/env python3
"""Step time of the default kernel on read populations the bench mix does not have (50 Mbp index, k=31, 2 M x 150 bp reads, both
strands): substitution rates up to 10 %, reads with indels, chimeric reads (two places glued together), reads of the other
strand only, long reads.  Looks for pathologies (quadratic restarts, fall-backs to the streaming search)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import finito_amd as fa
from finito_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
k = int(os.environ.get("K", "31")); L = int(os.environ.get("L", "150"))
g = synth.genome(50_000_000); u = synth.unitigs(g, k)
idx = fa.FinimizerIndex.build_on_device(u.as_tuple(), k, 0).to_device(0)   # (the device builder takes every k <= 255)
rng = np.random.default_rng(7)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
def windows(n, L):
    st = rng.integers(0, len(g) - 2 * L, n)
    return g[(st[:, None] + np.arange(L)[None, :]).reshape(-1)].reshape(n, L).copy()
def subst(m, rate):
    m = m.copy(); flat = m.reshape(-1)
    e = np.nonzero(rng.random(flat.size) < rate)[0]
    flat[e] = ACGT[(np.searchsorted(ACGT, flat[e]) + rng.integers(1, 4, e.size)) % 4]
    return m
def indels(m, per_read):
    out = m.copy()
    for _ in range(per_read):
        pos = rng.integers(10, L - 10, len(m)); dele = rng.random(len(m)) < 0.5
        for i in np.nonzero(dele)[0][:0]: pass
        # deletion: shift left from pos, pad the end with a random base; insertion: shift right from pos, random base at pos
        idx_ = np.arange(L)[None, :]
        src_del = np.minimum(idx_ + (idx_ >= pos[:, None]), L - 1)
        src_ins = np.maximum(idx_ - (idx_ > pos[:, None]), 0)
        src = np.where(dele[:, None], src_del, src_ins)
        out = np.take_along_axis(out, src, axis=1)
        out[np.arange(len(m)), np.where(dele, L - 1, pos)] = ACGT[rng.integers(0, 4, len(m))]
    return out
def chimera(m):
    other = windows(len(m), L); cut = rng.integers(20, L - 20, len(m))
    return np.where(np.arange(L)[None, :] < cut[:, None], m, other)
base = windows(n, L)
pops = [("1% substitutions (bench mix without random reads)", subst(base, 0.01)), ("5% substitutions", subst(base, 0.05)),
        ("10% substitutions", subst(base, 0.10)), ("1 indel per read + 1% substitutions", subst(indels(base, 1), 0.01)),
        ("3 indels per read", indels(base, 3)), ("chimeric reads", chimera(base)), ("error-free", base),
        ("random reads", ACGT[rng.integers(0, 4, (n, L))])]
offs = np.arange(n + 1, dtype=np.uint64) * L
for kern in (4, 3):
    fa.lib().fin_set_option(b"kernel", kern)
    for name, m in pops:
        b = idx.batch((np.ascontiguousarray(m.reshape(-1)), offs))
        b.run(fa.FIN_MERGED); b.run(fa.FIN_MERGED); b.run(fa.FIN_MERGED)
        _, npos = b.download(want_pairs=False)
        parts, _ = b.step_time_ms(skip_first=1)
        pc = b.pipeline_counts(48) if kern == 4 else None
        print("kernel %d  %-52s step %6.2f ms (pre-pass %.2f, search %.2f)  %.3g k-mers/s  found %.1f%%%s" % (
            kern, name, parts["step"], parts["probe_prepass"], parts["search"], b.n_kmers / parts["step"] * 1e3, 100.0 * npos / b.n_kmers,
            "  stream slots %s list %d" % (pc[6:6 + 4 * 3:4], pc[2]) if pc else ""), flush=True)
        b.close()
fa.lib().fin_set_option(b"kernel", 4)
# ---- repeat-rich genome (round 3): 45 % interspersed / tandem / segmental repeats, copies 1-10 % diverged, both orientations; the index
#      holds every canonical k-mer at its first occurrence (a disjoint string set: short pieces, probe strings that occur all over) ----
if k <= 32:
    idx.close()
    g2 = synth.repeat_genome(50_000_000); u2 = synth.spss(g2, k)
    idx2 = fa.FinimizerIndex.build_on_device(u2.as_tuple(), k, 0).to_device(0)
    r2 = synth.reads(g2, n, read_len=L)
    for kern, ktab in ((4, 1), (4, 0), (3, 1)):
        fa.lib().fin_set_option(b"kernel", kern); fa.lib().fin_set_option(b"kmer_table", ktab)
        b = idx2.batch(r2.as_tuple())
        b.run(fa.FIN_MERGED); b.run(fa.FIN_MERGED); b.run(fa.FIN_MERGED)
        got, npos = b.download()
        parts, _ = b.step_time_ms(skip_first=1)
        bad, checked, _ = synth.check_ground_truth(idx2, u2, r2, got)
        print("kernel %d  %-52s step %6.2f ms (pre-pass %.2f, search %.2f)  %.3g k-mers/s  found %.1f%%  ground truth: %d wrong of %d  overflow reads %d" % (
            kern, "repeat-rich genome, 1%% substitutions, k-mer table %s" % ("on" if ktab else "off"), parts["step"], parts["probe_prepass"], parts["search"],
            b.n_kmers / parts["step"] * 1e3, 100.0 * npos / b.n_kmers, bad, checked, b.overflow_reads()), flush=True)
        b.close()
    fa.lib().fin_set_option(b"kernel", 4); fa.lib().fin_set_option(b"kmer_table", 1)
