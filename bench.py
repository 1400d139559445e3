#!/usr/bin/env python3
"""bench.py -- localized k-mers/s of the search-fmin path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic reads whose ASCII bases are already resident in HBM, results left
in HBM: ingest kernel (2-bit packing of both strands = the reference's get_rc + base decoding), output prefill, probe pre-pass,
search kernel, overflow redo -- everything the reference does inside its timed region search_fmin.hh:46-71 except the text
formatting (reported separately as `end_to_end`).

N = 1 (default): BASELINE.json configs[2] ("chr1": 250 Mbp synthetic unitigs, k=31, t=1, 10 M x 150 bp reads).
N > 1: BASELINE.json configs[3] ("chr1x8": the same index, ONE seeded set of 100 M reads sharded by record across the N GPUs: strong
scaling -- rank r searches records [100 M r / N, 100 M (r+1) / N) in as many device batches as the 2^32-bases-per-batch limit asks for;
a step = one pass over the whole set; `--workload chr1x8 --gpus 1` is that mode's N = 1 point: 100 M reads in four batches on one GPU).
Every rank holds a replica of the index and its own shard of the read records; there is no collective on the data path
(torch.distributed only provides the barrier and the max-over-ranks clock).

`python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh child processes,
before anything touches the GPU); under torch.distributed.run it reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment.

Prints ONE JSON line on rank 0.  `roofline`: algorithmic bytes of the algorithm that runs (the lazy search: probe proofs, walk,
verified restarts), counted by its CPU restatement in oracle/ on a read sample, per launch / device time of the step from HIP
events on the launch stream; the reference algorithm's bytes are kept beside it under `reference_equivalent`.  `cpu_baseline`:
the CPU oracle in reference-shaped mode timed on the host (a port: the reference binary cannot be built here).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (genome bases, k, read_len, reads per GPU, what it is, kind of input)
    "chr1": (250_000_000, 31, 150, 10_000_000, "configs[2]: 250 Mbp synthetic unitigs k=31 t=1, 10 M 150 bp reads per GPU", "iid"),
    # (the one workload whose read count is the WHOLE JOB's: the N ranks share it -- strong scaling)
    "chr1x8": (250_000_000, 31, 150, 100_000_000, "configs[3]: 250 Mbp synthetic unitigs k=31 t=1, ONE set of 100 M 150 bp reads sharded by record across "
                                                  "the N GPUs (rank r: records [100 M r / N, 100 M (r+1) / N))", "iid"),
    "ecoli": (5_000_000, 31, 150, 1_000_000, "configs[1]: 5 Mbp synthetic unitigs k=31 t=1, 1 M 150 bp reads", "iid"),
    "k63": (250_000_000, 63, 250, 10_000_000, "configs[4] at t=1: 250 Mbp synthetic unitigs k=63, 10 M 250 bp reads", "iid"),
    "k127": (250_000_000, 127, 250, 10_000_000, "NOT a BASELINE config: the k63 workload's sizes at k = 127 (beyond the walk kernel's two-word look-ups: the compact k-mer table serves "
                                                "the pre-pass's fast path alone; host-built index)", "iid"),
    # beyond BASELINE.json: inputs that are not iid
    "chr1_repeats": (250_000_000, 31, 150, 10_000_000, "NOT a BASELINE config: configs[2]'s sizes on a repeat-rich genome (45 % interspersed / tandem / segmental "
                     "repeats, copies 1-10 % diverged, both orientations) as a disjoint string set that keeps every canonical k-mer at its first occurrence", "repeats"),
    "k63_repeats": (250_000_000, 63, 250, 10_000_000, "NOT a BASELINE config: the k63 workload's sizes on the repeat-rich genome of chr1_repeats (disjoint string set, every canonical "
                    "63-mer at its first occurrence)", "repeats"),
    "chr1_dups": (250_000_000, 31, 150, 10_000_000, "NOT a BASELINE config: configs[2] with 20 copies (1 % diverged) of a 100 kb block written into the genome -- "
                  "a unitig set that is NOT disjoint (a few duplicated k-mers)", "dups"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: chr1 on 1 GPU, chr1x8 on several")
    ap.add_argument("--genome", type=int, default=0, help="override genome size (bases)")
    ap.add_argument("--reads", type=int, default=0, help="override reads per GPU")
    ap.add_argument("--cpu-sample", type=int, default=60_000, help="reads in the CPU-baseline / parity / byte-count sample")
    ap.add_argument("--check-reads", type=int, default=0, help="reads per rank whose pairs are checked against the ground truth "
                                                               "(default: all on 1 GPU, 2 M per rank on several)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (and the oracle parity / byte-count sample)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive and CLI end-to-end legs")
    ap.add_argument("--no-legs", action="store_true", help="skip the comparison legs (kernel 2 = the reference's work; the index-only configuration)")
    ap.add_argument("--no-text", action="store_true", help="skip the step_with_text passes (profiling: every launch of the run then belongs to a default step)")
    ap.add_argument("--batch-reads", type=int, default=0, help="reads per device batch at most (default: what 2^32 - 2^20 bases hold; tests force several batches with it)")
    ap.add_argument("--parts-max-bases", type=int, default=0, help="> 0: the unitigs as a PARTITIONED index (fin_pindex: parts of at most this many bases, each an ordinary index "
                    "below 2^32 nodes, every read searched in every part) -- unitig sets beyond 2^32 nodes on one GPU; 1 GPU, iid workloads; the line's `value` is the set's")
    ap.add_argument("--no-verify", action="store_true", help="with --parts-max-bases: skip the build-time check that no k-mer lies in two parts")
    ap.add_argument("--kernel", type=int, default=-1)
    return ap.parse_args(argv)


def kernel_source_hash():
    """sha256 over the device code and its launcher: what a PMC traffic measurement is valid for"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "finito_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) or f == "fin_capi.cpp":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def make_inputs(synth, np, kind, gsize, k):
    """(genome, unitigs, skip): the seeded input of a workload.  skip: per k-mer start of the genome, 1 = not checked against the ground
    truth (k-mers with several places in a set that is not disjoint), or None."""
    if kind == "repeats":
        g = synth.repeat_genome(gsize)
        return g, synth.spss(g, k), None
    g = synth.genome(gsize)
    skip = None
    if kind == "dups":
        rng = np.random.default_rng(12345)
        L = min(100_000, gsize // 50); src = gsize // 10
        block = g[src:src + L].copy()
        skip = np.zeros(gsize, dtype=np.uint8); skip[max(0, src - k):src + L] = 1
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8); code = np.zeros(256, dtype=np.uint8); code[acgt] = np.arange(4, dtype=np.uint8)
        for i in range(20):
            c = block.copy()
            m = rng.random(L) < 0.01
            c[m] = acgt[(code[c[m]] + rng.integers(1, 4, int(m.sum()))) % 4]
            dst = int(gsize * (0.2 + 0.035 * i))
            g[dst:dst + L] = c; skip[max(0, dst - k):dst + L] = 1
    return g, synth.unitigs(g, k), skip


def spawn_ranks(args):
    """--gpus N without a launcher: start N fresh rank processes of this script (nothing here has touched the GPU or imported
    torch), forward rank 0's JSON line, fail if any rank fails."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 600
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    if rcs[0] != 0:          # rank 0 died: the others may be stuck in a barrier
        for p in procs[1:]:
            if p.poll() is None:
                p.kill()
    line = None
    for ln in (out0 or b"").decode(errors="replace").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(rcs) or line is None:
        log("rank exit codes:", rcs)
        sys.stdout.write((out0 or b"").decode(errors="replace"))
        raise SystemExit(1)
    print(line, flush=True)


def dry_run(args):
    """FINITO_BENCH_DRYRUN=1: the rank plumbing alone (rendezvous, barrier, max-over-ranks clock, rank 0's line) on gloo, no GPU --
    what tests/test_dist.py runs on a CPU-only machine"""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    el = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        wname = args.workload or ("chr1" if world == 1 else "chr1x8")
        print(json.dumps({"dryrun": True, "n_gpus": world, "max_over_ranks": float(el.item()), "config": {"workload": "%s = BASELINE.json %s" % (wname, WORKLOADS[wname][4])}}), flush=True)


def run_rank(args):
    if os.environ.get("FINITO_BENCH_DRYRUN"):
        return dry_run(args)
    import numpy as np
    import torch

    import finito_amd as fa
    from finito_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("note: WORLD_SIZE=%d, --gpus=%d; using WORLD_SIZE" % (world, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal switch: FINITO_BENCH_BACKEND=gloo with FINITO_BENCH_DEVICE=0 runs several ranks on one GPU (tests the rank
        # plumbing on a 1-GPU box); the real multi-GPU run is one rank per GPU over RCCL
        backend = os.environ.get("FINITO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if "FINITO_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["FINITO_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if args.kernel >= 0:
        assert fa.lib().fin_set_option(b"kernel", args.kernel) == 0
    for kv in filter(None, os.environ.get("FINITO_OPTS", "").split(",")):   # experiments: any process-wide option, "name=value,..."
        name, val = kv.split("=")
        assert fa.lib().fin_set_option(name.encode(), int(val)) == 0, kv
    if "FINITO_JTAB_T" in os.environ:   # experiments: depth of the jump table (default: by index size)
        assert fa.lib().fin_set_option(b"jtab_t", int(os.environ["FINITO_JTAB_T"])) == 0
    if "FINITO_FILT_F" in os.environ:   # experiments: depth of the pre-pass's absence filter (default: by index size; 0 = none)
        assert fa.lib().fin_set_option(b"filt_f", int(os.environ["FINITO_FILT_F"])) == 0
    if "FINITO_PTAB_T" in os.environ:   # experiments: depth of the prefix table (default: by index size)
        assert fa.lib().fin_set_option(b"ptab_t", int(os.environ["FINITO_PTAB_T"])) == 0

    wname = args.workload or ("chr1" if world == 1 else "chr1x8")
    gsize, k, read_len, n_reads, desc, kind = WORKLOADS[wname]
    if args.genome:
        gsize = args.genome
    if args.reads:
        n_reads = args.reads

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.parts_max_bases:
        return run_partitioned(args, fa, synth, np, torch, wname, gsize, k, read_len, n_reads, desc, kind, local_rank, world)

    # ---- inputs: index built once by rank 0, replicated through a container file in /dev/shm; reads sharded by record ----
    t0 = time.time()
    # (all ranks of a job share MASTER_PORT, whoever launched them)
    prefix = "/dev/shm/finito_bench_%s_%d_p%s" % (wname, gsize, os.environ.get("MASTER_PORT", str(os.getpid())))
    shared = ("g", "ubases", "uoffsets", "ugstart", "uglen", "urc")
    skip = None
    if rank == 0:
        g, u, skip = make_inputs(synth, np, kind, gsize, k)
        log("inputs (%s) generated in %.1f s: %d unitigs, %d bases" % (kind, time.time() - t0, len(u), int(u.offsets[-1])))
        t0 = time.time()
        build_how = "host"
        if not os.environ.get("FINITO_BENCH_HOST_BUILD"):   # the device builder: the same index bit for bit (tests/test_build_gpu.py), about 30 x sooner
            idx = fa.FinimizerIndex.build_on_device(u.as_tuple(), k, local_rank)
            build_how = "device (%s ms)" % {kk: round(vv) for kk, vv in idx.build_phase_ms.items()}
        else:
            idx = fa.FinimizerIndex.build(u.as_tuple(), k)
        build_s = time.time() - t0
        log("index built on the %s in %.2f s: %d nodes, %d k-mers, %d unitigs, %d finimizers, %.1f MB in HBM"
            % (build_how, build_s, idx.n_nodes, idx.n_kmers, idx.n_unitigs, idx.n_finimizers, idx.size_in_bytes() / 1e6))
        if world > 1:   # the other ranks take genome, unitigs and index from /dev/shm instead of making them again (16 host cores for 8 ranks)
            idx.serialize(prefix)
            for nm, arr in zip(shared, (g, u.bases, u.offsets, u.gstart, u.glen, u.rc)):
                np.save("%s.%s.npy" % (prefix, nm), arr)
    if dist is not None:
        dist.barrier()
        if rank != 0:
            idx = fa.FinimizerIndex().load(prefix)
            a = {nm: np.load("%s.%s.npy" % (prefix, nm), mmap_mode="r") for nm in shared}
            g = a["g"]
            u = synth.Unitigs(a["ubases"], a["uoffsets"], a["ugstart"], a["uglen"], a["urc"], k)
        dist.barrier()
        if rank == 0:
            for f in [prefix + ".finamd"] + ["%s.%s.npy" % (prefix, nm) for nm in shared]:
                try:
                    os.unlink(f)
                except OSError:
                    pass
    idx.to_device(local_rank)
    # STRONG scaling (configs[3], any N > 1): the job is ONE read set of n_reads records; rank r holds records [ceil(n_reads r / N), ceil(n_reads (r+1) / N))
    # -- what finito_amd.dist.shard_bounds gives for reads of one length -- in device batches of at most 2^32 - 2^20 bases each (a batch
    # addresses bases and k-mers with 32 bits), all resident at once; a record's content depends on its number only (fin_synth_reads_at).
    # N = 1 without --workload chr1x8: configs[2], one batch.
    strong = wname == "chr1x8"
    total_reads = n_reads
    r_lo, r_hi = (-(-total_reads * rank // world), -(-total_reads * (rank + 1) // world)) if strong else (0, n_reads)   # (ceil: the first record at or behind the r-th N-th of the bases)
    if not strong and world > 1:   # (another workload on several GPUs: every rank its own n_reads records -- weak scaling)
        r_lo, r_hi = rank * n_reads, (rank + 1) * n_reads
        total_reads = world * n_reads
    n_reads = r_hi - r_lo
    cap = max(1, ((1 << 32) - (1 << 20)) // max(read_len, 1))
    if args.batch_reads:
        cap = min(cap, args.batch_reads)
    n_batches = max(1, -(-n_reads // cap))
    per = -(-n_reads // n_batches)
    t1 = time.time()
    batches, reads = [], None
    for bi in range(n_batches):
        b_lo, b_hi = r_lo + bi * per, min(r_hi, r_lo + (bi + 1) * per)
        rd = synth.reads(g, b_hi - b_lo, read_len=read_len, seed=synth.SEED_READS, first=b_lo)
        batches.append(idx.batch(rd.as_tuple()))
        if bi == 0:
            reads = rd   # (the first batch's reads stay on the host: ground truth, oracle sample, text legs)
        del rd
    batch = batches[0]
    n_kmers = sum(b.n_kmers for b in batches)
    log("rank %d: records [%d, %d) of %d = %d reads (%d k-mers) resident in HBM in %d batch(es), generated+uploaded in %.1f s" % (rank, r_lo, r_hi, total_reads, n_reads, n_kmers, n_batches, time.time() - t1))
    stream = torch.cuda.current_stream().cuda_stream

    # ---- timed region: a step = one pass over the rank's whole shard ----
    for _ in range(args.warmup):
        for b in batches:
            b.run(fa.FIN_MERGED, stream)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        for b in batches:
            b.run(fa.FIN_MERGED, stream)
    barrier()
    elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if (dist is None or dist.get_backend() == "nccl") else "cpu")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # HIP events on the launch stream, timed launches only; several batches: a step's parts are the sums over its batches
    parts, parts_n = {}, 0
    for b in batches:
        p_b, parts_n = b.step_time_ms(skip_first=args.warmup)
        for kk, vv in p_b.items():
            parts[kk] = parts.get(kk, 0.0) + vv
    kern_ms = parts["step"]
    n_reads_rank, n_kmers_rank = n_reads, n_kmers
    n_reads, n_kmers_b0 = len(reads), batch.n_kmers   # (everything below the timed region looks at the first batch)

    # ---- several ranks: what the host side costs when all GPUs of the node are fed at once ("scaling limit is
    #      host-side: input parse and D2H of 8 B/k-mer over PCIe").  Every rank pushes the same number of reads from page-locked host buffers
    #      through fin_search_batch (H2D + step + D2H, pipelined) at the same time; max over ranks; never `value` ----
    pcie_all = None
    if world > 1 and not args.no_e2e:
        ns_p = min(n_reads, 2_000_000)
        subp = reads.subset(0, ns_p)
        nk_p = ns_p * max(0, read_len - k + 1)
        pin_b = fa.PinnedArray((ns_p * read_len,), np.uint8)
        pin_o = fa.PinnedArray((max(nk_p, 1), 2), np.int32)
        try:
            pin_b.array[:] = subp.bases
            idx.search_reads((pin_b.array, subp.offsets), fa.FIN_MERGED, out=pin_o.array)   # warm-up: buffers, streams
            barrier()
            tp = time.perf_counter()
            idx.search_reads((pin_b.array, subp.offsets), fa.FIN_MERGED, out=pin_o.array)
            barrier()
            dtp = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(dtp, op=dist.ReduceOp.MAX)
            pcie_all = {"reads_per_rank": ns_p, "kmers_all_ranks": world * nk_p, "seconds_max_over_ranks": float(dtp.item()),
                        "pcie_inclusive_kmers_per_s_all_ranks": world * nk_p / float(dtp.item()),
                        "note": "every rank at once: fin_search_batch from page-locked host buffers, pairs back in host memory (1.25 B in + 8 B out per k-mer over each GPU's PCIe link, %d host threads per rank); the `value` above keeps its inputs and outputs in HBM" % fa.host_threads()}
        finally:
            pin_b.close(); pin_o.close()

    # ---- the reference's own timed region (search_fmin.hh:46-71) ends with the output TEXT: one more measurement, outside `value`, of
    #      step + text formatting on the device (fin_text.hip), events on the same stream ----
    with_text = None
    if rank == 0 and read_len >= k and not args.no_text and n_batches == 1:
        tstream = torch.cuda.current_stream()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        legs_t = {}
        for mode in (2, 1, 0):   # text only (the reference keeps nothing but the text) | pairs + the fast path's records | pairs, text from the pairs (round 3)
            batch.text_mode(mode)
            ms_step, ms_text, tbytes = [], [], 0
            for i in range(3):
                e0.record(tstream); batch.run(fa.FIN_MERGED, stream); e1.record(tstream)
                tbytes = batch.format_text(); e2.record(tstream)
                torch.cuda.synchronize()
                if i:   # (the first pass allocates the text buffers)
                    ms_step.append(e0.elapsed_time(e1)); ms_text.append(e1.elapsed_time(e2))
            legs_t[mode] = {"ms_step": sum(ms_step) / len(ms_step), "ms_text": sum(ms_text) / len(ms_text), "text_bytes": tbytes}
        assert legs_t[0]["text_bytes"] == legs_t[1]["text_bytes"] == legs_t[2]["text_bytes"], legs_t
        with_text = dict(legs_t[2], text_mode=2,
                         pairs_kept={"ms_step": legs_t[1]["ms_step"], "ms_text": legs_t[1]["ms_text"], "text_mode": 1},
                         text_from_pairs={"ms_step": legs_t[0]["ms_step"], "ms_text": legs_t[0]["ms_text"], "text_mode": 0})
        # (the last pass ran in mode 0: the batch's pairs are complete for the checks below)

    # ---- checks on the results of the timed launches (rank 0 carries the oracle leg) ----
    n_check = args.check_reads or (n_reads if world == 1 and n_batches == 1 else min(n_reads, 2_000_000))
    n_check = min(n_check, n_reads)
    nk_read = max(0, read_len - k + 1)
    if n_check == n_reads:
        pairs, n_pos = batch.download()
    else:
        pairs = batch.download_range(0, n_check * nk_read)
        _, n_pos = batch.download(want_pairs=False)
    chk = reads if n_check == n_reads else reads.subset(0, n_check)
    if rank == 0 or kind == "iid":
        bad, checked, first_bad = synth.check_ground_truth(idx, u, chk, pairs, skip=skip)
        if bad:
            raise SystemExit("rank %d: %d of %d error-free k-mers localized wrongly (first bad read %d)" % (rank, bad, checked, first_bad))
        log("rank %d: ground truth ok on %d error-free k-mers of the first %d reads; %d of %d k-mers found (first batch)" % (rank, checked, n_check, n_pos, n_kmers_b0))
    else:
        # (the duplicated / repeat-rich generators' first-occurrence maps live on rank 0 only -- the other ranks' unitigs come from
        #  /dev/shm without them, and the plain check would call the duplicated k-mers wrong: rank 0 checks its whole shard)
        checked = 0
        log("rank %d: %d of %d k-mers found (ground truth of the '%s' generator is checked on rank 0)" % (rank, n_pos, n_kmers_b0, kind))

    out = None
    if rank == 0:
        kname = "v%d" % (args.kernel if args.kernel >= 0 else 4)
        # the whole job's k-mers per second: every rank's shard (reads of one length: the shards' k-mers add up to the set's)
        job_kmers = total_reads * nk_read
        value = job_kmers * args.steps / elapsed
        ptd = idx.prefix_table_depth(local_rank)
        out = {
            "metric": "localized k-mers/s (k=%d)" % k, "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s = BASELINE.json %s" % (wname, desc), "k": k, "t": 1, "index_bases": gsize,
                       "index_nodes": idx.n_nodes, "index_bytes_hbm": idx.size_in_bytes(), "index_build": build_how, "index_build_s": round(build_s, 3), "index_disjoint": idx.is_disjoint(), "unsafe_places": idx.unsafe_places(local_rank), "rc_pairs": idx.rc_pairs(local_rank), "second_strand_deferred": idx.defers_second_strand(local_rank), "anchor_table_build_ms": idx.anchor_build_ms(local_rank),
                       "prefix_table_bytes_hbm": 8 * 4 ** ptd if ptd > 0 else 0,
                       "jump_table_bytes_hbm": 8 * 4 ** idx.jump_table_depth(local_rank) if idx.jump_table_depth(local_rank) > 0 else 0,
                       "seed_table_bytes_hbm": idx.seed_table_bytes(local_rank), "kmer_table_bytes_hbm": idx.kmer_table_bytes(local_rank), "reads_per_gpu": n_reads_rank, "reads_total": total_reads, "batches_per_gpu": n_batches,
                       "read_len": read_len, "kmers_per_gpu_per_step": n_kmers_rank, "kmers_per_step": job_kmers, "strands": "both, merged",
                       "step": "ASCII reads resident in HBM -> 2-bit pack of both strands -> probe pre-pass -> search pipeline (writes every output "
                               "slot once; without a seed table: (-1,-1) prefill first, pairs overwrite) -> overflow redo; pairs left in HBM",
                       "derived_tables_bytes_hbm": None, "derived_tables_bytes_per_indexed_base": None,
                       "parallelism": "reads sharded by record, index replicated, no collective",
                       "kernel": kname, "ground_truth_checked_kmers": checked, "overflow_reads": sum(b.overflow_reads() for b in batches)},
        }
        cfg = out["config"]   # what the upload builds beside the index itself: every table, filter and bitmap of the replica
        cfg["string_filter_bytes_hbm"] = idx.string_filter_bytes(local_rank)
        cfg["derived_tables_bytes_hbm"] = idx.replica_table_bytes(local_rank)
        cfg["derived_tables_bytes_per_indexed_base"] = round(cfg["derived_tables_bytes_hbm"] / max(1, gsize), 1)
        cfg["replica_bits_per_kmer"] = round(8.0 * (cfg["index_bytes_hbm"] + cfg["derived_tables_bytes_hbm"]) / max(1, idx.n_kmers), 1)
        info = batch.run_info()   # what the timed runs decided (the per-run decision, not a guess from the replica)
        cfg["second_strand_deferred"] = info["deferred"]; cfg["fast_path"] = info["fast_path"]; cfg["lean_tables"] = idx.lean_tables(local_rank)
        pre_kernel = "fin_fast_prepass_kernel" if info["fast_path"] else "fin_pair_prepass_kernel" if info["deferred"] else "fin_probe_kernel"
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "one step = fin_pack_reads_kernel + " + (pre_kernel + " + fin_route_kernel + rounds x (fin_stream_kernel + fin_walk_kernel) + fin_search_v3_kernel on the rest" if kname == "v4" else "prefill + fin_probe_kernel + fin_search_%s_kernel" % kname),
                "kernel_ms": kern_ms, "kernel_ms_parts": parts, "timed_launches": parts_n}
        if kname == "v4":
            pc = batch.pipeline_counts(48)
            roof["pipeline_queue_slots"] = {"kernel3_list": pc[2], "stream_rounds": pc[6:6 + 4 * 9:4], "walk_rounds": pc[7:7 + 4 * 8:4]}
            roof["reads_finished_by_the_fast_path"] = pc[4 * 8 + 9]; roof["deferred_strands_searched"] = pc[4 * 8 + 8]
        if not args.no_cpu and world == 1:
            from oracle.oracle import Counters, LazyCounters, OracleIndex
            ns = min(args.cpu_sample, n_reads)
            t2 = time.time()
            oracle = OracleIndex.from_components(k, idx.components())
            log("oracle assembled from exported components in %.1f s" % (time.time() - t2))
            # (the sample is a stride over the whole batch, not its head)
            sidx = (np.arange(ns, dtype=np.int64) * n_reads) // ns
            sample = reads.take(sidx)
            ctr = Counters()
            exp, _, _ = oracle.search_batch(sample.as_tuple(), counters=ctr, n_threads=fa.host_threads())
            got_s = pairs.reshape(-1, nk_read, 2)[sidx[sidx < n_check]].reshape(-1, 2) if nk_read else pairs[:0]
            if not np.array_equal(got_s.astype(np.int64), exp[: got_s.shape[0]]):
                raise SystemExit("HIP output differs from the CPU oracle on the %d-read sample" % ns)
            # the algorithm the kernels run, restated on the CPU: same pairs, and its own byte count
            lctr = LazyCounters()
            lazy_kw = dict(ptab_t=ptd, jump_t=idx.jump_table_depth(local_rank), count_safe_checks=idx.unsafe_places(local_rank) > 0, rc_pairs=idx.rc_pairs(local_rank) > 0, filt_f=idx.filter_depth(local_rank), n_threads=fa.host_threads())
            lexp = oracle.search_batch_lazy(sample.as_tuple(), disjoint=kname in ("v3", "v4"), seeds=kname == "v4", kmer_table=kname == "v4" and idx.kmer_table_bytes(local_rank) > 0, defer=kname == "v4" and info["deferred"], fast=kname == "v4" and info["fast_path"], lean=kname == "v4" and idx.lean_tables(local_rank), counters=lctr, **lazy_kw)
            if not np.array_equal(lexp, exp):
                raise SystemExit("oracle: the lazy restatement differs from the faithful search on the %d-read sample" % ns)
            # timed leg: single thread, search + merge + text formatting exactly as the reference's timed region
            _, secs, _ = oracle.search_batch(sample.as_tuple(), want_pairs=False, format_text=True, n_threads=1)
            _, secs_nofmt, _ = oracle.search_batch(sample.as_tuple(), want_pairs=False, format_text=False, n_threads=1)
            ncores = fa.host_threads()
            _, secs_all, _ = oracle.search_batch(sample.as_tuple(), want_pairs=False, format_text=True, n_threads=ncores)
            sk = int(ctr.kmers)
            out["cpu_baseline"] = {"value": sk / secs, "unit": "k-mers/s", "cores": 1, "kind": "port",
                                   "sample": "%d reads spread evenly over rank 0's batch (%d k-mers), oracle in reference-shaped mode: two "
                                             "searches per read + merge + text formatting (search_fmin.hh:46-71)" % (ns, sk),
                                   "search_only_value": sk / secs_nofmt, "all_cores_value": sk / secs_all, "all_cores": ncores}
            lazy_bpk = lctr.algorithmic_bytes() / sk
            ref_bpk = ctr.algorithmic_bytes() / sk
            # kernels 3 and 4 run the lazy algorithm: its bytes; kernels 2 and 0 do all the reference's work: the reference's bytes
            bpk = lazy_bpk if kname in ("v3", "v4") else ref_bpk
            roof["achieved"] = bpk * n_kmers / (kern_ms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            roof["algorithmic_bytes_per_kmer"] = bpk
            roof["algorithmic_bytes_model"] = LazyCounters.MODEL if kname in ("v3", "v4") else "SURVEY.md 8(d): 64*(rank_lines + lcs_lines + 4*anchors + walked/256) + base_strands + 8*kmers on the faithful oracle's counters"
            roof["algorithmic_bytes_parts_per_kmer"] = {kk: vv / sk for kk, vv in lctr.parts().items()}
            if kname in ("v3", "v4") and parts:
                # the same bytes and the same HIP-event times, split by the part of the step that moves them
                roof["stages"] = {st: {"ms": parts.get(st), "algorithmic_bytes_per_kmer": by / sk,
                                       "achieved": by / sk * n_kmers / (parts[st] * 1e-3) / 1e9, "frac": by / sk * n_kmers / (parts[st] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                  for st, by in lctr.stage_bytes(output_in_search=kname == "v4" and info["no_prefill"]).items() if parts.get(st)}
            roof["lazy_counters_per_kmer"] ={kk: vv / sk for kk, vv in lctr.as_dict().items()}
            roof["reference_equivalent"] = {
                "note": "bytes of the REFERENCE algorithm (SURVEY.md 8(d) formula on the faithful oracle's counters) / the same time: "
                        "not a roofline fraction -- the kernels skip most of that work (CHANGELOG.md 4.6)",
                "algorithmic_bytes_per_kmer": ref_bpk, "gbps": ref_bpk * n_kmers / (kern_ms * 1e-3) / 1e9,
                "oracle_counters_per_base_strand": {kk: vv / ctr.base_strands for kk, vv in ctr.as_dict().items()
                                                    if kk in ("extends", "rank_lines", "drops", "lcs_lines", "lcs_entries", "anchors", "walked")}}
            out["config"]["parity"] = "bit-exact vs CPU oracle (faithful and lazy restatements) on %d reads spread evenly over the batch" % ns
            roof["algorithmic_bytes_sample"] = "counters of the lazy restatement on %d reads spread evenly over the batch (%.2f %% of it), scaled to the batch" % (ns, 100.0 * ns / n_reads)
            # like for like: `value` stops at pairs in HBM -> against the port's search-only rate; the reference's own region includes the
            # text -> step_with_text against the port's search+text rate
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["search_only_value"]
            out["speedup_note"] = "value / cpu_baseline.search_only_value (both stop at pairs); with the output text on both sides: step_with_text.speedup_vs_cpu_1core"
            if with_text:
                with_text["speedup_vs_cpu_1core"] = n_kmers / ((with_text["ms_step"] + with_text["ms_text"]) * 1e-3) / out["cpu_baseline"]["value"]
            if not args.no_legs and kname == "v4":
                out["comparison_legs"] = comparison_legs(fa, np, idx, reads, pairs, nk_read, k, local_rank, stream, oracle, sample, exp, ctr, lazy_kw, LazyCounters)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get("%s:%d:%s" % (wname, n_reads, kname))
                if ent:
                    if ent.get("kernel_src_sha16") == kernel_source_hash():
                        roof["traffic"] = ent["hbm_bytes_per_step"]
                        roof["traffic_source"] = ent.get("source")
                        roof["hbm_measured_gbps"] = ent["hbm_bytes_per_step"] / (kern_ms * 1e-3) / 1e9
                        roof["hbm_measured_frac"] = roof["hbm_measured_gbps"] / HBM_PEAK_GBS
                        # per stage: the counters' bytes of its kernels beside its algorithmic bytes -- the wasted-traffic ratio kernel by kernel
                        stage_of = lambda kn: ("ingest_prefill" if kn.startswith("fin_pack") or "fillBuffer" in kn else
                                               "probe_prepass" if kn.startswith(("fin_fast", "fin_pair_prepass", "fin_probe")) else "search")
                        by_stage = {}
                        for kn, bts in (ent.get("parts") or {}).items():
                            by_stage[stage_of(kn)] = by_stage.get(stage_of(kn), 0) + bts
                        roof["traffic_kernels"] = ent.get("parts")
                        for st, bts in by_stage.items():
                            if st in roof.get("stages", {}):
                                sg = roof["stages"][st]
                                sg["traffic"] = bts
                                sg["traffic_over_algorithmic"] = bts / max(1.0, sg["algorithmic_bytes_per_kmer"] * n_kmers)
                                sg["hbm_measured_frac"] = bts / (sg["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                    else:
                        roof["traffic_note"] = "profiles/traffic.json was measured on other kernel sources (%s): not reported" % ent.get("kernel_src_sha16")
            except Exception as e:   # a damaged traffic file must not cost the bench line
                roof["traffic_note"] = "profiles/traffic.json unreadable: %s" % e
        if with_text:
            tot = with_text["ms_step"] + with_text["ms_text"]
            with_text.update({"kmers_per_s": n_kmers / (tot * 1e-3), "ms": tot,
                              "note": "one step + fin_batch_format_text (the reference's timed region search_fmin.hh:46-71: both searches, merge, text) on one "
                                      "GPU in text-only mode (fin_batch_text_mode 2: the pairs of the reads the fast path finishes are never written, their text "
                                      "is made from the path's 32-byte records); pairs_kept: mode 1; text_from_pairs: mode 0 (round 3's formatter); HIP events on "
                                      "the launch stream, mean of 2 passes; text left in HBM; never `value`"})
            tb = 16.0 + with_text["text_bytes"] / n_kmers   # (priced as round 3 did: pairs read twice (lengths, write) + the text written -- the record path moves less)
            roof.setdefault("stages", {})["text"] = {"ms": with_text["ms_text"], "algorithmic_bytes_per_kmer": tb,
                                                     "achieved": tb * n_kmers / (with_text["ms_text"] * 1e-3) / 1e9,
                                                     "frac": tb * n_kmers / (with_text["ms_text"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out["step_with_text"] = with_text
        out["roofline"] = roof
        if pcie_all:
            out["end_to_end_all_ranks"] = pcie_all
        if not args.no_e2e and world == 1:
            try:
                out["end_to_end"] = end_to_end(fa, idx, reads, k, read_len, min(n_reads, 4_000_000), local_rank)
            except Exception as e:
                out["end_to_end"] = {"error": str(e)}
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def run_partitioned(args, fa, synth, np, torch, wname, gsize, k, read_len, n_reads, desc, kind, device, world):
    """--parts-max-bases: the workload's unitigs as a partitioned index (include/finito_amd.h: fin_pindex_*) on ONE GPU -- the way past 2^32 nodes.  A step =
    every part's step over the same resident reads + a merge pass each; the ground truth is checked on every error-free k-mer as in the default line."""
    if world != 1 or kind != "iid":
        raise SystemExit("--parts-max-bases: one GPU, an iid workload (the set must be disjoint)")
    t0 = time.time()
    g, u, _ = make_inputs(synth, np, kind, gsize, k)
    log("inputs (%s) generated in %.1f s: %d unitigs, %d bases" % (kind, time.time() - t0, len(u), int(u.offsets[-1])))
    t0 = time.time()
    pidx = fa.PartitionedIndex(u.as_tuple(), k, device, max_part_bases=args.parts_max_bases, verify=not args.no_verify)
    build_s = time.time() - t0
    nodes = pidx.part_nodes()
    log("partitioned index built in %.1f s (of which the disjointness check %.1f s): %d parts of %s nodes = %d nodes, %d k-mers, %d unitigs; index %.1f GB + tables %.1f GB in HBM"
        % (build_s, pidx.verify_seconds, pidx.n_parts, nodes, pidx.n_nodes, pidx.n_kmers, pidx.n_unitigs, pidx.size_in_bytes() / 1e9, pidx.replica_table_bytes() / 1e9))
    rd = synth.reads(g, n_reads, read_len=read_len, seed=synth.SEED_READS)
    b = pidx.batch(rd.as_tuple())
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        b.run(stream)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        b.run(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    ms_dev, _ = b.step_time_ms(skip_first=args.warmup)
    pairs, n_pos = b.download()
    bad, checked, first_bad = synth.check_ground_truth(pidx, u, rd, pairs)
    if bad:
        raise SystemExit("%d of %d error-free k-mers localized wrongly by the partitioned index (first bad read %d)" % (bad, checked, first_bad))
    log("ground truth ok on %d error-free k-mers of %d reads; %d of %d k-mers found" % (checked, n_reads, n_pos, b.n_kmers))
    value = b.n_kmers * args.steps / elapsed
    print(json.dumps({
        "metric": "localized k-mers/s (k=%d)" % k, "value": value, "unit": "k-mers/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s = BASELINE.json %s, its unitigs as a PARTITIONED index" % (wname, desc), "k": k, "read_len": read_len, "reads_per_gpu": n_reads,
                   "index_bases": pidx.total_len, "index_nodes": pidx.n_nodes, "index_nodes_beyond_2_32": pidx.n_nodes >= (1 << 32), "index_kmers": pidx.n_kmers,
                   "index_unitigs": pidx.n_unitigs, "parts": pidx.n_parts, "part_nodes": nodes, "parts_max_bases": args.parts_max_bases,
                   "index_build_s": round(build_s, 2), "disjointness_check_s": round(pidx.verify_seconds, 2), "shared_kmers": pidx.shared_kmers,
                   "index_bytes_hbm": pidx.size_in_bytes(), "derived_tables_bytes_hbm": pidx.replica_table_bytes(),
                   "kmers_found": n_pos, "ground_truth_kmers_checked": checked, "device_ms_per_step": ms_dev},
        "note": "every read is searched in every part (the parts' steps one after the other on one stream) and each part's pairs are merged into the set's "
                "result with the unitigs renumbered for the whole set (fin_set_merge_kernel); exact for a disjoint spectrum-preserving string set, which the build checked"}), flush=True)


def comparison_legs(fa, np, idx, reads, pairs, nk_read, k, device, stream, oracle, sample, exp, ctr, lazy_kw, LazyCounters):
    """VERDICT r3 #4 -- two short legs beside the headline, never `value`, each checked against the headline's pairs:
    kernel_2     the kernel that streams every base of both strands, i.e. does ALL the reference's work (SURVEY 8(d)'s formula prices exactly
                 that: the one line whose fraction by the reference's bytes is a roofline fraction);
    index_only   kernel 4 on a replica WITHOUT anchor table, k-mer table and string filter (options seed_anchors 0, kmer_table 0): the
                 configuration that lives closest to the reference's memory class -- the finimizer dictionaries answer every anchor."""
    import tempfile, shutil
    legs = {}
    n_reads = len(reads)
    sk = int(ctr.kmers)

    def timed(ix, sub, n_rep=3):
        b = ix.batch(sub.as_tuple())
        try:
            for _ in range(1 + n_rep):
                b.run(fa.FIN_MERGED, stream)
            got, _ = b.download()
            parts, n = b.step_time_ms(skip_first=1)
            return got, parts, b.n_kmers, b.run_info()
        finally:
            b.close()

    # kernel 2 on the first million reads
    nl = min(n_reads, 1_000_000)
    sub = reads.subset(0, nl)
    idx.set_option("kernel", 2)
    try:
        got, parts, nkm, info = timed(idx, sub)
    finally:
        idx.set_option("kernel", None)
    if not np.array_equal(got, pairs[: got.shape[0]]):
        raise SystemExit("kernel 2 differs from the default kernel on the first %d reads" % nl)
    ref_bpk = ctr.algorithmic_bytes() / sk
    legs["kernel_2"] = {"what": "fin_search_v2_kernel: every base of both strands through the streaming search (rarest_fmin_streaming_search, common.hh:78-186) -- the reference's work",
                        "reads": nl, "ms_per_step": parts["step"], "kmers_per_s": nkm / (parts["step"] * 1e-3), "algorithmic_bytes_per_kmer": ref_bpk,
                        "algorithmic_bytes_model": "SURVEY.md 8(d) on the faithful oracle's counters", "achieved": ref_bpk * nkm / (parts["step"] * 1e-3) / 1e9,
                        "frac": ref_bpk * nkm / (parts["step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "pairs": "equal to the default kernel's"}
    # the index-only configuration on the first two million reads
    tmp = tempfile.mkdtemp(prefix="finito_leg_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        idx.serialize(os.path.join(tmp, "idx"))
        ix2 = fa.FinimizerIndex().load(os.path.join(tmp, "idx"))
        ix2.set_option("seed_anchors", 0); ix2.set_option("kmer_table", 0)
        ix2.to_device(device)
        nl2 = min(n_reads, 2_000_000)
        got, parts, nkm, info = timed(ix2, reads.subset(0, nl2))
        if not np.array_equal(got, pairs[: got.shape[0]]):
            raise SystemExit("the index-only configuration differs from the default one on the first %d reads" % nl2)
        lc = LazyCounters()
        ns2 = min(len(sample), 20_000)
        s2 = sample.subset(0, ns2)
        kw2 = dict(lazy_kw, ptab_t=ix2.prefix_table_depth(device), jump_t=ix2.jump_table_depth(device), filt_f=ix2.filter_depth(device))   # (this replica's own tables)
        le = oracle.search_batch_lazy(s2.as_tuple(), disjoint=True, seeds=False, kmer_table=False, defer=False, fast=False, counters=lc, **kw2)
        if not np.array_equal(le, exp[: le.shape[0]]):
            raise SystemExit("oracle: the lazy restatement (no seeds) differs from the faithful search")
        bpk = lc.algorithmic_bytes() / int(lc.kmers)
        legs["index_only"] = {"what": "kernel 4 on a replica without anchor table, k-mer table and string filter (seed_anchors 0, kmer_table 0): probes, streaming search, finimizer dictionaries, walk, text re-anchoring",
                              "reads": nl2, "ms_per_step": parts["step"], "kmers_per_s": nkm / (parts["step"] * 1e-3), "kernel_ms_parts": parts,
                              "algorithmic_bytes_per_kmer": bpk, "achieved": bpk * nkm / (parts["step"] * 1e-3) / 1e9, "frac": bpk * nkm / (parts["step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "index_bytes_hbm": ix2.size_in_bytes(), "derived_tables_bytes_hbm": ix2.replica_table_bytes(device),
                              "derived_tables_bytes_per_indexed_base": round(ix2.replica_table_bytes(device) / max(1, ix2.total_len), 1),
                              "prefix_table_depth": ix2.prefix_table_depth(device), "pairs": "equal to the default configuration's"}
        ix2.close()
        if idx.lean_tables(device):
            # round 3's tables (lean_tables 0: prefix table T = 15 + anchor table beside the k-mer table) on the same two million reads
            ix3 = fa.FinimizerIndex().load(os.path.join(tmp, "idx"))
            ix3.set_option("lean_tables", 0)
            ix3.to_device(device)
            got, parts, nkm, info = timed(ix3, reads.subset(0, nl2))
            if not np.array_equal(got, pairs[: got.shape[0]]):
                raise SystemExit("the configuration with round 3's tables differs from the default one on the first %d reads" % nl2)
            gotd, partsd, nkmd, _ = timed(idx, reads.subset(0, nl2))
            legs["full_tables"] = {"what": "lean_tables 0: the prefix table (T = %d) and the anchor table beside the k-mer table, probes through the prefix table, seeds through the anchor table (round 3's configuration)" % ix3.prefix_table_depth(device),
                                   "reads": nl2, "ms_per_step": parts["step"], "kmers_per_s": nkm / (parts["step"] * 1e-3), "kernel_ms_parts": parts,
                                   "default_ms_per_step_same_reads": partsd["step"],
                                   "derived_tables_bytes_hbm": ix3.replica_table_bytes(device), "derived_tables_bytes_per_indexed_base": round(ix3.replica_table_bytes(device) / max(1, ix3.total_len), 1),
                                   "pairs": "equal to the default configuration's"}
            ix3.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return legs


def end_to_end(fa, idx, reads, k, read_len, ns, device):
    """What a drop-in caller sees (never `value`): (1) fin_search_batch from page-locked host buffers, pairs back in host memory
    (H2D + step + D2H, pipelined); (2) the `finito search-fmin` command, plain FASTQ in -> the reference's text out."""
    import numpy as np
    sub = reads.subset(0, ns)
    nk = ns * max(0, read_len - k + 1)
    res = {"reads": ns, "kmers": nk}
    pin_b = fa.PinnedArray((ns * read_len,), np.uint8)
    pin_o = fa.PinnedArray((max(nk, 1), 2), np.int32)
    try:
        pin_b.array[:] = sub.bases
        idx.search_reads((pin_b.array, sub.offsets), fa.FIN_MERGED, out=pin_o.array)   # warm-up: buffers, streams
        t = time.perf_counter()
        idx.search_reads((pin_b.array, sub.offsets), fa.FIN_MERGED, out=pin_o.array)
        dt = time.perf_counter() - t
        res["pcie_inclusive_kmers_per_s"] = nk / dt
        res["pcie_inclusive_note"] = "fin_search_batch, page-locked host buffers in and out, sub-batches pipelined over 3 streams"
        # the same reads with the results as RECORDS (fin_search_batch_records: 32 bytes per read the fast path finishes, the other reads' pairs in one stream)
        # -- and, separately, the host-side expansion back to pairs (fin_expand_records), which is memory-bound host work a consumer of runs does not need
        import ctypes as C
        pin_r = fa.PinnedArray((ns * 32,), np.uint8)
        try:
            L = fa.lib()
            L.fin_search_batch_records.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
            offs = np.ascontiguousarray(sub.offsets, dtype=np.uint64)
            got = C.c_uint64(0); err = C.create_string_buffer(512)
            def run_records():
                rc = L.fin_search_batch_records(idx.h, pin_b.array.ctypes.data_as(C.c_char_p), offs.ctypes.data_as(C.POINTER(C.c_uint64)), ns, pin_r.array.ctypes.data_as(C.c_void_p),
                                                pin_o.array.ctypes.data_as(C.c_void_p), nk, C.byref(got), err, 512)
                if rc != 0:
                    raise RuntimeError(err.value.decode(errors="replace"))
            run_records()
            t = time.perf_counter(); run_records(); dt_r = time.perf_counter() - t
            recs = pin_r.array.view(fa.RECORD_DTYPE)
            stream = pin_o.array[: int(got.value)]
            t = time.perf_counter(); pairs_x, npos_x = fa.expand_records(recs, stream, k); dt_x = time.perf_counter() - t
            ref_pairs, _ = idx.search_reads((pin_b.array, sub.offsets), fa.FIN_MERGED)
            res["records"] = {"pcie_inclusive_kmers_per_s": nk / dt_r, "bytes_to_host_per_read": (32.0 * ns + 8.0 * int(got.value)) / ns,
                              "reads_as_one_record": int(((recs["meta"] >> 16) != 0).sum()), "stream_pairs": int(got.value),
                              "expand_on_host_kmers_per_s": nk / dt_x, "expand_threads": fa.host_threads(),
                              "pairs_after_expansion": "equal to fin_search_batch's" if np.array_equal(pairs_x, ref_pairs) else "DIFFER",
                              "note": "fin_search_batch_records, page-locked buffers: a 32-byte record per read the fast path finishes instead of its pairs, the other reads' pairs in one stream; fin_expand_records (host threads) makes the same pairs"}
            if res["records"]["pairs_after_expansion"] == "DIFFER":
                raise SystemExit("records + stream do not expand to fin_search_batch's pairs")
            del recs, stream, pairs_x, ref_pairs
        finally:
            pin_r.close()
    finally:
        pin_b.close(); pin_o.close()
    cli = os.path.join(ROOT, "finito_amd", "finito")
    if os.path.exists(cli):
        import tempfile
        ncli = min(ns, 2_000_000)
        tmp = tempfile.mkdtemp(prefix="finito_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            idx.serialize(os.path.join(tmp, "idx"))
            fq = os.path.join(tmp, "reads.fastq")
            b = sub.bases[: ncli * read_len].reshape(ncli, read_len)
            rec = np.empty((ncli, 2 * read_len + 7), dtype=np.uint8)   # "@r\n" bases "\n+\n" quals "\n"
            rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
            rec[:, 3:3 + read_len] = b
            rec[:, 3 + read_len:6 + read_len] = np.frombuffer(b"\n+\n", dtype=np.uint8)
            rec[:, 6 + read_len:6 + 2 * read_len] = ord("I")
            rec[:, 6 + 2 * read_len] = ord("\n")
            rec.tofile(fq)
            t = time.perf_counter()
            p = subprocess.run([cli, "search-fmin", "-i", os.path.join(tmp, "idx"), "-q", fq, "-o", os.path.join(tmp, "out.txt"), "--gpus", "1"],
                               capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(device))))
            dt = time.perf_counter() - t
            if p.returncode != 0:
                res["cli_error"] = (p.stderr or "")[-300:]
            else:
                res["cli_us_per_kmer"] = 1e6 * dt / (ncli * max(0, read_len - k + 1))
                res["cli_kmers_per_s"] = ncli * max(0, read_len - k + 1) / dt
                res["cli_note"] = "`finito search-fmin` on %d reads, plain FASTQ -> reference text format, wall time of the whole process (start, index load + upload, search, formatting, write)" % ncli
                res["cli_output_bytes"] = os.path.getsize(os.path.join(tmp, "out.txt"))
                import re
                m = re.search(r"us/query: ([0-9.eE+-]+) \(excluding I/O", p.stderr or "")
                if m and float(m.group(1)) > 0:   # the command's own log line (search_fmin.hh:78): its pipeline without start-up and file I/O
                    res["cli_pipeline_us_per_kmer"] = float(m.group(1))
                    res["cli_pipeline_kmers_per_s"] = 1e6 / float(m.group(1))
        finally:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    else:
        run_rank(args)


if __name__ == "__main__":
    main()
