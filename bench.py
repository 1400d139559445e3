#!/usr/bin/env python3
"""bench.py -- localized k-mers/s of the search-fmin path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (both strands + merge, search_fmin.hh:43-72) over one batch of synthetic reads that
is already resident in HBM, results left in HBM.  Default workload = BASELINE.json configs[2] ("chr1"): 250 Mbp
synthetic unitigs, k=31, t=1, 10 M x 150 bp reads per GPU.  With N > 1 ranks every rank holds a replica of the index
and its own shard of the read records (weak scaling, no collective on the data path; torch.distributed is used only
for the barrier and the max-over-ranks clock).

Prints ONE JSON line on rank 0 with `roofline` (algorithmic bytes per launch, SURVEY.md 8(d), counted by the CPU
oracle on a read sample / average kernel duration from HIP events on the launch stream) and `cpu_baseline` (the CPU
oracle -- a port: the reference binary cannot be built here -- timed on the host on a bounded read sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (genome bases, k, read_len, reads per GPU, BASELINE.json config it is)
    "chr1": (250_000_000, 31, 150, 10_000_000, "configs[2]: 250 Mbp synthetic unitigs k=31 t=1, 10 M 150 bp reads per GPU"),
    "ecoli": (5_000_000, 31, 150, 1_000_000, "configs[1]: 5 Mbp synthetic unitigs k=31 t=1, 1 M 150 bp reads"),
    "k63": (250_000_000, 63, 250, 10_000_000, "configs[4] at t=1: 250 Mbp synthetic unitigs k=63, 10 M 250 bp reads"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="chr1", choices=sorted(WORKLOADS))
    ap.add_argument("--genome", type=int, default=0, help="override genome size (bases)")
    ap.add_argument("--reads", type=int, default=0, help="override reads per GPU")
    ap.add_argument("--cpu-sample", type=int, default=60_000, help="reads in the CPU-baseline / parity sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (and the oracle parity sample)")
    ap.add_argument("--kernel", type=int, default=-1)
    args = ap.parse_args()

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
    if "FINITO_PTAB_T" in os.environ:   # experiments: depth of the prefix table (default: by index size)
        assert fa.lib().fin_set_option(b"ptab_t", int(os.environ["FINITO_PTAB_T"])) == 0

    gsize, k, read_len, n_reads, desc = WORKLOADS[args.workload]
    if args.genome:
        gsize = args.genome
    if args.reads:
        n_reads = args.reads

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs: index built once by rank 0, replicated through a container file; reads sharded by record ----
    t0 = time.time()
    g = synth.genome(gsize)
    u = synth.unitigs(g, k)
    prefix = "/dev/shm/finito_bench_%s_%d_%d" % (args.workload, gsize, os.getppid() if world > 1 else os.getpid())
    if rank == 0:
        idx = fa.FinimizerIndex.build(u.as_tuple(), k)
        log("index built in %.1f s: %d nodes, %d k-mers, %d unitigs, %d finimizers, %.1f MB in HBM"
            % (time.time() - t0, idx.n_nodes, idx.n_kmers, idx.n_unitigs, idx.n_finimizers, idx.size_in_bytes() / 1e6))
        if world > 1:
            idx.serialize(prefix)
    if dist is not None:
        dist.barrier()
        if rank != 0:
            idx = fa.FinimizerIndex().load(prefix)
        dist.barrier()
        if rank == 0:
            try:
                os.unlink(prefix + ".finamd")
            except OSError:
                pass
    idx.to_device(local_rank)
    # rank r holds records [r*n_reads, (r+1)*n_reads) of the global read set (the generator is seeded per record)
    t1 = time.time()
    reads = synth.reads(g, n_reads, read_len=read_len, seed=synth.SEED_READS + 7919 * rank)
    batch = idx.batch(reads.as_tuple())
    n_kmers = batch.n_kmers
    log("rank %d: %d reads (%d k-mers) resident in HBM, generated+uploaded in %.1f s" % (rank, n_reads, n_kmers, time.time() - t1))
    stream = torch.cuda.current_stream().cuda_stream

    # ---- timed region ----
    for _ in range(args.warmup):
        batch.run(fa.FIN_MERGED, stream)
    barrier()
    warm_ms, warm_n = batch.kernel_time_ms()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        batch.run(fa.FIN_MERGED, stream)
    barrier()
    elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if (dist is None or dist.get_backend() == "nccl") else "cpu")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    all_ms, all_n = batch.kernel_time_ms()
    kern_ms = (all_ms * all_n - warm_ms * warm_n) / max(1, all_n - warm_n)   # average over the timed launches only
    pre_ms, srch_ms, parts_n = batch.kernel_time_parts_ms()   # kernel 3: probe pre-pass + search kernel (all launches, warm-up included)

    # ---- checks on the results of the timed launches (rank 0 carries the oracle leg) ----
    pairs, n_pos = batch.download()
    bad, checked, first_bad = synth.check_ground_truth(idx, u, reads, pairs)
    if bad:
        raise SystemExit("rank %d: %d of %d error-free k-mers localized wrongly (first bad read %d)" % (rank, bad, checked, first_bad))
    log("rank %d: ground truth ok on %d error-free k-mers; %d of %d k-mers found" % (rank, checked, n_pos, n_kmers))

    out = None
    if rank == 0:
        value = world * n_kmers * args.steps / elapsed
        out = {
            "metric": "localized k-mers/s (k=%d)" % k, "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s = BASELINE.json %s" % (args.workload, desc), "k": k, "t": 1, "index_bases": gsize,
                       "index_nodes": idx.n_nodes, "index_bytes_hbm": idx.size_in_bytes(),
                       "prefix_table_bytes_hbm": 8 * 4 ** idx.prefix_table_depth(local_rank) if idx.prefix_table_depth(local_rank) > 0 else 0, "reads_per_gpu": n_reads,
                       "read_len": read_len, "kmers_per_gpu_per_step": n_kmers, "strands": "both, merged",
                       "parallelism": "reads sharded by record, index replicated, no collective",
                       "kernel": "v%d" % (args.kernel if args.kernel >= 0 else 3), "ground_truth_checked_kmers": checked},
        }
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "fin_search_v%d_kernel" % (args.kernel if args.kernel >= 0 else 3), "kernel_ms": kern_ms}
        if parts_n:   # the hot path is two launches per step: kernel_ms is their sum (HIP events around both)
            roof["kernel"] += " + fin_probe_kernel (pre-pass)"
            roof["kernel_ms_parts"] = {"fin_probe_kernel": pre_ms, "fin_search_v3_kernel": srch_ms}
        if not args.no_cpu:
            from oracle.oracle import Counters, OracleIndex
            ns = min(args.cpu_sample, n_reads)
            t2 = time.time()
            oracle = OracleIndex.from_components(k, idx.components())
            log("oracle assembled from exported components in %.1f s" % (time.time() - t2))
            sample = reads.subset(0, ns)
            ctr = Counters()
            exp, _, _ = oracle.search_batch(sample.as_tuple(), counters=ctr, n_threads=fa.host_threads())
            if not np.array_equal(pairs[: exp.shape[0]].astype(np.int64), exp):
                raise SystemExit("HIP output differs from the CPU oracle on the %d-read sample" % ns)
            # timed leg: single thread, search + merge + text formatting exactly as the reference's timed region
            _, secs, _ = oracle.search_batch(sample.as_tuple(), want_pairs=False, format_text=True, n_threads=1)
            _, secs_nofmt, _ = oracle.search_batch(sample.as_tuple(), want_pairs=False, format_text=False, n_threads=1)
            ncores = fa.host_threads()
            _, secs_all, _ = oracle.search_batch(sample.as_tuple(), want_pairs=False, format_text=True, n_threads=ncores)
            sk = int(ctr.kmers)
            out["cpu_baseline"] = {"value": sk / secs, "unit": "k-mers/s", "cores": 1, "kind": "port",
                                   "sample": "first %d reads of rank 0's batch (%d k-mers), oracle in reference-shaped mode: two "
                                             "searches per read + merge + text formatting (search_fmin.hh:46-71)" % (ns, sk),
                                   "search_only_value": sk / secs_nofmt, "all_cores_value": sk / secs_all, "all_cores": ncores}
            bytes_per_kmer = ctr.algorithmic_bytes() / sk
            alg_bytes = bytes_per_kmer * n_kmers
            roof["achieved"] = alg_bytes / (kern_ms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            roof["algorithmic_bytes_per_kmer"] = bytes_per_kmer
            roof["oracle_counters_per_base_strand"] = {kk: vv / ctr.base_strands for kk, vv in ctr.as_dict().items()
                                                        if kk in ("extends", "rank_lines", "drops", "lcs_lines", "lcs_entries", "anchors", "walked")}
            out["config"]["parity"] = "bit-exact vs CPU oracle on the first %d reads" % ns
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"]
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get("%s:%d" % (args.workload, n_reads))
                if ent and ent.get("kernel", "v3") == out["config"]["kernel"]:
                    roof["traffic"] = ent["hbm_bytes_per_launch"]
                    roof["traffic_source"] = ent.get("source")
                    # what the kernel really moves: it skips work the reference algorithm does (walk mode, probes), so the
                    # algorithmic figure above (the reference's working set per k-mer / time) can exceed the HBM peak
                    roof["hbm_measured"] = ent["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9
                    roof["hbm_measured_frac"] = roof["hbm_measured"] / HBM_PEAK_GBS
            except Exception:
                pass
        out["roofline"] = roof
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
