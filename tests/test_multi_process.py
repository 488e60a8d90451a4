"""world_size-2 test of the N>1 path on CPU (gloo): the batch split, the statistics reduction and the property that
sharded results equal the single-process results.  The compute leg uses the oracle (tests may), the plumbing under
test is path_optimizer_amd/shard.py exactly as bench.py uses it."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_and_balance():
    from path_optimizer_amd.shard import shard_range

    for total in (0, 1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(32768, 8, 3) == (3 * 4096, 4 * 4096)  # BASELINE config 4


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import oracle_py
    from path_optimizer_amd import synth
    from path_optimizer_amd.shard import reduce_stats, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(6, world, rank)
    batch = synth.make_batch(2, B=hi - lo, first_path=lo)
    st, info, xs = oracle_py.solve_batch(batch, oracle_py.device_equivalent_params())
    tot = reduce_stats(float(info["iters"].sum()), float((info["status"] != 1).sum()), float(info["iters"].max()), 0.1 * (rank + 1))
    q.put((rank, lo, hi, xs, tot))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_split_matches_single_process():
    import torch.multiprocessing as mp

    from oracle import oracle_py
    from path_optimizer_amd import synth

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = synth.make_batch(2, B=6)
    st, info, xs = oracle_py.solve_batch(full, oracle_py.device_equivalent_params())
    cat = np.concatenate([g[3] for g in got])
    assert np.array_equal(cat, xs)  # shards are bit-identical to the single-process solve: no cross-path coupling
    for g in got:
        assert g[4][0] == float(info["iters"].sum()) and g[4][2] == float(info["iters"].max()) and abs(g[4][3] - 0.2) < 1e-12


def test_bench_n_gt_1_branch_dry_run():
    """bench.py's N>1 branch end to end on CPU: `torch.distributed.run` with 2 ranks, gloo, --dry-run (device times faked, everything else
    real: shard_range, the SUM / MAX reductions, the root gather of SURVEY §8e, the JSON line of rank 0)."""
    import json
    import subprocess

    port = 29700 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "96", "--dry-run", "--gather"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4 and d["config"]["batch_per_gpu"] == 96
    # faked device times: rank r takes 10 ms x steps x (1 + r/4) -> max over ranks = rank 1, imbalance = 1.25 / 1.125
    assert abs(d["ms_per_step"] - 12.5) < 1e-9 and abs(d["value"] - 2 * 96 * 4 / 0.05) < 1e-6
    assert abs(d["config"]["rank_time_max_over_mean"] - 1.25 / 1.125) < 1e-12
    # statistics are functions of the GLOBAL path id: the reductions must reproduce the single-process figures
    ids = np.arange(2 * 96)
    it = 25 * (1 + (ids * 2654435761 % 61))
    assert d["admm"]["iters_mean"] == it.mean() and d["admm"]["iters_max"] == it.max() and d["admm"]["refactorisations"] == int((it // 400).sum())
    # which ranks took part and what they ran on (so that the first real SCALE record can be checked the same way)
    rk = d["config"]["ranks"]
    assert rk["world"] == 2 and rk["rccl_ranks_seen"] == 2 and rk["backend"] == "gloo"
    assert [r["rank"] for r in rk["ranks"]] == [0, 1] and [r["paths"] for r in rk["ranks"]] == [[0, 96], [96, 192]]
    g = d["gather"]
    assert g["paths_on_root"] == 192 and g["iters_sum_on_root"] == float(it.sum()) and g["path_ids_in_order"] is True


def test_bench_gpus_flag_means_n_gpus_without_torchrun():
    """`python bench.py --gpus 2 --dry-run` started WITHOUT torchrun (the shape of the driver's N = 1 command): the script re-runs itself under
    torch.distributed.run with two ranks — n_gpus 2, two ranks seen, contiguous path ranges [0, B), [B, 2B) (the flag used to be ignored)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--dry-run"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    rk = d["config"]["ranks"]
    assert d["n_gpus"] == 2 and rk["world"] == 2 and rk["rccl_ranks_seen"] == 2
    assert [q["paths"] for q in rk["ranks"]] == [[0, 64], [64, 128]]


def test_bench_refuses_a_world_that_is_not_gpus():
    """Under torchrun with WORLD_SIZE != --gpus the bench exits non-zero instead of printing a line labelled with the wrong GPU count."""
    import subprocess

    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8", "--dry-run"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
