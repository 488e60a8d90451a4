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
