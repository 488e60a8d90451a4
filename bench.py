#!/usr/bin/env python
"""bench.py — QP paths/sec of the batched solve (BASELINE.json metric) on N GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (assembly + factorisation + ADMM to eps 1e-4 + output map, one fused
kernel launch) over one batch of synthetic planning instances already resident in HBM:
  N=1   BASELINE config 3: B=4096 paths, N=200 points, KP, per-path random obstacle clearances
  N>1   BASELINE config 4: the same generator, 4096 paths per GPU (contiguous shard of path ids), no
        data-path collective (paths are independent); RCCL is used only for the barrier / max-time reduction.
Rank 0 prints ONE JSON line.  `roofline.achieved` is ALGORITHMIC bytes (B_iter = 8*(128N+13C+9)
bytes per path-iteration + 8*(18N+8) compulsory I/O per path) / measured kernel time — NOT HBM traffic: the
solver state is LDS-resident, so `frac` can exceed what HBM could stream; measured HBM traffic (rocprofv3 PMC,
profiles/) is reported beside it as `traffic`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes(form, N, keep, iters_sum, B):
    C = (N + keep - 2) // keep if form != 2 else N - 1
    vals = {0: 128 * N + 13 * C + 9, 1: 139 * N + 32 * C + 9, 2: 119 * N - 20}[form]
    return 8.0 * vals * iters_sum + 8.0 * (18 * N + 8) * B, 8.0 * vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="paths per GPU")
    ap.add_argument("--config", type=int, default=0, help="BASELINE config id (default: 3 at N=1, 4 at N>1)")
    ap.add_argument("--cpu-sample", type=int, default=768, help="paths timed on the CPU oracle (rank 0, N=1 only)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(dev))  # RCCL; used for the barrier and 4 scalars only

    from path_optimizer_amd import binding, synth

    cfg = args.config or (3 if world == 1 else 4)
    B = args.batch
    from path_optimizer_amd.shard import shard_range

    lo, hi = shard_range(world * B, world, rank)  # weak scaling: fixed work per GPU, contiguous path ids
    batch = synth.make_batch(cfg, B=hi - lo, first_path=lo)
    dbatch = binding.DeviceBatch(batch, device=dev)
    eng = binding.Engine(local_rank)
    stream = torch.cuda.Stream(device=dev)  # a real (non-null) HIP stream shared by torch events and the engine
    eng.set_stream(stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.solve_batch_device(dbatch)
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        eng.solve_batch_device(dbatch)
        e1.record(stream)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    info = dbatch.info_numpy()
    stats = torch.tensor([float(info["iters"].sum()), float((info["status"] != 1).sum()), float(info["iters"].max())],
                         dtype=torch.float64, device=dev)
    if world > 1:
        tot = stats.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        iters_sum_all, unsolved_all, iters_max = float(tot[0]), float(tot[1]), float(mx[2])
    else:
        iters_sum_all, unsolved_all, iters_max = float(stats[0]), float(stats[1]), float(stats[2])

    if rank == 0:
        N, keep, form = batch.N, batch.keep, batch.formulation
        paths_per_s = world * B * args.steps / elapsed
        abytes, b_iter = algorithmic_bytes(form, N, keep, float(info["iters"].sum()), B)
        achieved = abytes / (kernel_ms * 1e-3) / 1e9  # GB/s, this rank's kernel
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "QP paths/sec at N=200 pts, batch=4096; ADMM iters to 1e-4",
            "value": paths_per_s,
            "unit": "paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {cfg}: KP, B={B} paths/GPU x {world} GPU, N={N} points, "
                                   "per-path random obstacle clearances, OSQP defaults (scaling 10, adaptive rho every 100 it) at eps_abs=eps_rel=1e-4",
                       "batch_per_gpu": B, "points": N, "formulation": "KP", "parallelism": f"batch-split x{world}"},
            "admm": {"iters_mean": iters_sum_all / (world * B), "iters_max": iters_max, "unsolved": int(unsolved_all),
                     "path_iters_per_s": iters_sum_all * args.steps / elapsed},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "kernel": "po::solve_kernel_fast<KP,SPL=4,NT=64,two-level>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_path_iter": b_iter,
                         "note": "algorithmic bytes / kernel time; state is LDS-resident so this is not HBM traffic",
                         # the bound that actually limits the kernel: fp64 VALU issue, v_fma_f64 = 8 cycles per
                         # wave-instruction measured (tools/ubench) -> 1024 SIMDs x 2.4 GHz x 64 lanes x 2 / 8 = 39.3 TFLOP/s
                         "secondary": {"bound": "fp64_valu", "unit": "TFLOP/s", "peak": 39.3,
                                       "achieved": float(info["iters"].sum()) * 1.0e5 / (kernel_ms * 1e-3) / 1e12,
                                       "note": "0.1 MFLOP per path-iteration (SURVEY.md §8a10)"}},
        }
        if world == 1 and args.cpu_sample > 0:
            from oracle import oracle_py  # CPU baseline leg only

            ns = min(args.cpu_sample, B)
            sample = batch.slice(0, ns)
            oracle_py.solve_batch(sample.slice(0, 2), oracle_py.device_equivalent_params())  # warm the ordering cache
            c0 = time.perf_counter()
            _, oinfo, _ = oracle_py.solve_batch(sample, oracle_py.device_equivalent_params(), want_x=False)
            c1 = time.perf_counter()
            out["cpu_baseline"] = {"value": ns / (c1 - c0), "unit": "paths/s", "cores": 1, "kind": "port",
                                   "sample": f"first {ns} paths of the same batch, oracle/libpo_oracle.so (OSQP-style ADMM, "
                                             f"sparse LDL', gcc -O3), {c1 - c0:.1f} s, mean iters {float(oinfo['iters'].mean()):.1f}",
                                   "host_cpus": os.cpu_count()}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
