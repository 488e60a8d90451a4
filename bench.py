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


def stage_legs(torch, binding, synth, eng, stream, dbatch, B, with_cpu):
    """Stages immediately before / after the QP on the same device (reported beside the headline, never part of `value`):
    corridor-bounds producer over a synthetic obstacle-distance map and the post-solve collision check of the solved batch."""
    d, res, px, py, _ = synth.make_distance_map(seed=3, size_x=600, size_y=600, resolution=0.2, pos=(1.0, -2.0), n_obstacles=60, r_range=(0.5, 3.0))
    eng.set_map(d, res, px, py)
    nb = 256
    P = synth.make_spline_paths(1, nb, 200)
    keys = ("ref_x", "ref_y", "ref_z", "ref_s", "knot_s", "knot_x", "knot_y")
    reps = -(-B // nb)
    t = {k: torch.from_numpy(np.ascontiguousarray(np.concatenate([P[k]] * reps, axis=0)[:B])).cuda() for k in keys}
    bounds = torch.zeros((B, 200, 4, 2), dtype=torch.float64, device="cuda"); nvalid = torch.zeros(B, dtype=torch.int32, device="cuda")
    nkeep = torch.zeros(B, dtype=torch.int32, device="cuda"); ok = torch.zeros_like(nkeep)

    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ms_b = timed(lambda: eng.bounds_batch_device(t, bounds, nvalid))
    ms_c = timed(lambda: eng.postcheck_batch_device(dbatch, nkeep, ok))
    st = {"map": "600 x 600 cells at 0.2 m (1.44 MB float32), 60 random discs",
          "bounds_producer": {"ms": ms_b, "paths_per_s": B / (ms_b * 1e-3), "circle_clearances_per_s": B * 200 * 4 / (ms_b * 1e-3),
                              "kept_states_mean": float(nvalid.float().mean().item()),
                              "workload": f"{B} spline reference paths x 200 states x 4 circles, <= 28 bilinear samples each"},
          "post_check": {"ms": ms_c, "paths_per_s": B / (ms_c * 1e-3), "states_per_s": B * 200 / (ms_c * 1e-3),
                         "ok_frac": float(ok.float().mean().item())}}
    # reference-smoothing QPs: TENSION2 (100 points) and the post-smoothing QP (60 layers), 4096 instances each
    from path_optimizer_amd.abi import INFO_DTYPE
    sm = {}
    sm_inputs = {}
    for name, kind, npts in (("tension2", 0, 100), ("post", 2, 60)):
        si = synth.make_smooth_inputs(30, 256, P=npts, kind=kind)
        sm_inputs[name] = (kind, si)
        tt = {k: torch.from_numpy(np.ascontiguousarray(np.concatenate([v] * reps, axis=0)[:B])).cuda() for k, v in si.items() if v is not None}
        so = dict(x=torch.zeros((B, npts), dtype=torch.float64, device="cuda"), y=torch.zeros((B, npts), dtype=torch.float64, device="cuda"),
                  s=torch.zeros((B, npts), dtype=torch.float64, device="cuda"), info=torch.zeros((B, 48), dtype=torch.uint8, device="cuda"))
        ms_s = timed(lambda: eng.smooth_batch_device(kind, tt, so))
        inf = so["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
        n_q, m_q = binding.smooth_dims(kind, npts)
        sm[name] = {"ms": ms_s, "qps_per_s": B / (ms_s * 1e-3), "qp_iters_per_s": float(inf["iters"].sum()) / (ms_s * 1e-3), "iters_mean": float(inf["iters"].mean()),
                    "iters_max": int(inf["iters"].max()), "unsolved": int((inf["status"] != 1).sum()), "points": npts, "n": n_q, "m": m_q}
    st["smoothing_qps"] = sm
    # DP lattice search and curvature-adaptive re-sampling on 4096 spline references
    spn, length, start = synth.make_search_inputs(9, nb)
    rp = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate([a] * reps, axis=0)[:B])).cuda()
    ts = {k: rp(spn[k]) for k in ("knot_s", "knot_x", "knot_y")}
    ts["length"] = rp(length)
    tstart = rp(start)
    Lc = 64
    so = dict(layer_s=torch.zeros((B, Lc), dtype=torch.float64, device="cuda"), lb=torch.zeros((B, Lc), dtype=torch.float64, device="cuda"),
              ub=torch.zeros((B, Lc), dtype=torch.float64, device="cuda"), l0=torch.zeros(B, dtype=torch.float64, device="cuda"),
              n_layers=torch.zeros(B, dtype=torch.int32, device="cuda"))
    ms_d = timed(lambda: eng.dp_search_batch_device(ts, tstart, Lc, so))
    nl = so["n_layers"].clamp(min=0).double()
    ro = {k: torch.zeros((B, 256), dtype=torch.float64, device="cuda") for k in ("ref_x", "ref_y", "ref_z", "ref_k", "ref_s")}
    ro["n_points"] = torch.zeros(B, dtype=torch.int32, device="cuda")
    ms_r = timed(lambda: eng.resample_batch_device(ts, 0.15, 0.3, 256, ro))
    st["dp_search"] = {"ms": ms_d, "paths_per_s": B / (ms_d * 1e-3), "layers_mean": float(nl.mean().item()),
                       "edge_evaluations_per_s": float(nl.sum().item()) * 34 * 34 / (ms_d * 1e-3), "workload": f"{B} spline references, 34 lateral samples per layer, layers every 1.5 m"}
    st["resample"] = {"ms": ms_r, "paths_per_s": B / (ms_r * 1e-3), "states_mean": float(ro["n_points"].double().mean().item())}
    # PathOptimizer::solve end to end (po_plan_batch_device): waypoints + start + goal -> final path, every stage on the device
    scn = synth.make_planning_scenes(2, 64)
    eng.set_map(*scn["map"])
    rs = -(-B // 64)
    # replicate the 64 scenes in a shuffled order: blocks go to the 8 XCDs round-robin, a period-64 pattern would pin scenes to XCDs
    perm = np.random.default_rng(5).permutation(64 * rs)[:B] % 64
    tp = {k: torch.from_numpy(np.ascontiguousarray(scn[k][perm])).cuda() for k in ("way_x", "way_y", "start", "goal")}
    Np = 320
    po_ = dict(states=torch.zeros((B, Np, 5), dtype=torch.float64, device="cuda"), n_states=torch.zeros(B, dtype=torch.int32, device="cuda"),
               ok=torch.zeros(B, dtype=torch.int32, device="cuda"), stage=torch.zeros(B, dtype=torch.int32, device="cuda"), info=torch.zeros((B, 48), dtype=torch.uint8, device="cuda"))
    way_len = float(np.hypot(np.diff(scn["way_x"], axis=1), np.diff(scn["way_y"], axis=1)).sum(axis=1).max())
    eng.plan_batch_device(tp, po_, Np, way_len); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.plan_batch_device(tp, po_, Np, way_len)
    torch.cuda.synchronize()
    ms_p = (time.perf_counter() - t0) / 3 * 1e3  # wall clock: the call synchronises mid-way to group the QPs by keep_control_steps_
    pinf = po_["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
    st["full_pipeline"] = {"ms": ms_p, "instances_per_s": B / (ms_p * 1e-3), "ok_frac": float(po_["ok"].double().mean().item()),
                           "states_mean": float(po_["n_states"].double().mean().item()), "qp_iters_mean": float(pinf["iters"].mean()),
                           "workload": f"{B} planning instances (24 waypoints over ~70 m, 60-disc map 700 x 700 cells): bSpline -> TENSION2 QP -> DP search -> post QP -> "
                                       "re-sampling (0.15..0.3 m) -> bounds -> KP QP -> collision check"}
    # serving pattern: consecutive batches from 3 host threads, one handle (= stream) each; the QP tail of one batch drains under the next
    import threading
    engs3, outs3 = [], []
    for _ in range(3):
        e3 = binding.Engine(torch.cuda.current_device()); e3.set_map(*scn["map"]); engs3.append(e3)
        outs3.append({k: torch.zeros_like(v) for k, v in po_.items()})
        e3.plan_batch_device(tp, outs3[-1], Np, way_len)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda i=i: [engs3[i].plan_batch_device(tp, outs3[i], Np, way_len) for _ in range(2)]) for i in range(3)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    ms_p3 = (time.perf_counter() - t0) / 6 * 1e3
    st["full_pipeline"]["pipelined_3_handles"] = {"ms_per_batch": ms_p3, "instances_per_s": B / (ms_p3 * 1e-3)}
    for e3 in engs3:
        e3.close()
    # the same at the reference's own stopping point: it never touches OSQP's eps (default 1e-3); 1e-4 above is this project's metric
    p3 = binding.default_params(); p3.eps_abs = p3.eps_rel = 1e-3
    e4 = binding.Engine(torch.cuda.current_device(), p3); e4.set_map(*scn["map"])
    e4.plan_batch_device(tp, po_, Np, way_len); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e4.plan_batch_device(tp, po_, Np, way_len)
    torch.cuda.synchronize()
    ms_p4 = (time.perf_counter() - t0) / 3 * 1e3
    pinf4 = po_["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
    st["full_pipeline"]["at_osqp_default_eps_1e-3"] = {"ms": ms_p4, "instances_per_s": B / (ms_p4 * 1e-3), "ok_frac": float(po_["ok"].double().mean().item()),
                                                       "qp_iters_mean": float(pinf4["iters"].mean())}
    e4.close()
    eng.set_map(d, res, px, py)
    if with_cpu:
        from oracle import oracle_py

        m = oracle_py.make_map(d, res, px, py); p = oracle_py.default_params()
        ms_ = oracle_py.make_map(*scn["map"])
        c6 = time.perf_counter()
        for b in range(16):
            oracle_py.path_optimizer_solve(p, ms_, scn["way_x"][b], scn["way_y"][b], scn["start"][b], scn["goal"][b])
        cpu_pipeline = 16 / (time.perf_counter() - c6)
        c0 = time.perf_counter()
        for b in range(64):
            oracle_py.bounds_path(p, m, *[P[k][b] for k in keys])
        c1 = time.perf_counter()
        states = dbatch.out_states[:64].cpu().numpy(); info = dbatch.info_numpy()[:64]
        oracle_py.postcheck_batch(p, m, states, info)
        c2 = time.perf_counter()
        st["cpu_port"] = {"bounds_paths_per_s": 64 / (c1 - c0), "post_check_paths_per_s": 64 / (c2 - c1), "cores": 1, "sample": "64 paths each, oracle (C)"}
        for name, (kind, si) in sm_inputs.items():
            c3 = time.perf_counter()
            oracle_py.smooth_batch(kind, p, {k: (None if v is None else v[:64]) for k, v in si.items()})
            st["cpu_port"][f"smoothing_{name}_qps_per_s"] = 64 / (time.perf_counter() - c3)
        c4 = time.perf_counter()
        for b in range(64):
            oracle_py.dp_search(p, m, spn["knot_s"][b], spn["knot_x"][b], spn["knot_y"][b], length[b], start[b], cap=Lc)
        c5 = time.perf_counter()
        for b in range(64):
            oracle_py.resample(p, spn["knot_s"][b], spn["knot_x"][b], spn["knot_y"][b], length[b], 0.15, 0.3, cap=256)
        st["cpu_port"]["full_pipeline_instances_per_s"] = cpu_pipeline
        st["cpu_port"]["dp_search_paths_per_s"] = 64 / (c5 - c4)
        st["cpu_port"]["resample_paths_per_s"] = 64 / (time.perf_counter() - c5)
    return st


def _cpu_slice(arg):
    """Worker of the all-cores CPU baseline leg (oracle, test infrastructure)."""
    from oracle import oracle_py

    batch, _ = arg
    oracle_py.solve_batch(batch, oracle_py.device_equivalent_params(), want_x=False)
    return batch.B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="paths per GPU")
    ap.add_argument("--config", type=int, default=0, help="BASELINE config id (default: 3 at N=1, 4 at N>1)")
    ap.add_argument("--streams", type=int, default=3, help="handles/HIP streams the consecutive steps are issued on round-robin "
                    "(independent batches: the stragglers of step k drain while step k+1 fills the CUs); 1 = strictly serial steps")
    ap.add_argument("--serial-leg", action="store_true", help="additionally time the same K steps strictly serially on one stream "
                    "and report them under \"serial\" (off by default so that a profile of the default command sees only the timed pattern)")
    ap.add_argument("--no-stages", action="store_true", help="skip the (untimed) legs for the stages around the QP: corridor-bounds producer "
                    "and post-solve collision check (SURVEY.md §8f-1/2)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="paths timed on the CPU oracle (rank 0, N=1 only)")
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
    # S independent handles, each with its own HIP stream and its own output buffers (inputs are shared, read-only)
    S = max(1, min(args.streams, max(args.steps, 1)))
    engs, streams, dbs = [], [], []
    for i in range(S):
        e = binding.Engine(local_rank)
        st = torch.cuda.Stream(device=dev)  # a real (non-null) HIP stream shared by torch events and the engine
        e.set_stream(st.cuda_stream)
        engs.append(e); streams.append(st)
        dbs.append(dbatch if i == 0 else dbatch.clone_outputs())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(steps, n_streams):
        """`steps` complete solves of the batch, step k on handle k % n_streams; returns (wall seconds, per-launch ms)."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k, (e0, e1) in enumerate(evs):
            i = k % n_streams
            e0.record(streams[i])
            engs[i].solve_batch_device(dbs[i])
            e1.record(streams[i])
        barrier()
        t1 = time.perf_counter()
        return t1 - t0, float(np.mean([a.elapsed_time(b) for a, b in evs]))

    # warm-up: issued exactly like the timed steps (round-robin over the handles) and timed per launch as well, so that the mean over
    # ALL launches of the process is available for comparison with rocprofv3's per-kernel average (which cannot tell warm-up from timed)
    _, warm_kernel_ms = run(args.warmup, S) if args.warmup > 0 else (0.0, None)
    elapsed, kernel_ms = run(args.steps, S)
    kernel_ms_all = kernel_ms if warm_kernel_ms is None else (warm_kernel_ms * args.warmup + kernel_ms * args.steps) / (args.warmup + args.steps)
    rank_time_max_over_mean = 1.0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tsum = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        rank_time_max_over_mean = float(t.item()) / (float(tsum.item()) / world)  # load imbalance between the shards
        elapsed = float(t.item())
    # the same K steps strictly one after the other on one stream (reported beside the headline, not as `value`)
    serial_elapsed, serial_kernel_ms = run(args.steps, 1) if (S > 1 and args.serial_leg) else ((elapsed, kernel_ms) if S == 1 else (None, None))
    if world > 1 and serial_elapsed is not None:
        t = torch.tensor([serial_elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        serial_elapsed = float(t.item())
    info = dbatch.info_numpy()
    stats = torch.tensor([float(info["iters"].sum()), float((info["status"] != 1).sum()), float(info["iters"].max()),
                          float(info["n_refactor"].sum())], dtype=torch.float64, device=dev)
    if world > 1:
        tot = stats.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        iters_sum_all, unsolved_all, iters_max, refac_all = float(tot[0]), float(tot[1]), float(mx[2]), float(tot[3])
    else:
        iters_sum_all, unsolved_all, iters_max, refac_all = float(stats[0]), float(stats[1]), float(stats[2]), float(stats[3])

    if rank == 0:
        N, keep, form = batch.N, batch.keep, batch.formulation
        paths_per_s = world * B * args.steps / elapsed
        abytes, b_iter = algorithmic_bytes(form, N, keep, float(info["iters"].sum()), B)
        achieved = abytes / (kernel_ms * 1e-3) / 1e9  # GB/s, this rank's kernel (per-launch duration: launches of different streams overlap)
        serial_achieved = None if serial_kernel_ms is None else abytes / (serial_kernel_ms * 1e-3) / 1e9
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
                       "batch_per_gpu": B, "points": N, "formulation": "KP", "parallelism": f"batch-split x{world}",
                       "rank_time_max_over_mean": rank_time_max_over_mean,
                       "streams_per_gpu": S},
            # the same K steps issued strictly serially on one stream (every step waits for the previous step's last straggler)
            "serial": None if serial_elapsed is None else {
                "value": world * B * args.steps / serial_elapsed, "ms_per_step": serial_elapsed / args.steps * 1e3,
                "kernel_ms": serial_kernel_ms, "roofline_achieved": serial_achieved, "roofline_frac": serial_achieved / 8000.0},
            "admm": {"iters_mean": iters_sum_all / (world * B), "iters_max": iters_max, "unsolved": int(unsolved_all),
                     "refactorisations": int(refac_all), "path_iters_per_s": iters_sum_all * args.steps / elapsed,
                     # distribution on rank 0's shard
                     "iters_min": int(info["iters"].min()), "iters_median": float(np.median(info["iters"])),
                     "iters_p95": float(np.percentile(info["iters"], 95)),
                     "schedule": "termination check every 25 it, adaptive rho every 100 it (by iteration count), max_iter 4000"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "frac_of_measured_copy_ceiling": achieved / 6290.0,  # MI355X_MICROARCH.md: 6.29 TB/s measured copy
                         "traffic": traffic, "kernel": "po::solve_kernel_fast<KP,SPL=4,NT=64,two-level,uniform-row-classes> (the general-variant launch that follows it only picks up deferred paths: none on this workload)", "kernel_ms": kernel_ms,
                         "kernel_ms_all_launches": kernel_ms_all,  # warm-up launches included: the figure rocprofv3's per-kernel average corresponds to
                         "algorithmic_bytes_per_path_iter": b_iter,
                         "note": "algorithmic bytes of one solve call / that call's duration (hipEvents on its stream around the equilibration kernel, the uniform-variant "
                                 "launch that does the work and the general-variant launch that follows it: rocprofv3 lists them separately); launches of the "
                                 f"{S} streams overlap, so one launch's duration is longer than ms_per_step; the state is LDS-resident, so this is not HBM traffic",
                         "aggregate_achieved": abytes * args.steps / elapsed / 1e9, "aggregate_frac": abytes * args.steps / elapsed / 1e9 / 8000.0,
                         # the bound that actually limits the kernel: fp64 VALU issue, v_fma_f64 = 8 cycles per
                         # wave-instruction measured (tools/ubench) -> 1024 SIMDs x 2.4 GHz x 64 lanes x 2 / 8 = 39.3 TFLOP/s
                         "secondary": {"bound": "fp64_valu", "unit": "TFLOP/s", "peak": 39.3,
                                       "achieved": float(info["iters"].sum()) * 1.0e5 * args.steps / elapsed / 1e12,
                                       "note": "0.1 MFLOP per path-iteration (SURVEY.md §8a10)"}},
        }
        if world == 1 and not args.no_stages:
            out["stages"] = stage_legs(torch, binding, synth, engs[0], streams[0], dbatch, B, args.cpu_sample > 0)
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
            # the reference's OWN solver classes (oracle/_ref/libpo_ref.so = src/solver/*.cpp compiled where they lie: dense-scratch
            # setHessianMatrix / setConstraintMatrix + getOptimizedPath; OSQP itself stood in by the oracle's ADMM), when that library was built
            try:
                from oracle import ref_py

                if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libpo_ref.so")):
                    nr = min(384, ns)  # ~5 s
                    rp = oracle_py.default_params()
                    inst = lambda b_: dict(ref_x=batch.ref_x[b_], ref_y=batch.ref_y[b_], ref_z=batch.ref_z[b_], ref_k=batch.ref_k[b_], ref_s=batch.ref_s[b_],
                                           bounds=batch.bounds[b_], x0=batch.x0[b_], goal_z=batch.goal_z[b_])
                    ref_py.solve("KP", inst(0), rp)
                    r0 = time.perf_counter()
                    for b_ in range(nr):
                        ref_py.solve("KP", inst(b_), rp)
                    r1 = time.perf_counter()
                    out["cpu_baseline_reference_code"] = {"value": nr / (r1 - r0), "unit": "paths/s", "cores": 1, "kind": "reference",
                                                          "sample": f"first {nr} paths through the reference's OsqpSolver::create(\"KP\")->solve() compiled from its own sources "
                                                                    f"(18.9 MB dense scratch per solve) with the oracle's ADMM in place of OSQP, {r1 - r0:.1f} s"}
            except Exception as e:
                out["cpu_baseline_reference_code"] = {"error": repr(e)}
            # the same sample on every host core, one path slice per process (the reference itself is single-threaded)
            try:
                import multiprocessing as mp

                nproc = max(1, min(os.cpu_count() or 1, 64))
                nmt = min(B, 48 * nproc)  # ~48 paths per core
                sample = batch.slice(0, nmt)
                ns = nmt
                parts = [(lo_, min(ns, lo_ + -(-ns // nproc))) for lo_ in range(0, ns, -(-ns // nproc))]
                # spawn (not fork): the parent holds a live HIP context; workers import numpy + the oracle only
                with mp.get_context("spawn").Pool(len(parts)) as pool:
                    pool.map_async(_cpu_slice, [(sample.slice(a, a + 2), None) for a, _ in parts], chunksize=1).get(timeout=180)  # warm
                    m0 = time.perf_counter()
                    pool.map_async(_cpu_slice, [(sample.slice(a, b_), None) for a, b_ in parts], chunksize=1).get(timeout=180)
                    m1 = time.perf_counter()
                out["cpu_baseline_all_cores"] = {"value": ns / (m1 - m0), "unit": "paths/s", "cores": len(parts), "kind": "port",
                                                 "sample": f"first {ns} paths of the batch split over {len(parts)} processes "
                                                           f"(one oracle instance per host core, capped at 64), {m1 - m0:.2f} s"}
            except Exception as e:  # never let the optional leg break the bench line
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
