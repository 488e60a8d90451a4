#!/usr/bin/env python
"""bench.py — QP paths/sec of the batched solve (BASELINE.json metric) on N GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` MEANS N GPUs: started without torchrun and N > 1, the script re-runs itself under `torch.distributed.run --nproc-per-node N`; started
under torchrun with WORLD_SIZE != N it exits non-zero; the printed line is refused unless N ranks answered.

A "step" is one pass of the hot path over one batch of synthetic planning instances already resident in HBM, at the HEADLINE setting:
equilibration + assembly + 25 ADMM iterations (the warm start) + the semismooth-Newton refinement of every path (factorisation, solve, line
search per step; 13.9 steps on average on config 3), certified at OSQP's termination test with eps 1e-8 — hence also at the metric's 1e-4 —
+ output map.  The OSQP-faithful solve ("ADMM to eps 1e-4", what the metric's words describe) is timed beside it: `value_osqp_faithful`.
  N=1   BASELINE config 3: B=4096 paths, N=200 points, KP, per-path random obstacle clearances
  N>1   BASELINE config 4: the same generator, 4096 paths per GPU (contiguous shard of path ids), no data-path collective
        (paths are independent); RCCL carries the barrier, the reduction of a few statistics and (--gather) the result gather.
`value` is what SURVEY.md §8d prescribes: K single-batch solves issued one after the other on ONE stream, timed between two
barriers (max over ranks), at the HEADLINE setting — the fastest setting at which EVERY path of the batch is within 1e-4 m (lateral-offset
RMS) of its exact optimum AND satisfies OSQP's termination test at eps 1e-4 (`config.accuracy` carries the count, measured
in the run against tests/golden/tight_full_c3.npz).  The OSQP-faithful default (no extension) is timed beside it as `osqp_default`.
`single_batch` holds the per-step hipEvent figures (median of K).  Everything bulky (other configs, stage legs, the table of settings) goes
to the details file (--details, default bench_details.json next to this script when writable); the printed line stays short enough for
the driver's record.
Rank 0 prints ONE JSON line.  `roofline` is the fp64 VALU issue roof (the solver state is LDS/register resident: HBM is not the
binding resource); SURVEY §8d's algorithmic-bytes contract figure is kept as `roofline.algorithmic_hbm_equivalent`.
`parity` is computed in the run: device vs the CPU oracle at identical settings on the CPU-baseline sample, and the lateral-offset
RMS against the exact optimum (tests/golden/tight_c3.npz) with and without the opt-in polish step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak: 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (half the 157.3 TF fp32 vector peak of MI355X_MICROARCH.md)
FP64_VALU_MEASURED_CEILING = {"1_wave_per_simd": 36.0, "2_waves": 52.0, "4_waves": 59.0}  # tools/ubench/fp64_rate.hip on this chip


def algorithmic_bytes(form, N, keep, iters_sum, B):
    C = (N + keep - 2) // keep if form != 2 else N - 1
    vals = {0: 128 * N + 13 * C + 9, 1: 139 * N + 32 * C + 9, 2: 119 * N - 20}[form]
    return 8.0 * vals * iters_sum + 8.0 * (18 * N + 8) * B, 8.0 * vals


def _load_json(path):
    try:
        return json.load(open(path))
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------------------
# GPU legs
# ---------------------------------------------------------------------------------------------------------------------------
def time_serial(torch, eng, stream, db, steps, barrier):
    """`steps` complete solves of the batch one after the other on one stream.  Returns (wall seconds between the barriers, [ms per step])."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier()
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        eng.solve_batch_device(db)
        e1.record(stream)
    barrier()
    t1 = time.perf_counter()
    return t1 - t0, [a.elapsed_time(b) for a, b in evs]


def time_pipelined(torch, engs, streams, dbs, steps, barrier):
    """step k on handle k % S: the stragglers of one batch drain while the next batch fills the CUs."""
    S = len(engs)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier()
    t0 = time.perf_counter()
    for k, (e0, e1) in enumerate(evs):
        i = k % S
        e0.record(streams[i])
        engs[i].solve_batch_device(dbs[i])
        e1.record(streams[i])
    barrier()
    t1 = time.perf_counter()
    return t1 - t0, [a.elapsed_time(b) for a, b in evs]


def config_legs(torch, binding, synth, dev, stream, steps=3):
    """The other BASELINE configs as quick legs: c1 (single path N=80), c2 (B=1024, N=120), c5 (KPC, B=4096, N=400) and
    the K formulation on config-3 corridors.  Single-batch figures, median of `steps` after one warm-up."""
    out = {}
    for name, cfg, kw in (("c1", 1, {}), ("c2", 2, {}), ("c5", 5, {}), ("k", 3, {"formulation": 2})):
        b = synth.make_batch(cfg, **kw)
        db = binding.DeviceBatch(b, device=dev)
        eng = binding.Engine(torch.cuda.current_device())
        eng.set_stream(stream.cuda_stream)
        eng.solve_batch_device(db)
        torch.cuda.synchronize()
        _, ms = time_serial(torch, eng, stream, db, steps, torch.cuda.synchronize)
        info = db.info_numpy()
        med = float(np.median(ms))
        out[name] = {"workload": f"BASELINE config {cfg}" + (" corridors, K formulation" if kw else "") + f": {['KP', 'KPC', 'K'][b.formulation]}, B={b.B}, N={b.N}",
                     "ms": med, "paths_per_s": b.B / (med * 1e-3), "path_iters_per_s": float(info["iters"].sum()) / (med * 1e-3),
                     "iters_mean": float(info["iters"].mean()), "iters_max": int(info["iters"].max()), "unsolved": int((info["status"] != 1).sum())}
        eng.close()
        # the same leg at the setting `value` is quoted at (HEADLINE: every path certified at refine_eps; tests/test_accuracy_full.py holds it to 1e-4 m on these shapes),
        # and with OSQP's adaptive-rho tolerance at 2 instead of 5 (the better choice for every shape but config 3)
        ref = {}
        gold, _gn = gold_for(b)
        dbx = binding.DeviceBatch(b, device=dev, want_x=True) if gold is not None else db
        # (round 6) KPC and K also with the equality rows' Newton penalty at 1e5 instead of the default 1e4 — a caller-side knob (po_params.refine_newton_rho_eq), NOT the one
        # setting `value` and the compliant_* figures are quoted at: 9 % fewer Newton steps on these two formulations (config 5 17.5 -> 15.9 ms, K 4.47 -> 4.11 ms), 4 % MORE
        # on KP (oracle, config 3: 14.3 -> 14.8 steps), and a rounding floor of rho_eq (a.x - b) ten times closer to refine_eps under extreme slack weights (DESIGN.md sections 11, 12)
        settings = (("headline_setting", dict(HEADLINE["params"])),) + ((("headline_setting_rho_eq_1e5", dict(HEADLINE["params"], refine_newton_rho_eq=1e5)),) if b.formulation != 0 else ())
        for tag, mut in settings:
            p = binding.default_params()
            if not hasattr(p, "refine_newton_rho"):
                break
            for k_, v_ in mut.items():
                setattr(p, k_, v_)
            eng = binding.Engine(torch.cuda.current_device(), p)
            eng.set_stream(stream.cuda_stream)
            eng.solve_batch_device(dbx)
            torch.cuda.synchronize()
            _, ms = time_serial(torch, eng, stream, dbx, steps, torch.cuda.synchronize)
            info = dbx.info_numpy()
            med = float(np.median(ms))
            ref[tag] = {"ms": med, "paths_per_s": b.B / (med * 1e-3), "iters_mean": float(info["iters"].mean()), "iters_max": int(info["iters"].max()),
                        "unsolved": int((info["status"] != 1).sum()), "certified": int((info["status_refine"] == 1).sum()), "r_prim_max": float(info["r_prim"].max()), "r_dual_max": float(info["r_dual"].max())}
            if gold is not None:
                acc = accuracy_of(dbx.out_x.cpu().numpy(), info, b, gold)
                ref[tag].update({"n_gt_1e-4_m": acc["n_gt_1e-4_m"], "max_m": acc["max_m"], "paths_checked": acc["paths"]})
            eng.close()
        if ref:
            out[name]["with_refinement"] = ref
            hs = ref["headline_setting"]
            # fp64-VALU roofline of this config at the headline setting (VERDICT r5 missing 4): the flop of one solve from the newest committed rocprofv3 PMC summary of THIS
            # config (tools/profile_cfg.sh -> profiles/r6*/<config>/pmc_summary.json: the work of a solve does not depend on the run), the time measured here
            prof = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r6*", name, "pmc_summary.json")))
            pj = _load_json(prof[-1]) if prof else None
            if pj and pj.get("fp64_flop_per_solve_all_kernels"):
                rn = pj.get("roofline_newton_kernels") or {}
                tf = pj["fp64_flop_per_solve_all_kernels"] / (hs["ms"] * 1e-3) / 1e12
                out[name]["roofline"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
                                         "fp64_flop_per_solve": pj["fp64_flop_per_solve_all_kernels"], "flop_source": os.path.relpath(prof[-1], ROOT) + " (rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, all kernels of a solve)",
                                         "newton_kernels": {"names": rn.get("kernels"), "fp64_flop_per_solve": rn.get("fp64_flop_per_solve"), "ms_in_the_profiled_run": rn.get("ms_per_solve"), "frac_in_the_profiled_run": rn.get("frac"),
                                                            "hbm_bytes_per_solve": rn.get("hbm_bytes_per_solve"), "resident_time_split": pj.get("resident_time_split")}}
                out[name]["compliant_fp64_roofline_frac"] = tf / FP64_VALU_PEAK_TFLOPS
            # the compliant figures of this config for the compact line: time, certified count, paths beyond the bar — beside the plain ones
            out[name].update({"compliant_ms": hs["ms"], "compliant_paths_per_s": hs["paths_per_s"], "compliant_certified": hs["certified"], "compliant_iters_max": hs["iters_max"],
                              "compliant_n_gt_1e-4_m": hs.get("n_gt_1e-4_m"), "compliant_max_m": hs.get("max_m")})
            if "headline_setting_rho_eq_1e5" in ref:
                h5 = ref["headline_setting_rho_eq_1e5"]
                out[name]["rho_eq_1e5"] = {"ms": h5["ms"], "paths_per_s": h5["paths_per_s"], "certified": h5["certified"], "n_gt_1e-4_m": h5.get("n_gt_1e-4_m"), "max_m": h5.get("max_m"), "iters_mean": h5["iters_mean"]}
    # BASELINE config 1 as the reference itself runs it: its REAL benchmark scene (src/test/path_optimizer_benchmark.cpp; map / way points / reference outputs
    # in tests/golden/benchmark_scene.npz).  Single planning instance: a latency figure, B = 1 fills one CU of 256.
    gpath = os.path.join(ROOT, "tests", "golden", "benchmark_scene.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        rows = {}
        for tag, eps in (("eps_1e-3_reference_default", 1e-3), ("eps_1e-4", 1e-4), ("headline_setting", -1e-4)):
            p = binding.default_params(); p.eps_abs = p.eps_rel = abs(eps)
            if eps < 0:  # the path QP at the headline setting (closer to the QP's optimum than either plain setting)
                if not hasattr(p, "refine_newton_rho"):
                    continue
                for k_, v_ in HEADLINE["params"].items():
                    setattr(p, k_, v_)
            eng = binding.Engine(torch.cuda.current_device(), p)
            eng.set_map(g["distance"], float(g["resolution"]), float(g["pos"][0]), float(g["pos"][1]))
            args = (g["way_x"][None], g["way_y"][None], g["start"][None], g["goal"][None])
            states, n, ok, stage, info = eng.plan_batch(*args, N=512)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); eng.plan_batch(*args, N=512); ts.append((time.perf_counter() - t0) * 1e3)
            ref = g["path1_e3" if eps == 1e-3 else "path1_e4"]  # (the refined run is compared with the eps 1e-4 reference output: it differs by the reference's own distance to the optimum)
            rows[tag] = {"solve_ms_host_to_host": float(np.median(ts)), "ok": int(ok[0]), "states": int(n[0]), "qp_iters": int(info["iters"][0]),
                         "max_abs_diff_vs_reference_compiled_PathOptimizer": float(np.abs(states[0, :n[0]] - ref).max()) if n[0] == len(ref) else None}
            eng.close()
        r3_ = rows.get("eps_1e-3_reference_default", {})
        out["c1_real_scene"] = {"ms": r3_.get("solve_ms_host_to_host"), "qp_iters": r3_.get("qp_iters"), "max_abs_diff_vs_reference": r3_.get("max_abs_diff_vs_reference_compiled_PathOptimizer"),
                                "workload": "the reference's benchmark scene: obstacles_for_benchmark.png (495 x 497 cells at 0.2 m), 100 way points, PathOptimizer::solve "
                                            "(bSpline -> TENSION2 QP -> DP search -> post QP -> re-sampling -> bounds -> KP QP, 132 states -> collision check), B = 1, "
                                            "host pointers in and out (po_plan_batch)", **rows}
    return out


# The setting `value` is quoted at — ONE setting for every shape (round 4): a short OSQP-faithful ADMM run (to the first termination check) as the warm start, then
# the Newton refinement (po_params.refine = 2: semismooth Newton on the augmented Lagrangian with a line search on the merit (safeguarded Newton on its piecewise-linear derivative)) until OSQP's termination test holds at refine_eps.
HEADLINE = {"label": "ADMM warm start (25 it) + Newton refinement (refine = 2, refine_eps 1e-8 + final correction steps), split launches; the Newton launch sliced in two by the engine (8 steps of every path, the rest longest-expected first)",
            "params": dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2),
            "algorithm": "EXTENSION (closer to the QP's optimum than the reference's OSQP run): certified per path, po_info.status_refine"}
OSQP_DEFAULT = {"label": "eps 1e-4, OSQP defaults only (no extension)", "params": {},
                "algorithm": "OSQP-FAITHFUL (identical to the reference's algorithm: same iteration counts as the CPU restatement of OSQP)"}


def make_params(binding, kw):
    p = binding.default_params()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def e_y_of(form, N, x):
    return x[:, 1:2 * N:2] if form == 2 else x[:, 0:3 * N:3]  # K orders the state pair (e_phi, e_y)


def gold_for(batch, first_path=0):
    """Exact optima (tests/golden/tight_full_*.npz, generator make_tight_full.py) of the paths of `batch` when they are a prefix of a golden set, else None."""
    name = None
    if batch.formulation == 0 and batch.N == 200 and batch.keep == 4:
        name = "c3"
    elif batch.formulation == 0 and batch.N == 120:
        name = "c2"
    elif batch.formulation == 1 and batch.N == 400:
        name = "c5"
    elif batch.formulation == 2 and batch.N == 200:
        name = "k"
    elif batch.formulation == 0 and batch.N == 231 and batch.keep == 3:
        name = "keep3"
    path = None if name is None else os.path.join(ROOT, "tests", "golden", f"tight_full_{name}.npz")
    if path is None or not os.path.exists(path) or first_path != 0:
        return None, None
    return np.load(path)["e_y"].astype(np.float64), name


def accuracy_of(x, info, batch, gold):
    """Per-path lateral-offset RMS against the exact optimum: the accuracy clause of the metric (<= 1e-4 m per path), as counts."""
    ng = min(len(gold), len(x))
    rms = np.sqrt(np.mean((e_y_of(batch.formulation, batch.N, x[:ng]) - gold[:ng]) ** 2, axis=1))
    sr = info["status_refine"][:ng]
    return {"paths": int(ng), "n_gt_1e-4_m": int((rms > 1e-4).sum()), "max_m": float(rms.max()), "p99_m": float(np.percentile(rms, 99)), "median_m": float(np.median(rms)),
            "certified_at_refine_eps": int((sr == 1).sum()), "not_certified": int((sr == -1).sum()), "max_m_certified": float(rms[sr == 1].max()) if (sr == 1).any() else None}


def run_setting(torch, binding, stream, db_x, kw, steps=3):
    """One setting on a batch resident in HBM (db_x keeps the raw QP solution): warm-up, `steps` timed single-batch solves, outputs of the last one."""
    eng = binding.Engine(torch.cuda.current_device(), make_params(binding, kw))
    eng.set_stream(stream.cuda_stream)
    eng.solve_batch_device(db_x)
    torch.cuda.synchronize()
    _, ms = time_serial(torch, eng, stream, db_x, steps, torch.cuda.synchronize)
    info = db_x.info_numpy().copy()
    x = db_x.out_x.cpu().numpy()
    eng.close()
    med = float(np.median(ms))
    row = {"ms": med, "paths_per_s": db_x.B / (med * 1e-3), "iters_mean": float(info["iters"].mean()), "iters_max": int(info["iters"].max()),
           "unsolved": int((info["status"] != 1).sum()), "path_iters_per_s": float(info["iters"].sum()) / (med * 1e-3)}
    return row, x, info


def scaling_preview(torch, binding, synth, dev, stream, B):
    """What a second GPU will see (no multi-GPU node has been available in any round).  Two things can cost a batch split its linearity:
    (1) the tail of a launch — does it amortise on a bigger shard?  BASELINE config 4's WHOLE batch (32 768 paths) as ONE launch sequence on this device, beside the
        4 096-path shard `value` is measured on: paths/s per GPU at both shard sizes;
    (2) the host side of E concurrent engines — E handles on E host threads pinned to distinct cores, each solving its own small batch (64 paths: the GPU work is
        ~0.1 ms, what is timed is the launch path: 1 scale + 2 warm-start + the Newton launch (a pair with a sort between them on batches of >= 2048 paths), the 4-byte read-back, the status sweep) — per-call wall time for E = 1, 2, 4, 8.
        (Python threads: ctypes releases the GIL for the call; ~50 us of interpreter time per call are included.)"""
    import threading

    out = {}
    eng = binding.Engine(torch.cuda.current_device(), make_params(binding, HEADLINE["params"]))
    eng.set_stream(stream.cuda_stream)
    big = synth.make_batch(4, B=32768)
    db = binding.DeviceBatch(big, device=dev)
    time_serial(torch, eng, stream, db, 1, torch.cuda.synchronize)
    _, ms = time_serial(torch, eng, stream, db, 3, torch.cuda.synchronize)
    info = db.info_numpy()
    med = float(np.median(ms))
    out["b32768_single_launch"] = {"workload": "BASELINE config 4 (KP, N=200), all 32768 paths on ONE device, one launch sequence", "ms": med, "paths_per_s": 32768 / (med * 1e-3),
                                   "iters_mean": float(info["iters"].mean()), "iters_max": int(info["iters"].max()), "certified": int((info["status_refine"] == 1).sum()),
                                   "unsolved": int((info["status"] != 1).sum())}
    del db
    small = synth.make_batch(3, B=64)
    per_e = {}
    for E in (1, 2, 4, 8):
        cores = sorted(os.sched_getaffinity(0))[:E]
        if len(cores) < E:
            break
        engs, strs, dbs = [], [], []
        for _ in range(E):
            e_ = binding.Engine(torch.cuda.current_device(), make_params(binding, HEADLINE["params"]))
            st_ = torch.cuda.Stream(device=dev)
            e_.set_stream(st_.cuda_stream)
            engs.append(e_); strs.append(st_); dbs.append(binding.DeviceBatch(small, device=dev))
        res = [None] * E
        gate = threading.Barrier(E)

        def work(k):
            try:
                os.sched_setaffinity(threading.get_native_id(), {cores[k]})
            except OSError:
                pass
            for _ in range(5):
                engs[k].solve_batch_device(dbs[k]); strs[k].synchronize()
            gate.wait()
            ts = []
            for _ in range(40):
                t0 = time.perf_counter(); engs[k].solve_batch_device(dbs[k]); strs[k].synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            res[k] = float(np.median(ts))

        th = [threading.Thread(target=work, args=(k,)) for k in range(E)]
        [t.start() for t in th]; [t.join() for t in th]
        per_e[str(E)] = {"per_call_ms_median_over_engines": float(np.median(res)), "per_call_ms_max": float(np.max(res))}
        [e_.close() for e_ in engs]
    out["engines_on_pinned_threads"] = {"batch_per_engine": 64, "calls": 40, "per_call_ms": per_e,
                                        "note": "one handle + stream + host thread per engine, threads pinned to distinct cores, all on device 0, in THIS process (the HIP runtime "
                                                "was started by torch with its default pool of 4 hardware queues); flat in E = the launch paths of concurrent engines do not serialise"}
    eng.close()
    # (round 6) WHERE the growth with E comes from: the same leg (a) with E PROCESSES — a runtime and a hardware-queue pool each — and (b) with E threads in a
    # fresh process whose runtime starts with GPU_MAX_HW_QUEUES = 16 (what libpo_hip.so sets before its first HIP call when the caller has not set the variable; a process
    # that imported torch first keeps the default 4).  Flat with processes and with the larger pool = the streams of one process share hardware queues, not the device.
    try:
        import subprocess

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import engines_procs as EP

        rows = {}
        for E in (1, 4, 8):
            r = EP.procs(E, "chain2")
            rows[str(E)] = float(np.median(r)) if r else None
        out["engines_in_processes"] = {"batch_per_engine": 64, "calls": EP.CALLS, "per_call_ms": rows, "note": "one PROCESS per engine (own HIP runtime and hardware queues), all on device 0"}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "engines_procs.py"), "--threads", "chain2"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
        line = [l for l in r.stdout.splitlines() if l.startswith("THREADS ")]
        out["engines_on_threads_hw_queues_16"] = {"batch_per_engine": 64, "calls": EP.CALLS, "per_call_ms": json.loads(line[0][8:]) if line else None,
                                                  "note": "E threads in ONE fresh process started with GPU_MAX_HW_QUEUES=16"}
    except Exception as exc:  # a secondary leg: never costs the bench its line
        out["engines_in_processes"] = {"error": repr(exc)[:200]}
    return out


def settings_table(torch, binding, batch, dev, stream, gold):
    """The table of settings on the full batch (details file): time, iterations and the accuracy counts of each."""
    db = binding.DeviceBatch(batch, device=dev, want_x=True)
    rows = []
    for label, kw in (("eps 1e-4 (OSQP defaults only)", {}),
                      ("eps 1e-4 + polish (OSQP: 1 pass)", dict(polish=1)),
                      (HEADLINE["label"], HEADLINE["params"]),
                      ("Newton refinement, fallback launch always issued: fully asynchronous (refine_chain = 3)", dict(HEADLINE["params"], refine_chain=3)),
                      ("Newton refinement, entered at 1e3 x eps (refine_rounds = 4)", dict(HEADLINE["params"], refine_rounds=4)),
                      ("Newton refinement, entered at 1e2 x eps (refine_rounds = 3)", dict(HEADLINE["params"], refine_rounds=3)),
                      ("Newton refinement, without the final correction steps", dict(HEADLINE["params"], refine_newton_final=0)),
                      ("Newton refinement, line search to 1e-4 and ONE correction step (the defaults profiles/r4b was taken with)", dict(HEADLINE["params"], refine_ls_tol=1e-4, refine_newton_final=1)),
                      ("Newton refinement, equality penalty fixed (refine_newton_rho_eq_max = 0)", dict(HEADLINE["params"], refine_newton_rho_eq_max=0.0)),
                      ("Newton refinement, refine_eps 3e-9", dict(HEADLINE["params"], refine_eps=3e-9)),
                      ("Newton refinement from the point eps 1e-4 stops at (refine_rounds = 1)", dict(HEADLINE["params"], refine_rounds=1)),
                      ("eps 1e-3 (OSQP's own default, what the reference runs) + Newton refinement", dict(HEADLINE["params"], eps_abs=1e-3, eps_rel=1e-3)),
                      ("eps 1e-5, max_iter 20000", dict(eps_abs=1e-5, eps_rel=1e-5, max_iter=20000)),
                      ("eps 1e-6, max_iter 20000", dict(eps_abs=1e-6, eps_rel=1e-6, max_iter=20000))):
        row, x, info = run_setting(torch, binding, stream, db, kw)
        row = {"setting": label, **row}
        if gold is not None:
            row["e_y_rms_vs_exact_optimum"] = accuracy_of(x, info, batch, gold)
        rows.append(row)
    return rows


def stage_legs_gpu(torch, binding, synth, eng, stream, dbatch, B):
    """Stages immediately before / after the QP on the same device (reported beside the headline, never part of `value`)."""
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
    from path_optimizer_amd.abi import INFO_BYTES, INFO_DTYPE
    sm = {}
    sm_inputs = {}
    for name, kind, npts in (("tension2", 0, 100), ("tension", 1, 100), ("post", 2, 60)):
        si = synth.make_smooth_inputs(30, 256, P=npts, kind=kind)
        sm_inputs[name] = (kind, si)
        tt = {k: torch.from_numpy(np.ascontiguousarray(np.concatenate([v] * reps, axis=0)[:B])).cuda() for k, v in si.items() if v is not None}
        so = dict(x=torch.zeros((B, npts), dtype=torch.float64, device="cuda"), y=torch.zeros((B, npts), dtype=torch.float64, device="cuda"),
                  s=torch.zeros((B, npts), dtype=torch.float64, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda"))
        ms_s = timed(lambda: eng.smooth_batch_device(kind, tt, so))
        inf = so["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
        n_q, m_q = binding.smooth_dims(kind, npts)
        sm[name] = {"ms": ms_s, "qps_per_s": B / (ms_s * 1e-3), "qp_iters_per_s": float(inf["iters"].sum()) / (ms_s * 1e-3), "iters_mean": float(inf["iters"].mean()),
                    "iters_max": int(inf["iters"].max()), "unsolved": int((inf["status"] != 1).sum()), "points": npts, "n": n_q, "m": m_q}
        if kind == 1:
            # TENSION reads the clearance of every way point from the map; the 60-disc map of the stage legs puts way points inside discs (unsolved at max_iter above).
            # Beside it: the map of tests/test_smooth.py and tools/smooth_bench.py, at the reference's own OSQP eps (1e-3, its default) and at the bench's.
            for eps_s in (1e-3, 1e-4):
                ps = binding.default_params(); ps.eps_abs = ps.eps_rel = eps_s
                e2 = binding.Engine(torch.cuda.current_device(), ps); e2.set_stream(stream.cuda_stream)
                e2.set_map(*synth.make_distance_map(3)[:4])
                ms2 = timed(lambda: e2.smooth_batch_device(kind, tt, so))
                inf2 = so["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
                sm[name]["synthetic_map_eps_%g" % eps_s] = {"ms": ms2, "qps_per_s": B / (ms2 * 1e-3), "iters_mean": float(inf2["iters"].mean()), "iters_max": int(inf2["iters"].max()),
                                                            "unsolved": int((inf2["status"] != 1).sum())}
                e2.close()
    st["smoothing_qps"] = sm
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
    perm = np.random.default_rng(5).permutation(64 * rs)[:B] % 64  # shuffled replication: a period-64 pattern would pin scenes to XCDs
    tp = {k: torch.from_numpy(np.ascontiguousarray(scn[k][perm])).cuda() for k in ("way_x", "way_y", "start", "goal")}
    Np = 320
    po_ = dict(states=torch.zeros((B, Np, 5), dtype=torch.float64, device="cuda"), n_states=torch.zeros(B, dtype=torch.int32, device="cuda"),
               ok=torch.zeros(B, dtype=torch.int32, device="cuda"), stage=torch.zeros(B, dtype=torch.int32, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda"))
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
                                       "re-sampling (0.15..0.3 m) -> bounds -> KP QP -> collision check; engine at the HEADLINE setting (the path QP runs the chained refinement rounds)"}
    p3 = binding.default_params(); p3.eps_abs = p3.eps_rel = 1e-3  # the reference's own stopping point: it never touches OSQP's eps (default 1e-3)
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
    # the same 4096 instances with the path QP's refinement phase (po_params.refine, include/po_hip.h): every path ends at residuals of 1e-6 or keeps its plain point
    for tag, kw in (("headline_setting", dict(HEADLINE["params"])), ("headline_setting_adapt_tol_2", dict(HEADLINE["params"], adapt_tol=2.0))):
        p5 = binding.default_params()
        for k_, v_ in kw.items():
            setattr(p5, k_, v_)
        e5 = binding.Engine(torch.cuda.current_device(), p5); e5.set_map(*scn["map"])
        e5.plan_batch_device(tp, po_, Np, way_len); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            e5.plan_batch_device(tp, po_, Np, way_len)
        torch.cuda.synchronize()
        ms_p5 = (time.perf_counter() - t0) / 3 * 1e3
        pinf5 = po_["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
        st["full_pipeline"][tag] = {"ms": ms_p5, "instances_per_s": B / (ms_p5 * 1e-3), "ok_frac": float(po_["ok"].double().mean().item()),
                                    "qp_iters_mean": float(pinf5["iters"].mean()), "qp_iters_max": int(pinf5["iters"].max())}
        e5.close()
    eng.set_map(d, res, px, py)
    ctx = dict(map=(d, res, px, py), scn=scn, P=P, keys=keys, sm_inputs=sm_inputs, spn=spn, length=length, start=start, Lc=Lc,
               states64=dbatch.out_states[:64].cpu().numpy(), info64=dbatch.info_numpy()[:64])
    return st, ctx


def stage_legs_cpu(st, ctx):
    """The same stages on one host core through the oracle (test infrastructure; CPU baseline legs only)."""
    from oracle import oracle_py

    d, res, px, py = ctx["map"]
    scn, P, keys = ctx["scn"], ctx["P"], ctx["keys"]
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
    oracle_py.postcheck_batch(p, m, ctx["states64"], ctx["info64"])
    c2 = time.perf_counter()
    st["cpu_port"] = {"bounds_paths_per_s": 64 / (c1 - c0), "post_check_paths_per_s": 64 / (c2 - c1), "cores": 1, "sample": "64 paths each, oracle (C)"}
    for name, (kind, si) in ctx["sm_inputs"].items():
        c3 = time.perf_counter()
        oracle_py.smooth_batch(kind, p, {k: (None if v is None else v[:64]) for k, v in si.items()}, m_map=m)
        st["cpu_port"][f"smoothing_{name}_qps_per_s"] = 64 / (time.perf_counter() - c3)
    spn, length, start, Lc = ctx["spn"], ctx["length"], ctx["start"], ctx["Lc"]
    c4 = time.perf_counter()
    for b in range(64):
        oracle_py.dp_search(p, m, spn["knot_s"][b], spn["knot_x"][b], spn["knot_y"][b], length[b], start[b], cap=Lc)
    c5 = time.perf_counter()
    for b in range(64):
        oracle_py.resample(p, spn["knot_s"][b], spn["knot_x"][b], spn["knot_y"][b], length[b], 0.15, 0.3, cap=256)
    st["cpu_port"]["full_pipeline_instances_per_s"] = cpu_pipeline
    st["cpu_port"]["dp_search_paths_per_s"] = 64 / (c5 - c4)
    st["cpu_port"]["resample_paths_per_s"] = 64 / (time.perf_counter() - c5)


def live_traffic(batch_paths, timeout_s=150):
    """HBM traffic, VALU instruction count and fp64 flop of the solve kernel measured IN THIS RUN: rocprofv3 PMC child passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU,
    then the four SQ_INSTS_VALU_*_F64 counters — separate passes, counters only, as MI355X_MICROARCH.md prescribes) over `bench.py --traffic-child` (3 solves of the
    same batch at the headline setting).  Returns None when rocprofv3 is missing, fails or times out (the bench line then keeps the committed fallback values)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    vals = {}
    vals_nk = {}  # the same counters for newton_kernel alone (the dominant kernel)

    def one_pass(ctrs):
        nk = {}
        d = tempfile.mkdtemp(prefix="po_pmc_", dir="/tmp")
        cmd = [exe, "--output-format", "csv", "--pmc", *ctrs, "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--traffic-child", "--batch", str(batch_paths)]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
        tot, disp = {c: 0.0 for c in ctrs}, set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                # every kernel of one solve at the headline setting: the warm-start launches (solve_kernel_fast, uniform + general variant), newton_kernel and
                # newton_fallback_kernel (scale_kernel / finalize_status_kernel: negligible, counted too).  Per SOLVE = the sum over the dispatches / the 3 solves of the child.
                kn = row.get("Kernel_Name", "")
                if ("solve_kernel_fast" in kn or "newton_" in kn or "scale_kernel" in kn or "finalize_status" in kn) and row.get("Counter_Name") in tot:
                    tot[row["Counter_Name"]] += float(row["Counter_Value"]); disp.add(row["Dispatch_Id"])
                    if "newton_kernel" in kn:
                        nk[row["Counter_Name"]] = nk.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        shutil.rmtree(d, ignore_errors=True)
        if r.returncode != 0 or not disp:
            return None
        for c in ctrs:
            vals_nk[c] = nk.get(c, 0.0) / 3.0
        return {c: v / 3.0 for c, v in tot.items()}

    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            got = one_pass([ctr])
            if got is None:
                return None
            vals[ctr] = got[ctr] * (1.0 if ctr.startswith("SQ_") else 1024.0)  # FETCH / WRITE in KB units
        res = {"hbm_bytes_per_launch": 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], "fetch_bytes_raw": vals["FETCH_SIZE"], "write_bytes": vals["WRITE_SIZE"],
               "valu_wave_instr_per_launch": vals["SQ_INSTS_VALU"],
               "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)",
               "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes over 3 solves of the same batch, every kernel of a solve summed"}
        try:
            f64 = one_pass(["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"])
        except Exception:
            f64 = None
        if f64 is not None and f64["SQ_INSTS_VALU_FMA_F64"] > 0:
            res["fp64_wave_instr_per_launch"] = sum(f64.values())
            res["fp64_flop_per_launch"] = 64.0 * (f64["SQ_INSTS_VALU_ADD_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] + 2.0 * f64["SQ_INSTS_VALU_FMA_F64"] + f64["SQ_INSTS_VALU_TRANS_F64"])
            res["fp64_counters"] = f64
            g = vals_nk
            if g.get("SQ_INSTS_VALU_FMA_F64", 0) > 0:
                res["newton_kernel_fp64_flop_per_launch"] = 64.0 * (g["SQ_INSTS_VALU_ADD_F64"] + g["SQ_INSTS_VALU_MUL_F64"] + 2.0 * g["SQ_INSTS_VALU_FMA_F64"] + g["SQ_INSTS_VALU_TRANS_F64"])
                res["newton_kernel_valu_wave_instr_per_launch"] = g.get("SQ_INSTS_VALU")
        return res
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------------------
# CPU legs (after the GPU legs: `gpu_done_s` marks the boundary for the driver's gpu_busy sampling)
# ---------------------------------------------------------------------------------------------------------------------------
def _cpu_slice(arg):
    """Worker of the all-cores CPU baseline leg (oracle, test infrastructure)."""
    from oracle import oracle_py

    batch, native, kw = arg
    if native:
        oracle_py.use_native()
    p = oracle_py.device_equivalent_params()
    for k, v in kw.items():
        setattr(p, k, v)
    oracle_py.solve_batch(batch, p, want_x=False)
    return batch.B


def _vs_oracle(dx, dinfo, oxs, oinfo, eps=1e-4):
    n_ = min(len(dx), len(oxs))
    dx, dinfo, oxs, oinfo = dx[:n_], dinfo[:n_], oxs[:n_], oinfo[:n_]
    same = dinfo["iters"] == oinfo["iters"]
    err = np.abs(dx - oxs).max(axis=1)
    di = np.abs(dinfo["iters"].astype(np.int64) - oinfo["iters"].astype(np.int64))
    return {"paths": int(n_), "status_equal": int((dinfo["status"] == oinfo["status"]).sum()), "iteration_count_equal": int(same.sum()),
            "iteration_count_within_1": int((di <= 1).sum()), "iteration_count_within_2": int((di <= 2).sum()), "iteration_count_max_difference": int(di.max()),
            "iters_mean_device": float(dinfo["iters"].mean()), "iters_mean_oracle": float(oinfo["iters"].mean()),
            "max_abs_dx_equal_count": float(err[same].max()) if same.any() else None,
            "max_abs_dx_other": float(err[~same].max()) if (~same).any() else 0.0,
            "other_within_10_eps": int((err[~same] <= 10 * eps).sum()) if (~same).any() else 0,
            "status_refine_equal": int((dinfo["status_refine"] == oinfo["status_refine"]).sum())}


def cpu_legs(out, details, batch, dev_samples, cpu_sample):
    """CPU legs (after the GPU legs).  dev_samples: {"headline": (x, info), "osqp_default": (x, info)} = the device's raw solutions of the first paths."""
    from oracle import oracle_py  # CPU baseline / checker legs only

    native = oracle_py.use_native()  # gcc -O3 -march=native build of the same source on THIS box; False: portable build
    B = batch.B
    ns = min(cpu_sample, B)
    sample = batch.slice(0, ns)
    build = "gcc -O3" + (" -march=native, built on this box" if native else ", portable x86-64 build")

    def opar(kw):
        p = oracle_py.device_equivalent_params()
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    oracle_py.solve_batch(sample.slice(0, 2), opar({}))  # warm the ordering cache
    # (i) the SAME setting as `value` (the oracle implements the refinement and its rounds too): the cpu_baseline of the line
    c0 = time.perf_counter()
    _, hinfo, hxs = oracle_py.solve_batch(sample, opar(HEADLINE["params"]), want_x=True)
    c1 = time.perf_counter()
    out["cpu_baseline"] = {"value": ns / (c1 - c0), "unit": "paths/s", "cores": 1, "kind": "port",
                           "sample": f"first {ns} paths of the same batch at the same setting as `value`, oracle (OSQP-style ADMM + the same refinement, sparse LDL', {build}), "
                                     f"{c1 - c0:.1f} s, mean iters {float(hinfo['iters'].mean()):.1f}",
                           "host_cpus": os.cpu_count()}
    # (ii) the OSQP-faithful default on the same sample
    nd = ns  # (every path of the sample: the OSQP-faithful leg is the one that is "identical to the reference's algorithm")
    c0 = time.perf_counter()
    _, dinfo_o, dxs_o = oracle_py.solve_batch(sample.slice(0, nd), opar({}), want_x=True)
    c1 = time.perf_counter()
    out["osqp_default"]["cpu_baseline"] = {"value": nd / (c1 - c0), "unit": "paths/s", "cores": 1, "kind": "port", "sample": f"first {nd} paths, {c1 - c0:.1f} s, mean iters {float(dinfo_o['iters'].mean()):.1f}"}
    # ---- parity: device vs oracle at identical settings, inside `config` so that the driver's parsed record keeps it ----
    if dev_samples.get("osqp_default") is not None:
        out["config"]["device_vs_oracle"] = {"osqp_default": _vs_oracle(*dev_samples["osqp_default"], dxs_o, dinfo_o)}
        if dev_samples.get("headline") is not None:
            out["config"]["device_vs_oracle"]["headline"] = _vs_oracle(*dev_samples["headline"], hxs, hinfo)
        out["config"]["device_vs_oracle"]["note"] = ("identical settings on both sides.  osqp_default (the OSQP-faithful leg): equal iteration count -> |dx| at round-off level; a residual "
                                                     "within round-off of a threshold flips one check (25 it): compared at 10 x eps, not dropped.  headline (extension): same Newton steps until "
                                                     "a decision falls inside one implementation's rounding noise (a row on its bound to the last bits in or out of the Newton matrix; the inexact "
                                                     "line search stopping one evaluation apart; a certified point's dual residual of ~1e-10 already 1e3 x below tolerance or not, which decides "
                                                     "whether a correction step follows): every fork ends at the same certified point — counts within 1 on most paths (iteration_count_within_1), "
                                                     "the points agree to <= 1e-5")
    # the reference's OWN solver classes (oracle/_ref/libpo_ref.so = src/solver/*.cpp compiled where they lie; OSQP itself stood in by the oracle's ADMM)
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
            details["cpu_baseline_reference_code"] = {"value": nr / (r1 - r0), "unit": "paths/s", "cores": 1, "kind": "reference",
                                                      "sample": f"first {nr} paths through the reference's OsqpSolver::create(\"KP\")->solve() compiled from its own sources "
                                                                f"(18.9 MB dense scratch per solve) with the oracle's ADMM in place of OSQP, {r1 - r0:.1f} s"}
    except Exception as e:
        details["cpu_baseline_reference_code"] = {"error": repr(e)}
    try:  # the same sample on every host core, one path slice per process (the reference itself is single-threaded)
        import multiprocessing as mp

        nproc = max(1, min(os.cpu_count() or 1, 64))
        nmt = min(B, 48 * nproc)
        sample = batch.slice(0, nmt)
        parts = [(lo_, min(nmt, lo_ + -(-nmt // nproc))) for lo_ in range(0, nmt, -(-nmt // nproc))]
        with mp.get_context("spawn").Pool(len(parts)) as pool:  # spawn (not fork): the parent holds a live HIP context
            pool.map_async(_cpu_slice, [(sample.slice(a, a + 2), native, HEADLINE["params"]) for a, _ in parts], chunksize=1).get(timeout=180)  # warm
            m0 = time.perf_counter()
            pool.map_async(_cpu_slice, [(sample.slice(a, b_), native, HEADLINE["params"]) for a, b_ in parts], chunksize=1).get(timeout=180)
            m1 = time.perf_counter()
        out["cpu_baseline"]["all_cores"] = {"value": nmt / (m1 - m0), "cores": len(parts),
                                            "sample": f"first {nmt} paths split over {len(parts)} processes (one oracle instance per host core, capped at 64), {m1 - m0:.2f} s"}
    except Exception as e:  # never let the optional leg break the bench line
        out["cpu_baseline"]["all_cores"] = {"error": repr(e)}


# ---------------------------------------------------------------------------------------------------------------------------
# dry run of the N>1 plumbing (no GPU: gloo, faked device times) — executed by tests/test_multi_process.py
# ---------------------------------------------------------------------------------------------------------------------------
class _DryEngine:
    """Stands in for binding.Engine + DeviceBatch where no GPU exists: produces deterministic per-path statistics so that the shard
    split, the reductions, the gather and the JSON assembly of the N>1 branch run for real."""

    def __init__(self, lo, hi, N):
        import torch

        ids = np.arange(lo, hi)
        self.iters = (25 * (1 + (ids * 2654435761 % 61))).astype(np.int32)  # multiples of 25 in [25, 1525], a function of the GLOBAL path id
        self.out_states = torch.from_numpy(np.repeat(ids.astype(np.float64)[:, None, None], N, axis=1).repeat(5, axis=2).copy())
        self.B = hi - lo

    def info_numpy(self):
        from path_optimizer_amd.abi import INFO_DTYPE

        info = np.zeros(self.B, dtype=INFO_DTYPE)
        info["status"] = 1
        info["iters"] = self.iters
        info["n_refactor"] = self.iters // 400
        return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="paths per GPU")
    ap.add_argument("--config", type=int, default=0, help="BASELINE config id (default: 3 at N=1, 4 at N>1)")
    ap.add_argument("--streams", type=int, default=3, help="handles/HIP streams of the secondary `pipelined` leg (0 = skip it)")
    ap.add_argument("--no-stages", action="store_true", help="skip the legs for the stages around the QP (SURVEY.md §8f)")
    ap.add_argument("--no-configs", action="store_true", help="skip the quick legs for BASELINE configs 1, 2, 5 and the K formulation")
    ap.add_argument("--no-parity", action="store_true", help="skip the accuracy / parity legs")
    ap.add_argument("--no-scaling-preview", action="store_true", help="skip the B = 32768 single-launch leg and the engines-on-pinned-threads leg (SURVEY §8e stand-ins)")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="paths timed on the CPU oracle (rank 0, N=1 only; 0 = no CPU legs)")
    ap.add_argument("--gather", action="store_true", help="N>1: gather every rank's states and info on rank 0 (SURVEY §8e collective 1) and report its time")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 PMC child passes that measure `roofline.traffic` in this run")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"), help="file for the bulky legs (other configs, stages, table of settings); '' = do not write")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: gloo backend, faked device times; exercises the shard split, reductions, gather and JSON "
                    "assembly of the N>1 branch (tests/test_multi_process.py)")
    args = ap.parse_args()

    # --gpus N means N GPUs.  Started WITHOUT torchrun (no WORLD_SIZE in the environment) and N > 1: re-run this very command under
    # `python -m torch.distributed.run --nproc-per-node N` (one process per GPU, rendezvous on 127.0.0.1) and hand its exit code back.  Started
    # under torchrun with a world that is not N: refuse — a line labelled n_gpus = world for a command that asked for N would be a wrong record.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1 and not args.traffic_child:
            import socket
            import subprocess

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch with --nproc-per-node {args.gpus}")

    import torch

    if args.traffic_child:  # profiled by live_traffic(): three solves of the batch at the headline setting, nothing else
        from path_optimizer_amd import binding, synth

        db = binding.DeviceBatch(synth.make_batch(3, B=args.batch))
        eng = binding.Engine(0, make_params(binding, HEADLINE["params"]))
        for _ in range(3):
            eng.solve_batch_device(db)
        torch.cuda.synchronize()
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    dev = "cpu" if dry else f"cuda:{local_rank}"
    if not dry:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))  # RCCL: barrier, 2 small reductions, optional gather

    from path_optimizer_amd import synth
    from path_optimizer_amd.shard import gather_to_root, reduce_stats, shard_range

    cfg = args.config or (3 if world == 1 else 4)
    B = args.batch
    lo, hi = shard_range(world * B, world, rank)  # weak scaling: fixed work per GPU, contiguous path ids

    def barrier():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    if dry:
        _, _, N0, _ = synth.CONFIGS[cfg]
        batch = None
        dbatch = _DryEngine(lo, hi, N0)
        barrier()
        elapsed = 0.010 * args.steps * (1.0 + 0.25 * rank)  # faked: rank r is 25 r % slower
        step_ms = [elapsed / args.steps * 1e3] * args.steps
        barrier()
        form, N, keep = 0, N0, 4
    else:
        from path_optimizer_amd import binding

        batch = synth.make_batch(cfg, B=hi - lo, first_path=lo)
        form, N, keep = batch.formulation, batch.N, batch.keep
        dbatch = binding.DeviceBatch(batch, device=dev)
        S = max(1, args.streams)
        engs, streams, dbs = [], [], []
        for i in range(S):
            e = binding.Engine(local_rank, make_params(binding, HEADLINE["params"]))
            st = torch.cuda.Stream(device=dev)  # a real (non-null) HIP stream shared by torch events and the engine
            e.set_stream(st.cuda_stream)
            engs.append(e); streams.append(st)
            dbs.append(dbatch if i == 0 else dbatch.clone_outputs())
        if args.warmup > 0:
            time_serial(torch, engs[0], streams[0], dbatch, args.warmup, barrier)
        # ---- THE timed region: exactly K single-batch solves, one after the other on one stream, barrier + synchronize on both sides ----
        elapsed, step_ms = time_serial(torch, engs[0], streams[0], dbatch, args.steps, barrier)

    phases = None
    if not dry:
        try:
            phases = engs[0].last_phase_ms()
        except Exception:
            phases = None
    info = dbatch.info_numpy()
    iters_sum_all, unsolved_all, iters_max, elapsed_max = reduce_stats(float(info["iters"].sum()), float((info["status"] != 1).sum()),
                                                                       float(info["iters"].max()), elapsed, device=None if dry else dev)
    elapsed_sum = elapsed
    refac_all = float(info["n_refactor"].sum())
    if world > 1:
        t = torch.tensor([elapsed, refac_all], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed_sum, refac_all = float(t[0]), float(t[1])
    rank_time_max_over_mean = elapsed_max / (elapsed_sum / world)  # load imbalance between the shards

    gather = None
    if world > 1 and args.gather:
        # SURVEY §8e collective (1): a single consumer of the whole batch — [B/G][N][5] states + po_info per rank -> rank 0
        info_t = torch.from_numpy(info.view(np.uint8).reshape(len(info), -1).copy()).to(dev) if dry else dbatch.out_info
        barrier()
        g0 = time.perf_counter()
        all_states = gather_to_root(dbatch.out_states, world, rank)
        all_info = gather_to_root(info_t, world, rank)
        barrier()
        g1 = time.perf_counter()
        gather = {"ms": (g1 - g0) * 1e3, "bytes_per_rank": int(dbatch.out_states.numel() * 8 + info_t.numel())}
        if rank == 0:
            from path_optimizer_amd.abi import INFO_DTYPE

            gi = all_info.cpu().numpy().view(INFO_DTYPE).reshape(-1)
            gather["paths_on_root"] = int(all_states.shape[0])
            gather["iters_sum_on_root"] = float(gi["iters"].sum())  # must equal the all-reduced sum
            if dry:
                gather["path_ids_in_order"] = bool((all_states[:, 0, 0].numpy() == np.arange(world * B)).all())

    # N > 1: which devices the ranks actually ran on (so that the first SCALE record can be checked)
    ranks_seen = None
    if world > 1:
        names = [None] * world
        me = {"rank": rank, "local_rank": local_rank, "device": "dry-run (cpu)" if dry else torch.cuda.get_device_name(local_rank), "paths": [lo, hi], "elapsed_s": elapsed}
        dist.all_gather_object(names, me)
        ranks_seen = {"backend": dist.get_backend(), "world": world, "rccl_ranks_seen": len([n for n in names if n is not None]), "ranks": names}
        if ranks_seen["rccl_ranks_seen"] != args.gpus or sorted(n["rank"] for n in names if n is not None) != list(range(args.gpus)):
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the ranks that answered are {ranks_seen}")

    out, details = None, {}
    if rank == 0:
        paths_per_s = world * B * args.steps / elapsed_max
        it_rank = float(info["iters"].sum())
        med_ms = float(np.median(step_ms))
        abytes, b_iter = algorithmic_bytes(form, N, keep, it_rank, B)
        valu = _load_json(os.path.join(ROOT, "profiles", "valu_latest.json")) or {}
        # fallback when the live PMC passes are unavailable: the fp64 flop of one solve of this very batch at the headline setting from the committed profile
        fl_solve = valu.get("fp64_flop_per_solve_headline_c3_b4096") if (cfg in (3, 4) and B == 4096) else None
        achieved_tf = None if fl_solve is None else fl_solve / (med_ms * 1e-3) / 1e12
        traffic = (_load_json(os.path.join(ROOT, "profiles", "traffic_latest.json")) or {}).get("hbm_bytes_per_launch")
        sr = info["status_refine"]
        out = {
            "metric": "QP paths/sec at N=200 pts, batch=4096; ADMM iters to 1e-4",
            "value": paths_per_s,
            # the figure that matches the metric's wording ("ADMM iters to 1e-4") and the library default: the OSQP-faithful leg (refine = 0), same batch, same K steps,
            # timed the same way further down (rank 0, N = 1; `osqp_default` has its details).  `value` itself = 25 ADMM iterations + the Newton refinement (`config.setting`)
            "value_osqp_faithful": None,
            "unit": "paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" + (" (DRY RUN: no GPU, faked device times)" if dry else ""),
            "config": {"workload": f"BASELINE config {cfg}: KP, B={B} paths/GPU x {world} GPU, N={N} points, per-path random obstacle clearances; one batch after the other on one stream",
                       "setting": HEADLINE["label"] + ": OSQP-faithful ADMM (scaling 10, check every 25) up to its first termination check as the warm start, then semismooth "
                                  "Newton on the augmented Lagrangian with a line search on the merit (safeguarded Newton on its piecewise-linear derivative) until OSQP's termination test holds at refine_eps — every path certified and within "
                                  "1e-4 m of its exact optimum (accuracy clause of the metric, see `accuracy`); the same setting on every BASELINE shape (`configs`); `osqp_default` = no extension",
                       "which_leg_is_what": {"value": HEADLINE["algorithm"], "osqp_default": OSQP_DEFAULT["algorithm"]},
                       "batch_per_gpu": B, "points": N, "formulation": "KP", "parallelism": f"batch-split x{world}",
                       "rank_time_max_over_mean": rank_time_max_over_mean},
            "single_batch": {"median_ms": med_ms, "min_ms": float(np.min(step_ms)), "max_ms": float(np.max(step_ms)), "paths_per_s": B / (med_ms * 1e-3),
                             "note": "hipEvents on the engine's stream around each step (equilibration + warm-start launches + the two sliced Newton launches and the sort between them + fallback launch + status sweep), rank 0"},
            "admm": {"iters_mean": iters_sum_all / (world * B), "iters_max": iters_max, "unsolved": int(unsolved_all),
                     "refactorisations": int(refac_all), "path_iters_per_s": iters_sum_all * args.steps / elapsed_max,
                     "iters_min": int(info["iters"].min()), "iters_median": float(np.median(info["iters"])), "iters_p95": float(np.percentile(info["iters"], 95)),
                     "certified_at_refine_eps": int((sr == 1).sum()), "not_certified": int((sr == -1).sum())},
            "roofline": {
                "bound": "fp64_valu", "unit": "TFLOP/s", "peak": FP64_VALU_PEAK_TFLOPS,
                "achieved": achieved_tf, "frac": None if achieved_tf is None else achieved_tf / FP64_VALU_PEAK_TFLOPS,
                "achieved_def": "fp64 VALU flop issued per solve / single-batch time.  Fallback (this value, when the live PMC passes are unavailable): "
                                "the fp64 flop of one solve of this batch at this setting from the committed profile (profiles/valu_latest.json) / single-batch time of this run",
                "peak_source": "MI355X fp64 vector peak = 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (half the 157.3 TF fp32 vector peak in MI355X_MICROARCH.md)",
                "kernel": "all kernels of one solve: po::solve_kernel_fast<KP,SPL=4,NT=64,two-level,uniform> (warm start) + po::newton_kernel<KP,4,64> (dominant, see `dominant_kernel`)", "kernel_ms": med_ms,
                "traffic": traffic,
                "traffic_note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, fallback value from profiles/traffic_latest.json (replaced below when the live PMC passes run)",
                "algorithmic_hbm_equivalent": {"GB_per_s": abytes / (med_ms * 1e-3) / 1e9, "frac_of_8TBps": abytes / (med_ms * 1e-3) / 1e9 / 8000.0,
                                               "algorithmic_bytes_per_path_iter": b_iter,
                                               "note": "SURVEY.md §8d contract figure; NOT a roofline: the state is LDS/register resident and never moves through HBM"}},
        }
        if phases is not None:
            out["single_batch"]["phases_ms_last_step"] = {k: phases[k] for k in ("warm_start", "newton", "fallback")}
        if ranks_seen is not None:
            out["config"]["ranks"] = ranks_seen
        if gather is not None:
            out["gather"] = gather

    # ---- secondary GPU legs (rank 0, N = 1) ----
    dev_samples = {}
    if not dry and world == 1:
        gold, gold_name = gold_for(batch, lo)
        ns = max(1, min(args.cpu_sample if args.cpu_sample > 0 else 256, B))
        full_x = binding.DeviceBatch(batch, device=dev, want_x=True)
        # the headline setting once more with the raw QP solution kept: accuracy of EVERY path of the batch `value` was measured on
        hrow, hx, hinfo = run_setting(torch, binding, streams[0], full_x, HEADLINE["params"], steps=1)
        dev_samples["headline"] = (hx[:ns].copy(), hinfo[:ns].copy())
        if gold is not None:
            out["config"]["accuracy"] = {"yardstick": f"exact optimum of every path (tests/golden/tight_full_{gold_name}.npz, KKT residuals <= 3e-14), lateral-offset RMS per path, bar 1e-4 m",
                                         **accuracy_of(hx, hinfo, batch, gold)}
            out["config"]["accuracy_clause_met"] = bool(out["config"]["accuracy"]["n_gt_1e-4_m"] == 0 and hrow["unsolved"] == 0 and len(gold) >= B)
        # the OSQP-faithful default (no extension), same batch, same K steps, timed the same way
        e0 = binding.Engine(local_rank); e0.set_stream(streams[0].cuda_stream)
        time_serial(torch, e0, streams[0], dbatch, 1, barrier)
        del_, dms = time_serial(torch, e0, streams[0], dbatch, args.steps, barrier)
        dinfo_full = dbatch.info_numpy().copy()
        e0.close()
        drow, dx, dinfo = run_setting(torch, binding, streams[0], full_x, {}, steps=1)
        dev_samples["osqp_default"] = (dx[:ns].copy(), dinfo[:ns].copy())
        out["osqp_default"] = {"setting": OSQP_DEFAULT["label"], "value": B * args.steps / del_, "ms_per_step": del_ / args.steps * 1e3, "median_ms": float(np.median(dms)),
                               "iters_mean": float(dinfo_full["iters"].mean()), "iters_max": int(dinfo_full["iters"].max()), "unsolved": int((dinfo_full["status"] != 1).sum()),
                               "path_iters_per_s": float(dinfo_full["iters"].sum()) / (float(np.median(dms)) * 1e-3)}
        out["value_osqp_faithful"] = out["osqp_default"]["value"]
        if gold is not None:
            out["osqp_default"]["accuracy"] = accuracy_of(dx, dinfo, batch, gold)
        time_serial(torch, engs[0], streams[0], dbatch, 1, barrier)  # (dbatch holds the headline outputs again for the stage legs below)
        info = dbatch.info_numpy()
        if args.streams > 1:
            # (the fully asynchronous variant of the split scheduling, refine_chain = 3: one host thread feeds all the streams, so no call may block)
            pengs = []
            for i in range(S):
                e_ = binding.Engine(local_rank, make_params(binding, dict(HEADLINE["params"], refine_chain=3)))
                e_.set_stream(streams[i].cuda_stream)
                pengs.append(e_)
            time_pipelined(torch, pengs, streams, dbs, min(args.steps, 3), barrier)
            pel, pms = time_pipelined(torch, pengs, streams, dbs, args.steps, barrier)
            [e_.close() for e_ in pengs]
            details["pipelined_3_streams" if S == 3 else f"pipelined_{S}_streams"] = {
                "paths_per_s": B * args.steps / pel, "ms_per_step": pel / args.steps * 1e3, "launch_ms_mean": float(np.mean(pms)),
                "note": f"the same K steps issued round-robin on {S} handles/streams (independent batches overlap: the stragglers of one drain under the next); NOT `value`"}
            out["pipelined_streams"] = {"streams": S, "paths_per_s": B * args.steps / pel}
        # caller-side scheduling hint po_batch_in.order = paths sorted by the PREVIOUS solve's iteration counts, longest first (identical inputs step to step: an upper bound)
        dbatch.set_order(np.argsort(-info["iters"].astype(np.int64), kind="stable"))
        time_serial(torch, engs[0], streams[0], dbatch, 1, barrier)
        hel, hms = time_serial(torch, engs[0], streams[0], dbatch, args.steps, barrier)
        dbatch.set_order(None)
        out["single_batch_with_order_hint"] = {"paths_per_s": B * args.steps / hel, "median_ms": float(np.median(hms)),
                                               "note": "po_batch_in.order = argsort(-iters of the previous solve); results bit-identical; never `value`"}
        # ---- host-pointer entry (what the drop-in's caller sees, SURVEY §8d "H2D/D2H reported separately"): po_solve_batch on the same batch, pageable caller arrays ----
        try:
            he = binding.Engine(local_rank, make_params(binding, HEADLINE["params"]))
            he.solve_batch(batch)
            hts, hph = [], []
            for _ in range(5):
                t0_ = time.perf_counter(); he.solve_batch(batch); hts.append((time.perf_counter() - t0_) * 1e3); hph.append(he.last_phase_ms())
            hmed = {k: float(np.median([q[k] for q in hph])) for k in hph[0]}
            he.close()
            # two host threads x two handles alternating on consecutive batches: the copies of one overlap the solve of the other
            import threading
            hes = [binding.Engine(local_rank, make_params(binding, HEADLINE["params"])) for _ in range(2)]
            for e_ in hes:
                e_.solve_batch(batch)
            nb_ = 6
            def _run(e_):
                for _ in range(nb_ // 2):
                    e_.solve_batch(batch)
            t0_ = time.perf_counter()
            ths = [threading.Thread(target=_run, args=(e_,)) for e_ in hes]
            [t_.start() for t_ in ths]; [t_.join() for t_ in ths]
            pipe_ms = (time.perf_counter() - t0_) * 1e3 / nb_
            [e_.close() for e_ in hes]
            out["host_to_host"] = {"ms": float(np.median(hts)), "paths_per_s": B / (float(np.median(hts)) * 1e-3), "pack_h2d_ms": hmed["pack_h2d"], "solve_ms": hmed["solve"], "d2h_ms": hmed["d2h"],
                                   "host_pack_ms": hmed["host_pack"], "host_unpack_ms": hmed["host_unpack"], "ratio_to_device_time": float(np.median(hts)) / hmed["solve"],
                                   "two_handles_alternating_ms_per_batch": pipe_ms,
                                   "note": "po_solve_batch (host pointers, pageable caller arrays: 85.5 MB in, 32.9 MB out): threaded pack into a pinned block while the slices already "
                                           "packed travel over PCIe, solve, one D2H into a pinned block, threaded unpack; never `value` (inputs resident in HBM there)"}
        except Exception as e_:
            out["host_to_host"] = {"error": repr(e_)}
        if not args.no_configs:
            details["configs"] = config_legs(torch, binding, synth, dev, streams[0])
            keep_ = ("ms", "paths_per_s", "iters_mean", "iters_max", "unsolved", "compliant_ms", "compliant_paths_per_s", "compliant_certified", "compliant_iters_max",
                     "compliant_n_gt_1e-4_m", "compliant_max_m", "compliant_fp64_roofline_frac", "rho_eq_1e5", "qp_iters", "max_abs_diff_vs_reference")
            out["configs"] = {k: {kk: v[kk] for kk in keep_ if kk in v} for k, v in details["configs"].items()}
            out["configs"]["note"] = "ms / paths_per_s: OSQP-faithful default at eps 1e-4; compliant_*: the headline setting (same as `value`) on the whole batch, against the exact optima"
        if not args.no_scaling_preview:
            try:
                details["scaling_preview"] = scaling_preview(torch, binding, synth, dev, streams[0], B)
                sp = details["scaling_preview"]
                out["scaling_preview"] = {"shard_4096_paths_per_s": out["single_batch"]["paths_per_s"], "shard_32768_paths_per_s": sp["b32768_single_launch"]["paths_per_s"],
                                          "engine_call_ms_by_E": {k: v["per_call_ms_median_over_engines"] for k, v in sp["engines_on_pinned_threads"]["per_call_ms"].items()},
                                          "engine_call_ms_by_E_processes": (sp.get("engines_in_processes") or {}).get("per_call_ms"),
                                          "engine_call_ms_by_E_threads_hw_queues_16": (sp.get("engines_on_threads_hw_queues_16") or {}).get("per_call_ms"),
                                          "note": "no multi-GPU node was available: one device at both shard sizes of a 1 -> 8 GPU split of config 4, and the host launch path of E concurrent engines (details file)"}
            except Exception as e_:
                out["scaling_preview"] = {"error": repr(e_)}
        if not args.no_parity:
            details["settings"] = settings_table(torch, binding, batch, dev, streams[0], gold)
            ok_all = [r for r in details["settings"] if r.get("e_y_rms_vs_exact_optimum", {}).get("n_gt_1e-4_m", 1) == 0 and r["unsolved"] == 0]
            best = max(ok_all, key=lambda r: r["paths_per_s"]) if ok_all else None
            out["config"]["fastest_setting_with_every_path_within_1e-4_m"] = None if best is None else {"setting": best["setting"], "paths_per_s": best["paths_per_s"], "ms": best["ms"]}
        stage_ctx = None
        if not args.no_stages:
            details["stages"], stage_ctx = stage_legs_gpu(torch, binding, synth, engs[0], streams[0], dbatch, B)
            fp = details["stages"].get("full_pipeline", {})
            out["stages"] = {"bounds_ms": details["stages"]["bounds_producer"]["ms"], "post_check_ms": details["stages"]["post_check"]["ms"],
                             "full_pipeline_instances_per_s": fp.get("instances_per_s"), "full_pipeline_ms": fp.get("ms")}
        if not args.no_live_traffic:
            lt = live_traffic(B)
            if lt is not None:
                rf = out["roofline"]
                rf["traffic"] = lt["hbm_bytes_per_launch"]
                rf["traffic_note"] = lt["source"] + f" (FETCH_SIZE {lt['fetch_bytes_raw'] / 1e6:.1f} MB x 2 + WRITE_SIZE {lt['write_bytes'] / 1e6:.1f} MB; compulsory I/O {8 * (18 * N + 8) * B / 1e6:.1f} MB)"
                rf["traffic_GBps"] = lt["hbm_bytes_per_launch"] / (out["single_batch"]["median_ms"] * 1e-3) / 1e9
                rf["valu_wave_instr_per_path_iter"] = lt["valu_wave_instr_per_launch"] / float(info["iters"].sum())  # SQ_INSTS_VALU of this run
                if lt.get("fp64_flop_per_launch"):
                    rf["achieved"] = lt["fp64_flop_per_launch"] / (out["single_batch"]["median_ms"] * 1e-3) / 1e12
                    rf["frac"] = rf["achieved"] / FP64_VALU_PEAK_TFLOPS
                    rf["fp64_wave_instr_per_path_iter"] = lt["fp64_wave_instr_per_launch"] / float(info["iters"].sum())
                    rf["achieved_def"] = ("fp64 VALU flop per launch measured in this run (rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of the solve kernel: wave-instructions x 64 lanes, "
                                          "FMA x 2) / single-batch time (hipEvents, median of the timed steps)")
                rf["frac_of_measured_issue_ceiling"] = {k: rf["achieved"] / v for k, v in FP64_VALU_MEASURED_CEILING.items()} if rf.get("achieved") else None
                nk_ms = (out["single_batch"].get("phases_ms_last_step") or {}).get("newton")
                if lt.get("newton_kernel_fp64_flop_per_launch") and nk_ms:
                    rf["dominant_kernel"] = {"kernel": "po::newton_kernel<KP,4,64>: the sliced pair of launches (8 steps of every path; the parked rest, longest expected first) with the 10 us sort between them (+ the fallback launch behind, empty here)", "launch_ms": nk_ms,
                                             "fp64_flop_per_launch": lt["newton_kernel_fp64_flop_per_launch"], "achieved": lt["newton_kernel_fp64_flop_per_launch"] / (nk_ms * 1e-3) / 1e12,
                                             "frac": lt["newton_kernel_fp64_flop_per_launch"] / (nk_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                             "launch_ms_source": "hipEvents on the engine's stream around the launch (po_last_phase_ms), last timed step"}
                    # what bounds it (round 6): the issue cadence of a lone wave per SIMD — from the newest committed counter attribution (tools/stall_pmc.sh; the split of a
                    # wave's resident time does not depend on the run), beside the same counters of a cache-resident loop of independent v_fma_f64 (tools/ubench/issue_mix.hip)
                    sa = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r6*", "stall_attribution.json")))
                    sj = _load_json(sa[-1]) if sa else None
                    try:
                        dk = sj["by_batch_size"]["4096"]["derived"]["po::newton_kernel<0, 4, 64, 1, 1>"]
                        rf["dominant_kernel"]["issue_attribution"] = {
                            "frac_issuing": dk["frac_issuing"], "frac_wait_any": dk["frac_wait_any"], "frac_wait_inst_any": dk["frac_wait_inst_any"], "cycles_per_instruction": dk["cycles_per_inst"],
                            "instruction_cache_miss_rate": dk["icache_miss_rate"], "lds_bank_conflict_of_lds_active": dk["lds_bank_conflict_of_active"],
                            "lone_wave_loop_of_independent_v_fma_f64": {"frac_issuing": 0.653, "frac_wait_any": 0.347, "frac_wait_inst_any": 0.0, "cycles_per_instruction": 6.13},
                            "reading": "a wave alone on its SIMD issues ANY independent VALU stream at 6 - 7 cycles per instruction (two waves: 3.4 - 3.8 per SIMD); the kernel sits at that "
                                       "cadence — no memory, LDS, scratch or instruction-fetch stall to remove; 38 % of its instructions are fp64, hence frac 0.13 of the fp64 lane peak",
                            "source": os.path.relpath(sa[-1], ROOT) + ", profiles/r6a/ubench/issue_mix_pmc.json"}
                    except (KeyError, TypeError):
                        pass
        torch.cuda.synchronize()
        out["gpu_done_s"] = time.time()  # everything after this timestamp is host-only (CPU baseline / checker legs)
        if args.cpu_sample > 0:
            cpu_legs(out, details, batch, dev_samples, args.cpu_sample)
            if stage_ctx is not None:
                stage_legs_cpu(details["stages"], stage_ctx)
    if rank == 0 and details and args.details:
        try:
            json.dump(details, open(args.details, "w"), indent=1)
            out["details_file"] = os.path.relpath(args.details, ROOT)
        except OSError:
            pass
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
