"""Throughput of the KP solve for every keep_control_steps value (arc-length spacing 1.2/keep), N=200 (keep=1: also N=100).  Dev tool (GPU box)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import np_twin as T
from path_optimizer_amd import binding, synth

eng = binding.Engine(0)
s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
ph = binding.default_params()
for k, v in dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2).items(): setattr(ph, k, v)  # bench.py HEADLINE
engh = binding.Engine(0, ph); engh.set_stream(s.cuda_stream)
def rand_batch(B, N, ds, seed):
    rng = np.random.default_rng(seed)
    insts = [T.random_instance(rng, N, ds=ds) for _ in range(B)]
    st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    return synth.Batch(0, B, N, 4, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"), st("x0"), np.array([i["goal_z"] for i in insts]))
out = []
for keep, N in [(1, 100), (1, 200), (2, 200), (3, 200), (4, 200), (5, 200), (6, 200), (7, 200), (8, 200), (9, 200), (10, 200), (12, 200), (14, 200), (16, 200), (17, 200)]:
    b = rand_batch(256, N, 1.2 / keep * 0.999, keep)
    b.keep = keep
    b = synth.replicate(b, 4096)
    db = binding.DeviceBatch(b)
    eng.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.solve_batch_device(db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    info = db.info_numpy()
    it = info["iters"]
    r = {"keep": keep, "N": N, "ms": dt * 1e3, "paths_per_s": 4096 / dt, "iters_mean": float(it.mean()), "iters_max": int(it.max()),
         "unsolved": int((info["status"] != 1).sum()), "path_iters_per_s": float(it.sum()) / dt}
    engh.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): engh.solve_batch_device(db)
    torch.cuda.synchronize()
    dth = (time.perf_counter() - t0) / 3
    ih = db.info_numpy()
    r.update(headline_ms=dth * 1e3, headline_paths_per_s=4096 / dth, headline_iters_mean=float(ih["iters"].mean()), headline_iters_max=int(ih["iters"].max()),
             headline_certified=int((ih["status_refine"] == 1).sum()), headline_solved=int((ih["status"] == 1).sum()))
    out.append(r)
    print(f"keep={keep:2d} N={N}: {dt*1e3:8.2f} ms {4096/dt:9.0f} paths/s  iters mean {it.mean():.0f} max {it.max()}  unsolved {r['unsolved']}  {r['path_iters_per_s']:.3e} path-iters/s | headline {dth*1e3:8.2f} ms {4096/dth:9.0f} paths/s its mean {ih['iters'].mean():.1f} max {ih['iters'].max()} certified {r['headline_certified']}/{r['headline_solved']}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/keep_sweep.json", "w"), indent=1)
