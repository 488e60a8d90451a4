"""Iteration statistics / time of config 3 vs the adaptive-rho interval (iterations).  Dev tool (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth
base = synth.make_batch(3, B=4096)
db = binding.DeviceBatch(base)
for ae, ce in [(100, 25), (75, 25), (50, 25), (25, 25), (50, 10), (20, 10), (40, 20), (30,10), (10, 10)]:
    p = binding.default_params(); p.adapt_every = ae; p.check_every = ce
    eng = binding.Engine(0, p)
    s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
    eng.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.solve_batch_device(db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    info = db.info_numpy(); it = info["iters"]
    print(f"adapt {ae:3d} check {ce:2d}: {dt*1e3:7.2f} ms {4096/dt:8.0f} paths/s  iters mean {it.mean():.0f} med {np.median(it):.0f} p95 {np.percentile(it,95):.0f} p99 {np.percentile(it,99):.0f} max {it.max()}  refactor mean {info['n_refactor'].mean():.2f} max {info['n_refactor'].max()} unsolved {(info['status']!=1).sum()}", flush=True)
