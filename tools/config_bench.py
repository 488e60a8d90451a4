"""Throughput of every BASELINE config (fixed-B runs of the fused kernel).  Dev tool (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth

def run(cfg, B, form=None, N=None, nb=256):
    base = synth.make_batch(cfg, B=min(B, nb), formulation=form, N=N)
    batch = synth.replicate(base, B)
    db = binding.DeviceBatch(batch)
    eng = binding.Engine(0)
    s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
    eng.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.solve_batch_device(db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    info = db.info_numpy()
    it = info["iters"]
    print(f"cfg {cfg} form {batch.formulation} B={B} N={batch.N}: {dt*1e3:8.2f} ms  {B/dt:10.0f} paths/s  iters mean {it.mean():.0f} max {it.max()} unsolved {(info['status']!=1).sum()}  path-iters/s {it.sum()/dt:.3e}", flush=True)

run(1, 1)
run(2, 1024)
run(3, 4096)
run(5, 4096)
run(3, 4096, form=2)   # K formulation on config-3 data
run(3, 16384)
