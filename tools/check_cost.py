"""Dev tool: what the termination checks and the adaptive-rho refactorisations cost on top of the bare iteration (fixed 400 iterations)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import np_twin as T
from path_optimizer_amd import binding, synth


def rand_batch(B, N, ds, seed):
    rng = np.random.default_rng(seed)
    insts = [T.random_instance(rng, N, ds=ds) for _ in range(B)]
    st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    return synth.Batch(0, B, N, 4, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"), st("x0"), np.array([i["goal_z"] for i in insts]))


for keep, N in [(4, 200), (3, 231)]:
    b = rand_batch(64, N, 1.2 / keep * 0.999, keep)
    b.keep = keep
    b = synth.replicate(b, 4096)
    for name, chk, adp in (("bare", 0, 0), ("checks", 25, 0), ("checks+adapt", 25, 100)):
        p = binding.default_params(); p.max_iter = 400; p.check_every = chk; p.adapt_every = adp; p.eps_abs = 0.0; p.eps_rel = 0.0
        p.eps_prim_inf = 1e-30; p.eps_dual_inf = 1e-30
        eng = binding.Engine(0, p)
        db = binding.DeviceBatch(b)
        eng.solve_batch_device(db); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.solve_batch_device(db)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        info = db.info_numpy()
        print(f"keep={keep} N={N} {name:13s}: {dt*1e3:7.2f} ms  {info['iters'].sum()/dt:.3e} path-iters/s  iters mean {info['iters'].mean():.0f} refactor mean {info['n_refactor'].mean():.2f}", flush=True)
        eng.close()
