"""Dev tool: cost of ONE ADMM iteration of the path QP per (keep, N): fixed 200 iterations, no termination checks / rho adaption,
4096 paths per launch -> path-iterations/s free of the iteration-count tail."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import np_twin as T
from path_optimizer_amd import binding, synth

p = binding.default_params(); p.max_iter = 200; p.check_every = 0; p.adapt_every = 0
eng = binding.Engine(0, p)


def rand_batch(B, N, ds, seed):
    rng = np.random.default_rng(seed)
    insts = [T.random_instance(rng, N, ds=ds) for _ in range(B)]
    st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    return synth.Batch(0, B, N, 4, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"), st("x0"), np.array([i["goal_z"] for i in insts]))


for keep, N in [(4, 200), (4, 232), (4, 256), (3, 192), (3, 200), (3, 231), (2, 128), (2, 200), (6, 231), (8, 231)]:
    b = rand_batch(64, N, 1.2 / keep * 0.999, keep)
    b.keep = keep
    b = synth.replicate(b, 4096)
    db = binding.DeviceBatch(b)
    eng.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.solve_batch_device(db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    it = db.info_numpy()["iters"]
    print(f"keep={keep} N={N}: {dt*1e3:7.2f} ms  {it.sum()/dt:.3e} path-iters/s  ({dt/ (it.mean()) *1e6:.2f} us per iteration of the whole batch)", flush=True)
