"""Dev tool (GPU box): how much of the cold single-batch time is the luck of the start order.  The headline setting (and variants) on BASELINE config 3 under random start
orders (fresh-first queue policy forced, so that an order only permutes the starts): median / min / max over the orders."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth
full = synth.make_batch(3)
db = binding.DeviceBatch(full)
base = dict(refine=1, refine_rounds=3, refine_extra_rounds=2)
for tag, kw in (("headline", {}), ("adapt_tol 3", dict(adapt_tol=3.0)), ("adapt_tol 2", dict(adapt_tol=2.0)), ("adapt_tol 1.5", dict(adapt_tol=1.5))):
    p = binding.default_params()
    for k, v in {**base, **kw}.items(): setattr(p, k, v)
    eng = binding.Engine(0, p); eng.debug_set("queue_policy", 0)
    res = []
    for seed in range(-1, 8):
        db.set_order(None if seed < 0 else np.random.default_rng(seed).permutation(full.B).astype(np.int32))
        eng.solve_batch_device(db); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.solve_batch_device(db); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res.append(float(np.median(ts)))
    db.set_order(None)
    info = db.info_numpy()
    print("%-14s natural order %.2f ms | 8 random orders: median %.2f min %.2f max %.2f | iters mean %.0f max %d" % (tag, res[0], np.median(res[1:]), np.min(res[1:]), np.max(res[1:]), info["iters"].mean(), info["iters"].max()), flush=True)
    eng.close()
