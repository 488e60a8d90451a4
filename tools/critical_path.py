"""Dev tool (GPU box): latency of the hardest paths of BASELINE config 3 solved ALONE (B = 1: the pure critical path, no queueing) at the headline setting,
next to the batch time; and the batch with those paths placed first (order hint)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth
kw = dict(refine=1, refine_rounds=3, refine_extra_rounds=2)
def mk(k):
    p = binding.default_params()
    for a, v in k.items(): setattr(p, a, v)
    return p
full = synth.make_batch(3, B=4096)
db = binding.DeviceBatch(full)
eng = binding.Engine(0, mk(kw))
def timed(d, n=5):
    eng.solve_batch_device(d); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); eng.solve_batch_device(d); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
tb = timed(db)
info = db.info_numpy().copy()
order = np.argsort(-info["iters"].astype(np.int64), kind="stable")
print("batch %.2f ms; hardest paths" % tb, [(int(i), int(info["iters"][i]), int(info["n_refactor"][i])) for i in order[:6]])
for i in order[:4]:
    d1 = binding.DeviceBatch(full.slice(int(i), int(i) + 1))
    t1 = timed(d1)
    print("path %d alone: %.2f ms (%d iters, %d refactorisations) -> %.2f us per iteration all-in" % (i, t1, info["iters"][i], info["n_refactor"][i], t1 * 1e3 / info["iters"][i]))
db.set_order(order)
print("batch with the previous solve's longest-first order: %.2f ms" % timed(db))
db.set_order(None)
for kw2 in (dict(), dict(refine=1), dict(refine=1, refine_rounds=3)):
    e2 = binding.Engine(0, mk(kw2)); e2.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter(); e2.solve_batch_device(db); torch.cuda.synchronize(); tt = (time.perf_counter() - t0) * 1e3
    i2 = db.info_numpy(); j = int(np.argmax(i2["iters"]))
    d1 = binding.DeviceBatch(full.slice(j, j + 1)); e2.solve_batch_device(d1); torch.cuda.synchronize()
    t0 = time.perf_counter(); e2.solve_batch_device(d1); torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) * 1e3
    print(kw2, "batch %.2f ms, longest path %d (%d iters) alone %.2f ms" % (tt, j, i2["iters"][j], t1))
