"""Dev tool (GPU box): how much of the Newton launch at the headline setting is TAIL (slots idle while the last long paths finish)?
BASELINE config 3: Newton steps per path, greedy list scheduling of the 4096 paths onto 1024 wave slots simulated in a random order and longest-first, and the
launch measured with the engine's own order, and with po_batch_in.order = longest first (from the previous solve's po_info.iters)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heapq
import numpy as np, torch
from path_optimizer_amd import binding, synth
HEADLINE = dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2)
batch = synth.make_batch(int(os.environ.get("AB_CFG", "3")), B=4096)
db = binding.DeviceBatch(batch)
p = binding.default_params()
for k, v in HEADLINE.items(): setattr(p, k, v)
eng = binding.Engine(0, p); s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
def run(n=7):
    for _ in range(2): eng.solve_batch_device(db); torch.cuda.synchronize()
    ts, nw = [], []
    for _ in range(n):
        eng.solve_batch_device(db); torch.cuda.synchronize(); ts.append(eng.last_kernel_ms()); nw.append(eng.last_phase_ms()["newton"])
    return float(np.median(ts)), float(np.median(nw))
base = run()
info = db.info_numpy()
steps = info["iters"] - 25
def sim(order, slots=1024):
    h = [0.0] * slots
    for i in order:
        t = heapq.heappop(h); heapq.heappush(h, t + steps[i])
    return max(h)
rng = np.random.default_rng(0)
out = {"steps_mean": float(steps.mean()), "steps_max": int(steps.max()), "steps_p99": float(np.percentile(steps, 99)), "hist": np.bincount(steps).tolist(),
       "ideal_steps_per_slot": float(steps.sum() / 1024), "sim_random_order": float(np.mean([sim(rng.permutation(4096)) for _ in range(5)])),
       "sim_longest_first": float(sim(np.argsort(-steps, kind="stable"))), "ms_total_newton_engine_order": base}
db.set_order(np.argsort(-steps, kind="stable"))
out["ms_total_newton_longest_first"] = run()
db.set_order(np.argsort(steps, kind="stable"))
out["ms_total_newton_shortest_first"] = run()
print(json.dumps(out))
