import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as o
from path_optimizer_amd import binding, synth
sc = synth.make_planning_scenes(1, 8)
mp = o.make_map(*sc["map"])
eng = binding.Engine(0)
eng.set_map(*sc["map"])
t0 = time.time()
states, n, ok, stage, info = eng.plan_batch(sc["way_x"], sc["way_y"], sc["start"], sc["goal"], N=512)
print("device", time.time() - t0, n, ok, stage, info["iters"])
p = o.default_params()
for b in range(8):
    ok2, path2, tr = o.path_optimizer_solve(p, mp, sc["way_x"][b], sc["way_y"][b], sc["start"][b], sc["goal"][b])
    m = min(len(path2), n[b])
    print(b, "oracle ok", ok2, len(path2), "iters", tr["qp"]["iters"], "dev n", n[b], "max diff", np.abs(states[b, :m] - path2[:m]).max() if m else None)
