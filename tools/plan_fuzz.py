"""Dev tool: po_plan_batch against the composed oracle pipeline on many random scenes (clean and cluttered); prints every disagreement."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle_py as o
from path_optimizer_amd import binding, synth

tot = agree = 0
for seed, near in ((11, 0), (12, 2), (13, 3), (14, 1), (15, 2)):
    sc = synth.make_planning_scenes(seed, 16, near=near, n_way=int(16 + 2 * (seed % 5)))
    eng = binding.Engine(0)
    eng.set_map(*sc["map"])
    states, n, ok, stage, info = eng.plan_batch(sc["way_x"], sc["way_y"], sc["start"], sc["goal"], N=640)
    mp = o.make_map(*sc["map"])
    p = o.default_params()
    for b in range(16):
        ook, opath, tr = o.path_optimizer_solve(p, mp, sc["way_x"][b], sc["way_y"][b], sc["start"][b], sc["goal"][b])
        same = bool(ok[b]) == bool(ook) and n[b] == len(opath) and (len(opath) == 0 or np.abs(states[b, :n[b]] - opath).max() < 1e-5)
        tot += 1; agree += same
        if not same:
            d = np.abs(states[b, :min(n[b], len(opath))] - opath[:min(n[b], len(opath))]).max() if min(n[b], len(opath)) else None
            print("seed", seed, "scene", b, "device ok/n/stage/iters", ok[b], n[b], stage[b], info["iters"][b], "oracle ok/n/iters", ook, len(opath),
                  tr["qp"]["iters"] if "qp" in tr else None, "dp layers", tr["dp"][0] if "dp" in tr else None, "max diff", d, flush=True)
    eng.close()
print("agree", agree, "of", tot)
