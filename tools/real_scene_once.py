"""Dev tool (GPU box): PathOptimizer::solve on the reference's benchmark scene, one instance, a few calls (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_amd import binding
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "benchmark_scene.npz"))
p = binding.default_params(); p.eps_abs = p.eps_rel = 1e-3
eng = binding.Engine(0, p)
eng.set_map(g["distance"], float(g["resolution"]), float(g["pos"][0]), float(g["pos"][1]))
args = (g["way_x"][None], g["way_y"][None], g["start"][None], g["goal"][None])
eng.plan_batch(*args, N=512)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); out = eng.plan_batch(*args, N=512); ts.append((time.perf_counter() - t0) * 1e3)
print("median ms %.2f" % np.median(ts), "ok", out[2], "iters", out[4]["iters"])
