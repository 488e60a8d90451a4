"""Dev tool: throughput of the f-4 stages (DP lattice search, re-sampling) on 4096 spline paths."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from path_optimizer_amd import binding, synth  # noqa: E402

B = 4096
eng = binding.Engine(0)
d, res, px, py, _ = synth.make_distance_map(seed=3, size_x=600, size_y=600, resolution=0.2, pos=(1.0, -2.0), n_obstacles=60, r_range=(0.5, 3.0))
eng.set_map(d, res, px, py)
sp, length, start = synth.make_search_inputs(9, 256)
rep = lambda a: np.ascontiguousarray(np.concatenate([a] * (B // 256)))
t = {k: torch.from_numpy(rep(sp[k])).cuda() for k in ("knot_s", "knot_x", "knot_y")}
t["length"] = torch.from_numpy(rep(length)).cuda()
st = torch.from_numpy(rep(start)).cuda()
L = 64
out = dict(layer_s=torch.zeros((B, L), dtype=torch.float64, device="cuda"), lb=torch.zeros((B, L), dtype=torch.float64, device="cuda"),
           ub=torch.zeros((B, L), dtype=torch.float64, device="cuda"), l0=torch.zeros(B, dtype=torch.float64, device="cuda"),
           n_layers=torch.zeros(B, dtype=torch.int32, device="cuda"))
eng.dp_search_batch_device(t, st, L, out); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    eng.dp_search_batch_device(t, st, L, out)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
nl = out["n_layers"].cpu().numpy()
print("dp search: ms/4096 %.3f paths/s %.0f layers mean %.1f edges/s %.3g" % (dt * 1e3, B / dt, nl.clip(0).mean(), nl.clip(0).sum() * 34 * 34 / dt))
N = 256
ro = {k: torch.zeros((B, N), dtype=torch.float64, device="cuda") for k in ("ref_x", "ref_y", "ref_z", "ref_k", "ref_s")}
ro["n_points"] = torch.zeros(B, dtype=torch.int32, device="cuda")
eng.resample_batch_device(t, 0.15, 0.3, N, ro); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    eng.resample_batch_device(t, 0.15, 0.3, N, ro)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print("resample: ms/4096 %.3f paths/s %.0f states mean %.1f" % (dt * 1e3, B / dt, ro["n_points"].float().mean().item()))
