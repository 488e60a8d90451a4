"""Dev tool: po_plan_batch_device on 4096 planning instances (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from path_optimizer_amd import binding, synth  # noqa: E402
from path_optimizer_amd.abi import INFO_BYTES
from path_optimizer_amd.abi import INFO_DTYPE  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scn = synth.make_planning_scenes(2, 64)
par = binding.default_params()
for kv in sys.argv[3:]:  # e.g. refine=2 eps_abs=3e-4 eps_rel=3e-4
    k_, v_ = kv.split("=")
    setattr(par, k_, type(getattr(par, k_))(float(v_)))
eng = binding.Engine(0, par)
eng.set_map(*scn["map"])
rs = -(-B // 64)
perm = np.random.default_rng(5).permutation(64 * rs)[:B] % 64  # shuffled replication; "periodic" as 2nd argument: scene b % 64 (worst case for a round-robin XCD dispatch)
if len(sys.argv) > 2 and sys.argv[2] == "periodic":
    perm = np.arange(B) % 64
tp = {k: torch.from_numpy(np.ascontiguousarray(scn[k][perm])).cuda() for k in ("way_x", "way_y", "start", "goal")}
Np = 320
out = dict(states=torch.zeros((B, Np, 5), dtype=torch.float64, device="cuda"), n_states=torch.zeros(B, dtype=torch.int32, device="cuda"),
           ok=torch.zeros(B, dtype=torch.int32, device="cuda"), stage=torch.zeros(B, dtype=torch.int32, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda"))
way_len = float(np.hypot(np.diff(scn["way_x"], axis=1), np.diff(scn["way_y"], axis=1)).sum(axis=1).max())
eng.plan_batch_device(tp, out, Np, way_len); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.plan_batch_device(tp, out, Np, way_len)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
inf = out["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
print("plan: %.2f ms / %d = %.0f instances/s, ok %.3f, states mean %.1f, QP iters mean %.0f max %d" % (ms, B, B / ms * 1e3, out["ok"].double().mean().item(),
      out["n_states"].double().mean().item(), inf["iters"].mean(), inf["iters"].max()))
