"""Dev tool: throughput of the reference-smoothing QP engine (po_smooth_batch_device) per kind, 4096 instances per launch."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from path_optimizer_amd import binding, synth  # noqa: E402
from path_optimizer_amd.abi import INFO_BYTES
from path_optimizer_amd.abi import INFO_DTYPE  # noqa: E402

B = 4096
for eps in (1e-3, 1e-4):
    p = binding.default_params(); p.eps_abs = p.eps_rel = eps
    eng = binding.Engine(0, p)
    dist, res, px, py, _ = synth.make_distance_map(3); eng.set_map(dist, res, px, py)
    for kind, P in ((0, 100), (1, 100), (2, 60)):
        inp = synth.make_smooth_inputs(30, 256, P=P, kind=kind)
        rep = {k: (None if v is None else np.concatenate([v] * (B // 256))) for k, v in inp.items()}
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in rep.items() if v is not None}
        out = dict(x=torch.zeros((B, P), dtype=torch.float64, device="cuda"), y=torch.zeros((B, P), dtype=torch.float64, device="cuda"),
                   s=torch.zeros((B, P), dtype=torch.float64, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda"))
        eng.smooth_batch_device(kind, t, out); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            eng.smooth_batch_device(kind, t, out)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 3
        info = out["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
        print("eps", eps, "kind", kind, "P", P, "ms/4096 %.2f" % (dt * 1e3), "QP/s %.0f" % (B / dt), "iters mean %.1f max %d" % (info["iters"].mean(), info["iters"].max()),
              "refactor mean %.2f" % info["n_refactor"].mean(), "solved", (info["status"] == 1).mean(), "QP-iters/s %.3g" % (info["iters"].sum() / dt))
