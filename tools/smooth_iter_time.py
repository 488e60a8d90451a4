"""Dev tool (GPU box): time per ADMM iteration of the reference-smoothing QP engine at a fixed iteration count (eps 1e-12, max_iter 200), by batch size and waves per QP."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_amd import binding, synth  # noqa: E402
from path_optimizer_amd.abi import INFO_BYTES, INFO_DTYPE  # noqa: E402

IT = 200
p = binding.default_params(); p.eps_abs = p.eps_rel = 1e-12; p.max_iter = IT
kinds = [int(k) for k in sys.argv[1:]] or [1, 0]
for kind in kinds:
    P = 100
    base = synth.make_smooth_inputs(30, 256, P=P, kind=kind)
    for waves in (0, 1, 4, 8):
        eng = binding.Engine(0, p)
        dist, res, px, py, _ = synth.make_distance_map(3); eng.set_map(dist, res, px, py)
        eng.debug_set("smooth_waves", waves)
        row = []
        for B in (1, 256, 768, 1536, 4096):
            rep = {k: (None if v is None else np.concatenate([v] * ((B + 255) // 256))[:B]) for k, v in base.items()}
            t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in rep.items() if v is not None}
            out = dict(x=torch.zeros((B, P), dtype=torch.float64, device="cuda"), y=torch.zeros((B, P), dtype=torch.float64, device="cuda"),
                       s=torch.zeros((B, P), dtype=torch.float64, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda"))
            eng.smooth_batch_device(kind, t, out); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3): eng.smooth_batch_device(kind, t, out)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 3
            info = out["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
            row.append("B %d: %.2f ms, %.1f us/iter (iters %d..%d, refactor mean %.1f)" % (B, dt * 1e3, dt * 1e6 / max(1, info["iters"].max()), info["iters"].min(), info["iters"].max(), info["n_refactor"].mean()))
        print("kind", kind, "P", P, "waves", waves, "|", " | ".join(row), flush=True)
        eng.close()
