"""Dev tool (GPU box): latency of the reference-smoothing QP engine on small batches, one, four and eight waves per QP (po_debug_set "smooth_waves") and the automatic choice."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_amd import binding, synth  # noqa: E402
from path_optimizer_amd.abi import INFO_BYTES
from path_optimizer_amd.abi import INFO_DTYPE  # noqa: E402

p = binding.default_params(); p.eps_abs = p.eps_rel = 1e-3
eng = binding.Engine(0, p)
dist, res, px, py, _ = synth.make_distance_map(3); eng.set_map(dist, res, px, py)
for kind, P in ((1, 100), (1, 250)) if os.environ.get("KIND1") else ((0, 100), (0, 150), (0, 250), (2, 60), (2, 100), (2, 250)):
    base = synth.make_smooth_inputs(30, 256, P=P, kind=kind)
    for B in (1, 768, 4096):
        rep = {k: (None if v is None else np.concatenate([v] * ((B + 255) // 256))[:B]) for k, v in base.items()}
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in rep.items() if v is not None}
        res_ = {}
        for one in ("1", "4", "8", "0", "n1", "n4"):
            eng.debug_set("smooth_waves", int(one.lstrip("n"))); eng.debug_set("smooth_nopad", 1 if one[0] == "n" else 0)
            out = dict(x=torch.zeros((B, P), dtype=torch.float64, device="cuda"), y=torch.zeros((B, P), dtype=torch.float64, device="cuda"),
                       s=torch.zeros((B, P), dtype=torch.float64, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda"))
            eng.smooth_batch_device(kind, t, out); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5): eng.smooth_batch_device(kind, t, out)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 5
            info = out["info"].cpu().numpy().view(INFO_DTYPE).reshape(-1)
            res_[one] = (dt * 1e3, out["x"].cpu().numpy(), info["iters"].copy(), info["status"].copy())
        a = res_["1"]
        print("kind", kind, "P", P, "B", B, "ms: 1 wave %.3f  4 waves %.3f  8 waves %.3f  auto %.3f | natural layout: 1 wave %.3f  4 waves %.3f" % tuple(res_[k][0] for k in ("1", "4", "8", "0", "n1", "n4")),
              "max|dx| %.1e" % max(np.abs(a[1] - res_[k][1]).max() for k in ("4", "8", "0", "n1", "n4")),
              "iters/status equal", all(bool((a[2] == res_[k][2]).all() and (a[3] == res_[k][3]).all()) for k in ("4", "8", "0", "n1", "n4")), flush=True)
