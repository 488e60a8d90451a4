import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from path_optimizer_amd import binding, synth
from path_optimizer_amd.abi import INFO_BYTES
p = binding.default_params(); p.eps_abs = p.eps_rel = 1e-12; p.max_iter = 200
for kind in (1, 0):
    for waves in (1, 4):
        eng = binding.Engine(0, p)
        dist, res, px, py, _ = synth.make_distance_map(3); eng.set_map(dist, res, px, py)
        eng.debug_set("smooth_waves", waves); eng.debug_set("smooth_debug", 1)
        P, B = 100, 256
        base = synth.make_smooth_inputs(30, 256, P=P, kind=kind)
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in base.items() if v is not None}
        out = dict(x=torch.zeros((B, P), dtype=torch.float64, device="cuda"), y=torch.zeros((B, P), dtype=torch.float64, device="cuda"),
                   s=torch.zeros((B, P), dtype=torch.float64, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda"))
        print("kind", kind, "waves", waves, file=sys.stderr, flush=True)
        eng.smooth_batch_device(kind, t, out); torch.cuda.synchronize()
        eng.close()
