"""Dev tool (GPU box): single-batch time of the headline setting under the queue policies of the chained rounds (po_debug_set "queue_policy") and speculation settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
full = synth.make_batch(cfg)
db = binding.DeviceBatch(full)
for spec in (1, 0, 2, -1):
    for pol in (0, 1, 2, 3):
        p = binding.default_params(); p.refine, p.refine_rounds, p.refine_extra_rounds, p.refine_speculate = 1, 3, 2, spec
        eng = binding.Engine(0, p); eng.debug_set("queue_policy", pol)
        eng.solve_batch_device(db); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); eng.solve_batch_device(db); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        info = db.info_numpy()
        print("speculate from %2d policy %d: median %.2f ms min %.2f  (unsolved %d, certified %d)" % (spec, pol, np.median(ts), np.min(ts), (info["status"] != 1).sum(), (info["status_refine"] == 1).sum()), flush=True)
        eng.close()
