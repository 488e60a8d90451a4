"""Dev tool (GPU box): keep-3 refinement, scan-factorisation fallback counters (debug_cycles) and device vs oracle counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import np_twin as T
from oracle import oracle_py as O
from path_optimizer_amd import binding, synth
from diag_refine import keep_batch
for keep, N, ds in ((3, 100, 0.3), (5, 100, 0.22), (4, 90, 0.25)):
    b = keep_batch(keep, N, ds)
    p = binding.default_params(); p.refine = 1
    eng = binding.Engine(0, p); eng.debug_set("debug_cycles", 1)
    st, info, xs = eng.solve_batch(b, want_x=True)
    ost, oinfo, oxs = O.solve_batch(b, O.device_equivalent_params(p))
    print(keep, N, "dev", info["iters"].tolist(), "orc", oinfo["iters"].tolist(), "dx %.2e" % np.abs(xs - oxs).max(), flush=True)
