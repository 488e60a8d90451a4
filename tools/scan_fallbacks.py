"""Dev tool (GPU box): how often the scan factorisation's cross-check falls back to the sequential chain (PO_DEBUG_CYCLES counters of path 0), plain and refined solves."""
import os, sys
os.environ["PO_DEBUG_CYCLES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import np_twin as T
from path_optimizer_amd import binding, synth
for keep, N, ds in ((3, 100, 0.3), (5, 100, 0.22), (4, 200, 0.25), (6, 100, 0.19), (2, 128, 0.5)):
    rng = np.random.default_rng(keep)
    insts = [T.random_instance(rng, N, ds=ds) for _ in range(1)]
    stk = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    b = synth.Batch(0, 1, N, keep, stk("ref_x"), stk("ref_y"), stk("ref_z"), stk("ref_k"), stk("ref_s"), stk("bounds"), stk("x0"), np.array([i["goal_z"] for i in insts]))
    for kw in (dict(), dict(refine=1), dict(refine=1, refine_rounds=3), dict(polish=1, polish_passes=4)):
        p = binding.default_params()
        for k, v in kw.items(): setattr(p, k, v)
        print("== keep", keep, "N", N, kw, flush=True)
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
for cfg, form in ((5, None), (3, 2)):
    b = synth.make_batch(cfg, B=1) if form is None else synth.make_batch(cfg, B=1, formulation=form)
    for kw in (dict(), dict(refine=1)):
        p = binding.default_params()
        for k, v in kw.items(): setattr(p, k, v)
        print("== cfg", cfg, form, kw, flush=True)
        binding.Engine(0, p).solve_batch(b)
