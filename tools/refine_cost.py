"""Dev tool (GPU box): what refinement iterations cost next to plain ones, one-wave (keep 4, N 200) and two-wave (keep 3, N 231) blocks; PO_DEBUG_CYCLES=1: one path, per-phase clocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import np_twin as T
from path_optimizer_amd import binding, synth
def mk(B, N, keep, ds):
    rng = np.random.default_rng(7)
    insts = [T.random_instance(rng, N, ds=ds) for _ in range(min(B, 64))]
    insts = (insts * (B // len(insts) + 1))[:B]
    stk = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    return synth.Batch(0, B, N, keep, stk("ref_x"), stk("ref_y"), stk("ref_z"), stk("ref_k"), stk("ref_s"), stk("bounds"), stk("x0"), np.array([i["goal_z"] for i in insts]))
dbg = os.environ.get("PO_DEBUG_CYCLES")
for keep, N, ds in ((3, 231, 0.3), (4, 200, 0.3)):
    b = mk(1 if dbg else 1024, N, keep, ds)
    for kw in (dict(eps_abs=1e-3, eps_rel=1e-3), dict(eps_abs=1e-3, eps_rel=1e-3, refine=1), dict(), dict(refine=1, refine_rounds=3)):
        p = binding.default_params()
        for k, v in kw.items(): setattr(p, k, v)
        eng = binding.Engine(0, p)
        print("== keep", keep, "N", N, kw, flush=True)
        eng.solve_batch(b)
        t0 = time.perf_counter(); st, info, _ = eng.solve_batch(b, want_x=True); dt = time.perf_counter() - t0
        print("   ms %.2f iters mean %.0f (main %.0f) refactor mean %.1f solved %.3f kernel_ms %.2f" % (dt * 1e3, info["iters"].mean(), info["status_polish"].mean(), info["n_refactor"].mean(), (info["status"] == 1).mean(), eng.last_kernel_ms() if hasattr(eng, "last_kernel_ms") else -1), flush=True)
