"""Dev tool (GPU box): device vs oracle iteration counts of the refinement for chosen shapes and switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import np_twin as T
from oracle import oracle_py as O
from path_optimizer_amd import binding, synth

def keep_batch(keep, N, ds):
    rng = np.random.default_rng(keep)
    insts = [T.random_instance(rng, N, ds=ds) for _ in range(8)]
    stk = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    return synth.Batch(0, 8, N, keep, stk("ref_x"), stk("ref_y"), stk("ref_z"), stk("ref_k"), stk("ref_s"), stk("bounds"), stk("x0"), np.array([i["goal_z"] for i in insts]))

if __name__ == "__main__":
  for keep, N, ds in ((3, 100, 0.3), (4, 90, 0.25), (2, 70, 0.5)):
      b = keep_batch(keep, N, ds)
      for kw in (dict(refine=1, refine_adapt=0, refine_eps=1e-6), dict(refine=1, refine_adapt=0), dict(refine=1, refine_adapt=1), dict(refine=0)):
          p = binding.default_params()
          for k, v in kw.items(): setattr(p, k, v)
          st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
          ost, oinfo, oxs = O.solve_batch(b, O.device_equivalent_params(p))
          print(keep, N, kw, "dev", info["iters"].tolist(), "orc", oinfo["iters"].tolist(), "nref", info["n_refactor"].tolist(), oinfo["n_refactor"].tolist(), "dx %.2e" % np.abs(xs - oxs).max(), flush=True)

  # the chained test's batch: path 30 (MAX_ITER)
  b = synth.make_batch(3, B=33)
  b.n_points = np.full(33, b.N, dtype=np.int32); b.n_points[1::5] = b.N - 7; b.n_points[2::7] = b.N // 2
  b.bounds[3::4, 30:45, 0, :] = (-1e30, 1e30)
  for kw in (dict(), dict(refine=1), dict(refine=1, refine_rounds=4, refine_chain=0), dict(refine=1, refine_rounds=4, refine_chain=0, refine_adapt=0), dict(refine=1, refine_rounds=4, refine_chain=0, refine_eps=1e-6)):
      p = binding.default_params()
      for k, v in kw.items(): setattr(p, k, v)
      st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
      ost, oinfo, oxs = O.solve_batch(b, O.device_equivalent_params(p))
      print(kw, "path30 dev", info[30], "orc", oinfo[30], "status", info["status"].tolist(), flush=True)
