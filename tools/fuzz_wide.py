"""Dev tool (GPU box): the two sweeps of tests/test_gpu_fuzz.py with the case generator re-pointed at the WIDE role-split shapes — KP, keep_control_steps_ 9 .. 16 (spacing 1.2 / keep),
path lengths up to the one-wave limit 32 keep, ragged lengths, the same random parameter block.  `python tools/fuzz_wide.py LO HI`"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import np_twin as T
import test_gpu_fuzz as F
from oracle import oracle_py
from path_optimizer_amd import synth


def _case(seed):
    rng = np.random.default_rng(5000 + seed)
    form = T.PO_KP
    keep = int(rng.integers(9, 17))
    ds = 1.2 / keep * 0.999
    N = int(rng.integers(6, min(512, 32 * keep) + 1))  # (beyond 32 keep the single-level chain takes over, where `refine` is ignored)
    B = 5
    narrow = bool(rng.integers(0, 2))
    insts = [T.random_instance(rng, N, ds=ds, narrow=narrow) for _ in range(B)]
    stk = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    b = synth.Batch(form, B, N, 4, stk("ref_x"), stk("ref_y"), stk("ref_z"), stk("ref_k"), stk("ref_s"), stk("bounds"), stk("x0"), np.array([i["goal_z"] for i in insts]), None, None)
    if rng.integers(0, 2):
        npts = rng.integers(max(3, N // 3), N + 1, size=B).astype(np.int32)
        npts[0] = N
        b.n_points = npts
    return rng, form, b


F._case = _case
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    for fn in (F.test_random_case_matches_oracle, F.test_random_case_newton_matches_oracle):
        try:
            fn(oracle_py, seed)
        except Exception as e:
            bad += 1
            print("FAIL", seed, fn.__name__, repr(e)[:300], flush=True)
print("wide seeds", lo, hi, "failures", bad)
