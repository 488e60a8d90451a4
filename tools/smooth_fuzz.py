"""Dev tool (GPU box): the smoothing-QP engine against the oracle on random sizes, ragged batches and batch sizes that pick every kernel variant."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from path_optimizer_amd import binding, synth
from oracle import oracle_py as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dist, res, px, py, _ = synth.make_distance_map(3)
eng = binding.Engine(0); eng.set_map(dist, res, px, py)
omap = O.make_map(dist, res, px, py)
worst = 0.0; bad = 0; t0 = time.time()
for case in range(n_cases):
    kind = int(rng.integers(0, 3)) if len(sys.argv) <= 3 else int(sys.argv[3])
    P = int(rng.integers(6, 261 if kind != 1 else 121))
    B = int(rng.choice([1, 2, 5, 17]))
    eps = float(rng.choice([1e-3, 1e-4]))
    inp = synth.make_smooth_inputs(int(rng.integers(1 << 30)), B, P=P, kind=kind, ragged=bool(rng.integers(0, 2)), jitter_ds=True)
    p = binding.default_params(); p.eps_abs = p.eps_rel = eps
    wv = str(rng.choice(["", "1", "4", "8"]))  # forced waves per QP (the launcher's own choice when empty)
    e2 = binding.Engine(0, p); e2.set_map(dist, res, px, py)
    e2.debug_set("smooth_waves", int(wv or 0))
    if len(sys.argv) > 4: e2.debug_set("smooth_seq", int(sys.argv[4]))
    dev = e2.smooth_batch(kind, inp, want_raw=True)
    op = O.default_params(); op.eps_abs = op.eps_rel = eps
    orc = O.smooth_batch(kind, op, inp, m_map=omap, want_raw=True)
    e2.close()
    same = dev[3]["iters"] == orc[3]["iters"]
    ok = np.array_equal(dev[3]["status"], orc[3]["status"]) and same.mean() >= 0.5
    err = float(np.abs(dev[4][same] - orc[4][same]).max()) if same.any() else 0.0
    worst = max(worst, err)
    if not ok or err > 1e-6:
        bad += 1
        print("MISMATCH" if not ok else "large", "kind", kind, "P", P, "B", B, "eps", eps, "waves", wv, "status equal", np.array_equal(dev[3]["status"], orc[3]["status"]), "same iters", same.mean(), "err", err, flush=True)
print("cases", n_cases, "mismatches", bad, "worst |d raw| on equal-iteration instances %.2e" % worst, "%.0f s" % (time.time() - t0))
