"""Dev tool (GPU box): the headline setting (round 4: Newton refinement, every scheduling; `r3` as third argument: round 3's chained rounds with speculation) against the oracle on random batches of every formulation / keep / ragged lengths; prints disagreements.
python tools/headline_fuzz.py [cases] [seed] [r3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle_py as O
from path_optimizer_amd import binding, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; worst = 0.0; t0 = time.time(); tot = 0; eq = 0
for case in range(n_cases):
    form = int(rng.choice([0, 0, 0, 1, 2]))
    cfg = 5 if form == 1 else 3
    B = int(rng.choice([1, 3, 16, 40]))
    first = int(rng.integers(0, 4000))
    kw = {}
    if form == 2: kw["formulation"] = 2
    if form == 0 and rng.integers(0, 3) == 0: kw.update(N=int(rng.integers(60, 240)), ds=float(rng.choice([0.3, 0.25, 0.4])))
    b = synth.make_batch(cfg, B=B, first_path=first, **kw)
    if rng.integers(0, 2):  # ragged
        npts = rng.integers(max(8, b.N // 2), b.N + 1, B).astype(np.int32); npts[0] = b.N
        b.n_points = npts
    p = binding.default_params()
    if len(sys.argv) > 3 and sys.argv[3] == "r3":
        p.refine, p.refine_rounds, p.refine_extra_rounds = 1, 3, 2
        p.refine_chain = int(rng.integers(0, 2)); p.refine_speculate = int(rng.choice([1, 0, -1, 2]))
    else:
        p.refine, p.refine_rounds, p.refine_extra_rounds, p.refine_eps = 2, 5, 2, 1e-8
        p.refine_chain = int(rng.choice([2, 2, 3, 1, 0])); p.refine_speculate = int(rng.choice([1, -1]))
        if rng.integers(0, 3) == 0: b.bounds *= float(rng.choice([0.5, 0.7]))  # narrower corridors: soft margins bind, degenerate optima
    if rng.integers(0, 4) == 0: p.max_iter = int(rng.choice([300, 700, 1500]))
    try:
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
    except Exception as e:
        print("case", case, "device error", e, form, B, kw); bad += 1; continue
    ost, oinfo, oxs = O.solve_batch(b, O.device_equivalent_params(p), want_x=True)
    same = np.abs(info["iters"].astype(int) - oinfo["iters"].astype(int)) <= (3 if p.refine == 2 else 0)
    ok = np.array_equal(info["status"], oinfo["status"]) and np.array_equal(info["status_refine"][same], oinfo["status_refine"][same])
    cmpx = same & (info["status"] == 1) & (oinfo["status"] == 1)  # (an unsolved path's iterate is not a result)
    err = float(np.abs(xs[cmpx] - oxs[cmpx]).max()) if cmpx.any() else 0.0
    cert = cmpx & (info["status_refine"] == 1) & (oinfo["status_refine"] == 1)
    err_c = float(np.abs(xs[cert] - oxs[cert]).max()) if cert.any() else 0.0
    tot += B; eq += int(same.sum()); worst = max(worst, err)
    if not ok or err > 1e-5 or same.mean() < 0.6:
        bad += 1
        print("MISMATCH case", case, "form", form, "B", B, kw, "chain", p.refine_chain, "spec", p.refine_speculate, "max_iter", p.max_iter, "| status equal", np.array_equal(info["status"], oinfo["status"]),
              "same iters %.2f" % same.mean(), "err %.2e (certified on both: %.2e, n solved %d certified %d)" % (err, err_c, cmpx.sum(), cert.sum()), "dev status", info["status"][~same][:5], "orc", oinfo["status"][~same][:5], flush=True)
print("cases", n_cases, "paths", tot, "equal iteration counts", eq, "mismatching cases", bad, "worst |dx| on equal-count paths %.2e" % worst, "%.0f s" % (time.time() - t0))
