"""Dev tool (GPU box): the accuracy clause on the WHOLE batch.  Every path of a set against its exact optimum (tests/golden/tight_full_<set>.npz) for a list of
settings: count of paths whose lateral-offset RMS exceeds 1e-4 m, max / p99, status_refine counts, single-batch time.  python tools/accuracy_full.py [set ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import torch

from make_tight_full import SETS, batch_of, e_y_of
from path_optimizer_amd import binding

SETTINGS = [
    ("eps 1e-4 (OSQP-faithful default)", {}),
    ("refine, rounds 3, one launch per round", dict(refine=1, refine_rounds=3, refine_chain=0)),
    ("refine, rounds 3, chained", dict(refine=1, refine_rounds=3, refine_chain=1)),
    ("refine, rounds 3 + 2 below eps, one launch per round", dict(refine=1, refine_rounds=3, refine_extra_rounds=2, refine_chain=0)),
    ("refine, rounds 3 + 2 below eps, chained (headline)", dict(refine=1, refine_rounds=3, refine_extra_rounds=2)),
    ("refine, rounds 2 + 3 below eps, chained", dict(refine=1, refine_rounds=2, refine_extra_rounds=3)),
    ("refine, rounds 4 + 2 below eps, chained", dict(refine=1, refine_rounds=4, refine_extra_rounds=2)),
    ("refine, 1 round + 2 below eps, chained", dict(refine=1, refine_rounds=1, refine_extra_rounds=2)),
    ("headline + adapt_tol 3", dict(refine=1, refine_rounds=3, refine_extra_rounds=2, adapt_tol=3.0)),
    ("headline + adapt_tol 2", dict(refine=1, refine_rounds=3, refine_extra_rounds=2, adapt_tol=2.0)),
    ("headline + adapt_tol 1.5", dict(refine=1, refine_rounds=3, refine_extra_rounds=2, adapt_tol=1.5)),
    ("OSQP default + adapt_tol 2", dict(adapt_tol=2.0)),
]
if os.environ.get("PO_ACC_ONLY"):
    SETTINGS = [s for s in SETTINGS if any(k in s[0] for k in os.environ["PO_ACC_ONLY"].split(","))]


def main():
    names = sys.argv[1:] or ["c3"]
    out = {}
    for name in names:
        gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"tight_full_{name}.npz"))["e_y"].astype(np.float64)
        b = batch_of(name)
        db = binding.DeviceBatch(b, want_x=True)
        rows = []
        for label, kw in SETTINGS:
            p = binding.default_params()
            for k, v in kw.items():
                setattr(p, k, v)
            eng = binding.Engine(0, p)
            eng.solve_batch_device(db); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); eng.solve_batch_device(db); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            info = db.info_numpy().copy()
            x = db.out_x.cpu().numpy()
            ey = np.stack([e_y_of(b.formulation, b.N, x[i]) for i in range(b.B)])
            rms = np.sqrt(np.mean((ey - gold) ** 2, axis=1))
            bad = np.where(rms > 1e-4)[0]
            row = {"setting": label, "ms_median": float(np.median(ts)), "ms_min": float(np.min(ts)), "paths_per_s": b.B / (float(np.median(ts)) * 1e-3), "paths": int(b.B),
                   "unsolved": int((info["status"] != 1).sum()), "iters_mean": float(info["iters"].mean()), "iters_max": int(info["iters"].max()),
                   "n_gt_1e-4": int(len(bad)), "max_m": float(rms.max()), "p99_m": float(np.percentile(rms, 99)), "median_m": float(np.median(rms)),
                   "status_refine": {str(k): int((info["status_refine"] == k).sum()) for k in (-1, 0, 1)},
                   "gt_1e-4_by_status_refine": {str(k): int((info["status_refine"][bad] == k).sum()) for k in (-1, 0, 1)},
                   "max_m_certified": float(rms[info["status_refine"] == 1].max()) if (info["status_refine"] == 1).any() else None,
                   "worst": [(int(i), float(rms[i]), int(info["status_refine"][i]), int(info["iters"][i]), float(info["r_prim"][i]), float(info["r_dual"][i])) for i in bad[np.argsort(-rms[bad])][:8]]}
            rows.append(row)
            print(json.dumps(row), flush=True)
            eng.close()
        out[name] = rows
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/accuracy_full.json", "w"), indent=1)


if __name__ == "__main__":
    main()
