"""Generates tests/golden/plan_ref.npz from the REFERENCE's own graphSearchDp / buildReferenceFromSpline / updateLimits
(oracle/_ref/libpo_ref_smooth.so: the reference's sources compiled where they lie).  Run in the build container:
python tests/golden/make_plan_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as o, ref_py as r  # noqa: E402
from path_optimizer_amd import synth  # noqa: E402

MAP_KW = dict(size_x=600, size_y=600, resolution=0.2, pos=(1.0, -2.0), n_obstacles=40, r_range=(0.5, 2.0))


def main():
    dist, res, px, py, _ = synth.make_distance_map(3, **MAP_KW)
    mp = o.make_map(dist, res, px, py)
    seed, B = 5, 12
    sp, length, start = synth.make_search_inputs(seed, B)
    out = dict(seed=seed, B=B, dp_n=np.zeros(B, np.int32), dp_l0=np.zeros(B), rs_n=np.zeros(B, np.int32))
    for b in range(B):
        ks, kx, ky = sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b]
        n, ls, lb, ub, l0 = r.dp_search(mp, ks, kx, ky, length[b], start[b])
        out["dp_n"][b] = n; out["dp_l0"][b] = l0; out[f"dp_{b}"] = np.stack([ls, lb, ub])
        nr, rr = r.resample(ks, kx, ky, length[b], 0.15, 0.3)
        out["rs_n"][b] = nr; out[f"rs_{b}"] = np.stack(rr)
    rng = np.random.default_rng(1)
    v = rng.uniform(0, 15, 200); v[:3] = [0, 0.0001, 0.00011]; a = rng.uniform(-4.5, 4.5, 200)
    out["lim_v"], out["lim_a"] = v, a
    out["lim_k"], out["lim_kp"] = r.limits(v, a)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "plan_ref.npz"), **out)
    print("wrote plan_ref.npz")


if __name__ == "__main__":
    main()
