"""Generates tests/golden/smooth_ref.npz from the REFERENCE's own smoother classes (oracle/_ref/libpo_ref_smooth.so = the reference's
reference_path_smoother / tension_smoother / tension_smoother_2 sources compiled where they lie; only OSQP is the oracle's ADMM).
Run in the build container (needs /root/reference):  python tests/golden/make_smooth_golden.py
Per kind (0 TENSION2, 1 TENSION, 2 POST) and instance: the inputs, the QP the reference handed to OsqpEigen (P upper CSC, A CSC, q, l, u)
and what osqpSmooth returned (result lists) / the QP solution (POST), at OSQP's default eps = 1e-3 (what the reference runs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as o, ref_py as r  # noqa: E402
from path_optimizer_amd import synth  # noqa: E402

MAP_SEED, MAP_KW = 3, dict(size_x=400, size_y=300, resolution=0.2, pos=(10.0, -5.0))


def main():
    p = o.default_params()
    p.eps_abs = p.eps_rel = 1e-3
    dist, res, px, py, _ = synth.make_distance_map(MAP_SEED, **MAP_KW)
    mp = o.make_map(dist, res, px, py)
    out = {}
    for kind, P in ((0, 48), (1, 40), (2, 33)):
        inp = synth.make_smooth_inputs(11, 6, P=P, kind=kind, ragged=True, jitter_ds=True)
        for key, val in inp.items():
            out[f"k{kind}_{key}"] = val
        sp = synth.make_spline_paths(2, 6, N=200)
        for b in range(6):
            n = int(inp["n_points"][b])
            if kind < 2:
                ref = r.osqp_smooth(kind, p, inp["x"][b, :n], inp["y"][b, :n], inp["angle"][b, :n], inp["k"][b, :n], inp["s"][b, :n], m_map=mp)
                out[f"k{kind}_{b}_out"] = np.stack([ref["out_x"], ref["out_y"], ref["out_s"]])
            else:
                ref = r.post_smooth(p, inp["s"][b, :n], inp["lb"][b, :n], inp["ub"][b, :n], inp["l0"][b], sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b])
            assert ref["rc"] == 1
            out[f"k{kind}_{b}_x"] = ref["x"]
            for name in ("q", "l", "u"):
                out[f"k{kind}_{b}_{name}"] = ref[name]
            for name in ("P", "A"):
                M = ref[name]
                out[f"k{kind}_{b}_{name}p"] = M.indptr.astype(np.int32); out[f"k{kind}_{b}_{name}i"] = M.indices.astype(np.int32); out[f"k{kind}_{b}_{name}x"] = M.data
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "smooth_ref.npz"), **out)
    print("wrote smooth_ref.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
