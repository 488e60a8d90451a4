"""BASELINE config 1 on the reference's REAL benchmark scene (/root/reference/src/test/path_optimizer_benchmark.cpp: obstacles_for_benchmark.png,
the 100 way points, start / goal): the only input fixture the reference holds.  Fixture tests/golden/benchmark_scene.npz = the outputs of the
reference-compiled PathOptimizer on it (generator tests/golden/make_benchmark_golden.py): BM_optimizePath's solve() and
BM_optimizePathWithoutSmoothing's solveWithoutSmoothing() on the result.

CPU: the composed oracle reproduces the reference's final path; the map recipe is reproducible where the PNG exists.
GPU: po_plan_batch (every stage on the device) reproduces solve(); bounds producer -> QP -> collision check reproduces solveWithoutSmoothing()."""
import os
import sys

import numpy as np
import pytest

from path_optimizer_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "benchmark_scene.npz")
HAVE_REF = os.path.isdir("/root/reference")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def _map(g):
    return g["distance"], float(g["resolution"]), float(g["pos"][0]), float(g["pos"][1])


def test_fixture_is_the_benchmark(g):
    assert g["distance"].shape == (495, 497) and g["distance"].dtype == np.float32 and len(g["way_x"]) == 100
    assert g["path1_e4"].shape == (132, 5) and g["path2_e4"].shape == (132, 5)  # the reference's re-sampling gives 132 states on this scene
    assert (g["qp_e4"][:, 0] == 1).all() and (g["qp_e3"][:, 0] == 1).all()
    assert abs(g["path1_e4"][0, 0] - g["start"][0]) < 0.5 and np.hypot(*(g["path1_e4"][-1, :2] - g["goal"][:2])) < 1.5


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference/obstacles_for_benchmark.png")
def test_map_recipe_reproduces_fixture(g):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_benchmark_golden as MB

    d, res, px, py = MB.benchmark_map()
    assert np.array_equal(d, g["distance"]) and res == 0.2 and px == 0 and py == 0
    assert np.array_equal(np.array(MB.X_LIST), g["way_x"]) and np.array_equal(np.array(MB.START), g["start"])


@pytest.mark.parametrize("tag,eps", [("e3", 1e-3), ("e4", 1e-4)])
def test_oracle_pipeline_reproduces_reference_on_the_benchmark(oracle, g, tag, eps):
    mp = oracle.make_map(*_map(g))
    p = oracle.default_params()
    p.eps_abs = p.eps_rel = eps
    ok, path, tr = oracle.path_optimizer_solve(p, mp, g["way_x"], g["way_y"], g["start"], g["goal"])
    ref = g[f"path1_{tag}"]
    assert ok and path.shape == ref.shape and np.abs(path - ref).max() < 1e-9
    assert tr["qp"]["iters"] == g[f"qp_{tag}"][0, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("tag,eps", [("e3", 1e-3), ("e4", 1e-4)])
def test_device_reproduces_the_reference_benchmark(g, tag, eps):
    from path_optimizer_amd import binding

    p = binding.default_params()
    p.eps_abs = p.eps_rel = eps
    eng = binding.Engine(0, p)
    eng.set_map(*_map(g))
    # BM_optimizePath: PathOptimizer::solve
    states, n, ok, stage, info = eng.plan_batch(g["way_x"][None], g["way_y"][None], g["start"][None], g["goal"][None], N=512)
    ref1 = g[f"path1_{tag}"]
    assert ok[0] == 1 and stage[0] == 0 and n[0] == len(ref1)
    assert info["status"][0] == 1 and info["iters"][0] == g[f"qp_{tag}"][0, 1] and info["n_refactor"][0] == g[f"qp_{tag}"][0, 2]
    assert np.abs(states[0, :n[0]] - ref1).max() < 1e-6, np.abs(states[0, :n[0]] - ref1).max()
    # BM_optimizePathWithoutSmoothing: the optimised path becomes the reference (setReference), bounds against the spline the first solve left
    # (reference_path_->updateBounds), vehicle error (0, 0), QP, collision check
    path1 = states[0, :n[0]]
    P = dict(ref_x=path1[None, :, 0], ref_y=path1[None, :, 1], ref_z=path1[None, :, 2], ref_s=path1[None, :, 4],
             knot_s=g[f"knot_s_{tag}"][None], knot_x=g[f"knot_x_{tag}"][None], knot_y=g[f"knot_y_{tag}"][None])
    bd, nv = eng.bounds_batch(P)
    N2 = int(nv[0])
    keep = binding.keep_control_steps(0, path1[:N2, 4])
    b = synth.Batch(0, 1, N2, keep, *(np.ascontiguousarray(path1[None, :N2, c]) for c in (0, 1, 2, 3, 4)), np.ascontiguousarray(bd[:, :N2]),
                    np.array([[0.0, 0.0, g["start"][3]]]), np.array([g["goal"][2]]))
    st2, info2, _ = eng.solve_batch(b)
    nk, ok2 = eng.postcheck_batch(st2, info2)
    ref2 = g[f"path2_{tag}"]
    assert info2["status"][0] == 1 and info2["iters"][0] == g[f"qp_{tag}"][1, 1] and info2["n_refactor"][0] == g[f"qp_{tag}"][1, 2]
    assert ok2[0] == 1 and nk[0] == len(ref2)
    assert np.abs(st2[0, :nk[0]] - ref2).max() < 1e-6, np.abs(st2[0, :nk[0]] - ref2).max()
