"""reference re-sampling (ReferencePathImpl::buildReferenceFromSpline), limits (updateLimits) and the DP lattice search
(ReferencePathSmoother::graphSearchDp) — /root/reference/src/data_struct/reference_path_impl.cpp:203-235,474-499,
/root/reference/src/reference_path_smoother/reference_path_smoother.cpp:110-300, /root/reference/src/tools/tools.cpp:34-112.

CPU: the oracle against the reference's own classes compiled through oracle/ref_shim (live where /root/reference exists, otherwise the
committed fixtures tests/golden/plan_ref.npz generated from them).  GPU: the device kernels against the oracle through the C ABI."""
import os

import numpy as np
import pytest

from path_optimizer_amd import synth

HAVE_REF = os.path.isdir("/root/reference")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "plan_ref.npz")
MAP_KW = dict(size_x=600, size_y=600, resolution=0.2, pos=(1.0, -2.0), n_obstacles=40, r_range=(0.5, 2.0))
DBL_MAX = np.finfo(np.float64).max


@pytest.fixture(scope="module")
def dmap():
    return synth.make_distance_map(3, **MAP_KW)


@pytest.fixture(scope="module")
def omap(oracle, dmap):
    dist, res, px, py, _ = dmap
    return oracle.make_map(dist, res, px, py)


def _ulp_close(a, b, ulps=4):
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


# ------------------------------------------------------------------ CPU
@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present: covered by the committed fixtures instead")
def test_oracle_matches_reference_live(oracle, omap):
    from oracle import ref_py

    p = oracle.default_params()
    sp, length, start = synth.make_search_inputs(4, 24)
    same = 0
    for b in range(24):
        ks, kx, ky = sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b]
        r = ref_py.dp_search(omap, ks, kx, ky, length[b], start[b])
        o = oracle.dp_search(p, omap, ks, kx, ky, length[b], start[b])
        assert r[0] == o[0] and r[4] == o[4]
        same += all(np.array_equal(x, y) for x, y in zip(r[1:4], o[1:4]))
        nr, rr = ref_py.resample(ks, kx, ky, length[b], 0.15, 0.3)
        no, oo = oracle.resample(p, ks, kx, ky, length[b], 0.15, 0.3)
        assert nr == no and all(_ulp_close(x, y) for x, y in zip(rr, oo))
    assert same >= 23
    far = start[0] + np.array([40.0, 40.0, 0.0])  # vehicle further than the lateral range from the spline: graphSearchDp returns false
    assert ref_py.dp_search(omap, sp["knot_s"][0], sp["knot_x"][0], sp["knot_y"][0], length[0], far)[0] == -1
    assert oracle.dp_search(p, omap, sp["knot_s"][0], sp["knot_x"][0], sp["knot_y"][0], length[0], far)[0] == -1
    assert ref_py.resample(sp["knot_s"][0], sp["knot_x"][0], sp["knot_y"][0], 0.0, 0.15, 0.3)[0] == -1
    assert oracle.resample(p, sp["knot_s"][0], sp["knot_x"][0], sp["knot_y"][0], 0.0, 0.15, 0.3)[0] == -1
    rng = np.random.default_rng(1)
    v = rng.uniform(0, 15, 200); v[:3] = [0, 0.0001, 0.00011]; a = rng.uniform(-4.5, 4.5, 200)
    rk, rkp = ref_py.limits(v, a)
    ok, okp = oracle.limits(p, v, a)
    assert np.array_equal(rk, ok, equal_nan=True) and np.array_equal(rkp, okp) and rk[0] == DBL_MAX and np.isnan(rk).any()


def test_oracle_matches_reference_fixtures(oracle, omap):
    g = np.load(GOLD)
    p = oracle.default_params()
    B = int(g["B"])
    sp, length, start = synth.make_search_inputs(int(g["seed"]), B)
    same = 0
    for b in range(B):
        ks, kx, ky = sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b]
        n, ls, lb, ub, l0 = oracle.dp_search(p, omap, ks, kx, ky, length[b], start[b])
        assert n == int(g["dp_n"][b]) and l0 == g["dp_l0"][b]
        same += np.array_equal(ls, g[f"dp_{b}"][0]) and np.array_equal(lb, g[f"dp_{b}"][1]) and np.array_equal(ub, g[f"dp_{b}"][2])
        no, oo = oracle.resample(p, ks, kx, ky, length[b], 0.15, 0.3)
        assert no == int(g["rs_n"][b]) and all(_ulp_close(x, y) for x, y in zip(oo, g[f"rs_{b}"]))
    assert same >= B - 1
    ok, okp = oracle.limits(p, g["lim_v"], g["lim_a"])
    assert np.array_equal(ok, g["lim_k"], equal_nan=True) and np.array_equal(okp, g["lim_kp"])


def test_plan_abi_symbols_and_argument_checks():
    from path_optimizer_amd import binding
    from path_optimizer_amd.abi import PO_ERR_INVALID

    L = binding.lib()
    for sym in ("po_resample_batch", "po_resample_batch_device", "po_limits_batch", "po_limits_batch_device", "po_dp_search_batch", "po_dp_search_batch_device"):
        getattr(L, sym)
    assert L.po_resample_batch(None, None, 0, 0, 0, None, None, None, None, None, None) == PO_ERR_INVALID
    assert L.po_dp_search_batch(None, None, None, 0, None, None, None, None, None) == PO_ERR_INVALID
    assert L.po_limits_batch(None, 0, 0, None, None, None, None, None) == PO_ERR_INVALID


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def engine(dmap):
    from path_optimizer_amd import binding

    e = binding.Engine(0)
    dist, res, px, py, _ = dmap
    e.set_map(dist, res, px, py)
    return e


@pytest.mark.gpu
def test_device_dp_search_matches_oracle(engine, oracle, omap):
    B = 96
    sp, length, start = synth.make_search_inputs(6, B)
    start[5] += np.array([40.0, 40.0, 0.0])  # graphSearchDp returns false
    ls, lb, ub, l0, nl = engine.dp_search_batch(sp, length, start, 64)
    p = oracle.default_params()
    ident = 0
    for b in range(B):
        n, ols, olb, oub, ol0 = oracle.dp_search(p, omap, sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b], length[b], start[b], cap=64)
        assert abs(l0[b] - ol0) < 1e-9
        if n < 0:
            assert nl[b] == n and not ls[b].any()
            continue
        assert np.abs(ls[b, :min(n, nl[b])] - ols[:min(n, nl[b])]).max() < 1e-9  # same layers (findClosestPoint + running sum)
        ok = nl[b] == n and np.abs(lb[b, :n] - olb).max() < 1e-9 and np.abs(ub[b, :n] - oub).max() < 1e-9
        ident += ok
        assert not ls[b, max(nl[b], 0):].any() and not lb[b, max(nl[b], 0):].any()
    assert nl[5] == -1
    assert ident == B - 1, ident  # same layer count and the same corridor on EVERY path (values to round-off: device sin/cos/atan2 differ from glibc in the last ulp)
    assert engine.dp_search_batch(sp, length, start, 8)[4].min() == -2  # cap too small is flagged, not truncated


@pytest.mark.gpu
def test_device_resample_and_limits_match_oracle(engine, oracle):
    B = 64
    sp, length, start = synth.make_search_inputs(7, B)
    length[3] = 0.0  # "Cannot build reference line from spline!"
    for dsm, dsl, N in ((0.15, 0.3, 256), (0.5, 1.0, 96)):
        out = engine.resample_batch(sp, length, dsm, dsl, N)
        p = oracle.default_params()
        for b in range(B):
            n, oo = oracle.resample(p, sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b], length[b], dsm, dsl, cap=N)
            assert out["n_points"][b] == n
            if n <= 0:
                continue
            for key, ov in zip(("ref_x", "ref_y", "ref_z", "ref_k", "ref_s"), oo):
                assert np.abs(out[key][b, :n] - ov).max() < 1e-9 and not out[key][b, n:].any()
    assert out["n_points"][3] == -1
    assert (np.delete(engine.resample_batch(sp, length, 0.15, 0.3, 32)["n_points"], 3) == -2).all()  # N too small is flagged, not truncated
    rng = np.random.default_rng(2)
    v = rng.uniform(0, 15, (8, 50)); v[0, :3] = [0, 0.0001, 0.00011]; a = rng.uniform(-4.5, 4.5, (8, 50))
    npts = np.array([50, 50, 20, 50, 3, 50, 50, 50], dtype=np.int32)
    mk, mkp = engine.limits_batch(v, a, npts)
    p = oracle.default_params()
    for b in range(8):
        ok, okp = oracle.limits(p, v[b], a[b])
        n = npts[b]
        assert np.array_equal(mk[b, :n], ok[:n], equal_nan=True) and np.array_equal(mkp[b, :n], okp[:n]) and not mk[b, n:].any()


@pytest.mark.gpu
def test_device_chain_search_to_post_smoothing(engine, oracle, omap):
    """graphSearchDp -> postSmooth's QP on the device, fed from one stage to the next, against the oracle doing the same."""
    B = 32
    sp, length, start = synth.make_search_inputs(8, B)
    ls, lb, ub, l0, nl = engine.dp_search_batch(sp, length, start, 64)
    keep = nl >= 4
    assert keep.sum() >= B // 2
    inp = dict(x=None, y=None, angle=None, k=None, s=ls[keep], lb=lb[keep], ub=ub[keep], l0=l0[keep], n_points=nl[keep])
    dx, _, _, info, raw = engine.smooth_batch(2, inp, want_raw=True)
    ox, _, _, oinfo, oraw = oracle.smooth_batch(2, oracle.default_params(), inp, want_raw=True)
    assert np.array_equal(info["status"], oinfo["status"])
    same = info["iters"] == oinfo["iters"]
    assert same.mean() >= 0.98 and np.abs(dx[same] - ox[same]).max() < 1e-7
    if (~same).any():
        assert np.abs(dx[~same] - ox[~same]).max() < 1e-2
