"""Post-solve step (collision check + truncation rule of PathOptimizer::optimizePath).
CPU: the C restatement against the reference's own collision_checker.cpp / car_geometry.cpp / Map.cpp / tools.cpp (live where
/root/reference exists, and through the committed fixture tests/golden/post_ref.npz everywhere).  GPU: the HIP kernels through the
C ABI against the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_post_golden as G  # noqa: E402
from path_optimizer_amd import synth  # noqa: E402
from path_optimizer_amd.abi import INFO_DTYPE  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post_ref.npz")


@pytest.fixture(scope="module")
def scene(oracle):
    d, res, px, py, discs = synth.make_distance_map(**G.MAP_ARGS)
    return dict(d=d, res=res, px=px, py=py, discs=discs, m=oracle.make_map(d, res, px, py), states=G.post_cases())


def _solved(B):
    info = np.zeros(B, dtype=INFO_DTYPE)
    info["status"] = 1
    return info


def _with_s(states):
    """arc length exactly as the output map / optimizePath accumulate it"""
    st = states.copy()
    for b in range(st.shape[0]):
        s = 0.0
        for i in range(1, st.shape[1]):
            dx, dy = st[b, i, 0] - st[b, i - 1, 0], st[b, i, 1] - st[b, i - 1, 1]
            s += np.sqrt(dx * dx + dy * dy)
            st[b, i, 4] = s
    return st


def test_map_interpolates_the_distance_field(oracle, scene):
    rng = np.random.default_rng(0)
    lx, ly = scene["d"].shape[0] * scene["res"], scene["d"].shape[1] * scene["res"]
    xy = np.stack([rng.uniform(scene["px"] - 0.49 * lx, scene["px"] + 0.49 * lx, 4000), rng.uniform(scene["py"] - 0.49 * ly, scene["py"] + 0.49 * ly, 4000)], axis=1)
    dist, inside = oracle.map_distance(scene["m"], xy)
    assert inside.all()
    discs = scene["discs"]
    true = np.maximum(np.min(np.hypot(xy[:, 0:1] - discs[:, 0], xy[:, 1:2] - discs[:, 1]) - discs[:, 2], axis=1), 0)
    assert np.abs(dist - true).max() < 0.08  # bilinear on a 0.2 m grid
    out = np.array([[scene["px"] + 0.5 * lx + 0.01, scene["py"]], [scene["px"], scene["py"] - 0.5 * ly - 0.01], [1e9, 0.0]])
    dist, inside = oracle.map_distance(scene["m"], out)
    assert not inside.any() and (dist == 0).all()  # Map.cpp:20-21
    # the four cell centres around a point reproduce the stored values exactly
    i, j = 17, 230
    cx = scene["px"] + 0.5 * lx - (i + 0.5) * scene["res"]; cy = scene["py"] + 0.5 * ly - (j + 0.5) * scene["res"]
    dist, _ = oracle.map_distance(scene["m"], [[cx, cy]])
    assert abs(dist[0] - float(scene["d"][i, j])) < 1e-6


def test_oracle_matches_reference_fixture(oracle, scene):
    g = np.load(GOLD)
    p = oracle.default_params()
    st = _with_s(scene["states"])
    free = np.array([[oracle.collision_free(p, scene["m"], *st[b, i, :3]) for i in range(st.shape[1])] for b in range(st.shape[0])], dtype=np.int8)
    assert np.array_equal(free, g["free"])
    nv, ok = oracle.postcheck_batch(p, scene["m"], st, _solved(st.shape[0]))
    assert np.array_equal(nv, g["n_valid"]) and np.array_equal(ok, g["ok"])
    assert 0 < ok.mean() < 1 and (nv < st.shape[1]).any()  # the scene exercises all three outcomes
    info = _solved(st.shape[0]); info["status"][::3] = -2
    nv2, ok2 = oracle.postcheck_batch(p, scene["m"], st, info)
    assert (nv2[::3] == 0).all() and (ok2[::3] == 0).all() and np.array_equal(nv2[1::3], nv[1::3])  # "QP failed." -> false
    p.enable_collision_check = 0
    nv3, ok3 = oracle.postcheck_batch(p, scene["m"], st, _solved(st.shape[0]))
    assert (nv3 == st.shape[1]).all() and (ok3 == 1).all()


def test_oracle_matches_reference_live(oracle, scene):
    ref_py = pytest.importorskip("oracle.ref_py")
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference tree not present")
    rng = np.random.default_rng(5)
    p = oracle.default_params()
    m = scene["m"]
    for _ in range(1500):
        x, y, z = rng.uniform(-65, 65), rng.uniform(-65, 65), rng.uniform(-4, 4)
        assert oracle.collision_free(p, m, x, y, z) == ref_py.collision_free(m, x, y, z)
        d, _ = oracle.map_distance(m, [[x, y]])
        assert d[0] == ref_py.map_distance(m, x, y)
    st = scene["states"]
    for b in range(0, st.shape[0], 5):
        ok, nv, s = ref_py.postcheck(m, st[b])
        stb = _with_s(st[b:b + 1])
        assert np.array_equal(stb[0, :max(nv, 1), 4], s[:max(nv, 1)])  # same running arc length, bit for bit
        nvo, oko = oracle.postcheck_batch(p, m, stb, _solved(1))
        assert (nvo[0], oko[0]) == (nv, ok)


# ------------------------------------------------------------------ GPU ------------------------------------------------------------------
@pytest.fixture(scope="module")
def binding():
    from path_optimizer_amd import binding as b

    b.lib()
    return b


@pytest.mark.gpu
def test_device_map_sampling_matches_oracle(binding, oracle, scene):
    eng = binding.Engine(0)
    with pytest.raises(binding.PoError):
        eng.map_sample(np.zeros((1, 2)))  # no map yet
    eng.set_map(scene["d"], scene["res"], scene["px"], scene["py"])
    rng = np.random.default_rng(2)
    xy = np.stack([rng.uniform(-70, 70, 20000), rng.uniform(-70, 70, 20000)], axis=1)  # inside, near the edges and outside
    d, ins = eng.map_sample(xy)
    od, oins = oracle.map_distance(scene["m"], xy)
    assert np.array_equal(ins, oins)
    assert np.array_equal(d, od)  # same arithmetic, float result: bit-exact


@pytest.mark.gpu
def test_device_postcheck_matches_oracle_and_fixture(binding, oracle, scene):
    g = np.load(GOLD)
    eng = binding.Engine(0)
    eng.set_map(scene["d"], scene["res"], scene["px"], scene["py"])
    st = _with_s(scene["states"])
    B = st.shape[0]
    nv, ok = eng.postcheck_batch(st, _solved(B))
    # index / decision outputs are exact: the fixture was produced by the reference's own collision checker (glibc sin / cos), the device uses the portable
    # sin / cos of include/po_pmath.h (within one ulp of glibc) — on this scene no clearance sits within an ulp of a radius, so every decision is the same
    assert np.array_equal(nv, g["n_valid"]) and np.array_equal(ok, g["ok"]), (nv, g["n_valid"])
    info = _solved(B); info["status"][::4] = -3
    npts = np.full(B, st.shape[1], dtype=np.int32); npts[1::4] = 57
    nv2, ok2 = eng.postcheck_batch(st, info, npts)
    with oracle.portable_math():  # the oracle on the device's own IEEE operation sequence (include/po_pmath.h): bit-identical decisions
        onv, ook = oracle.postcheck_batch(oracle.default_params(), scene["m"], st, info, npts)
    assert np.array_equal(nv2, onv) and np.array_equal(ok2, ook)
    onv_g, ook_g = oracle.postcheck_batch(oracle.default_params(), scene["m"], st, info, npts)  # glibc mode (the mode pinned against the reference): same decisions here
    assert np.array_equal(nv2, onv_g) and np.array_equal(ok2, ook_g)
    assert (nv2[::4] == 0).all() and (ok2[::4] == 0).all() and (nv2[1::4] <= 57).all()
    p = binding.default_params(); p.enable_collision_check = 0
    nv3, ok3 = binding.Engine(0, p).postcheck_batch(st, _solved(B))
    assert (nv3 == st.shape[1]).all() and (ok3 == 1).all()


@pytest.mark.gpu
def test_solve_then_postcheck_on_device(binding, oracle, scene):
    """The intended pipeline: solve on the device, check the outputs where they lie (device pointers), only n_valid/ok come back."""
    import torch

    batch = synth.make_batch(3, B=64)
    db = binding.DeviceBatch(batch)
    eng = binding.Engine(0)
    eng.set_map(scene["d"], scene["res"], scene["px"], scene["py"])
    eng.solve_batch_device(db)
    nv = torch.zeros(batch.B, dtype=torch.int32, device="cuda"); ok = torch.zeros_like(nv)
    eng.postcheck_batch_device(db, nv, ok)
    torch.cuda.synchronize()
    states = db.out_states.cpu().numpy(); info = db.info_numpy()
    with oracle.portable_math():
        onv, ook = oracle.postcheck_batch(oracle.default_params(), scene["m"], states, info)
    assert np.array_equal(nv.cpu().numpy(), onv) and np.array_equal(ok.cpu().numpy(), ook)
    assert (onv < batch.N).any() and (onv == batch.N).any()
