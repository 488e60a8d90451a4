"""GPU parity tests (run with -m gpu on the MI355X box).  Everything goes through the C ABI
(libpo_hip.so via path_optimizer_amd.binding); the oracle is only the checker."""
import numpy as np
import pytest

import np_twin as T
from path_optimizer_amd import synth

pytestmark = pytest.mark.gpu
FORMS = [(T.PO_KP, "KP"), (T.PO_KPC, "KPC"), (T.PO_K, "K")]


@pytest.fixture(scope="module")
def binding():
    from path_optimizer_amd import binding as b

    b.lib()
    return b


def _compare_every_path(info, oinfo, xs, oxs, st, ost, min_same, eps=1e-4, check_every=25):
    """Every path is compared: same iteration count -> round-off level (1e-6); a path whose residual sat within round-off of eps at a termination
    check stops one check earlier / later on one side -> compared at 10 x eps, and the counts may differ by exactly one check interval."""
    assert np.array_equal(info["status"], oinfo["status"])
    same = info["iters"] == oinfo["iters"]
    assert same.mean() >= min_same, (info["iters"], oinfo["iters"])
    assert np.abs(xs - oxs)[same].max() < 1e-6 and np.abs(st - ost)[same].max() < 1e-6
    if (~same).any():
        assert (np.abs(info["iters"].astype(int) - oinfo["iters"].astype(int))[~same] == check_every).all(), (info["iters"][~same], oinfo["iters"][~same])
        assert np.abs(xs - oxs)[~same].max() < 10 * eps and np.abs(st - ost)[~same].max() < 10 * eps
    return same


def _rand_batch(form, B, N, ds=0.25, seed=0, narrow=False):
    rng = np.random.default_rng(seed)
    insts = [T.random_instance(rng, N, ds=ds, narrow=narrow) for _ in range(B)]
    st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    keep = 1 if form == T.PO_K else (4 if form == T.PO_KPC else None)
    return synth.Batch(form, B, N, keep or 4, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"),
                       st("x0"), np.array([i["goal_z"] for i in insts]), st("max_k") if form == T.PO_KPC else None,
                       st("max_kp") if form == T.PO_KPC else None)


@pytest.mark.parametrize("form,name", FORMS)
@pytest.mark.parametrize("N", [2, 5, 23, 64, 130])
def test_device_assembly_matches_oracle(binding, oracle, form, name, N):
    b = _rand_batch(form, 3, N, seed=N + form)
    eng = binding.Engine(0)
    l, u, dyn = eng.assemble_batch(b)
    p = oracle.default_params()
    for i in range(b.B):
        P, A, lo, uo = oracle.assemble(form, p, N, b.keep, b.ref_k[i], b.ref_s[i], b.ref_z[i, -1], b.bounds[i], b.x0[i], b.goal_z[i],
                                       None if b.max_k is None else b.max_k[i], None if b.max_kp is None else b.max_kp[i])
        if form == T.PO_K:  # device atan/cos may differ from glibc in the last ulp
            np.testing.assert_allclose(l[i], lo, rtol=4e-16, atol=0)
            np.testing.assert_allclose(u[i], uo, rtol=4e-16, atol=0)
        else:
            assert np.array_equal(l[i], lo) and np.array_equal(u[i], uo)  # bit-exact
        Ad = A.toarray()
        for t in range(N - 1):
            if form == T.PO_K:
                ref = [Ad[2 * (t + 1), 2 * t + 1], Ad[2 * (t + 1) + 1, 2 * t], Ad[2 * (t + 1), 2 * N + t]]
                np.testing.assert_allclose(dyn[i, t], ref, rtol=4e-16)
            else:
                ref = [Ad[3 * (t + 1), 3 * t + 1], Ad[3 * (t + 1) + 1, 3 * t], Ad[3 * (t + 1) + 2, 3 * N + t // b.keep]]
                assert np.array_equal(dyn[i, t], ref)


@pytest.mark.parametrize("form,name", FORMS)
def test_device_scaling_block_matches_oracle(binding, oracle, form, name):
    """The per-path equilibration the kernel consumes equals the oracle's class-level Ruiz factors."""
    for N, ds in ((40, 0.25), (77, 0.3), (30, 1.0)):
        b = _rand_batch(form, 2, N, ds=ds, seed=N)
        if form == T.PO_KP:
            b.keep = binding.keep_control_steps(form, b.ref_s[0])
        p = binding.default_params()
        blk = binding.Engine(0, p).scaling_batch(b)
        po = oracle.default_params()
        ds_nom = np.max(np.diff(b.ref_s[0])[:9])
        D, E, c = oracle.class_scaling(form, po, N, b.keep, ds_nom, 10)
        n, m, C = oracle.dims(form, N, b.keep)
        if form == T.PO_K:
            dv = [D[3], D[2], D[2 * N + 1], D[3 * N]]          # e_y, e_phi, delta, S
            er_dyn = [E[2], E[3]]
            er_loc = [E[2 * N + 2], E[2 * N + 3], E[4 * N + 1], E[5 * N], E[6 * N - 1 + 3], E[6 * N - 1 + 4], E[6 * N - 1 + 5], E[9 * N], E[10 * N]]
        else:
            dv = [D[3], D[4], D[5], D[3 * N + C + 1]]
            er_dyn = [E[3], E[4], E[5]]
            if form == T.PO_KP:
                cb = 5 * N + C
                er_loc = [E[3 * N + 1], E[4 * N + C + 1], E[cb + 2], E[cb + 3], E[cb + 2 * N + 1], E[cb + 3 * N + 1], E[cb + 4 * N + 1], E[cb + 5 * N + 1]]
            else:
                sb, cb = 5 * N + 2 * C, 7 * N + 3 * C
                er_loc = [E[3 * N + 1], E[4 * N + 1], E[sb + 1], E[sb + N + 1], E[cb + 3], E[cb + 4], E[cb + 5], E[cb + 3 * N + 1], E[cb + 4 * N + 1]]
        er = np.array(er_loc + er_dyn)
        for i in range(b.B):
            np.testing.assert_allclose(blk[i, 63], c, rtol=1e-13)
            np.testing.assert_allclose(blk[i, 24:24 + len(er)], er, rtol=1e-13)
            np.testing.assert_allclose(blk[i, 0:len(er)], er ** 2 / c, rtol=1e-13)
            np.testing.assert_allclose(blk[i, 48:52], po.sigma / (c * np.array(dv) ** 2), rtol=1e-13)
            np.testing.assert_allclose(blk[i, 56:60], c * np.array(dv), rtol=1e-13)
            if form == T.PO_K:  # first / last steering variable and their box rows: own Ruiz factors (variable class 6, row class 11, alternate block)
                for dend, eend in ((D[2 * N], E[4 * N]), (D[3 * N - 2], E[5 * N - 2])):
                    np.testing.assert_allclose(blk[i, 48 + 6], po.sigma / (c * dend ** 2), rtol=1e-13)
                    np.testing.assert_allclose(blk[i, 56 + 6], c * dend, rtol=1e-13)
                    np.testing.assert_allclose(blk[i, 24 + 11], eend, rtol=1e-13)
                alt = np.array(er_loc); alt[2] = E[4 * N]
                np.testing.assert_allclose(blk[i, 24 + 12:24 + 21], alt, rtol=1e-13)
                np.testing.assert_allclose(blk[i, 12:21], alt ** 2 / c, rtol=1e-13)
                assert D[2 * N] != D[2 * N + 1] and E[4 * N] != E[4 * N + 1]


@pytest.mark.parametrize("form,name", FORMS)
@pytest.mark.parametrize("scaling", [0, 10])
def test_fixed_iteration_iterates_match_oracle(binding, oracle, form, name, scaling):
    """Same ADMM, same number of iterations, no termination test: iterates must agree to round-off."""
    b = _rand_batch(form, 6, 50, seed=3 + form, narrow=True)
    for iters, adapt in ((1, 0), (2, 0), (40, 0), (120, 50)):
        p = binding.default_params()
        p.max_iter, p.check_every, p.adapt_every, p.scaling = iters, 0, adapt, scaling
        po = oracle.default_params()
        po.max_iter, po.check_every, po.adapt_every, po.scaling = iters, 0, adapt, -scaling
        eng = binding.Engine(0, p)
        st, info, xs = eng.solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, po)
        assert (info["iters"] == iters).all() and (oinfo["iters"] == iters).all()
        assert np.array_equal(info["n_refactor"], oinfo["n_refactor"])
        np.testing.assert_allclose(info["rho"], oinfo["rho"], rtol=1e-8)
        assert np.abs(xs - oxs).max() < 1e-8, (name, iters, np.abs(xs - oxs).max())
        np.testing.assert_allclose(info["r_prim"], oinfo["r_prim"], rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(info["r_dual"], oinfo["r_dual"], rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(info["obj"], oinfo["obj"], rtol=1e-6, atol=1e-12)  # 0.5 x'Px at exit (OSQP's info.obj_val)


@pytest.mark.parametrize("cfg,B", [(1, 1), (2, 48), (3, 96), (5, 16)])
def test_baseline_configs_match_oracle(binding, oracle, cfg, B):
    """BASELINE configs at the project's termination (eps 1e-4): same iteration counts, same solution."""
    b = synth.make_batch(cfg, B=B)
    eng = binding.Engine(0)
    st, info, xs = eng.solve_batch(b, want_x=True)
    ost, oinfo, oxs = oracle.solve_batch(b, oracle.device_equivalent_params())
    same = _compare_every_path(info, oinfo, xs, oxs, st, ost, 0.98)
    ey = (xs - oxs)[:, 0:3 * b.N:3]
    rms = np.sqrt((ey ** 2).mean(axis=1))
    assert rms[same].max() < 1e-6 and rms.max() < 1e-4, rms  # bar: <= 1e-4 m lateral-offset RMS vs the oracle at identical settings; measured ~1e-9
    np.testing.assert_allclose(info["obj"][same], oinfo["obj"][same], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("form,name", FORMS)
def test_device_matches_literal_ruiz(binding, oracle, form, name):
    """The device's class-level equilibration against the oracle running OSQP's Ruiz passes LITERALLY on the assembled P and A
    (scaling = +10, the reference's setting): same iteration counts and the same solution for every formulation — including K, whose
    first and last steering variable get their own factors."""
    import copy

    b = copy.copy(synth.make_batch(3, B=12)); b.formulation = form
    if form == T.PO_K:
        b.keep = 1
    if form == T.PO_KPC:
        b.max_k = np.full((b.B, b.N), 0.2); b.max_kp = np.full((b.B, b.N), 0.05)
    st, info, xs = binding.Engine(0).solve_batch(b, want_x=True)
    po = oracle.default_params(); assert po.scaling == 10
    ost, oinfo, oxs = oracle.solve_batch(b, po)
    assert (info["status"] == 1).all() and np.array_equal(info["status"], oinfo["status"])
    _compare_every_path(info, oinfo, xs, oxs, st, ost, 0.98)


@pytest.mark.parametrize("form,name", [(T.PO_KP, "KP"), (T.PO_KPC, "KPC")])
def test_mixed_uniform_and_general_paths(binding, oracle, form, name):
    """The two-level mapping is two launches: paths whose row classes are the same on every stage go through the uniform-class
    variant, the others are deferred to the general one (po_fast.inc, solve_kernel_fast).  A batch that mixes both — free rows
    (infinite clearances) and equality rows (lb == ub) on some stages of every other path — against the oracle."""
    import copy

    b = copy.copy(synth.make_batch(3, B=16)); b.formulation = form
    b.bounds = b.bounds.copy()
    if form == T.PO_KPC:
        b.max_k = np.full((b.B, b.N), 0.2); b.max_kp = np.full((b.B, b.N), 0.05)
    for i in range(1, b.B, 2):
        b.bounds[i, 30:45, 0, :] = (-1e30, 1e30)       # circle 0 unconstrained on 15 stages: free rows (rho_min)
        if i % 4 == 1:
            b.bounds[i, 100:104, 2, :] = 0.05            # circle 2 pinned: lb == ub -> equality rows (1e3 rho)
    st, info, xs = binding.Engine(0).solve_batch(b, want_x=True)
    ost, oinfo, oxs = oracle.solve_batch(b, oracle.device_equivalent_params())
    assert np.array_equal(info["status"], oinfo["status"]), (info["status"], oinfo["status"])
    same = _compare_every_path(info, oinfo, xs, oxs, st, ost, 0.98)
    assert same[0::2].sum() >= 7 and same[1::2].sum() >= 6  # both kinds of path are covered by the comparison


def test_keep_quirk_and_ragged_sizes(binding, oracle):
    """ds = 0.3 -> keep = 3 (truncation quirk); N not a multiple of 64 or of keep."""
    for N, ds in ((7, 0.3), (65, 0.3), (127, 0.5), (200, 2.0), (90, 0.2), (150, 0.15), (300, 0.25), (511, 0.3)):
        b = _rand_batch(T.PO_KP, 2, N, ds=ds, seed=N)
        b.keep = binding.keep_control_steps(T.PO_KP, b.ref_s[0])
        assert b.keep == oracle.keep_steps(T.PO_KP, b.ref_s[0])
        p = binding.default_params(); p.max_iter, p.check_every, p.adapt_every = 60, 0, 0
        po = oracle.device_equivalent_params(); po.max_iter, po.check_every, po.adapt_every = 60, 0, 0
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, po)
        assert np.abs(xs - oxs).max() < 1e-8, (N, ds, np.abs(xs - oxs).max())


@pytest.mark.parametrize("keep", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20])
def test_every_keep_control_steps_value(binding, oracle, keep):
    """keep_control_steps_ = int(1.2 / spacing) takes every value 1..8 in the reference's own pipeline (spacing 0.15..1.0 m,
    path_optimizer.cpp:171-172); larger values through the API (9 .. 16: the wide role-split shapes, N <= 32 keep; beyond: the single-level chain).
    Fixed-iteration iterates, then the full run (termination, adaptive rho, infeasibility certificate) against the oracle, for sizes that are / are not multiples of keep."""
    ds = 1.2 / keep * 0.999
    for N in (keep + 2, 41, 97 if keep == 1 else 150) + ((32 * keep - 1, 32 * keep + 1) if 9 <= keep <= 12 else ()):  # (9 .. 12: either side of the one-wave limit of the wide shapes)
        b = _rand_batch(T.PO_KP, 3, N, ds=ds, seed=keep * 100 + N, narrow=True)
        b.keep = keep
        assert binding.keep_control_steps(T.PO_KP, b.ref_s[0]) == keep
        p = binding.default_params(); p.max_iter, p.check_every, p.adapt_every = 60, 0, 25
        po = oracle.device_equivalent_params(); po.max_iter, po.check_every, po.adapt_every = 60, 0, 25
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, po)
        assert np.abs(xs - oxs).max() < 1e-8, (keep, N, np.abs(xs - oxs).max())
        assert np.array_equal(info["n_refactor"], oinfo["n_refactor"])
        if N > 20:  # one path gets a corridor that jumps sideways within one step: primal infeasible
            j = N // 2
            b.bounds[1, j, :, :] = [0.9, 1.0]
            b.bounds[1, j + 1, :, :] = [-1.0, -0.9]
        st, info, xs = binding.Engine(0).solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, oracle.device_equivalent_params())
        assert np.array_equal(info["status"], oinfo["status"]), (keep, N, info, oinfo)
        assert np.array_equal(info["iters"], oinfo["iters"]), (keep, N, info["iters"], oinfo["iters"])
        ok = info["status"] == 1
        # 1e-5, not 1e-6: on the tiny instances (N = keep + 2) the residuals reach round-off level before the last rho adaption, whose
        # estimate sqrt(r_prim / r_dual) then amplifies last-bit differences between the two implementations (measured 0.9e-6 .. 1.4e-6
        # on one N = 4 path, 1e-11 elsewhere; iterates agree to 1e-15 up to that refactorisation)
        assert np.abs(xs - oxs)[ok].max() < 1e-5


@pytest.mark.parametrize("keep", [4, 3, 2, 1])
def test_scan_row_and_wave_boundaries(binding, oracle, keep):
    """Path lengths that put the number of chunks just below / on / above the boundaries of the scan structure: the 16-lane DPP rows of the
    one-wave scan (16, 32, 48 chunks), the wave (64), the 8-lane segments and the second wave of the segmented scan (65 ... 128), ragged too."""
    ds = 1.2 / keep * 0.999
    sizes = sorted({keep * c + d for c in (15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 72, 73, 96, 127, 128) for d in (0, 1) if keep * c + d <= 512 and keep * c + d >= 3})
    sizes = [n for n in sizes if n <= (256 if keep == 1 else 512)]
    for N in sizes:
        b = _rand_batch(T.PO_KP, 2, N, ds=ds, seed=7 * N + keep)
        b.keep = keep
        p = binding.default_params(); p.max_iter, p.check_every, p.adapt_every = 40, 0, 25
        po = oracle.device_equivalent_params(); po.max_iter, po.check_every, po.adapt_every = 40, 0, 25
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, po)
        assert np.abs(xs - oxs).max() < 1e-8, (keep, N, np.abs(xs - oxs).max())
        assert np.array_equal(info["n_refactor"], oinfo["n_refactor"])
    # ragged: the longest path sizes the block, the shorter ones end inside a row / a wave
    N = keep * 66 if keep * 66 <= 256 or keep > 1 else 256
    b = _rand_batch(T.PO_KP, 4, N, ds=ds, seed=99 + keep)
    b.keep = keep
    b.n_points = np.array([N, keep * 16 + 1, keep * 33, max(keep + 2, 5)], dtype=np.int32)
    p = binding.default_params(); p.max_iter, p.check_every, p.adapt_every = 40, 0, 25
    po = oracle.device_equivalent_params(); po.max_iter, po.check_every, po.adapt_every = 40, 0, 25
    st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
    ost, oinfo, oxs = oracle.solve_batch(b, po)
    assert np.abs(xs - oxs).max() < 1e-8 and np.abs(st - ost).max() < 1e-8


@pytest.mark.parametrize("form,name", [(T.PO_KPC, "KPC"), (T.PO_K, "K")])
def test_scan_boundaries_other_formulations(binding, oracle, form, name):
    """KPC / K at the same structural boundaries (uniform variants with their per-stage patches: K's end stages and final stage, KPC's
    curvature-slack box row), fixed iteration count, incl. a rho adaption."""
    for N in (63, 64, 65, 127, 128, 129, 130, 255, 256, 257, 300):
        b = _rand_batch(form, 2, N, seed=31 * N + form)
        if form == T.PO_KPC:  # make the class of the curvature-slack box row change along the path (max_k above / below kappa_max)
            b.max_k = b.max_k.copy(); b.max_k[:, ::3] = 0.5
        p = binding.default_params(); p.max_iter, p.check_every, p.adapt_every = 40, 0, 25
        po = oracle.device_equivalent_params(); po.max_iter, po.check_every, po.adapt_every = 40, 0, 25
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, po)
        assert np.abs(xs - oxs).max() < 1e-8, (name, N, np.abs(xs - oxs).max())
        assert np.array_equal(info["n_refactor"], oinfo["n_refactor"])


def test_api_errors_and_empty(binding):
    eng = binding.Engine(0)
    b = _rand_batch(T.PO_KP, 2, 10)
    b.B = 0
    st, info, xs = eng.solve_batch(b)
    assert st.shape[0] == 0
    b = _rand_batch(T.PO_KP, 1, 10)
    b.formulation = 7
    with pytest.raises(binding.PoError):
        eng.solve_batch(b)
    b = _rand_batch(T.PO_KPC, 1, 10)
    b.keep = 3
    with pytest.raises(binding.PoError):
        eng.solve_batch(b)
    b = _rand_batch(T.PO_KP, 1, 10)
    b.max_k = None
    b.formulation = T.PO_KPC
    with pytest.raises(binding.PoError):
        eng.solve_batch(b)


def test_infeasible_corridor_certificate(binding, oracle):
    """A corridor that jumps 1.9 m sideways in one 0.25 m step: OSQP's primal-infeasibility certificate fires on the
    device at the same termination check as in the oracle; neighbours in the batch are unaffected."""
    for form in (T.PO_KP, T.PO_KPC, T.PO_K):
        b = _rand_batch(form, 3, 30, seed=5)
        b.bounds[1, 10, :, :] = [0.9, 1.0]
        b.bounds[1, 11, :, :] = [-1.0, -0.9]
        st, info, xs = binding.Engine(0).solve_batch(b)
        ost, oinfo, oxs = oracle.solve_batch(b, oracle.device_equivalent_params())
        assert np.array_equal(info["status"], oinfo["status"]) and info["iters"][1] == oinfo["iters"][1], (form, info, oinfo)
        if form != T.PO_KPC:  # (in KPC OSQP's certificate does not fire within max_iter either: both end as MAX_ITER)
            assert info["status"][1] == -3
        assert info["status"][0] == 1 and info["status"][2] == 1 and np.array_equal(info["iters"][[0, 2]], oinfo["iters"][[0, 2]])


def test_full_size_properties(binding):
    """BASELINE config 3 at full size (B=4096, N=200) through size-independent properties."""
    small = synth.make_batch(3, B=256)
    big = synth.replicate(small, 4096)
    eng = binding.Engine(0)
    st, info, xs = eng.solve_batch(big, want_x=True)
    assert (info["status"] == 1).mean() > 0.98  # a handful of near-degenerate corridors hit max_iter (as on the oracle)
    ok = info["status"] == 1
    # determinism / independence of batch position: replicas are bit-identical
    assert np.array_equal(xs[:256], xs[256:512]) and np.array_equal(xs[:256], xs[3840:])
    # solution satisfies the hard corridor rows and the curvature box within the primal tolerance
    N = 200
    p = binding.default_params()
    ey, ephi, k = xs[:, 0:3 * N:3], xs[:, 1:3 * N:3], xs[:, 2:3 * N:3]
    for d, c in ((p.d[0], 0), (p.d[2], 2)):
        val = ey + d * ephi
        assert (val[ok] <= big.bounds[ok][:, :, c, 1] + 2e-3).all() and (val[ok] >= big.bounds[ok][:, :, c, 0] - 2e-3).all()
    assert np.abs(k[ok]).max() <= np.tan(p.max_steer) / p.wheel_base + 2e-3
    # initial state pinned, arc length monotone
    assert np.abs(ey[ok, 0] - big.x0[ok, 0]).max() < 2e-3
    assert (np.diff(st[:, :, 4], axis=1) > 0).all()


def test_host_side_cpp_mirror(binding):
    """The C++ OsqpSolver mirror (path_optimizer_amd/host) driven like path_optimizer.cpp:182-183."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "path_optimizer_amd", "host")], stdout=subprocess.DEVNULL)
    for form in ("KP", "KPC", "K"):
        r = subprocess.run([os.path.join(root, "path_optimizer_amd", "host", "host_test"), form, "60", "3"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "single ok=1" in r.stdout and "batch rc=0" in r.stdout and "create(KCP)=nullptr" in r.stdout
        assert r.stdout.count("status=1") == 3
        assert "check free=1 hit=1" in r.stdout and "map stages FAILED" not in r.stdout  # Map / updateBounds / CollisionChecker mirrors
        assert "top-level ok" in r.stdout  # PathOptimizer(start, end, map).solve(points, &path) / solveWithoutSmoothing
        assert "smoothing stages ok" in r.stdout  # TensionSmoother2 / graphSearchDp / postSmooth / buildReferenceFromSpline / updateLimits mirrors


def test_host_side_cpp_multi_device_solve_batch(binding):
    """OsqpSolver::solveBatch over several engines (contiguous shards, one host thread + handle + stream per device, results in place) returns
    states and po_info bit-identical to the one-engine call: G = hipGetDeviceCount() engines (1 on a single-GPU box, 8 on a full node), and — so that the split,
    the threads and the in-place writes run on a single-GPU box too — 1 / 3 / 5 engines on device 0 with a batch that does not divide evenly."""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "path_optimizer_amd", "host")], stdout=subprocess.DEVNULL)
    exe = os.path.join(root, "path_optimizer_amd", "host", "host_test")
    for form, nb, shards in (("KP", "13", None), ("KP", "13", "3"), ("KPC", "7", "5"), ("K", "4", "8"), ("KP", "2", "1")):
        env = dict(os.environ)
        env.pop("PO_HOST_TEST_SHARDS", None)
        if shards:
            env["PO_HOST_TEST_SHARDS"] = shards
        r = subprocess.run([exe, form, "60", nb], capture_output=True, text=True, env=env)
        m = re.search(r"multi-device rc=(-?\d+) engines=(\d+) \(visible devices (\d+)\) bit_identical=(\d)", r.stdout)
        assert m, r.stdout + r.stderr
        assert m.group(1) == "0" and m.group(4) == "1", r.stdout
        assert int(m.group(2)) == max(int(m.group(3)), int(shards or 0)), r.stdout
        assert r.stdout.count("device shard") == int(m.group(2))
        assert r.returncode == 0, r.stdout + r.stderr


def test_deterministic_and_device_pointer_entry(binding):
    """Same inputs twice -> bit-identical outputs; the device-pointer entry (inputs resident in HBM, caller's stream)
    returns exactly what the host-pointer entry returns."""
    import torch

    b = synth.make_batch(3, B=48)
    eng = binding.Engine(0)
    st1, info1, xs1 = eng.solve_batch(b, want_x=True)
    st2, info2, xs2 = eng.solve_batch(b, want_x=True)
    assert np.array_equal(xs1, xs2) and np.array_equal(st1, st2) and np.array_equal(info1["iters"], info2["iters"])
    db = binding.DeviceBatch(b, want_x=True)
    s = torch.cuda.Stream()
    eng.set_stream(s.cuda_stream)
    eng.solve_batch_device(db)
    s.synchronize()
    assert eng.last_kernel_ms() > 0
    assert np.array_equal(db.out_x.cpu().numpy(), xs1) and np.array_equal(db.out_states.cpu().numpy(), st1)
    assert np.array_equal(db.info_numpy()["iters"], info1["iters"])
    eng.set_stream(None)


@pytest.mark.parametrize("form,name", FORMS)
def test_tiny_and_unsupported_sizes(binding, oracle, form, name):
    """N = 2 and N = 3 (one or two transitions) solve like the oracle; sizes beyond the on-chip tile are refused with
    PO_ERR_UNSUPPORTED instead of being computed somewhere else."""
    for N in (2, 3):
        b = _rand_batch(form, 2, N, seed=N)
        p = binding.default_params(); p.max_iter, p.check_every, p.adapt_every = 30, 0, 0
        po = oracle.device_equivalent_params(); po.max_iter, po.check_every, po.adapt_every = 30, 0, 0
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, po)
        assert np.abs(xs - oxs).max() < 1e-8
    big = _rand_batch(form, 1, 1200, seed=1)
    with pytest.raises(binding.PoError, match="unsupported"):
        binding.Engine(0).solve_batch(big)


@pytest.mark.parametrize("form,name", FORMS)
def test_ragged_batch(binding, oracle, form, name):
    """Paths of different lengths in one launch (po_batch_in.n_points): every path equals its own single-path solve."""
    N = 150
    b = _rand_batch(form, 6, N, seed=11 + form)
    b.n_points = np.array([150, 2, 37, 149, 64, 101], dtype=np.int32)
    st, info, xs = binding.Engine(0).solve_batch(b, want_x=True)
    ost, oinfo, oxs = oracle.solve_batch(b, oracle.device_equivalent_params())
    assert np.array_equal(info["status"], oinfo["status"]) and np.array_equal(info["iters"], oinfo["iters"]), (info, oinfo)
    assert np.abs(xs - oxs).max() < 1e-6 and np.abs(st - ost).max() < 1e-6
    for i, n in enumerate(b.n_points):
        assert not st[i, n:].any()  # rows beyond the path's length are zero
    bad = _rand_batch(form, 2, 20, seed=1)
    bad.n_points = np.array([20, 21], dtype=np.int32)
    with pytest.raises(binding.PoError):
        binding.Engine(0).solve_batch(bad)


@pytest.mark.parametrize("form,name", FORMS)
def test_frozen_tight_optima_on_device(binding, oracle, form, name):
    """The device ADMM driven to eps 1e-9 lands on the frozen KKT-certified optima (tests/golden/tight_*.npz) — a check that
    does not go through the oracle's ADMM at all — and at eps 1e-4 its gap to them equals the oracle's."""
    from test_oracle import _tight_batch

    b, xg, ey = _tight_batch(form)
    p = binding.default_params(); p.eps_abs = p.eps_rel = 1e-9; p.max_iter = 200000
    st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
    assert (info["status"] == 1).all(), info
    assert np.abs(xs - xg).max() < 1e-6, np.abs(xs - xg).max()
    st, info, xs = binding.Engine(0).solve_batch(b, want_x=True)
    ost, oinfo, oxs = oracle.solve_batch(b, oracle.device_equivalent_params())
    assert np.array_equal(info["iters"], oinfo["iters"]) and (info["status"] == 1).all()
    rms = np.sqrt((((xs - xg)[:, ey]) ** 2).mean(axis=1))
    orms = np.sqrt((((oxs - xg)[:, ey]) ** 2).mean(axis=1))
    assert rms.max() < 2e-3 and np.abs(rms - orms).max() < 1e-7


def test_non_finite_inputs_are_never_reported_solved(binding):
    """A NaN / Inf in a caller's arrays must not come back as PO_STATUS_SOLVED: the residual norms are fmax-accumulated (fmax drops NaN), so the
    iterate is tested for non-finite values explicitly (PO_STATUS_NON_FINITE).  Neighbours in the batch are unaffected."""
    from path_optimizer_amd.abi import PO_STATUS_SOLVED

    b = synth.make_batch(3, B=8)
    clean = binding.Engine(0).solve_batch(b, want_x=True)
    b.ref_k[1, 50] = np.nan
    b.bounds[3, 10, 0, 1] = np.nan
    b.x0[4, 0] = np.inf
    b.ref_s[6, 120] = np.nan
    st, info, xs = binding.Engine(0).solve_batch(b, want_x=True)
    badp = np.array([1, 3, 4, 6])
    good = np.array([0, 2, 5, 7])
    assert (info["status"][badp] == -8).all(), info["status"]  # PO_STATUS_NON_FINITE
    assert (info["status"][good] == PO_STATUS_SOLVED).all()
    assert np.array_equal(xs[good], clean[2][good]) and np.array_equal(info["iters"][good], clean[1]["iters"][good])
    # K (no held controls) and KPC (two waves per path) go through the same test
    for form, cfg in ((2, 3), (1, 5)):
        bb = synth.make_batch(cfg, B=3, formulation=form)
        bb.ref_k[1, 7] = np.nan
        _, inf2, _ = binding.Engine(0).solve_batch(bb)
        assert inf2["status"][1] == -8 and inf2["status"][0] == PO_STATUS_SOLVED and inf2["status"][2] == PO_STATUS_SOLVED


def test_order_hint_changes_scheduling_only(binding):
    """po_batch_in.order (workgroup i solves path order[i]) is a scheduling hint: bit-identical results; a non-permutation is rejected."""
    b = synth.make_batch(3, B=40)
    eng = binding.Engine(0)
    st0, info0, x0 = eng.solve_batch(b, want_x=True)
    order = np.argsort(-info0["iters"].astype(np.int64), kind="stable")
    st1, info1, x1 = eng.solve_batch(b, want_x=True, order=order)
    assert np.array_equal(x0, x1) and np.array_equal(st0, st1) and np.array_equal(info0["iters"], info1["iters"])
    bad = order.copy(); bad[3] = bad[4]
    with pytest.raises(binding.PoError):
        eng.solve_batch(b, order=bad)
    p = binding.default_params(); p.polish = 1
    st2, info2, x2 = binding.Engine(0, p).solve_batch(b, want_x=True, order=order[::-1].copy())
    st3, info3, x3 = binding.Engine(0, p).solve_batch(b, want_x=True)
    assert np.array_equal(x2, x3) and np.array_equal(info2["status_polish"], info3["status_polish"])
