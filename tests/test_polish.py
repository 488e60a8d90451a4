"""OSQP's polish step (po_params.polish, opt-in like in OSQP; SURVEY.md App. B / BASELINE.md §3 accuracy clause).

CPU: the oracle's polish (full reduced KKT system, sparse LDL', iterative refinement: oracle/po_oracle.c) against the exact optima of
tests/golden/tight_c3.npz.  GPU: the device polish (condensed form through the block LDL' of the ADMM iteration, csrc/po_fast.inc
polish_kernel) against the oracle's — two different formulations of the same published algorithm — and against the exact optima."""
import os

import numpy as np
import pytest

from path_optimizer_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tight_c3.npz")


def _rms(xs, gold, N):
    return np.sqrt(np.mean((xs[:, 0:3 * N:3] - gold[:len(xs)]) ** 2, axis=1))


def test_oracle_polish_reaches_the_exact_optimum(oracle):
    b = synth.make_batch(3, B=48)
    gold = np.load(GOLD)["e_y"]
    base = oracle.device_equivalent_params()
    _, i0, x0 = oracle.solve_batch(b, base)
    r0 = _rms(x0, gold, b.N)
    res = {}
    for passes in (1,):
        p = oracle.device_equivalent_params()
        p.polish = 1
        _, info, xs = oracle.solve_batch(b, p)
        assert np.array_equal(info["iters"], i0["iters"]) and (info["status"] == 1).all()
        ok = info["status_polish"] == 1
        assert set(np.unique(info["status_polish"])) <= {1, -1}
        r = _rms(xs, gold, b.N)
        # an unsuccessful polish keeps the ADMM solution bit for bit; an adopted one never has larger residuals (polish_successful)
        assert np.array_equal(xs[~ok], x0[~ok])
        assert (info["r_prim"][ok] < i0["r_prim"][ok]).all() and (info["r_dual"][ok] <= i0["r_dual"][ok]).all()
        res[passes] = (ok.mean(), (r <= 1e-4).mean(), np.median(r[ok]))
    assert (r0 <= 1e-4).mean() < 0.6           # ADMM at eps 1e-4 alone: about half of the paths within 1e-4 m of the optimum
    assert res[1][1] >= 0.8
    assert res[1][2] < 1e-9  # where the active set was identified the polished point IS the optimum


def test_oracle_polish_off_is_the_default(oracle):
    p = oracle.default_params()
    assert p.polish == 0 and p.polish_delta == 1e-6 and p.polish_refine_iter == 3  # OSQP defaults
    b = synth.make_batch(2, B=2)
    _, info, _ = oracle.solve_batch(b, oracle.device_equivalent_params())
    assert (info["status_polish"] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("passes", [1])
def test_device_polish_matches_oracle_and_optimum(oracle, passes):
    from path_optimizer_amd import binding

    b = synth.make_batch(3, B=96)
    gold = np.load(GOLD)["e_y"]
    p = binding.default_params()
    assert p.polish == 0 and p.polish_delta == 1e-6 and p.polish_refine_iter == 3
    p.polish = 1
    st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
    st0, info0, xs0 = binding.Engine(0).solve_batch(b, want_x=True)
    po = oracle.device_equivalent_params(p)
    ost, oinfo, oxs = oracle.solve_batch(b, po)
    assert (info["status"] == 1).all() and np.array_equal(info["iters"], info0["iters"])
    ok, ook = info["status_polish"] == 1, oinfo["status_polish"] == 1
    assert set(np.unique(info["status_polish"])) <= {1, -1}
    assert np.array_equal(xs[~ok], xs0[~ok]) and np.array_equal(st[~ok], st0[~ok])  # rejected: the ADMM solution, untouched
    same_admm = info["iters"] == oinfo["iters"]
    both = ok & ook & same_admm
    assert np.array_equal(ok[same_admm], ook[same_admm]), (ok.sum(), ook.sum(), np.where((ok != ook) & same_admm)[0])  # adopt / reject: the same decision on every path whose ADMM run agrees
    assert both.sum() >= 0.6 * b.B
    # condensed block LDL' (device) vs full quasi-definite LDL' (oracle): same polished point
    assert np.abs(xs[both] - oxs[both]).max() < 1e-6, np.abs(xs[both] - oxs[both]).max()
    assert np.abs(st[both] - ost[both]).max() < 1e-6
    r = _rms(xs, gold, b.N)
    r0 = _rms(xs0, gold, b.N)
    assert np.median(r[ok]) < 1e-8 and (r <= 1e-4).mean() >= (0.8 if passes == 1 else 0.88) and (r <= 1e-4).mean() > (r0 <= 1e-4).mean() + 0.25
    assert (info["r_prim"][ok] < info0["r_prim"][ok]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("form,cfg,B", [(1, 5, 12), (2, 3, 24)])
def test_device_polish_other_formulations(oracle, form, cfg, B):
    """KPC (two waves per path) and K: the polished point satisfies the oracle's solver-independent KKT check far below the ADMM tolerance."""
    from path_optimizer_amd import binding

    b = synth.make_batch(cfg, B=B, formulation=form)
    p = binding.default_params()
    p.polish = 1
    st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
    po = oracle.device_equivalent_params(p)
    ost, oinfo, oxs = oracle.solve_batch(b, po)
    ok, ook = info["status_polish"] == 1, oinfo["status_polish"] == 1
    same = info["iters"] == oinfo["iters"]
    assert ok.mean() >= 0.5 and np.array_equal(ok[same], ook[same]), np.where((ok != ook) & same)[0]
    both = ok & ook & same
    assert np.abs(xs[both] - oxs[both]).max() < 1e-6
    assert (info["r_prim"][ok] < 1e-7).all()


@pytest.mark.gpu
def test_device_polish_on_ragged_batches_and_other_keep_values(oracle):
    """Polish after the general (non-uniform) loop variant, on a ragged batch, and for keep_control_steps_ 2 and 3 (other chunk shapes / multi-wave blocks)."""
    import np_twin as T
    from path_optimizer_amd import binding

    for keep, N, ds in ((4, 90, 0.25), (3, 100, 0.3), (2, 70, 0.5)):
        rng = np.random.default_rng(keep)
        insts = [T.random_instance(rng, N, ds=ds) for _ in range(10)]
        stk = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
        b = synth.Batch(0, 10, N, keep, stk("ref_x"), stk("ref_y"), stk("ref_z"), stk("ref_k"), stk("ref_s"), stk("bounds"), stk("x0"), np.array([i["goal_z"] for i in insts]))
        assert binding.keep_control_steps(0, b.ref_s[0]) == keep
        b.n_points = np.array([N, N - 1, N - 5, N // 2, N, 7, N - 2, N, 31, N], dtype=np.int32)
        p = binding.default_params()
        p.polish = 1
        st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
        ost, oinfo, oxs = oracle.solve_batch(b, oracle.device_equivalent_params(p))
        assert np.array_equal(info["status"], oinfo["status"])
        same = (info["iters"] == oinfo["iters"]) & (info["status"] == 1)
        ok, ook = info["status_polish"] == 1, oinfo["status_polish"] == 1
        assert np.array_equal(ok[same], ook[same]), (keep, ok, ook)
        both = ok & ook & same
        assert both.sum() >= 3 and np.abs(xs[both] - oxs[both]).max() < 1e-6 and np.abs(st[both] - ost[both]).max() < 1e-6, keep
