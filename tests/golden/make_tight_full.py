"""Generator of tests/golden/tight_full_<set>.npz: the EXACT optimum (lateral offsets e_y) of EVERY path of the batches the metric is quoted on — the
yardstick of the accuracy clause "<= 1e-4 m lateral-offset RMS per path" (SURVEY.md §8d parity bar) on the whole batch, not a sample.

    set     batch                                                                  paths
    c3      BASELINE config 3: KP, N = 200, per-path random obstacle clearances    4096  (the batch `value` is measured on)
    c2      BASELINE config 2: KP, N = 120, fixed corridor                         1024
    c5      BASELINE config 5: KPC, N = 400, curvature / curvature-rate limits     4096  (round 4: the whole batch, was 256)
    k       config-3 corridors, K formulation, N = 200                             4096  (round 4: the whole batch, was 256)
    keep3   config-3 generator at N = 231, ds = 0.3 m -> keep_control_steps_ = 3   1024  (the shape the reference's own pipeline hands the QP)

Method (test infrastructure, CPU only; the same as make_tight_c3.py): oracle ADMM to eps 1e-6, then primal-dual active-set iteration on the full KKT system
with scipy's sparse LU (an algorithm independent of both ADMM implementations), accepted only when the active set reproduces itself, i.e. the point satisfies
the KKT conditions; every point is then certified by the oracle's solver-independent po_oracle_kkt_check (stationarity, primal violation, complementarity:
<= 2e-6 asserted; measured <= 3e-14 on KP / K, 4e-7 absolute on KPC, whose curvature-rate slack rows carry the weight 1e5).  Paths the active-set iteration does not settle fall back to ADMM at eps 1e-10.
e_y is stored as float32 (|rounding| <= 1.2e-7 m for |e_y| < 2 m: three orders below the 1e-4 m bar) so that 4096 x 200 values stay a 3 MB fixture.

    python tests/golden/make_tight_full.py [set ...]        (default: all sets; one worker process per host core)
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

SETS = {
    # name: (config, kwargs of synth.make_batch, paths)
    "c3": (3, {}, 4096),
    "c2": (2, {}, 1024),
    "c5": (5, {}, 4096),
    "k": (3, {"formulation": 2}, 4096),
    "keep3": (3, {"N": 231, "ds": 0.3}, 1024),
}


def batch_of(name, B=None, first=0):
    from path_optimizer_amd import synth

    cfg, kw, nb = SETS[name]
    return synth.make_batch(cfg, B=nb if B is None else B, first_path=first, **kw)


def e_y_of(form, N, x):
    return x[1:2 * N:2] if form == 2 else x[0:3 * N:3]  # K orders the state pair (e_phi, e_y), solver_k_as_input.cpp:156


def _work(arg):
    name, lo, hi = arg
    import scipy.sparse as sp

    from make_tight_c3 import active_set_refine
    from oracle import oracle_py as O
    from path_optimizer_amd.abi import PoParams

    batch = batch_of(name, hi - lo, lo)
    p = O.device_equivalent_params()

    def par(eps, mi):
        q = PoParams.from_buffer_copy(bytes(p))
        q.eps_abs = q.eps_rel = eps
        q.max_iter = mi
        return q

    N, keep, form = batch.N, batch.keep, batch.formulation
    ey = np.zeros((hi - lo, N)); cert = np.zeros((hi - lo, 3)); method = np.zeros(hi - lo, dtype=np.int8)
    for b in range(hi - lo):
        mk = None if batch.max_k is None else batch.max_k[b]
        mkp = None if batch.max_kp is None else batch.max_kp[b]
        P, A, l, u = O.assemble(form, p, N, keep, batch.ref_k[b], batch.ref_s[b], batch.ref_z[b, -1], batch.bounds[b], batch.x0[b], batch.goal_z[b], mk, mkp)
        Pf = (P + sp.triu(P, 1).T).tocsc()
        xs = ys = None
        for k, eps in enumerate((1e-6, 1e-7, 1e-8)):
            x, y, z, info = O.qp_solve(P, A, l, u, par(eps, 200000))
            xs, ys, ok = active_set_refine(Pf, A, l, u, x, y, z)
            if ok:
                method[b] = k + 1
                break
        if xs is None:
            xs, ys, _, info = O.qp_solve(P, A, l, u, par(1e-10, 4000000))
            method[b] = 9
        c = O.kkt_check(P, A, l, u, xs, ys)
        cert[b] = (c["stationarity"], c["primal_violation"], c["complementarity"])
        ey[b] = e_y_of(form, N, xs)
    return lo, ey, cert, method


def main():
    names = sys.argv[1:] or list(SETS)
    nproc = int(os.environ.get("PO_GOLDEN_PROCS", os.cpu_count() or 1))
    for name in names:
        nb = SETS[name][2]
        step = max(8, -(-nb // (4 * nproc)))
        jobs = [(name, lo, min(nb, lo + step)) for lo in range(0, nb, step)]
        with mp.get_context("spawn").Pool(min(nproc, len(jobs))) as pool:
            parts = sorted(pool.map(_work, jobs, chunksize=1), key=lambda t: t[0])
        ey = np.concatenate([t[1] for t in parts]); cert = np.concatenate([t[2] for t in parts]); method = np.concatenate([t[3] for t in parts])
        assert len(ey) == nb and cert.max() < 2e-6, (len(ey), cert.max())  # (KPC: rows weighted 1e5 -> stationarity ~4e-7 absolute)
        b0 = batch_of(name, 1)
        np.savez_compressed(os.path.join(HERE, f"tight_full_{name}.npz"), e_y=ey.astype(np.float32), kkt_max=cert.max(axis=0), method=method,
                            note=f"exact optima of set '{name}' ({['KP', 'KPC', 'K'][b0.formulation]}, N = {b0.N}, keep = {b0.keep}, {nb} paths, path ids 0..{nb - 1}): "
                                 "e_y[b, j] as float32; kkt_max = max over paths of (stationarity, primal violation, complementarity); method 1..3 = active set from ADMM at "
                                 "1e-6 / 1e-7 / 1e-8, 9 = ADMM at 1e-10")
        print(name, "paths", nb, "methods", np.bincount(method), "max kkt", cert.max(axis=0), flush=True)


if __name__ == "__main__":
    main()
