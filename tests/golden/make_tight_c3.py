"""Generator of tests/golden/tight_c3.npz: the EXACT optimum (lateral offsets e_y) of the first 256 paths of BASELINE config 3
(KP, N = 200, per-path random obstacle clearances) — the yardstick of the accuracy clause "<= 1e-4 m lateral-offset RMS"
(SURVEY.md §8d parity bar).

Method (test infrastructure, CPU only): oracle ADMM to eps 1e-6, then primal-dual active-set refinement on the full KKT system with
scipy's sparse LU (an algorithm independent of both ADMM implementations); accepted only when the active set is self-consistent, i.e.
the point satisfies the KKT conditions exactly (checked again by the oracle's solver-independent po_oracle_kkt_check: stationarity,
primal violation, complementarity / dual sign <= 1e-8).  Paths the refinement does not settle fall back to ADMM at eps 1e-10.

    python tests/golden/make_tight_c3.py [n_paths=256] [config=3]      (config 2: tests/golden/tight_c2.npz, N = 120)
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from path_optimizer_amd import synth  # noqa: E402
from path_optimizer_amd.abi import PoParams  # noqa: E402


def kkt_solve(Pf, A, l, u, low, upp, delta=1e-7, refine=4):
    m, n = A.shape
    act = low | upp
    idx = np.where(act)[0]
    Ar = A.tocsr()[idx]
    b = np.where(low, l, u)[act]
    na = len(idx)
    K = sp.bmat([[Pf, Ar.T], [Ar, sp.csr_matrix((na, na))]], format="csc")
    lu = spl.splu((K + sp.diags(np.r_[np.full(n, delta), np.full(na, -delta)])).tocsc())
    rhs = np.r_[np.zeros(n), b]
    sol = lu.solve(rhs)
    for _ in range(refine):
        sol = sol + lu.solve(rhs - K @ sol)
    y = np.zeros(m)
    y[act] = sol[n:]
    return sol[:n], y


def active_set_refine(Pf, A, l, u, x, y, z, passes=12):
    eq = (u - l) < 1e-9
    low = ((z - l) < -y) | eq
    upp = ((u - z) < y) & ~eq
    for _ in range(passes):
        xp, yp = kkt_solve(Pf, A, l, u, low, upp)
        zp = A @ xp
        nlow = ((zp - l) < -yp) | eq
        nupp = ((u - zp) < yp) & ~eq
        if (nlow == low).all() and (nupp == upp).all():
            return xp, yp, True
        low, upp = nlow, nupp
    return None, None, False


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    batch = synth.make_batch(cfg, B=nb)
    p = O.device_equivalent_params()

    def par(eps, mi):
        q = PoParams.from_buffer_copy(bytes(p))
        q.eps_abs = q.eps_rel = eps
        q.max_iter = mi
        return q

    N, keep = batch.N, batch.keep
    ey = np.zeros((nb, N))
    cert = np.zeros((nb, 3))
    method = np.zeros(nb, dtype=np.int32)
    for b in range(nb):
        P, A, l, u = O.assemble(0, p, N, keep, batch.ref_k[b], batch.ref_s[b], batch.ref_z[b, -1], batch.bounds[b], batch.x0[b], batch.goal_z[b])
        Pf = (P + sp.triu(P, 1).T).tocsc()
        xs = ys = None
        for k, eps in enumerate((1e-6, 1e-7, 1e-8)):
            x, y, z, info = O.qp_solve(P, A, l, u, par(eps, 100000))
            xs, ys, ok = active_set_refine(Pf, A, l, u, x, y, z)
            if ok:
                method[b] = k + 1
                break
        if xs is None:
            xs, ys, _, info = O.qp_solve(P, A, l, u, par(1e-10, 4000000))
            method[b] = 9
        c = O.kkt_check(P, A, l, u, xs, ys)
        cert[b] = (c["stationarity"], c["primal_violation"], c["complementarity"])
        ey[b] = xs[0:3 * N:3]
        print(b, method[b], cert[b], flush=True)
    assert cert.max() < 1e-7, cert.max()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tight_c%d.npz" % cfg), e_y=ey, kkt=cert, method=method,
                        note="exact optima of synth.make_batch(%d, B=%d): e_y[b, j] = x[3 j]; kkt = (stationarity, primal violation, complementarity)" % (cfg, nb))
    print("methods", np.bincount(method), "max kkt", cert.max(axis=0))


if __name__ == "__main__":
    main()
