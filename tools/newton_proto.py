"""Dev tool (CPU, numpy / scipy): prototype of a GLOBALISED refinement — semismooth Newton on the augmented Lagrangian with a backtracking line
search — from the point the type-based ADMM stops at, on the class-scaled problem the device iterates on.  Used in round 4 to decide what replaces the
activity-weighted ADMM refinement (whose activity set cycles on ~0.3 % of BASELINE config 3).

    python tools/newton_proto.py <set> <eps_entry> [path ids ... | first:count]
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_tight_full import SETS, batch_of, e_y_of  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from path_optimizer_amd.abi import PoParams  # noqa: E402

INF = 1e20
LS = os.environ.get("LS", "exact")
NBIS = int(os.environ.get("NBIS", "20"))
RTOL = float(os.environ.get("RTOL", "5e-8"))
EPS = float(os.environ.get("EPS", "1e-7"))
LSTOL = float(os.environ.get("LSTOL", "1e-9"))
MAXEV = int(os.environ.get("MAXEV", "40"))


def scaled_problem(batch, b, p):
    form, N, keep = batch.formulation, batch.N, batch.keep
    mk = None if batch.max_k is None else batch.max_k[b]
    mkp = None if batch.max_kp is None else batch.max_kp[b]
    P, A, l, u = O.assemble(form, p, N, keep, batch.ref_k[b], batch.ref_s[b], batch.ref_z[b, -1], batch.bounds[b], batch.x0[b], batch.goal_z[b], mk, mkp)
    ds = float(batch.ref_s[b, 1] - batch.ref_s[b, 0])
    D, E, c = O.class_scaling(form, p, N, keep, ds, 10)
    Pf = (P + sp.triu(P, 1).T).tocsc()
    Ps = (sp.diags(D) @ Pf @ sp.diags(D) * c).tocsc()
    As = (sp.diags(E) @ A @ sp.diags(D)).tocsc()
    return Ps, As, l * E, u * E, D, E, c


def newton_al(Ps, As, l, u, x, y, rho_in, rho_eq, sigma=1e-6, tol=1e-9, max_newton=60, max_outer=20, verbose=False, armijo=1e-4):
    """Semismooth Newton on phi(x) = 1/2 x'Px + sum_i rho_i/2 dist^2(a_i x + y_i/rho_i, [l_i,u_i]) with backtracking, multiplier updates when the inner
    problem is solved.  Returns x, y, stats."""
    n, m = Ps.shape[0], As.shape[0]
    eq = (u - l) < 1e-4
    free = (l < -INF) & (u > INF)
    rho = np.where(eq, rho_eq, rho_in)
    rho[free] = 0.0
    AsT = As.T.tocsc()
    nfac = 0; nls = 0; nouter = 0
    I = sp.identity(n, format="csc")

    def parts(xx, yy):
        w = As @ xx + yy / np.where(rho > 0, rho, 1.0)
        pw = np.clip(w, l, u)
        r = np.where(rho > 0, rho * (w - pw), 0.0)
        return w, pw, r

    def phi(xx, yy):
        w, pw, r = parts(xx, yy)
        return 0.5 * xx @ (Ps @ xx) + 0.5 * np.sum(np.where(rho > 0, rho * (w - pw) ** 2, 0.0))

    hist = []
    for outer in range(max_outer):
        for k in range(max_newton):
            w, pw, r = parts(x, y)
            g = Ps @ x + AsT @ r
            gn = np.abs(g).max()
            if gn < RTOL * (1 + max(np.abs(Ps @ x).max(), np.abs(AsT @ r).max())):
                break
            J = (w < l) | (w > u)
            M = (Ps + sigma * I + AsT @ sp.diags(np.where(J, rho, 0.0)) @ As).tocsc()
            d = spl.splu(M).solve(-g)
            nfac += 1
            if LS == "armijo":
                f0 = phi(x, y); slope = g @ d
                t = 1.0
                while True:
                    nls += 1
                    if phi(x + t * d, y) <= f0 + armijo * t * slope or t < 1e-8:
                        break
                    t *= 0.5
            elif LS == "none":
                t = 1.0
            else:
                # exact line search on the convex piecewise quadratic: root of psi'(t) = d'grad phi(x + t d), by bisection with NBIS evaluations after bracketing
                sA = As @ d; dPd = d @ (Ps @ d); dPx = d @ (Ps @ x)
                w0 = As @ x + y / np.where(rho > 0, rho, 1.0)
                def dpsi(tt):
                    ww = w0 + tt * sA
                    return dPx + tt * dPd + np.sum(np.where(rho > 0, rho * (ww - np.clip(ww, l, u)) * sA, 0.0))
                def dpsi2(tt):  # value and slope of the current linear piece
                    ww = w0 + tt * sA
                    out = (ww < l) | (ww > u)
                    return dPx + tt * dPd + np.sum(np.where(rho > 0, rho * (ww - np.clip(ww, l, u)) * sA, 0.0)), dPd + np.sum(np.where(out & (rho > 0), rho * sA * sA, 0.0))
                # safeguarded Newton on the piecewise-linear psi' (finite: lands on the root once the piece holds it)
                lo, hi = 0.0, np.inf
                t = 1.0
                f0 = abs(dpsi2(0.0)[0])
                for ev in range(MAXEV):
                    nls += 1
                    f, fp = dpsi2(t)
                    if abs(f) <= LSTOL * f0:
                        break
                    if f < 0: lo = t
                    else: hi = t
                    tn = t - f / fp if fp > 0 else (2 * t if hi == np.inf else 0.5 * (lo + hi))
                    if not (lo < tn < hi):
                        tn = 2 * t if hi == np.inf else 0.5 * (lo + hi)
                    t = tn
            x = x + t * d
            if verbose:
                print(f"   outer {outer} newton {k} |g| {gn:.2e} nact {int(J.sum())} t {t:.9f} f0 {(g @ d):.6e}")
        # multiplier update
        w, pw, r = parts(x, y)
        y = r.copy()
        nouter += 1
        ax = As @ x
        rp = np.abs(ax - np.clip(ax, l, u)).max()
        rd = np.abs(Ps @ x + AsT @ y).max()
        hist.append((nfac, rp, rd))
        if verbose:
            print(f" outer {outer}: nfac {nfac} rp {rp:.2e} rd {rd:.2e}")
        if rp < EPS * (1 + max(np.abs(ax).max(), np.abs(np.clip(ax, l, u)).max())) and rd < EPS * (1 + max(np.abs(Ps @ x).max(), np.abs(AsT @ y).max())):
            break
    return x, y, dict(nfac=nfac, nls=nls, nouter=nouter, hist=hist)


def main():
    name = sys.argv[1]; eps_entry = float(sys.argv[2])
    ids = []
    for a in sys.argv[3:]:
        if ":" in a:
            lo, cnt = a.split(":"); ids += list(range(int(lo), int(lo) + int(cnt)))
        else:
            ids.append(int(a))
    rho_in = float(os.environ.get("RHO", "10")); rho_eq = float(os.environ.get("RHO_EQ", str(1e3 * rho_in)))
    verbose = os.environ.get("V", "0") == "1"
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"tight_full_{name}.npz"))["e_y"].astype(np.float64)
    p = O.device_equivalent_params()
    out = []
    for b in ids:
        batch = batch_of(name, 1, b)
        Ps, As, l, u, D, E, c = scaled_problem(batch, 0, p)
        q = PoParams.from_buffer_copy(bytes(p)); q.scaling = 0; q.eps_abs = q.eps_rel = eps_entry; q.max_iter = 20000
        x, y, z, info = O.qp_solve(sp.triu(Ps).tocsc(), As, l, u, q)
        xs, ys, st = newton_al(Ps, As, l, u, x.copy(), y.copy(), rho_in, rho_eq, verbose=verbose)
        ey = e_y_of(batch.formulation, batch.N, xs * D)
        rms = np.sqrt(np.mean((ey - gold[b]) ** 2))
        ey0 = e_y_of(batch.formulation, batch.N, x * D)
        rms0 = np.sqrt(np.mean((ey0 - gold[b]) ** 2))
        out.append((b, info["iters"], st["nfac"], st["nls"], st["nouter"], rms0, rms))
        print(f"path {b}: admm {info['iters']} it -> newton fac {st['nfac']} ls {st['nls']} outer {st['nouter']}  rms before {rms0:.2e} after {rms:.2e}", flush=True)
    o = np.array(out)
    print(f"summary {len(o)} paths: admm mean {o[:,1].mean():.0f} max {o[:,1].max():.0f}; fac mean {o[:,2].mean():.1f} p95 {np.percentile(o[:,2],95):.0f} max {o[:,2].max():.0f}; "
          f"ls mean {o[:,3].mean():.1f}; outer mean {o[:,4].mean():.1f} max {o[:,4].max():.0f}; rms max {o[:,6].max():.2e} n>1e-4 {(o[:,6]>1e-4).sum()}")


if __name__ == "__main__":
    main()
