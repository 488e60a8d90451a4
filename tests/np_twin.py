"""Independent numpy/scipy twin of the oracle (tests only).

A second restatement of SURVEY.md App. A (assembly, via dense scratch like the reference itself:
Eigen::MatrixXd::Zero + sparseView, src/solver/solver_kp_as_input.cpp:47,73) and App. B (OSQP-style
ADMM) written with different machinery (dense numpy writes, scipy.sparse.linalg.splu on the KKT
system) so that a bug in oracle/po_oracle.c and a bug here are unlikely to coincide.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

INF = 1e30
PO_KP, PO_KPC, PO_K = 0, 1, 2


def wrap(a):
    while True:
        if a > math.pi:
            a -= 2 * math.pi
        elif a < -math.pi:
            a += 2 * math.pi
        else:
            return a


def dims(form, N, keep):
    if form == PO_KP:
        C = (N + keep - 2) // keep
        return 5 * N + C, 11 * N + C + 2, C
    if form == PO_KPC:
        C = (N + keep - 2) // keep
        return 6 * N + C, 12 * N + 3 * C + 2, C
    return 4 * N - 1, 11 * N - 1, N - 1


def _end_window(p, goal_z, ref_z_last):
    lo, hi = -INF, INF
    if p.constraint_end_heading:
        psi = wrap(goal_z - ref_z_last)
        if psi < 70 * math.pi / 180:
            lo, hi = psi - 5 * math.pi / 180, psi + 5 * math.pi / 180
    return lo, hi


def assemble_np(form, p, N, keep, ref_k, ref_s, ref_z_last, bounds, x0, goal_z, max_k=None, max_kp=None):
    """Dense-scratch assembly exactly like the reference; returns (P_upper_csc, A_csc, l, u)."""
    n, m, C = dims(form, N, keep)
    d1, d2, d3, d4 = p.d[0], p.d[1], p.d[2], p.d[3]
    H = np.zeros((n, n))
    A = np.zeros((m, n))
    l = np.zeros(m)
    u = np.zeros(m)
    kmax = math.tan(p.max_steer) / p.wheel_base
    lb = bounds[:, :, 0]
    ub = bounds[:, :, 1]
    elo, ehi = _end_window(p, goal_z, ref_z_last)
    if form in (PO_KP, PO_KPC):
        ss, cs = 3 * N, C
        idx = np.arange(N)
        H[3 * idx, 3 * idx] += p.w_dev
        H[3 * idx + 2, 3 * idx + 2] += p.w_curv
        H[ss + cs + idx, ss + cs + idx] += p.w_slack
        cidx = np.arange(C)
        H[ss + cidx, ss + cidx] += keep * p.w_curv_rate
        if form == PO_KP:
            H[ss + cs + N + idx, ss + cs + N + idx] += p.w_slack
        else:
            H[ss + cs + N + idx, ss + cs + N + idx] += p.w_k_slack
            H[ss + cs + 2 * N + cidx, ss + cs + 2 * N + cidx] += p.w_kp_slack * keep
        A[np.arange(ss), np.arange(ss)] = -1
        l[0:3] = u[0:3] = -np.asarray(x0)
        for i in range(N - 1):
            k = ref_k[i]
            ds = ref_s[i + 1] - ref_s[i]
            a = np.zeros((3, 3)); a[0, 1] = 1; a[1, 2] = 1; a[1, 0] = -(k * k)
            A[3 * (i + 1):3 * (i + 1) + 3, 3 * i:3 * i + 3] = a * ds + np.eye(3)
            A[3 * (i + 1):3 * (i + 1) + 3, ss + i // keep] = np.array([0, 0, 1.0]) * ds
            kp = (ref_k[i + 1] - k) / ds
            c = ds * (np.array([0, 0, kp]) - a @ np.array([0, 0, k]) - np.array([0, 0, 1.0]) * kp)
            l[3 * (i + 1):3 * (i + 1) + 3] = -c
            u[3 * (i + 1):3 * (i + 1) + 3] = -c
        if form == PO_KP:
            vb = 3 * N; cb = vb + 2 * N + C; eb = cb + 6 * N
            A[vb + idx, 3 * idx + 2] = 1
            A[vb + N + C + idx, ss + cs + idx] = 1
            A[vb + N + cidx, ss + cidx] = 1
            l[vb + idx] = -kmax; u[vb + idx] = kmax
            l[vb + N + C + idx] = 0; u[vb + N + C + idx] = p.margin
            l[vb + N + cidx] = -INF; u[vb + N + cidx] = INF
            for r, dd in ((0, d1), (1, d3)):
                A[cb + 2 * idx + r, 3 * idx] = 1
                A[cb + 2 * idx + r, 3 * idx + 1] = dd
            for blk, dd, sg in ((2, d4, -1), (3, d4, 1), (4, d2, -1), (5, d2, 1)):
                A[cb + blk * N + idx, 3 * idx] = 1
                A[cb + blk * N + idx, 3 * idx + 1] = dd
                A[cb + blk * N + idx, ss + cs + idx] = sg
            l[cb + 2 * idx] = lb[:, 0]; u[cb + 2 * idx] = ub[:, 0]
            l[cb + 2 * idx + 1] = lb[:, 2]; u[cb + 2 * idx + 1] = ub[:, 2]
            u[cb + 2 * N + idx] = ub[:, 3] - p.margin; l[cb + 2 * N + idx] = -INF
            l[cb + 3 * N + idx] = lb[:, 3] + p.margin; u[cb + 3 * N + idx] = INF
            u[cb + 4 * N + idx] = ub[:, 1] - p.margin; l[cb + 4 * N + idx] = -INF
            l[cb + 5 * N + idx] = lb[:, 1] + p.margin; u[cb + 5 * N + idx] = INF
            A[eb, ss - 3] = 1; A[eb + 1, ss - 2] = 1
            l[eb], u[eb] = -1, 1
            l[eb + 1], u[eb + 1] = elo, ehi
        else:
            kl = 3 * N; ku = kl + N; kpl = ku + N; kpu = kpl + C; sb = kpu + C
            cb = sb + 2 * N + C; eb = cb + 5 * N; s0 = ss + cs
            A[kl + idx, 3 * idx + 2] = 1; A[kl + idx, s0 + N + idx] = 1
            A[ku + idx, 3 * idx + 2] = 1; A[ku + idx, s0 + N + idx] = -1
            A[sb + idx, s0 + idx] = 1; A[sb + N + idx, s0 + N + idx] = 1
            A[kpl + cidx, ss + cidx] = 1; A[kpl + cidx, s0 + 2 * N + cidx] = 1
            A[kpu + cidx, ss + cidx] = 1; A[kpu + cidx, s0 + 2 * N + cidx] = -1
            A[sb + 2 * N + cidx, s0 + 2 * N + cidx] = 1
            for r, dd in ((0, d1), (1, d2), (2, d4)):
                A[cb + 3 * idx + r, 3 * idx] = 1
                A[cb + 3 * idx + r, 3 * idx + 1] = dd
            for blk, sg in ((3, -1), (4, 1)):
                A[cb + blk * N + idx, 3 * idx] = 1
                A[cb + blk * N + idx, 3 * idx + 1] = d3
                A[cb + blk * N + idx, s0 + idx] = sg
            A[eb, ss - 3] = 1; A[eb + 1, ss - 2] = 1
            mk = np.asarray(max_k); mkp = np.asarray(max_kp)
            l[kl + idx] = -mk; u[kl + idx] = INF
            l[ku + idx] = -INF; u[ku + idx] = mk
            l[sb + idx] = 0; u[sb + idx] = p.margin
            l[sb + N + idx] = 0; u[sb + N + idx] = np.maximum(kmax - mk, 0.0)
            l[kpl + cidx] = -mkp[:C]; u[kpl + cidx] = INF
            l[kpu + cidx] = -INF; u[kpu + cidx] = mkp[:C]
            l[sb + 2 * N + cidx] = 0; u[sb + 2 * N + cidx] = INF
            l[cb + 3 * idx] = lb[:, 0]; u[cb + 3 * idx] = ub[:, 0]
            l[cb + 3 * idx + 1] = lb[:, 1]; u[cb + 3 * idx + 1] = ub[:, 1]
            l[cb + 3 * idx + 2] = lb[:, 3]; u[cb + 3 * idx + 2] = ub[:, 3]
            u[cb + 3 * N + idx] = ub[:, 2] - p.margin; l[cb + 3 * N + idx] = -INF
            l[cb + 4 * N + idx] = lb[:, 2] + p.margin; u[cb + 4 * N + idx] = INF
            l[eb], u[eb] = -INF, INF
            l[eb + 1], u[eb + 1] = elo, ehi
    else:  # K
        L = p.wheel_base
        w_c, w_cr, w_pq, w_e = p.k_w_curv, p.k_w_curv_rate, p.k_w_dev, p.w_slack
        cs = N - 1
        for i in range(N):
            H[2 * i + 1, 2 * i + 1] = w_pq
        for i in range(cs):
            H[2 * N + i, 2 * N + i] = (w_c + w_cr) if i in (0, cs - 1) else (2 * w_cr + w_c)
            if i + 1 < cs:
                H[2 * N + i, 2 * N + i + 1] = H[2 * N + i + 1, 2 * N + i] = -w_cr
        H[3 * N - 1:, 3 * N - 1:] = np.eye(N) * w_e
        A[np.arange(2 * N), np.arange(2 * N)] = -1
        for i in range(N - 1):
            k = ref_k[i]; rs = ref_s[i + 1] - ref_s[i]; dl = math.atan(k * L)
            A[2 * (i + 1):2 * (i + 1) + 2, 2 * i:2 * i + 2] = np.array([[1, -rs * (k * k)], [rs, 1]])
            cd = math.cos(dl)
            A[2 * (i + 1):2 * (i + 1) + 2, 2 * N + i] = [rs / L / (cd * cd), 0]  # gcc folds pow(x,2) to x*x
            l[2 + 2 * i] = u[2 + 2 * i] = rs * dl / L / (cd * cd)
        A[2 * N + np.arange(4 * N - 1), np.arange(4 * N - 1)] = 1
        idx = np.arange(N)
        for r, dd in ((0, d1), (1, d3), (2, d4)):
            A[6 * N - 1 + 3 * idx + r, 2 * idx] = dd
            A[6 * N - 1 + 3 * idx + r, 2 * idx + 1] = 1
        for base, sg in ((9 * N - 1, -1), (10 * N - 1, 1)):
            A[base + idx, 2 * idx] = d2
            A[base + idx, 2 * idx + 1] = 1
            A[base + idx, 3 * N - 1 + idx] = sg
        l[0:2] = u[0:2] = [-x0[1], -x0[0]]
        l[2 * N:4 * N] = -INF; u[2 * N:4 * N] = INF
        if elo > -INF:
            l[4 * N - 2], u[4 * N - 2] = elo, ehi
        l[4 * N:5 * N - 1] = -p.max_steer; u[4 * N:5 * N - 1] = p.max_steer
        l[5 * N - 1:6 * N - 1] = 0; u[5 * N - 1:6 * N - 1] = p.margin
        for r, c in ((0, 0), (1, 2), (2, 3)):
            l[6 * N - 1 + 3 * idx + r] = lb[:, c]; u[6 * N - 1 + 3 * idx + r] = ub[:, c]
        l[9 * N - 1 + idx] = -INF; u[9 * N - 1 + idx] = ub[:, 1] - p.margin
        l[10 * N - 1 + idx] = lb[:, 1] + p.margin; u[10 * N - 1 + idx] = INF
    P = sp.triu(sp.csc_matrix(H)).tocsc()
    P.eliminate_zeros()
    Ac = sp.csc_matrix(A)
    Ac.eliminate_zeros()
    return P, Ac, l, u


def admm_np(P, A, l, u, p, max_iter=None):
    """Plain OSQP-style ADMM (no scaling) with scipy's sparse LU on the KKT matrix."""
    n, m = P.shape[0], A.shape[0]
    Pf = (P + sp.triu(P, 1).T).tocsc()
    rho = p.rho0
    loose = (l < -INF * 1e-4) & (u > INF * 1e-4)
    eq = (~loose) & (u - l < 1e-4)

    def rho_vec(r):
        v = np.full(m, r)
        v[eq] = 1e3 * r
        v[loose] = 1e-6
        return v

    def factor(rv):
        K = sp.bmat([[Pf + p.sigma * sp.eye(n), A.T], [A, -sp.diags(1.0 / rv)]]).tocsc()
        return spla.splu(K)

    rv = rho_vec(rho)
    lu = factor(rv)
    x = np.zeros(n); z = np.zeros(m); y = np.zeros(m)
    iters = 0
    nref = 0
    status = -10
    rp = rd = 0.0
    for it in range(1, (max_iter or p.max_iter) + 1):
        iters = it
        sol = lu.solve(np.concatenate([p.sigma * x, z - y / rv]))
        xt = sol[:n]; zt = z + (sol[n:] - y) / rv
        xn = p.alpha * xt + (1 - p.alpha) * x
        v = p.alpha * zt + (1 - p.alpha) * z + y / rv
        zn = np.clip(v, l, u)
        y = y + rv * (p.alpha * zt + (1 - p.alpha) * z - zn)
        x, z = xn, zn
        check = it % p.check_every == 0
        adapt = p.adapt_every > 0 and it % p.adapt_every == 0
        if check or adapt:
            Ax = A @ x; Px = Pf @ x; Aty = A.T @ y
            rp = np.abs(Ax - z).max(); rd = np.abs(Px + Aty).max()
            pn = max(np.abs(z).max(), np.abs(Ax).max()); dn = max(np.abs(Aty).max(), np.abs(Px).max())
            if check and rp < p.eps_abs + p.eps_rel * pn and rd < p.eps_abs + p.eps_rel * dn:
                status = 1
                break
            if adapt:
                rn = rho * math.sqrt((rp / (pn + 1e-10)) / (rd / (dn + 1e-10) + 1e-10))
                rn = min(max(rn, 1e-6), 1e6)
                if rn > rho * p.adapt_tol or rn < rho / p.adapt_tol:
                    rho = rn
                    rv = rho_vec(rho)
                    lu = factor(rv)
                    nref += 1
    if status != 1:
        status = -2
    return x, y, z, dict(status=status, iters=iters, n_refactor=nref, r_prim=rp, r_dual=rd, rho=rho)


def random_instance(rng, N, ds=0.25, form=PO_KP, narrow=False):
    """Small random planning instance (not a BASELINE config) for oracle/twin cross-checks."""
    s = ds * np.arange(N)
    k = 0.05 * np.sin(2 * math.pi * s / rng.uniform(8, 30) + rng.uniform(0, 6.28)) + rng.uniform(-0.01, 0.01)
    z = rng.uniform(-3, 3) + np.concatenate(([0.0], np.cumsum(k[:-1] * ds)))
    x = np.concatenate(([0.0], np.cumsum(np.cos(z[:-1]) * ds)))
    y = np.concatenate(([0.0], np.cumsum(np.sin(z[:-1]) * ds)))
    w = rng.uniform(0.8, 1.2, size=(N, 4)) if narrow else rng.uniform(1.5, 2.5, size=(N, 4))
    w2 = rng.uniform(0.8, 1.2, size=(N, 4)) if narrow else rng.uniform(1.5, 2.5, size=(N, 4))
    bounds = np.stack([-w, w2], axis=-1)
    x0 = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.05, 0.05), k[0]])
    goal = z[-1] + rng.uniform(-0.05, 0.05)
    v = rng.uniform(3, 15, size=N)
    max_k = 0.4 * 9.8 / v ** 2
    max_kp = 0.1 / v
    return dict(ref_x=x, ref_y=y, ref_z=z, ref_k=k, ref_s=s, bounds=bounds, x0=x0, goal_z=goal, max_k=max_k, max_kp=max_kp)


def kkt_certificate(P, A, l, u, x, act_tol=1e-6):
    """Solver-independent optimality certificate of a primal point x of  min 1/2 x'Px  s.t.  l <= Ax <= u  (P upper-triangular CSC as the
    assemblers return it).  No multipliers are taken from any solver: the rows active at x are read off Ax, and scipy's bounded least squares finds the
    multipliers of the right sign (y <= 0 at a lower bound, y >= 0 at an upper bound, free on equality rows) that minimise ||Px + A'y||.  x is optimal
    iff the primal violation and that minimum are zero; returns both (inf-norms), the second also relative to ||Px||_inf."""
    import scipy.sparse as sp
    from scipy.optimize import lsq_linear

    P = sp.csc_matrix(P)
    Pf = P + P.T - sp.diags(P.diagonal())
    A = sp.csr_matrix(A)
    x = np.asarray(x, dtype=np.float64)
    ax = A @ x
    fin_l, fin_u = l > -1e20, u < 1e20
    viol = float(max(0.0, np.max(np.where(fin_l, l - ax, 0.0)), np.max(np.where(fin_u, ax - u, 0.0))))
    eq = fin_l & fin_u & (u - l < 1e-9)
    at_l = fin_l & (ax - l <= act_tol * (1.0 + np.abs(np.where(fin_l, l, 0.0))))
    at_u = fin_u & (u - ax <= act_tol * (1.0 + np.abs(np.where(fin_u, u, 0.0))))
    rows = np.flatnonzero(eq | at_l | at_u)
    g = Pf @ x
    if rows.size == 0:
        r = float(np.abs(g).max())
        return dict(primal_violation=viol, stationarity=r, stationarity_rel=r / max(1.0, float(np.abs(g).max())), n_active=0)
    free = eq[rows] | (at_l[rows] & at_u[rows])
    lb = np.where(free | at_l[rows], -np.inf, 0.0)
    ub = np.where(free | at_u[rows], np.inf, 0.0)
    At = sp.csc_matrix(A[rows].T)
    # column scaling (rows of A differ by orders of magnitude between the formulations' slack / curvature rows): plain diagonal preconditioning
    cn = np.sqrt(np.asarray(At.multiply(At).sum(axis=0)).ravel())
    cn[cn == 0] = 1.0
    res = lsq_linear(At @ sp.diags(1.0 / cn), -g, bounds=(lb, ub), method="trf", tol=1e-13, lsmr_tol="auto", max_iter=400)
    r = float(np.abs(At @ (res.x / cn) + g).max())
    if r > 1e-9 * max(1.0, float(np.abs(g).max())) and At.shape[0] * At.shape[1] <= 4_000_000:
        # the iterative (LSMR) inner solver may stop short on ill-conditioned row sets (one control held over a whole short path: found by tools/fuzz_wide.py, 1.8e-3 where the exact
        # solve gives 4e-14): settle it with the dense exact solver
        Ad = (At @ sp.diags(1.0 / cn)).toarray()
        res = lsq_linear(Ad, -g, bounds=(lb, ub), method="bvls", tol=1e-15, max_iter=5000)
        r = min(r, float(np.abs(At @ (res.x / cn) + g).max()))
    return dict(primal_violation=viol, stationarity=r, stationarity_rel=r / max(1.0, float(np.abs(g).max())), n_active=int(rows.size))
