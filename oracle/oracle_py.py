"""ctypes loader for the CPU oracle (oracle/libpo_oracle.so).  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import math

import numpy as np

from path_optimizer_amd.abi import INFO_DTYPE, PoBatchIn, PoBatchOut, PoInfo, PoMap, PoParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libpo_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "po_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libpo_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.po_oracle_wrap_angle.restype = C.c_double
        _LIB.po_oracle_wrap_angle.argtypes = [C.c_double]
    return _LIB


def use_native() -> bool:
    """CPU-baseline timing only: switch this process to a `gcc -O3 -march=native` build of the same source, compiled on the box that
    times it into oracle/_native/ (git-ignored).  Returns False (and keeps the portable build) when that build fails.
    The portable build (no -march, no FMA contraction) stays the checker of the parity tests."""
    global _LIB
    so = os.path.join(_HERE, "_native", "libpo_oracle.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(so)
    except Exception:
        return False
    L.po_oracle_wrap_angle.restype = C.c_double
    L.po_oracle_wrap_angle.argtypes = [C.c_double]
    _LIB = L
    return True


def set_portable_math(on: bool) -> None:
    """Map stages only (bounds, DP search, re-sampling, projections, collision geometry).  False (default): glibc's sin / cos / atan2 / pow and the reference's
    spline elimination order — the mode that is pinned bit for bit against the reference's own binaries.  True: include/po_pmath.h and the device's spline
    derivation, i.e. the exact IEEE operation sequence of the HIP kernels — device and oracle then agree BIT FOR BIT (tests/test_pmath.py)."""
    lib().po_oracle_set_portable_math(1 if on else 0)


class portable_math:
    """with oracle_py.portable_math(): ...   (restores the previous mode)"""

    def __enter__(self):
        self.prev = lib().po_oracle_get_portable_math()
        set_portable_math(True)

    def __exit__(self, *a):
        set_portable_math(bool(self.prev))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_KEEP = []


def _i32(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.int32)
    _KEEP.append(a)
    del _KEEP[:-4]
    return a


def default_params() -> PoParams:
    p = PoParams()
    lib().po_oracle_default_params(C.byref(p))
    return p


def dims(form, N, keep):
    n, m, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib().po_oracle_dims(form, N, keep, C.byref(n), C.byref(m), C.byref(c))
    if rc:
        raise ValueError(f"po_oracle_dims rc={rc}")
    return n.value, m.value, c.value


def keep_steps(form, ref_s):
    ref_s = np.ascontiguousarray(ref_s, dtype=np.float64)
    return lib().po_oracle_keep(form, _p(ref_s), len(ref_s))


def assemble(form, params, N, keep, ref_k, ref_s, ref_z_last, bounds, x0, goal_z, max_k=None, max_kp=None):
    """Returns (P, A, l, u) with P upper-triangular scipy CSC and A scipy CSC, reference ordering."""
    import scipy.sparse as sp

    n, m, _ = dims(form, N, keep)
    L = lib()
    ab, pb = L.po_oracle_nnz_bound_A(form, N, keep), L.po_oracle_nnz_bound_P(form, N, keep)
    Pp = np.zeros(n + 1, np.int32); Pi = np.zeros(pb, np.int32); Px = np.zeros(pb)
    Ap = np.zeros(n + 1, np.int32); Ai = np.zeros(ab, np.int32); Ax = np.zeros(ab)
    l = np.zeros(m); u = np.zeros(m)
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    ref_k, ref_s, bounds, x0, max_k, max_kp = map(f, (ref_k, ref_s, bounds, x0, max_k, max_kp))
    zl = C.c_double(float(ref_z_last))
    rc = L.po_oracle_assemble(form, C.byref(params), N, keep, _p(ref_k), _p(ref_s), C.byref(zl), _p(bounds), _p(x0),
                              C.c_double(float(goal_z)), _p(max_k), _p(max_kp), _p(Pp), _p(Pi), _p(Px), _p(Ap), _p(Ai), _p(Ax), _p(l), _p(u))
    if rc:
        raise ValueError(f"po_oracle_assemble rc={rc}")
    P = sp.csc_matrix((Px[:Pp[n]], Pi[:Pp[n]], Pp), shape=(n, n))
    A = sp.csc_matrix((Ax[:Ap[n]], Ai[:Ap[n]], Ap), shape=(m, n))
    return P, A, l, u


def qp_solve(P, A, l, u, params, q=None, perm=None):
    """OSQP-style ADMM on a scipy (P upper CSC, A CSC). Returns x, y, z, info(dict)."""
    n, m = P.shape[0], A.shape[0]
    P = P.tocsc(); A = A.tocsc()
    P.sort_indices(); A.sort_indices()
    Pp = P.indptr.astype(np.int32); Pi = P.indices.astype(np.int32); Px = P.data.astype(np.float64)
    Ap = A.indptr.astype(np.int32); Ai = A.indices.astype(np.int32); Ax = A.data.astype(np.float64)
    l = np.ascontiguousarray(l, np.float64); u = np.ascontiguousarray(u, np.float64)
    x = np.zeros(n); y = np.zeros(m); z = np.zeros(m)
    qq = None if q is None else np.ascontiguousarray(q, np.float64)
    pm = None if perm is None else np.ascontiguousarray(perm, np.int32)
    info = PoInfo()
    rc = lib().po_oracle_qp_solve(n, m, _p(Pp), _p(Pi), _p(Px), _p(qq), _p(Ap), _p(Ai), _p(Ax), _p(l), _p(u),
                                  C.byref(params), _p(pm), _p(x), _p(y), _p(z), C.byref(info))
    if rc:
        raise RuntimeError(f"po_oracle_qp_solve rc={rc}")
    return x, y, z, {k: getattr(info, k) for k, _ in PoInfo._fields_}


def kkt_check(P, A, l, u, x, y, q=None):
    n, m = P.shape[0], A.shape[0]
    P = P.tocsc(); A = A.tocsc()
    Pp = P.indptr.astype(np.int32); Pi = P.indices.astype(np.int32); Px = P.data.astype(np.float64)
    Ap = A.indptr.astype(np.int32); Ai = A.indices.astype(np.int32); Ax = A.data.astype(np.float64)
    res = np.zeros(4)
    l = np.ascontiguousarray(l, np.float64); u = np.ascontiguousarray(u, np.float64)
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    qq = None if q is None else np.ascontiguousarray(q, np.float64)
    lib().po_oracle_kkt_check(n, m, _p(Pp), _p(Pi), _p(Px), _p(qq), _p(Ap), _p(Ai), _p(Ax), _p(l), _p(u), _p(x), _p(y), _p(res))
    return dict(stationarity=res[0], primal_violation=res[1], complementarity=res[2], objective=res[3])


def class_scaling(form, params, N, keep, ds_nom, passes=10):
    """Class-level Ruiz factors expanded to the reference ordering: D [n], E [m], c."""
    n, m, _ = dims(form, N, keep)
    D = np.ones(n); E = np.ones(m); c = C.c_double(1.0)
    rc = lib().po_oracle_class_scaling(form, C.byref(params), N, keep, C.c_double(float(ds_nom)), passes, _p(D), _p(E), C.byref(c))
    if rc:
        raise ValueError(f"po_oracle_class_scaling rc={rc}")
    return D, E, c.value


def device_equivalent_params(params=None):
    """Oracle parameters that run the SAME algorithm as the device engine with `params`
    (device scaling = k class-level passes  <->  oracle scaling = -k)."""
    p = default_params() if params is None else params
    q = PoParams.from_buffer_copy(bytes(p))
    q.scaling = -abs(p.scaling)
    return q


def _batch_structs(batch, want_x):
    n, m, _ = dims(batch.formulation, batch.N, batch.keep)
    bi = PoBatchIn(batch.formulation, batch.B, batch.N, batch.keep, _p(batch.ref_x), _p(batch.ref_y), _p(batch.ref_z),
                   _p(batch.ref_k), _p(batch.ref_s), _p(batch.bounds), _p(batch.x0), _p(batch.goal_z), _p(batch.max_k), _p(batch.max_kp), _p(_i32(getattr(batch, 'n_points', None))))
    states = np.zeros((batch.B, batch.N, 5)); info = np.zeros(batch.B, dtype=INFO_DTYPE)
    xs = np.zeros((batch.B, n)) if want_x else None
    bo = PoBatchOut(_p(states), _p(info), _p(xs))
    return bi, bo, states, info, xs


def solve_batch(batch, params=None, want_x=True):
    """Sequential CPU solve of a synth.Batch. Returns states [B,N,5], info (structured), x [B,n]."""
    params = params or default_params()
    bi, bo, states, info, xs = _batch_structs(batch, want_x)
    rc = lib().po_oracle_solve_batch(C.byref(params), C.byref(bi), C.byref(bo))
    if rc:
        raise RuntimeError(f"po_oracle_solve_batch rc={rc}")
    return states, info, xs


def output_map(form, N, xsol, ref_x, ref_y, ref_z):
    out = np.zeros((N, 5))
    f = lambda a: np.ascontiguousarray(a, np.float64)
    xsol, ref_x, ref_y, ref_z = map(f, (xsol, ref_x, ref_y, ref_z))
    lib().po_oracle_output(form, N, _p(xsol), _p(ref_x), _p(ref_y), _p(ref_z), _p(out))
    return out


# ---- post-solve step ----
def make_map(dist, resolution, pos_x, pos_y) -> PoMap:
    """po_map over a float32 array dist[size_x, size_y] (index (i, j): i along x, j along y; stored column-major like
    grid_map's Eigen::MatrixXf, i.e. the numpy array is kept Fortran-ordered)."""
    d = np.asfortranarray(dist, dtype=np.float32)
    m = PoMap(d.ctypes.data_as(C.c_void_p), d.shape[0], d.shape[1], float(resolution), float(pos_x), float(pos_y))
    m._keep = d
    return m


def map_distance(m: PoMap, xy):
    L = lib()
    L.po_oracle_map_distance.restype = C.c_double
    L.po_oracle_map_distance.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.po_oracle_map_inside.argtypes = [C.c_void_p, C.c_double, C.c_double]
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    d = np.array([L.po_oracle_map_distance(C.byref(m), x, y) for x, y in xy])
    ins = np.array([L.po_oracle_map_inside(C.byref(m), x, y) for x, y in xy], dtype=np.int32)
    return d, ins


def collision_free(params, m: PoMap, x, y, z) -> int:
    L = lib()
    L.po_oracle_collision_free.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double]
    return L.po_oracle_collision_free(C.byref(params), C.byref(m), x, y, z)


def postcheck_batch(params, m: PoMap, states, info, n_points=None):
    """po_oracle_postcheck per path. states [B,N,5], info structured. Returns n_valid [B], ok [B]."""
    B, N = states.shape[0], states.shape[1]
    nv = np.zeros(B, dtype=np.int32); ok = np.zeros(B, dtype=np.int32)
    L = lib()
    for b in range(B):
        n = N if n_points is None else int(n_points[b])
        s = np.ascontiguousarray(states[b], dtype=np.float64)
        k = C.c_int(0)
        ok[b] = L.po_oracle_postcheck(C.byref(params), C.byref(m), n, _p(s), int(info["status"][b]), C.byref(k))
        nv[b] = k.value
    return nv, ok


# ---- corridor-bounds producer ----
def spline_eval(ks, kv, at):
    L = lib()
    L.po_oracle_spline_eval.restype = C.c_double
    ks = np.ascontiguousarray(ks, np.float64); kv = np.ascontiguousarray(kv, np.float64)
    K = len(ks)
    a = np.zeros(K); b = np.zeros(K); c = np.zeros(K)
    L.po_oracle_spline_fit(K, _p(ks), _p(kv), _p(a), _p(b), _p(c))
    return np.array([L.po_oracle_spline_eval(K, _p(ks), _p(kv), _p(a), _p(b), _p(c), C.c_double(float(t))) for t in np.atleast_1d(at)])


def bounds_path(params, m: PoMap, ref_x, ref_y, ref_z, ref_s, ks, kx, ky):
    """updateBoundsImproved for one path. Returns bounds [N,4,2] (lb, ub; rows >= n_valid are zero), n_valid."""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ref_x, ref_y, ref_z, ref_s, ks, kx, ky = map(f, (ref_x, ref_y, ref_z, ref_s, ks, kx, ky))
    N = len(ref_x)
    out = np.zeros((N, 4, 2))
    n = lib().po_oracle_bounds_path(C.byref(params), C.byref(m), N, _p(ref_x), _p(ref_y), _p(ref_z), _p(ref_s), len(ks), _p(ks), _p(kx), _p(ky), _p(out))
    return out, n


# ---- reference-smoothing QPs ----
def smooth_dims(kind, P):
    n, m = C.c_int(), C.c_int()
    rc = lib().po_oracle_smooth_dims(kind, P, C.byref(n), C.byref(m))
    if rc:
        raise ValueError(f"po_oracle_smooth_dims rc={rc}")
    return n.value, m.value


def _smooth_args(inp, b, n):
    f = lambda key: None if inp.get(key) is None else np.ascontiguousarray(inp[key][b, :n], dtype=np.float64)
    return [f(k) for k in ("x", "y", "angle", "k", "s", "lb", "ub")], float(inp["l0"][b]) if inp.get("l0") is not None else 0.0


def smooth_assemble(kind, params, inp, b=0, m_map=None):
    """(P upper CSC, q, A CSC, l, u) of instance b in the reference's variable / row order."""
    import scipy.sparse as sp

    n_pts = int(inp["n_points"][b]) if inp.get("n_points") is not None else inp["s"].shape[1]
    n, m = smooth_dims(kind, n_pts)
    arrs, l0 = _smooth_args(inp, b, n_pts)
    Pp = np.zeros(n + 1, np.int32); Pi = np.zeros(16 * n_pts + 16, np.int32); Px = np.zeros(16 * n_pts + 16)
    Ap = np.zeros(n + 1, np.int32); Ai = np.zeros(4 * m + 16, np.int32); Ax = np.zeros(4 * m + 16)
    q = np.zeros(n); l = np.zeros(m); u = np.zeros(m)
    L = lib()
    L.po_oracle_smooth_assemble.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_double] + [C.c_void_p] * 9
    rc = L.po_oracle_smooth_assemble(kind, C.byref(params), None if m_map is None else C.byref(m_map), n_pts, *[_p(a) for a in arrs], l0,
                                     _p(Pp), _p(Pi), _p(Px), _p(q), _p(Ap), _p(Ai), _p(Ax), _p(l), _p(u))
    if rc:
        raise ValueError(f"po_oracle_smooth_assemble rc={rc}")
    Pm = sp.csc_matrix((Px[:Pp[n]], Pi[:Pp[n]], Pp), shape=(n, n))
    Am = sp.csc_matrix((Ax[:Ap[n]], Ai[:Ap[n]], Ap), shape=(m, n))
    return Pm, q, Am, l, u


def smooth_batch(kind, params, inp, m_map=None, want_raw=False):
    """po_oracle_smooth_solve over a batch: out_x, out_y, out_s [B,P], info [B] (+ raw [B,n_max])."""
    B, P = inp["s"].shape
    ox = np.zeros((B, P)); oy = np.zeros((B, P)); os_ = np.zeros((B, P))
    info = np.zeros(B, dtype=INFO_DTYPE)
    nmax, _ = smooth_dims(kind, P)
    raw = np.zeros((B, nmax)) if want_raw else None
    L = lib()
    L.po_oracle_smooth_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_double] + [C.c_void_p] * 5
    for b in range(B):
        n_pts = int(inp["n_points"][b]) if inp.get("n_points") is not None else P
        arrs, l0 = _smooth_args(inp, b, n_pts)
        one = PoInfo()
        tx = np.zeros(n_pts); ty = np.zeros(n_pts); ts = np.zeros(n_pts); tr = np.zeros(nmax)
        rc = L.po_oracle_smooth_solve(kind, C.byref(params), None if m_map is None else C.byref(m_map), n_pts, *[_p(a) for a in arrs], l0,
                                      _p(tx), _p(ty), _p(ts), _p(tr), C.byref(one))
        if rc < 0:
            raise ValueError(f"po_oracle_smooth_solve rc={rc}")
        ox[b, :n_pts], oy[b, :n_pts], os_[b, :n_pts] = tx, ty, ts
        for name, _ in INFO_DTYPE:
            info[b][name] = getattr(one, name)
        if want_raw:
            raw[b] = tr
    return ox, oy, os_, info, raw


# ---- reference re-sampling, limits and the DP lattice search ----
def dp_search(params, m: PoMap, ks, kx, ky, length, start, cap=512):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky, start = map(f, (ks, kx, ky, start))
    ls = np.zeros(cap); lb = np.zeros(cap); ub = np.zeros(cap); l0 = C.c_double(0)
    L = lib()
    L.po_oracle_dp_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    n = L.po_oracle_dp_search(C.byref(params), C.byref(m), len(ks), _p(ks), _p(kx), _p(ky), float(length), _p(start), cap, _p(ls), _p(lb), _p(ub), C.byref(l0))
    return n, ls[:max(n, 0)], lb[:max(n, 0)], ub[:max(n, 0)], l0.value


def resample(params, ks, kx, ky, max_s, ds_smaller, ds_larger, cap=4096):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky = map(f, (ks, kx, ky))
    out = [np.zeros(cap) for _ in range(5)]
    L = lib()
    L.po_oracle_resample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 5
    n = L.po_oracle_resample(C.byref(params), len(ks), _p(ks), _p(kx), _p(ky), float(max_s), float(ds_smaller), float(ds_larger), cap, *[_p(o) for o in out])
    return n, [o[:max(n, 0)] for o in out]


def limits(params, v, a):
    v = np.ascontiguousarray(v, np.float64); a = np.ascontiguousarray(a, np.float64)
    mk = np.zeros(len(v)); mkp = np.zeros(len(v))
    L = lib()
    L.po_oracle_limits.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
    L.po_oracle_limits(C.byref(params), len(v), _p(v), _p(a), _p(mk), _p(mkp))
    return mk, mkp


# ---- the remaining glue stages of PathOptimizer::solve and the composed pipeline (restatement of path_optimizer.cpp:40-230) ----
def bspline(px, py, cap=4096):
    px = np.ascontiguousarray(px, np.float64); py = np.ascontiguousarray(py, np.float64)
    x = np.zeros(cap); y = np.zeros(cap); s = np.zeros(cap)
    n = lib().po_oracle_bspline_sample(len(px), _p(px), _p(py), cap, _p(x), _p(y), _p(s))
    return n, x[:max(n, 0)], y[:max(n, 0)], s[:max(n, 0)]


def segment_raw(ks, kx, ky, cap=4096):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky = map(f, (ks, kx, ky))
    out = [np.zeros(cap) for _ in range(5)]
    n = lib().po_oracle_segment_raw(len(ks), _p(ks), _p(kx), _p(ky), cap, *[_p(o) for o in out])
    return n, [o[:max(n, 0)] for o in out]  # x, y, s, angle, k


def post_project(ks, kx, ky, layer_s, offsets):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky, layer_s, offsets = map(f, (ks, kx, ky, layer_s, offsets))
    L = len(layer_s)
    x = np.zeros(L); y = np.zeros(L); s = np.zeros(L)
    rc = lib().po_oracle_post_project(len(ks), _p(ks), _p(kx), _p(ky), L, _p(layer_s), _p(offsets), _p(x), _p(y), _p(s))
    if rc:
        raise ValueError(f"po_oracle_post_project rc={rc}")
    return x, y, s


def segment_init(ks, kx, ky, length, start, goal, exact_position=0):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky, start, goal = map(f, (ks, kx, ky, start, goal))
    out = np.zeros(3)
    L = lib()
    L.po_oracle_segment_init.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    ok = L.po_oracle_segment_init(len(ks), _p(ks), _p(kx), _p(ky), float(length), _p(start), _p(goal), exact_position, _p(out))
    return ok, out[0], out[1], out[2]


def densify(params, m: PoMap, states, status, cap=4096):
    """optimizePath's densifying output branch for one path: (ok, out [n,5])."""
    st = np.ascontiguousarray(states, dtype=np.float64)
    out = np.zeros((cap, 5)); n = C.c_int(0)
    ok = lib().po_oracle_densify(C.byref(params), C.byref(m), st.shape[0], _p(st), int(status), cap, _p(out), C.byref(n))
    return ok, out[:max(n.value, 0)]


def path_optimizer_solve(params, m: PoMap, px, py, start, goal, smooth_params=None):
    """PathOptimizer::solve restated stage by stage on the oracle (start = x, y, heading, k; goal = x, y, heading).
    Returns (ok, path [n,5], trace dict of the intermediate results).  `smooth_params`: OSQP settings of the smoothing QPs (default: same)."""
    from path_optimizer_amd import synth
    from path_optimizer_amd.abi import PO_KP

    sp_ = smooth_params or params
    tr = {}
    n, bx, by, bs = bspline(px, py)
    if n < 0:
        return False, np.zeros((0, 5)), tr
    tr["bspline"] = (bx, by, bs)
    n, (rx, ry, rs, ra, rk) = segment_raw(bs, bx, by)
    if n < 3:
        return False, np.zeros((0, 5)), tr
    inp = dict(x=rx[None], y=ry[None], angle=ra[None], k=rk[None], s=rs[None], lb=None, ub=None, l0=None)
    ox, oy, os_, info, _ = smooth_batch(1 if params.smoothing_method == 1 else 0, sp_, inp, m_map=m)
    tr["tension2"] = (ox[0], oy[0], os_[0], info[0])
    if info["status"][0] != 1:
        return False, np.zeros((0, 5)), tr
    k1s, k1x, k1y = os_[0], ox[0], oy[0]
    len1 = k1s[-1] + 3
    nl, ls, lb, ub, l0 = dp_search(params, m, k1s, k1x, k1y, len1, start[:3])
    tr["dp"] = (nl, ls, lb, ub, l0)
    if nl < 4:  # graphSearchDp false, or postSmooth "Ref is short"
        return False, np.zeros((0, 5)), tr
    pin = dict(x=None, y=None, angle=None, k=None, s=ls[None], lb=lb[None], ub=ub[None], l0=np.array([l0]))
    off, _, _, info2, _ = smooth_batch(2, sp_, pin)
    if info2["status"][0] != 1:
        return False, np.zeros((0, 5)), tr
    k2x, k2y, k2s = post_project(k1s, k1x, k1y, ls, off[0])
    tr["post"] = (k2s, k2x, k2y, off[0])
    ok, e0, e1, len2 = segment_init(k2s, k2x, k2y, k2s[-1], start[:3], goal[:2], exact_position=int(params.enable_exact_position))
    tr["init"] = (ok, e0, e1, len2)
    if not ok:
        return False, np.zeros((0, 5)), tr
    raw_out = bool(params.enable_raw_output)
    nr, (qx, qy, qz, qk, qs) = resample(params, k2s, k2x, k2y, len2, 0.15 if raw_out else 0.5, params.output_spacing if raw_out else 1.0)
    bounds, nv = bounds_path(params, m, qx, qy, qz, qs, k2s, k2x, k2y)
    tr["reference"] = (qx, qy, qz, qk, qs, nv)
    if nv < 2:
        return False, np.zeros((0, 5)), tr
    form = params.optimization_method if params.optimization_method in (1, 2) else PO_KP
    keep = keep_steps(form, qs[:nv])
    lim = (np.full((1, nv), math.tan(params.max_steer) / params.wheel_base), np.full((1, nv), np.finfo(np.float64).max)) if form == 1 else (None, None)
    batch = synth.Batch(form, 1, nv, keep, qx[None, :nv].copy(), qy[None, :nv].copy(), qz[None, :nv].copy(), qk[None, :nv].copy(), qs[None, :nv].copy(),
                        bounds[None, :nv].copy(), np.array([[e0, e1, start[3]]]), np.array([goal[2]]), lim[0], lim[1])
    states, sinfo, _ = solve_batch(batch, params, want_x=False)
    tr["qp"] = sinfo[0]
    if not raw_out:
        okd, dense = densify(params, m, states[0], sinfo["status"][0])
        return bool(okd), dense, tr
    nvalid, okv = postcheck_batch(params, m, states, sinfo)
    return bool(okv[0]), states[0, :nvalid[0]], tr
