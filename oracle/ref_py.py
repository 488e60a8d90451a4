"""ctypes loader for oracle/_ref/libpo_ref.so — the REFERENCE's own solver sources (src/solver/*.cpp,
src/config/planning_flags.cpp, src/data_struct/vehicle_state_frenet.cpp) compiled where they lie under
/root/reference against the stand-in headers in oracle/ref_shim/.  TEST INFRASTRUCTURE ONLY.

Available only where /root/reference exists (this container); tests that need it skip otherwise and fall back
to the committed fixtures under tests/golden/ that were generated from it (tests/golden/make_golden.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from path_optimizer_amd.abi import PoParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libpo_ref.so")
_LIB = None
_LIBB = None
NAMES = {0: "KP", 1: "KPC", 2: "K"}


def available() -> bool:
    return os.path.exists(_SO) or os.path.isdir("/root/reference")


def lib():
    global _LIB
    if _LIB is None:
        if os.path.isdir("/root/reference"):
            subprocess.check_call(["make", "-C", _HERE, "libpo_oracle.so", "ref"], stdout=subprocess.DEVNULL)
        C.CDLL(os.path.join(_HERE, "libpo_oracle.so"), mode=C.RTLD_GLOBAL)
        _LIB = C.CDLL(_SO)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def flags():
    out = np.zeros(16)
    lib().po_ref_flags(_p(out))
    keys = ["d1", "d2", "d3", "d4", "KP_curvature_weight", "KP_curvature_rate_weight", "KP_deviation_weight", "KP_slack_weight",
            "K_curvature_weight", "K_curvature_rate_weight", "K_deviation_weight", "expected_safety_margin", "max_steering_angle",
            "wheel_base", "constraint_end_heading", "mu"]
    return dict(zip(keys, out))


def solve(type_name: str, inst: dict, params: PoParams):
    """Runs the reference's OsqpSolver::create(type)->solve(). Returns dict(rc, states, P, A, q, l, u, x)."""
    import scipy.sparse as sp

    f = lambda k: None if inst.get(k) is None else np.ascontiguousarray(inst[k], dtype=np.float64)
    N = len(inst["ref_s"])
    states = np.zeros((N, 5))
    ns = C.c_int(0)
    rc = lib().po_ref_solve(type_name.encode(), N, _p(f("ref_x")), _p(f("ref_y")), _p(f("ref_z")), _p(f("ref_k")), _p(f("ref_s")),
                            _p(f("bounds")), _p(f("x0")), C.c_double(float(inst["goal_z"])), _p(f("max_k")), _p(f("max_kp")),
                            C.byref(params), _p(states), C.byref(ns))
    if rc < 0:
        return dict(rc=rc)
    n, m, pnz, anz = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib().po_ref_get_dims(C.byref(n), C.byref(m), C.byref(pnz), C.byref(anz))
    n, m, pnz, anz = n.value, m.value, pnz.value, anz.value
    Pp = np.zeros(n + 1, np.int32); Pi = np.zeros(pnz, np.int32); Px = np.zeros(pnz)
    Ap = np.zeros(n + 1, np.int32); Ai = np.zeros(anz, np.int32); Ax = np.zeros(anz)
    q = np.zeros(n); l = np.zeros(m); u = np.zeros(m); x = np.zeros(n)
    lib().po_ref_get_qp(_p(Pp), _p(Pi), _p(Px), _p(Ap), _p(Ai), _p(Ax), _p(q), _p(l), _p(u), _p(x))
    return dict(rc=rc, states=states[:ns.value], P=sp.csc_matrix((Px, Pi, Pp), shape=(n, n)), A=sp.csc_matrix((Ax, Ai, Ap), shape=(m, n)),
                q=q, l=l, u=u, x=x, n=n, m=m)


# ---- post-solve step: the reference's own CollisionChecker / CarGeometry / Map / tools (ref_shim/ref_glue_post.cpp) ----
def collision_free(m, x, y, z) -> int:
    L = lib()
    L.po_ref_collision_free.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    return L.po_ref_collision_free(C.byref(m), x, y, z)


def map_distance(m, x, y) -> float:
    L = lib()
    L.po_ref_map_distance.restype = C.c_double
    L.po_ref_map_distance.argtypes = [C.c_void_p, C.c_double, C.c_double]
    return L.po_ref_map_distance(C.byref(m), x, y)


def postcheck(m, states):
    """optimizePath's raw-output tail on one solved path: returns ok, n_valid, s [n]."""
    s = np.ascontiguousarray(states, dtype=np.float64)
    n = s.shape[0]
    nv = C.c_int(0)
    so = np.zeros(n)
    ok = lib().po_ref_postcheck(C.byref(m), n, _p(s), C.byref(nv), _p(so))
    return ok, nv.value, so


# ---- corridor-bounds producer: the reference's real ReferencePathImpl (ref_shim/ref_glue_bounds.cpp, own library) ----
def lib_bounds():
    global _LIBB
    if _LIBB is None:
        lib()
        _LIBB = C.CDLL(os.path.join(_HERE, "_ref", "libpo_ref_bounds.so"))
    return _LIBB


def spline_eval(ks, kv, at):
    ks = np.ascontiguousarray(ks, np.float64); kv = np.ascontiguousarray(kv, np.float64); at = np.ascontiguousarray(np.atleast_1d(at), np.float64)
    out = np.zeros(len(at))
    lib_bounds().po_ref_spline_eval(len(ks), _p(ks), _p(kv), len(at), _p(at), _p(out))
    return out


def bounds_path(m, ref_x, ref_y, ref_z, ref_s, ks, kx, ky):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ref_x, ref_y, ref_z, ref_s, ks, kx, ky = map(f, (ref_x, ref_y, ref_z, ref_s, ks, kx, ky))
    N = len(ref_x)
    out = np.zeros((N, 4, 2))
    n = lib_bounds().po_ref_bounds_path(C.byref(m), N, _p(ref_x), _p(ref_y), _p(ref_z), _p(ref_s), len(ks), _p(ks), _p(kx), _p(ky), _p(out))
    return out, n


# ---- reference-smoothing QPs: the reference's real smoother classes (ref_shim/ref_glue_smooth.cpp, own library) ----
_LIBS = None


def lib_smooth():
    global _LIBS
    if _LIBS is None:
        lib()
        _LIBS = C.CDLL(os.path.join(_HERE, "_ref", "libpo_ref_smooth.so"))
    return _LIBS


def smooth_flags():
    out = np.zeros(6)
    lib_smooth().po_ref_smooth_flags(_p(out))
    return out


def _captured_smooth():
    import scipy.sparse as sp

    L = lib_smooth()
    n, m, pnz, anz = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    L.po_ref_smooth_get_dims(C.byref(n), C.byref(m), C.byref(pnz), C.byref(anz))
    n, m, pnz, anz = n.value, m.value, pnz.value, anz.value
    Pp = np.zeros(n + 1, np.int32); Pi = np.zeros(pnz, np.int32); Px = np.zeros(pnz)
    Ap = np.zeros(n + 1, np.int32); Ai = np.zeros(anz, np.int32); Ax = np.zeros(anz)
    q = np.zeros(n); l = np.zeros(m); u = np.zeros(m); x = np.zeros(n)
    L.po_ref_smooth_get_qp(_p(Pp), _p(Pi), _p(Px), _p(Ap), _p(Ai), _p(Ax), _p(q), _p(l), _p(u), _p(x))
    return dict(P=sp.csc_matrix((Px, Pi, Pp), shape=(n, n)), A=sp.csc_matrix((Ax, Ai, Ap), shape=(m, n)), q=q, l=l, u=u, x=x, n=n, m=m)


def osqp_smooth(kind, params, x, y, angle, k, s, m_map=None):
    """TensionSmoother2::osqpSmooth (kind 0) / TensionSmoother::osqpSmooth (kind 1) of the reference on one instance."""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    x, y, angle, k, s = map(f, (x, y, angle, k, s))
    P = len(x)
    rx = np.zeros(P); ry = np.zeros(P); rs = np.zeros(P)
    L = lib_smooth()
    L.po_ref_osqp_smooth.argtypes = [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 9
    rc = L.po_ref_osqp_smooth(kind, None if m_map is None else C.byref(m_map), P, _p(x), _p(y), _p(angle), _p(k), _p(s), C.byref(params), _p(rx), _p(ry), _p(rs))
    out = _captured_smooth()
    out.update(rc=rc, out_x=rx, out_y=ry, out_s=rs)
    return out


def post_smooth(params, layer_s, lb, ub, l0, ks, kx, ky):
    """ReferencePathSmoother::postSmooth of the reference: captured QP + samples of the re-fitted spline."""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    layer_s, lb, ub, ks, kx, ky = map(f, (layer_s, lb, ub, ks, kx, ky))
    Lr = len(layer_s)
    send = C.c_double(0)
    nx = np.zeros(Lr); ny = np.zeros(Lr)
    L = lib_smooth()
    L.po_ref_post_smooth.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int] + [C.c_void_p] * 7
    rc = L.po_ref_post_smooth(Lr, _p(layer_s), _p(lb), _p(ub), float(l0), len(ks), _p(ks), _p(kx), _p(ky), C.byref(params), C.byref(send), _p(nx), _p(ny))
    out = _captured_smooth()
    out.update(rc=rc, s_end=send.value, new_x=nx, new_y=ny)
    return out


# ---- f-4: the reference's own graphSearchDp / buildReferenceFromSpline / updateLimits (same library) ----
def dp_search(m_map, ks, kx, ky, length, start, cap=512):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky, start = map(f, (ks, kx, ky, start))
    ls = np.zeros(cap); lb = np.zeros(cap); ub = np.zeros(cap); l0 = C.c_double(0)
    L = lib_smooth()
    L.po_ref_dp_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 5
    n = L.po_ref_dp_search(C.byref(m_map), len(ks), _p(ks), _p(kx), _p(ky), float(length), _p(start), _p(ls), _p(lb), _p(ub), C.byref(l0))
    return n, ls[:max(n, 0)], lb[:max(n, 0)], ub[:max(n, 0)], l0.value


def resample(ks, kx, ky, max_s, ds_smaller, ds_larger, cap=4096):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky = map(f, (ks, kx, ky))
    out = [np.zeros(cap) for _ in range(5)]
    L = lib_smooth()
    L.po_ref_resample.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 5
    n = L.po_ref_resample(len(ks), _p(ks), _p(kx), _p(ky), float(max_s), float(ds_smaller), float(ds_larger), cap, *[_p(o) for o in out])
    return n, [o[:max(n, 0)] for o in out]


def limits(v, a):
    v = np.ascontiguousarray(v, np.float64); a = np.ascontiguousarray(a, np.float64)
    mk = np.zeros(len(v)); mkp = np.zeros(len(v))
    lib_smooth().po_ref_limits(len(v), _p(v), _p(a), _p(mk), _p(mkp))
    return mk, mkp


# ---- the remaining glue stages and the reference's top-level PathOptimizer (same library) ----
def bspline(px, py, cap=4096):
    px = np.ascontiguousarray(px, np.float64); py = np.ascontiguousarray(py, np.float64)
    x = np.zeros(cap); y = np.zeros(cap); s = np.zeros(cap)
    n = lib_smooth().po_ref_bspline(len(px), _p(px), _p(py), cap, _p(x), _p(y), _p(s))
    return n, x[:n], y[:n], s[:n]


def segment_raw(ks, kx, ky, cap=4096):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky = map(f, (ks, kx, ky))
    out = [np.zeros(cap) for _ in range(5)]
    n = lib_smooth().po_ref_segment_raw(len(ks), _p(ks), _p(kx), _p(ky), cap, *[_p(o) for o in out])
    return n, [o[:max(n, 0)] for o in out]  # x, y, s, angle, k


def segment_smoothed(m_map, ks, kx, ky, length, start, goal, cap=4096):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    ks, kx, ky, start, goal = map(f, (ks, kx, ky, start, goal))
    out4 = np.zeros(4); states = np.zeros((cap, 5))
    L = lib_smooth()
    L.po_ref_segment_smoothed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    ok = L.po_ref_segment_smoothed(C.byref(m_map), len(ks), _p(ks), _p(kx), _p(ky), float(length), _p(start), _p(goal), _p(out4), cap, _p(states))
    return ok, out4[0], out4[1], out4[2], states[:int(out4[3])]


def path_optimizer_solve(m_map, params, px, py, start, goal, cap=4096):
    """PathOptimizer(start_state, end_state, map).solve(reference_points, &final_path): returns (bool, path [n,5])."""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    px, py, start, goal = map(f, (px, py, start, goal))
    path = np.zeros((cap, 5)); n = C.c_int(0)
    ok = lib_smooth().po_ref_path_optimizer_solve(C.byref(m_map), len(px), _p(px), _p(py), _p(start), _p(goal), C.byref(params), cap, _p(path), C.byref(n))
    return ok, path[:n.value]


def path_optimizer_solve_without_smoothing(m_map, params, ref, ks, kx, ky, start, goal, cap=4096):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    rx, ry, rz, rk, rs = map(f, ref)
    ks, kx, ky, start, goal = map(f, (ks, kx, ky, start, goal))
    path = np.zeros((cap, 5)); n = C.c_int(0)
    ok = lib_smooth().po_ref_path_optimizer_solve_without_smoothing(C.byref(m_map), len(rx), _p(rx), _p(ry), _p(rz), _p(rk), _p(rs), len(ks), _p(ks), _p(kx), _p(ky),
                                                                    _p(start), _p(goal), C.byref(params), cap, _p(path), C.byref(n))
    return ok, path[:n.value]


def benchmark_scene(m_map, params, px, py, start, goal, cap=4096, kcap=4096):
    """The reference's benchmark on one PathOptimizer object: solve(points) then solveWithoutSmoothing(result) (path_optimizer_benchmark.cpp).
    Returns dict(ok1, ok2, path1, path2, knot_s, knot_x, knot_y, max_s, qp1, qp2) — qp* = po_info of the path QP of each call."""
    from path_optimizer_amd.abi import PoInfo

    f = lambda a: np.ascontiguousarray(a, np.float64)
    px, py, start, goal = map(f, (px, py, start, goal))
    p1 = np.zeros((cap, 5)); p2 = np.zeros((cap, 5)); n1 = C.c_int(0); n2 = C.c_int(0)
    ks = np.zeros(kcap); kx = np.zeros(kcap); ky = np.zeros(kcap); K = C.c_int(0); max_s = C.c_double(0)
    q1, q2 = PoInfo(), PoInfo()
    rc = lib_smooth().po_ref_benchmark(C.byref(m_map), len(px), _p(px), _p(py), _p(start), _p(goal), C.byref(params), cap, _p(p1), C.byref(n1), _p(p2), C.byref(n2),
                                      kcap, _p(ks), _p(kx), _p(ky), C.byref(K), C.byref(max_s), C.byref(q1), C.byref(q2))
    g = lambda q: {k: getattr(q, k) for k, _ in PoInfo._fields_}
    k = K.value
    return dict(ok1=bool(rc & 1), ok2=bool(rc & 2), path1=p1[:n1.value], path2=p2[:n2.value], knot_s=ks[:k], knot_x=kx[:k], knot_y=ky[:k], max_s=max_s.value,
                qp1=g(q1), qp2=g(q2))
