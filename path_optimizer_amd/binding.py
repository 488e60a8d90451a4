"""ctypes binding over the C ABI of libpo_hip.so (include/po_hip.h).

Plumbing only: the product is the shared library.  There is NO CPU fallback — importing works without a
GPU (so the symbol/ABI tests can run), but every compute call goes to the HIP kernels and raises
PoError if the library or a device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from .abi import (INFO_BYTES, INFO_DTYPE, PO_ABI_VERSION, PO_ERR_HIP, PO_ERR_UNSUPPORTED, PO_OK, PoBatchIn, PoBatchOut, PoParams)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PO_LIB") or os.path.join(_HERE, "libpo_hip.so")  # PO_LIB: dev builds (make dev), A/B experiments
_LIB = None

EXPORTS = ["po_default_params", "po_problem_dims", "po_keep_control_steps", "po_create", "po_destroy", "po_set_stream", "po_debug_set", "po_device_count",
           "po_solve_batch", "po_solve_batch_device", "po_assemble_batch", "po_scaling_batch", "po_last_kernel_ms", "po_last_phase_ms", "po_debug_get", "po_strerror",
           "po_last_hip_error", "po_version", "po_set_map", "po_postcheck_batch", "po_postcheck_batch_device", "po_bounds_batch",
           "po_bounds_batch_device", "po_map_sample", "po_smooth_dims", "po_smooth_batch", "po_smooth_batch_device",
           "po_resample_batch", "po_resample_batch_device", "po_limits_batch", "po_limits_batch_device", "po_dp_search_batch",
           "po_dp_search_batch_device", "po_bspline_batch_device", "po_segment_raw_batch_device", "po_post_project_batch_device",
           "po_segment_init_batch_device", "po_plan_batch", "po_plan_batch_device", "po_densify_batch", "po_densify_batch_device"]


class PoError(RuntimeError):
    pass


_ENV_DEBUG = {"PO_IDENTITY_ORDER": "identity_order", "PO_DEBUG_CYCLES": "debug_cycles", "PO_SMOOTH_SEQ": "smooth_seq",
              "PO_SMOOTH_WAVES": "smooth_waves", "PO_SMOOTH_NOPAD": "smooth_nopad", "PO_SMOOTH_DEBUG": "smooth_debug", "PO_DP_ONE_WAVE": "dp_one_wave", "PO_NEWTON_SLICE": "newton_slice"}


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PoError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:  # if torch lives in this process too, let it load ITS libamdhip64 first so that both share one HIP runtime
            import torch  # noqa: F401
        except Exception:  # torch is optional plumbing (allocator / streams), never required by the library
            pass
        L = C.CDLL(LIB_PATH)
        L.po_strerror.restype = C.c_char_p
        L.po_last_hip_error.restype = C.c_char_p
        L.po_version.restype = C.c_char_p
        # the structs of abi.py are laid out for ONE ABI: a stale libpo_hip.so (or an older dev build picked through PO_LIB) would be driven with shifted fields
        ver = L.po_version().decode()
        parts = ver.split()
        if len(parts) < 2 or not parts[1].isdigit() or int(parts[1]) != PO_ABI_VERSION:
            raise PoError(f"{LIB_PATH} reports '{ver}' but this binding is written against PO_ABI_VERSION {PO_ABI_VERSION} (include/po_hip.h): rebuild the library")
        L.po_create.argtypes = [C.c_int, C.POINTER(PoParams), C.POINTER(C.c_void_p)]
        L.po_destroy.argtypes = [C.c_void_p]
        L.po_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.po_solve_batch.argtypes = [C.c_void_p, C.POINTER(PoBatchIn), C.POINTER(PoBatchOut)]
        L.po_solve_batch_device.argtypes = [C.c_void_p, C.POINTER(PoBatchIn), C.POINTER(PoBatchOut)]
        L.po_assemble_batch.argtypes = [C.c_void_p, C.POINTER(PoBatchIn), C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_scaling_batch.argtypes = [C.c_void_p, C.POINTER(PoBatchIn), C.c_void_p]
        L.po_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.po_last_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.po_debug_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
        _LIB = L
    return _LIB


def _check(rc: int):
    if rc != PO_OK:
        L = lib()
        msg = L.po_strerror(rc).decode()
        if rc == PO_ERR_HIP:
            msg += ": " + L.po_last_hip_error().decode()
        raise PoError(f"libpo_hip: {msg} (rc={rc})")


def default_params() -> PoParams:
    p = PoParams()
    lib().po_default_params(C.byref(p))
    return p


def problem_dims(form: int, N: int, keep: int):
    n, m, c = C.c_int(), C.c_int(), C.c_int()
    _check(lib().po_problem_dims(form, N, keep, C.byref(n), C.byref(m), C.byref(c)))
    return n.value, m.value, c.value


def smooth_dims(kind: int, P: int):
    n, m = C.c_int(), C.c_int()
    _check(lib().po_smooth_dims(kind, P, C.byref(n), C.byref(m)))
    return n.value, m.value


def keep_control_steps(form: int, ref_s) -> int:
    ref_s = np.ascontiguousarray(ref_s, dtype=np.float64)
    k = lib().po_keep_control_steps(form, ref_s.ctypes.data_as(C.c_void_p), len(ref_s))
    if k < 0:
        _check(k)
    return k


def _np(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_KEEP = []


def _i32(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.int32)
    _KEEP.append(a)  # keep alive for the duration of the call
    del _KEEP[:-4]
    return a


class Engine:
    """One handle = one HIP device + one stream (po_create / po_destroy)."""

    def __init__(self, device: int = 0, params: PoParams | None = None):
        self.params = params or default_params()
        self._h = C.c_void_p()
        _check(lib().po_create(device, C.byref(self.params), C.byref(self._h)))
        # developer conveniences of THIS Python plumbing (tools/*.py, A/B runs): PO_* environment variables are translated into po_debug_set calls
        # here; the C library itself reads no environment variable.  A switch this build does not have is
        # reported and ignored; any other failure destroys the handle before it propagates.
        try:
            for env, key in _ENV_DEBUG.items():
                v = os.environ.get(env)
                if v not in (None, "", "0"):
                    rc = lib().po_debug_set(self._h, key.encode(), int(v) if v.lstrip("-").isdigit() else 1)
                    if rc == PO_ERR_UNSUPPORTED:
                        sys.stderr.write(f"[path_optimizer_amd] {env} ignored: this build of libpo_hip.so has no '{key}' switch\n")
                    else:
                        _check(rc)
        except Exception:
            self.close()
            raise

    def debug_set(self, key: str, value: int):
        """po_debug_set: developer A/B switches (identity_order, debug_cycles, host_threads, smooth_seq, smooth_waves, smooth_nopad, smooth_debug, dp_one_wave, newton_slice)."""
        _check(lib().po_debug_set(self._h, key.encode(), int(value)))

    def debug_get(self, key: str) -> int:
        v = C.c_longlong()
        _check(lib().po_debug_get(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def close(self):
        if self._h:
            lib().po_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, raw_stream: int | None):
        _check(lib().po_set_stream(self._h, C.c_void_p(raw_stream or 0)))

    # ---- host-pointer path (H2D + solve + D2H) ----
    def solve_batch(self, batch, want_x: bool = False, order=None):
        """order: optional permutation of range(B) — scheduling hint (po_batch_in.order), results do not depend on it."""
        n, m, _ = problem_dims(batch.formulation, batch.N, batch.keep)
        bi = PoBatchIn(batch.formulation, batch.B, batch.N, batch.keep, _np(batch.ref_x), _np(batch.ref_y), _np(batch.ref_z),
                       _np(batch.ref_k), _np(batch.ref_s), _np(batch.bounds), _np(batch.x0), _np(batch.goal_z), _np(batch.max_k), _np(batch.max_kp), _np(_i32(getattr(batch, 'n_points', None))),
                       _np(_i32(order)))
        states = np.zeros((batch.B, batch.N, 5))
        info = np.zeros(batch.B, dtype=INFO_DTYPE)
        xs = np.zeros((batch.B, n)) if want_x else None
        bo = PoBatchOut(_np(states), _np(info), _np(xs))
        _check(lib().po_solve_batch(self._h, C.byref(bi), C.byref(bo)))
        return states, info, xs

    def assemble_batch(self, batch):
        n, m, _ = problem_dims(batch.formulation, batch.N, batch.keep)
        bi = PoBatchIn(batch.formulation, batch.B, batch.N, batch.keep, _np(batch.ref_x), _np(batch.ref_y), _np(batch.ref_z),
                       _np(batch.ref_k), _np(batch.ref_s), _np(batch.bounds), _np(batch.x0), _np(batch.goal_z), _np(batch.max_k), _np(batch.max_kp), _np(_i32(getattr(batch, 'n_points', None))))
        l = np.zeros((batch.B, m)); u = np.zeros((batch.B, m)); dyn = np.zeros((batch.B, batch.N - 1, 3))
        _check(lib().po_assemble_batch(self._h, C.byref(bi), _np(l), _np(u), _np(dyn)))
        return l, u, dyn

    def scaling_batch(self, batch):
        bi = PoBatchIn(batch.formulation, batch.B, batch.N, batch.keep, _np(batch.ref_x), _np(batch.ref_y), _np(batch.ref_z),
                       _np(batch.ref_k), _np(batch.ref_s), _np(batch.bounds), _np(batch.x0), _np(batch.goal_z), _np(batch.max_k), _np(batch.max_kp), _np(_i32(getattr(batch, 'n_points', None))))
        out = np.zeros((batch.B, 64))
        _check(lib().po_scaling_batch(self._h, C.byref(bi), _np(out)))
        return out

    # ---- device-pointer path: tensors are torch CUDA(=HIP) tensors already resident in HBM ----
    def solve_batch_device(self, dev: "DeviceBatch"):
        bi = PoBatchIn(dev.formulation, dev.B, dev.N, dev.keep, *(None if t is None else t.data_ptr() for t in
                       (dev.ref_x, dev.ref_y, dev.ref_z, dev.ref_k, dev.ref_s, dev.bounds, dev.x0, dev.goal_z, dev.max_k, dev.max_kp, dev.n_points,
                        getattr(dev, "order", None))))
        bo = PoBatchOut(dev.out_states.data_ptr(), dev.out_info.data_ptr(), None if dev.out_x is None else dev.out_x.data_ptr())
        _check(lib().po_solve_batch_device(self._h, C.byref(bi), C.byref(bo)))

    # ---- post-solve step ----
    def set_map(self, dist, resolution, pos_x, pos_y):
        """Upload the obstacle-distance layer dist[size_x, size_y] (float32; kept column-major like grid_map's MatrixXf)."""
        from .abi import PoMap

        d = np.asfortranarray(dist, dtype=np.float32)
        m = PoMap(d.ctypes.data_as(C.c_void_p), d.shape[0], d.shape[1], float(resolution), float(pos_x), float(pos_y))
        _check(lib().po_set_map(self._h, C.byref(m)))

    def postcheck_batch(self, states, info, n_points=None):
        """Host-pointer entry: states [B,N,5], info structured array -> n_valid [B], ok [B]."""
        states = np.ascontiguousarray(states, dtype=np.float64)
        info = np.ascontiguousarray(info)
        B, N = states.shape[0], states.shape[1]
        nv = np.zeros(B, dtype=np.int32); ok = np.zeros(B, dtype=np.int32)
        _check(lib().po_postcheck_batch(self._h, B, N, _np(_i32(n_points)), _np(states), _np(info), _np(nv), _np(ok)))
        return nv, ok

    def densify_batch(self, states, info, M: int, n_points=None):
        """optimizePath's densifying output branch (host-pointer entry): states [B,N,5], info -> out [B,M,5], n_out [B], ok [B]."""
        states = np.ascontiguousarray(states, dtype=np.float64)
        info = np.ascontiguousarray(info)
        B, N = states.shape[0], states.shape[1]
        out = np.zeros((B, M, 5)); n = np.zeros(B, dtype=np.int32); ok = np.zeros(B, dtype=np.int32)
        _check(lib().po_densify_batch(self._h, B, N, _np(_i32(n_points)), _np(states), _np(info), M, _np(out), _np(n), _np(ok)))
        return out, n, ok

    def postcheck_batch_device(self, dev: "DeviceBatch", n_valid, ok):
        """Device-pointer entry on the outputs of solve_batch_device; n_valid / ok are int32 torch tensors [B]."""
        _check(lib().po_postcheck_batch_device(self._h, dev.B, dev.N, None if dev.n_points is None else C.c_void_p(dev.n_points.data_ptr()),
                                               C.c_void_p(dev.out_states.data_ptr()), C.c_void_p(dev.out_info.data_ptr()),
                                               C.c_void_p(n_valid.data_ptr()), C.c_void_p(ok.data_ptr())))

    # ---- corridor-bounds producer ----
    def bounds_batch(self, paths: dict, n_points=None, n_knots=None):
        """Host-pointer entry. paths: dict of [B,N] ref_x/ref_y/ref_z/ref_s and [B,K] knot_s/knot_x/knot_y (synth.make_spline_paths).
        Returns bounds [B,N,4,2] (lb, ub), n_valid [B]."""
        from .abi import PoBoundsIn

        f = lambda k: np.ascontiguousarray(paths[k], dtype=np.float64)
        arr = {k: f(k) for k in ("ref_x", "ref_y", "ref_z", "ref_s", "knot_s", "knot_x", "knot_y")}
        B, N = arr["ref_x"].shape
        K = arr["knot_s"].shape[1]
        bi = PoBoundsIn(B, N, K, _np(arr["ref_x"]), _np(arr["ref_y"]), _np(arr["ref_z"]), _np(arr["ref_s"]), _np(_i32(n_points)),
                        _np(arr["knot_s"]), _np(arr["knot_x"]), _np(arr["knot_y"]), _np(_i32(n_knots)))
        bounds = np.zeros((B, N, 4, 2)); nv = np.zeros(B, dtype=np.int32)
        _check(lib().po_bounds_batch(self._h, C.byref(bi), _np(bounds), _np(nv)))
        return bounds, nv

    def bounds_batch_device(self, t: dict, bounds, n_valid):
        """Device-pointer entry: t holds torch tensors (same keys as bounds_batch), bounds [B,N,4,2] f64 and n_valid [B] i32 are outputs."""
        from .abi import PoBoundsIn

        B, N = t["ref_x"].shape
        K = t["knot_s"].shape[1]
        p = lambda k: None if t.get(k) is None else C.c_void_p(t[k].data_ptr())
        bi = PoBoundsIn(B, N, K, p("ref_x"), p("ref_y"), p("ref_z"), p("ref_s"), p("n_points"), p("knot_s"), p("knot_x"), p("knot_y"), p("n_knots"))
        _check(lib().po_bounds_batch_device(self._h, C.byref(bi), C.c_void_p(bounds.data_ptr()), C.c_void_p(n_valid.data_ptr())))

    # ---- reference-smoothing QPs ----
    def smooth_batch(self, kind: int, inp: dict, want_raw: bool = False):
        """Host-pointer entry. inp: dict of [B,P] arrays x, y, angle, k, s (TENSION2 / TENSION) or s, lb, ub + l0 [B] (POST), optional
        n_points [B] (synth.make_smooth_inputs).  Returns out_x, out_y, out_s [B,P], info [B] (+ raw [B,n_max] in the reference order)."""
        from .abi import PoSmoothIn, PoSmoothOut

        f = lambda k: None if inp.get(k) is None else np.ascontiguousarray(inp[k], dtype=np.float64)
        arr = {k: f(k) for k in ("x", "y", "angle", "k", "s", "lb", "ub", "l0")}
        B, P = arr["s"].shape
        si = PoSmoothIn(kind, B, P, _np(_i32(inp.get("n_points"))), *[_np(arr[k]) for k in ("x", "y", "angle", "k", "s", "lb", "ub", "l0")])
        ox = np.zeros((B, P)); oy = np.zeros((B, P)); os_ = np.zeros((B, P))
        info = np.zeros(B, dtype=INFO_DTYPE)
        raw = np.zeros((B, smooth_dims(kind, P)[0])) if want_raw else None
        so = PoSmoothOut(_np(ox), _np(oy), _np(os_), _np(info), _np(raw))
        _check(lib().po_smooth_batch(self._h, C.byref(si), C.byref(so)))
        return ox, oy, os_, info, raw

    def smooth_batch_device(self, kind: int, t: dict, out: dict):
        """Device-pointer entry: t / out hold torch tensors (keys as above; out: x, y, s [B,P] f64, info [B,sizeof(po_info)] u8, optional raw)."""
        from .abi import PoSmoothIn, PoSmoothOut

        B, P = t["s"].shape
        p = lambda d, k: None if d.get(k) is None else C.c_void_p(d[k].data_ptr())
        si = PoSmoothIn(kind, B, P, p(t, "n_points"), *[p(t, k) for k in ("x", "y", "angle", "k", "s", "lb", "ub", "l0")])
        so = PoSmoothOut(p(out, "x"), p(out, "y"), p(out, "s"), p(out, "info"), p(out, "raw"))
        _check(lib().po_smooth_batch_device(self._h, C.byref(si), C.byref(so)))

    # ---- reference re-sampling, limits, DP lattice search ----
    @staticmethod
    def _spline_in(sp: dict, length, n_knots=None):
        from .abi import PoSplineIn

        arr = {k: np.ascontiguousarray(sp[k], dtype=np.float64) for k in ("knot_s", "knot_x", "knot_y")}
        arr["length"] = np.ascontiguousarray(length, dtype=np.float64)
        B, K = arr["knot_s"].shape
        nk = _i32(n_knots)
        return PoSplineIn(B, K, _np(arr["knot_s"]), _np(arr["knot_x"]), _np(arr["knot_y"]), _np(nk), _np(arr["length"])), arr, B

    def resample_batch(self, sp: dict, length, ds_smaller: float, ds_larger: float, N: int, n_knots=None):
        """buildReferenceFromSpline over a batch of splines (knots knot_s/knot_x/knot_y [B,K], length [B]).  Returns dict ref_x, ref_y,
        ref_z, ref_k, ref_s [B,N] and n_points [B]."""
        si, keep, B = self._spline_in(sp, length, n_knots)
        out = {k: np.zeros((B, N)) for k in ("ref_x", "ref_y", "ref_z", "ref_k", "ref_s")}
        npts = np.zeros(B, dtype=np.int32)
        _check(lib().po_resample_batch(self._h, C.byref(si), C.c_double(ds_smaller), C.c_double(ds_larger), N, _np(out["ref_x"]), _np(out["ref_y"]),
                                       _np(out["ref_z"]), _np(out["ref_k"]), _np(out["ref_s"]), _np(npts)))
        out["n_points"] = npts
        return out

    def limits_batch(self, v, a, n_points=None):
        v = np.ascontiguousarray(v, dtype=np.float64); a = np.ascontiguousarray(a, dtype=np.float64)
        B, N = v.shape
        mk = np.zeros((B, N)); mkp = np.zeros((B, N))
        _check(lib().po_limits_batch(self._h, B, N, _np(_i32(n_points)), _np(v), _np(a), _np(mk), _np(mkp)))
        return mk, mkp

    def dp_search_batch(self, sp: dict, length, start, L: int, n_knots=None):
        """graphSearchDp over a batch.  start [B,3] = (x, y, heading).  Returns layer_s, lb, ub [B,L], l0 [B], n_layers [B]."""
        si, keep, B = self._spline_in(sp, length, n_knots)
        start = np.ascontiguousarray(start, dtype=np.float64)
        ls = np.zeros((B, L)); lb = np.zeros((B, L)); ub = np.zeros((B, L)); l0 = np.zeros(B); nl = np.zeros(B, dtype=np.int32)
        _check(lib().po_dp_search_batch(self._h, C.byref(si), _np(start), L, _np(ls), _np(lb), _np(ub), _np(l0), _np(nl)))
        return ls, lb, ub, l0, nl

    def dp_search_batch_device(self, t: dict, start, L: int, out: dict):
        """Device-pointer entry: t holds torch tensors knot_s/knot_x/knot_y [B,K], length [B]; start [B,3]; out: layer_s, lb, ub [B,L], l0 [B], n_layers [B] i32."""
        from .abi import PoSplineIn

        B, K = t["knot_s"].shape
        p = lambda d, k: None if d.get(k) is None else C.c_void_p(d[k].data_ptr())
        si = PoSplineIn(B, K, p(t, "knot_s"), p(t, "knot_x"), p(t, "knot_y"), p(t, "n_knots"), p(t, "length"))
        _check(lib().po_dp_search_batch_device(self._h, C.byref(si), C.c_void_p(start.data_ptr()), L, p(out, "layer_s"), p(out, "lb"), p(out, "ub"),
                                               p(out, "l0"), p(out, "n_layers")))

    def resample_batch_device(self, t: dict, ds_smaller: float, ds_larger: float, N: int, out: dict):
        from .abi import PoSplineIn

        B, K = t["knot_s"].shape
        p = lambda d, k: None if d.get(k) is None else C.c_void_p(d[k].data_ptr())
        si = PoSplineIn(B, K, p(t, "knot_s"), p(t, "knot_x"), p(t, "knot_y"), p(t, "n_knots"), p(t, "length"))
        _check(lib().po_resample_batch_device(self._h, C.byref(si), C.c_double(ds_smaller), C.c_double(ds_larger), N, p(out, "ref_x"), p(out, "ref_y"),
                                              p(out, "ref_z"), p(out, "ref_k"), p(out, "ref_s"), p(out, "n_points")))

    # ---- PathOptimizer::solve for a batch of planning instances ----
    def plan_batch(self, way_x, way_y, start, goal, N: int = 512, n_way=None, max_length: float = 0.0):
        """Host-pointer entry of po_plan_batch.  way_x / way_y [B,W], start [B,4] (x, y, heading, k), goal [B,3].
        Returns states [B,N,5], n_states [B], ok [B], stage [B], info [B]."""
        from .abi import PoPlanIn, PoPlanOut

        wx = np.ascontiguousarray(way_x, dtype=np.float64); wy = np.ascontiguousarray(way_y, dtype=np.float64)
        st = np.ascontiguousarray(start, dtype=np.float64); gl = np.ascontiguousarray(goal, dtype=np.float64)
        B, W = wx.shape
        pi = PoPlanIn(B, W, _np(_i32(n_way)), _np(wx), _np(wy), _np(st), _np(gl), float(max_length), N)
        states = np.zeros((B, N, 5)); n = np.zeros(B, dtype=np.int32); ok = np.zeros(B, dtype=np.int32); stage = np.zeros(B, dtype=np.int32)
        info = np.zeros(B, dtype=INFO_DTYPE)
        po = PoPlanOut(_np(states), _np(n), _np(ok), _np(stage), _np(info))
        _check(lib().po_plan_batch(self._h, C.byref(pi), C.byref(po)))
        return states, n, ok, stage, info

    def plan_batch_device(self, t: dict, out: dict, N: int, max_length: float):
        """Device-pointer entry: t: way_x, way_y [B,W], start [B,4], goal [B,3] (+ n_way); out: states [B,N,5], n_states, ok (+ stage, info [B,sizeof(po_info)] u8)."""
        from .abi import PoPlanIn, PoPlanOut

        B, W = t["way_x"].shape
        p = lambda d, k: None if d.get(k) is None else C.c_void_p(d[k].data_ptr())
        pi = PoPlanIn(B, W, p(t, "n_way"), p(t, "way_x"), p(t, "way_y"), p(t, "start"), p(t, "goal"), float(max_length), N)
        po = PoPlanOut(p(out, "states"), p(out, "n_states"), p(out, "ok"), p(out, "stage"), p(out, "info"))
        _check(lib().po_plan_batch_device(self._h, C.byref(pi), C.byref(po)))

    def map_sample(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        d = np.zeros(len(xy)); ins = np.zeros(len(xy), dtype=np.int32)
        _check(lib().po_map_sample(self._h, len(xy), _np(xy), _np(d), _np(ins)))
        return d, ins

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(lib().po_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_phase_ms(self) -> dict:
        """po_last_phase_ms: where the last solve spent its time (host-pointer entry: pack + H2D / solve / D2H / host pack / host unpack; split scheduling: the solve's own phases)."""
        ms = (C.c_float * 8)()
        _check(lib().po_last_phase_ms(self._h, ms))
        keys = ("pack_h2d", "solve", "d2h", "host_pack", "host_unpack", "warm_start", "newton", "fallback")
        return {k: float(ms[i]) for i, k in enumerate(keys)}


class DeviceBatch:
    """A synth.Batch uploaded once into HBM as torch tensors (torch is only the allocator here)."""

    def __init__(self, batch, device="cuda:0", want_x=False):
        import torch

        self.formulation, self.B, self.N, self.keep = batch.formulation, batch.B, batch.N, batch.keep
        up = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.ref_x, self.ref_y, self.ref_z, self.ref_k, self.ref_s = map(up, (batch.ref_x, batch.ref_y, batch.ref_z, batch.ref_k, batch.ref_s))
        self.bounds, self.x0, self.goal_z, self.max_k, self.max_kp = map(up, (batch.bounds, batch.x0, batch.goal_z, batch.max_k, batch.max_kp))
        npts = getattr(batch, 'n_points', None)
        self.n_points = None if npts is None else torch.from_numpy(np.ascontiguousarray(npts, dtype=np.int32)).to(device)
        n, _, _ = problem_dims(batch.formulation, batch.N, batch.keep)
        self.out_states = torch.zeros((batch.B, batch.N, 5), dtype=torch.float64, device=device)
        self.out_info = torch.zeros((batch.B, INFO_BYTES), dtype=torch.uint8, device=device)  # sizeof(po_info)
        self.out_x = torch.zeros((batch.B, n), dtype=torch.float64, device=device) if want_x else None
        self.order = None  # optional int32 [B] device tensor: scheduling hint (po_batch_in.order), see set_order()

    def set_order(self, order):
        """Scheduling hint for the next solves of this batch: a permutation of range(B), e.g. np.argsort(-previous_info["iters"], kind="stable")
        (longest path first).  None: the engine's own mixing."""
        import torch

        self.order = None if order is None else torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).to(self.out_states.device)

    def clone_outputs(self):
        """Same (shared, read-only) inputs, fresh output buffers: lets several handles solve the batch concurrently."""
        import copy

        import torch

        d = copy.copy(self)
        d.out_states = torch.zeros_like(self.out_states)
        d.out_info = torch.zeros_like(self.out_info)
        d.out_x = None if self.out_x is None else torch.zeros_like(self.out_x)
        return d

    def info_numpy(self):
        return self.out_info.cpu().numpy().view(INFO_DTYPE).reshape(-1)
