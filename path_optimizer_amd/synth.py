"""Synthetic planning instances for the BASELINE.json configs.

The reference ships no input fixtures for the QP stage beyond the benchmark's waypoint list
(/root/reference/src/test/path_optimizer_benchmark.cpp:47-82), which needs grid_map/OpenCV to turn
into corridor bounds.  These generators produce reference paths + corridor bounds of the same shape
the solver reads through ReferencePath::{getReferenceStates,getBounds,getMaxKList,getMaxKpList}
(/root/reference/include/path_optimizer/data_struct/reference_path.hpp:34-37).

RNG: numpy default_rng(SeedSequence([20260924, config_id, path_id])), draws in the order listed in
SURVEY.md §8d.  Arc length s_i = 0.25*i is exact in binary so keep_control_steps_ == 4
(the truncation quirk of solver_kp_as_input.cpp:17 is exercised separately in the tests).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

PO_KP, PO_KPC, PO_K = 0, 1, 2
SEED0 = 20260924
DS = 0.25
# FLAGS_d1..d4 defaults (planning_flags.cpp:8-14,31-37)
D_OFFSETS = (-3.0 / 8.0 * 4.9 + 1.45, -1.0 / 8.0 * 4.9 + 1.45, 1.0 / 8.0 * 4.9 + 1.45, 3.0 / 8.0 * 4.9 + 1.45)


@dataclasses.dataclass
class Batch:
    """Host-side SoA batch in the layout of po_batch_in (include/po_hip.h)."""

    formulation: int
    B: int
    N: int
    keep: int
    ref_x: np.ndarray  # [B,N]
    ref_y: np.ndarray
    ref_z: np.ndarray
    ref_k: np.ndarray
    ref_s: np.ndarray
    bounds: np.ndarray  # [B,N,4,2] (lb, ub)
    x0: np.ndarray  # [B,3]
    goal_z: np.ndarray  # [B]
    max_k: np.ndarray | None = None  # [B,N]
    max_kp: np.ndarray | None = None
    n_points: np.ndarray | None = None  # optional [B] int32: ragged batch (paths shorter than N)

    def slice(self, lo: int, hi: int) -> "Batch":
        f = lambda a: None if a is None else np.ascontiguousarray(a[lo:hi])
        return Batch(self.formulation, hi - lo, self.N, self.keep, f(self.ref_x), f(self.ref_y), f(self.ref_z),
                     f(self.ref_k), f(self.ref_s), f(self.bounds), f(self.x0), f(self.goal_z), f(self.max_k), f(self.max_kp), f(self.n_points))


CONFIGS = {
    # id: (formulation, B, N, bounds kind)
    1: (PO_KP, 1, 80, "fixed"),
    2: (PO_KP, 1024, 120, "fixed"),
    3: (PO_KP, 4096, 200, "obstacles"),
    4: (PO_KP, 32768, 200, "obstacles"),
    5: (PO_KPC, 4096, 400, "obstacles+limits"),
}


def _one_path(config_id: int, path_id: int, N: int, kind: str, form: int, DS: float = DS):
    rng = np.random.default_rng(np.random.SeedSequence([SEED0, config_id, path_id]))
    a1 = rng.uniform(0.0, 0.06)
    a2 = rng.uniform(0.0, 0.03)
    lam1 = rng.uniform(30.0, 80.0)
    lam2 = rng.uniform(8.0, 20.0)
    ph1 = rng.uniform(0.0, 2 * math.pi)
    ph2 = rng.uniform(0.0, 2 * math.pi)
    z0 = rng.uniform(-math.pi, math.pi)
    ey0 = rng.uniform(-0.3, 0.3)
    ephi0 = rng.uniform(-0.05, 0.05)
    dgoal = rng.uniform(-0.05, 0.05)
    s = DS * np.arange(N)
    k = a1 * np.sin(2 * math.pi * s / lam1 + ph1) + a2 * np.sin(2 * math.pi * s / lam2 + ph2)
    z = z0 + np.concatenate(([0.0], np.cumsum(k[:-1] * DS)))
    x = np.concatenate(([0.0], np.cumsum(np.cos(z[:-1]) * DS)))
    y = np.concatenate(([0.0], np.cumsum(np.sin(z[:-1]) * DS)))
    bounds = np.empty((N, 4, 2))
    if kind == "fixed":
        bounds[:, :, 0] = -2.0
        bounds[:, :, 1] = 2.0
    else:
        w = rng.uniform(1.8, 2.5)
        nobs = int(rng.integers(1, 4))
        left = np.full(N, w)
        right = np.full(N, w)
        idx = np.arange(N)
        for _ in range(nobs):
            side = int(rng.integers(0, 2))
            c = int(rng.integers(20, N - 20 + 1))
            hl = int(rng.integers(5, 21))
            g = rng.uniform(0.4, w - 0.5)
            t = np.where(np.abs(idx - c) <= hl, 0.5 * (1.0 + np.cos(math.pi * (idx - c) / hl)), 0.0)
            if side == 0:
                left = left - g * t
            else:
                right = right - g * t
        left = np.maximum(left, 0.5)
        right = np.maximum(right, 0.5)
        for j, d in enumerate(D_OFFSETS):
            sh = np.clip(idx + int(round(d / DS)), 0, N - 1)
            bounds[:, j, 1] = left[sh]
            bounds[:, j, 0] = -right[sh]
    max_k = max_kp = None
    if form == PO_KPC:
        lamv = rng.uniform(40.0, 120.0)
        phv = rng.uniform(0.0, 2 * math.pi)
        v = 9.0 + 6.0 * np.sin(2 * math.pi * s / lamv + phv)
        # ReferencePathImpl::updateLimits, directly-given reference (reference_path_impl.cpp:223-233), a = 0
        ay = math.sqrt((0.4 * 9.8) ** 2 - 0.0)
        max_k = ay / v ** 2
        max_kp = 0.1 / v
    x0 = np.array([ey0, ephi0, k[0]])
    return x, y, z, k, s, bounds, x0, z[-1] + dgoal, max_k, max_kp


def keep_control_steps(form: int, ref_s: np.ndarray) -> int:
    """keep_control_steps_ exactly as the reference computes it (solver.cpp:19,22-27; solver_kp_as_input.cpp:17: truncating cast)."""
    if form == PO_KPC:
        return 4
    if form == PO_K:
        return 1
    interval = 0.0
    for i in range(1, min(len(ref_s), 10)):
        interval = max(interval, float(ref_s[i]) - float(ref_s[i - 1]))
    return max(int(1.2 / interval), 1)


def make_batch(config_id: int, B: int | None = None, first_path: int = 0, N: int | None = None,
               formulation: int | None = None, ds: float = DS) -> Batch:
    """Batch of `B` paths of BASELINE config `config_id`, path ids first_path..first_path+B-1.
    ds: arc-length spacing (default 0.25 m as SURVEY.md §8d: keep_control_steps_ = 4).  ds = 0.3 with N = 231 is the shape the reference's
    own pipeline hands the QP (re-sampled spacing 0.3 m: 1.2 / 0.30000000000000004 truncates to keep = 3, solver_kp_as_input.cpp:17)."""
    form, B0, N0, kind = CONFIGS[config_id]
    if formulation is not None:
        form = formulation
    B = B0 if B is None else B
    N = N0 if N is None else N
    kind_eff = kind
    if form == PO_KPC and "limits" not in kind_eff:
        kind_eff = kind_eff + "+limits"
    rx = np.empty((B, N)); ry = np.empty((B, N)); rz = np.empty((B, N)); rk = np.empty((B, N)); rs = np.empty((B, N))
    bd = np.empty((B, N, 4, 2)); x0 = np.empty((B, 3)); gz = np.empty(B)
    mk = np.empty((B, N)) if form == PO_KPC else None
    mkp = np.empty((B, N)) if form == PO_KPC else None
    for b in range(B):
        x, y, z, k, s, bnd, xi, g, a, c = _one_path(config_id, first_path + b, N, "fixed" if kind == "fixed" else "obstacles", form, ds)
        rx[b], ry[b], rz[b], rk[b], rs[b], bd[b], x0[b], gz[b] = x, y, z, k, s, bnd, xi, g
        if form == PO_KPC:
            mk[b], mkp[b] = a, c
    keep = keep_control_steps(form, rs[0]) if B > 0 else 4
    return Batch(form, B, N, keep, rx, ry, rz, rk, rs, bd, x0, gz, mk, mkp)


def replicate(batch: Batch, B: int) -> Batch:
    """Tile a smaller batch up to B paths (used only to size throughput runs quickly)."""
    reps = -(-B // batch.B)
    f = lambda a: None if a is None else np.ascontiguousarray(np.concatenate([a] * reps, axis=0)[:B])
    return Batch(batch.formulation, B, batch.N, batch.keep, f(batch.ref_x), f(batch.ref_y), f(batch.ref_z), f(batch.ref_k),
                 f(batch.ref_s), f(batch.bounds), f(batch.x0), f(batch.goal_z), f(batch.max_k), f(batch.max_kp))


def make_distance_map(seed: int, size_x: int = 400, size_y: int = 300, resolution: float = 0.2, pos=(10.0, -5.0), n_obstacles: int = 25,
                      r_range=(0.5, 2.5)):
    """Synthetic obstacle-distance layer (stand-in for cv::distanceTransform * resolution of the benchmark's PNG,
    /root/reference/src/test/path_optimizer_benchmark.cpp:28-44): distance from each cell centre to the nearest of
    `n_obstacles` random discs (0 inside a disc), float32 [size_x, size_y] in grid_map's index convention
    (cell (0,0) at the largest x and y).  Returns dist, resolution, pos_x, pos_y, discs [n,3] (x, y, r)."""
    rng = np.random.default_rng(np.random.SeedSequence([SEED0, 77, seed]))
    lx, ly = size_x * resolution, size_y * resolution
    cx = pos[0] + 0.5 * lx - (np.arange(size_x) + 0.5) * resolution
    cy = pos[1] + 0.5 * ly - (np.arange(size_y) + 0.5) * resolution
    discs = np.stack([rng.uniform(pos[0] - 0.5 * lx, pos[0] + 0.5 * lx, n_obstacles), rng.uniform(pos[1] - 0.5 * ly, pos[1] + 0.5 * ly, n_obstacles),
                      rng.uniform(r_range[0], r_range[1], n_obstacles)], axis=1)
    X, Y = np.meshgrid(cx, cy, indexing="ij")
    d = np.full((size_x, size_y), np.inf)
    for ox, oy, r in discs:
        d = np.minimum(d, np.hypot(X - ox, Y - oy) - r)
    return np.maximum(d, 0.0).astype(np.float32), resolution, pos[0], pos[1], discs


def make_spline_paths(seed: int, B: int, N: int = 200, ds: float = 0.3, knot_ds: float = 1.5, start_box: float = 8.0):
    """Reference paths for the corridor-bounds producer: smooth planar curves given as spline knots (s_k, x_k, y_k) every
    `knot_ds` metres (what ReferencePath::setSpline is built from) plus N reference states sampled every `ds` metres
    (x, y, heading, s) as buildReferenceFromSpline would hand them over.  Returns dict of arrays [B, ...]."""
    rng = np.random.default_rng(np.random.SeedSequence([SEED0, 78, seed]))
    length = (N - 1) * ds
    K = int(math.ceil(length / knot_ds)) + 4
    ks = knot_ds * np.arange(K)
    out = dict(ref_x=np.zeros((B, N)), ref_y=np.zeros((B, N)), ref_z=np.zeros((B, N)), ref_s=np.tile(ds * np.arange(N), (B, 1)),
               knot_s=np.tile(ks, (B, 1)), knot_x=np.zeros((B, K)), knot_y=np.zeros((B, K)))
    fine = np.linspace(0, ks[-1], 8 * K)
    for b in range(B):
        k = rng.uniform(0, 0.05) * np.sin(2 * math.pi * fine / rng.uniform(25, 70) + rng.uniform(0, 6.28)) + rng.uniform(-0.01, 0.01)
        z = rng.uniform(-math.pi, math.pi) + np.concatenate(([0.0], np.cumsum(0.5 * (k[1:] + k[:-1]) * np.diff(fine))))
        x = rng.uniform(-start_box, start_box) + np.concatenate(([0.0], np.cumsum(np.cos(0.5 * (z[1:] + z[:-1])) * np.diff(fine))))
        y = rng.uniform(-start_box, start_box) + np.concatenate(([0.0], np.cumsum(np.sin(0.5 * (z[1:] + z[:-1])) * np.diff(fine))))
        out["knot_x"][b] = np.interp(ks, fine, x); out["knot_y"][b] = np.interp(ks, fine, y)
        s = out["ref_s"][b]
        out["ref_x"][b] = np.interp(s, fine, x); out["ref_y"][b] = np.interp(s, fine, y); out["ref_z"][b] = np.interp(s, fine, z)
    return out


def make_smooth_inputs(seed: int, B: int, P: int = 100, kind: int = 0, ragged: bool = False, jitter_ds: bool = False):
    """Inputs of the reference-smoothing QPs in the layout of po_smooth_in.
    kind 0/1 (TENSION2 / TENSION): the five lists ReferencePathSmoother::segmentRawReference hands to osqpSmooth
    (/root/reference/src/reference_path_smoother/reference_path_smoother.cpp:50-91): a raw planner path = smooth curve + a lateral
    wiggle of a few metres wavelength, sampled every 1.0 m (the reference's fixed delta_s), with heading and curvature of the raw
    curve at the samples.  kind 2 (POST): layers_s_list_ every FLAGS_search_longitudial_spacing = 1.5 m (last gap shorter), corridor
    bounds layers_bounds_ as graphSearchDp leaves them (multiples of 0.2 m around the chosen lattice node, within +-6 m), and the
    vehicle's lateral offset.  jitter_ds: non-uniform spacing (test only).  Returns dict of [B,P] arrays (+ n_points if ragged)."""
    rng = np.random.default_rng(np.random.SeedSequence([SEED0, 79, seed, kind]))
    out = dict(x=np.zeros((B, P)), y=np.zeros((B, P)), angle=np.zeros((B, P)), k=np.zeros((B, P)), s=np.zeros((B, P)),
               lb=np.zeros((B, P)), ub=np.zeros((B, P)), l0=np.zeros(B))
    npts = np.full(B, P, dtype=np.int32)
    for b in range(B):
        n = P if not ragged else int(rng.integers(max(4, P // 3), P + 1))
        npts[b] = n
        if kind == 2:
            gaps = np.full(n - 1, 1.5)
            gaps[-1] = rng.uniform(0.2, 1.5)
            if jitter_ds:
                gaps *= rng.uniform(0.6, 1.2, n - 1)
            s = np.concatenate(([0.0], np.cumsum(gaps)))
            centre = 1.5 * np.sin(2 * math.pi * s / rng.uniform(25, 60) + rng.uniform(0, 6.28)) * rng.uniform(0, 1)
            lo = centre - 0.2 * rng.integers(2, 12, n)
            hi = centre + 0.2 * rng.integers(2, 12, n)
            out["lb"][b, :n] = np.maximum(np.round(lo / 0.2) * 0.2, -6.0)
            out["ub"][b, :n] = np.minimum(np.round(hi / 0.2) * 0.2, 6.0)
            out["lb"][b, 0], out["ub"][b, 0] = -10.0, 10.0
            out["l0"][b] = rng.uniform(-0.5, 0.5)
            out["s"][b, :n] = s
            continue
        gaps = np.ones(n - 1)
        if jitter_ds:
            gaps = rng.uniform(0.6, 1.4, n - 1)
        s = np.concatenate(([0.0], np.cumsum(gaps)))
        fine = np.linspace(0.0, s[-1], 40 * n)
        kk = rng.uniform(0, 0.06) * np.sin(2 * math.pi * fine / rng.uniform(30, 80) + rng.uniform(0, 6.28))
        z = rng.uniform(-math.pi, math.pi) + np.concatenate(([0.0], np.cumsum(0.5 * (kk[1:] + kk[:-1]) * np.diff(fine))))
        xb = rng.uniform(-8, 8) + np.concatenate(([0.0], np.cumsum(np.cos(0.5 * (z[1:] + z[:-1])) * np.diff(fine))))
        yb = rng.uniform(-8, 8) + np.concatenate(([0.0], np.cumsum(np.sin(0.5 * (z[1:] + z[:-1])) * np.diff(fine))))
        e = rng.uniform(0.05, 0.3) * np.sin(2 * math.pi * fine / rng.uniform(4, 12) + rng.uniform(0, 6.28))
        xr, yr = xb - e * np.sin(z), yb + e * np.cos(z)
        dx, dy = np.gradient(xr, fine), np.gradient(yr, fine)
        ddx, ddy = np.gradient(dx, fine), np.gradient(dy, fine)
        ang = np.arctan2(dy, dx)
        cur = (dx * ddy - dy * ddx) / np.power(dx * dx + dy * dy, 1.5)
        idx = np.clip(np.searchsorted(fine, s), 0, len(fine) - 1)
        out["x"][b, :n], out["y"][b, :n], out["angle"][b, :n], out["k"][b, :n], out["s"][b, :n] = xr[idx], yr[idx], ang[idx], cur[idx], s
    if ragged:
        out["n_points"] = npts
    return out


def make_search_inputs(seed: int, B: int, N: int = 200, ds: float = 0.3, max_offset: float = 0.6):
    """Inputs of the DP lattice search / re-sampling: spline reference paths (make_spline_paths), the arc length the
    reference attaches to each (shorter than the last knot), and a vehicle start state near the beginning of the path (lateral offset
    within +-max_offset, small heading error).  Returns (paths dict, length [B], start [B,3])."""
    sp = make_spline_paths(seed, B, N=N, ds=ds)
    rng = np.random.default_rng(np.random.SeedSequence([SEED0, 80, seed]))
    length = sp["ref_s"][:, -1] - rng.uniform(0.0, 3.0, B)
    i0 = rng.integers(0, 8, B)
    rows = np.arange(B)
    z0 = sp["ref_z"][rows, i0]
    off = rng.uniform(-max_offset, max_offset, B)
    start = np.stack([sp["ref_x"][rows, i0] - off * np.sin(z0), sp["ref_y"][rows, i0] + off * np.cos(z0), z0 + rng.uniform(-0.1, 0.1, B)], axis=1)
    return sp, np.ascontiguousarray(length), np.ascontiguousarray(start)


def make_planning_scenes(seed: int, B: int, n_way: int = 24, way_ds: float = 3.0, noise: float = 0.25, map_kw: dict | None = None, clear: float = 3.2,
                         n_discs: int = 60, near: int = 0):
    """Whole planning instances for PathOptimizer::solve (/root/reference/src/test/path_optimizer_benchmark.cpp:47-100 hands it a
    list of waypoints, a start and a goal state over an obstacle-distance map): a smooth curve inside the map, waypoints every `way_ds`
    metres jittered laterally by +-`noise` (what a coarse planner returns), start = first waypoint with the curve's heading, goal =
    last waypoint; obstacles (discs) kept >= 3 m away from the curve's first 6 m and otherwise scattered beside it.
    Returns dict: way_x, way_y [B,n_way], start [B,4] (x, y, heading, k), goal [B,3], and the shared map tuple (dist, res, px, py)."""
    rng = np.random.default_rng(np.random.SeedSequence([SEED0, 81, seed]))
    kw = dict(size_x=700, size_y=700, resolution=0.2, pos=(0.0, 0.0))
    kw.update(map_kw or {})
    lx, ly = kw["size_x"] * kw["resolution"], kw["size_y"] * kw["resolution"]
    out = dict(way_x=np.zeros((B, n_way)), way_y=np.zeros((B, n_way)), start=np.zeros((B, 4)), goal=np.zeros((B, 3)))
    curves = []
    for b in range(B):
        fine = np.linspace(0.0, way_ds * (n_way - 1), 40 * n_way)
        kk = rng.uniform(0.0, 0.035) * np.sin(2 * math.pi * fine / rng.uniform(40, 90) + rng.uniform(0, 6.28))
        z = rng.uniform(-math.pi, math.pi) + np.concatenate(([0.0], np.cumsum(0.5 * (kk[1:] + kk[:-1]) * np.diff(fine))))
        x = np.concatenate(([0.0], np.cumsum(np.cos(0.5 * (z[1:] + z[:-1])) * np.diff(fine))))
        y = np.concatenate(([0.0], np.cumsum(np.sin(0.5 * (z[1:] + z[:-1])) * np.diff(fine))))
        x += kw["pos"][0] - 0.5 * (x.max() + x.min()) + rng.uniform(-10, 10)
        y += kw["pos"][1] - 0.5 * (y.max() + y.min()) + rng.uniform(-10, 10)
        idx = np.searchsorted(fine, way_ds * np.arange(n_way)).clip(0, len(fine) - 1)
        e = rng.uniform(-noise, noise, n_way)
        e[0] = e[-1] = 0.0
        out["way_x"][b] = x[idx] - e * np.sin(z[idx]); out["way_y"][b] = y[idx] + e * np.cos(z[idx])
        out["start"][b] = [x[0], y[0], z[0], kk[0]]
        out["goal"][b] = [x[idx[-1]], y[idx[-1]], z[idx[-1]]]
        curves.append((x, y, fine))
    # one shared map: discs beside the curves, never closer than `clear` to any curve point (wider near the starts)
    discs = []
    tries = 0
    while len(discs) < n_discs and tries < 4000:
        tries += 1
        ox, oy = rng.uniform(kw["pos"][0] - 0.45 * lx, kw["pos"][0] + 0.45 * lx), rng.uniform(kw["pos"][1] - 0.45 * ly, kw["pos"][1] + 0.45 * ly)
        r = rng.uniform(0.5, 2.0)
        okd = True
        for x, y, fine in curves:
            d = np.hypot(x - ox, y - oy) - r
            if d.min() < clear or d[fine < 8.0].min() < 5.0:
                okd = False
                break
        if okd:
            discs.append((ox, oy, r))
    for x, y, fine in curves:  # `near` discs per curve right beside it (1.3 .. 2.6 m of free space), wherever the other curves run
        for _ in range(near):
            i = int(rng.integers(np.searchsorted(fine, 10.0), len(fine) - 1))
            r = rng.uniform(0.5, 1.5)
            gap = r + rng.uniform(1.3, 2.6)
            hz = math.atan2(y[i + 1] - y[i - 1], x[i + 1] - x[i - 1]) + (math.pi / 2 if rng.uniform() < 0.5 else -math.pi / 2)
            discs.append((x[i] + gap * math.cos(hz), y[i] + gap * math.sin(hz), r))
    cx = kw["pos"][0] + 0.5 * lx - (np.arange(kw["size_x"]) + 0.5) * kw["resolution"]
    cy = kw["pos"][1] + 0.5 * ly - (np.arange(kw["size_y"]) + 0.5) * kw["resolution"]
    X, Y = np.meshgrid(cx, cy, indexing="ij")
    d = np.full((kw["size_x"], kw["size_y"]), 30.0)
    for ox, oy, r in discs:
        d = np.minimum(d, np.hypot(X - ox, Y - oy) - r)
    out["map"] = (np.maximum(d, 0.0).astype(np.float32), kw["resolution"], kw["pos"][0], kw["pos"][1])
    return out
