// po_post.hip — kernels around the QP hot path: post-solve collision check, map sampling, corridor-bounds producer.
// Compiled with -ffp-contract=off (see Makefile): these kernels mirror plain IEEE double arithmetic of the reference's C++,
// expression by expression, so that thresholds (clearance < radius) fall on the same side as on the CPU.
#include <hip/hip_runtime.h>

#include "../../include/po_hip.h"
#include "../../include/po_pmath.h"  // portable sin / cos / atan2: the same IEEE operation sequence as the oracle's portable-math mode (bit-exact map stages)
#define PO_MAP_DEVICE_CODE
#include "po_map.hpp"

namespace po {

// One block per path: first colliding state, then optimizePath's return value (path_optimizer.cpp:183-200).
__global__ __launch_bounds__(128) void postcheck_kernel(DevMap m, DevCar c, int B, int N, const int *n_points, const double *states,
                                                        const po_info *info, int *n_valid, int *ok) {
    const int b = blockIdx.x;
    __shared__ int first;
    int n = n_points ? n_points[b] : N;
    n = n < 0 ? 0 : (n > N ? N : n);
    if (threadIdx.x == 0) first = n;
    __syncthreads();
    const bool solved = info[b].status == PO_STATUS_SOLVED;
    if (solved && c.enable) {
        const double *s = states + (size_t)b * N * 5;
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (!collision_free(m, c, s[5 * i], s[5 * i + 1], s[5 * i + 2])) atomicMin(&first, i);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!solved) { n_valid[b] = 0; ok[b] = 0; }
        else if (first >= n) { n_valid[b] = n; ok[b] = 1; }
        else { n_valid[b] = first; ok[b] = (first > 0 && states[((size_t)b * N + first - 1) * 5 + 4] >= 20.0) ? 1 : 0; }
    }
}

__global__ void map_sample_kernel(DevMap m, int n, const double *xy, double *dist, int *inside) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dist[i] = map_distance(m, xy[2 * i], xy[2 * i + 1]);
    inside[i] = map_inside(m, xy[2 * i], xy[2 * i + 1]) ? 1 : 0;
}


// ---- natural cubic spline through (x_i, y_i): what the reference gets from tk::spline::set_points (src/tools/spline.cpp:154-271, a third-party GPL file that is
// NOT reproduced here).  Derived independently from the textbook formulation: the second derivatives M_i solve the tridiagonal system
//   h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} = 6 ((y_{i+1} - y_i) / h_i - (y_i - y_{i-1}) / h_{i-1}),   M_0 = M_{K-1} = 0,
// by the Thomas algorithm; the piece on [x_i, x_{i+1}] is  a_i (t - x_i)^3 + b_i (t - x_i)^2 + c_i (t - x_i) + y_i  with  b_i = M_i / 2,
// a_i = (M_{i+1} - M_i) / (6 h_i),  c_i = (y_{i+1} - y_i) / h_i - h_i (2 M_i + M_{i+1}) / 6; beyond the last knot the reference's class continues with the end
// slope and zero curvature, which is what entry K-1 holds.  Same spline as the reference's to a few ulp (its LU on the scaled system rounds differently).
// One thread per spline; a, b, c [K] out, w = 3K doubles of scratch.
__device__ void spline_fit(int K, const double *x, const double *y, double *a, double *b, double *c, double *w) {
    double *cp = w, *dp = w + K, *M = w + 2 * K;  // Thomas: modified super-diagonal, modified rhs, solution
    cp[0] = 0.0; dp[0] = 0.0;                     // row 0: M_0 = 0
    for (int i = 1; i < K - 1; ++i) {
        const double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
        const double rhs = 6.0 * ((y[i + 1] - y[i]) / hr - (y[i] - y[i - 1]) / hl);
        const double piv = 2.0 * (hl + hr) - hl * cp[i - 1];
        cp[i] = hr / piv;
        dp[i] = (rhs - hl * dp[i - 1]) / piv;
    }
    M[K - 1] = 0.0;
    for (int i = K - 2; i >= 1; --i) M[i] = dp[i] - cp[i] * M[i + 1];
    M[0] = 0.0;
    for (int i = 0; i < K - 1; ++i) {
        const double h = x[i + 1] - x[i];
        b[i] = 0.5 * M[i];
        a[i] = (M[i + 1] - M[i]) / (6.0 * h);
        c[i] = (y[i + 1] - y[i]) / h - h * (2.0 * M[i] + M[i + 1]) / 6.0;
    }
    const double h = x[K - 1] - x[K - 2];
    b[K - 1] = 0.0;
    a[K - 1] = 0.0;
    c[K - 1] = 3.0 * a[K - 2] * h * h + 2.0 * b[K - 2] * h + c[K - 2];
}
__device__ __forceinline__ double spline_eval(int K, const double *x, const double *y, const double *a, const double *b, const double *c, double at) {
    int lo = 0, hi = K;  // std::lower_bound
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (x[mid] < at) lo = mid + 1; else hi = mid; }
    const int idx = lo - 1 > 0 ? lo - 1 : 0;
    const double h = at - x[idx];
    if (at < x[0]) return (b[0] * h + c[0]) * h + y[0];
    if (at > x[K - 1]) return (b[K - 1] * h + c[K - 1]) * h + y[K - 1];
    return ((a[idx] * h + b[idx]) * h + c[idx]) * h + y[idx];
}
__device__ __forceinline__ int knots_of(const DevBounds &in, int b) {
    int k = in.n_knots ? in.n_knots[b] : in.K;
    return k < 3 ? 3 : (k > in.K ? in.K : k);
}
__global__ void spline_fit_kernel(DevBounds in) {
    const int b = blockIdx.x, which = threadIdx.x;  // 0: x(s), 1: y(s)
    if (which > 1) return;
    const int K = knots_of(in, b);
    double *co = in.coef + ((size_t)b * 2 + which) * 6 * in.K;
    spline_fit(K, in.knot_s + (size_t)b * in.K, (which ? in.knot_y : in.knot_x) + (size_t)b * in.K, co, co + in.K, co + 2 * in.K, co + 3 * in.K);
}
__device__ __forceinline__ double wrap_pi(double a) {  // constraintAngle, tools.hpp:24-35
    for (int it = 0; it < 64; ++it) {
        if (a > M_PI) a -= 2 * M_PI;
        else if (a < -M_PI) a += 2 * M_PI;
        else break;
    }
    return a;
}
// Lateral clearance of one covering circle: what ReferencePathImpl::getClearanceWithDirectionStrict returns with the shipped
// FLAGS_enable_simple_boundary_decision = true (reference_path_impl.cpp:283-472).  Written here as rays and marches; the arithmetic of every probe position and of the
// two results is the reference's (same operands in the same order: a threshold on an interpolated distance must fall on the same side).
//   * two rays from the circle centre along the left / right normal of the heading;
//   * coarse march in 0.5 m steps, at most 10: from a free centre outwards until a probe is blocked (clearance < radius), the bound is one step back; from a blocked centre
//     first outwards on both sides until a probe is free, the nearer side wins, the corridor lies wholly on that side: one bound where it becomes free, the other where it is
//     blocked again;
//   * fine march: each bound pushed outwards in 0.1 m steps, at most 4, and taken back one step when the probe is blocked.
__device__ void clearance_strict(const DevMap &m, double radius, double sx, double sy, double sz, double &left_bound, double &right_bound) {
    struct Ray { double c, s; };
    const double hl = wrap_pi(sz + M_PI_2), hr = wrap_pi(sz - M_PI_2);
    const Ray L{po_pcos(hl), po_psin(hl)}, R{po_pcos(hr), po_psin(hr)};
    constexpr double kCoarse = 0.5, kFine = 0.1;
    const int n_coarse = (int)(5.0 / kCoarse), n_fine = (int)(kCoarse / kFine);
    auto probe = [&](const Ray &r, double d) { return map_distance(m, sx + d * r.c, sy + d * r.s); };
    auto march = [&](const Ray &r, double d, bool until_blocked) {  // distance reached when the probe first is blocked / free (or after n_coarse steps)
        for (int k = 0; k != n_coarse; ++k) {
            d += kCoarse;
            const double clr = probe(r, d);
            if (until_blocked ? clr < radius : clr > radius) break;
        }
        return d;
    };
    if (map_distance(m, sx, sy) > radius) {
        const double dr = march(R, 0.0, true), dl = march(L, 0.0, true);
        right_bound = -(dr - kCoarse);
        left_bound = dl - kCoarse;
    } else {
        const double fr = march(R, 0.0, false), fl = march(L, 0.0, false);
        if (fl < fr) {
            right_bound = fl;
            left_bound = march(L, fl, true) - kCoarse;
        } else {
            left_bound = -fr;
            right_bound = -(march(R, fr, true) - kCoarse);
        }
    }
    for (int k = 1; k != n_fine; ++k) {
        left_bound += kFine;
        if (probe(L, left_bound) < radius) { left_bound -= kFine; break; }
    }
    for (int k = 1; k != n_fine; ++k) {
        right_bound -= kFine;
        if (probe(R, right_bound) < radius) { right_bound += kFine; break; }
    }
}
// One block per path, one thread per (state, circle): updateBoundsImproved (:142-201)
__global__ __launch_bounds__(256) void bounds_kernel(DevMap m, DevBounds in, double *bounds, int *n_valid) {
    const int b = blockIdx.x;
    __shared__ int first;
    int n = in.n_points ? in.n_points[b] : in.N;
    n = n < 0 ? 0 : (n > in.N ? in.N : n);
    if (threadIdx.x == 0) first = n;
    __syncthreads();
    const int K = knots_of(in, b);
    const double *ks = in.knot_s + (size_t)b * in.K, *kx = in.knot_x + (size_t)b * in.K, *ky = in.knot_y + (size_t)b * in.K;
    const double *cx_ = in.coef + ((size_t)b * 2) * 6 * in.K, *cy_ = cx_ + 6 * in.K;
    double *out = bounds + (size_t)b * in.N * 8;
    for (int t = threadIdx.x; t < 4 * in.N; t += blockDim.x) {
        const int i = t >> 2, j = t & 3;
        double lb = 0, ub = 0;
        if (i < n) {
            const size_t o = (size_t)b * in.N + i;
            const double x = in.ref_x[o], y = in.ref_y[o], z = in.ref_z[o], s = in.ref_s[o];
            const double cz = po_pcos(z), sz = po_psin(z), len = in.d[j];
            const double ccx = x + len * cz, ccy = y + len * sz;
            // getApproxState (:121-140)
            const double px = spline_eval(K, ks, kx, cx_, cx_ + in.K, cx_ + 2 * in.K, s + len);
            const double py = spline_eval(K, ks, ky, cy_, cy_ + in.K, cy_ + 2 * in.K, s + len);
            const double v1x = ccx - x, v1y = ccy - y, v2x = px - x, v2y = py - y;
            const double nrm = sqrt(v1x * v1x + v1y * v1y);
            const double proj = (v1x * v2x + v1y * v2y) / (0.001 > nrm ? 0.001 : nrm);
            const double move = fabs(len) - proj;
            const int sign = len >= 0 ? 1 : -1;
            const double ax = px + sign * move * cz, ay = py + sign * move * sz;
            double l, r;
            clearance_strict(m, in.radius, ax, ay, z, l, r);
            const double dx = ax - ccx, dy = ay - ccy;
            const double off = -dx * sz + dy * cz;  // global2Local(c_j, c_jj).y
            ub = l + off; lb = r + off;
            if (fabs(ub - lb) < 1e-6) atomicMin(&first, i);  // isEqual -> "Path is blocked"
        }
        out[2 * t] = lb; out[2 * t + 1] = ub;
    }
    __syncthreads();
    const int nv = first;
    for (int t = threadIdx.x; t < 4 * in.N; t += blockDim.x)
        if ((t >> 2) >= nv) { out[2 * t] = 0; out[2 * t + 1] = 0; }
    if (threadIdx.x == 0) n_valid[b] = nv;
}


// =====================================================================================================================
// reference re-sampling, limits, DP lattice search
// =====================================================================================================================
// tk::spline::deriv, src/tools/spline.cpp:273-318 (the order-2 left-extrapolation branch keeps its `* h`, as written there)
__device__ __forceinline__ double spline_deriv(int K, const double *x, const double *a, const double *b, const double *c, int order, double at) {
    int lo = 0, hi = K;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (x[mid] < at) lo = mid + 1; else hi = mid; }
    const int idx = lo - 1 > 0 ? lo - 1 : 0;
    const double h = at - x[idx];
    if (at < x[0]) return order == 1 ? 2.0 * b[0] * h + c[0] : 2.0 * b[0] * h;
    if (at > x[K - 1]) return order == 1 ? 2.0 * b[K - 1] * h + c[K - 1] : 2.0 * b[K - 1];
    return order == 1 ? (3.0 * a[idx] * h + 2.0 * b[idx]) * h + c[idx] : 6.0 * a[idx] * h + 2.0 * b[idx];
}
// the pair x(s), y(s) of one path, staged in LDS: s, x, y knots and a, b, c of both splines = 9 arrays of K doubles
// the interval search of spline_eval / spline_deriv done once per argument (seven dependent LDS reads for K ~ 70): same arithmetic afterwards
struct SplAt { int idx, side; double h; };  // side: -1 left of the first knot, +1 right of the last, 0 inside
__device__ __forceinline__ SplAt spline_at(int K, const double *x, double at) {
    int lo = 0, hi = K;  // std::lower_bound
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (x[mid] < at) lo = mid + 1; else hi = mid; }
    SplAt p;
    p.idx = lo - 1 > 0 ? lo - 1 : 0;
    p.h = at - x[p.idx];
    p.side = at < x[0] ? -1 : (at > x[K - 1] ? 1 : 0);
    return p;
}
__device__ __forceinline__ double spline_eval_at(int K, const double *y, const double *a, const double *b, const double *c, const SplAt &p) {
    if (p.side < 0) return (b[0] * p.h + c[0]) * p.h + y[0];
    if (p.side > 0) return (b[K - 1] * p.h + c[K - 1]) * p.h + y[K - 1];
    return ((a[p.idx] * p.h + b[p.idx]) * p.h + c[p.idx]) * p.h + y[p.idx];
}
__device__ __forceinline__ double spline_deriv_at(int K, const double *a, const double *b, const double *c, int order, const SplAt &p) {
    if (p.side < 0) return order == 1 ? 2.0 * b[0] * p.h + c[0] : 2.0 * b[0] * p.h;
    if (p.side > 0) return order == 1 ? 2.0 * b[K - 1] * p.h + c[K - 1] : 2.0 * b[K - 1];
    return order == 1 ? (3.0 * a[p.idx] * p.h + 2.0 * b[p.idx]) * p.h + c[p.idx] : 6.0 * a[p.idx] * p.h + 2.0 * b[p.idx];
}
struct Spl2 {
    int K;
    const double *s, *vx, *vy, *ax, *bx, *cx, *ay, *by, *cy;
    __device__ __forceinline__ SplAt at(double t) const { return spline_at(K, s, t); }
    __device__ __forceinline__ double x(const SplAt &p) const { return spline_eval_at(K, vx, ax, bx, cx, p); }
    __device__ __forceinline__ double y(const SplAt &p) const { return spline_eval_at(K, vy, ay, by, cy, p); }
    __device__ __forceinline__ double dx(int o, const SplAt &p) const { return spline_deriv_at(K, ax, bx, cx, o, p); }
    __device__ __forceinline__ double dy(int o, const SplAt &p) const { return spline_deriv_at(K, ay, by, cy, o, p); }
    __device__ __forceinline__ double heading(const SplAt &p) const { return po_patan2(dy(1, p), dx(1, p)); }
    __device__ __forceinline__ double curvature(const SplAt &p) const {
        const double x1 = dx(1, p), y1 = dy(1, p), x2 = dx(2, p), y2 = dy(2, p);
        return (x1 * y2 - y1 * x2) / po_ppow15(x1 * x1 + y1 * y1);
    }
    __device__ __forceinline__ double x(double t) const { return x(at(t)); }
    __device__ __forceinline__ double y(double t) const { return y(at(t)); }
    __device__ __forceinline__ double dx(int o, double t) const { return dx(o, at(t)); }
    __device__ __forceinline__ double dy(int o, double t) const { return dy(o, at(t)); }
    __device__ __forceinline__ double heading(double t) const { return heading(at(t)); }      // getHeading, tools.cpp:34-38
    __device__ __forceinline__ double curvature(double t) const { return curvature(at(t)); }  // getCurvature, tools.cpp:40-46
};
// Knots into LDS, then the two natural-spline fits (x(s) and y(s)) by lanes 0 and 1 right there: 15 K doubles of LDS
// (s, x, y | a, b, c + 3 K scratch for each spline).  No separate fit kernel, no coefficient round trip through HBM.
__device__ __forceinline__ Spl2 stage_spline(const DevSpline &in, int b, double *lds) {
    int K = in.n_knots ? in.n_knots[b] : in.K;
    K = K < 3 ? 3 : (K > in.K ? in.K : K);
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        lds[i] = in.knot_s[(size_t)b * in.K + i];
        lds[K + i] = in.knot_x[(size_t)b * in.K + i];
        lds[2 * K + i] = in.knot_y[(size_t)b * in.K + i];
    }
    __syncthreads();
    double *cx = lds + 3 * K, *cy = lds + 9 * K;
    if (threadIdx.x < 2) {
        double *co = threadIdx.x ? cy : cx;
        spline_fit(K, lds, threadIdx.x ? lds + 2 * K : lds + K, co, co + K, co + 2 * K, co + 3 * K);
    }
    __syncthreads();
    Spl2 S{K, lds, lds + K, lds + 2 * K, cx, cx + K, cx + 2 * K, cy, cy + K, cy + 2 * K};
    return S;
}

// ReferencePathImpl::buildReferenceFromSpline (reference_path_impl.cpp:474-499).  One block per path: lane 0 walks the arc-length
// sequence (each step depends on the curvature at the previous one), then all lanes fill x, y, heading of the emitted states.
__global__ __launch_bounds__(64) void resample_kernel(DevSpline in, DevResample r) {
    extern __shared__ double lds[];
    const int b = blockIdx.x;
    const size_t o = (size_t)b * r.N;
    const double max_s = in.length[b];
    __shared__ int s_n;
    if (max_s <= 0 || (in.n_knots && (in.n_knots[b] < 3 || in.n_knots[b] > in.K))) {  // "Cannot build reference line from spline!"
        for (int i = threadIdx.x; i < r.N; i += 64) { r.x[o + i] = 0; r.y[o + i] = 0; r.z[o + i] = 0; r.k[o + i] = 0; r.s[o + i] = 0; }
        if (threadIdx.x == 0) r.n_points[b] = -1;
        return;
    }
    const Spl2 S = stage_spline(in, b, lds);
    if (threadIdx.x == 0) {
        const double large_k = 0.2, small_k = 0.08;
        double tmp_s = 0;
        int n = 0;
        while (tmp_s <= max_s) {
            if (n == r.N) { n = -2; break; }
            const double k = S.curvature(tmp_s);
            r.k[o + n] = k; r.s[o + n] = tmp_s;
            ++n;
            if (r.dynamic) {
                const double k_share = fabs(k) > large_k ? 1 : fabs(k) < small_k ? 0 : (fabs(k) - small_k) / (large_k - small_k);
                tmp_s += r.ds_large - k_share * (r.ds_large - r.ds_small);
            } else tmp_s += r.ds_large;
        }
        s_n = n;
        r.n_points[b] = n;
    }
    __threadfence_block();
    __syncthreads();
    const int n = s_n < 0 ? 0 : s_n;
    for (int i = threadIdx.x; i < r.N; i += 64) {
        if (i < n) {
            const SplAt pi_ = S.at(r.s[o + i]);
            r.x[o + i] = S.x(pi_); r.y[o + i] = S.y(pi_); r.z[o + i] = S.heading(pi_);
        } else { r.x[o + i] = 0; r.y[o + i] = 0; r.z[o + i] = 0; r.k[o + i] = 0; r.s[o + i] = 0; }
    }
}

// ReferencePathImpl::updateLimits (reference_path_impl.cpp:203-235), states-given-directly branch.  pow(x, 2) == x * x.
__global__ void limits_kernel(int B, int N, const int *n_points, const double *v, const double *a, double *max_k, double *max_kp, double mu, double max_kp_rate) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * N) return;
    const int b = (int)(t / N), i = (int)(t - (size_t)b * N);
    if (n_points && i >= n_points[b]) { max_k[t] = 0; max_kp[t] = 0; return; }
    const double mg = mu * 9.8, ref_v = v[t], ref_ax = a[t];
    const double ay_allowed = sqrt(mg * mg - ref_ax * ref_ax);
    max_k[t] = ref_v > 0.0001 ? ay_allowed / (ref_v * ref_v) : 1.7976931348623157e308;
    max_kp[t] = ref_v > 0.0001 ? max_kp_rate / ref_v : 1.7976931348623157e308;
}

// ReferencePathSmoother::graphSearchDp (reference_path_smoother.cpp:147-300) + calculateCostAt (:110-145).
// One block (one wave) per path, lane = lateral sample.  Layers are visited in order (each needs the costs of the previous one); inside a
// layer every lane scans the <= 64 nodes of the previous layer from LDS.  Parent indices of all layers stay in LDS for the walk back.
constexpr int kDpMaxLayers = 512, kDpMaxLat = 64;
// NW waves per instance: every wave carries all lateral samples (lane = node of the current layer) and evaluates the edges from every NW-th node of the
// previous layer; the partial minima meet in LDS (cost first, then the smaller previous index: the reference's first-minimum order).  NW = 8 for small
// batches (B <= 512): a single planning instance otherwise spends 47 layers x 34 x 34 edges, each with an atan2, on one wave — 2.35 ms against 0.92 ms.
template <int NW> __global__ __launch_bounds__(64 * NW) void dp_search_kernel(DevMap m, DevSpline in, DevSearch q) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t o = (size_t)b * q.L;
    if (in.n_knots && (in.n_knots[b] < 3 || in.n_knots[b] > in.K)) {  // no spline (an earlier stage of a pipeline failed): nothing to search
        for (int i = threadIdx.x; i < q.L; i += 64 * NW) { q.layer_s[o + i] = 0; q.lb[o + i] = 0; q.ub[o + i] = 0; }
        if (threadIdx.x == 0) { q.n_layers[b] = -1; q.l0[b] = 0; }
        return;
    }
    const Spl2 S = stage_spline(in, b, lds);
    const int LM = q.L < kDpMaxLayers ? q.L : kDpMaxLayers;  // layer tables are sized by the caller's capacity (LDS decides the occupancy)
    double *ls = lds + 15 * S.K;               // [LM] layer arc lengths
    double *nx = ls + LM;                      // node x, y, dir, cost of the previous / current layer (2 x 4 x 64)
    double *lat = nx + 2 * 4 * kDpMaxLat;      // [kDpMaxLat] lateral offsets
    unsigned long long *fmask = reinterpret_cast<unsigned long long *>(lat + kDpMaxLat);  // [LM] feasibility bits
    unsigned char *parent = reinterpret_cast<unsigned char *>(fmask + LM);             // [LM][64]
    unsigned char *chosen = parent + LM * kDpMaxLat;                                    // [LM]
    double *red = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(chosen + LM) + 15) & ~(uintptr_t)15);  // NW > 1: [NW][3][64] partial (cost, direction, index)
    __shared__ int s_L, s_rc;
    const double length = in.length[b], sx = q.start[3 * b], sy = q.start[3 * b + 1], sz = q.start[3 * b + 2];
    const double search_threshold = 1.45;
    // ---- findClosestPoint (tools.cpp:71-112): 0.5 m grid (lanes), then Newton (lane 0) ----
    double tmp_s0 = 0;
    if (length > 0) {
        double best = 1.7976931348623157e308;
        int bestk = 0x7fffffff;
        for (int k = lane; 0.5 * k <= length; k += 64) {  // tmp_s = 0.5 k exactly (the reference's running sum of 0.5 is exact)
            const SplAt pk = S.at(0.5 * k);
            const double ddx = S.x(pk) - sx, ddy = S.y(pk) - sy;
            const double d = sqrt(ddx * ddx + ddy * ddy);
            if (d < best) { best = d; bestk = k; }
        }
        for (int off = 32; off > 0; off >>= 1) {  // first minimum in scan order
            const double ob = __shfl_xor(best, off);
            const int ok = __shfl_xor(bestk, off);
            if (ob < best || (ob == best && ok < bestk)) { best = ob; bestk = ok; }
        }
        double cur_s = 0.5 * bestk, prev_s = cur_s;
        for (int i = 0; i < 20; ++i) {
            const SplAt pc = S.at(cur_s);
            const double px = S.x(pc), py = S.y(pc), dx = S.dx(1, pc), dy = S.dy(1, pc), ddx = S.dx(2, pc), ddy = S.dy(2, pc);
            const double j = (px - sx) * dx + (py - sy) * dy;
            const double hh = dx * dx + (px - sx) * ddx + dy * dy + (py - sy) * ddy;
            cur_s -= j / hh;
            if (fabs(cur_s - prev_s) < 1e-5) break;
            prev_s = cur_s;
        }
        tmp_s0 = cur_s < length ? cur_s : length;
    }
    // ---- layers (running sum, like the reference) ----
    if (threadIdx.x == 0) {
        const double search_ds = length > 6 ? q.long_spacing : 0.5;
        const int cap = LM;
        double t = tmp_s0;
        int L = 0, rc = 0;
        while (t < length) {
            if (L >= cap) { rc = -2; break; }
            ls[L++] = t;
            t += search_ds;
        }
        if (!rc) { if (L >= cap) rc = -2; else ls[L++] = length; }
        s_L = L; s_rc = rc;
    }
    __syncthreads();
    int L = s_L, rc = s_rc;
    double vl = 0;
    int start_idx = 0;
    if (!rc) {
        const SplAt pv = S.at(ls[0]);
        const double pxr = S.x(pv), pyr = S.y(pv), pz = S.heading(pv);
        const double dx = sx - pxr, dy = sy - pyr;
        vl = -dx * po_psin(pz) + dy * po_pcos(pz);  // global2Local(proj_point, start_state).y
        if (fabs(vl) > q.range) rc = -1;
        start_idx = (int)((q.range + vl) / q.lat_spacing);
    }
    if (rc) {
        for (int i = threadIdx.x; i < q.L; i += 64 * NW) { q.layer_s[o + i] = 0; q.lb[o + i] = 0; q.ub[o + i] = 0; }
        if (threadIdx.x == 0) { q.n_layers[b] = rc; q.l0[b] = vl; }
        return;
    }
    // lateral offsets by the reference's running sum; every lane keeps its own
    int nlat = 0;
    double my_l = 0;
    for (double cur_l = -q.range; cur_l <= q.range && nlat < kDpMaxLat; cur_l += q.lat_spacing) { if (nlat == lane) my_l = cur_l; ++nlat; }
    const bool act = lane < nlat;
    lat[lane] = my_l;
    __syncthreads();
    int max_layer = 0;
    double last_cost = 1.7976931348623157e308;  // this lane's cost in layer max_layer
    for (int i = 0; i < L; ++i) {
        const double cur_s = ls[i];
        const SplAt pa = S.at(cur_s);
        const double rx = S.x(pa), ry = S.y(pa), rh = S.heading(pa), rk = S.curvature(pa), rr = 1 / rk;
        double *cur = nx + (i & 1) * 4 * kDpMaxLat, *prv = nx + ((i & 1) ^ 1) * 4 * kDpMaxLat;
        const double x = rx + my_l * po_pcos(rh + M_PI_2), y = ry + my_l * po_psin(rh + M_PI_2);
        const double dis = map_inside(m, x, y) ? map_distance(m, x, y) : -1;
        bool feas = act && !((rk < 0 && my_l < rr) || (rk > 0 && my_l > rr) || dis < search_threshold);
        double cost = 1.7976931348623157e308, dir = 0;
        int par = 255;
        if (i == 0) { feas = act && lane == start_idx; if (feas) { dir = sz; cost = 0.0; } }
        const unsigned long long fm = __ballot(feas);
        if (lane == 0) fmask[i] = fm;
        if (i > 0 && feas) {  // calculateCostAt
            double self = 0;
            if (dis < 3.0) self += (3.0 - dis) / 3.0 * 0.5;
            self += fabs(my_l) / q.range * 1.0;
            const unsigned long long pm = fmask[i - 1];
            const double ps = ls[i - 1];
            double min_cost = 1.7976931348623157e308;
            // only previous nodes within reach can pass the test below: scan this lane's window of them, in ascending order like the reference's loop
            // over all of them (ties keep the first index).  The window is a superset by two samples either side; the exact test still decides.
            int wr = (int)((cur_s - ps) / q.lat_spacing) + 2;
            wr = wr < nlat ? wr : nlat;
            const int k0 = lane - wr > 0 ? lane - wr : 0, k1 = lane + wr < nlat - 1 ? lane + wr : nlat - 1;
            for (int k = k0 + wv; k <= k1; k += NW) {
                if (!((pm >> k) & 1ull)) continue;
                if (fabs(lat[k] - my_l) > (cur_s - ps)) continue;
                const double direction = po_patan2(y - prv[kDpMaxLat + k], x - prv[k]);
                const double edge = fabs(wrap_pi(direction - prv[2 * kDpMaxLat + k])) / M_PI_2 * 16.0 + fabs(wrap_pi(direction - rh)) / M_PI_2 * 0.5;
                const double total = self + edge + prv[3 * kDpMaxLat + k];
                if (total < min_cost) { min_cost = total; par = k; dir = direction; }
            }
            if (par != 255) cost = min_cost;
        }
        if constexpr (NW > 1) {  // the waves' partial minima -> the minimum over all previous nodes, first index on ties (what the sequential scan finds)
            red[(wv * 3 + 0) * 64 + lane] = cost; red[(wv * 3 + 1) * 64 + lane] = dir; red[(wv * 3 + 2) * 64 + lane] = (double)par;
            __syncthreads();
            if (i > 0) {
                cost = 1.7976931348623157e308; par = 255; dir = 0;
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
                    const double c2 = red[(w2 * 3 + 0) * 64 + lane];
                    const int p2 = (int)red[(w2 * 3 + 2) * 64 + lane];
                    if (p2 != 255 && (c2 < cost || (c2 == cost && p2 < par))) { cost = c2; par = p2; dir = red[(w2 * 3 + 1) * 64 + lane]; }
                }
            }
        }
        const bool any = __any(par != 255);
        if (i != 0 && !any) break;
        if (act) { cur[lane] = x; cur[kDpMaxLat + lane] = y; cur[2 * kDpMaxLat + lane] = dir; cur[3 * kDpMaxLat + lane] = cost; parent[i * kDpMaxLat + lane] = (unsigned char)par; }
        max_layer = i;
        last_cost = act ? cost : 1.7976931348623157e308;
        __syncthreads();
    }
    // ---- cheapest node of the last reachable layer (first minimum), walk back ----
    double bc = last_cost;
    int bj = (act && last_cost < 1.7976931348623157e308) ? lane : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(bc, off);
        const int oj = __shfl_xor(bj, off);
        if (oc < bc || (oc == bc && oj < bj)) { bc = oc; bj = oj; }
    }
    int count = 0;
    if (bj != 0x7fffffff) {
        if (threadIdx.x == 0) {
            int j = bj;
            for (int i = max_layer; i >= 0; --i) { chosen[i] = (unsigned char)j; j = parent[i * kDpMaxLat + j]; }
        }
        count = max_layer + 1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < q.L; i += 64 * NW) {
        double lo = 0, hi = 0, sv = 0;
        if (i < count) {
            sv = ls[i];
            if (i == 0) { lo = -10; hi = 10; }
            else {
                const int j = chosen[i];
                const unsigned long long fm = fmask[i];
                int ja = j, jb = j;  // rough bounds: the run of consecutive feasible samples around j
                if ((fm >> j) & 1ull) {
                    while (ja > 0 && ((fm >> (ja - 1)) & 1ull)) --ja;
                    while (jb < nlat - 1 && ((fm >> (jb + 1)) & 1ull)) ++jb;
                }
                const double check_s = 0.2, check_limit = 6.0;
                hi = check_s + lat[jb]; lo = -check_s + lat[ja];
                const SplAt ps = S.at(sv);
                const double rx = S.x(ps), ry = S.y(ps), rh = S.heading(ps);
                while (hi < check_limit) {
                    const double px2 = rx + hi * po_pcos(rh + M_PI_2), py2 = ry + hi * po_psin(rh + M_PI_2);
                    if (map_inside(m, px2, py2) && map_distance(m, px2, py2) > search_threshold) hi += check_s;
                    else { hi -= check_s; break; }
                }
                while (lo > -check_limit) {
                    const double px2 = rx + lo * po_pcos(rh + M_PI_2), py2 = ry + lo * po_psin(rh + M_PI_2);
                    if (map_inside(m, px2, py2) && map_distance(m, px2, py2) > search_threshold) lo -= check_s;
                    else { lo += check_s; break; }
                }
            }
        }
        q.layer_s[o + i] = sv; q.lb[o + i] = lo; q.ub[o + i] = hi;
    }
    if (threadIdx.x == 0) { q.n_layers[b] = count; q.l0[b] = vl; }
}


// =====================================================================================================================
// The remaining glue stages of PathOptimizer::solve (path_optimizer.cpp:40-178) and the plumbing of po_plan_batch
// =====================================================================================================================
// tinyspline's clamped B-spline (library absent from /root/reference: restated, parity unpinned): knots 0 (i <= deg), (i - deg)/(n - deg), 1 (i >= n)
__device__ __forceinline__ double bs_knot(int i, int n, int deg) {
    if (i <= deg) return 0.0;
    if (i >= n) return 1.0;
    const double fac = (1.0 - 0.0) / (double)(n + deg + 1 - 2 * deg - 1);
    return fac * (double)(i - deg) + 0.0;
}
__device__ void bspline_eval(int n, int deg, const double *cx, const double *cy, double u, double &ox, double &oy) {
    if (u >= 1.0) { ox = cx[n - 1]; oy = cy[n - 1]; return; }
    if (u <= 0.0) { ox = cx[0]; oy = cy[0]; return; }
    int k = deg;
    while (k + 1 < n && bs_knot(k + 1, n, deg) <= u) ++k;
    double dx[8], dy[8];
    for (int j = 0; j <= deg; ++j) { dx[j] = cx[k - deg + j]; dy[j] = cy[k - deg + j]; }
    for (int r = 1; r <= deg; ++r)
        for (int j = deg; j >= r; --j) {
            const int i = k - deg + j;
            const double ki = bs_knot(i, n, deg), kj = bs_knot(i + deg - r + 1, n, deg);
            const double a = (u - ki) / (kj - ki);
            dx[j] = (1.0 - a) * dx[j - 1] + a * dx[j];
            dy[j] = (1.0 - a) * dy[j - 1] + a * dy[j];
        }
    ox = dx[deg]; oy = dy[deg];
}
// ReferencePathSmoother::bSpline (reference_path_smoother.cpp:495-532): x_list_, y_list_, s_list_ from the input points.
__global__ __launch_bounds__(64) void bspline_kernel(int B, int W, const int *n_way, const double *wx, const double *wy, int M, double *x, double *y, double *s,
                                                     int *n_samples) {
    extern __shared__ double lds[];  // control points x, y [W]
    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t o = (size_t)b * M;
    __shared__ int s_m, s_deg;
    const int n = n_way ? n_way[b] : W;
    double *cx = lds, *cy = lds + W;
    for (int i = lane; i < W; i += 64) { cx[i] = i < n ? wx[(size_t)b * W + i] : 0.0; cy[i] = i < n ? wy[(size_t)b * W + i] : 0.0; }
    __syncthreads();
    if (lane == 0) {
        int m = -1, degree = 3;
        if (n >= 4 && n <= W) {  // "Few reference points."
            double length = 0;
            for (int i = 0; i + 1 < n; ++i) { const double ddx = cx[i] - cx[i + 1], ddy = cy[i] - cy[i + 1]; length += sqrt(ddx * ddx + ddy * ddy); }
            const double average_length = length / (n - 1);
            degree = average_length > 10 ? 3 : (average_length > 5 ? 4 : 5);
            if (n > degree) {  // tinyspline throws otherwise
                const double delta_t = 1.0 / length;
                double tmp_t = 0;
                m = 0;
                while (tmp_t < 1) {
                    if (m >= M - 1) { m = -2; break; }
                    s[o + m] = tmp_t;  // parameter values parked in the output row until the evaluation below
                    ++m;
                    tmp_t += delta_t;
                }
                if (m >= 0) { s[o + m] = 1.0; ++m; }
            }
        }
        s_m = m; s_deg = degree;
        n_samples[b] = m;
    }
    __threadfence_block();
    __syncthreads();
    const int m = s_m < 0 ? 0 : s_m;
    for (int i = lane; i < M; i += 64) {
        double ox = 0, oy = 0;
        if (i < m) bspline_eval(n, s_deg, cx, cy, s[o + i], ox, oy);
        x[o + i] = ox; y[o + i] = oy;
    }
    __threadfence_block();
    __syncthreads();
    if (lane == 0) {
        double acc = 0;
        for (int i = 0; i < M; ++i) {
            if (i > 0 && i < m) { const double ddx = x[o + i] - x[o + i - 1], ddy = y[o + i] - y[o + i - 1]; acc += sqrt(ddx * ddx + ddy * ddy); }
            s[o + i] = i < m ? acc : 0.0;
        }
    }
}

// ReferencePathSmoother::segmentRawReference (reference_path_smoother.cpp:50-91): 1 m stations on the spline through the dense lists.
__global__ __launch_bounds__(64) void segment_raw_kernel(DevSpline in, int P, double *x, double *y, double *s, double *angle, double *k, int *n_points) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t o = (size_t)b * P;
    const int Kb = in.n_knots ? in.n_knots[b] : in.K;
    int n = -1;
    if (Kb >= 3 && Kb <= in.K) {
        const double max_s = in.knot_s[(size_t)b * in.K + Kb - 1];
        n = 1;
        double last = 0;
        while (last < max_s) {  // s_list: 0, 1, 2, ... until the last station is >= max_s
            if (n >= P) { n = -2; break; }
            last += 1.0;
            ++n;
        }
    }
    if (n <= 0) {
        for (int i = lane; i < P; i += 64) { x[o + i] = 0; y[o + i] = 0; s[o + i] = 0; angle[o + i] = 0; k[o + i] = 0; }
        if (lane == 0) n_points[b] = n;
        return;
    }
    const Spl2 S = stage_spline(in, b, lds);
    for (int i = lane; i < P; i += 64) {
        if (i < n) {
            const double at = (double)i;
            const SplAt pa = S.at(at);
            const double dx = S.dx(1, pa), dy = S.dy(1, pa), ddx = S.dx(2, pa), ddy = S.dy(2, pa);
            angle[o + i] = po_patan2(dy, dx);
            k[o + i] = (dx * ddy - dy * ddx) / po_ppow15(dx * dx + dy * dy);
            x[o + i] = S.x(pa); y[o + i] = S.y(pa); s[o + i] = at;
        } else { x[o + i] = 0; y[o + i] = 0; s[o + i] = 0; angle[o + i] = 0; k[o + i] = 0; }
    }
    if (lane == 0) n_points[b] = n;
}

// The tail of ReferencePathSmoother::postSmooth (reference_path_smoother.cpp:568-590): QP offsets re-projected onto the spline.
__global__ __launch_bounds__(64) void post_project_kernel(DevSpline in, int L, const int *n_layers, const double *layer_s, const double *off, double *x, double *y,
                                                          double *s, double *length_out) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t o = (size_t)b * L;
    const int Kb = in.n_knots ? in.n_knots[b] : in.K;
    int n = n_layers ? n_layers[b] : L;
    if (Kb < 3 || Kb > in.K || n < 1 || n > L) n = 0;
    if (n == 0) {
        for (int i = lane; i < L; i += 64) { x[o + i] = 0; y[o + i] = 0; s[o + i] = 0; }
        if (lane == 0 && length_out) length_out[b] = 0;
        return;
    }
    const Spl2 S = stage_spline(in, b, lds);
    for (int i = lane; i < L; i += 64) {
        double ox = 0, oy = 0;
        if (i < n) {
            const double ref_s = layer_s[o + i];
            const SplAt pr = S.at(ref_s);
            const double ref_dir = S.heading(pr);
            ox = S.x(pr) + off[o + i] * po_pcos(ref_dir + M_PI_2);
            oy = S.y(pr) + off[o + i] * po_psin(ref_dir + M_PI_2);
        }
        x[o + i] = ox; y[o + i] = oy;
    }
    __threadfence_block();
    __syncthreads();
    if (lane == 0) {
        double acc = 0;
        for (int i = 0; i < L; ++i) {
            if (i > 0 && i < n) { const double ddx = x[o + i] - x[o + i - 1], ddy = y[o + i] - y[o + i - 1]; acc += sqrt(ddx * ddx + ddy * ddy); }
            s[o + i] = i < n ? acc : 0.0;
        }
        if (length_out) length_out[b] = acc;
    }
}

// PathOptimizer::segmentSmoothedPath before the re-sampling (path_optimizer.cpp:119-169): initial offset / heading error, goal trim.
__global__ __launch_bounds__(64) void segment_init_kernel(DevSpline in, const double *start, int start_stride, const double *goal, int goal_stride, int exact, double *init,
                                                          int *ok) {
    extern __shared__ double lds[];
    const int b = blockIdx.x;
    const int Kb = in.n_knots ? in.n_knots[b] : in.K;
    const double length = in.length[b];
    if (Kb < 3 || Kb > in.K || length == 0) {  // "Smoothed path is empty!"
        if (threadIdx.x == 0) { init[3 * b] = 0; init[3 * b + 1] = 0; init[3 * b + 2] = length; ok[b] = 0; }
        return;
    }
    const Spl2 S = stage_spline(in, b, lds);
    if (threadIdx.x != 0) return;
    const double sx = start[(size_t)b * start_stride], sy = start[(size_t)b * start_stride + 1], sz = start[(size_t)b * start_stride + 2];
    const double gx = goal[(size_t)b * goal_stride], gy = goal[(size_t)b * goal_stride + 1];
    const double fx = S.x(0), fy = S.y(0), fz = S.heading(0);
    const double dx = fx - sx, dy = fy - sy;
    const double local_y = -dx * po_psin(sz) + dy * po_pcos(sz);
    const double min_distance = sqrt((sx - fx) * (sx - fx) + (sy - fy) * (sy - fy));
    const double e0 = local_y < 0 ? min_distance : -min_distance, e1 = wrap_pi(sz - fz);
    int good = !(fabs(e1) > 75 * M_PI / 180);
    double len = length;
    if (good) {
        const double ex = gx - S.x(length), ey = gy - S.y(length);
        const double end_distance = sqrt(ex * ex + ey * ey);
        if (!(fabs(end_distance - 0) < 1e-6)) {
            const double dsr = exact ? 0.1 : 0.5;
            double tmp_s = length - dsr, min_dis = end_distance, min_s = length;
            while (tmp_s > 0) {
                const double px = S.x(tmp_s), py = S.y(tmp_s);
                const double d = sqrt((px - gx) * (px - gx) + (py - gy) * (py - gy));
                if (d < min_dis) { min_dis = d; min_s = tmp_s; }
                if (d > 8 && min_dis < 8) break;
                tmp_s -= dsr;
            }
            len = min_s;
        }
    }
    init[3 * b] = e0; init[3 * b + 1] = e1; init[3 * b + 2] = len; ok[b] = good;
}

// optimizePath's densifying output branch (path_optimizer.cpp:201-226): splines x(s), y(s) through the solved states (fitted by two lanes
// in LDS), samples every `spacing` on all lanes, first colliding sample by atomicMin.
__global__ __launch_bounds__(128) void densify_kernel(DevMap m, DevCar c, int B, int N, const int *n_points, const double *states, const po_info *info, double spacing,
                                                      int M, double *out, int *n_out, int *ok) {
    extern __shared__ double lds[];  // knots s, x, y [N] + 2 x (a, b, c, 3 scratch) [N]
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int first;
    int n = n_points ? n_points[b] : N;
    n = n < 0 ? 0 : (n > N ? N : n);
    double *o = out + (size_t)b * M * 5;
    const bool solved = info[b].status == PO_STATUS_SOLVED && n >= 3;
    if (!solved) {
        for (int i = tid; i < M * 5; i += blockDim.x) o[i] = 0;
        if (tid == 0) { n_out[b] = 0; ok[b] = 0; }
        return;
    }
    double *ks = lds, *kx = lds + N, *ky = lds + 2 * N, *cx = lds + 3 * N, *cy = lds + 9 * N;
    const double *st = states + (size_t)b * N * 5;
    for (int i = tid; i < n; i += blockDim.x) { kx[i] = st[5 * i]; ky[i] = st[5 * i + 1]; ks[i] = st[5 * i + 4]; }
    __syncthreads();
    if (tid < 2) spline_fit(n, ks, tid ? ky : kx, (tid ? cy : cx), (tid ? cy : cx) + N, (tid ? cy : cx) + 2 * N, (tid ? cy : cx) + 3 * N);
    __syncthreads();
    const Spl2 S{n, ks, kx, ky, cx, cx + N, cx + 2 * N, cy, cy + N, cy + 2 * N};
    const double s_end = ks[n - 1];
    int total = 0;  // samples i with i * spacing <= s_end (same float test as the reference's loop)
    while ((double)total * spacing <= s_end) ++total;
    const int lim = total < M ? total : M;
    if (tid == 0) first = total;
    __syncthreads();
    for (int i = tid; i < M; i += blockDim.x) {
        double v[5] = {0, 0, 0, 0, 0};
        if (i < lim) {
            const double at = (double)i * spacing;
            const SplAt pa = S.at(at);
            v[0] = S.x(pa); v[1] = S.y(pa); v[2] = S.heading(pa); v[3] = S.curvature(pa); v[4] = at;
            if (c.enable && !collision_free(m, c, v[0], v[1], v[2])) atomicMin(&first, i);
        }
        for (int j = 0; j < 5; ++j) o[5 * i + j] = v[j];
    }
    __syncthreads();
    const int f = first;
    if (f < total || total > M) {  // truncated by a collision (or by the capacity)
        const int keepn = f < lim ? f : lim;
        for (int i = tid; i < M; i += blockDim.x)
            if (i >= keepn) for (int j = 0; j < 5; ++j) o[5 * i + j] = 0;
    }
    if (tid == 0) {
        if (f >= total && total <= M) { n_out[b] = total; ok[b] = 1; }
        else if (f >= lim && total > M) { n_out[b] = -2; ok[b] = 0; }  // M too small before any collision
        else { n_out[b] = f; ok[b] = (f > 0 && (double)(f - 1) * spacing >= 20.0) ? 1 : 0; }
    }
}

// ---- plumbing of po_plan_batch: per-instance gates between stages (a failed instance turns every later stage into a no-op) ----
__global__ void plan_gate_kernel(PlanGate g) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.B) return;
    int st = g.stage[b];
    switch (g.mode) {
        case 0:  // before everything
            st = 0;
            for (int i = 0; i < 3; ++i) g.start3[3 * b + i] = g.start[4 * b + i];
            break;
        case 1:  // after TENSION2: few points / bSpline -> 1, QP not solved -> 2; length = result_s.back() + 3 (TensionSmoother::smooth)
            if (!st && (g.cnt2[b] == -2 || g.cnt[b] == -2)) st = 9;  // intermediate capacity (max_length too small)
            if (!st && (g.cnt2[b] < 3 || g.cnt[b] < 3)) st = 1;
            if (!st && g.info[b].status != PO_STATUS_SOLVED) st = 2;
            g.length[b] = st ? 0.0 : g.s[(size_t)b * g.stride + g.cnt[b] - 1] + 3;
            if (st) g.cnt[b] = 0;
            break;
        case 2:  // after the search: false -> 3, fewer than 4 layers ("Ref is short") -> 4
            if (!st && g.cnt[b] == -2) st = 9;  // layer capacity
            if (!st && g.cnt[b] < 0) st = 3;
            if (!st && g.cnt[b] < 4) st = 4;
            if (st) g.cnt[b] = 0;
            break;
        case 3:  // after the post-smoothing QP
            if (!st && g.info[b].status != PO_STATUS_SOLVED) st = 4;
            if (st) g.cnt[b] = 0;
            break;
        case 4:  // after segmentSmoothedPath's first half
            if (!st && !g.ok[b]) st = 5;
            g.length[b] = st ? 0.0 : g.init[3 * b + 2];
            break;
        case 5: {  // after re-sampling + bounds: blocked reference -> 6; QP inputs
            if (!st && g.cnt[b] < 2) st = 6;
            if (st) g.cnt[b] = 0;
            int keep = 0;
            if (!st) {  // OsqpSolver::OsqpSolver (solver.cpp:19-27) + SolverKpAsInput (solver_kp_as_input.cpp:17)
                double interval = 0;
                for (int i = 1; i < g.cnt[b] && i < 10; ++i) interval = fmax(interval, g.ref_s[(size_t)b * g.ref_stride + i] - g.ref_s[(size_t)b * g.ref_stride + i - 1]);
                const double q = 1.2 / interval;
                keep = (q >= 2147483647.0 || q != q) ? 1 : (int)q;
                keep = keep > 1 ? keep : 1;
            }
            g.keep[b] = keep;
            g.x0[3 * b] = g.init[3 * b]; g.x0[3 * b + 1] = g.init[3 * b + 1]; g.x0[3 * b + 2] = g.start[4 * b + 3];
            g.goal_z[b] = g.goal[3 * b + 2];
            break;
        }
        case 6:  // after the path QP
            if (!st && g.info[b].status != PO_STATUS_SOLVED) st = 7;
            if (st) g.cnt[b] = 0;
            break;
        case 7:  // after the collision check: cnt = n_kept, ok = optimizePath's value
            if (!st && g.cnt[b] == -2) st = 9;  // densifying branch (ok is 0 already): output capacity too small (po_densify_batch's n_out = -2) is a capacity
                                                                 // problem (stage 9, no states), not a failed collision check
            if (!st && !g.ok[b]) st = 8;
            if (st && st != 8) g.cnt[b] = 0;
            break;
    }
    g.stage[b] = st;
}
// gather the rows of one keep-group into a compact batch (stride Ng) and scatter its results back
__global__ void plan_gather_kernel(PlanRows r) {
    const int g = blockIdx.x, b = r.idx[g];
    for (int i = threadIdx.x; i < r.Ng; i += blockDim.x) {
        const size_t so = (size_t)b * r.N + i, d = (size_t)g * r.Ng + i;
        r.g_x[d] = r.ref_x[so]; r.g_y[d] = r.ref_y[so]; r.g_z[d] = r.ref_z[so]; r.g_k[d] = r.ref_k[so]; r.g_s[d] = r.ref_s[so];
        for (int j = 0; j < 8; ++j) r.g_bounds[d * 8 + j] = r.bounds[so * 8 + j];
    }
    if (threadIdx.x == 0) {
        for (int j = 0; j < 3; ++j) r.g_x0[3 * g + j] = r.x0[3 * b + j];
        r.g_goal[g] = r.goal_z[b]; r.g_n[g] = r.n_valid[b];
    }
}
__global__ void plan_scatter_kernel(PlanRows r) {
    const int g = blockIdx.x, b = r.idx[g];
    for (int i = threadIdx.x; i < r.N * 5; i += blockDim.x) {
        const int row = i / 5;
        r.states[(size_t)b * r.N * 5 + i] = row < r.Ng ? r.g_states[(size_t)g * r.Ng * 5 + i] : 0.0;
    }
    if (threadIdx.x == 0) r.info[b] = r.g_info[g];
}
__global__ void plan_clear_kernel(int B, int N, const int *stage, double *states, po_info *info) {  // rows of instances that never reached the QP
    const int b = blockIdx.x;
    if (!stage[b]) return;
    for (int i = threadIdx.x; i < N * 5; i += blockDim.x) states[(size_t)b * N * 5 + i] = 0.0;
    if (threadIdx.x == 0) { po_info z{}; z.status = PO_STATUS_UNSOLVED; info[b] = z; }
}

}  // namespace po

extern "C" hipError_t po_launch_postcheck(const po::DevMap *m, const po::DevCar *c, int B, int N, const int *n_points, const double *states,
                                          const po_info *info, int *n_valid, int *ok, hipStream_t st) {
    hipLaunchKernelGGL(po::postcheck_kernel, dim3(B), dim3(128), 0, st, *m, *c, B, N, n_points, states, info, n_valid, ok);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_map_sample(const po::DevMap *m, int n, const double *xy, double *dist, int *inside, hipStream_t st) {
    hipLaunchKernelGGL(po::map_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *m, n, xy, dist, inside);
    return hipGetLastError();
}

extern "C" hipError_t po_launch_bounds(const po::DevMap *m, const po::DevBounds *in, double *bounds, int *n_valid, hipStream_t st) {
    hipLaunchKernelGGL(po::spline_fit_kernel, dim3(in->B), dim3(64), 0, st, *in);
    hipLaunchKernelGGL(po::bounds_kernel, dim3(in->B), dim3(256), 0, st, *m, *in, bounds, n_valid);
    return hipGetLastError();
}

extern "C" size_t po_spline_lds_bytes(int K) { return sizeof(double) * 15 * (size_t)K; }
// LDS of the one-wave variant (what the capacity check uses: the launcher falls back to it when the 8-wave variant's reduction scratch does not fit)
extern "C" size_t po_dp_lds_bytes(int K, int L) {
    const size_t LM = (size_t)(L < po::kDpMaxLayers ? L : po::kDpMaxLayers);
    return sizeof(double) * (15 * (size_t)K + LM + 2 * 4 * po::kDpMaxLat + po::kDpMaxLat) + 8 * LM + LM * po::kDpMaxLat + LM + 32;
}
static const size_t kDpEightWaveScratch = sizeof(double) * 8 * 3 * 64;
extern "C" hipError_t po_launch_resample(const po::DevSpline *in, const po::DevResample *r, hipStream_t st) {
    hipLaunchKernelGGL(po::resample_kernel, dim3(in->B), dim3(64), po_spline_lds_bytes(in->K), st, *in, *r);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_limits(int B, int N, const int *n_points, const double *v, const double *a, double *max_k, double *max_kp, double mu,
                                       double rate, hipStream_t st) {
    const size_t tot = (size_t)B * N;
    hipLaunchKernelGGL(po::limits_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B, N, n_points, v, a, max_k, max_kp, mu, rate);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_dp_search(const po::DevMap *m, const po::DevSpline *in, const po::DevSearch *q, int one_wave, hipStream_t st) {
    const size_t lds1 = po_dp_lds_bytes(in->K, q->L), lds4 = lds1 + kDpEightWaveScratch;
    // few instances (a planner's own call: B = 1): eight waves per instance share the edge evaluations of a layer; a full batch keeps one wave per instance
    // (one_wave: the caller's A/B switch, po_debug_set "dp_one_wave")
    if (in->B <= 512 && !one_wave && lds4 <= 160 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&po::dp_search_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(po::dp_search_kernel<8>, dim3(in->B), dim3(512), lds4, st, *m, *in, *q);
        return hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&po::dp_search_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(po::dp_search_kernel<1>, dim3(in->B), dim3(64), lds1, st, *m, *in, *q);
    return hipGetLastError();
}

extern "C" hipError_t po_launch_bspline(int B, int W, const int *n_way, const double *wx, const double *wy, int M, double *x, double *y, double *s, int *n_samples, hipStream_t st) {
    hipLaunchKernelGGL(po::bspline_kernel, dim3(B), dim3(64), sizeof(double) * 2 * (size_t)W, st, B, W, n_way, wx, wy, M, x, y, s, n_samples);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_segment_raw(const po::DevSpline *in, int P, double *x, double *y, double *s, double *angle, double *k, int *n_points, hipStream_t st) {
    hipLaunchKernelGGL(po::segment_raw_kernel, dim3(in->B), dim3(64), po_spline_lds_bytes(in->K), st, *in, P, x, y, s, angle, k, n_points);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_post_project(const po::DevSpline *in, int L, const int *n_layers, const double *layer_s, const double *off, double *x, double *y, double *s,
                                             double *length_out, hipStream_t st) {
    hipLaunchKernelGGL(po::post_project_kernel, dim3(in->B), dim3(64), po_spline_lds_bytes(in->K), st, *in, L, n_layers, layer_s, off, x, y, s, length_out);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_segment_init(const po::DevSpline *in, const double *start, int start_stride, const double *goal, int goal_stride, int exact, double *init,
                                             int *ok, hipStream_t st) {
    hipLaunchKernelGGL(po::segment_init_kernel, dim3(in->B), dim3(64), po_spline_lds_bytes(in->K), st, *in, start, start_stride, goal, goal_stride, exact, init, ok);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_plan_gate(const po::PlanGate *g, hipStream_t st) {
    hipLaunchKernelGGL(po::plan_gate_kernel, dim3((g->B + 127) / 128), dim3(128), 0, st, *g);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_plan_gather(const po::PlanRows *r, hipStream_t st) {
    hipLaunchKernelGGL(po::plan_gather_kernel, dim3(r->G), dim3(128), 0, st, *r);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_plan_scatter(const po::PlanRows *r, hipStream_t st) {
    hipLaunchKernelGGL(po::plan_scatter_kernel, dim3(r->G), dim3(128), 0, st, *r);
    return hipGetLastError();
}
extern "C" hipError_t po_launch_plan_clear(int B, int N, const int *stage, double *states, po_info *info, hipStream_t st) {
    hipLaunchKernelGGL(po::plan_clear_kernel, dim3(B), dim3(128), 0, st, B, N, stage, states, info);
    return hipGetLastError();
}

extern "C" hipError_t po_launch_densify(const po::DevMap *m, const po::DevCar *c, int B, int N, const int *n_points, const double *states, const po_info *info,
                                        double spacing, int M, double *out, int *n_out, int *ok, hipStream_t st) {
    const size_t lds = sizeof(double) * 15 * (size_t)N;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&po::densify_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(po::densify_kernel, dim3(B), dim3(128), lds, st, *m, *c, B, N, n_points, states, info, spacing, M, out, n_out, ok);
    return hipGetLastError();
}
