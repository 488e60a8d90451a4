// po_post.hip — kernels around the QP hot path: post-solve collision check, map sampling, corridor-bounds producer.
// Compiled with -ffp-contract=off (see Makefile): these kernels mirror plain IEEE double arithmetic of the reference's C++,
// expression by expression, so that thresholds (clearance < radius) fall on the same side as on the CPU.
#include <hip/hip_runtime.h>

#include "../../include/po_hip.h"
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


// ---- tk::spline (src/tools/spline.cpp:154-271), natural boundary conditions ------------------------------------------------
// fit: one thread per spline; a, b, c [K] out, w = 3K doubles of scratch (diagonal, saved 1/diagonal, rhs)
__device__ void spline_fit(int K, const double *x, const double *y, double *a, double *b, double *c, double *w) {
    double *lo = a, *up = c, *di = w, *sd = w + K, *rh = w + 2 * K;
    for (int i = 1; i < K - 1; ++i) {
        lo[i] = 1.0 / 3.0 * (x[i] - x[i - 1]);
        di[i] = 2.0 / 3.0 * (x[i + 1] - x[i - 1]);
        up[i] = 1.0 / 3.0 * (x[i + 1] - x[i]);
        rh[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    }
    di[0] = 2.0; up[0] = 0.0; rh[0] = 0.0; lo[0] = 0.0;
    di[K - 1] = 2.0; lo[K - 1] = 0.0; rh[K - 1] = 0.0; up[K - 1] = 0.0;
    for (int i = 0; i < K; ++i) {  // band_matrix::lu_decompose preconditioning (:70-84)
        sd[i] = 1.0 / di[i];
        if (i > 0) lo[i] *= sd[i];
        if (i < K - 1) up[i] *= sd[i];
        di[i] = 1.0;
    }
    for (int k = 0; k + 1 < K; ++k) {  // Gauss (:86-100)
        const double xx = -lo[k + 1] / di[k];
        lo[k + 1] = -xx;
        di[k + 1] = di[k + 1] + xx * up[k];
    }
    for (int i = 0; i < K; ++i) rh[i] = (rh[i] * sd[i]) - (i > 0 ? lo[i] * rh[i - 1] : 0.0);            // l_solve
    for (int i = K - 1; i >= 0; --i) b[i] = (rh[i] - (i < K - 1 ? up[i] * b[i + 1] : 0.0)) / di[i];   // r_solve
    for (int i = 0; i < K - 1; ++i) {
        a[i] = 1.0 / 3.0 * (b[i + 1] - b[i]) / (x[i + 1] - x[i]);
        c[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - 1.0 / 3.0 * (2.0 * b[i] + b[i + 1]) * (x[i + 1] - x[i]);
    }
    const double h = x[K - 1] - x[K - 2];
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
// getClearanceWithDirectionStrict (reference_path_impl.cpp:283-472, simple boundary decision)
__device__ void clearance_strict(const DevMap &m, double radius, double sx, double sy, double sz, double &left_bound, double &right_bound) {
    left_bound = 0; right_bound = 0;
    const double delta_s = 0.5;
    const double la = wrap_pi(sz + M_PI_2), ra = wrap_pi(sz - M_PI_2);
    const int n = (int)(5.0 / delta_s);
    const double cl = cos(la), sl = sin(la), cr = cos(ra), sr = sin(ra);
    if (map_distance(m, sx, sy) > radius) {
        double right_s = 0, left_s = 0;
        for (int j = 0; j != n; ++j) { right_s += delta_s; if (map_distance(m, sx + right_s * cr, sy + right_s * sr) < radius) break; }
        for (int j = 0; j != n; ++j) { left_s += delta_s; if (map_distance(m, sx + left_s * cl, sy + left_s * sl) < radius) break; }
        right_bound = -(right_s - delta_s);
        left_bound = left_s - delta_s;
    } else {
        double right_s = 0, left_s = 0;
        for (int j = 0; j != n; ++j) { right_s += delta_s; if (map_distance(m, sx + right_s * cr, sy + right_s * sr) > radius) break; }
        for (int j = 0; j != n; ++j) { left_s += delta_s; if (map_distance(m, sx + left_s * cl, sy + left_s * sl) > radius) break; }
        if (left_s < right_s) {
            right_bound = left_s;
            for (int j = 0; j != n; ++j) { left_s += delta_s; if (map_distance(m, sx + left_s * cl, sy + left_s * sl) < radius) break; }
            left_bound = left_s - delta_s;
        } else {
            left_bound = -right_s;
            for (int j = 0; j != n; ++j) { right_s += delta_s; if (map_distance(m, sx + right_s * cr, sy + right_s * sr) < radius) break; }
            right_bound = -(right_s - delta_s);
        }
    }
    const double smaller_ds = 0.1;
    const int nf = (int)(delta_s / smaller_ds);
    for (int i = 1; i != nf; ++i) {
        left_bound += smaller_ds;
        if (map_distance(m, sx + left_bound * cl, sy + left_bound * sl) < radius) { left_bound -= smaller_ds; break; }
    }
    for (int i = 1; i != nf; ++i) {
        right_bound -= smaller_ds;
        if (map_distance(m, sx + right_bound * cr, sy + right_bound * sr) < radius) { right_bound += smaller_ds; break; }
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
            const double cz = cos(z), sz = sin(z), len = in.d[j];
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
