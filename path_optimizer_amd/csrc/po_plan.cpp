// po_plan.cpp — PathOptimizer::solve for a batch of planning instances: the orchestration of the device stages
// (/root/reference/src/path_optimizer/path_optimizer.cpp:40-85,119-230 and ReferencePathSmoother::solve,
// src/reference_path_smoother/reference_path_smoother.cpp:34-48), plus the device-pointer entries of the small glue stages.
// Everything here goes through the public C ABI of the other stages; intermediates live in one arena of the handle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/po_hip.h"
#include "po_map.hpp"

// po_capi.cpp
extern "C" void *po_internal_arena(po_handle h, size_t bytes);
extern "C" void *po_internal_plan_coef(po_handle h, size_t bytes);
extern "C" hipStream_t po_internal_stream(po_handle h);
extern "C" int po_internal_device(po_handle h);
extern "C" const po_params *po_internal_params(po_handle h);
extern "C" int po_internal_has_map(po_handle h);
extern "C" int po_internal_hip_fail(hipError_t e, const char *what);
extern "C" void *po_internal_plan_host(po_handle h, size_t bytes);
#include <mutex>
extern "C" std::mutex *po_internal_plan_mutex(po_handle h);
// po_post.hip
extern "C" size_t po_spline_lds_bytes(int K);
extern "C" hipError_t po_launch_bspline(int B, int W, const int *n_way, const double *wx, const double *wy, int M, double *x, double *y, double *s, int *n_samples, hipStream_t st);
extern "C" hipError_t po_launch_segment_raw(const po::DevSpline *in, int P, double *x, double *y, double *s, double *angle, double *k, int *n_points, hipStream_t st);
extern "C" hipError_t po_launch_post_project(const po::DevSpline *in, int L, const int *n_layers, const double *layer_s, const double *off, double *x, double *y, double *s,
                                             double *length_out, hipStream_t st);
extern "C" hipError_t po_launch_segment_init(const po::DevSpline *in, const double *start, int start_stride, const double *goal, int goal_stride, int exact, double *init,
                                             int *ok, hipStream_t st);
extern "C" hipError_t po_launch_plan_gate(const po::PlanGate *g, hipStream_t st);
extern "C" hipError_t po_launch_plan_gather(const po::PlanRows *r, hipStream_t st);
extern "C" hipError_t po_launch_plan_scatter(const po::PlanRows *r, hipStream_t st);
extern "C" hipError_t po_launch_plan_clear(int B, int N, const int *stage, double *states, po_info *info, hipStream_t st);

#define HIP_TRY(x)                                               \
    do {                                                         \
        if (po_internal_hip_fail((x), #x)) return PO_ERR_HIP;    \
    } while (0)
#define PO_TRY(x)                    \
    do {                             \
        const int rc_ = (x);         \
        if (rc_ != PO_OK) return rc_; \
    } while (0)

namespace {
int dev_spline(po_handle h, const po_spline_in *in, po::DevSpline *D) {
    if (!in || in->B < 0 || in->K < 3 || (in->B > 0 && (!in->knot_s || !in->knot_x || !in->knot_y))) return PO_ERR_INVALID;
    if (po_spline_lds_bytes(in->K) > 64 * 1024) return PO_ERR_UNSUPPORTED;
    (void)h;  // the spline coefficients are fitted in LDS by each consumer kernel
    D->B = in->B; D->K = in->K; D->knot_s = in->knot_s; D->knot_x = in->knot_x; D->knot_y = in->knot_y; D->n_knots = in->n_knots; D->length = in->length;
    D->coef = nullptr;
    return PO_OK;
}
struct Arena {  // bump allocator over the handle's plan arena
    char *p = nullptr;
    size_t off = 0, cap = 0;
    template <typename T> T *take(size_t n) {
        off = (off + 15) & ~(size_t)15;
        T *r = reinterpret_cast<T *>(p + off);
        off += sizeof(T) * n;
        return r;
    }
};
}  // namespace

extern "C" {

int po_bspline_batch_device(po_handle h, int B, int W, const int *n_way, const double *way_x, const double *way_y, int M, double *x, double *y, double *s, int *n_samples) {
    if (!h || B < 0 || W < 1 || M < 2 || (B > 0 && (!way_x || !way_y || !x || !y || !s || !n_samples))) return PO_ERR_INVALID;
    if (B == 0) return PO_OK;
    if (sizeof(double) * 2 * (size_t)W > 64 * 1024) return PO_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(po_internal_device(h)));
    HIP_TRY(po_launch_bspline(B, W, n_way, way_x, way_y, M, x, y, s, n_samples, po_internal_stream(h)));
    return PO_OK;
}

int po_segment_raw_batch_device(po_handle h, const po_spline_in *raw, int P, double *x, double *y, double *s, double *angle, double *k, int *n_points) {
    if (!h || P < 1 || !raw || (raw->B > 0 && (!x || !y || !s || !angle || !k || !n_points))) return PO_ERR_INVALID;
    po::DevSpline D{};
    PO_TRY(dev_spline(h, raw, &D));
    if (raw->B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(po_internal_device(h)));
    HIP_TRY(po_launch_segment_raw(&D, P, x, y, s, angle, k, n_points, po_internal_stream(h)));
    return PO_OK;
}

int po_post_project_batch_device(po_handle h, const po_spline_in *spline, int L, const int *n_layers, const double *layer_s, const double *offsets, double *x, double *y,
                                 double *s, double *length) {
    if (!h || L < 1 || !spline || (spline->B > 0 && (!layer_s || !offsets || !x || !y || !s))) return PO_ERR_INVALID;
    po::DevSpline D{};
    PO_TRY(dev_spline(h, spline, &D));
    if (spline->B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(po_internal_device(h)));
    HIP_TRY(po_launch_post_project(&D, L, n_layers, layer_s, offsets, x, y, s, length, po_internal_stream(h)));
    return PO_OK;
}

int po_segment_init_batch_device(po_handle h, const po_spline_in *spline, const double *start, int start_stride, const double *goal, int goal_stride, double *init, int *ok) {
    if (!h || !spline || start_stride < 3 || goal_stride < 2 || (spline->B > 0 && (!spline->length || !start || !goal || !init || !ok))) return PO_ERR_INVALID;
    po::DevSpline D{};
    PO_TRY(dev_spline(h, spline, &D));
    if (spline->B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(po_internal_device(h)));
    HIP_TRY(po_launch_segment_init(&D, start, start_stride, goal, goal_stride, po_internal_params(h)->enable_exact_position, init, ok, po_internal_stream(h)));
    return PO_OK;
}

static int plan_device_locked(po_handle h, const po_plan_in *in, const po_plan_out *out) {
    if (!h || !in || !out || in->B < 0 || in->W < 4 || in->N < 2) return PO_ERR_INVALID;
    if (in->B > 0 && (!in->way_x || !in->way_y || !in->start || !in->goal || !out->states || !out->n_states || !out->ok)) return PO_ERR_INVALID;
    if (!(in->max_length > 0)) return PO_ERR_INVALID;  // the device entry cannot look at the waypoints
    if (!po_internal_has_map(h)) return PO_ERR_INVALID;
    const int B = in->B;
    if (B == 0) return PO_OK;
    const po_params *prm = po_internal_params(h);
    hipStream_t st = po_internal_stream(h);
    HIP_TRY(hipSetDevice(po_internal_device(h)));
    // capacities from the waypoint polyline length: a B-spline is never longer than its control polygon
    const double Lmax = in->max_length;
    const int M = (int)std::ceil(Lmax) + 6, P = (int)std::ceil(Lmax) + 6;
    const int Lc = std::min(512, std::max((int)std::ceil((Lmax + 3) / prm->search_long_spacing), 14) + 8);  // layers every search_long_spacing (0.5 m when <= 6 m long)
    const int N = in->N;
    const size_t bM = (size_t)B * M, bP = (size_t)B * P, bL = (size_t)B * Lc, bN = (size_t)B * N;
    size_t need = sizeof(double) * (3 * bM + 8 * bP + 7 * bL + 13 * bN + 5 * bN + 16 * (size_t)B) + sizeof(po_info) * 3 * (size_t)B + sizeof(int) * 16 * (size_t)B + 4096;
    need += sizeof(double) * (20 * bN + 8 * (size_t)B) + sizeof(po_info) * (size_t)B;  // group staging + KPC limits (worst case: one group of everything)
    const bool raw_out = prm->enable_raw_output != 0;
    if (!raw_out) need += sizeof(double) * 5 * bN + 64;  // QP states before the densifying output branch
    Arena A;
    A.p = static_cast<char *>(po_internal_arena(h, need));
    if (!A.p) return PO_ERR_NOMEM;
    A.cap = need;
    double *bs_x = A.take<double>(bM), *bs_y = A.take<double>(bM), *bs_s = A.take<double>(bM);
    double *rw_x = A.take<double>(bP), *rw_y = A.take<double>(bP), *rw_s = A.take<double>(bP), *rw_a = A.take<double>(bP), *rw_k = A.take<double>(bP);
    double *t2_x = A.take<double>(bP), *t2_y = A.take<double>(bP), *t2_s = A.take<double>(bP);
    double *ly_s = A.take<double>(bL), *ly_lb = A.take<double>(bL), *ly_ub = A.take<double>(bL), *ly_off = A.take<double>(bL);
    double *k2_x = A.take<double>(bL), *k2_y = A.take<double>(bL), *k2_s = A.take<double>(bL);
    double *rf_x = A.take<double>(bN), *rf_y = A.take<double>(bN), *rf_z = A.take<double>(bN), *rf_k = A.take<double>(bN), *rf_s = A.take<double>(bN);
    double *bnd = A.take<double>(8 * bN);
    double *len1 = A.take<double>(B), *len2 = A.take<double>(B), *len3 = A.take<double>(B), *l0 = A.take<double>(B), *start3 = A.take<double>(3 * (size_t)B);
    double *init = A.take<double>(3 * (size_t)B), *x0 = A.take<double>(3 * (size_t)B), *goal_z = A.take<double>(B);
    po_info *info1 = A.take<po_info>(B), *info2 = A.take<po_info>(B), *info3 = out->info ? out->info : A.take<po_info>(B);
    int *n_bs = A.take<int>(B), *n_raw = A.take<int>(B), *n_lay = A.take<int>(B), *n_ref = A.take<int>(B), *n_val = A.take<int>(B), *okseg = A.take<int>(B);
    int *keep = A.take<int>(B), *stage = out->stage ? out->stage : A.take<int>(B), *gidx = A.take<int>(B), *g_n = A.take<int>(B);
    double *g_x = A.take<double>(bN), *g_y = A.take<double>(bN), *g_z = A.take<double>(bN), *g_k = A.take<double>(bN), *g_s = A.take<double>(bN);
    double *g_b = A.take<double>(8 * bN), *g_x0 = A.take<double>(3 * (size_t)B), *g_goal = A.take<double>(B), *g_states = A.take<double>(5 * bN);
    po_info *g_info = A.take<po_info>(B);
    double *lim_k = A.take<double>(bN), *lim_kp = A.take<double>(bN);
    double *qp_states = raw_out ? out->states : A.take<double>(5 * bN);  // optimizePath's two output branches (path_optimizer.cpp:191 / :201)
    if (A.off > A.cap) return PO_ERR_NOMEM;

    po::PlanGate G{};
    G.B = B; G.stage = stage; G.start = in->start; G.goal = in->goal;
    G.mode = 0; G.start3 = start3;
    HIP_TRY(po_launch_plan_gate(&G, st));
    // 1. bSpline, 2. segmentRawReference
    PO_TRY(po_bspline_batch_device(h, B, in->W, in->n_way, in->way_x, in->way_y, M, bs_x, bs_y, bs_s, n_bs));
    po_spline_in raw{B, M, bs_s, bs_x, bs_y, n_bs, nullptr};
    PO_TRY(po_segment_raw_batch_device(h, &raw, P, rw_x, rw_y, rw_s, rw_a, rw_k, n_raw));
    // 3. TensionSmoother2::osqpSmooth
    const int smoother = prm->smoothing_method == PO_SMOOTH_TENSION ? PO_SMOOTH_TENSION : PO_SMOOTH_TENSION2;  // ReferencePathSmoother::create
    po_smooth_in s1{smoother, B, P, n_raw, rw_x, rw_y, rw_a, rw_k, rw_s, nullptr, nullptr, nullptr};
    po_smooth_out o1{t2_x, t2_y, t2_s, info1, nullptr};
    PO_TRY(po_smooth_batch_device(h, &s1, &o1));
    G.mode = 1; G.cnt = n_raw; G.cnt2 = n_bs; G.info = info1; G.s = t2_s; G.stride = P; G.length = len1;
    HIP_TRY(po_launch_plan_gate(&G, st));
    // 4. graphSearchDp on the smoothed spline (knots = the result lists, max_s = result_s.back() + 3)
    po_spline_in sp1{B, P, t2_s, t2_x, t2_y, n_raw, len1};
    PO_TRY(po_dp_search_batch_device(h, &sp1, start3, Lc, ly_s, ly_lb, ly_ub, l0, n_lay));
    G.mode = 2; G.cnt = n_lay;
    HIP_TRY(po_launch_plan_gate(&G, st));
    // 5. postSmooth: QP, then the re-projection -> second spline
    po_smooth_in s2{PO_SMOOTH_POST, B, Lc, n_lay, nullptr, nullptr, nullptr, nullptr, ly_s, ly_lb, ly_ub, l0};
    po_smooth_out o2{ly_off, nullptr, nullptr, info2, nullptr};
    PO_TRY(po_smooth_batch_device(h, &s2, &o2));
    G.mode = 3; G.info = info2;
    HIP_TRY(po_launch_plan_gate(&G, st));
    PO_TRY(po_post_project_batch_device(h, &sp1, Lc, n_lay, ly_s, ly_off, k2_x, k2_y, k2_s, len2));
    // 6. segmentSmoothedPath: initial errors, goal trim, re-sampling, corridor bounds
    po_spline_in sp2{B, Lc, k2_s, k2_x, k2_y, n_lay, len2};
    PO_TRY(po_segment_init_batch_device(h, &sp2, in->start, 4, in->goal, 3, init, okseg));
    G.mode = 4; G.init = init; G.ok = okseg; G.length = len3;
    HIP_TRY(po_launch_plan_gate(&G, st));
    po_spline_in sp3{B, Lc, k2_s, k2_x, k2_y, n_lay, len3};
    // path_optimizer.cpp:171-172: 0.15 / FLAGS_output_spacing when the QP states are output directly, 0.5 / 1.0 when they are densified later
    PO_TRY(po_resample_batch_device(h, &sp3, raw_out ? 0.15 : 0.5, raw_out ? prm->output_spacing : 1.0, N, rf_x, rf_y, rf_z, rf_k, rf_s, n_ref));
    po_bounds_in bi{B, N, Lc, rf_x, rf_y, rf_z, rf_s, n_ref, k2_s, k2_x, k2_y, n_lay};
    PO_TRY(po_bounds_batch_device(h, &bi, bnd, n_val));
    G.mode = 5; G.cnt = n_val; G.ref_s = rf_s; G.ref_stride = N; G.x0 = x0; G.goal_z = goal_z; G.keep = keep;
    HIP_TRY(po_launch_plan_gate(&G, st));
    // 7. the path QP, grouped by keep_control_steps_ (one launch = one keep)
    std::vector<int> h_stage(B), h_keep(B), h_nval(B), h_nref(B);
    HIP_TRY(hipMemcpyAsync(h_stage.data(), stage, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_keep.data(), keep, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_nval.data(), n_val, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_nref.data(), n_ref, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int form = prm->optimization_method == PO_K ? PO_K : (prm->optimization_method == PO_KPC ? PO_KPC : PO_KP);
    std::map<int, std::vector<int>> groups;
    bool capacity = false;
    for (int b = 0; b < B; ++b) {
        if (h_nref[b] == -2 && h_stage[b] == 6) { h_stage[b] = 9; capacity = true; }  // N too small for the re-sampled reference
        if (h_stage[b] == 0) groups[form == PO_KP ? h_keep[b] : (form == PO_KPC ? 4 : 1)].push_back(b);  // K has no held control, KPC fixes keep = 4
    }
    if (form == PO_KPC) {
        std::vector<double> hk((size_t)bN, std::tan(prm->max_steer) / prm->wheel_base), hkp((size_t)bN, 1.7976931348623157e308);
        HIP_TRY(hipMemcpyAsync(lim_k, hk.data(), sizeof(double) * bN, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(lim_kp, hkp.data(), sizeof(double) * bN, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    for (auto &kv : groups) {
        const std::vector<int> &ids = kv.second;
        const int Gn = (int)ids.size();
        int Ng = 2;
        for (int b : ids) Ng = std::max(Ng, h_nval[b]);
        HIP_TRY(hipMemcpyAsync(gidx, ids.data(), sizeof(int) * Gn, hipMemcpyHostToDevice, st));
        po::PlanRows R{};
        R.G = Gn; R.N = N; R.Ng = Ng; R.idx = gidx;
        R.ref_x = rf_x; R.ref_y = rf_y; R.ref_z = rf_z; R.ref_k = rf_k; R.ref_s = rf_s; R.bounds = bnd; R.x0 = x0; R.goal_z = goal_z; R.n_valid = n_val;
        R.g_x = g_x; R.g_y = g_y; R.g_z = g_z; R.g_k = g_k; R.g_s = g_s; R.g_bounds = g_b; R.g_x0 = g_x0; R.g_goal = g_goal; R.g_n = g_n;
        R.g_states = g_states; R.g_info = g_info; R.states = qp_states; R.info = info3;
        HIP_TRY(po_launch_plan_gather(&R, st));
        // KPC with a spline-built reference: updateLimits() has no speed profile ("Reference states must be given directly!") and falls back to
        // max_k = tan(max_steering_angle) / wheel_base, max_kp = DBL_MAX (reference_path_impl.cpp:214-222)
        po_batch_in qi{form, Gn, Ng, kv.first, g_x, g_y, g_z, g_k, g_s, g_b, g_x0, g_goal, form == PO_KPC ? lim_k : nullptr, form == PO_KPC ? lim_kp : nullptr, g_n};
        po_batch_out qo{g_states, g_info, nullptr};
        const int rc = po_solve_batch_device(h, &qi, &qo);
        if (rc == PO_ERR_UNSUPPORTED) {  // does not fit the on-chip tile: flagged per instance, the others go on
            for (int b : ids) h_stage[b] = 9;
            capacity = true;
            HIP_TRY(hipStreamSynchronize(st));  // gidx is reused by the next group
            continue;
        }
        PO_TRY(rc);
        HIP_TRY(po_launch_plan_scatter(&R, st));
        HIP_TRY(hipStreamSynchronize(st));  // gidx / staging are reused by the next group
    }
    if (capacity) HIP_TRY(hipMemcpyAsync(stage, h_stage.data(), sizeof(int) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(po_launch_plan_clear(B, N, stage, qp_states, info3, st));
    G.mode = 6; G.cnt = n_val; G.info = info3;
    HIP_TRY(po_launch_plan_gate(&G, st));
    // 8. the tail of optimizePath: arc length + collision check + truncation rule
    if (raw_out) PO_TRY(po_postcheck_batch_device(h, B, N, n_val, out->states, info3, out->n_states, out->ok));
    else PO_TRY(po_densify_batch_device(h, B, N, n_val, qp_states, info3, N, out->states, out->n_states, out->ok));
    G.mode = 7; G.cnt = out->n_states; G.ok = out->ok;
    HIP_TRY(po_launch_plan_gate(&G, st));
    return PO_OK;
}

int po_plan_batch_device(po_handle h, const po_plan_in *in, const po_plan_out *out) {
    if (!h) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(*po_internal_plan_mutex(h));
    return plan_device_locked(h, in, out);
}

int po_plan_batch(po_handle h, const po_plan_in *in, const po_plan_out *out) {
    if (!h || !in || !out || in->B < 0 || in->W < 4 || in->N < 2) return PO_ERR_INVALID;
    if (in->B > 0 && (!in->way_x || !in->way_y || !in->start || !in->goal || !out->states || !out->n_states || !out->ok)) return PO_ERR_INVALID;
    const int B = in->B;
    if (B == 0) return PO_OK;
    double Lmax = in->max_length;
    if (!(Lmax > 0)) {
        Lmax = 1.0;
        for (int b = 0; b < B; ++b) {
            const int n = in->n_way ? in->n_way[b] : in->W;
            double len = 0;
            for (int i = 0; i + 1 < n && i + 1 < in->W; ++i) len += std::hypot(in->way_x[(size_t)b * in->W + i + 1] - in->way_x[(size_t)b * in->W + i], in->way_y[(size_t)b * in->W + i + 1] - in->way_y[(size_t)b * in->W + i]);
            Lmax = std::max(Lmax, len);
        }
    }
    hipStream_t st = po_internal_stream(h);
    HIP_TRY(hipSetDevice(po_internal_device(h)));
    const size_t bw = sizeof(double) * (size_t)B * in->W, bstates = sizeof(double) * (size_t)B * in->N * 5;
    const size_t total = 2 * bw + sizeof(double) * 7 * (size_t)B + sizeof(int) * 4 * (size_t)B + bstates + sizeof(po_info) * (size_t)B + 256;
    std::lock_guard<std::mutex> g(*po_internal_plan_mutex(h));
    void *raw = po_internal_plan_host(h, total);  // staging block of the host-pointer entry (the arena belongs to the device entry)
    if (!raw) return PO_ERR_NOMEM;
    Arena A; A.p = static_cast<char *>(raw); A.cap = total;
    double *wx = A.take<double>((size_t)B * in->W), *wy = A.take<double>((size_t)B * in->W), *d_start = A.take<double>(4 * (size_t)B), *d_goal = A.take<double>(3 * (size_t)B);
    double *d_states = A.take<double>((size_t)B * in->N * 5);
    po_info *d_info = A.take<po_info>(B);
    int *d_nway = A.take<int>(B), *d_n = A.take<int>(B), *d_ok = A.take<int>(B), *d_stage = A.take<int>(B);
    int rc = PO_OK;
    auto fail = [&](int code) { return code; };
    if (po_internal_hip_fail(hipMemcpyAsync(wx, in->way_x, bw, hipMemcpyHostToDevice, st), "H2D way_x")) return fail(PO_ERR_HIP);
    if (po_internal_hip_fail(hipMemcpyAsync(wy, in->way_y, bw, hipMemcpyHostToDevice, st), "H2D way_y")) return fail(PO_ERR_HIP);
    if (po_internal_hip_fail(hipMemcpyAsync(d_start, in->start, sizeof(double) * 4 * B, hipMemcpyHostToDevice, st), "H2D start")) return fail(PO_ERR_HIP);
    if (po_internal_hip_fail(hipMemcpyAsync(d_goal, in->goal, sizeof(double) * 3 * B, hipMemcpyHostToDevice, st), "H2D goal")) return fail(PO_ERR_HIP);
    if (in->n_way && po_internal_hip_fail(hipMemcpyAsync(d_nway, in->n_way, sizeof(int) * B, hipMemcpyHostToDevice, st), "H2D n_way")) return fail(PO_ERR_HIP);
    po_plan_in din = *in;
    din.way_x = wx; din.way_y = wy; din.start = d_start; din.goal = d_goal; din.n_way = in->n_way ? d_nway : nullptr; din.max_length = Lmax;
    po_plan_out dout{d_states, d_n, d_ok, d_stage, d_info};
    rc = plan_device_locked(h, &din, &dout);
    if (rc != PO_OK) return fail(rc);
    if (po_internal_hip_fail(hipMemcpyAsync(out->states, d_states, bstates, hipMemcpyDeviceToHost, st), "D2H states")) return fail(PO_ERR_HIP);
    if (po_internal_hip_fail(hipMemcpyAsync(out->n_states, d_n, sizeof(int) * B, hipMemcpyDeviceToHost, st), "D2H n")) return fail(PO_ERR_HIP);
    if (po_internal_hip_fail(hipMemcpyAsync(out->ok, d_ok, sizeof(int) * B, hipMemcpyDeviceToHost, st), "D2H ok")) return fail(PO_ERR_HIP);
    if (out->stage && po_internal_hip_fail(hipMemcpyAsync(out->stage, d_stage, sizeof(int) * B, hipMemcpyDeviceToHost, st), "D2H stage")) return fail(PO_ERR_HIP);
    if (out->info && po_internal_hip_fail(hipMemcpyAsync(out->info, d_info, sizeof(po_info) * B, hipMemcpyDeviceToHost, st), "D2H info")) return fail(PO_ERR_HIP);
    if (po_internal_hip_fail(hipStreamSynchronize(st), "sync")) return fail(PO_ERR_HIP);
    return PO_OK;
}

}  // extern "C"
