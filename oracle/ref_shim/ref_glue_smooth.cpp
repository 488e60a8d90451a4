// ref_glue_smooth.cpp — drives the REFERENCE's own reference-smoothing QP code, compiled from
// /root/reference/src/reference_path_smoother/{reference_path_smoother,tension_smoother,tension_smoother_2,angle_diff_smoother}.cpp,
// src/data_struct/{reference_path,reference_path_impl}.cpp, src/tools/{Map,tools,spline}.cpp and src/config/planning_flags.cpp,
// unmodified, against the stand-in headers in this directory.  TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libpo_ref_smooth.so.
//
// Pins: TensionSmoother2::{setHessianMatrix,setGradient,setConstraintMatrix,osqpSmooth's output loop},
// TensionSmoother::{setHessianMatrix,setConstraintMatrix,osqpSmooth} (incl. the map clearance of every point),
// ReferencePathSmoother::{setPostHessianMatrix,setPostConstraintMatrix,postSmooth} and the flag defaults they read.
// Does NOT pin OSQP (absent): OsqpEigen::Solver::solve() forwards to the oracle's ADMM.
#include <cstring>
#include <vector>

#define private public      // the entry points are private members of the reference's classes; access only, no layout change
#define protected public
#include "path_optimizer/reference_path_smoother/tension_smoother_2.hpp"
#include "path_optimizer/path_optimizer.hpp"
#include "path_optimizer/tools/spline.h"
#undef private
#undef protected
#include "OsqpEigen/OsqpEigen.h"
#include "path_optimizer/config/planning_flags.hpp"
#include "path_optimizer/data_struct/reference_path.hpp"
#include "path_optimizer/tools/Map.hpp"
#include "path_optimizer/data_struct/vehicle_state_frenet.hpp"
#include "path_optimizer/tools/spline.h"

namespace OsqpEigen {
static Captured g_cap;
static po_params g_params;
Captured &last_captured() { return g_cap; }
const po_params &shim_params() { return g_params; }
}  // namespace OsqpEigen

void updateConfig();  // planning_flags.cpp

extern "C" {

void po_ref_smooth_flags(double *out /*[6]*/) {
    const double v[6] = {FLAGS_tension_2_deviation_weight, FLAGS_tension_2_curvature_weight, FLAGS_tension_2_curvature_rate_weight,
                         FLAGS_cartesian_curvature_weight, FLAGS_cartesian_curvature_rate_weight, FLAGS_cartesian_deviation_weight};
    std::memcpy(out, v, sizeof(v));
}

// kind 0: TensionSmoother2::osqpSmooth, kind 1: TensionSmoother::osqpSmooth (virtual dispatch on the real classes).
// Returns what osqpSmooth returned (1/0).  The captured QP is read back with po_ref_smooth_get_*.
int po_ref_osqp_smooth(int kind, const po_map *m, int P, const double *x, const double *y, const double *angle, const double *k,
                       const double *s, const po_params *admm, double *rx, double *ry, double *rs) {
    using namespace PathOptimizationNS;
    updateConfig();
    OsqpEigen::g_params = *admm;
    po_map empty{};
    grid_map::GridMap gm(m ? *m : empty);
    Map map(gm);
    std::vector<State> input(4);
    State start;
    std::vector<double> vx(x, x + P), vy(y, y + P), va(angle, angle + P), vk(k, k + P), vs(s, s + P), ox, oy, os;
    bool ok;
    if (kind == 0) { TensionSmoother2 sm(input, start, map); ok = static_cast<TensionSmoother &>(sm).osqpSmooth(vx, vy, va, vk, vs, &ox, &oy, &os); }
    else { TensionSmoother sm(input, start, map); ok = sm.osqpSmooth(vx, vy, va, vk, vs, &ox, &oy, &os); }
    for (size_t i = 0; i < ox.size(); ++i) { rx[i] = ox[i]; ry[i] = oy[i]; rs[i] = os[i]; }
    return ok ? 1 : 0;
}

// ReferencePathSmoother::postSmooth on given layers (layers_s_list_, layers_bounds_, vehicle_l_wrt_smoothed_ref_) and a reference
// spline through (ks, kx, ky).  Outputs: the new spline's knots (s_list, x_list, y_list) re-read from the ReferencePath.
int po_ref_post_smooth(int L, const double *layer_s, const double *lb, const double *ub, double l0, int K, const double *ks,
                       const double *kx, const double *ky, const po_params *admm, double *new_s_end, double *new_x_at, double *new_y_at) {
    using namespace PathOptimizationNS;
    updateConfig();
    OsqpEigen::g_params = *admm;
    po_map empty{};
    grid_map::GridMap gm(empty);
    Map map(gm);
    std::vector<State> input(4);
    State start;
    TensionSmoother2 sm(input, start, map);
    sm.layers_s_list_.assign(layer_s, layer_s + L);
    sm.layers_bounds_.clear();
    for (int i = 0; i < L; ++i) sm.layers_bounds_.emplace_back(lb[i], ub[i]);
    sm.vehicle_l_wrt_smoothed_ref_ = l0;
    ReferencePath ref;
    tk::spline xs, ys;
    xs.set_points(std::vector<double>(ks, ks + K), std::vector<double>(kx, kx + K));
    ys.set_points(std::vector<double>(ks, ks + K), std::vector<double>(ky, ky + K));
    ref.setSpline(xs, ys, ks[K - 1]);
    const bool ok = sm.postSmooth(&ref);
    if (ok) {  // sample the re-fitted spline at L abscissae spread over its length
        *new_s_end = ref.getLength();
        for (int i = 0; i < L; ++i) {
            const double at = ref.getLength() * i / (L - 1);
            new_x_at[i] = ref.getXS(at);
            new_y_at[i] = ref.getYS(at);
        }
    }
    return ok ? 1 : 0;
}

// ReferencePathSmoother::graphSearchDp (private; reference_path_smoother.cpp:147-300) on a spline reference and an obstacle map.
// Returns -1 when it returns false, else the number of layers kept; outputs layers_s_list_, layers_bounds_, vehicle_l_wrt_smoothed_ref_.
int po_ref_dp_search(const po_map *m, int K, const double *ks, const double *kx, const double *ky, double length, const double *start,
                     double *layer_s, double *lb, double *ub, double *l0) {
    using namespace PathOptimizationNS;
    updateConfig();
    grid_map::GridMap gm(*m);
    Map map(gm);
    std::vector<State> input(4);
    State st(start[0], start[1], start[2]);
    TensionSmoother2 sm(input, st, map);
    ReferencePath ref;
    tk::spline xs, ys;
    xs.set_points(std::vector<double>(ks, ks + K), std::vector<double>(kx, kx + K));
    ys.set_points(std::vector<double>(ks, ks + K), std::vector<double>(ky, ky + K));
    ref.setSpline(xs, ys, length);
    const bool ok = sm.graphSearchDp(&ref);
    *l0 = sm.vehicle_l_wrt_smoothed_ref_;
    if (!ok) return -1;
    for (size_t i = 0; i < sm.layers_bounds_.size(); ++i) { layer_s[i] = sm.layers_s_list_[i]; lb[i] = sm.layers_bounds_[i].first; ub[i] = sm.layers_bounds_[i].second; }
    return (int)sm.layers_bounds_.size();
}

// ReferencePath::buildReferenceFromSpline (reference_path_impl.cpp:474-499): returns the number of states, -1 if it returns false.
int po_ref_resample(int K, const double *ks, const double *kx, const double *ky, double max_s, double ds_smaller, double ds_larger, int cap,
                    double *ox, double *oy, double *oz, double *ok, double *os) {
    using namespace PathOptimizationNS;
    updateConfig();
    ReferencePath ref;
    tk::spline xs, ys;
    xs.set_points(std::vector<double>(ks, ks + K), std::vector<double>(kx, kx + K));
    ys.set_points(std::vector<double>(ks, ks + K), std::vector<double>(ky, ky + K));
    ref.setSpline(xs, ys, max_s);
    if (!ref.buildReferenceFromSpline(ds_smaller, ds_larger)) return -1;
    const auto &st = ref.getReferenceStates();
    for (size_t i = 0; i < st.size() && (int)i < cap; ++i) { ox[i] = st[i].x; oy[i] = st[i].y; oz[i] = st[i].z; ok[i] = st[i].k; os[i] = st[i].s; }
    return (int)st.size();
}

// ReferencePath::updateLimits (reference_path_impl.cpp:203-235) with FLAGS_optimization_method = "KPC" and states given directly.
void po_ref_limits(int N, const double *v, const double *a, double *max_k, double *max_kp) {
    using namespace PathOptimizationNS;
    updateConfig();
    const std::string keep = FLAGS_optimization_method;
    FLAGS_optimization_method = "KPC";
    ReferencePath ref;
    std::vector<State> states;
    for (int i = 0; i < N; ++i) { State s(0.1 * i, 0, 0, 0, 0.1 * i); s.v = v[i]; s.a = a[i]; states.push_back(s); }
    ref.setReference(states);
    ref.updateLimits();
    for (int i = 0; i < N; ++i) { max_k[i] = ref.getMaxKList()[i]; max_kp[i] = ref.getMaxKpList()[i]; }
    FLAGS_optimization_method = keep;
}

// ReferencePathSmoother::bSpline (private) on the given input points: x_list_, y_list_, s_list_.
int po_ref_bspline(int n, const double *px, const double *py, int cap, double *x, double *y, double *s) {
    using namespace PathOptimizationNS;
    po_map empty{};
    grid_map::GridMap gm(empty);
    Map map(gm);
    std::vector<State> input;
    for (int i = 0; i < n; ++i) input.emplace_back(px[i], py[i]);
    State st;
    TensionSmoother2 sm(input, st, map);
    sm.bSpline();
    const int m = (int)sm.x_list_.size();
    for (int i = 0; i < m && i < cap; ++i) { x[i] = sm.x_list_[i]; y[i] = sm.y_list_[i]; s[i] = sm.s_list_[i]; }
    return m;
}

// ReferencePathSmoother::segmentRawReference (protected) on given dense lists.
int po_ref_segment_raw(int K, const double *ks, const double *kx, const double *ky, int cap, double *x, double *y, double *s, double *angle, double *k) {
    using namespace PathOptimizationNS;
    po_map empty{};
    grid_map::GridMap gm(empty);
    Map map(gm);
    std::vector<State> input(4);
    State st;
    TensionSmoother2 sm(input, st, map);
    sm.s_list_.assign(ks, ks + K); sm.x_list_.assign(kx, kx + K); sm.y_list_.assign(ky, ky + K);
    std::vector<double> vx, vy, vs, va, vk;
    if (!sm.segmentRawReference(&vx, &vy, &vs, &va, &vk)) return -1;
    const int n = (int)vs.size();
    for (int i = 0; i < n && i < cap; ++i) { x[i] = vx[i]; y[i] = vy[i]; s[i] = vs[i]; angle[i] = va[i]; k[i] = vk[i]; }
    return n;
}

// PathOptimizer::segmentSmoothedPath (private) on a spline reference: returns its value; out = init offset, heading error, length after the
// goal trim, number of reference states, then the states (x, y, z, k, s) of the re-sampled reference (before any bound truncation: map huge).
int po_ref_segment_smoothed(const po_map *m, int K, const double *ks, const double *kx, const double *ky, double length, const double *start /*x,y,z,k*/,
                            const double *goal /*x,y,z*/, double *out4, int cap, double *states /*[cap][5]*/) {
    using namespace PathOptimizationNS;
    updateConfig();
    grid_map::GridMap gm(*m);
    State st(start[0], start[1], start[2], start[3]), en(goal[0], goal[1], goal[2]);
    PathOptimizer opt(st, en, gm);
    tk::spline xs, ys;
    xs.set_points(std::vector<double>(ks, ks + K), std::vector<double>(kx, kx + K));
    ys.set_points(std::vector<double>(ks, ks + K), std::vector<double>(ky, ky + K));
    opt.reference_path_->setSpline(xs, ys, length);
    const bool ok = opt.segmentSmoothedPath();
    const auto err = opt.vehicle_state_->getInitError();
    out4[0] = err[0]; out4[1] = err[1]; out4[2] = opt.reference_path_->getLength();
    const auto &rs = opt.reference_path_->getReferenceStates();
    out4[3] = (double)rs.size();
    for (size_t i = 0; i < rs.size() && (int)i < cap; ++i) { states[5 * i] = rs[i].x; states[5 * i + 1] = rs[i].y; states[5 * i + 2] = rs[i].z; states[5 * i + 3] = rs[i].k; states[5 * i + 4] = rs[i].s; }
    return ok ? 1 : 0;
}

// The reference's top-level entry points.  PathOptimizer::solve(reference_points, &final_path) and solveWithoutSmoothing (which needs the
// spline a previous solve() left in reference_path_: set here from the knots).  Returns the bool; path [n][5] = x, y, z, k, s.
int po_ref_path_optimizer_solve(const po_map *m, int n_pts, const double *px, const double *py, const double *start /*x,y,z,k*/, const double *goal /*x,y,z*/,
                                const po_params *admm, int cap, double *path, int *n_path) {
    using namespace PathOptimizationNS;
    updateConfig();
    OsqpEigen::g_params = *admm;
    const bool keep_raw = FLAGS_enable_raw_output;
    FLAGS_enable_raw_output = admm->enable_raw_output != 0;  // the two output branches of optimizePath
    const std::string keep_sm = FLAGS_smoothing_method, keep_om = FLAGS_optimization_method;
    FLAGS_smoothing_method = admm->smoothing_method == PO_SMOOTH_TENSION ? "TENSION" : "TENSION2";
    FLAGS_optimization_method = admm->optimization_method == PO_K ? "K" : (admm->optimization_method == PO_KPC ? "KPC" : "KP");
    const bool keep_ex = FLAGS_enable_exact_position;
    FLAGS_enable_exact_position = admm->enable_exact_position != 0;
    struct Restore { bool v; std::string sm, om; bool ex; ~Restore() { FLAGS_enable_raw_output = v; FLAGS_smoothing_method = sm; FLAGS_optimization_method = om; FLAGS_enable_exact_position = ex; } } restore{keep_raw, keep_sm, keep_om, keep_ex};
    grid_map::GridMap gm(*m);
    State st(start[0], start[1], start[2], start[3]), en(goal[0], goal[1], goal[2]);
    PathOptimizer opt(st, en, gm);
    std::vector<State> pts, result;
    for (int i = 0; i < n_pts; ++i) pts.emplace_back(px[i], py[i]);
    const bool ok = opt.solve(pts, &result);
    *n_path = (int)result.size();
    for (size_t i = 0; i < result.size() && (int)i < cap; ++i) { path[5 * i] = result[i].x; path[5 * i + 1] = result[i].y; path[5 * i + 2] = result[i].z; path[5 * i + 3] = result[i].k; path[5 * i + 4] = result[i].s; }
    return ok ? 1 : 0;
}
int po_ref_path_optimizer_solve_without_smoothing(const po_map *m, int N, const double *rx, const double *ry, const double *rz, const double *rk, const double *rs,
                                                  int K, const double *ks, const double *kx, const double *ky, const double *start, const double *goal,
                                                  const po_params *admm, int cap, double *path, int *n_path) {
    using namespace PathOptimizationNS;
    updateConfig();
    OsqpEigen::g_params = *admm;
    grid_map::GridMap gm(*m);
    State st(start[0], start[1], start[2], start[3]), en(goal[0], goal[1], goal[2]);
    PathOptimizer opt(st, en, gm);
    tk::spline xs, ys;
    xs.set_points(std::vector<double>(ks, ks + K), std::vector<double>(kx, kx + K));
    ys.set_points(std::vector<double>(ks, ks + K), std::vector<double>(ky, ky + K));
    opt.reference_path_->setSpline(xs, ys, ks[K - 1]);
    std::vector<State> pts, result;
    for (int i = 0; i < N; ++i) pts.emplace_back(rx[i], ry[i], rz[i], rk[i], rs[i]);
    const bool ok = opt.solveWithoutSmoothing(pts, &result);
    *n_path = (int)result.size();
    for (size_t i = 0; i < result.size() && (int)i < cap; ++i) { path[5 * i] = result[i].x; path[5 * i + 1] = result[i].y; path[5 * i + 2] = result[i].z; path[5 * i + 3] = result[i].k; path[5 * i + 4] = result[i].s; }
    return ok ? 1 : 0;
}

// The reference's own benchmark, src/test/path_optimizer_benchmark.cpp: BM_optimizePath = PathOptimizer(start, goal, map).solve(points, &path1)
// (:84-100) and BM_optimizePathWithoutSmoothing = solveWithoutSmoothing(path1, &path2) ON THE SAME OBJECT (:153-158: it reads the spline the first
// solve left in reference_path_).  Also returns that spline's knots and length, and the iteration count / status of the path QP of each call.
// Return value: bit 0 = solve() returned true, bit 1 = solveWithoutSmoothing() returned true.
int po_ref_benchmark(const po_map *m, int n_pts, const double *px, const double *py, const double *start /*x,y,z,k*/, const double *goal /*x,y,z*/,
                     const po_params *admm, int cap, double *path1, int *n1, double *path2, int *n2, int kcap, double *ks, double *kx, double *ky, int *K,
                     double *max_s, po_info *qp1, po_info *qp2) {
    using namespace PathOptimizationNS;
    updateConfig();
    OsqpEigen::g_params = *admm;
    grid_map::GridMap gm(*m);
    State st(start[0], start[1], start[2], start[3]), en(goal[0], goal[1], goal[2]);
    PathOptimizer opt(st, en, gm);
    std::vector<State> pts, r1, r2;
    for (int i = 0; i < n_pts; ++i) pts.emplace_back(px[i], py[i]);
    const bool ok1 = opt.solve(pts, &r1);
    *qp1 = OsqpEigen::g_cap.info;
    auto put = [cap](const std::vector<State> &r, double *path, int *n) {
        *n = (int)r.size();
        for (size_t i = 0; i < r.size() && (int)i < cap; ++i) { path[5 * i] = r[i].x; path[5 * i + 1] = r[i].y; path[5 * i + 2] = r[i].z; path[5 * i + 3] = r[i].k; path[5 * i + 4] = r[i].s; }
    };
    put(r1, path1, n1);
    const tk::spline &xs = opt.reference_path_->getXS(), &ys = opt.reference_path_->getYS();
    *K = (int)xs.m_x.size();
    for (int i = 0; i < *K && i < kcap; ++i) { ks[i] = xs.m_x[i]; kx[i] = xs.m_y[i]; ky[i] = ys.m_y[i]; }
    *max_s = opt.reference_path_->getLength();
    bool ok2 = false;
    *n2 = 0;
    if (ok1) { ok2 = opt.solveWithoutSmoothing(r1, &r2); *qp2 = OsqpEigen::g_cap.info; put(r2, path2, n2); }
    return (ok1 ? 1 : 0) | (ok2 ? 2 : 0);
}

void po_ref_smooth_get_dims(int *n, int *m, int *pnz, int *anz) {
    const auto &c = OsqpEigen::g_cap;
    *n = c.n; *m = c.m; *pnz = (int)c.Px.size(); *anz = (int)c.Ax.size();
}
void po_ref_smooth_get_qp(int *Pp, int *Pi, double *Px, int *Ap, int *Ai, double *Ax, double *q, double *l, double *u, double *x) {
    const auto &c = OsqpEigen::g_cap;
    std::memcpy(Pp, c.Pp.data(), sizeof(int) * c.Pp.size()); std::memcpy(Pi, c.Pi.data(), sizeof(int) * c.Pi.size());
    std::memcpy(Px, c.Px.data(), sizeof(double) * c.Px.size());
    std::memcpy(Ap, c.Ap.data(), sizeof(int) * c.Ap.size()); std::memcpy(Ai, c.Ai.data(), sizeof(int) * c.Ai.size());
    std::memcpy(Ax, c.Ax.data(), sizeof(double) * c.Ax.size());
    std::memcpy(q, c.q.data(), sizeof(double) * c.q.size());
    std::memcpy(l, c.l.data(), sizeof(double) * c.l.size()); std::memcpy(u, c.u.data(), sizeof(double) * c.u.size());
    if (x && !c.x.empty()) std::memcpy(x, c.x.data(), sizeof(double) * c.x.size());
}
}
