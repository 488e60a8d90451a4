// ref_glue.cpp — drives the REFERENCE's own solver classes (compiled from /root/reference/src/solver/*.cpp,
// src/config/planning_flags.cpp and src/data_struct/vehicle_state_frenet.cpp, unmodified, against the stand-in
// headers in this directory) and exposes what they assembled.  TEST INFRASTRUCTURE ONLY; built into oracle/_ref/.
//
// What this pins: the reference's setHessianMatrix / setConstraintMatrix / getOptimizedPath and its flag defaults
// are the real code.  What it does NOT pin: OSQP itself (absent) — OsqpEigen::Solver::solve() forwards to the oracle.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "OsqpEigen/OsqpEigen.h"
#include "path_optimizer/config/planning_flags.hpp"
#include "path_optimizer/data_struct/data_struct.hpp"
#include "path_optimizer/data_struct/reference_path.hpp"
#include "path_optimizer/data_struct/vehicle_state_frenet.hpp"
#include "path_optimizer/solver/solver.hpp"

namespace OsqpEigen {
static Captured g_cap;
static po_params g_params;
Captured &last_captured() { return g_cap; }
const po_params &shim_params() { return g_params; }
}  // namespace OsqpEigen

// Our own definition of the reference's pimpl: only the containers the solver reads
// (include/path_optimizer/data_struct/reference_path.hpp:34-37).
namespace PathOptimizationNS {
class ReferencePathImpl {
 public:
    std::vector<State> states;
    std::vector<CoveringCircleBounds> bounds;
    std::vector<double> max_k, max_kp;
};
ReferencePath::ReferencePath() : reference_path_impl_(std::make_shared<ReferencePathImpl>()) {}
std::size_t ReferencePath::getSize() const { return reference_path_impl_->states.size(); }
const std::vector<State> &ReferencePath::getReferenceStates() const { return reference_path_impl_->states; }
const std::vector<CoveringCircleBounds> &ReferencePath::getBounds() const { return reference_path_impl_->bounds; }
const std::vector<double> &ReferencePath::getMaxKList() const { return reference_path_impl_->max_k; }
const std::vector<double> &ReferencePath::getMaxKpList() const { return reference_path_impl_->max_kp; }
}  // namespace PathOptimizationNS

void updateConfig();  // planning_flags.cpp

extern "C" {

// FLAGS the hot path reads, as the reference's own planning_flags.cpp + updateConfig() define them.
void po_ref_flags(double *out /*[16]*/) {
    updateConfig();
    const double v[16] = {FLAGS_d1, FLAGS_d2, FLAGS_d3, FLAGS_d4, FLAGS_KP_curvature_weight, FLAGS_KP_curvature_rate_weight,
                          FLAGS_KP_deviation_weight, FLAGS_KP_slack_weight, FLAGS_K_curvature_weight, FLAGS_K_curvature_rate_weight,
                          FLAGS_K_deviation_weight, FLAGS_expected_safety_margin, FLAGS_max_steering_angle, FLAGS_wheel_base,
                          FLAGS_constraint_end_heading ? 1.0 : 0.0, FLAGS_mu};
    std::memcpy(out, v, sizeof(v));
}

// Run OsqpSolver::create(type)->solve() of the reference on one path.  Returns: -1 unknown type (create() == nullptr),
// 0 solve() returned false, 1 true.  Captured QP: n, m, nnz and CSC arrays can then be read with po_ref_get_*.
int po_ref_solve(const char *type, int N, const double *ref_x, const double *ref_y, const double *ref_z, const double *ref_k,
                 const double *ref_s, const double *bounds /*[N][4][2] lb,ub*/, const double *x0 /*[3]*/, double goal_z,
                 const double *max_k, const double *max_kp, const po_params *admm, double *out_states /*[N][5]*/, int *n_out_states) {
    using namespace PathOptimizationNS;
    updateConfig();
    OsqpEigen::g_params = *admm;
    ReferencePath ref;
    auto &impl = *reinterpret_cast<std::shared_ptr<ReferencePathImpl> *>(&ref);  // sole data member (reference_path.hpp:49)
    for (int i = 0; i < N; ++i) {
        impl->states.emplace_back(ref_x[i], ref_y[i], ref_z[i], ref_k[i], ref_s[i]);
        CoveringCircleBounds b;
        CoveringCircleBounds::SingleCircleBounds *c[4] = {&b.c0, &b.c1, &b.c2, &b.c3};
        for (int j = 0; j < 4; ++j) { c[j]->lb = bounds[(i * 4 + j) * 2 + 0]; c[j]->ub = bounds[(i * 4 + j) * 2 + 1]; }
        impl->bounds.push_back(b);
        if (max_k) impl->max_k.push_back(max_k[i]);
        if (max_kp) impl->max_kp.push_back(max_kp[i]);
    }
    State start(ref_x[0], ref_y[0], ref_z[0], x0[2], 0), goal(ref_x[N - 1], ref_y[N - 1], goal_z);
    VehicleState vs(start, goal, x0[0], x0[1]);
    std::string t(type);
    auto solver = OsqpSolver::create(t, ref, vs, (size_t)N);
    if (!solver) return -1;
    std::vector<State> path;
    const bool ok = solver->solve(&path);
    *n_out_states = (int)path.size();
    for (size_t i = 0; i < path.size(); ++i) {
        out_states[5 * i + 0] = path[i].x; out_states[5 * i + 1] = path[i].y; out_states[5 * i + 2] = path[i].z;
        out_states[5 * i + 3] = path[i].k; out_states[5 * i + 4] = path[i].s;
    }
    return ok ? 1 : 0;
}

void po_ref_get_dims(int *n, int *m, int *pnz, int *anz) {
    const auto &c = OsqpEigen::g_cap;
    *n = c.n; *m = c.m; *pnz = (int)c.Px.size(); *anz = (int)c.Ax.size();
}
void po_ref_get_qp(int *Pp, int *Pi, double *Px, int *Ap, int *Ai, double *Ax, double *q, double *l, double *u, double *x) {
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
