// Host-side mirror of the reference's top-level class, backed by libpo_hip.so:
//
//   reference (include/path_optimizer/path_optimizer.hpp:22-55)                     here
//   PathOptimizer(const State &start_state, const State &end_state,                  PathOptimizer(start_state, end_state, const Map &map)
//                 const grid_map::GridMap &map)                                        (Map = the "distance" layer uploaded once, map_tools.hpp)
//   bool solve(const std::vector<State> &reference_points,                           same signature: po_plan_batch with B = 1
//              std::vector<State> *final_path)        (src/path_optimizer/path_optimizer.cpp:40-85)
//   bool solveWithoutSmoothing(reference_points, final_path)  (:87-117)              same signature (+ the knots of the spline a previous solve() left
//                                                                                      behind, which the reference reads through reference_path_)
//   — new —                                                                          static solveBatch(): many planning instances in one call
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "po_hip.h"
#include "data_struct.hpp"
#include "map_tools.hpp"
#include "solver.hpp"

namespace PathOptimizationNS {

struct PlanningProblem {  // what one PathOptimizer object is constructed with and handed in solve()
    State start_state, end_state;
    std::vector<State> reference_points;
};

class PathOptimizer {
 public:
    PathOptimizer() = delete;
    PathOptimizer(const State &start_state, const State &end_state, const Map &map) : start_(start_state), end_(end_state), map_(map) {}
    PathOptimizer(const PathOptimizer &) = delete;
    PathOptimizer &operator=(const PathOptimizer &) = delete;

    // Call this to get the optimized path.
    bool solve(const std::vector<State> &reference_points, std::vector<State> *final_path) {
        if (!final_path) throw std::invalid_argument("final_path == nullptr");  // CHECK_NOTNULL
        if (reference_points.empty()) return false;                            // "Empty input, quit path optimization"
        PlanningProblem pr{start_, end_, reference_points};
        std::vector<std::vector<State>> paths;
        std::vector<int> stage;
        const std::vector<bool> ok = solveBatch(&pr, 1, map_, &paths, &stage);
        *final_path = std::move(paths[0]);
        last_stage_ = stage[0];
        return ok[0];
    }

    // "Calculate once more based on the previous result": the reference states are given directly; `knots` = the spline of the reference.
    bool solveWithoutSmoothing(const std::vector<State> &reference_points, const SplineKnots &knots, std::vector<State> *final_path) {
        if (!final_path) throw std::invalid_argument("final_path == nullptr");
        if (reference_points.empty()) return false;
        ReferencePath ref;
        ref.setReference(reference_points);
        updateBounds(ref, knots, map_);  // reference_path_->updateBounds(*grid_map_); updateLimits() is a no-op for "KP"
        VehicleState vs(start_, end_, 0, 0);  // vehicle_state_->setInitError(0, 0)
        if (ref.getSize() < 2) return false;
        // OsqpSolver::create(FLAGS_optimization_method, ...): the formulation comes from the engine's parameter block, as in po_plan_batch.  KPC needs
        // ReferencePath::updateLimits' lists; for a reference given directly without a speed profile they are kappa_max / DBL_MAX (reference_path_impl.cpp:214-222)
        const po_params &pp = map_.engine()->params();
        const int form = pp.optimization_method == PO_K ? PO_K : (pp.optimization_method == PO_KPC ? PO_KPC : PO_KP);
        if (form == PO_KPC) {
            const double kmax = std::tan(pp.max_steer) / pp.wheel_base;
            ref.setLimits(std::vector<double>(ref.getSize(), kmax), std::vector<double>(ref.getSize(), 1.7976931348623157e308));
        }
        OsqpSolver solver(form, ref, vs, ref.getSize(), map_.engine());
        std::vector<State> path;
        if (!solver.solve(&path)) return false;  // "QP failed."
        std::vector<std::vector<State>> one{path};
        std::vector<po_info> info(1);
        info[0].status = PO_STATUS_SOLVED;
        const std::vector<bool> ok = CollisionChecker(map_).checkPaths(&one, info);
        *final_path = std::move(one[0]);
        return ok[0];
    }

    // Many planning instances in one call (one PathOptimizer::solve each).  stage (optional): see po_plan_out in po_hip.h.
    static std::vector<bool> solveBatch(const PlanningProblem *problems, size_t B, const Map &map, std::vector<std::vector<State>> *final_paths,
                                        std::vector<int> *stage = nullptr, int max_states = 0) {
        std::vector<bool> ok(B, false);
        final_paths->assign(B, {});
        if (B == 0) return ok;
        size_t W = 4;
        double longest = 1.0;
        for (size_t b = 0; b < B; ++b) {
            const auto &rp = problems[b].reference_points;
            if (rp.size() > W) W = rp.size();
            double len = 0;
            for (size_t i = 0; i + 1 < rp.size(); ++i) len += std::hypot(rp[i + 1].x - rp[i].x, rp[i + 1].y - rp[i].y);
            if (len > longest) longest = len;
        }
        const int N = max_states > 0 ? max_states : (int)((longest + 3) / 0.15) + 8;  // worst case: 0.15 m spacing everywhere
        std::vector<double> wx(B * W, 0.0), wy(B * W, 0.0), st(B * 4), gl(B * 3), states(B * (size_t)N * 5);
        std::vector<int> nw(B), n(B), okv(B), stg(B);
        for (size_t b = 0; b < B; ++b) {
            const auto &rp = problems[b].reference_points;
            nw[b] = (int)rp.size();
            for (size_t i = 0; i < rp.size(); ++i) { wx[b * W + i] = rp[i].x; wy[b * W + i] = rp[i].y; }
            const State &s = problems[b].start_state, &e = problems[b].end_state;
            st[4 * b] = s.x; st[4 * b + 1] = s.y; st[4 * b + 2] = s.z; st[4 * b + 3] = s.k;
            gl[3 * b] = e.x; gl[3 * b + 1] = e.y; gl[3 * b + 2] = e.z;
        }
        po_plan_in in{(int)B, (int)W, nw.data(), wx.data(), wy.data(), st.data(), gl.data(), longest, N};
        po_plan_out out{states.data(), n.data(), okv.data(), stg.data(), nullptr};
        const int rc = po_plan_batch(map.engine()->handle(), &in, &out);
        if (rc != PO_OK) throw std::runtime_error(std::string("po_plan_batch: ") + po_strerror(rc) + " " + po_last_hip_error());
        for (size_t b = 0; b < B; ++b) {
            ok[b] = okv[b] != 0;
            for (int i = 0; i < n[b]; ++i) {
                const double *q = &states[(b * (size_t)N + (size_t)i) * 5];
                (*final_paths)[b].emplace_back(q[0], q[1], q[2], q[3], q[4]);
            }
        }
        if (stage) *stage = stg;
        return ok;
    }
    int lastStage() const { return last_stage_; }

 private:
    State start_, end_;
    const Map &map_;
    int last_stage_ = 0;
};

}  // namespace PathOptimizationNS
