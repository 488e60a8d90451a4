// Host-side mirror of the reference's solver interface, backed by libpo_hip.so (C ABI, include/po_hip.h).
//
//   reference                                                    here
//   OsqpSolver::create(type, ref, vehicle, horizon)              same signature, same "K"/"KP"/"KPC" strings (solver.cpp:30-44)
//   virtual bool OsqpSolver::solve(std::vector<State>*)          same: true iff the QP status is `solved` (solver.cpp:46-77)
//   (none: one path per call)                                    OsqpSolver::solveBatch(): many independent planning instances
//
// Differences, on purpose: an unknown type still yields nullptr like the reference, but solveBatch
// reports it as an error instead of silently producing an empty "successful" path; parameters are captured in a
// po_params block at construction instead of being read from gflags globals during assembly.
#pragma once
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "po_hip.h"  // repo-root include/ on the include path
#ifdef PO_USE_REFERENCE_TYPES
// drop-in build (host/dropin/path_optimizer/solver/solver.hpp): State, CoveringCircleBounds, ReferencePath and VehicleState are the REFERENCE's own
// classes; this header only reads them through the getters the reference's solver reads them through
#include "path_optimizer/data_struct/data_struct.hpp"
#include "path_optimizer/data_struct/reference_path.hpp"
#include "path_optimizer/data_struct/vehicle_state_frenet.hpp"
#else
#include "data_struct.hpp"
#endif

namespace PathOptimizationNS {

// One engine (HIP device + stream) shared by every solver object of the process unless told otherwise.
class PoEngine {
 public:
    explicit PoEngine(int device = 0, const po_params *params = nullptr) {
        po_params p;
        if (params) p = *params; else po_default_params(&p);
        const int rc = po_create(device, &p, &h_);
        if (rc != PO_OK) throw std::runtime_error(std::string("po_create: ") + po_strerror(rc) + " " + po_last_hip_error());
        params_ = p;
    }
    ~PoEngine() { if (h_) po_destroy(h_); }
    PoEngine(const PoEngine &) = delete;
    PoEngine &operator=(const PoEngine &) = delete;
    po_handle handle() const { return h_; }
    const po_params &params() const { return params_; }
    static PoEngine &instance() { static PoEngine e; return e; }
 private:
    po_handle h_{};
    po_params params_{};
};

struct PlanningInstance {  // what one OsqpSolver object holds references to in the reference (solver.hpp:50-51)
    const ReferencePath *reference_path;
    const VehicleState *vehicle_state;
};

class OsqpSolver {
 public:
    OsqpSolver() = delete;
    OsqpSolver(int formulation, const ReferencePath &reference_path, const VehicleState &vehicle_state, const size_t &horizon, PoEngine *engine = nullptr)
        : formulation_(formulation), horizon_(horizon), reference_path_(reference_path), vehicle_state_(vehicle_state),
          engine_(engine ? engine : &PoEngine::instance()) {}
    virtual ~OsqpSolver() = default;

    static std::unique_ptr<OsqpSolver> create(std::string &type, const ReferencePath &reference_path, const VehicleState &vehicle_state,
                                              const size_t &horizon) {
        int f;
        if (type == "K") f = PO_K;
        else if (type == "KP") f = PO_KP;
        else if (type == "KPC") f = PO_KPC;
        else return nullptr;  // reference: LOG(ERROR) << "No such solver!" and nullptr (solver.cpp:40-43)
        return std::unique_ptr<OsqpSolver>(new OsqpSolver(f, reference_path, vehicle_state, horizon));
    }

    // Same contract as the reference: fills optimized_path and returns true iff the solver status is `solved`.
    virtual bool solve(std::vector<State> *optimized_path) {
        if (!optimized_path) return false;
        std::vector<std::vector<State>> out;
        std::vector<po_info> info;
        const PlanningInstance inst{&reference_path_, &vehicle_state_};
        const int rc = solveBatch(formulation_, &inst, 1, horizon_, &out, &info, engine_);
        optimized_path->clear();
        if (rc != PO_OK || info[0].status != PO_STATUS_SOLVED) return false;  // solver.cpp:72-73: any failure -> false
        *optimized_path = std::move(out[0]);
        return true;
    }

    // New: B independent planning instances of the same formulation and horizon in one launch.
    static int solveBatch(int formulation, const PlanningInstance *inst, size_t B, size_t horizon, std::vector<std::vector<State>> *paths,
                          std::vector<po_info> *info, PoEngine *engine = nullptr) {
        if (!inst || !paths || !info || horizon < 2) return PO_ERR_INVALID;
        PoEngine *eng = engine ? engine : &PoEngine::instance();
        const size_t N = horizon;
        std::vector<double> rx(B * N), ry(B * N), rz(B * N), rk(B * N), rs(B * N), bd(B * N * 8), x0(B * 3), gz(B), mk, mkp;
        if (formulation == PO_KPC) { mk.resize(B * N); mkp.resize(B * N); }
        int keep = 0;
        for (size_t b = 0; b < B; ++b) {  // AoS -> SoA pack
            const auto &st = inst[b].reference_path->getReferenceStates();
            const auto &bnd = inst[b].reference_path->getBounds();
            if (st.size() < N || bnd.size() < N) return PO_ERR_INVALID;
            for (size_t i = 0; i < N; ++i) {
                const size_t o = b * N + i;
                rx[o] = st[i].x; ry[o] = st[i].y; rz[o] = st[i].z; rk[o] = st[i].k; rs[o] = st[i].s;
                const CoveringCircleBounds::SingleCircleBounds *c[4] = {&bnd[i].c0, &bnd[i].c1, &bnd[i].c2, &bnd[i].c3};
                for (int j = 0; j < 4; ++j) { bd[o * 8 + 2 * j] = c[j]->lb; bd[o * 8 + 2 * j + 1] = c[j]->ub; }
            }
            if (formulation == PO_KPC) {
                const auto &a = inst[b].reference_path->getMaxKList();
                const auto &c = inst[b].reference_path->getMaxKpList();
                if (a.size() < N || c.size() < N) return PO_ERR_INVALID;
                for (size_t i = 0; i < N; ++i) { mk[b * N + i] = a[i]; mkp[b * N + i] = c[i]; }
            }
            const auto e = inst[b].vehicle_state->getInitError();
            x0[b * 3] = e[0]; x0[b * 3 + 1] = e[1]; x0[b * 3 + 2] = inst[b].vehicle_state->getStartState().k;
            gz[b] = inst[b].vehicle_state->getEndState().z;
            const int kb = po_keep_control_steps(formulation, &rs[b * N], (int)N);  // solver.cpp:22-27 + solver_kp_as_input.cpp:17
            if (kb < 0) return kb;
            if (b == 0) keep = kb;
            else if (kb != keep) return PO_ERR_INVALID;  // one batch = one (N, keep)
        }
        int n, m, C;
        int rc = po_problem_dims(formulation, (int)N, keep, &n, &m, &C);
        if (rc) return rc;
        po_batch_in in{formulation, (int)B, (int)N, keep, rx.data(), ry.data(), rz.data(), rk.data(), rs.data(), bd.data(), x0.data(), gz.data(),
                       formulation == PO_KPC ? mk.data() : nullptr, formulation == PO_KPC ? mkp.data() : nullptr, nullptr};
        std::vector<double> states(B * N * 5);
        info->assign(B, po_info{});
        po_batch_out out{states.data(), info->data(), nullptr};
        rc = po_solve_batch(eng->handle(), &in, &out);
        if (rc) return rc;
        paths->assign(B, {});
        for (size_t b = 0; b < B; ++b) {
            auto &p = (*paths)[b];
            p.reserve(N);
            for (size_t i = 0; i < N; ++i) {
                const double *s = &states[(b * N + i) * 5];
                p.emplace_back(s[0], s[1], s[2], s[3], s[4]);  // v = a = 0, like getOptimizedPath
            }
        }
        return PO_OK;
    }

 protected:
    const int formulation_;
    const size_t horizon_{};
    const ReferencePath &reference_path_;
    const VehicleState &vehicle_state_;
    PoEngine *engine_;
};

}  // namespace PathOptimizationNS
