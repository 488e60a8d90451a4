// Host-side mirror of the reference-smoothing stages, backed by libpo_hip.so:
//
//   reference                                                                     here
//   TensionSmoother2::osqpSmooth(x, y, angle, k, s, &rx, &ry, &rs)                 TensionSmoother2::osqpSmooth — same signature
//     (src/reference_path_smoother/tension_smoother_2.cpp:163-218)                   (po_smooth_batch, PO_SMOOTH_TENSION2)
//   TensionSmoother::osqpSmooth (tension_smoother.cpp:186-236)                     TensionSmoother::osqpSmooth (PO_SMOOTH_TENSION; needs the Map)
//   ReferencePathSmoother::graphSearchDp (reference_path_smoother.cpp:147-300)     graphSearchDp(knots, length, start_state, map, &layers)
//   ReferencePathSmoother::postSmooth's QP (:534-566)                              postSmoothOffsets(layers, &offsets)
//   ReferencePath::buildReferenceFromSpline(ds_smaller, ds_larger)                 buildReferenceFromSpline(&ref, knots, max_s, ...)
//     (src/data_struct/reference_path_impl.cpp:474-499)                              (po_resample_batch)
//   ReferencePath::updateLimits() (reference_path_impl.cpp:203-235)                updateLimits(&ref)  (po_limits_batch; State.v / State.a)
// The reference keeps these as private members working on member vectors; the arguments here are those vectors.
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "po_hip.h"
#include "data_struct.hpp"
#include "map_tools.hpp"
#include "solver.hpp"

namespace PathOptimizationNS {

class TensionSmoother2 {
 public:
    explicit TensionSmoother2(PoEngine *engine = nullptr) : engine_(engine ? engine : &PoEngine::instance()) {}
    virtual ~TensionSmoother2() = default;
    virtual bool osqpSmooth(const std::vector<double> &x_list, const std::vector<double> &y_list, const std::vector<double> &angle_list,
                            const std::vector<double> &k_list, const std::vector<double> &s_list, std::vector<double> *result_x_list,
                            std::vector<double> *result_y_list, std::vector<double> *result_s_list) {
        return run(kind(), x_list, y_list, angle_list, k_list, s_list, result_x_list, result_y_list, result_s_list);
    }
    const po_info &lastInfo() const { return info_; }
 protected:
    virtual int kind() const { return PO_SMOOTH_TENSION2; }
    bool run(int knd, const std::vector<double> &x, const std::vector<double> &y, const std::vector<double> &a, const std::vector<double> &k,
             const std::vector<double> &s, std::vector<double> *rx, std::vector<double> *ry, std::vector<double> *rs) {
        const size_t P = x.size();
        if (y.size() != P || a.size() != P || s.size() != P || k.size() < P - (P ? 1 : 0)) return false;  // CHECK_EQ in the reference
        std::vector<double> kk(k);
        kk.resize(P, 0.0);
        rx->assign(P, 0.0); ry->assign(P, 0.0); rs->assign(P, 0.0);
        po_smooth_in in{knd, 1, (int)P, nullptr, x.data(), y.data(), a.data(), kk.data(), s.data(), nullptr, nullptr, nullptr};
        po_smooth_out out{rx->data(), ry->data(), rs->data(), &info_, nullptr};
        const int rc = po_smooth_batch(engine_->handle(), &in, &out);
        if (rc != PO_OK) { rx->clear(); ry->clear(); rs->clear(); return false; }
        return info_.status == PO_STATUS_SOLVED;  // OsqpEigen: solve() is true only for OSQP_SOLVED
    }
    PoEngine *engine_;
    po_info info_{};
};

class TensionSmoother : public TensionSmoother2 {
 public:
    explicit TensionSmoother(const Map &map) : TensionSmoother2(map.engine()) {}
 protected:
    int kind() const override { return PO_SMOOTH_TENSION; }
};

struct SearchLayers {  // layers_s_list_, layers_bounds_, vehicle_l_wrt_smoothed_ref_ of ReferencePathSmoother
    std::vector<double> s;
    std::vector<std::pair<double, double>> bounds;
    double vehicle_l = 0;
};

// graphSearchDp: false when the reference returns false (vehicle further than FLAGS_search_lateral_range from the spline).
inline bool graphSearchDp(const SplineKnots &knots, double length, const State &start_state, const Map &map, SearchLayers *out) {
    const int K = (int)knots.s.size();
    int L = (int)(length / (length > 6 ? map.engine()->params().search_long_spacing : 0.5)) + 4;
    std::vector<double> ls((size_t)L), lb((size_t)L), ub((size_t)L);
    const double start[3] = {start_state.x, start_state.y, start_state.z};
    double l0 = 0;
    int n = 0;
    po_spline_in sp{1, K, knots.s.data(), knots.x.data(), knots.y.data(), nullptr, &length};
    const int rc = po_dp_search_batch(map.engine()->handle(), &sp, start, L, ls.data(), lb.data(), ub.data(), &l0, &n);
    if (rc != PO_OK) throw std::runtime_error(std::string("po_dp_search_batch: ") + po_strerror(rc));
    out->vehicle_l = l0;
    out->s.clear(); out->bounds.clear();
    if (n < 0) return false;
    for (int i = 0; i < n; ++i) { out->s.push_back(ls[(size_t)i]); out->bounds.emplace_back(lb[(size_t)i], ub[(size_t)i]); }
    return true;
}

// The QP of postSmooth: lateral offset of every layer.  false for fewer than 4 layers ("Ref is short") or an unsolved QP.
inline bool postSmoothOffsets(const SearchLayers &layers, std::vector<double> *offsets, PoEngine *engine = nullptr) {
    PoEngine *e = engine ? engine : &PoEngine::instance();
    const size_t L = layers.s.size();
    if (L < 4 || layers.bounds.size() != L) return false;
    std::vector<double> lb(L), ub(L);
    for (size_t i = 0; i < L; ++i) { lb[i] = layers.bounds[i].first; ub[i] = layers.bounds[i].second; }
    offsets->assign(L, 0.0);
    po_info info{};
    po_smooth_in in{PO_SMOOTH_POST, 1, (int)L, nullptr, nullptr, nullptr, nullptr, nullptr, layers.s.data(), lb.data(), ub.data(), &layers.vehicle_l};
    po_smooth_out out{offsets->data(), nullptr, nullptr, &info, nullptr};
    const int rc = po_smooth_batch(e->handle(), &in, &out);
    return rc == PO_OK && info.status == PO_STATUS_SOLVED;
}

inline bool buildReferenceFromSpline(ReferencePath *ref, const SplineKnots &knots, double max_s, double delta_s_smaller, double delta_s_larger,
                                     PoEngine *engine = nullptr) {
    PoEngine *e = engine ? engine : &PoEngine::instance();
    if (max_s <= 0) return false;
    const int N = (int)(max_s / delta_s_smaller) + 4;
    std::vector<double> x((size_t)N), y((size_t)N), z((size_t)N), k((size_t)N), s((size_t)N);
    int n = 0;
    po_spline_in sp{1, (int)knots.s.size(), knots.s.data(), knots.x.data(), knots.y.data(), nullptr, &max_s};
    const int rc = po_resample_batch(e->handle(), &sp, delta_s_smaller, delta_s_larger, N, x.data(), y.data(), z.data(), k.data(), s.data(), &n);
    if (rc != PO_OK) throw std::runtime_error(std::string("po_resample_batch: ") + po_strerror(rc));
    if (n < 0) return false;
    std::vector<State> st;
    for (int i = 0; i < n; ++i) st.emplace_back(x[(size_t)i], y[(size_t)i], z[(size_t)i], k[(size_t)i], s[(size_t)i]);
    ref->setReference(std::move(st));
    return true;
}

inline void updateLimits(ReferencePath *ref, PoEngine *engine = nullptr) {
    PoEngine *e = engine ? engine : &PoEngine::instance();
    const auto &st = ref->getReferenceStates();
    const size_t N = st.size();
    if (N == 0) return;  // "Empty reference, updateLimits() fail!"
    std::vector<double> v(N), a(N), mk(N), mkp(N);
    for (size_t i = 0; i < N; ++i) { v[i] = st[i].v; a[i] = st[i].a; }
    if (po_limits_batch(e->handle(), 1, (int)N, nullptr, v.data(), a.data(), mk.data(), mkp.data()) != PO_OK) throw std::runtime_error("po_limits_batch failed");
    ref->setLimits(std::move(mk), std::move(mkp));
}

}  // namespace PathOptimizationNS
