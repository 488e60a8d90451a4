// Host-side mirror of the reference's classes around the obstacle-distance map, backed by libpo_hip.so:
//
//   reference                                                          here
//   PathOptimizationNS::Map (include/path_optimizer/tools/Map.hpp)      Map: owns the layer "distance", uploads it to the engine once
//     getObstacleDistance(pos) / isInside(pos)  (src/tools/Map.cpp)       same names (device sampling, po_map_sample)
//   ReferencePath::updateBounds(const Map&)                             updateBounds(ReferencePath&, knots, map): po_bounds_batch
//     (-> ReferencePathImpl::updateBoundsImproved)                        fills the bounds and truncates the states like the reference
//   CollisionChecker::isSingleStateCollisionFreeImproved(State)         same name; checkPaths(): the batched tail of optimizePath
//     (src/tools/collision_checker.cpp)                                   (po_postcheck_batch)
//
// The reference builds its Map from a grid_map::GridMap; here the caller hands over that layer's raw buffer
// (gm["distance"].data(), gm.getSize(), gm.getResolution(), gm.getPosition()) — see INTEGRATION.md §C.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "po_hip.h"
#include "data_struct.hpp"
#include "solver.hpp"

namespace PathOptimizationNS {

class Map {
 public:
    Map(const float *distance_col_major, int size_x, int size_y, double resolution, double pos_x, double pos_y, PoEngine *engine = nullptr)
        : engine_(engine ? engine : &PoEngine::instance()) {
        po_map m{distance_col_major, size_x, size_y, resolution, pos_x, pos_y};
        const int rc = po_set_map(engine_->handle(), &m);
        if (rc != PO_OK) throw std::runtime_error(std::string("po_set_map: ") + po_strerror(rc));
    }
    double getObstacleDistance(double x, double y) const { double d; int in; sample(x, y, &d, &in); return d; }
    bool isInside(double x, double y) const { double d; int in; sample(x, y, &d, &in); return in != 0; }
    PoEngine *engine() const { return engine_; }
 private:
    void sample(double x, double y, double *d, int *in) const {
        const double xy[2] = {x, y};
        if (po_map_sample(engine_->handle(), 1, xy, d, in) != PO_OK) throw std::runtime_error("po_map_sample failed");
    }
    PoEngine *engine_;
};

// The knots x_s_ / y_s_ were set from (tk::spline::set_points; ReferencePath::setSpline in the reference).
struct SplineKnots { std::vector<double> s, x, y; };

// ReferencePath::updateBounds for many reference paths at once.  Returns PO_OK or an API error code.
inline int updateBoundsBatch(ReferencePath *const *refs, const SplineKnots *const *knots, size_t B, const Map &map) {
    if (B == 0) return PO_OK;
    size_t N = 0, K = 0;
    for (size_t b = 0; b < B; ++b) { if (refs[b]->getSize() > N) N = refs[b]->getSize(); if (knots[b]->s.size() > K) K = knots[b]->s.size(); }
    if (N < 1 || K < 3) return PO_ERR_INVALID;
    std::vector<double> rx(B * N), ry(B * N), rz(B * N), rs(B * N), ks(B * K), kx(B * K), ky(B * K), bounds(B * N * 8);
    std::vector<int> npts(B), nk(B), nvalid(B);
    for (size_t b = 0; b < B; ++b) {
        const auto &st = refs[b]->getReferenceStates();
        npts[b] = (int)st.size(); nk[b] = (int)knots[b]->s.size();
        if (nk[b] < 3 || knots[b]->x.size() != knots[b]->s.size() || knots[b]->y.size() != knots[b]->s.size()) return PO_ERR_INVALID;
        for (size_t i = 0; i < st.size(); ++i) { rx[b * N + i] = st[i].x; ry[b * N + i] = st[i].y; rz[b * N + i] = st[i].z; rs[b * N + i] = st[i].s; }
        for (int i = 0; i < nk[b]; ++i) { ks[b * K + i] = knots[b]->s[i]; kx[b * K + i] = knots[b]->x[i]; ky[b * K + i] = knots[b]->y[i]; }
        for (size_t i = (size_t)nk[b]; i < K; ++i) ks[b * K + i] = ks[b * K + i - 1] + 1.0;  // padding stays increasing
    }
    po_bounds_in in{(int)B, (int)N, (int)K, rx.data(), ry.data(), rz.data(), rs.data(), npts.data(), ks.data(), kx.data(), ky.data(), nk.data()};
    const int rc = po_bounds_batch(map.engine()->handle(), &in, bounds.data(), nvalid.data());
    if (rc != PO_OK) return rc;
    for (size_t b = 0; b < B; ++b) {
        std::vector<CoveringCircleBounds> out((size_t)nvalid[b]);
        for (int i = 0; i < nvalid[b]; ++i) {
            CoveringCircleBounds::SingleCircleBounds *c[4] = {&out[i].c0, &out[i].c1, &out[i].c2, &out[i].c3};
            for (int j = 0; j < 4; ++j) { c[j]->lb = bounds[((b * N + i) * 4 + j) * 2]; c[j]->ub = bounds[((b * N + i) * 4 + j) * 2 + 1]; }
        }
        std::vector<State> st = refs[b]->getReferenceStates();
        st.resize((size_t)nvalid[b]);  // reference_path_impl.cpp:198-200
        refs[b]->setReference(std::move(st));
        refs[b]->setBounds(std::move(out));
    }
    return PO_OK;
}
inline void updateBounds(ReferencePath &ref, const SplineKnots &knots, const Map &map) {
    ReferencePath *r = &ref; const SplineKnots *k = &knots;
    const int rc = updateBoundsBatch(&r, &k, 1, map);
    if (rc != PO_OK) throw std::runtime_error(std::string("po_bounds_batch: ") + po_strerror(rc));
}

class CollisionChecker {
 public:
    explicit CollisionChecker(const Map &map) : map_(map) {}
    bool isSingleStateCollisionFreeImproved(const State &current) const {
        const double st[5] = {current.x, current.y, current.z, current.k, 0.0};
        po_info info{}; info.status = PO_STATUS_SOLVED;
        int nv = 0, ok = 0;
        if (po_postcheck_batch(map_.engine()->handle(), 1, 1, nullptr, st, &info, &nv, &ok) != PO_OK) throw std::runtime_error("po_postcheck_batch failed");
        return nv == 1;
    }
    // The tail of optimizePath (path_optimizer.cpp:183-200) for a batch of solved paths: truncates each path at its first colliding
    // state and returns, per path, what optimizePath returns.
    std::vector<bool> checkPaths(std::vector<std::vector<State>> *paths, const std::vector<po_info> &info) const {
        const size_t B = paths->size();
        size_t N = 1;
        for (auto &p : *paths) if (p.size() > N) N = p.size();
        std::vector<double> st(B * N * 5, 0.0);
        std::vector<int> npts(B), nv(B), ok(B);
        for (size_t b = 0; b < B; ++b) {
            npts[b] = (int)(*paths)[b].size();
            for (size_t i = 0; i < (*paths)[b].size(); ++i) {
                const State &s = (*paths)[b][i];
                double *o = &st[(b * N + i) * 5];
                o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.k; o[4] = s.s;
            }
        }
        std::vector<bool> out(B, false);
        if (B == 0) return out;
        if (po_postcheck_batch(map_.engine()->handle(), (int)B, (int)N, npts.data(), st.data(), info.data(), nv.data(), ok.data()) != PO_OK)
            throw std::runtime_error("po_postcheck_batch failed");
        for (size_t b = 0; b < B; ++b) { (*paths)[b].resize((size_t)nv[b]); out[b] = ok[b] != 0; }
        return out;
    }
 private:
    const Map &map_;
};

}  // namespace PathOptimizationNS
