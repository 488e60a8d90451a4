// ref_glue_bounds.cpp — drives the REFERENCE's own ReferencePathImpl::updateBoundsImproved (compiled from
// /root/reference/src/data_struct/reference_path_impl.cpp, src/tools/{Map,tools,spline}.cpp, src/config/planning_flags.cpp,
// unmodified, against the stand-in headers in this directory).  TEST INFRASTRUCTURE ONLY; built into
// oracle/_ref/libpo_ref_bounds.so (its own library: ref_glue.cpp defines a stub ReferencePathImpl of the same name).
// Pins: updateBoundsImproved, getApproxState, getClearanceWithDirectionStrict (default flags: simple boundary decision),
// tk::spline::set_points / operator(), global2Local, constraintAngle, isEqual.  Does NOT pin grid_map's interpolation.
#include <vector>

#include "path_optimizer/config/planning_flags.hpp"
#include "path_optimizer/data_struct/data_struct.hpp"
#include "path_optimizer/data_struct/reference_path_impl.hpp"
#include "path_optimizer/tools/Map.hpp"
#include "path_optimizer/tools/spline.h"

void updateConfig();  // planning_flags.cpp

// This library is compiled at -O0 (see oracle/Makefile: a missing return statement in data_struct.hpp:76-79 is miscompiled at
// -O1+).  The reference's Release build folds pow(x, 2) into x * x (gcc does so from -O1 on); at -O0 the call reaches libm,
// whose pow differs from x * x by one ulp once in a few thousand arguments.  Interpose pow inside this library (-Bsymbolic)
// so that the arithmetic is the Release build's.
#include <dlfcn.h>
extern "C" double pow(double x, double y) {
    if (y == 2.0) return x * x;
    static double (*real)(double, double) = (double (*)(double, double))dlsym(RTLD_NEXT, "pow");
    return real(x, y);
}

extern "C" {

// tk::spline through (ks, kv): value at `n` abscissae
void po_ref_spline_eval(int K, const double *ks, const double *kv, int n, const double *at, double *out) {
    PathOptimizationNS::tk::spline s;
    s.set_points(std::vector<double>(ks, ks + K), std::vector<double>(kv, kv + K));
    for (int i = 0; i < n; ++i) out[i] = s(at[i]);
}

// FLAGS_enable_simple_boundary_decision of the reference-compiled code (default true, src/config/planning_flags.cpp:84); returns the previous value
int po_ref_set_simple_boundary_decision(int on) {
    const int prev = FLAGS_enable_simple_boundary_decision ? 1 : 0;
    FLAGS_enable_simple_boundary_decision = on != 0;
    return prev;
}

// One path: reference states + the knots its x(s), y(s) splines were set from -> bounds [n_valid][4][2] (lb, ub).
int po_ref_bounds_path(const po_map *m, int N, const double *ref_x, const double *ref_y, const double *ref_z, const double *ref_s, int K,
                       const double *ks, const double *kx, const double *ky, double *bounds /*[N][4][2]*/) {
    using namespace PathOptimizationNS;
    updateConfig();
    grid_map::GridMap gm(*m);
    Map map(gm);
    tk::spline xs, ys;
    xs.set_points(std::vector<double>(ks, ks + K), std::vector<double>(kx, kx + K));
    ys.set_points(std::vector<double>(ks, ks + K), std::vector<double>(ky, ky + K));
    ReferencePathImpl impl;
    std::vector<State> states;
    for (int i = 0; i < N; ++i) states.emplace_back(ref_x[i], ref_y[i], ref_z[i], 0.0, ref_s[i]);
    impl.setReference(states);
    impl.setSpline(xs, ys, ks[K - 1]);
    impl.updateBoundsImproved(map);
    const auto &b = impl.getBounds();
    for (size_t i = 0; i < b.size(); ++i) {
        const CoveringCircleBounds::SingleCircleBounds *c[4] = {&b[i].c0, &b[i].c1, &b[i].c2, &b[i].c3};
        for (int j = 0; j < 4; ++j) { bounds[(i * 4 + j) * 2] = c[j]->lb; bounds[(i * 4 + j) * 2 + 1] = c[j]->ub; }
    }
    return (int)b.size();
}
}
