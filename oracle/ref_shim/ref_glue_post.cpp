// ref_glue_post.cpp — drives the REFERENCE's own CollisionChecker / CarGeometry / Map / tools (compiled from
// /root/reference/src/tools/{collision_checker,car_geometry,Map,tools,spline}.cpp, unmodified, against the stand-in headers
// in this directory).  TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libpo_ref.so.
// Pins: isSingleStateCollisionFree{,Improved}, the seven footprint circles, local2Global, distance, Map's inside/0.0 rule.
// Does NOT pin grid_map's interpolation (the shim forwards it to the oracle).
#include <vector>

#include "path_optimizer/config/planning_flags.hpp"
#include "path_optimizer/data_struct/data_struct.hpp"
#include "path_optimizer/tools/collosion_checker.hpp"
#include "path_optimizer/tools/tools.hpp"

void updateConfig();  // planning_flags.cpp

extern "C" {

int po_ref_collision_free(const po_map *m, double x, double y, double heading) {
    using namespace PathOptimizationNS;
    updateConfig();
    grid_map::GridMap gm(*m);
    CollisionChecker cc(gm);
    State st(x, y, heading);
    return cc.isSingleStateCollisionFreeImproved(st) ? 1 : 0;
}

double po_ref_map_distance(const po_map *m, double x, double y) {
    grid_map::GridMap gm(*m);
    PathOptimizationNS::Map map(gm);
    return map.getObstacleDistance(Eigen::Vector2d(x, y));
}

// The raw-output branch of optimizePath's tail (path_optimizer.cpp:191-200) driven through the reference's own
// distance() and collision checker: arc length accumulated point by point, stop at the first colliding state.
int po_ref_postcheck(const po_map *m, int n, const double *states /*[n][5]*/, int *n_valid, double *s_out /*[n]*/) {
    using namespace PathOptimizationNS;
    updateConfig();
    grid_map::GridMap gm(*m);
    CollisionChecker cc(gm);
    std::vector<State> path;
    for (int i = 0; i < n; ++i) path.emplace_back(states[5 * i], states[5 * i + 1], states[5 * i + 2], states[5 * i + 3], 0.0);
    double s = 0;
    for (int i = 0; i < n; ++i) {
        if (i > 0) s += distance(path[i - 1], path[i]);
        path[i].s = s;
        s_out[i] = s;
        if (FLAGS_enable_collision_check && !cc.isSingleStateCollisionFreeImproved(path[i])) {
            *n_valid = i;
            return i > 0 && path[i - 1].s >= 20;
        }
    }
    *n_valid = n;
    return 1;
}
}
