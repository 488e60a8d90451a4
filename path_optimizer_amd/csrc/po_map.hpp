// po_map.hpp — obstacle-distance map sampling and the post-solve collision check, gfx950.
//
// Reference: PathOptimizationNS::Map::{getObstacleDistance,isInside} (src/tools/Map.cpp:16-26),
// CollisionChecker::isSingleStateCollisionFree{,Improved} (src/tools/collision_checker.cpp:17-59),
// CarGeometry (src/tools/car_geometry.cpp:38-72), the output loop of PathOptimizer::optimizePath
// (src/path_optimizer/path_optimizer.cpp:183-200).
// grid_map (ROS package grid_map_core, NOT in /root/reference, un-pinned) is restated from its published sources:
// GridMap::isInside -> checkIfPositionWithinMap, GridMap::atPosition(INTER_LINEAR) -> atPositionLinearInterpolated with the
// nearest-cell fallback, index/position conversions of GridMapMath.cpp.  "parity unpinned" for that part.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/po_hip.h"
#include "../../include/po_pmath.h"  // portable sin / cos / atan2: the same IEEE operation sequence as the oracle's portable-math mode (bit-exact map stages)

namespace po {

struct DevMap {
    const float *d;  // column-major [sy][sx]
    int sx, sy;
    double res, px, py;
};
struct DevCar {  // CollisionChecker's CarGeometry, built on the host exactly like car_geometry.cpp:38-56
    double bx, br;            // bounding circle (local x, radius); local y = 0
    double cx[6], cy[6], cr[6];
    int enable;
};

// corridor-bounds producer launch arguments (po_post.hip)
struct DevBounds {
    int B, N, K;
    const double *ref_x, *ref_y, *ref_z, *ref_s;
    const int *n_points;
    const double *knot_s, *knot_x, *knot_y;
    const int *n_knots;
    double d[4], radius;
    double *coef;  // [B][2][6][K]: per spline a, b, c + 3K scratch
};

// re-sampling / DP search launch arguments (po_post.hip), SURVEY.md §8f-4
struct DevSpline {
    int B, K;
    const double *knot_s, *knot_x, *knot_y;
    const int *n_knots;
    const double *length;
    double *coef;  // [B][2][6][K] as in DevBounds
};
struct DevSearch {
    double range, long_spacing, lat_spacing;
    const double *start;  // [B][3]
    int L;                // stride / cap of the per-layer outputs
    double *layer_s, *lb, *ub, *l0;
    int *n_layers;
};
struct DevResample {
    double ds_small, ds_large;
    int dynamic, N;
    double *x, *y, *z, *k, *s;
    int *n_points;
};

// plumbing of po_plan_batch (po_post.hip kernels, po_plan.cpp orchestration)
struct PlanGate {
    int B, mode;
    int *stage;
    int *cnt;               // the count array the next stage reads (n_points / n_layers / n_valid): zeroed for failed instances
    const int *cnt2;        // mode 1: n_samples of the B-spline stage
    const po_info *info;    // modes 1, 3, 6: the QP info to test
    const double *s; int stride;  // mode 1: result_s (for length = s.back() + 3)
    double *length;         // modes 1, 4: length handed to the next spline stage
    const double *init; const int *ok;  // mode 4
    const double *start, *goal;         // mode 5: [B][4], [B][3]
    const double *ref_s; int ref_stride;  // mode 5: reference arc lengths (keep_control_steps)
    double *x0, *goal_z; int *keep;     // mode 5 outputs
    double *start3;         // mode 0: start [B][3] for the search
};
struct PlanRows {
    int G, N, Ng;
    const int *idx;  // [G] instance of each group row
    const double *ref_x, *ref_y, *ref_z, *ref_k, *ref_s, *bounds, *x0, *goal_z;
    const int *n_valid;
    double *g_x, *g_y, *g_z, *g_k, *g_s, *g_bounds, *g_x0, *g_goal;
    int *g_n;
    double *g_states; po_info *g_info;  // solve outputs of the group
    double *states; po_info *info;      // [B][N][5], [B]
};

#ifdef PO_MAP_DEVICE_CODE  // kernels and device functions: po_kernels.hip only (po_capi.cpp needs just the structs)
// checkIfPositionWithinMap (GridMapMath.cpp): t = -(p - mapPos - 0.5*len); 0 <= t < len on both axes
__device__ __forceinline__ bool map_inside(const DevMap &m, double x, double y) {
    const double lx = m.sx * m.res, ly = m.sy * m.res;
    const double tx = -(__dsub_rn(__dsub_rn(x, m.px), __dmul_rn(0.5, lx)));
    const double ty = -(__dsub_rn(__dsub_rn(y, m.py), __dmul_rn(0.5, ly)));
    return tx >= 0.0 && ty >= 0.0 && tx < lx && ty < ly;
}
// getIndexFromPosition: idx = (int)( -((p - 0.5*len - mapPos) / res) )   (C++ double->int conversion truncates)
__device__ __forceinline__ void map_index(const DevMap &m, double x, double y, int &ix, int &iy) {
    const double lx = m.sx * m.res, ly = m.sy * m.res;
    ix = (int)(-(__dsub_rn(__dsub_rn(x, __dmul_rn(0.5, lx)), m.px) / m.res));
    iy = (int)(-(__dsub_rn(__dsub_rn(y, __dmul_rn(0.5, ly)), m.py) / m.res));
}
__device__ __forceinline__ bool map_index_ok(const DevMap &m, int ix, int iy) { return ix >= 0 && iy >= 0 && ix < m.sx && iy < m.sy; }
// getPositionFromIndex: p = (mapPos + (0.5*len - 0.5*res)) + res * (-idx); leaves p untouched when idx is out of range
__device__ __forceinline__ void map_position(const DevMap &m, int ix, int iy, double &x, double &y) {
    if (!map_index_ok(m, ix, iy)) return;
    const double lx = m.sx * m.res, ly = m.sy * m.res;
    x = __dadd_rn(__dadd_rn(m.px, __dsub_rn(__dmul_rn(0.5, lx), __dmul_rn(0.5, m.res))), __dmul_rn(m.res, (double)(-ix)));
    y = __dadd_rn(__dadd_rn(m.py, __dsub_rn(__dmul_rn(0.5, ly), __dmul_rn(0.5, m.res))), __dmul_rn(m.res, (double)(-iy)));
}
// GridMap::atPosition(layer, p, INTER_LINEAR) for a position that isInside (the only way the reference calls it)
__device__ __forceinline__ double map_at_linear(const DevMap &m, double x, double y) {
    int i0x, i0y;
    map_index(m, x, y, i0x, i0y);
    double ptx = 0, pty = 0;
    map_position(m, i0x, i0y, ptx, pty);
    int ix[4], iy[4], sh[4];
    ix[0] = i0x; iy[0] = i0y;
    bool up;
    if (x >= ptx) { ix[1] = i0x - 1; iy[1] = i0y; up = true; } else { ix[1] = i0x + 1; iy[1] = i0y; up = false; }
    if (y >= pty) {
        ix[2] = i0x; iy[2] = i0y - 1;
        if (up) { sh[0] = 0; sh[1] = 1; sh[2] = 2; sh[3] = 3; } else { sh[0] = 1; sh[1] = 0; sh[2] = 3; sh[3] = 2; }
    } else {
        ix[2] = i0x; iy[2] = i0y + 1;
        if (up) { sh[0] = 2; sh[1] = 3; sh[2] = 0; sh[3] = 1; } else { sh[0] = 3; sh[1] = 2; sh[2] = 1; sh[3] = 0; }
    }
    ix[3] = ix[1]; iy[3] = iy[2];
    const long long nbuf = (long long)m.sx * m.sy;
    float f[4];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int jx = 0, jy = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (sh[i] == k) { jx = ix[k]; jy = iy[k]; }
        // the library tests the LINEAR index (size_t) against [0, size]: a row index of -1 wraps into the previous column
        const long long lin = (long long)jy * m.sx + jx;
        if (lin < 0 || lin > nbuf) ok = false;
        f[i] = (lin >= 0 && lin < nbuf) ? m.d[lin] : 0.0f;  // (lin == size would read one past the buffer in the library)
    }
    if (ok) {
        int jx = 0, jy = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (sh[0] == k) { jx = ix[k]; jy = iy[k]; }
        map_position(m, jx, jy, ptx, pty);
        const double rx = __dsub_rn(x, ptx) / m.res, ry = __dsub_rn(y, pty) / m.res;
        const double fx = __dsub_rn(1.0, rx), fy = __dsub_rn(1.0, ry);
        double v = __dmul_rn(__dmul_rn((double)f[0], fx), fy);
        v = __dadd_rn(v, __dmul_rn(__dmul_rn((double)f[1], rx), fy));
        v = __dadd_rn(v, __dmul_rn(__dmul_rn((double)f[2], fx), ry));
        v = __dadd_rn(v, __dmul_rn(__dmul_rn((double)f[3], rx), ry));
        return (double)(float)v;  // the library returns float
    }
    // INTER_NEAREST fallback
    return map_index_ok(m, i0x, i0y) ? (double)m.d[(long long)i0y * m.sx + i0x] : 0.0;
}
// Map::getObstacleDistance (Map.cpp:16-22)
__device__ __forceinline__ double map_distance(const DevMap &m, double x, double y) { return map_inside(m, x, y) ? map_at_linear(m, x, y) : 0.0; }

// CollisionChecker::isSingleStateCollisionFreeImproved (collision_checker.cpp:42-59)
__device__ __forceinline__ bool collision_free(const DevMap &m, const DevCar &c, double x, double y, double z) {
    const double cz = po_pcos(z), sz = po_psin(z);
    // local2Global (tools.cpp:50-55): x = tx cos - ty sin + rx ; y = tx sin + ty cos + ry
    const double bx = __dadd_rn(__dsub_rn(__dmul_rn(c.bx, cz), __dmul_rn(0.0, sz)), x);
    const double by = __dadd_rn(__dadd_rn(__dmul_rn(c.bx, sz), __dmul_rn(0.0, cz)), y);
    if (!map_inside(m, bx, by)) return false;
    if (!(map_at_linear(m, bx, by) < c.br)) return true;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        const double gx = __dadd_rn(__dsub_rn(__dmul_rn(c.cx[k], cz), __dmul_rn(c.cy[k], sz)), x);
        const double gy = __dadd_rn(__dadd_rn(__dmul_rn(c.cx[k], sz), __dmul_rn(c.cy[k], cz)), y);
        if (!map_inside(m, gx, gy)) return false;
        if (map_at_linear(m, gx, gy) < c.cr[k]) return false;
    }
    return true;
}

#endif  // PO_MAP_DEVICE_CODE

}  // namespace po
