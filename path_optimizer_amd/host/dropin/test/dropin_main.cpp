// Drop-in proof: the reference's OWN src/path_optimizer/path_optimizer.cpp — unchanged, compiled where it lies — linked against
// host/dropin/path_optimizer/solver/solver.hpp + libpo_hip.so instead of src/solver/*.cpp + OSQP.  This driver is the body of the reference's
// benchmark (src/test/path_optimizer_benchmark.cpp:84-100): PathOptimizer(start, goal, map).solve(points, &path).
//   dropin_test <scene.bin>   scene.bin = int32 size_x, size_y, n_pts | f64 resolution, pos_x, pos_y, start[4], goal[3] | f32 distance[size_y][size_x]
//                             (column-major, like po_map) | f64 x[n_pts], y[n_pts]
// prints "ok <0|1> n <states>" and one "x y z k s" line per state.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "path_optimizer/path_optimizer.hpp"
#include "path_optimizer/data_struct/data_struct.hpp"
#include "grid_map_core/grid_map_core.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[3];
    double d[10];
    if (std::fread(hdr, sizeof(int), 3, f) != 3 || std::fread(d, sizeof(double), 10, f) != 10) return 2;
    std::vector<float> dist((size_t)hdr[0] * hdr[1]);
    std::vector<double> x(hdr[2]), y(hdr[2]);
    if (std::fread(dist.data(), sizeof(float), dist.size(), f) != dist.size() || std::fread(x.data(), sizeof(double), x.size(), f) != x.size() ||
        std::fread(y.data(), sizeof(double), y.size(), f) != y.size()) return 2;
    std::fclose(f);
    po_map m{dist.data(), hdr[0], hdr[1], d[0], d[1], d[2]};
    grid_map::GridMap map(m);
    using PathOptimizationNS::State;
    State start(d[3], d[4], d[5], d[6]), goal(d[7], d[8], d[9]);
    std::vector<State> points, path;
    for (int i = 0; i < hdr[2]; ++i) points.emplace_back(x[i], y[i]);
    PathOptimizationNS::PathOptimizer opt(start, goal, map);
    const bool ok = opt.solve(points, &path);
    std::printf("ok %d n %zu\n", ok ? 1 : 0, path.size());
    for (const auto &s : path) std::printf("%.17g %.17g %.17g %.17g %.17g\n", s.x, s.y, s.z, s.k, s.s);
    return 0;
}
