// Host-side check of the OsqpSolver mirror: reads like the call site in the reference
// (src/path_optimizer/path_optimizer.cpp:182-183).  Needs a GPU; prints a few numbers the pytest wrapper compares
// with the oracle.  Usage: host_test <KP|KPC|K> <N> <B>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "path_optimizer_amd/solver.hpp"

using namespace PathOptimizationNS;

int main(int argc, char **argv) {
    std::string type = argc > 1 ? argv[1] : "KP";
    const size_t N = argc > 2 ? (size_t)std::atoi(argv[2]) : 60, B = argc > 3 ? (size_t)std::atoi(argv[3]) : 3;
    std::vector<ReferencePath> refs(B);
    std::vector<VehicleState> vs(B);
    for (size_t b = 0; b < B; ++b) {  // deterministic toy instances: sinusoidal curvature, +-(1.6..2.0) m corridor
        std::vector<State> st;
        std::vector<CoveringCircleBounds> bd;
        std::vector<double> mk, mkp;
        double z = 0.3 * (double)b, x = 0, y = 0;
        for (size_t i = 0; i < N; ++i) {
            const double s = 0.25 * (double)i, k = 0.04 * std::sin(0.2 * s + (double)b);
            st.emplace_back(x, y, z, k, s);
            x += std::cos(z) * 0.25; y += std::sin(z) * 0.25; z += k * 0.25;
            CoveringCircleBounds c;
            const double w = 1.6 + 0.4 * std::sin(0.1 * (double)i + (double)b);
            c.c0.lb = c.c1.lb = c.c2.lb = c.c3.lb = -w;
            c.c0.ub = c.c1.ub = c.c2.ub = c.c3.ub = w;
            bd.push_back(c);
            mk.push_back(0.4 * 9.8 / 64.0); mkp.push_back(0.1 / 8.0);
        }
        refs[b].setReference(st);
        refs[b].setBounds(bd);
        refs[b].setLimits(mk, mkp);
        vs[b] = VehicleState(State(0, 0, st[0].z, st[0].k), State(x, y, st[N - 1].z + 0.02), 0.2 - 0.1 * (double)b, 0.03);
    }
    // 1) the reference's call pattern, one path
    auto solver = OsqpSolver::create(type, refs[0], vs[0], N);
    if (!solver) { std::printf("create: nullptr\n"); return 2; }
    std::vector<State> path;
    const bool ok = solver->solve(&path);
    std::printf("single ok=%d n=%zu x[last]=%.12f y[last]=%.12f s[last]=%.12f\n", (int)ok, path.size(), path.empty() ? 0.0 : path.back().x,
                path.empty() ? 0.0 : path.back().y, path.empty() ? 0.0 : path.back().s);
    // 2) the batched entry gives the same first path
    std::vector<PlanningInstance> inst;
    for (size_t b = 0; b < B; ++b) inst.push_back({&refs[b], &vs[b]});
    std::vector<std::vector<State>> paths;
    std::vector<po_info> info;
    const int formulation = type == "K" ? PO_K : (type == "KP" ? PO_KP : PO_KPC);
    const int rc = OsqpSolver::solveBatch(formulation, inst.data(), B, N, &paths, &info);
    std::printf("batch rc=%d\n", rc);
    if (rc) return 3;
    double dmax = 0;
    for (size_t i = 0; i < N && ok; ++i) dmax = std::fmax(dmax, std::fabs(paths[0][i].x - path[i].x) + std::fabs(paths[0][i].y - path[i].y));
    std::printf("batch0_vs_single=%.3e\n", dmax);
    for (size_t b = 0; b < B; ++b) std::printf("path %zu status=%d iters=%d rho=%.6f s_end=%.9f\n", b, info[b].status, info[b].iters, info[b].rho, paths[b].back().s);
    std::string bad = "KCP";
    std::printf("create(KCP)=%s\n", OsqpSolver::create(bad, refs[0], vs[0], N) ? "object" : "nullptr");
    return ok && dmax < 1e-12 ? 0 : 1;
}
