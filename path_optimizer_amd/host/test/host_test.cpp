// Host-side check of the OsqpSolver mirror: reads like the call site in the reference
// (src/path_optimizer/path_optimizer.cpp:182-183).  Needs a GPU; prints a few numbers the pytest wrapper compares
// with the oracle.  Usage: host_test <KP|KPC|K> <N> <B>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "path_optimizer_amd/map_tools.hpp"
#include "path_optimizer_amd/path_optimizer.hpp"
#include "path_optimizer_amd/smoother.hpp"
#include "path_optimizer_amd/solver.hpp"

using namespace PathOptimizationNS;

// host_test bench <KP|KPC|K> <N> <B> [reps] [engines] [corridor half-width]: what the drop-in's caller pays for a batch — OsqpSolver::solveBatch host to host (AoS -> SoA pack, po_solve_batch =
// pinned staging + H2D + solve + D2H, unpack into std::vector<State>) at the setting bench.py quotes `value` at; printed per repetition (SURVEY.md §8d "H2D/D2H reported separately")
static int bench_main(int argc, char **argv) {
    const std::string type = argc > 2 ? argv[2] : "KP";
    const size_t N = argc > 3 ? (size_t)std::atoi(argv[3]) : 200, B = argc > 4 ? (size_t)std::atoi(argv[4]) : 4096;
    const int reps = argc > 5 ? std::atoi(argv[5]) : 5;
    const int formulation = type == "KP" ? PO_KP : (type == "KPC" ? PO_KPC : PO_K);
    // corridor half-width 1.6 +- 0.4 m by default: NARROWER than the soft margin on part of every path, a degenerate and slow case for the refinement (tests/test_newton.py
    // rebuilds this batch in numpy); argv[7] = 2.6 gives BASELINE-like corridors
    const double w0 = argc > 7 ? std::atof(argv[7]) : 1.6;
    std::vector<ReferencePath> refs(B);
    std::vector<VehicleState> vs(B);
    for (size_t b = 0; b < B; ++b) {
        std::vector<State> st;
        std::vector<CoveringCircleBounds> bd;
        std::vector<double> mk, mkp;
        double z = 0.3 * (double)(b % 17), x = 0, y = 0;
        for (size_t i = 0; i < N; ++i) {
            const double s = 0.25 * (double)i, k = 0.04 * std::sin(0.2 * s + (double)(b % 31));
            st.emplace_back(x, y, z, k, s);
            x += std::cos(z) * 0.25; y += std::sin(z) * 0.25; z += k * 0.25;
            CoveringCircleBounds c;
            const double w = w0 + 0.4 * std::sin(0.1 * (double)i + (double)(b % 13));
            c.c0.lb = c.c1.lb = c.c2.lb = c.c3.lb = -w;
            c.c0.ub = c.c1.ub = c.c2.ub = c.c3.ub = w;
            bd.push_back(c);
            mk.push_back(0.4 * 9.8 / 64.0); mkp.push_back(0.1 / 8.0);
        }
        refs[b].setReference(st); refs[b].setBounds(bd); refs[b].setLimits(mk, mkp);
        vs[b] = VehicleState(State(0, 0, st[0].z, st[0].k), State(x, y, st[N - 1].z + 0.02), 0.2 - 0.01 * (double)(b % 29), 0.03);
    }
    std::vector<PlanningInstance> inst(B);
    for (size_t b = 0; b < B; ++b) inst[b] = {&refs[b], &vs[b]};
    po_params p;
    po_default_params(&p);
    p.refine = 2; p.refine_rounds = 5; p.refine_extra_rounds = 2; p.refine_eps = 1e-8; p.refine_chain = 2;  // bench.py HEADLINE
    PoEngine eng(0, &p);
    std::vector<std::vector<State>> paths;
    std::vector<po_info> info;
    for (int r = 0; r < reps + 1; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = OsqpSolver::solveBatch(formulation, inst.data(), B, N, &paths, &info, &eng);
        const double tot = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        float ph[8] = {0};
        po_last_phase_ms(eng.handle(), ph);
        int solved = 0, cert = 0;
        for (const po_info &i : info) { solved += i.status == PO_STATUS_SOLVED; cert += i.status_refine == 1; }
        std::printf("bench rep %d rc=%d B=%zu N=%zu: solveBatch host-to-host %.2f ms = AoS->SoA pack %.2f + po_solve_batch %.2f (staging pack + H2D %.2f, solve %.2f, D2H %.2f, unpack %.2f) + State unpack %.2f; solved %d certified %d%s\n",
                    r, rc, B, N, tot, OsqpSolver::lastShardMs()[0], OsqpSolver::lastShardMs()[1], ph[0], ph[1], ph[2], ph[4], OsqpSolver::lastShardMs()[2], solved, cert, r == 0 ? " (warm-up)" : "");
    }
    // the same batch split over E engines on ONE device (own handle, stream, pinned staging and host thread each; `host_test bench KP 200 4096 5 4`): slice k's H2D, solve and D2H
    // overlap the other slices' — what SURVEY.md §8e asks the multi-device path not to serialise on, measured where there is one device
    const int E = argc > 6 ? std::atoi(argv[6]) : 0;
    if (E > 1) {
        std::vector<std::unique_ptr<PoEngine>> own;
        std::vector<PoEngine *> engs;
        for (int e = 0; e < E; ++e) { own.emplace_back(new PoEngine(0, &p)); engs.push_back(own.back().get()); }
        const int pt = OsqpSolver::packThreads();
        OsqpSolver::packThreads() = std::max(1, pt / E);
        for (int r = 0; r < reps + 1; ++r) {
            std::vector<double> dms;
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = OsqpSolver::solveBatch(formulation, inst.data(), B, N, &paths, &info, engs, &dms);
            const double tot = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            int solved = 0, cert = 0;
            for (const po_info &i : info) { solved += i.status == PO_STATUS_SOLVED; cert += i.status_refine == 1; }
            std::printf("bench rep %d rc=%d B=%zu N=%zu over %d engines on device 0: solveBatch host-to-host %.2f ms; per shard", r, rc, B, N, E, tot);
            for (double d : dms) std::printf(" %.2f", d);
            std::printf("; solved %d certified %d%s\n", solved, cert, r == 0 ? " (warm-up)" : "");
        }
        OsqpSolver::packThreads() = pt;
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "bench") return bench_main(argc, argv);
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
    // 2b) the batch split over every visible device: one host thread + handle + stream per device, contiguous shards, results in place.
    //     G = hipGetDeviceCount(): 1 on a single-GPU box (the split then degenerates to one shard), 8 on a full node.  With PO_HOST_TEST_SHARDS=k the same
    //     device is used through k engines, which exercises the split, the threads and the in-place writes on a single-GPU box.
    {
        std::vector<PoEngine *> engines = PoEngine::allDevices();
        std::vector<std::unique_ptr<PoEngine>> extra;
        const char *ks = std::getenv("PO_HOST_TEST_SHARDS");
        const int want = ks ? std::atoi(ks) : 0;
        while (want > 0 && (int)engines.size() < want) { extra.emplace_back(new PoEngine(0)); engines.push_back(extra.back().get()); }
        std::vector<std::vector<State>> mpaths;
        std::vector<po_info> minfo;
        std::vector<double> dev_ms;
        const int mrc = OsqpSolver::solveBatch(formulation, inst.data(), B, N, &mpaths, &minfo, engines, &dev_ms);
        bool same = mrc == PO_OK && mpaths.size() == paths.size();
        for (size_t b = 0; b < B && same; ++b) {
            same = same && std::memcmp(&minfo[b], &info[b], sizeof(po_info)) == 0 && mpaths[b].size() == paths[b].size();
            for (size_t i = 0; i < N && same; ++i) {
                const State &p = mpaths[b][i], &q = paths[b][i];
                same = same && p.x == q.x && p.y == q.y && p.z == q.z && p.k == q.k && p.s == q.s;
            }
        }
        double mx = 0, sum = 0;
        for (double t : dev_ms) { mx = std::fmax(mx, t); sum += t; }
        std::printf("multi-device rc=%d engines=%zu (visible devices %d) bit_identical=%d device_ms max=%.3f mean=%.3f max_over_mean=%.3f\n", mrc, engines.size(),
                    po_device_count(), (int)same, mx, sum / (double)dev_ms.size(), dev_ms.empty() || sum == 0 ? 0.0 : mx / (sum / (double)dev_ms.size()));
        for (size_t g = 0; g < dev_ms.size(); ++g) std::printf("  device shard %zu: %.3f ms\n", g, dev_ms[g]);
        if (!same) return 7;
    }
    // 3) the stages either side of the solve: bounds from a distance map, collision check of the result
    {
        const int sx = 500, sy = 500;
        const double res = 0.2;
        std::vector<float> dist((size_t)sx * sy);
        const double ox = 12.0, oy = 2.5, orad = 1.0;  // one disc obstacle left of the first reference path
        for (int j = 0; j < sy; ++j)
            for (int i = 0; i < sx; ++i) {
                const double cx = 0.0 + 0.5 * sx * res - (i + 0.5) * res, cy = 0.0 + 0.5 * sy * res - (j + 0.5) * res;
                const double dd = std::sqrt((cx - ox) * (cx - ox) + (cy - oy) * (cy - oy)) - orad;
                dist[(size_t)j * sx + i] = (float)(dd > 0 ? dd : 0);
            }
        Map map(dist.data(), sx, sy, res, 0.0, 0.0);
        std::printf("map d(0,0)=%.6f inside(60,0)=%d\n", map.getObstacleDistance(0, 0), (int)map.isInside(60, 0));
        ReferencePath straight;
        SplineKnots kn;
        std::vector<State> st;
        for (int i = 0; i < 100; ++i) st.emplace_back(0.25 * i - 20.0, -15.0, 0.0, 0.0, 0.25 * i);  // far from the disc: free corridor
        for (int i = 0; i < 30; ++i) { kn.s.push_back(1.0 * i); kn.x.push_back(1.0 * i - 20.0); kn.y.push_back(-15.0); }
        straight.setReference(st);
        updateBounds(straight, kn, map);
        const auto &b0 = straight.getBounds();
        std::printf("bounds n=%zu c0=[%.9f, %.9f] c3=[%.9f, %.9f]\n", b0.size(), b0[10].c0.lb, b0[10].c0.ub, b0[10].c3.lb, b0[10].c3.ub);
        bool bounds_ok = b0.size() == 100;
        for (const auto &c : b0) bounds_ok = bounds_ok && std::fabs(c.c0.ub - 4.9) < 1e-9 && std::fabs(c.c0.lb + 4.9) < 1e-9 && std::fabs(c.c3.ub - 4.9) < 1e-9;
        ReferencePath blocked;  // runs straight through the disc: there the corridor lies entirely on one side of the reference
        st.clear();
        for (int i = 0; i < 100; ++i) st.emplace_back(0.25 * i, 2.5, 0.0, 0.0, 0.25 * i);
        SplineKnots kb;
        for (int i = 0; i < 30; ++i) { kb.s.push_back(1.0 * i); kb.x.push_back(1.0 * i); kb.y.push_back(2.5); }
        blocked.setReference(st);
        updateBounds(blocked, kb, map);
        int one_sided = 0;
        for (const auto &c : blocked.getBounds()) one_sided += (c.c1.lb * c.c1.ub > 0) ? 1 : 0;
        std::printf("path through the disc keeps %zu of 100 states, %d with a one-sided corridor\n", blocked.getSize(), one_sided);
        CollisionChecker cc(map);
        const bool free0 = cc.isSingleStateCollisionFreeImproved(State(-10, -15, 0.3)), hit = !cc.isSingleStateCollisionFreeImproved(State(11.0, 2.5, 0.0));
        std::vector<std::vector<State>> two(2);
        for (int i = 0; i < 120; ++i) { two[0].emplace_back(0.25 * i - 20, -15.0, 0.0, 0.0, 0.25 * i); two[1].emplace_back(0.25 * i - 16.0, 2.5, 0.0, 0.0, 0.25 * i); }
        std::vector<po_info> inf(2); inf[0].status = inf[1].status = PO_STATUS_SOLVED;
        const auto okv = cc.checkPaths(&two, inf);
        std::printf("check free=%d hit=%d kept=%zu,%zu ok=%d,%d\n", (int)free0, (int)hit, two[0].size(), two[1].size(), (int)okv[0], (int)okv[1]);
        if (!(bounds_ok && free0 && hit && two[0].size() == 120 && two[1].size() < 120 && okv[0] && blocked.getSize() == 100 && one_sided > 3 && one_sided < 30)) {
            std::printf("map stages FAILED\n");
            return 4;
        }
        // 4) the reference-smoothing stages: TensionSmoother2 -> graphSearchDp -> postSmooth QP -> re-sampling -> limits
        std::vector<double> xl, yl, al, kl, sl, rx, ry, rs;
        for (int i = 0; i < 40; ++i) {  // a wiggly raw reference along y = -15, sampled every metre like segmentRawReference does
            const double s = 1.0 * i, w = 0.3 * std::sin(0.9 * s), dw = 0.27 * std::cos(0.9 * s), ddw = -0.243 * std::sin(0.9 * s);
            xl.push_back(s - 20.0); yl.push_back(-15.0 + w); al.push_back(std::atan2(dw, 1.0)); kl.push_back(ddw / std::pow(1 + dw * dw, 1.5)); sl.push_back(s);
        }
        TensionSmoother2 ts;
        const bool sm_ok = ts.osqpSmooth(xl, yl, al, kl, sl, &rx, &ry, &rs);
        double rough_in = 0, rough_out = 0;
        for (int i = 1; i + 1 < 40; ++i) { rough_in += std::fabs(yl[i + 1] - 2 * yl[i] + yl[i - 1]); rough_out += std::fabs(ry[i + 1] - 2 * ry[i] + ry[i - 1]); }
        std::printf("tension2 ok=%d iters=%d roughness %.4f -> %.4f start (%.6f, %.6f)\n", (int)sm_ok, ts.lastInfo().iters, rough_in, rough_out, rx[0], ry[0]);
        SplineKnots sk;  // x_spline.set_points(result_s_list, result_x_list) in TensionSmoother::smooth
        sk.s = rs; sk.x = rx; sk.y = ry;
        SearchLayers layers;
        const bool dp_ok = graphSearchDp(sk, rs.back(), State(-20.0, -14.6, 0.05), map, &layers);
        std::vector<double> offsets;
        const bool ps_ok = dp_ok && postSmoothOffsets(layers, &offsets);
        std::printf("search ok=%d layers=%zu l0=%.6f post ok=%d offset[0]=%.6f offset[last]=%.6f\n", (int)dp_ok, layers.s.size(), layers.vehicle_l, (int)ps_ok,
                    offsets.empty() ? 0.0 : offsets.front(), offsets.empty() ? 0.0 : offsets.back());
        ReferencePath resampled;
        const bool rs_ok = buildReferenceFromSpline(&resampled, sk, rs.back(), 0.15, 0.3);
        std::vector<State> withv = resampled.getReferenceStates();
        for (auto &q : withv) { q.v = 8.0; q.a = 1.0; }
        resampled.setReference(withv);
        updateLimits(&resampled);
        const double mk_expect = std::sqrt(0.4 * 9.8 * 0.4 * 9.8 - 1.0) / 64.0;
        std::printf("resample ok=%d n=%zu ds=%.3f max_k=%.9f (expect %.9f) max_kp=%.9f\n", (int)rs_ok, resampled.getSize(),
                    resampled.getSize() > 1 ? resampled.getReferenceStates()[1].s : 0.0, resampled.getMaxKList().empty() ? 0.0 : resampled.getMaxKList()[0], mk_expect,
                    resampled.getMaxKpList().empty() ? 0.0 : resampled.getMaxKpList()[0]);
        const bool stages_ok = sm_ok && rough_out < 0.5 * rough_in && std::fabs(rx[0] - xl[0]) < 1e-3 && std::fabs(ry[0] - yl[0]) < 1e-3 && dp_ok && layers.s.size() >= 20 &&
                               std::fabs(layers.vehicle_l) < 1.0 && ps_ok && std::fabs(offsets.front() - layers.vehicle_l) < 1e-3 && rs_ok && resampled.getSize() > 100 &&
                               std::fabs(resampled.getMaxKList()[0] - mk_expect) < 1e-12 && std::fabs(resampled.getMaxKpList()[0] - 0.1 / 8.0) < 1e-15;
        std::printf("smoothing stages %s\n", stages_ok ? "ok" : "FAILED");
        if (!stages_ok) return 5;
        // 5) the top-level class, used like path_optimizer_benchmark.cpp:84-100: PathOptimizer(start, end, map).solve(points, &result)
        std::vector<State> pts;
        for (int i = 0; i < 20; ++i) pts.emplace_back(3.0 * i - 30.0, -15.0 + 0.3 * std::sin(1.3 * i), 0.0);
        State start_state(-30.0, -15.0, 0.0, 0.0), end_state(27.0, pts.back().y, 0.0);
        PathOptimizer path_optimizer(start_state, end_state, map);
        std::vector<State> result;
        const bool po_ok = path_optimizer.solve(pts, &result);
        std::printf("PathOptimizer::solve ok=%d stage=%d n=%zu end=(%.4f, %.4f) s_end=%.4f\n", (int)po_ok, path_optimizer.lastStage(), result.size(),
                    result.empty() ? 0.0 : result.back().x, result.empty() ? 0.0 : result.back().y, result.empty() ? 0.0 : result.back().s);
        std::vector<State> again;
        SplineKnots kk;
        for (int i = 0; i < 20; ++i) { kk.s.push_back(3.0 * i); kk.x.push_back(3.0 * i - 30.0); kk.y.push_back(-15.0); }
        std::vector<State> direct;
        for (int i = 0; i < 150; ++i) direct.emplace_back(0.3 * i - 30.0, -15.0, 0.0, 0.0, 0.3 * i);
        const bool ws_ok = path_optimizer.solveWithoutSmoothing(direct, kk, &again);
        std::printf("solveWithoutSmoothing ok=%d n=%zu y[50]=%.6f\n", (int)ws_ok, again.size(), again.size() > 50 ? again[50].y : 0.0);
        std::vector<State> none;
        const bool empty_ok = path_optimizer.solve(std::vector<State>(), &none);
        const bool top_ok = po_ok && result.size() > 100 && std::fabs(result.back().x - 27.0) < 1.5 && std::fabs(result.back().y + 15.0) < 1.0 && ws_ok && again.size() == 150 &&
                            std::fabs(again[50].y + 15.0) < 1e-6 && !empty_ok;
        std::printf("top-level %s\n", top_ok ? "ok" : "FAILED");
        if (!top_ok) return 6;
    }
    std::string bad = "KCP";
    std::printf("create(KCP)=%s\n", OsqpSolver::create(bad, refs[0], vs[0], N) ? "object" : "nullptr");
    return ok && dmax < 1e-12 ? 0 : 1;
}
