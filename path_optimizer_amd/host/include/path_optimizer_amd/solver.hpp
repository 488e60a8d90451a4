// Host-side mirror of the reference's solver interface, backed by libpo_hip.so (C ABI, include/po_hip.h).
//
//   reference                                                    here
//   OsqpSolver::create(type, ref, vehicle, horizon)              same signature, same "K"/"KP"/"KPC" strings (solver.cpp:30-44)
//   virtual bool OsqpSolver::solve(std::vector<State>*)          same: true iff the QP status is `solved` (solver.cpp:46-77)
//   (none: one path per call)                                    OsqpSolver::solveBatch(): many independent planning instances, on one device or — given
//                                                                several engines — split over the GPUs of the node (contiguous shards,
//                                                                one host thread + one handle + one stream per device, no collective)
//
// Differences, on purpose: an unknown type still yields nullptr like the reference, but solveBatch
// reports it as an error instead of silently producing an empty "successful" path; parameters are captured in a
// po_params block at construction instead of being read from gflags globals during assembly.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
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
    // One engine per visible HIP device (device 0 first), created once per process with the default parameters: what the multi-device solveBatch takes.
    // Thread-safe (function-local static, initialised exactly once); an EMPTY vector means no HIP device is visible — solveBatch reports that as its own error.
    static const std::vector<PoEngine *> &allDevices() {
        struct All {
            std::vector<std::unique_ptr<PoEngine>> own;
            std::vector<PoEngine *> view;
            All() {
                const int n = po_device_count();
                for (int d = 0; d < n; ++d) { own.emplace_back(new PoEngine(d)); view.push_back(own.back().get()); }
            }
        };
        static const All all;
        return all.view;
    }
    // Grow-only SoA staging of this engine's batches (solveBatch): a fresh 85 MB allocation per call costs more in first-touch page faults than the pack itself.
    // A handle runs one batch at a time (po_solve_batch is not re-entrant on one handle), so callers hold mutex() from the pack to the unpack.
    std::mutex &mutex() { return mu_; }
    double *stagingIn(size_t n) { return grow(in_, in_cap_, n); }
    double *stagingOut(size_t n) { return grow(out_, out_cap_, n); }
 private:
    static double *grow(std::unique_ptr<double[]> &b, size_t &cap, size_t n) {
        if (n > cap) { b.reset(new double[n]); cap = n; }  // (uninitialised: every element a caller reads was written by its pack / by po_solve_batch)
        return b.get();
    }
    po_handle h_{};
    po_params params_{};
    std::mutex mu_;
    std::unique_ptr<double[]> in_, out_;
    size_t in_cap_ = 0, out_cap_ = 0;
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
        paths->assign(B, {});
        info->assign(B, po_info{});
        int keep = 0;
        return B == 0 ? PO_OK : solveShard(formulation, inst, 0, B, horizon, paths, info->data(), eng, &keep);
    }

    // New: the same batch split over several devices.  Engine g of G solves the contiguous shard [g B / G, (g + 1) B / G) (the first B % G
    // shards one path longer: the split of path_optimizer_amd/shard.py and bench.py) on its own host thread, handle and stream; paths are independent QPs, so
    // there is no exchange between devices and the results — states and po_info, written in place — are bit-identical to the single-engine call.
    // device_ms (optional): wall time of each shard's pack + H2D + solve + D2H, for the load-imbalance figure max / mean.
    // One batch = one (horizon, keep_control_steps_): PO_ERR_INVALID otherwise, as for one engine.  The first error of any shard is returned.
    static int solveBatch(int formulation, const PlanningInstance *inst, size_t B, size_t horizon, std::vector<std::vector<State>> *paths,
                          std::vector<po_info> *info, const std::vector<PoEngine *> &engines, std::vector<double> *device_ms = nullptr) {
        if (engines.empty()) return PO_ERR_HIP;  // (PoEngine::allDevices() with no HIP device visible: not a malformed call)
        if (!inst || !paths || !info || horizon < 2) return PO_ERR_INVALID;
        for (PoEngine *e : engines) if (!e) return PO_ERR_INVALID;
        const size_t G = engines.size();
        paths->assign(B, {});
        info->assign(B, po_info{});
        if (device_ms) device_ms->assign(G, 0.0);
        if (B == 0) return PO_OK;
        std::vector<int> rc(G, PO_OK), keep(G, 0);
        std::vector<size_t> lo(G), hi(G);
        for (size_t g = 0; g < G; ++g) { const size_t base = B / G, rem = B % G; lo[g] = g * base + (g < rem ? g : rem); hi[g] = lo[g] + base + (g < rem ? 1 : 0); }
        auto work = [&](size_t g) {
            const auto t0 = std::chrono::steady_clock::now();
            if (hi[g] > lo[g]) rc[g] = solveShard(formulation, inst, lo[g], hi[g], horizon, paths, info->data(), engines[g], &keep[g]);
            if (device_ms) (*device_ms)[g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        };
        std::vector<std::thread> th;
        for (size_t g = 1; g < G; ++g) th.emplace_back(work, g);
        work(0);
        for (auto &t : th) t.join();
        int k0 = 0;
        for (size_t g = 0; g < G; ++g) {
            if (rc[g] != PO_OK) return rc[g];
            if (hi[g] > lo[g]) { if (k0 == 0) k0 = keep[g]; else if (keep[g] != k0) return PO_ERR_INVALID; }  // one batch = one keep_control_steps_
        }
        return PO_OK;
    }

 private:
    // paths [lo, hi) of the batch on one engine; results written in place ((*paths)[b], info[b])
    static int solveShard(int formulation, const PlanningInstance *inst, size_t lo, size_t hi, size_t horizon, std::vector<std::vector<State>> *paths,
                          po_info *info, PoEngine *eng, int *keep_out) {
        const size_t N = horizon, B = hi - lo;
        std::lock_guard<std::mutex> engine_lock(eng->mutex());
        const auto tp0 = std::chrono::steady_clock::now();
        double *const buf = eng->stagingIn(B * N * (13 + (formulation == PO_KPC ? 2 : 0)) + B * 4);
        double *rx = buf, *ry = rx + B * N, *rz = ry + B * N, *rk = rz + B * N, *rs = rk + B * N, *bd = rs + B * N, *x0 = bd + B * N * 8, *gz = x0 + B * 3;
        double *mk = gz + B, *mkp = mk + (formulation == PO_KPC ? B * N : 0);
        // AoS -> SoA pack on several host threads: contiguous slices of the shard (the multi-device caller already runs one thread per device)
        const size_t nthr = std::max<size_t>(1, std::min<size_t>({(size_t)packThreads(), B / 64 + 1, (size_t)16}));
        std::vector<int> trc(nthr, PO_OK), tkeep(nthr, 0);
        auto pack = [&](size_t t) {
            const size_t b0 = B * t / nthr, b1 = B * (t + 1) / nthr;
            int keep_t = 0;
            for (size_t b = b0; b < b1; ++b) {
                const PlanningInstance &pi = inst[lo + b];
                if (!pi.reference_path || !pi.vehicle_state) { trc[t] = PO_ERR_INVALID; return; }
                const auto &st = pi.reference_path->getReferenceStates();
                const auto &bnd = pi.reference_path->getBounds();
                if (st.size() < N || bnd.size() < N) { trc[t] = PO_ERR_INVALID; return; }
                for (size_t i = 0; i < N; ++i) {
                    const size_t o = b * N + i;
                    rx[o] = st[i].x; ry[o] = st[i].y; rz[o] = st[i].z; rk[o] = st[i].k; rs[o] = st[i].s;
                    const CoveringCircleBounds::SingleCircleBounds *c[4] = {&bnd[i].c0, &bnd[i].c1, &bnd[i].c2, &bnd[i].c3};
                    for (int j = 0; j < 4; ++j) { bd[o * 8 + 2 * j] = c[j]->lb; bd[o * 8 + 2 * j + 1] = c[j]->ub; }
                }
                if (formulation == PO_KPC) {
                    const auto &a = pi.reference_path->getMaxKList();
                    const auto &c = pi.reference_path->getMaxKpList();
                    if (a.size() < N || c.size() < N) { trc[t] = PO_ERR_INVALID; return; }
                    for (size_t i = 0; i < N; ++i) { mk[b * N + i] = a[i]; mkp[b * N + i] = c[i]; }
                }
                const auto e = pi.vehicle_state->getInitError();
                x0[b * 3] = e[0]; x0[b * 3 + 1] = e[1]; x0[b * 3 + 2] = pi.vehicle_state->getStartState().k;
                gz[b] = pi.vehicle_state->getEndState().z;
                const int kb = po_keep_control_steps(formulation, &rs[b * N], (int)N);  // solver.cpp:22-27 + solver_kp_as_input.cpp:17
                if (kb < 0) { trc[t] = kb; return; }
                if (b == b0) keep_t = kb;
                else if (kb != keep_t) { trc[t] = PO_ERR_INVALID; return; }  // one batch = one (N, keep)
            }
            tkeep[t] = keep_t;
        };
        runThreads(nthr, pack);
        int keep = 0;
        for (size_t t = 0; t < nthr; ++t) {
            if (trc[t] != PO_OK) return trc[t];
            if (B * t / nthr == B * (t + 1) / nthr) continue;  // (empty slice)
            if (keep == 0) keep = tkeep[t];
            else if (tkeep[t] != keep) return PO_ERR_INVALID;
        }
        *keep_out = keep;
        const auto tp1 = std::chrono::steady_clock::now();
        int n, m, C;
        int rc = po_problem_dims(formulation, (int)N, keep, &n, &m, &C);
        if (rc) return rc;
        po_batch_in in{formulation, (int)B, (int)N, keep, rx, ry, rz, rk, rs, bd, x0, gz, formulation == PO_KPC ? mk : nullptr, formulation == PO_KPC ? mkp : nullptr, nullptr, nullptr};
        double *const states = eng->stagingOut(B * N * 5);
        po_batch_out out{states, info + lo, nullptr};
        rc = po_solve_batch(eng->handle(), &in, &out);
        if (rc) return rc;
        const auto tp2 = std::chrono::steady_clock::now();
        auto unpack = [&](size_t t) {
            for (size_t b = B * t / nthr; b < B * (t + 1) / nthr; ++b) {
                auto &p = (*paths)[lo + b];
                p.clear();
                p.reserve(N);
                for (size_t i = 0; i < N; ++i) {
                    const double *s = &states[(b * N + i) * 5];
                    p.emplace_back(s[0], s[1], s[2], s[3], s[4]);  // v = a = 0, like getOptimizedPath
                }
            }
        };
        runThreads(nthr, unpack);
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        lastShardMs()[0] = ms(tp0, tp1); lastShardMs()[1] = ms(tp1, tp2); lastShardMs()[2] = ms(tp2, tp3);
        return PO_OK;
    }

 public:
    // host threads of the AoS <-> SoA pack / unpack of one shard (default: min(16, half the hardware threads); 1 = the caller's thread only)
    static int &packThreads() { static int n = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2)); return n; }
    // what the calling thread's last shard spent where, ms: AoS -> SoA pack, po_solve_batch (pinned staging + H2D + solve + D2H: po_last_phase_ms splits it), unpack into State vectors
    static double *lastShardMs() { static thread_local double v[3] = {0, 0, 0}; return v; }

 private:
    template <class Fn> static void runThreads(size_t n, Fn fn) {
        std::vector<std::thread> th;
        for (size_t t = 1; t < n; ++t) th.emplace_back(fn, t);
        fn(0);
        for (auto &x : th) x.join();
    }

 protected:
    const int formulation_;
    const size_t horizon_{};
    const ReferencePath &reference_path_;
    const VehicleState &vehicle_state_;
    PoEngine *engine_;
};

}  // namespace PathOptimizationNS
