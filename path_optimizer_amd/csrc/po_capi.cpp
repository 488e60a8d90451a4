// po_capi.cpp — the C ABI of libpo_hip.so (include/po_hip.h): handle management, host<->device staging,
// kernel launches.  No torch types, no oracle, no CPU fallback: every entry point that computes runs the HIP
// kernels or returns an error code.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/po_hip.h"
#include "po_device.hpp"
#include "po_map.hpp"
#include "po_smooth.hpp"

extern "C" hipError_t po_launch_solve(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out);
extern "C" hipError_t po_launch_finalize_status(po_info *info, int B, hipStream_t st);
extern "C" hipError_t po_launch_mark_unavailable(po_info *info, int B, int refine, int polish, hipStream_t st);
extern "C" hipError_t po_launch_polish(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st);
extern "C" hipError_t po_launch_newton(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st);
extern "C" hipError_t po_launch_newton_fallback(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st);
extern "C" int po_polish_state_doubles(int form, int N, int C, int keep);
extern "C" int po_newton_park_doubles(int form, int N, int C, int keep);
extern "C" int po_shape_threads(int form, int N, int C, int keep);
extern "C" hipError_t po_launch_nw_sort(const int *keys, int B, int *list, hipStream_t st);
extern "C" int po_has_polish_kernel(int form, int N, int C, int keep);
extern "C" hipError_t po_launch_scale(int form, const po::DevBatch *in, const po::DevParams *P, int passes, double *sc, hipStream_t st);
extern "C" hipError_t po_launch_assemble(int form, const po::DevBatch *in, const po::DevParams *P, double *l, double *u, double *dyn, hipStream_t st);
extern "C" size_t po_lds_bytes(int form, int N, int C, int keep);
extern "C" hipError_t po_launch_postcheck(const po::DevMap *m, const po::DevCar *c, int B, int N, const int *n_points, const double *states,
                                          const po_info *info, int *n_valid, int *ok, hipStream_t st);
extern "C" hipError_t po_launch_densify(const po::DevMap *m, const po::DevCar *c, int B, int N, const int *n_points, const double *states, const po_info *info,
                                        double spacing, int M, double *out, int *n_out, int *ok, hipStream_t st);
extern "C" hipError_t po_launch_bounds(const po::DevMap *m, const po::DevBounds *in, double *bounds, int *n_valid, hipStream_t st);
extern "C" hipError_t po_launch_smooth(const po::DevSmooth *a, hipStream_t st);
extern "C" size_t po_smooth_lds_bytes(int kind, int P);
extern "C" size_t po_smooth_scratch_doubles(int kind, int P);
extern "C" hipError_t po_launch_resample(const po::DevSpline *in, const po::DevResample *r, hipStream_t st);
extern "C" hipError_t po_launch_limits(int B, int N, const int *n_points, const double *v, const double *a, double *max_k, double *max_kp, double mu, double rate, hipStream_t st);
extern "C" hipError_t po_launch_dp_search(const po::DevMap *m, const po::DevSpline *in, const po::DevSearch *q, int one_wave, hipStream_t st);
extern "C" size_t po_dp_lds_bytes(int K, int L);
extern "C" size_t po_spline_lds_bytes(int K);
extern "C" hipError_t po_launch_map_sample(const po::DevMap *m, int n, const double *xy, double *dist, int *inside, hipStream_t st);

namespace {
thread_local std::string g_hip_err;
bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIP_TRY(x)                                   \
    do {                                             \
        if (!hip_ok((x), #x)) return PO_ERR_HIP;     \
    } while (0)

struct DevBuf {  // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PO_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (!hip_ok(hipMalloc(&p, bytes), "hipMalloc")) return PO_ERR_NOMEM;
        cap = bytes;
        return PO_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
struct HostBuf {  // grow-only PINNED host buffer (hipHostMalloc): the staging area of the host-pointer entry — DMA engines read / write it directly,
                  // so the H2D / D2H copies run at PCIe speed and asynchronously (a copy from pageable memory is staged by the runtime, synchronously)
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PO_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        if (!hip_ok(hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc")) return PO_ERR_NOMEM;
        cap = bytes;
        return PO_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};
struct CopySeg { char *dst; const char *src; size_t bytes; };
// memcpy of a list of segments on `nthr` host threads (pieces of <= 2 MB handed out round-robin: every thread streams through every array)
void parallel_copy(const std::vector<CopySeg> &segs, int nthr) {
    std::vector<CopySeg> pieces;
    const size_t kPiece = (size_t)2 << 20;
    for (const CopySeg &sg : segs)
        for (size_t o = 0; o < sg.bytes; o += kPiece) pieces.push_back({sg.dst + o, sg.src + o, std::min(kPiece, sg.bytes - o)});
    if (nthr <= 1 || pieces.size() <= 1) {
        for (const CopySeg &pc : pieces) std::memcpy(pc.dst, pc.src, pc.bytes);
        return;
    }
    nthr = (int)std::min<size_t>((size_t)nthr, pieces.size());
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; ++t)
        th.emplace_back([&pieces, t, nthr] { for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)nthr) std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes); });
    for (size_t i = 0; i < pieces.size(); i += (size_t)nthr) std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes);
    for (std::thread &t : th) t.join();
}
}  // namespace

struct po_handle_s {
    int device = 0;
    po_params params{};
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evh[4] = {nullptr, nullptr, nullptr, nullptr};  // host-pointer entry: start, H2D done, (ev0 .. ev1 = the solve), D2H done; evh[3]: solve phase mark (po_last_phase_ms)
    hipEvent_t evp[2] = {nullptr, nullptr};                    // split scheduling (refine = 2): end of the warm-start launches, end of the Newton launch
    bool timed = false, timed_host = false, timed_phases = false;
    double host_pack_ms = 0.0, host_unpack_ms = 0.0;
    HostBuf pin_in, pin_out;   // pinned staging of the host-pointer entry
    int host_threads = 0;      // pack / unpack threads (0: min(8, hardware threads); po_debug_set "host_threads")
    DevBuf pol_buf;  // per-lane ADMM state handed from the solve kernels to newton_kernel / polish_kernel (po_params.refine / polish)
    DevBuf fb_buf;   // refine = 2: the work list of newton_fallback_kernel
    DevBuf nw_state_buf, nw_idx_buf;  // sliced Newton launches: the parked paths' blocks; keys [B] + list [B + 1]
    bool nw_slice_forced = false;
    int nw_last_B = 0;  // ... and the batch size of the last sliced solve (po_debug_get "newton_parked")
    int wave_slots = 1024;  // paths the device runs at a time (one wave per SIMD: 4 per CU); batches below two rounds of that are not sliced (no queueing tail to remove)
    int nw_slice = 8;  // steps of the first of the two Newton launches (po_debug_set "newton_slice"; 0: one launch).  Scheduling only.
    HostBuf fb_host; // ... and the pinned word its count is read back into (refine_chain = 2)
    // developer switches (po_debug_set; the library reads no environment variable): identity_order (block i solves path i), debug_cycles (per-phase shader
    // clocks of path 0 on stderr; synchronises), smoothing / DP-search A/B switches
    bool env_identity = false, env_cycles = false, env_smooth_seq = false, env_smooth_nopad = false, env_smooth_debug = false, env_dp_one_wave = false;
    int env_smooth_waves = 0;
    DevBuf in_buf, out_buf, asm_buf, scale_buf, dbg_buf, map_buf, post_buf, coef_buf, bnd_buf, smooth_buf, smooth_io, plan_coef, plan_io, plan_arena, plan_host;
    po::DevMap map{};  // obstacle-distance layer (po_set_map); map.d == nullptr until set
    std::mutex mu;
    std::mutex plan_mu;  // held for a whole po_plan_batch* call: its stages share the plan arena
};

extern "C" {

void po_default_params(po_params *p) {
    if (!p) return;
    // car geometry and weights: /root/reference/src/config/planning_flags.cpp:8-14,18-43,102-119
    const double car_length = 4.9, rear_axle_to_center = 1.45;
    std::memset(p, 0, sizeof(*p));
    p->d[0] = -3.0 / 8.0 * car_length + rear_axle_to_center;
    p->d[1] = -1.0 / 8.0 * car_length + rear_axle_to_center;
    p->d[2] = 1.0 / 8.0 * car_length + rear_axle_to_center;
    p->d[3] = 3.0 / 8.0 * car_length + rear_axle_to_center;
    p->w_curv = 10; p->w_curv_rate = 200; p->w_dev = 0; p->w_slack = 3;
    p->k_w_curv = 50; p->k_w_curv_rate = 200; p->k_w_dev = 0;
    p->w_k_slack = 500; p->w_kp_slack = 25000;
    p->margin = 1.3;
    p->max_steer = 30.0 * M_PI / 180.0;
    p->wheel_base = 2.85;
    p->enable_collision_check = 1;
    p->car_width = 2.0; p->car_length = 4.9; p->rear_axle_to_center = 1.45; p->safety_margin = 0.0;  /* planning_flags.cpp:18-29 */
    p->constraint_end_heading = 1;
    p->scaling = 10;  // OSQP default (the reference leaves it untouched)
    p->eps_abs = 1e-4; p->eps_rel = 1e-4; p->eps_prim_inf = 1e-4; p->eps_dual_inf = 1e-4;
    p->rho0 = 0.1; p->sigma = 1e-6; p->alpha = 1.6; p->adapt_tol = 5.0;
    p->max_iter = 4000; p->check_every = 25; p->adapt_every = 100;
    // reference-smoothing QPs, planning_flags.cpp:76-86
    p->t2_w_dev = 0.005; p->t2_w_curv = 1; p->t2_w_curv_rate = 10;
    p->cart_w_curv = 1; p->cart_w_curv_rate = 50; p->cart_w_dev = 0.0;
    /* planning_flags.cpp:41-43,57-63,137 */
    p->mu = 0.4; p->max_curvature_rate = 0.1; p->search_lateral_range = 10.0; p->search_long_spacing = 1.5; p->search_lat_spacing = 0.6;
    p->enable_dynamic_segmentation = 1;
    p->enable_raw_output = 1; p->output_spacing = 0.3; /* planning_flags.cpp:127-129 */
    p->polish = 0; p->polish_delta = 1e-6; p->polish_refine_iter = 3;  /* OSQP defaults (polish off) */
    p->refine = 0; p->refine_eps = 1e-7; p->refine_rounds = 1; p->refine_chain = 2; p->refine_extra_rounds = 0;
    p->refine_newton_rho = 100.0; p->refine_newton_rho_eq = 1e4; p->refine_newton_rho_max = 1e5; p->refine_ls_tol = 0.6; p->refine_ls_max = 30; p->refine_newton_max = 300; p->refine_newton_final = 3; p->refine_newton_escalate = 12; p->refine_newton_rho_eq_max = 1e6; /* refine = 2 */
}

int po_problem_dims(int form, int N, int keep, int *n, int *m, int *C) {
    if (N < 2) return PO_ERR_INVALID;
    int c, nn, mm;
    if (form == PO_KP) {  // solver_kp_as_input.cpp:13-24
        if (keep < 1) return PO_ERR_INVALID;
        c = (N + keep - 2) / keep; nn = 5 * N + c; mm = 11 * N + c + 2;
    } else if (form == PO_KPC) {  // solver_kp_as_input_constrained.cpp:13-24
        if (keep != 4) return PO_ERR_INVALID;
        c = (N + keep - 2) / keep; nn = 6 * N + c; mm = 12 * N + 3 * c + 2;
    } else if (form == PO_K) {  // solver_k_as_input.cpp:14-20
        c = N - 1; nn = 4 * N - 1; mm = 11 * N - 1;
    } else {
        return PO_ERR_INVALID;
    }
    if (n) *n = nn;
    if (m) *m = mm;
    if (C) *C = c;
    return PO_OK;
}

int po_keep_control_steps(int form, const double *ref_s, int N) {
    if (!ref_s || N < 2) return PO_ERR_INVALID;
    if (form == PO_KPC) return 4;
    if (form == PO_K) return 1;
    if (form != PO_KP) return PO_ERR_INVALID;
    double interval = 0;  // solver.cpp:19,22-27
    for (int i = 1; i < N && i < 10; ++i) interval = std::fmax(interval, ref_s[i] - ref_s[i - 1]);
    const double q = 1.2 / interval;  // solver_kp_as_input.cpp:17 (truncating cast)
    const int k = (q >= 2147483647.0 || q != q) ? 1 : static_cast<int>(q);
    return k > 1 ? k : 1;
}

// Several engines in ONE process (OsqpSolver::solveBatch(..., engines), one host thread + handle + stream each): the HIP runtime maps a process's streams onto a pool of
// GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a hardware queue run their kernels one after the other — measured on one MI355X with E engines
// solving 64-path batches side by side (tools/engines_procs.py, round 6): per call 1.21 / 1.22 / 1.82 / 4.24 ms for E = 1 / 2 / 4 / 8 with the default pool, 1.23 / 2.39 /
// 4.77 / 7.16 with a pool of 2, 1.22 / 1.23 / 1.36 / 1.53 with a pool of 16; E PROCESSES (a pool each): 1.22 / 1.46 / 1.66 / 1.88.  The pool size is read once, when the
// runtime starts; so, BEFORE this library's first HIP call, the variable is set to 16 unless the caller has set it (overwrite = 0; a process whose runtime is already up —
// e.g. one that imported torch first — keeps what it started with: export GPU_MAX_HW_QUEUES there).
static void runtime_prepare() {
    static std::once_flag once;
    std::call_once(once, [] { setenv("GPU_MAX_HW_QUEUES", "16", 0); });
}

int po_device_count(void) {
    runtime_prepare();
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int po_create(int device, const po_params *params, po_handle *out) {
    if (!params || !out) return PO_ERR_INVALID;
    if (params->scaling < 0 || params->scaling > 100) return PO_ERR_INVALID;
    if (params->refine != 0 && params->refine != 2) return PO_ERR_INVALID;  // (refine = 1 was removed with ABI 5, include/po_hip.h)
    // the rounds of the refinement (refine_rounds regular ones + refine_extra_rounds below eps) are counted in 5 bits of the hand-back status: 32 or more used to switch the
    // refinement off silently — refused here instead
    if (params->refine == 2 && (params->refine_rounds > 1 ? params->refine_rounds : 1) + (params->refine_extra_rounds > 0 ? params->refine_extra_rounds : 0) >= 32) return PO_ERR_INVALID;
    runtime_prepare();
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return PO_ERR_INVALID;
    HIP_TRY(hipSetDevice(device));
    po_handle_s *h = new (std::nothrow) po_handle_s;
    if (!h) return PO_ERR_NOMEM;
    h->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->wave_slots = 4 * cus; }
    h->params = *params;
    if (!hip_ok(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(hipEventCreate(&h->ev0), "hipEventCreate") || !hip_ok(hipEventCreate(&h->ev1), "hipEventCreate")) {
        delete h;
        return PO_ERR_HIP;
    }
    for (hipEvent_t &e : h->evh) if (!hip_ok(hipEventCreate(&e), "hipEventCreate")) { delete h; return PO_ERR_HIP; }
    for (hipEvent_t &e : h->evp) if (!hip_ok(hipEventCreate(&e), "hipEventCreate")) { delete h; return PO_ERR_HIP; }
    h->stream = h->own_stream;
    *out = h;
    return PO_OK;
}

int po_destroy(po_handle h) {
    if (!h) return PO_ERR_INVALID;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->pol_buf.release();
    h->fb_buf.release();
    h->nw_state_buf.release(); h->nw_idx_buf.release();
    h->in_buf.release(); h->out_buf.release(); h->asm_buf.release(); h->scale_buf.release(); h->dbg_buf.release(); h->map_buf.release(); h->post_buf.release(); h->coef_buf.release(); h->bnd_buf.release(); h->smooth_buf.release(); h->smooth_io.release(); h->plan_coef.release(); h->plan_io.release(); h->plan_arena.release(); h->plan_host.release();
    h->pin_in.release(); h->pin_out.release(); h->fb_host.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->evh) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->evp) if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return PO_OK;
}

int po_debug_set(po_handle h, const char *key, int value) {
    if (!h || !key) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    const std::string k(key);
    if (k == "identity_order") h->env_identity = value != 0;
    else if (k == "host_threads") h->host_threads = value < 0 ? 0 : value;
    else if (k == "debug_cycles") h->env_cycles = value != 0;
    else if (k == "newton_slice") { h->nw_slice = value > 0 ? (int)value : 0; h->nw_slice_forced = value > 0; }  // (set explicitly: also on batches the engine would not slice)
    else if (k == "smooth_seq") h->env_smooth_seq = value != 0;
    else if (k == "smooth_waves") h->env_smooth_waves = value;
    else if (k == "smooth_nopad") h->env_smooth_nopad = value != 0;
    else if (k == "smooth_debug") h->env_smooth_debug = value != 0;
    else if (k == "dp_one_wave") h->env_dp_one_wave = value != 0;
    else return PO_ERR_INVALID;
    return PO_OK;
}

int po_debug_get(po_handle h, const char *key, long long *value) {
    if (!h || !key || !value) return PO_ERR_INVALID;
    const std::string k(key);
    std::lock_guard<std::mutex> g(h->mu);
    if (k == "fallback_paths") {  // split scheduling of refine = 2: how many paths the last solve's Newton launch handed to the fallback launch
        *value = 0;
        if (!h->fb_buf.p) return PO_OK;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
        int c = 0;
        HIP_TRY(hipMemcpy(&c, h->fb_buf.p, sizeof(int), hipMemcpyDeviceToHost));
        *value = c;
        return PO_OK;
    }
    if (k == "newton_parked") {  // sliced Newton launches: how many paths of the last solve went on into the second launch (-1: the last solve was not sliced)
        *value = -1;
        if (!h->nw_idx_buf.p || h->nw_last_B <= 0) return PO_OK;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
        int c = 0;
        HIP_TRY(hipMemcpy(&c, static_cast<int *>(h->nw_idx_buf.p) + h->nw_last_B, sizeof(int), hipMemcpyDeviceToHost));
        *value = c;
        return PO_OK;
    }
    if (k == "newton_list_ok") {  // sliced Newton launches: is the second launch's list what nw_sort_kernel promises — every parked path exactly once, keys non-increasing, ties in path
        // order?  1 yes, 0 no, -1 the last solve was not sliced.  (The results of a solve do not depend on the list's ORDER, so only this check sees an ordering bug.)
        *value = -1;
        if (!h->nw_idx_buf.p || h->nw_last_B <= 0) return PO_OK;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const int B = h->nw_last_B;
        std::vector<int> kl(2 * (size_t)B + 1);
        HIP_TRY(hipMemcpy(kl.data(), h->nw_idx_buf.p, sizeof(int) * kl.size(), hipMemcpyDeviceToHost));
        const int *keys = kl.data(), *list = kl.data() + B;
        const int n = list[0];
        int parked = 0;
        for (int b = 0; b < B; ++b) parked += keys[b] >= 0;
        bool ok = n == parked;
        std::vector<char> seen((size_t)B, 0);
        for (int i = 0; ok && i < n; ++i) {
            const int b = list[1 + i];
            ok = b >= 0 && b < B && keys[b] >= 0 && !seen[(size_t)b];
            if (ok) seen[(size_t)b] = 1;
            if (ok && i > 0) { const int a = list[i]; ok = keys[a] > keys[b] || (keys[a] == keys[b] && a < b); }
        }
        *value = ok ? 1 : 0;
        return PO_OK;
    }
    return PO_ERR_INVALID;
}

int po_set_stream(po_handle h, void *hip_stream) {
    if (!h) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return PO_OK;
}

static int make_dev_params(const po_handle_s *h, int form, int keep, po::DevParams *D) {
    const po_params &p = h->params;
    D->d1 = p.d[0]; D->d2 = p.d[1]; D->d3 = p.d[2]; D->d4 = p.d[3];
    if (form == PO_K) {
        D->w_dev = p.k_w_dev; D->w_c = p.k_w_curv; D->w_cr = p.k_w_curv_rate; D->w_s1 = p.w_slack; D->w_s2 = 0; D->w_u = 0; D->w_su = 0;
    } else {
        D->w_dev = p.w_dev; D->w_c = p.w_curv; D->w_cr = p.w_curv_rate; D->w_s1 = p.w_slack;
        D->w_s2 = (form == PO_KPC) ? p.w_k_slack : 0.0;
        D->w_u = keep * p.w_curv_rate;
        D->w_su = (form == PO_KPC) ? p.w_kp_slack * keep : 0.0;
    }
    D->margin = p.margin;
    D->kmax = std::tan(p.max_steer) / p.wheel_base;
    D->max_steer = p.max_steer;
    D->wheel_base = p.wheel_base;
    D->sigma = p.sigma; D->alpha = p.alpha; D->rho0 = p.rho0; D->eps_abs = p.eps_abs; D->eps_rel = p.eps_rel;
    D->eps_pinf = p.eps_prim_inf; D->adapt_tol = p.adapt_tol;
    D->max_iter = p.max_iter; D->check_every = p.check_every; D->adapt_every = p.adapt_every;
    D->end_heading = p.constraint_end_heading;
    D->polish = p.polish; D->pol_delta = p.polish_delta > 0 ? p.polish_delta : 1e-6; D->pol_refine = p.polish_refine_iter < 0 ? 0 : p.polish_refine_iter;
    D->refine = p.refine; D->ref_eps = p.refine_eps; D->ref_rounds = p.refine_rounds;
    D->ref_extra = (p.refine && p.refine_extra_rounds > 0) ? p.refine_extra_rounds : 0;
    // every refine_newton_* field is defaulted when it is not positive (a zero-initialised po_params must not silently disable the penalty growth) and held inside what
    // the rest of the engine allows; the ESCALATED caps (refine_newton_escalate: up to 100 x) are clamped inside the phase to kRhoMax / 1e8 (po_fast.inc, oracle alike)
    D->ref_nw_rho = p.refine_newton_rho > 0 ? std::min(p.refine_newton_rho, po::kRhoMax) : 100.0;
    D->ref_nw_rho_eq = p.refine_newton_rho_eq > 0 ? std::min(p.refine_newton_rho_eq, 1e8) : 1e4;
    D->ref_nw_rho_max = p.refine_newton_rho_max > 0 ? std::min(p.refine_newton_rho_max, po::kRhoMax) : 1e5;
    D->ref_nw_rho_eq_max = p.refine_newton_rho_eq_max > 0 ? std::min(p.refine_newton_rho_eq_max, 1e8) : (p.refine_newton_rho_eq_max < 0 ? 1e6 : 0.0);  // 0: the equality penalty never grows (documented switch)
    D->ref_ls_tol = p.refine_ls_tol > 0 ? p.refine_ls_tol : 1e-4;
    D->ref_nw_final = p.refine_newton_final; D->ref_nw_esc = p.refine_newton_escalate; D->ref_nw_slice = 0; D->ref_ls_max = p.refine_ls_max > 0 ? p.refine_ls_max : 30; D->ref_nw_max = p.refine_newton_max > 0 ? p.refine_newton_max : 300;
    return PO_OK;
}

static int validate(const po_batch_in *in, int *n, int *m, int *C) {
    if (!in) return PO_ERR_INVALID;
    if (in->B < 0) return PO_ERR_INVALID;
    int rc = po_problem_dims(in->formulation, in->N, in->keep, n, m, C);
    if (rc) return rc;
    if (!in->ref_x || !in->ref_y || !in->ref_z || !in->ref_k || !in->ref_s || !in->bounds || !in->x0 || !in->goal_z) return PO_ERR_INVALID;
    if (in->formulation == PO_KPC && (!in->max_k || !in->max_kp)) return PO_ERR_INVALID;
    if (po_lds_bytes(in->formulation, in->N, *C, in->keep) > 160 * 1024) return PO_ERR_UNSUPPORTED;
    return PO_OK;
}

static void fill_dev_batch(const po_handle_s *h, po::DevBatch *D, const po_batch_in *in, const po_batch_out *out, int n, int m, int C) {
    D->B = in->B; D->N = in->N; D->keep = in->keep; D->C = C;
    D->ref_x = in->ref_x; D->ref_y = in->ref_y; D->ref_z = in->ref_z; D->ref_k = in->ref_k; D->ref_s = in->ref_s;
    D->bounds = in->bounds; D->x0 = in->x0; D->goal_z = in->goal_z; D->max_k = in->max_k; D->max_kp = in->max_kp;
    D->n_points = in->n_points;
    D->order = in->order;
    D->out_states = out ? out->states : nullptr;
    D->out_info = out ? out->info : nullptr;
    D->out_x = out ? out->x : nullptr;
    D->dbg_cycles = nullptr;
    D->only_deferred = 0;
    D->perm_bits = 0;  // block -> path permutation (po_debug_set "identity_order": blockIdx order, dev tool)
    if (!h->env_identity && in->B > 8)
        while ((1 << D->perm_bits) < in->B) ++D->perm_bits;
    D->scale = nullptr;
    D->pol_state = nullptr; D->pol_stride = 0;
    D->round = 0;
    D->fb_list = nullptr;
    D->nw_phase = 0; D->nw_state = nullptr; D->nw_stride = 0; D->nw_keys = nullptr; D->nw_list = nullptr;
    D->nw_follows = 0;
    D->n = n; D->m = m;
}

int po_solve_batch_device(po_handle h, const po_batch_in *in, const po_batch_out *out) {
    if (!h || !out || !out->states || !out->info) return PO_ERR_INVALID;
    int n, m, C;
    int rc = validate(in, &n, &m, &C);
    if (rc) return rc;
    if (in->B == 0) return PO_OK;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    h->timed_host = false;  // (set again by po_solve_batch when this call is its inner solve: po_last_phase_ms never mixes this solve's events with an older call's)
    po::DevParams P;
    make_dev_params(h, in->formulation, in->keep, &P);
    po::DevBatch D;
    fill_dev_batch(h, &D, in, out, n, m, C);
    const bool dbg = h->env_cycles;
    if (dbg) {
        if ((rc = h->dbg_buf.ensure(sizeof(long long) * 16 * (size_t)in->B))) return rc;
        D.dbg_cycles = static_cast<long long *>(h->dbg_buf.p);
        HIP_TRY(hipMemsetAsync(D.dbg_cycles, 0, sizeof(long long) * 16 * (size_t)in->B, h->stream));
    }
    if ((rc = h->scale_buf.ensure(sizeof(double) * 64 * (size_t)in->B))) return rc;
    D.scale = static_cast<double *>(h->scale_buf.p);
    bool polish = false;
    if (h->params.polish || h->params.refine) {  // OSQP's polish (opt-in) and the Newton refinement pick the ADMM state up from pol_buf, where the solve kernels leave it
        const int sd = po_polish_state_doubles(in->formulation, in->N, C, in->keep);
        if (sd > 0) {  // (shapes on the single-level mapping have neither kernel: status_polish / status_refine = PO_NOT_AVAILABLE, see below)
            if ((rc = h->pol_buf.ensure(sizeof(double) * (size_t)sd * (size_t)in->B))) return rc;
            D.pol_state = static_cast<double *>(h->pol_buf.p);
            D.pol_stride = sd;
            polish = h->params.polish && po_has_polish_kernel(in->formulation, in->N, C, in->keep);  // (role-split shapes, keep 6 .. 16: no polish kernel -> status_polish = PO_NOT_AVAILABLE)
        }
    }
    const int rounds_total = (h->params.refine_rounds > 1 ? h->params.refine_rounds : 1) + P.ref_extra;
    // refine = 2: plain warm-start launches, the Newton refinement as its own launch, then the (nearly always empty) fallback launch for the later rounds
    const bool split = h->params.refine == 2 && D.pol_state != nullptr && rounds_total < 32;
    if (split && ((rc = h->fb_buf.ensure(sizeof(int) * ((size_t)in->B + 1))) || (rc = h->fb_host.ensure(64)))) return rc;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    h->timed_phases = false;
    // per-path equilibration (h->params.scaling class-level Ruiz passes; 0 -> identity), then the fused solve
    HIP_TRY(po_launch_scale(in->formulation, &D, &P, h->params.scaling, static_cast<double *>(h->scale_buf.p), h->stream));
    if (split) {
        po::DevParams P1 = P;  // the warm start: the plain solve kernels, stopped where round 0 of the rounds stops (10^(R-1) x eps)
        for (int r = 1; r < (h->params.refine_rounds > 1 ? h->params.refine_rounds : 1); ++r) { P1.eps_abs *= 10.0; P1.eps_rel *= 10.0; }
        D.nw_follows = 1;  // (newton_kernel writes the outputs of the paths this launch reports SOLVED)
        HIP_TRY(po_launch_solve(in->formulation, &D, &P1, h->stream, nullptr));
        D.nw_follows = 0;
        HIP_TRY(hipEventRecord(h->evp[0], h->stream));
        D.fb_list = static_cast<int *>(h->fb_buf.p);
        HIP_TRY(hipMemsetAsync(D.fb_list, 0, sizeof(int), h->stream));
        // SLICED LAUNCHES (engine-internal scheduling): the Newton launch is two — every path for nw_slice steps, the unfinished ones parked with a
        // priority key; a one-workgroup sort; the parked paths in order of expected remaining work, longest first.  One launch in engine order ends on a tail of a few
        // long paths (30 % of it on BASELINE config 3).  The operations and their order are unchanged: statuses and certificates do not depend on the slicing, solutions agree to round-off.
        // Left alone the engine slices where it was measured to pay: one wave per path (NT = 64: 4 x CUs paths at a time) and at least two rounds of them — BASELINE config 3
        // 4.76 -> 4.06 ms, K 5.41 -> 4.85, keep 8 6.15 -> 5.75, 2 048 paths of config 3 2.62 -> 2.53; two-wave shapes gain or lose 1 % (keep 2 / 3) or lose 10 % (KPC at N = 400:
        // eight rounds, the re-entry of a 2 x 36 KB state per path), a batch of one round has no queueing tail to remove (config 2: + 10 % for the re-entry).
        const bool slice_auto = in->B >= 2 * h->wave_slots && po_shape_threads(in->formulation, in->N, C, in->keep) == 64;
        int pd = (h->nw_slice > 0 && (slice_auto || h->nw_slice_forced)) ? po_newton_park_doubles(in->formulation, in->N, C, in->keep) : 0;
        // (the parking blocks are 39 KB per path on top of the state block: a batch they do not fit beside is solved unsliced — scheduling is not worth an out-of-memory error)
        if (pd > 0 && (h->nw_state_buf.ensure(sizeof(double) * (size_t)pd * (size_t)in->B) || h->nw_idx_buf.ensure(sizeof(int) * (2 * (size_t)in->B + 1)))) {
            (void)hipGetLastError();
            h->nw_state_buf.release();
            pd = 0;
        }
        h->nw_last_B = pd > 0 ? in->B : 0;
        if (pd > 0) {
            D.nw_state = static_cast<double *>(h->nw_state_buf.p); D.nw_stride = pd;
            D.nw_keys = static_cast<int *>(h->nw_idx_buf.p); D.nw_list = D.nw_keys + in->B;
            P.ref_nw_slice = h->nw_slice;
            HIP_TRY(hipMemsetAsync(D.nw_keys, 0xFF, sizeof(int) * (size_t)in->B, h->stream));  // (a path no workgroup of the first launch reaches — a malformed caller-side order — is not parked)
            D.nw_phase = 1;
            HIP_TRY(po_launch_newton(in->formulation, &D, &P, h->stream));
            HIP_TRY(po_launch_nw_sort(D.nw_keys, in->B, D.nw_list, h->stream));
            D.nw_phase = 2;
            HIP_TRY(po_launch_newton(in->formulation, &D, &P, h->stream));
            D.nw_phase = 0;
        } else {
            HIP_TRY(po_launch_newton(in->formulation, &D, &P, h->stream));
        }
        HIP_TRY(hipEventRecord(h->evp[1], h->stream));
        h->timed_phases = true;
        // The paths newton_kernel did not certify (rare) are on a device-side work list; newton_fallback_kernel takes them through the later rounds.  Its launch alone
        // costs 0.5 ms whatever its grid (1.2 KB of private segment per lane: the runtime re-provisions scratch for it; 7 % of a BASELINE config-3 solve), so
        // refine_chain = 2 reads the 4-byte count back and launches it only when there is something on the list — the call then returns when the Newton launch has
        // finished (it blocks).  refine_chain = 3: always launched, the call stays asynchronous.
        bool need_fb = true;
        // a stream that is being captured cannot be waited on (the copy below would never run and a synchronise invalidates the capture): always issue the launch then
        hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(h->stream, &cap_st) == hipSuccess && cap_st != hipStreamCaptureStatusNone;
        if (h->params.refine_chain != 3 && !capturing) {
            // the count lands in a pinned word the host SPINS on: a blocking hipStreamSynchronize wakes up through an interrupt — measured 0.5 ms, what the launch it
            // is meant to save costs; falls back to the blocking wait after 20 ms of spinning (a long solve: the wake-up latency no longer matters)
            volatile int *cnt = static_cast<volatile int *>(h->fb_host.p);
            *cnt = -1;
            HIP_TRY(hipMemcpyAsync(const_cast<int *>(cnt), D.fb_list, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            const auto ts0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (*cnt == -1) {
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - ts0 > std::chrono::milliseconds(20)) { HIP_TRY(hipStreamSynchronize(h->stream)); break; }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#else
                std::this_thread::yield();
#endif
            }
            need_fb = *cnt != 0;
        }
        if (need_fb) HIP_TRY(po_launch_newton_fallback(in->formulation, &D, &P, h->stream));
    } else {
        HIP_TRY(po_launch_solve(in->formulation, &D, &P, h->stream, nullptr));
    }
    if (split) HIP_TRY(po_launch_finalize_status(D.out_info, in->B, h->stream));
    if (polish) HIP_TRY(po_launch_polish(in->formulation, &D, &P, h->stream));
    {   // asked for, but this shape has no kernel for it: say so per path (PO_NOT_AVAILABLE) instead of leaving the "off" value 0
        const int no_refine = h->params.refine == 2 && !split, no_polish = h->params.polish && !polish;
        if (no_refine || no_polish) HIP_TRY(po_launch_mark_unavailable(D.out_info, in->B, no_refine, no_polish, h->stream));
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    if (dbg) {
        long long c4[16];
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(c4, D.dbg_cycles, sizeof(c4), hipMemcpyDeviceToHost));
        {   // batch totals of the factorisation counters
            std::vector<long long> all(16 * (size_t)in->B);
            HIP_TRY(hipMemcpy(all.data(), D.dbg_cycles, sizeof(long long) * all.size(), hipMemcpyDeviceToHost));
            long long nf = 0, nb = 0;
            double worst = 0;
            for (int b = 0; b < in->B; ++b) {
                nf += all[16 * (size_t)b + 13]; nb += all[16 * (size_t)b + 14];
                double w; std::memcpy(&w, &all[16 * (size_t)b + 15], sizeof(w)); worst = std::fmax(worst, w);
            }
            std::fprintf(stderr, "[po] batch of %d (form %d, N %d, keep %d): %lld scan factorisations, %lld fell back to the sequential chain; worst relative disagreement scan vs recursion %.3e\n", in->B, in->formulation, in->N, in->keep, nf, nb, worst);
        }
        std::fprintf(stderr, "[po] path0 cycles: rhs %lld solve %lld update %lld over %lld iterations | solve phases F1 %lld F2 %lld F3B1 %lld B2 %lld B3 %lld | first factorisation: blocks %lld chain %lld | uniform row classes %lld | scan factorisations %lld, fell back to the sequential chain %lld\n",
                     c4[0], c4[1], c4[2], c4[3], c4[4], c4[5], c4[6], c4[7], c4[8], c4[10], c4[11], c4[12], c4[13], c4[14]);
    }
    return PO_OK;
}

int po_solve_batch(po_handle h, const po_batch_in *in, const po_batch_out *out) {
    if (!h || !out || !out->states || !out->info) return PO_ERR_INVALID;
    int n, m, C;
    int rc = validate(in, &n, &m, &C);
    if (rc) return rc;
    if (in->B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t B = in->B, N = in->N;
    const bool kpc = in->formulation == PO_KPC;
    // one staging block: 5 ref arrays + bounds(8) + (max_k, max_kp) per point, x0(3) + goal per path, then n_points / order (ints)
    const size_t per_pt = 13 + (kpc ? 2 : 0);
    if (in->n_points)
        for (size_t b = 0; b < B; ++b)
            if (in->n_points[b] < 2 || in->n_points[b] > in->N) return PO_ERR_INVALID;
    if (in->order) {  // scheduling hint: must be a permutation (checked here; the device-pointer entry trusts its caller)
        std::vector<char> seen(B, 0);
        for (size_t b = 0; b < B; ++b) {
            const int v = in->order[b];
            if (v < 0 || (size_t)v >= B || seen[(size_t)v]) return PO_ERR_INVALID;
            seen[(size_t)v] = 1;
        }
    }
    const size_t in_bytes = sizeof(double) * (B * N * per_pt + B * 4 + 2 * ((B + 1) / 2 + 1));
    const size_t out_bytes = sizeof(double) * (B * N * 5 + (out->x ? B * (size_t)n : 0)) + sizeof(po_info) * B;
    {
        std::lock_guard<std::mutex> g(h->mu);
        if ((rc = h->in_buf.ensure(in_bytes)) || (rc = h->out_buf.ensure(out_bytes)) || (rc = h->pin_in.ensure(in_bytes)) || (rc = h->pin_out.ensure(out_bytes))) return rc;
    }
    // ---- pack: caller's (pageable) arrays -> the pinned block, on several host threads; the block goes to the device in slices as they fill ----
    double *d = static_cast<double *>(h->in_buf.p);
    char *pin = static_cast<char *>(h->pin_in.p);
    po_batch_in din = *in;
    std::vector<CopySeg> segs;
    size_t o = 0;
    auto up = [&](const double *src, size_t cnt, const double **dst) {
        *dst = d + o;
        segs.push_back({pin + o * sizeof(double), reinterpret_cast<const char *>(src), cnt * sizeof(double)});
        o += cnt;
    };
    up(in->ref_x, B * N, &din.ref_x); up(in->ref_y, B * N, &din.ref_y); up(in->ref_z, B * N, &din.ref_z); up(in->ref_k, B * N, &din.ref_k); up(in->ref_s, B * N, &din.ref_s);
    up(in->bounds, B * N * 8, &din.bounds); up(in->x0, B * 3, &din.x0); up(in->goal_z, B, &din.goal_z);
    if (kpc) { up(in->max_k, B * N, &din.max_k); up(in->max_kp, B * N, &din.max_kp); }
    size_t used = o * sizeof(double);
    if (in->n_points) {
        segs.push_back({pin + o * sizeof(double), reinterpret_cast<const char *>(in->n_points), sizeof(int) * B});
        din.n_points = reinterpret_cast<const int *>(d + o);
        used = o * sizeof(double) + sizeof(int) * B;
    }
    if (in->order) {
        const size_t oo = o * sizeof(double) + sizeof(int) * ((B + 1) / 2) * 2;
        segs.push_back({pin + oo, reinterpret_cast<const char *>(in->order), sizeof(int) * B});
        din.order = reinterpret_cast<const int *>(reinterpret_cast<char *>(d) + oo);
        used = oo + sizeof(int) * B;
    }
    const int nthr = h->host_threads > 0 ? h->host_threads : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    HIP_TRY(hipEventRecord(h->evh[0], h->stream));
    {
        const auto t0 = std::chrono::steady_clock::now();
        // the block in slices of 32 MB: slice k is on its way over PCIe while slice k + 1 is being packed (gaps between the int arrays travel as they are)
        const size_t kSlice = (size_t)32 << 20;
        for (size_t lo = 0; lo < used; lo += kSlice) {
            const size_t hi = std::min(used, lo + kSlice);
            std::vector<CopySeg> cur;
            for (const CopySeg &sg : segs) {
                const size_t s0 = (size_t)(sg.dst - pin), s1 = s0 + sg.bytes;
                const size_t a0 = std::max(s0, lo), a1 = std::min(s1, hi);
                if (a0 < a1) cur.push_back({pin + a0, sg.src + (a0 - s0), a1 - a0});
            }
            parallel_copy(cur, nthr);
            HIP_TRY(hipMemcpyAsync(reinterpret_cast<char *>(d) + lo, pin + lo, hi - lo, hipMemcpyHostToDevice, h->stream));
        }
        h->host_pack_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    HIP_TRY(hipEventRecord(h->evh[1], h->stream));
    po_batch_out dout;
    char *ob = static_cast<char *>(h->out_buf.p);
    dout.states = reinterpret_cast<double *>(ob);
    dout.x = out->x ? dout.states + B * N * 5 : nullptr;
    dout.info = reinterpret_cast<po_info *>(ob + sizeof(double) * (B * N * 5 + (out->x ? B * (size_t)n : 0)));
    rc = po_solve_batch_device(h, &din, &dout);
    if (rc) return rc;
    // ---- D2H into the pinned block (one copy), then unpack on the host threads ----
    char *pout = static_cast<char *>(h->pin_out.p);
    HIP_TRY(hipMemcpyAsync(pout, ob, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->evh[2], h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<CopySeg> us;
        us.push_back({reinterpret_cast<char *>(out->states), pout, sizeof(double) * B * N * 5});
        if (out->x) us.push_back({reinterpret_cast<char *>(out->x), pout + sizeof(double) * B * N * 5, sizeof(double) * B * (size_t)n});
        us.push_back({reinterpret_cast<char *>(out->info), pout + sizeof(double) * (B * N * 5 + (out->x ? B * (size_t)n : 0)), sizeof(po_info) * B});
        parallel_copy(us, nthr);
        h->host_unpack_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    h->timed_host = true;
    return PO_OK;
}

// What the last po_solve_batch (host pointers) spent where, ms: [0] pack + H2D (stream time from the start of the call to the last H2D slice: the pack runs
// under the copies), [1] the solve (= po_last_kernel_ms), [2] D2H, [3] host pack alone (wall), [4] host unpack (wall).  And, for the split scheduling of
// refine = 2, the solve's own phases: [5] equilibration + warm-start launches, [6] the Newton launch, [7] the per-round fallback launches + status sweep (0 otherwise).
int po_last_phase_ms(po_handle h, float *ms8) {
    if (!h || !ms8 || !h->timed) return PO_ERR_INVALID;
    for (int i = 0; i < 8; ++i) ms8[i] = 0.0f;
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(&ms8[1], h->ev0, h->ev1));
    if (h->timed_host) {
        HIP_TRY(hipEventSynchronize(h->evh[2]));
        HIP_TRY(hipEventElapsedTime(&ms8[0], h->evh[0], h->evh[1]));
        HIP_TRY(hipEventElapsedTime(&ms8[2], h->ev1, h->evh[2]));
        ms8[3] = (float)h->host_pack_ms; ms8[4] = (float)h->host_unpack_ms;
    }
    if (h->timed_phases) {
        HIP_TRY(hipEventElapsedTime(&ms8[5], h->ev0, h->evp[0]));
        HIP_TRY(hipEventElapsedTime(&ms8[6], h->evp[0], h->evp[1]));
        HIP_TRY(hipEventElapsedTime(&ms8[7], h->evp[1], h->ev1));
    }
    return PO_OK;
}

int po_assemble_batch(po_handle h, const po_batch_in *in, double *l, double *u, double *dyn) {
    if (!h || !l || !u || !dyn) return PO_ERR_INVALID;
    if (in && in->n_points) return PO_ERR_INVALID;  // diagnostics work on uniform batches only
    int n, m, C;
    int rc = validate(in, &n, &m, &C);
    if (rc) return rc;
    if (in->B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t B = in->B, N = in->N;
    const bool kpc = in->formulation == PO_KPC;
    const size_t per_pt = 13 + (kpc ? 2 : 0);
    const size_t in_bytes = sizeof(double) * (B * N * per_pt + B * 4);
    const size_t a_bytes = sizeof(double) * (2 * B * (size_t)m + B * (N - 1) * 3);
    {
        std::lock_guard<std::mutex> g(h->mu);
        if ((rc = h->in_buf.ensure(in_bytes)) || (rc = h->asm_buf.ensure(a_bytes))) return rc;
    }
    double *d = static_cast<double *>(h->in_buf.p);
    po_batch_in din = *in;
    size_t o = 0;
    auto up = [&](const double *src, size_t cnt, const double **dst) -> int {
        *dst = d + o;
        if (!hip_ok(hipMemcpyAsync(d + o, src, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream), "H2D")) return PO_ERR_HIP;
        o += cnt;
        return PO_OK;
    };
    if ((rc = up(in->ref_x, B * N, &din.ref_x)) || (rc = up(in->ref_y, B * N, &din.ref_y)) || (rc = up(in->ref_z, B * N, &din.ref_z)) ||
        (rc = up(in->ref_k, B * N, &din.ref_k)) || (rc = up(in->ref_s, B * N, &din.ref_s)) || (rc = up(in->bounds, B * N * 8, &din.bounds)) ||
        (rc = up(in->x0, B * 3, &din.x0)) || (rc = up(in->goal_z, B, &din.goal_z)))
        return rc;
    if (kpc && ((rc = up(in->max_k, B * N, &din.max_k)) || (rc = up(in->max_kp, B * N, &din.max_kp)))) return rc;
    double *dl = static_cast<double *>(h->asm_buf.p), *du = dl + B * (size_t)m, *dd = du + B * (size_t)m;
    HIP_TRY(hipMemsetAsync(dl, 0, a_bytes, h->stream));
    po::DevParams P;
    make_dev_params(h, in->formulation, in->keep, &P);
    po::DevBatch D;
    fill_dev_batch(h, &D, &din, nullptr, n, m, C);
    HIP_TRY(po_launch_assemble(in->formulation, &D, &P, dl, du, dd, h->stream));
    HIP_TRY(hipMemcpyAsync(l, dl, sizeof(double) * B * (size_t)m, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(u, du, sizeof(double) * B * (size_t)m, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(dyn, dd, sizeof(double) * B * (N - 1) * 3, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

int po_scaling_batch(po_handle h, const po_batch_in *in, double *out) {
    if (!h || !out) return PO_ERR_INVALID;
    if (in && in->n_points) return PO_ERR_INVALID;  // diagnostics work on uniform batches only
    int n, m, C;
    int rc = validate(in, &n, &m, &C);
    if (rc) return rc;
    if (in->B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t B = in->B, N = in->N;
    {
        std::lock_guard<std::mutex> g(h->mu);
        if ((rc = h->in_buf.ensure(sizeof(double) * B * N)) || (rc = h->scale_buf.ensure(sizeof(double) * 64 * B))) return rc;
    }
    po_batch_in din = *in;
    HIP_TRY(hipMemcpyAsync(h->in_buf.p, in->ref_s, sizeof(double) * B * N, hipMemcpyHostToDevice, h->stream));
    din.ref_s = static_cast<const double *>(h->in_buf.p);
    po::DevParams P;
    make_dev_params(h, in->formulation, in->keep, &P);
    po::DevBatch D;
    fill_dev_batch(h, &D, &din, nullptr, n, m, C);
    HIP_TRY(po_launch_scale(in->formulation, &D, &P, h->params.scaling, static_cast<double *>(h->scale_buf.p), h->stream));
    HIP_TRY(hipMemcpyAsync(out, h->scale_buf.p, sizeof(double) * 64 * B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

int po_last_kernel_ms(po_handle h, float *ms) {
    if (!h || !ms || !h->timed) return PO_ERR_INVALID;
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return PO_OK;
}

// ---- post-solve step -------------------------------------------------------------------------------------------
static po::DevCar make_car(const po_params &p) {
    // CollisionChecker ctor (collision_checker.cpp:9-15) + CarGeometry::setCircles (car_geometry.cpp:38-56)
    po::DevCar c{};
    const double width = p.car_width, back = p.car_length / 2.0 - p.rear_axle_to_center, front = p.car_length / 2.0 + p.rear_axle_to_center;
    const double length = front + back;
    c.bx = (front - back) / 2.0;
    c.br = std::sqrt((length / 2) * (length / 2) + (width / 2) * (width / 2));
    const double shift = width / 4.0, small_r = std::sqrt(2 * (shift * shift));
    const double large_r = std::sqrt(width * width + ((length - width) / 2.0) * ((length - width) / 2.0)) / 2;
    const double px[4] = {-back + shift, -back + shift, front - shift, front - shift};          // rr, rl, fr, fl
    const double py[4] = {-width / 2.0 + shift, width / 2.0 - shift, -width / 2.0 + shift, width / 2.0 - shift};
    for (int i = 0; i < 4; ++i) { c.cx[i] = px[i]; c.cy[i] = py[i]; c.cr[i] = small_r; }
    c.cx[4] = c.bx + (length - width) / 4; c.cy[4] = 0; c.cr[4] = large_r;  // fm
    c.cx[5] = c.bx - (length - width) / 4; c.cy[5] = 0; c.cr[5] = large_r;  // rm
    c.enable = p.enable_collision_check;
    return c;
}

int po_set_map(po_handle h, const po_map *map) {
    if (!h || !map || !map->distance || map->size_x < 1 || map->size_y < 1 || !(map->resolution > 0)) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = sizeof(float) * (size_t)map->size_x * map->size_y;
    if (int rc = h->map_buf.ensure(bytes)) return rc;
    HIP_TRY(hipMemcpyAsync(h->map_buf.p, map->distance, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->map = po::DevMap{static_cast<const float *>(h->map_buf.p), map->size_x, map->size_y, map->resolution, map->pos_x, map->pos_y};
    return PO_OK;
}

int po_postcheck_batch_device(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info, int *n_valid, int *ok) {
    if (!h || B < 0 || N < 1 || (B > 0 && (!states || !info || !n_valid || !ok))) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->map.d && h->params.enable_collision_check) return PO_ERR_INVALID;  // no map set
    if (B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const po::DevCar car = make_car(h->params);
    HIP_TRY(po_launch_postcheck(&h->map, &car, B, N, n_points, states, info, n_valid, ok, h->stream));
    return PO_OK;
}

int po_densify_batch_device(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info, int M, double *out_states, int *n_out, int *ok) {
    if (!h || B < 0 || N < 3 || M < 1 || (B > 0 && (!states || !info || !out_states || !n_out || !ok))) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->map.d && h->params.enable_collision_check) return PO_ERR_INVALID;  // no map set
    if (!(h->params.output_spacing > 0)) return PO_ERR_INVALID;
    if (B == 0) return PO_OK;
    if (sizeof(double) * 15 * (size_t)N > 160 * 1024) return PO_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(h->device));
    const po::DevCar car = make_car(h->params);
    HIP_TRY(po_launch_densify(&h->map, &car, B, N, n_points, states, info, h->params.output_spacing, M, out_states, n_out, ok, h->stream));
    return PO_OK;
}

int po_densify_batch(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info, int M, double *out_states, int *n_out, int *ok) {
    if (!h || B < 0 || N < 3 || M < 1 || (B > 0 && (!states || !info || !out_states || !n_out || !ok))) return PO_ERR_INVALID;
    if (B == 0) return PO_OK;
    const size_t bs = sizeof(double) * 5 * (size_t)B * N, bo = sizeof(double) * 5 * (size_t)B * M, bi = sizeof(po_info) * (size_t)B, bn = sizeof(int) * (size_t)B;
    char *base = nullptr;
    {
        std::lock_guard<std::mutex> g(h->mu);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = h->post_buf.ensure(bs + bo + bi + 3 * bn + 64)) return rc;
        base = static_cast<char *>(h->post_buf.p);
        HIP_TRY(hipMemcpyAsync(base, states, bs, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(base + bs + bo, info, bi, hipMemcpyHostToDevice, h->stream));
        if (n_points) HIP_TRY(hipMemcpyAsync(base + bs + bo + bi, n_points, bn, hipMemcpyHostToDevice, h->stream));
    }
    int *dn = reinterpret_cast<int *>(base + bs + bo + bi);
    const int rc = po_densify_batch_device(h, B, N, n_points ? dn : nullptr, reinterpret_cast<const double *>(base), reinterpret_cast<const po_info *>(base + bs + bo), M,
                                           reinterpret_cast<double *>(base + bs), dn + B, dn + 2 * B);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipMemcpyAsync(out_states, base + bs, bo, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(n_out, dn + B, bn, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ok, dn + 2 * B, bn, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

int po_postcheck_batch(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info, int *n_valid, int *ok) {
    if (!h || B < 0 || N < 1 || (B > 0 && (!states || !info || !n_valid || !ok))) return PO_ERR_INVALID;
    if (B == 0) return PO_OK;
    const size_t bs = sizeof(double) * 5 * (size_t)B * N, bi = sizeof(po_info) * (size_t)B, bn = sizeof(int) * (size_t)B;
    char *base = nullptr;
    {
        std::lock_guard<std::mutex> g(h->mu);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = h->post_buf.ensure(bs + bi + 3 * bn + 64)) return rc;
        base = static_cast<char *>(h->post_buf.p);
        HIP_TRY(hipMemcpyAsync(base, states, bs, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(base + bs, info, bi, hipMemcpyHostToDevice, h->stream));
        if (n_points) HIP_TRY(hipMemcpyAsync(base + bs + bi, n_points, bn, hipMemcpyHostToDevice, h->stream));
    }
    int *dn = reinterpret_cast<int *>(base + bs + bi);
    const int rc = po_postcheck_batch_device(h, B, N, n_points ? dn : nullptr, reinterpret_cast<const double *>(base),
                                             reinterpret_cast<const po_info *>(base + bs), dn + B, dn + 2 * B);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipMemcpyAsync(n_valid, dn + B, bn, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ok, dn + 2 * B, bn, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

// ---- corridor-bounds producer ------------------------------------------------------------------------------------
int po_bounds_batch_device(po_handle h, const po_bounds_in *in, double *bounds, int *n_valid) {
    if (!h || !in || in->B < 0 || in->N < 1 || in->K < 3) return PO_ERR_INVALID;
    if (in->B > 0 && (!in->ref_x || !in->ref_y || !in->ref_z || !in->ref_s || !in->knot_s || !in->knot_x || !in->knot_y || !bounds || !n_valid)) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->map.d) return PO_ERR_INVALID;  // po_set_map first
    if (in->B == 0) return PO_OK;
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = h->coef_buf.ensure(sizeof(double) * (size_t)in->B * 2 * 6 * in->K)) return rc;
    po::DevBounds D{};
    D.B = in->B; D.N = in->N; D.K = in->K;
    D.ref_x = in->ref_x; D.ref_y = in->ref_y; D.ref_z = in->ref_z; D.ref_s = in->ref_s; D.n_points = in->n_points;
    D.knot_s = in->knot_s; D.knot_x = in->knot_x; D.knot_y = in->knot_y; D.n_knots = in->n_knots;
    const po_params &p = h->params;
    for (int j = 0; j < 4; ++j) D.d[j] = p.d[j];
    D.radius = std::sqrt((p.car_length / 8) * (p.car_length / 8) + (p.car_width / 2) * (p.car_width / 2)) + p.safety_margin;  // planning_flags.cpp:9
    D.coef = static_cast<double *>(h->coef_buf.p);
    HIP_TRY(po_launch_bounds(&h->map, &D, bounds, n_valid, h->stream));
    return PO_OK;
}

int po_bounds_batch(po_handle h, const po_bounds_in *in, double *bounds, int *n_valid) {
    if (!h || !in || in->B < 0 || in->N < 1 || in->K < 3) return PO_ERR_INVALID;
    if (in->B > 0 && (!in->ref_x || !in->ref_y || !in->ref_z || !in->ref_s || !in->knot_s || !in->knot_x || !in->knot_y || !bounds || !n_valid)) return PO_ERR_INVALID;
    if (in->B == 0) return PO_OK;
    const size_t bn = sizeof(double) * (size_t)in->B * in->N, bk = sizeof(double) * (size_t)in->B * in->K, bi = sizeof(int) * (size_t)in->B;
    const size_t bo = sizeof(double) * (size_t)in->B * in->N * 8;
    char *base = nullptr;
    {
        std::lock_guard<std::mutex> g(h->mu);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = h->bnd_buf.ensure(4 * bn + 3 * bk + 3 * bi + bo + 64)) return rc;
        base = static_cast<char *>(h->bnd_buf.p);
        const void *src[7] = {in->ref_x, in->ref_y, in->ref_z, in->ref_s, in->knot_s, in->knot_x, in->knot_y};
        size_t off = 0;
        for (int i = 0; i < 7; ++i) { const size_t sz = i < 4 ? bn : bk; HIP_TRY(hipMemcpyAsync(base + off, src[i], sz, hipMemcpyHostToDevice, h->stream)); off += sz; }
        if (in->n_points) HIP_TRY(hipMemcpyAsync(base + off, in->n_points, bi, hipMemcpyHostToDevice, h->stream));
        if (in->n_knots) HIP_TRY(hipMemcpyAsync(base + off + bi, in->n_knots, bi, hipMemcpyHostToDevice, h->stream));
    }
    po_bounds_in d = *in;
    const double *pd = reinterpret_cast<const double *>(base);
    d.ref_x = pd; d.ref_y = pd + (size_t)in->B * in->N; d.ref_z = pd + 2 * (size_t)in->B * in->N; d.ref_s = pd + 3 * (size_t)in->B * in->N;
    const double *pk = pd + 4 * (size_t)in->B * in->N;
    d.knot_s = pk; d.knot_x = pk + (size_t)in->B * in->K; d.knot_y = pk + 2 * (size_t)in->B * in->K;
    int *pi = reinterpret_cast<int *>(base + 4 * bn + 3 * bk);
    d.n_points = in->n_points ? pi : nullptr;
    d.n_knots = in->n_knots ? pi + in->B : nullptr;
    double *dout = reinterpret_cast<double *>(base + 4 * bn + 3 * bk + 3 * bi + (8 - (3 * bi) % 8) % 8);
    const int rc = po_bounds_batch_device(h, &d, dout, pi + 2 * in->B);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipMemcpyAsync(bounds, dout, bo, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(n_valid, pi + 2 * in->B, bi, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

// ---- reference-smoothing QPs -----------------------------------------------------------------------
int po_smooth_dims(int kind, int P, int *n, int *m) {
    int nn, mm;
    if (kind == PO_SMOOTH_TENSION2) { if (P < 3) return PO_ERR_INVALID; nn = 4 * P - 1; mm = 3 * (P - 1) + 2; }  // tension_smoother_2.cpp:177-178
    else if (kind == PO_SMOOTH_TENSION) { if (P < 3) return PO_ERR_INVALID; nn = 3 * P; mm = 3 * P; }            // tension_smoother.cpp:201-202
    else if (kind == PO_SMOOTH_POST) { if (P < 4) return PO_ERR_INVALID; nn = 3 * P; mm = 3 * P - 2; }           // reference_path_smoother.cpp:536,544-545
    else return PO_ERR_INVALID;
    if (n) *n = nn;
    if (m) *m = mm;
    return PO_OK;
}

static int smooth_args_ok(const po_smooth_in *in, const po_smooth_out *out) {
    if (!in || !out || in->B < 0) return 0;
    if (po_smooth_dims(in->kind, in->P, nullptr, nullptr)) return 0;
    if (in->B == 0) return 1;
    if (!in->s || !out->x || !out->info) return 0;
    if (in->kind == PO_SMOOTH_POST) return in->lb && in->ub && in->l0;
    return in->x && in->y && in->angle && in->k && out->y && out->s;
}

int po_smooth_batch_device(po_handle h, const po_smooth_in *in, const po_smooth_out *out) {
    if (!h || !smooth_args_ok(in, out)) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    if (in->kind == PO_SMOOTH_TENSION && !h->map.d) return PO_ERR_INVALID;  // po_set_map first (clearance of every point)
    if (in->B == 0) return PO_OK;
    if (po_smooth_lds_bytes(in->kind, in->P) > 160 * 1024) return PO_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(h->device));
    po::DevSmooth D{};
    D.kind = in->kind; D.B = in->B; D.P = in->P; D.n_points = in->n_points;
    D.x = in->x; D.y = in->y; D.angle = in->angle; D.k = in->k; D.s = in->s; D.lb = in->lb; D.ub = in->ub; D.l0 = in->l0;
    D.out_x = out->x; D.out_y = out->y; D.out_s = out->s; D.info = out->info; D.raw = out->raw;
    int nmax = 0;
    po_smooth_dims(in->kind, in->P, &nmax, nullptr);
    D.raw_stride = nmax;
    const po_params &p = h->params;
    const double w[6] = {p.t2_w_dev, p.t2_w_curv, p.t2_w_curv_rate, p.cart_w_curv, p.cart_w_curv_rate, p.cart_w_dev};
    for (int i = 0; i < 6; ++i) D.w[i] = w[i];
    D.sigma = p.sigma; D.alpha = p.alpha; D.rho0 = p.rho0; D.eps_abs = p.eps_abs; D.eps_rel = p.eps_rel;
    D.eps_pinf = p.eps_prim_inf; D.eps_dinf = p.eps_dual_inf; D.adapt_tol = p.adapt_tol;
    D.max_iter = p.max_iter; D.check_every = p.check_every; D.adapt_every = p.adapt_every; D.scaling = p.scaling;
    D.scratch_stride = po_smooth_scratch_doubles(in->kind, in->P);
    if (int rc = h->smooth_buf.ensure(sizeof(double) * D.scratch_stride * (size_t)in->B)) return rc;
    D.scratch = static_cast<double *>(h->smooth_buf.p);
    D.map = h->map;
    D.perm_bits = 0;
    D.seq_band = h->env_smooth_seq ? 1 : 0; D.waves = h->env_smooth_waves; D.nopad = h->env_smooth_nopad ? 1 : 0;  // developer A/B switches (read once at po_create)
    if (!h->env_identity && in->B > 8)
        while ((1 << D.perm_bits) < in->B) ++D.perm_bits;
    const bool dbg = h->env_smooth_debug;  // dev tool: per-phase cycle totals of instance 0..B-1 printed to stderr
    if (dbg) {
        if (int rc = h->dbg_buf.ensure(sizeof(long long) * 8 * (size_t)in->B)) return rc;
        D.dbg_cycles = static_cast<long long *>(h->dbg_buf.p);
    }
    if (h->timed) HIP_TRY(hipEventRecord(h->ev0, h->stream));
    HIP_TRY(po_launch_smooth(&D, h->stream));
    if (dbg) {
        std::vector<long long> hc(8 * (size_t)in->B);
        HIP_TRY(hipMemcpyAsync(hc.data(), D.dbg_cycles, sizeof(long long) * hc.size(), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        double tot[8] = {0};
        for (int b = 0; b < in->B; ++b) for (int i = 0; i < 8; ++i) tot[i] += (double)hc[8 * (size_t)b + i];
        std::fprintf(stderr, "po_smooth kind %d: mean cycles/instance setup %.0f factor %.0f rhs %.0f solve %.0f update %.0f check %.0f out %.0f\n", in->kind,
                     tot[0] / in->B, tot[1] / in->B, tot[2] / in->B, tot[3] / in->B, tot[4] / in->B, tot[5] / in->B, tot[6] / in->B);
    }
    if (h->timed) HIP_TRY(hipEventRecord(h->ev1, h->stream));
    return PO_OK;
}

int po_smooth_batch(po_handle h, const po_smooth_in *in, const po_smooth_out *out) {
    if (!h || !smooth_args_ok(in, out)) return PO_ERR_INVALID;
    if (in->B == 0) return PO_OK;
    int nmax = 0;
    po_smooth_dims(in->kind, in->P, &nmax, nullptr);
    const size_t bp = sizeof(double) * (size_t)in->B * in->P, bb = sizeof(double) * (size_t)in->B, bi = sizeof(int) * (size_t)in->B;
    const size_t binfo = sizeof(po_info) * (size_t)in->B, braw = out->raw ? sizeof(double) * (size_t)in->B * nmax : 0;
    char *base = nullptr;
    const void *src[7] = {in->x, in->y, in->angle, in->k, in->s, in->lb, in->ub};
    {
        std::lock_guard<std::mutex> g(h->mu);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = h->smooth_io.ensure(10 * bp + bb + bi + binfo + braw + 64)) return rc;
        base = static_cast<char *>(h->smooth_io.p);
        for (int i = 0; i < 7; ++i)
            if (src[i]) HIP_TRY(hipMemcpyAsync(base + i * bp, src[i], bp, hipMemcpyHostToDevice, h->stream));
        if (in->l0) HIP_TRY(hipMemcpyAsync(base + 10 * bp, in->l0, bb, hipMemcpyHostToDevice, h->stream));
        if (in->n_points) HIP_TRY(hipMemcpyAsync(base + 10 * bp + bb + binfo + braw, in->n_points, bi, hipMemcpyHostToDevice, h->stream));
    }
    po_smooth_in d = *in;
    const double *pd = reinterpret_cast<const double *>(base);
    const size_t np = (size_t)in->B * in->P;
    d.x = in->x ? pd : nullptr; d.y = in->y ? pd + np : nullptr; d.angle = in->angle ? pd + 2 * np : nullptr; d.k = in->k ? pd + 3 * np : nullptr;
    d.s = pd + 4 * np; d.lb = in->lb ? pd + 5 * np : nullptr; d.ub = in->ub ? pd + 6 * np : nullptr;
    d.l0 = in->l0 ? reinterpret_cast<const double *>(base + 10 * bp) : nullptr;
    d.n_points = in->n_points ? reinterpret_cast<const int *>(base + 10 * bp + bb + binfo + braw) : nullptr;
    po_smooth_out dout{};
    double *po_ = reinterpret_cast<double *>(base + 7 * bp);
    dout.x = po_; dout.y = po_ + np; dout.s = po_ + 2 * np;
    dout.info = reinterpret_cast<po_info *>(base + 10 * bp + bb);
    dout.raw = out->raw ? reinterpret_cast<double *>(base + 10 * bp + bb + binfo) : nullptr;
    const int rc = po_smooth_batch_device(h, &d, &dout);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipMemcpyAsync(out->x, dout.x, bp, hipMemcpyDeviceToHost, h->stream));
    if (out->y) HIP_TRY(hipMemcpyAsync(out->y, dout.y, bp, hipMemcpyDeviceToHost, h->stream));
    if (out->s) HIP_TRY(hipMemcpyAsync(out->s, dout.s, bp, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(out->info, dout.info, binfo, hipMemcpyDeviceToHost, h->stream));
    if (out->raw) HIP_TRY(hipMemcpyAsync(out->raw, dout.raw, braw, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

// ---- reference re-sampling, limits, DP lattice search -----------------------------------------------
static int spline_args_ok(const po_spline_in *in) {
    return in && in->B >= 0 && in->K >= 3 && (in->B == 0 || (in->knot_s && in->knot_x && in->knot_y && in->length));
}
static int make_dev_spline(po_handle h, const po_spline_in *in, po::DevSpline *D) {
    (void)h;  // the spline coefficients are fitted in LDS by each consumer kernel
    D->B = in->B; D->K = in->K; D->knot_s = in->knot_s; D->knot_x = in->knot_x; D->knot_y = in->knot_y; D->n_knots = in->n_knots;
    D->length = in->length; D->coef = nullptr;
    return PO_OK;
}

int po_resample_batch_device(po_handle h, const po_spline_in *in, double ds_smaller, double ds_larger, int N, double *ref_x, double *ref_y,
                             double *ref_z, double *ref_k, double *ref_s, int *n_points) {
    if (!h || !spline_args_ok(in) || N < 1 || !(ds_smaller <= ds_larger) || !(ds_smaller > 0)) return PO_ERR_INVALID;  // CHECK_LE(delta_s_smaller, delta_s_larger)
    if (in->B > 0 && (!ref_x || !ref_y || !ref_z || !ref_k || !ref_s || !n_points)) return PO_ERR_INVALID;
    if (in->B == 0) return PO_OK;
    if (po_spline_lds_bytes(in->K) > 64 * 1024) return PO_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    po::DevSpline D{};
    if (int rc = make_dev_spline(h, in, &D)) return rc;
    po::DevResample R{ds_smaller, ds_larger, h->params.enable_dynamic_segmentation, N, ref_x, ref_y, ref_z, ref_k, ref_s, n_points};
    HIP_TRY(po_launch_resample(&D, &R, h->stream));
    return PO_OK;
}

int po_limits_batch_device(po_handle h, int B, int N, const int *n_points, const double *v, const double *a, double *max_k, double *max_kp) {
    if (!h || B < 0 || N < 1 || (B > 0 && (!v || !a || !max_k || !max_kp))) return PO_ERR_INVALID;
    if (B == 0) return PO_OK;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(po_launch_limits(B, N, n_points, v, a, max_k, max_kp, h->params.mu, h->params.max_curvature_rate, h->stream));
    return PO_OK;
}

int po_dp_search_batch_device(po_handle h, const po_spline_in *in, const double *start, int L, double *layer_s, double *lb, double *ub, double *l0,
                              int *n_layers) {
    if (!h || !spline_args_ok(in) || L < 1) return PO_ERR_INVALID;
    if (in->B > 0 && (!start || !layer_s || !lb || !ub || !l0 || !n_layers)) return PO_ERR_INVALID;
    const po_params &p = h->params;
    if (!(p.search_lat_spacing > 0) || !(p.search_long_spacing > 0) || !(p.search_lateral_range > 0) ||
        2 * p.search_lateral_range / p.search_lat_spacing + 1 > 64) return PO_ERR_UNSUPPORTED;  // one wave per path: <= 64 lateral samples
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->map.d) return PO_ERR_INVALID;  // po_set_map first
    if (in->B == 0) return PO_OK;
    if (po_dp_lds_bytes(in->K, L) > 160 * 1024) return PO_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(h->device));
    po::DevSpline D{};
    if (int rc = make_dev_spline(h, in, &D)) return rc;
    po::DevSearch Q{p.search_lateral_range, p.search_long_spacing, p.search_lat_spacing, start, L, layer_s, lb, ub, l0, n_layers};
    HIP_TRY(po_launch_dp_search(&h->map, &D, &Q, h->env_dp_one_wave ? 1 : 0, h->stream));
    return PO_OK;
}

// host-pointer wrappers: stage the spline batch, run the device entry, copy back
namespace {
struct StagedSpline { po_spline_in d; char *next; };
int stage_spline_batch(po_handle h, const po_spline_in *in, size_t extra_bytes, StagedSpline *out) {
    const size_t bk = sizeof(double) * (size_t)in->B * in->K, bb = sizeof(double) * (size_t)in->B, bi = sizeof(int) * (size_t)in->B;
    if (int rc = h->plan_io.ensure(3 * bk + bb + bi + 8 + extra_bytes + 64)) return rc;
    char *base = static_cast<char *>(h->plan_io.p);
    const void *src[3] = {in->knot_s, in->knot_x, in->knot_y};
    for (int i = 0; i < 3; ++i) HIP_TRY(hipMemcpyAsync(base + i * bk, src[i], bk, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(base + 3 * bk, in->length, bb, hipMemcpyHostToDevice, h->stream));
    if (in->n_knots) HIP_TRY(hipMemcpyAsync(base + 3 * bk + bb, in->n_knots, bi, hipMemcpyHostToDevice, h->stream));
    out->d = *in;
    out->d.knot_s = reinterpret_cast<const double *>(base); out->d.knot_x = reinterpret_cast<const double *>(base + bk);
    out->d.knot_y = reinterpret_cast<const double *>(base + 2 * bk); out->d.length = reinterpret_cast<const double *>(base + 3 * bk);
    out->d.n_knots = in->n_knots ? reinterpret_cast<const int *>(base + 3 * bk + bb) : nullptr;
    out->next = base + ((3 * bk + bb + bi + 7) & ~(size_t)7);
    return PO_OK;
}
}  // namespace

int po_resample_batch(po_handle h, const po_spline_in *in, double ds_smaller, double ds_larger, int N, double *ref_x, double *ref_y, double *ref_z,
                      double *ref_k, double *ref_s, int *n_points) {
    if (!h || !spline_args_ok(in) || N < 1) return PO_ERR_INVALID;
    if (in->B > 0 && (!ref_x || !ref_y || !ref_z || !ref_k || !ref_s || !n_points)) return PO_ERR_INVALID;
    if (in->B == 0) return PO_OK;
    const size_t bn = sizeof(double) * (size_t)in->B * N, bi = sizeof(int) * (size_t)in->B;
    StagedSpline S{};
    {
        std::lock_guard<std::mutex> g(h->mu);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = stage_spline_batch(h, in, 5 * bn + bi, &S)) return rc;
    }
    double *o = reinterpret_cast<double *>(S.next);
    int *on = reinterpret_cast<int *>(S.next + 5 * bn);
    const size_t n1 = (size_t)in->B * N;
    if (int rc = po_resample_batch_device(h, &S.d, ds_smaller, ds_larger, N, o, o + n1, o + 2 * n1, o + 3 * n1, o + 4 * n1, on)) return rc;
    std::lock_guard<std::mutex> g(h->mu);
    double *dst[5] = {ref_x, ref_y, ref_z, ref_k, ref_s};
    for (int i = 0; i < 5; ++i) HIP_TRY(hipMemcpyAsync(dst[i], o + i * n1, bn, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(n_points, on, bi, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

int po_limits_batch(po_handle h, int B, int N, const int *n_points, const double *v, const double *a, double *max_k, double *max_kp) {
    if (!h || B < 0 || N < 1 || (B > 0 && (!v || !a || !max_k || !max_kp))) return PO_ERR_INVALID;
    if (B == 0) return PO_OK;
    const size_t bn = sizeof(double) * (size_t)B * N, bi = sizeof(int) * (size_t)B;
    char *base = nullptr;
    {
        std::lock_guard<std::mutex> g(h->mu);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = h->plan_io.ensure(4 * bn + bi + 64)) return rc;
        base = static_cast<char *>(h->plan_io.p);
        HIP_TRY(hipMemcpyAsync(base, v, bn, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(base + bn, a, bn, hipMemcpyHostToDevice, h->stream));
        if (n_points) HIP_TRY(hipMemcpyAsync(base + 4 * bn, n_points, bi, hipMemcpyHostToDevice, h->stream));
    }
    double *d = reinterpret_cast<double *>(base);
    const size_t n1 = (size_t)B * N;
    if (int rc = po_limits_batch_device(h, B, N, n_points ? reinterpret_cast<const int *>(base + 4 * bn) : nullptr, d, d + n1, d + 2 * n1, d + 3 * n1)) return rc;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipMemcpyAsync(max_k, d + 2 * n1, bn, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(max_kp, d + 3 * n1, bn, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

int po_dp_search_batch(po_handle h, const po_spline_in *in, const double *start, int L, double *layer_s, double *lb, double *ub, double *l0, int *n_layers) {
    if (!h || !spline_args_ok(in) || L < 1) return PO_ERR_INVALID;
    if (in->B > 0 && (!start || !layer_s || !lb || !ub || !l0 || !n_layers)) return PO_ERR_INVALID;
    if (in->B == 0) return PO_OK;
    const size_t bl = sizeof(double) * (size_t)in->B * L, bs = sizeof(double) * 3 * (size_t)in->B, bb = sizeof(double) * (size_t)in->B, bi = sizeof(int) * (size_t)in->B;
    StagedSpline S{};
    {
        std::lock_guard<std::mutex> g(h->mu);
        HIP_TRY(hipSetDevice(h->device));
        if (int rc = stage_spline_batch(h, in, 3 * bl + bs + bb + bi, &S)) return rc;
        HIP_TRY(hipMemcpyAsync(S.next, start, bs, hipMemcpyHostToDevice, h->stream));
    }
    double *dstart = reinterpret_cast<double *>(S.next);
    double *o = dstart + 3 * (size_t)in->B;
    const size_t n1 = (size_t)in->B * L;
    double *dl0 = o + 3 * n1;
    int *dn = reinterpret_cast<int *>(dl0 + in->B);
    if (int rc = po_dp_search_batch_device(h, &S.d, dstart, L, o, o + n1, o + 2 * n1, dl0, dn)) return rc;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(hipMemcpyAsync(layer_s, o, bl, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(lb, o + n1, bl, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ub, o + 2 * n1, bl, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(l0, dl0, bb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(n_layers, dn, bi, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

int po_map_sample(po_handle h, int n, const double *xy, double *dist, int *inside) {
    if (!h || n < 0 || (n > 0 && (!xy || !dist || !inside))) return PO_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->map.d) return PO_ERR_INVALID;
    if (n == 0) return PO_OK;
    HIP_TRY(hipSetDevice(h->device));
    const size_t bx = sizeof(double) * 2 * (size_t)n, bd = sizeof(double) * (size_t)n, bi = sizeof(int) * (size_t)n;
    if (int rc = h->post_buf.ensure(bx + bd + bi)) return rc;
    char *base = static_cast<char *>(h->post_buf.p);
    HIP_TRY(hipMemcpyAsync(base, xy, bx, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(po_launch_map_sample(&h->map, n, reinterpret_cast<const double *>(base), reinterpret_cast<double *>(base + bx),
                                 reinterpret_cast<int *>(base + bx + bd), h->stream));
    HIP_TRY(hipMemcpyAsync(dist, base + bx, bd, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(inside, base + bx + bd, bi, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PO_OK;
}

// internal accessors for po_plan.cpp (not part of include/po_hip.h)
void *po_internal_arena(po_handle h, size_t bytes) { return h->plan_arena.ensure(bytes) == PO_OK ? h->plan_arena.p : nullptr; }
void *po_internal_plan_coef(po_handle h, size_t bytes) { return h->plan_coef.ensure(bytes) == PO_OK ? h->plan_coef.p : nullptr; }
hipStream_t po_internal_stream(po_handle h) { return h->stream; }
int po_internal_device(po_handle h) { return h->device; }
const po_params *po_internal_params(po_handle h) { return &h->params; }
int po_internal_has_map(po_handle h) { return h->map.d != nullptr; }
int po_internal_hip_fail(hipError_t e, const char *what) { return hip_ok(e, what) ? 0 : 1; }
void *po_internal_plan_host(po_handle h, size_t bytes) { return h->plan_host.ensure(bytes) == PO_OK ? h->plan_host.p : nullptr; }
std::mutex *po_internal_plan_mutex(po_handle h) { return &h->plan_mu; }

const char *po_strerror(int code) {
    switch (code) {
        case PO_OK: return "ok";
        case PO_ERR_INVALID: return "invalid argument";
        case PO_ERR_HIP: return "HIP runtime error (see po_last_hip_error)";
        case PO_ERR_UNSUPPORTED: return "unsupported configuration (path too long for the on-chip tile)";
        case PO_ERR_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}
const char *po_last_hip_error(void) { return g_hip_err.c_str(); }
#define PO_STR2(x) #x
#define PO_STR(x) PO_STR2(x)
const char *po_version(void) { return "po_hip " PO_STR(PO_ABI_VERSION) " (gfx950)"; }

}  // extern "C"
