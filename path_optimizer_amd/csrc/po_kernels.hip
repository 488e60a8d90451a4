// po_kernels.hip — gfx950 (MI355X) kernels of libpo_hip.so.
//
//   scale_kernel    (po_scale.hpp)  per-path class-level Ruiz equilibration (OSQP's default scaling = 10)
//   solve_kernel_fast (po_fast.inc) the fused batched QP solve: assembly + factorisation + ADMM + output map
//                                   in ONE launch, per-path state resident in VGPRs / LDS
//   assemble_kernel                 diagnostic: the assembled bounds / data-dependent A entries in the
//                                   reference row order, for bit-exact comparison with the oracle
//
// Replaces, per path, the work behind /root/reference/src/solver/solver.cpp:46-77 (setHessianMatrix,
// setConstraintMatrix, osqp_setup, osqp_solve, getOptimizedPath).
#include "po_solve_common.hpp"

namespace po {

// diagnostic assembly kernel: l,u in reference row order + data-dependent A entries per transition
template <int F> __global__ __launch_bounds__(64) void assemble_kernel(DevBatch in, DevParams P, double *l, double *u, double *dyn) {
    using T = FormTraits<F>;
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    Ctx<F> cx(P, in, smem, b);
    const int N = cx.N, C = cx.C;
    double *lb = l + (size_t)b * in.m, *ub = u + (size_t)b * in.m;
    for (int j = cx.lane; j < N; j += 64) {
        const StageIn si = cx.stage_in(j);
        AsmFn<F> fn{lb, ub, j, N, C, 0};
        local_rows<F>(si, P, fn);
        if (T::NEND && si.last) { fn.kind = 1; end_rows<F>(si, P, fn); }
        double bin[3];
        if (j == 0) cx.init_bounds(bin);
        else {
            const Dyn<F> d = cx.dyn(j - 1);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) bin[r] = d.b[r];
            double *dd = dyn + ((size_t)b * (N - 1) + (j - 1)) * 3;
            if constexpr (F == F_K) { dd[0] = d.f[0][0]; dd[1] = d.f[1][1]; dd[2] = d.f[0][2]; }
            else { dd[0] = d.f[0][1]; dd[1] = d.f[1][0]; dd[2] = d.beta[2]; }
        }
#pragma unroll
        for (int r = 0; r < T::NDYN; ++r) { lb[T::NDYN * j + r] = bin[r]; ub[T::NDYN * j + r] = bin[r]; }
        if constexpr (F == F_K) {
            if (si.last) { /* delta_{N-1} has no box row */ }
        }
    }
    if constexpr (T::HAS_U)
        for (int c = cx.lane; c < C; c += 64) {
            AsmFn<F> fn{lb, ub, c, N, C, 2};
            const double mkp = (F == F_KPC) ? in.max_kp[cx.po + c] : 0.0;
            ctl_rows<F>(mkp, fn);
        }
}

}  // namespace po

// the solve kernels of one formulation and one loop variant live in their own object (po_solve_form.hip)
#define PO_DECL(name) extern "C" hipError_t name(const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out)
PO_DECL(po_launch_solve_kp); PO_DECL(po_launch_solve_kp_uni);
PO_DECL(po_launch_solve_kp_w); PO_DECL(po_launch_solve_kp_w_uni);  // the wide role-split shapes of keep 9 .. 16 (objects of their own)
PO_DECL(po_launch_solve_kpc); PO_DECL(po_launch_solve_kpc_uni);
PO_DECL(po_launch_solve_k); PO_DECL(po_launch_solve_k_uni);
#undef PO_DECL

namespace po {
// After the launches of a solve that hands paths from launch to launch (the Newton refinement's rounds): no internal "in flight" status may reach the caller
// (a path that a malformed caller-side order skipped): anything at or below kStatusDeferred becomes UNSOLVED.
__global__ void finalize_status_kernel(po_info *info, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && info[b].status <= kStatusDeferred) { info[b].status = PO_STATUS_UNSOLVED; info[b].status_polish = 0; }
}
// A caller that asked for the Newton refinement (po_params.refine = 2) or the polish on a shape that has no kernel for it (the single-level mapping; polish: also the role-split
// shapes) can SEE that: status_refine / status_polish = -2 (PO_NOT_AVAILABLE) on every path instead of 0, which also means "off" (include/po_hip.h, ABI 6).
__global__ void mark_unavailable_kernel(po_info *info, int B, int refine, int polish) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        if (refine) info[b].status_refine = PO_NOT_AVAILABLE;
        if (polish) info[b].status_polish = PO_NOT_AVAILABLE;
    }
}
}  // namespace po
extern "C" hipError_t po_launch_mark_unavailable(po_info *info, int B, int refine, int polish, hipStream_t st) {
    hipLaunchKernelGGL(po::mark_unavailable_kernel, dim3((B + 255) / 256), dim3(256), 0, st, info, B, refine, polish);
    return hipGetLastError();
}
// Two launches on the same stream for the two-level mapping: the uniform-row-class variant first (solves what it can, defers the rest),
// then the general variant (po_fast.inc, solve_kernel_fast).
extern "C" hipError_t po_launch_solve(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out) {
    using namespace po;
    hipError_t e;
    if (form == F_KP) {
        e = po_launch_solve_kp_uni(in, P, st, lds_out);
        if (e == kNotMyShape) {  // not a shape of keep 1 .. 8: the wide objects
            e = po_launch_solve_kp_w_uni(in, P, st, lds_out);
            return e != hipSuccess ? e : po_launch_solve_kp_w(in, P, st, lds_out);
        }
        return e != hipSuccess ? e : po_launch_solve_kp(in, P, st, lds_out);
    }
    if (form == F_KPC) { e = po_launch_solve_kpc_uni(in, P, st, lds_out); return e != hipSuccess ? e : po_launch_solve_kpc(in, P, st, lds_out); }
    e = po_launch_solve_k_uni(in, P, st, lds_out);
    return e != hipSuccess ? e : po_launch_solve_k(in, P, st, lds_out);
}
// no internal "in flight" status reaches the caller (finalize_status_kernel)
extern "C" hipError_t po_launch_finalize_status(po_info *info, int B, hipStream_t st) {
    hipLaunchKernelGGL(po::finalize_status_kernel, dim3((B + 255) / 256), dim3(256), 0, st, info, B);
    return hipGetLastError();
}

#define PO_DECLP(name) extern "C" hipError_t name(const po::DevBatch *in, const po::DevParams *P, hipStream_t st)
PO_DECLP(po_launch_polish_kp); PO_DECLP(po_launch_polish_kpc); PO_DECLP(po_launch_polish_k);
#undef PO_DECLP
#define PO_DECLP(name) extern "C" hipError_t name(const po::DevBatch *in, const po::DevParams *P, hipStream_t st)
PO_DECLP(po_launch_newton_kp); PO_DECLP(po_launch_newton_kpc); PO_DECLP(po_launch_newton_k);
PO_DECLP(po_launch_newton_kp_fb); PO_DECLP(po_launch_newton_kpc_fb); PO_DECLP(po_launch_newton_k_fb);
PO_DECLP(po_launch_newton_kp_b); PO_DECLP(po_launch_newton_kp_b_fb);  // KP's role-split shapes (second Newton object)
PO_DECLP(po_launch_newton_kp_c); PO_DECLP(po_launch_newton_kp_c_fb);  // KP's multi-group shapes (third)
PO_DECLP(po_launch_newton_kp_w1); PO_DECLP(po_launch_newton_kp_w1_fb); PO_DECLP(po_launch_newton_kp_w2); PO_DECLP(po_launch_newton_kp_w2_fb); PO_DECLP(po_launch_newton_kp_w3); PO_DECLP(po_launch_newton_kp_w3_fb);  // keep 9 .. 16
#undef PO_DECLP
// the Newton refinement of round 0 as its own launch (po_params.refine = 2), and the fallback launch for what it hands back
extern "C" hipError_t po_launch_newton(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st) {
    using namespace po;
    if (form == F_KP) {  // (an object answers kNotMyShape for a shape it does not hold; any other code is that launch's own failure and is returned as it is)
        hipError_t e = po_launch_newton_kp(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_b(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_c(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_w1(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_w2(in, P, st);
        return e == kNotMyShape ? po_launch_newton_kp_w3(in, P, st) : e;
    }
    return form == F_KPC ? po_launch_newton_kpc(in, P, st) : po_launch_newton_k(in, P, st);
}
extern "C" hipError_t po_launch_newton_fallback(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st) {
    using namespace po;
    if (form == F_KP) {
        hipError_t e = po_launch_newton_kp_fb(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_b_fb(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_c_fb(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_w1_fb(in, P, st);
        if (e == kNotMyShape) e = po_launch_newton_kp_w2_fb(in, P, st);
        return e == kNotMyShape ? po_launch_newton_kp_w3_fb(in, P, st) : e;
    }
    return form == F_KPC ? po_launch_newton_kpc_fb(in, P, st) : po_launch_newton_k_fb(in, P, st);
}
extern "C" int po_polish_state_doubles_kp(int N, int C, int keep);
extern "C" int po_polish_state_doubles_kpc(int N, int C, int keep);
extern "C" int po_polish_state_doubles_k(int N, int C, int keep);
// OSQP's polish (po_params.polish) on the paths the two solve launches reported solved; same stream, after them
extern "C" hipError_t po_launch_polish(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st) {
    using namespace po;
    return form == F_KP ? po_launch_polish_kp(in, P, st) : (form == F_KPC ? po_launch_polish_kpc(in, P, st) : po_launch_polish_k(in, P, st));
}
extern "C" int po_polish_state_doubles_kp_park(int N, int C, int keep);
extern "C" int po_polish_state_doubles_kp_w(int N, int C, int keep);
extern "C" int po_polish_state_doubles_kp_w_park(int N, int C, int keep);
extern "C" int po_polish_state_doubles_kpc_park(int N, int C, int keep);
extern "C" int po_polish_state_doubles_k_park(int N, int C, int keep);
extern "C" int po_newton_park_doubles(int form, int N, int C, int keep) {
    using namespace po;
    if (form == F_KP) { const int d = po_polish_state_doubles_kp_park(N, C, keep); return d ? d : po_polish_state_doubles_kp_w_park(N, C, keep); }
    return form == F_KPC ? po_polish_state_doubles_kpc_park(N, C, keep) : po_polish_state_doubles_k_park(N, form == F_K ? 0 : C, keep);
}
namespace po {
// the parked paths (keys[b] >= 0) in descending key order, ties in path order (a stable counting sort: deterministic): list[0] = count, list[1 ..] = path ids.
// One workgroup of kNwSortThreads threads, thread t owns the contiguous range of paths [t * per, (t + 1) * per).
constexpr int kNwKeys = 32, kNwSortThreads = 256;  // (32 keys x 257 counters = 33 KB of LDS)
__global__ __launch_bounds__(kNwSortThreads) void nw_sort_kernel(const int *keys, int B, int *list) {
    // (round 6) a thread's counters live in its own COLUMN of the LDS table (no conflicts, one read-modify-write per path instead of a 32-way compare over registers), its
    // first 16 keys are loaded at once and kept for the scatter pass, and the rows are scanned by whole waves: 22 -> ~10 us for 4 096 paths
    __shared__ int cnt[kNwKeys][kNwSortThreads + 1];
    __shared__ int rowbase[kNwKeys];
    const int t = threadIdx.x, per = (B + kNwSortThreads - 1) / kNwSortThreads, lo = t * per, hi = min(B, lo + per);
    constexpr int kCache = 16;
    int kc[kCache];
#pragma unroll
    for (int i = 0; i < kCache; ++i) kc[i] = (lo + i < hi) ? keys[lo + i] : -1;
#pragma unroll
    for (int k = 0; k < kNwKeys; ++k) cnt[k][t] = 0;
#pragma unroll
    for (int i = 0; i < kCache; ++i)
        if (kc[i] >= 0) cnt[kc[i] < kNwKeys ? kc[i] : kNwKeys - 1][t] += 1;
    for (int b = lo + kCache; b < hi; ++b) {
        const int k = keys[b];
        if (k >= 0) cnt[k < kNwKeys ? k : kNwKeys - 1][t] += 1;
    }
    __syncthreads();
    // exclusive prefix over (key descending, thread ascending): per key a scan of its row of kNwSortThreads counters (lane l takes kNwSortThreads / 64 consecutive counters, a
    // 6-step shuffle scan over the lanes' sums), then the rows are offset by the totals of the higher keys
    {
        constexpr int kPerLane = kNwSortThreads / 64, kWaves = kNwSortThreads / 64;
        const int lane = t & 63, wv = t >> 6;
        for (int k = wv; k < kNwKeys; k += kWaves) {
            int v[kPerLane], sum = 0;
#pragma unroll
            for (int i = 0; i < kPerLane; ++i) { v[i] = cnt[k][lane * kPerLane + i]; sum += v[i]; }
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            int run = incl - sum;
#pragma unroll
            for (int i = 0; i < kPerLane; ++i) { cnt[k][lane * kPerLane + i] = run; run += v[i]; }
            if (lane == 63) cnt[k][kNwSortThreads] = incl;
        }
    }
    __syncthreads();
    if (t < kNwKeys) {
        int run = 0;
        for (int j = kNwKeys - 1; j > t; --j) run += cnt[j][kNwSortThreads];
        rowbase[t] = run;
        if (t == 0) list[0] = run + cnt[0][kNwSortThreads];
    }
    __syncthreads();
    auto place = [&](int b, int k) {  // (own column again: the next free position of this thread's paths with this key)
        const int kk = k < kNwKeys ? k : kNwKeys - 1;
        const int pos = rowbase[kk] + cnt[kk][t];
        cnt[kk][t] += 1;
        list[1 + pos] = b;
    };
#pragma unroll
    for (int i = 0; i < kCache; ++i)
        if (kc[i] >= 0) place(lo + i, kc[i]);
    for (int b = lo + kCache; b < hi; ++b) {
        const int k = keys[b];
        if (k >= 0) place(b, k);
    }
}

}  // namespace po
// sliced Newton launches: the parked paths ordered by expected remaining work (one small workgroup; ~10 us)
extern "C" hipError_t po_launch_nw_sort(const int *keys, int B, int *list, hipStream_t st) {
    hipLaunchKernelGGL(po::nw_sort_kernel, dim3(1), dim3(po::kNwSortThreads), 0, st, keys, B, list);
    return hipGetLastError();
}
extern "C" int po_polish_state_doubles(int form, int N, int C, int keep) {
    using namespace po;
    if (form == F_KP) { const int d = po_polish_state_doubles_kp(N, C, keep); return d ? d : po_polish_state_doubles_kp_w(N, C, keep); }
    return form == F_KPC ? po_polish_state_doubles_kpc(N, C, keep) : po_polish_state_doubles_k(N, form == F_K ? 0 : C, keep);
}
// threads per path of the shape this batch runs in (0: unsupported)
extern "C" int po_shape_threads(int form, int N, int C, int keep) {
    po::Shape s;
    return po::resolve_shape(form, N, form == po::F_K ? 0 : C, keep, &s) ? s.nt : 0;
}
extern "C" int po_has_polish_kernel_kp(int N, int C, int keep);
extern "C" int po_has_polish_kernel_kpc(int N, int C, int keep);
extern "C" int po_has_polish_kernel_k(int N, int C, int keep);
// does the shape of this batch have a polish kernel?  (the role-split shapes of keep 6 .. 16 and the single-level mapping do not)
extern "C" int po_has_polish_kernel(int form, int N, int C, int keep) {
    using namespace po;
    return form == F_KP ? po_has_polish_kernel_kp(N, C, keep) : (form == F_KPC ? po_has_polish_kernel_kpc(N, C, keep) : po_has_polish_kernel_k(N, form == F_K ? 0 : C, keep));
}

extern "C" hipError_t po_launch_scale(int form, const po::DevBatch *in, const po::DevParams *P, int passes, double *sc, hipStream_t st) {
    using namespace po;
    const int bs = 64, gs = (in->B + bs - 1) / bs;
    if (form == F_KP) hipLaunchKernelGGL(scale_kernel<F_KP>, dim3(gs), dim3(bs), 0, st, *in, *P, passes, sc);
    else if (form == F_KPC) hipLaunchKernelGGL(scale_kernel<F_KPC>, dim3(gs), dim3(bs), 0, st, *in, *P, passes, sc);
    else hipLaunchKernelGGL(scale_kernel<F_K>, dim3(gs), dim3(bs), 0, st, *in, *P, passes, sc);
    return hipGetLastError();
}

extern "C" hipError_t po_launch_assemble(int form, const po::DevBatch *in, const po::DevParams *P, double *l, double *u, double *dyn, hipStream_t st) {
    using namespace po;
    if (form == F_KP) hipLaunchKernelGGL(assemble_kernel<F_KP>, dim3(in->B), dim3(64), 0, st, *in, *P, l, u, dyn);
    else if (form == F_KPC) hipLaunchKernelGGL(assemble_kernel<F_KPC>, dim3(in->B), dim3(64), 0, st, *in, *P, l, u, dyn);
    else hipLaunchKernelGGL(assemble_kernel<F_K>, dim3(in->B), dim3(64), 0, st, *in, *P, l, u, dyn);
    return hipGetLastError();
}

extern "C" size_t po_lds_bytes(int form, int N, int C, int keep) {
    using namespace po;
    Shape s;
    if (!resolve_shape(form, N, C, keep, &s)) return (size_t)1 << 30;
    return lds_of(form, N, C, s);
}

