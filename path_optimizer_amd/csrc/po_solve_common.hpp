// po_solve_common.hpp — what every translation unit that sees the solve kernels shares: the per-path context, the fused solve kernel
// (po_fast.inc) and its launch / shape-selection wrappers.  The instantiations themselves are compiled per formulation
// (po_solve_form.hip, -DPO_FORM=0/1/2) so that the three sets build in parallel; po_kernels.hip holds the dispatcher.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/po_hip.h"
#include "po_device.hpp"
#include "po_scale.hpp"

namespace po {

// -------------------------------------------------------------------------------------------------------
// per-path context
// -------------------------------------------------------------------------------------------------------
template <int F> struct Ctx {
    using T = FormTraits<F>;
    const DevParams &P;
    const DevBatch &in;
    double *S;  // LDS base (unused by the diagnostic kernel)
    int N, C, keep, lane;
    size_t po;  // path offset (b*N)
    int b;
    double elo, ehi;

    __device__ Ctx(const DevParams &P_, const DevBatch &in_, double *S_, int b_)
        : P(P_), in(in_), S(S_), N(in_.N), C(in_.C), keep(in_.keep), lane(threadIdx.x), po((size_t)b_ * in_.N), b(b_) {
        // end-heading window: solver_kp_as_input.cpp:193-202 (signed test, preserved)
        elo = -kInf;
        ehi = kInf;
        if (P.end_heading) {
            const double psi = wrap_angle(in.goal_z[b] - in.ref_z[po + N - 1]);
            if (psi < 70 * kPi / 180) {
                elo = psi - 5 * kPi / 180;
                ehi = psi + 5 * kPi / 180;
            }
        }
    }
    __device__ __forceinline__ StageIn stage_in(int j) const {
        StageIn s;
        const double4 *bp = reinterpret_cast<const double4 *>(in.bounds + (po + j) * 8);
        const double4 b0 = bp[0], b1 = bp[1];
        s.lb[0] = b0.x; s.ub[0] = b0.y; s.lb[1] = b0.z; s.ub[1] = b0.w;
        s.lb[2] = b1.x; s.ub[2] = b1.y; s.lb[3] = b1.z; s.ub[3] = b1.w;
        s.maxk = (F == F_KPC) ? in.max_k[po + j] : 0.0;
        s.elo = elo; s.ehi = ehi; s.j = j; s.N = N; s.last = (j == N - 1);
        return s;
    }
    __device__ __forceinline__ Dyn<F> dyn(int i) const {  // transition i -> i+1
        const double k = in.ref_k[po + i];
        const double ds = __dsub_rn(in.ref_s[po + i + 1], in.ref_s[po + i]);
        return make_dyn<F>(k, ds, P);
    }
    __device__ __forceinline__ void init_bounds(double bnd[3]) const {
        // rows 0..2: -X_0 = -x0 (solver_kp_as_input.cpp:143-147); K: x0 = (heading_err, offset) (:154-158)
        if constexpr (F == F_K) {
            bnd[0] = -in.x0[b * 3 + 1];
            bnd[1] = -in.x0[b * 3 + 0];
        } else {
            bnd[0] = -in.x0[b * 3 + 0];
            bnd[1] = -in.x0[b * 3 + 1];
            bnd[2] = -in.x0[b * 3 + 2];
        }
    }
};

__device__ __forceinline__ void inv3_sym(const double g[6], double gi[6]) {
    // g = [g00 g01 g02 g11 g12 g22]; SPD
    const double c00 = g[3] * g[5] - g[4] * g[4];
    const double c01 = g[2] * g[4] - g[1] * g[5];
    const double c02 = g[1] * g[4] - g[2] * g[3];
    const double det = g[0] * c00 + g[1] * c01 + g[2] * c02;
    const double id = 1.0 / det;
    gi[0] = c00 * id;
    gi[1] = c01 * id;
    gi[2] = c02 * id;
    gi[3] = (g[0] * g[5] - g[2] * g[2]) * id;
    gi[4] = (g[1] * g[2] - g[0] * g[4]) * id;
    gi[5] = (g[0] * g[3] - g[1] * g[1]) * id;
}


struct Resid { double rp, rd, nAx, nz, nPx, nAty, rps, rds, nAxs, nzs, nPxs, nAtys, obj; int bad; };  // obj = 0.5 x'Px (q = 0)


#include "po_fast.inc"

}  // namespace po

// ---- launch wrappers used by the C ABI (po_capi.cpp) ----
namespace po {
template <class K> hipError_t launch_grid(K kern, const DevBatch *in, const DevParams *P, int grid, int nt, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, st, *in, *P);
    return hipGetLastError();
}
template <class K> hipError_t launch1(K kern, const DevBatch *in, const DevParams *P, int nt, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(in->rq != nullptr ? in->B + in->rq_cap : in->B), dim3(nt), lds, st, *in, *P);
    return hipGetLastError();
}
// thread-block shape: NT threads x SPL stages per thread must cover N, and NT >= C (one control per thread).
// Two-level mode needs chunk == control group (SPL == keep), or no held controls at all (K).
struct Shape { int nt, spl; bool two; };
inline bool pick_shape(int form, int N, int C, int keep, Shape *s) {
    const bool no_u = (form == F_K);
    if (no_u) C = 0;  // K has no held controls: its N-1 steering variables live inside the nodes
    // two-level path: one chunk of `spl` stages per thread; with held controls the chunk IS the control group (spl == keep).
    // keep 1..8 covers what the reference produces (spacing 0.15..1.0 m, path_optimizer.cpp:171-172); KPC is keep == 4 only.
    const int keep_max = form == F_KP ? 8 : 4;
    if (no_u || (keep >= 1 && keep <= keep_max)) {
        const int spl = no_u ? (N <= 128 ? 2 : 4) : keep;
        for (int nt : {64, 128, 256}) {
            if (nt == 256 && spl != 1) break;
            if (nt == 128 && spl > 4) break;
            if (N <= nt * spl && C <= nt) { *s = {nt, spl, true}; return true; }
        }
    }
    const int cand[5][2] = {{64, 2}, {64, 4}, {128, 4}, {256, 2}, {256, 4}};
    for (auto &c : cand)
        if (N <= c[0] * c[1] && C <= c[0]) { *s = {c[0], c[1], false}; return true; }
    return false;
}
inline size_t lds_of(int form, int N, int C, const Shape &s) {
    return form == F_KP ? lds_bytes_fast<F_KP>(N, C, s.spl, s.two, s.nt) : (form == F_KPC ? lds_bytes_fast<F_KPC>(N, C, s.spl, s.two, s.nt) : lds_bytes_fast<F_K>(N, C, s.spl, s.two, s.nt));
}
// pick_shape, then fall back to the single-level path when the two-level tables of a long, finely chunked path
// (prefix products: 9 * chunks * log2(chunks) doubles) exceed the 160 KB of LDS
inline bool resolve_shape(int form, int N, int C, int keep, Shape *s) {
    if (!pick_shape(form, N, C, keep, s)) return false;
    if (s->two && lds_of(form, N, C, *s) > 160 * 1024) return pick_shape(form, N, C, /*keep (forces the single-level candidates)*/ 0, s);
    return true;
}
// UNI = true: the uniform-row-class variant of the two-level kernels (no-op for shapes that use the single-level mapping and for K on multi-wave blocks); UNI = false: the general variant.  po_launch_solve issues them in this order on one stream.
template <int F> inline bool has_uni_variant(const Shape &s) { return s.two && (F != F_K || s.nt == 64); }  // K: one-wave blocks only (Fast::classify)
// REF: kernels that carry the refinement phase (po_params.refine) around the loop — every variant exists with and without (the phase costs the hot loop
// a few % even when it is not taken); those with are launched only when po_params.refine is set.
template <int F, bool UNI, int REF> hipError_t launch_form(const DevBatch *in_, const DevParams *P, hipStream_t st, size_t *lds_out) {
    Shape s;
    if (!resolve_shape(F, in_->N, in_->C, in_->keep, &s)) return hipErrorInvalidValue;
    const size_t lds = lds_bytes_fast<F>(in_->N, in_->C, s.spl, s.two, s.nt);
    if (lds_out) *lds_out = lds;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    DevBatch copy = *in_;
    copy.only_deferred = (!UNI && has_uni_variant<F>(s)) ? 1 : 0;
    const DevBatch *in = &copy;
#define PO_L(SPL_, NT_, TWO_) return launch1(&solve_kernel_fast<F, SPL_, NT_, TWO_, UNI && TWO_, REF>, in, P, NT_, lds, st)
    if constexpr (UNI) {
        if (!has_uni_variant<F>(s)) return hipSuccess;
    }
#ifdef PO_WITH_SPLIT
    if constexpr (UNI && F != F_K) {
        // keep == 4 on one-wave blocks (BASELINE configs 1-3): the stage-split two-wave mapping (Fast<..., NW = 2>): 2 stages per lane, <= 256 registers, two waves per SIMD
        if (in->use_split && s.two && s.spl == 4 && s.nt == 64 && in->keep == 4) {
            const size_t lds2 = lds_bytes_fast<F>(in->N, in->C, 2, true, 128, 2);
            if (lds_out) *lds_out = lds2;
            if (lds2 <= 160 * 1024) return launch1(&solve_kernel_split<F>, in, P, 128, lds2, st);
        }
    }
#endif
#ifdef PO_DEV_HEADLINE  // dev builds: only the BASELINE config-3 variant (seconds to compile); -DPO_DEV_SPL=k: the one-wave variant of keep k instead
#ifndef PO_DEV_SPL
#define PO_DEV_SPL 4
#endif
    if (s.two && s.spl == PO_DEV_SPL && s.nt == 64) PO_L(PO_DEV_SPL, 64, true);
    return hipErrorInvalidValue;
#else
    if (s.two) {
        if constexpr (F == F_KP) {
            if (s.spl == 1 && s.nt == 64) PO_L(1, 64, true);
            if (s.spl == 1 && s.nt == 128) PO_L(1, 128, true);
            if (s.spl == 1) PO_L(1, 256, true);
            if (s.spl == 5) PO_L(5, 64, true);
            if (s.spl == 6) PO_L(6, 64, true);
            if (s.spl == 7) PO_L(7, 64, true);
            if (s.spl == 8) PO_L(8, 64, true);
        }
        if (s.spl == 2 && s.nt == 64) PO_L(2, 64, true);
        if (s.spl == 2) PO_L(2, 128, true);
        if (s.spl == 3 && s.nt == 64) PO_L(3, 64, true);
        if (s.spl == 3) PO_L(3, 128, true);
        if (s.nt == 64) PO_L(4, 64, true);
        PO_L(4, 128, true);
    }
    if constexpr (!UNI) {
        if (s.nt == 64 && s.spl == 2) PO_L(2, 64, false);
        if (s.nt == 64) PO_L(4, 64, false);
        if (s.nt == 128) PO_L(4, 128, false);
        if (s.spl == 2) PO_L(2, 256, false);
        PO_L(4, 256, false);
    }
    return hipErrorInvalidValue;
#endif
#undef PO_L
}
// ---- polish (po_params.polish): same shape as the solve launch; two-level shapes only (every case the reference produces) ----
template <int F, int SPL_, int NT_> inline int state_doubles_of() { return Fast<F, SPL_, NT_, true>::kStateDoubles * NT_; }
#define PO_POLISH_SHAPES(X)                                                                          \
    if constexpr (F == F_KP) {                                                                       \
        if (s.spl == 1 && s.nt == 64) X(1, 64); if (s.spl == 1 && s.nt == 128) X(1, 128); if (s.spl == 1) X(1, 256); \
        if (s.spl == 5) X(5, 64); if (s.spl == 6) X(6, 64); if (s.spl == 7) X(7, 64); if (s.spl == 8) X(8, 64);       \
    }                                                                                                \
    if (s.spl == 2 && s.nt == 64) X(2, 64); if (s.spl == 2) X(2, 128);                               \
    if (s.spl == 3 && s.nt == 64) X(3, 64); if (s.spl == 3) X(3, 128);                               \
    if (s.nt == 64) X(4, 64); X(4, 128);
// doubles per path of the state block the solve kernels leave for the polish (0: shape without a polish kernel)
template <int F> inline int polish_state_doubles(int N, int C, int keep) {
    Shape s;
    if (!resolve_shape(F, N, C, keep, &s) || !s.two) return 0;
#ifdef PO_DEV_HEADLINE
    return (s.spl == 4 && s.nt == 64) ? state_doubles_of<F, 4, 64>() : 0;
#else
#define PO_X(SPL_, NT_) return state_doubles_of<F, SPL_, NT_>()
    PO_POLISH_SHAPES(PO_X)
#undef PO_X
    return 0;
#endif
}
// the Newton refinement of round 0 as its own launch (po_params.refine = 2, refine_chain = 2 / 3): same shapes as the polish.  FB: the fallback launch behind it
// (newton_fallback_kernel walks the work list of the paths newton_kernel handed back; a small fixed grid)
template <int F, bool FB> hipError_t launch_newton(const DevBatch *in, const DevParams *P, hipStream_t st) {
    Shape s;
    if (!resolve_shape(F, in->N, in->C, in->keep, &s) || !s.two) return hipErrorInvalidValue;
    const size_t lds = lds_bytes_fast<F>(in->N, in->C, s.spl, s.two, s.nt);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define PO_X(SPL_, NT_) { if constexpr (FB) return launch_grid(&newton_fallback_kernel<F, SPL_, NT_>, in, P, kFallbackGrid, NT_, lds, st); else return launch1(&newton_kernel<F, SPL_, NT_>, in, P, NT_, lds, st); }
#ifdef PO_DEV_HEADLINE
    if (s.spl == 4 && s.nt == 64) PO_X(4, 64)
    return hipErrorInvalidValue;
#else
    PO_POLISH_SHAPES(PO_X)
    return hipErrorInvalidValue;
#endif
#undef PO_X
}
template <int F> hipError_t launch_polish(const DevBatch *in, const DevParams *P, hipStream_t st) {
    Shape s;
    if (!resolve_shape(F, in->N, in->C, in->keep, &s) || !s.two) return hipErrorInvalidValue;
    const size_t lds = lds_bytes_fast<F>(in->N, in->C, s.spl, s.two, s.nt);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#ifdef PO_DEV_HEADLINE
    if (s.spl == 4 && s.nt == 64) return launch1(&polish_kernel<F, 4, 64>, in, P, 64, lds, st);
    return hipErrorInvalidValue;
#else
#define PO_X(SPL_, NT_) return launch1(&polish_kernel<F, SPL_, NT_>, in, P, NT_, lds, st)
    PO_POLISH_SHAPES(PO_X)
#undef PO_X
    return hipErrorInvalidValue;
#endif
}
}  // namespace po

