// po_scale.hpp — class-level ("structured") Ruiz equilibration, device side.
//
// The reference runs OSQP with its default scaling = 10 Ruiz passes (it only touches verbosity/warm_start,
// /root/reference/src/solver/solver.cpp:48-49).  On these QPs every Ruiz pass returns the SAME factor for all
// variables / rows of one kind (the pattern repeats along the path and the data-dependent entries never attain a
// column's inf-norm), so the identical iteration is run ONCE per path on a one-stage template with the path's
// nominal arc-length step (max of the first <= 9 gaps — the reference's own reference_interval_, solver.cpp:22-27).
// In unscaled variables a Ruiz-scaled OSQP is plain ADMM with per-row step rho_j * E_j^2 / c and per-variable
// proximal weight sigma / (c D_k^2), which is how the solve kernel consumes the result.
#pragma once
#include "po_device.hpp"

namespace po {

constexpr int kNVC = 8;    // variable classes: 0 e_y, 1 e_phi, 2 c, 3 s1, 4 s2, 5 u, 6 su (K: first / last steering variable), 7 dead
constexpr int kNRC = 24;   // row classes, order: [local | dyn | ctl | end]
constexpr int kScStride = 64;
constexpr int kScW = 0, kScE = 24, kScSig = 48, kScCD = 56, kScC = 63;
// K only: the first and the last steering variable carry w_c + w_cr on the diagonal of R instead of w_c + 2 w_cr
// (solver_k_as_input.cpp:62-76), so OSQP's Ruiz passes give them — and their box rows — their own factors.  They are variable
// class 6 and row class kKEndRow; a second copy of the stage-local W / E entries with the box row replaced sits kScAlt
// entries further on, and the kernel points the row functors of the stages 0 and N-2 at it (Fast::Wloc(j) / Eloc(j)).
constexpr int kKEndRow = 11, kKEndVar = 6, kScAlt = 12;
constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4;

struct ClassModel {
    int nr;
    double Pmax[kNVC], cnt[kNVC];
    double a[kNRC][kNVC];
    int tgt[kNRC];
};

// Records the coefficient pattern of each row class by running the row generators on a template stage.
struct PatFn {
    ClassModel *M;
    int base;
    template <int MASK, class TL, class TU> __host__ __device__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL, TU) {
        double *a = M->a[base + r];
        a[0] = c0; a[1] = c1; a[2] = c2; a[3] = c3; a[4] = c4;
    }
};
struct PatCtlFn {  // control rows use slots (.,.,u,su,.) -> variable classes 5, 6
    ClassModel *M;
    int base;
    template <int MASK, class TL, class TU> __host__ __device__ void row(int r, double, double, double c2, double c3, double, TL, TU) {
        double *a = M->a[base + r];
        a[5] = c2; a[6] = c3;
    }
};

__device__ inline double limit_scaling(double v) {
    v = v < kMinScaling ? 1.0 : v;
    return v > kMaxScaling ? kMaxScaling : v;
}

template <int F> __device__ void class_scaling(const DevParams &P, int N, int keep, int C, double ds_nom, int passes, double *out /*[kScStride]*/) {
#pragma clang fp contract(off)
    using T = FormTraits<F>;
    ClassModel M;
    M.nr = T::NLOC + T::NDYN + T::NCTL + T::NEND;
    for (int r = 0; r < kNRC; ++r) { M.tgt[r] = -1; for (int v = 0; v < kNVC; ++v) M.a[r][v] = 0; }
    StageIn s;
    for (int c = 0; c < 4; ++c) { s.lb[c] = -1; s.ub[c] = 1; }
    s.maxk = 0; s.elo = -kInf; s.ehi = kInf; s.j = 0; s.N = N; s.last = 0;
    PatFn pf{&M, 0};
    local_rows<F>(s, P, pf);
    const Dyn<F> d = make_dyn<F>(0.0, ds_nom, P);  // template: k_ref = 0
    for (int r = 0; r < T::NDYN; ++r) {
        double *a = M.a[T::NLOC + r];
        a[0] = d.f[r][0]; a[1] = d.f[r][1]; a[2] = d.f[r][2]; a[5] = d.beta[r];
        M.tgt[T::NLOC + r] = dyn_tau<F>(r);
    }
    if constexpr (T::NCTL > 0) { PatCtlFn cf{&M, T::NLOC + T::NDYN}; ctl_rows<F>(0.0, cf); }
    if constexpr (T::NEND > 0) { PatFn ef{&M, T::NLOC + T::NDYN + T::NCTL}; end_rows<F>(s, P, ef); }
    if constexpr (F == F_K) {
        static_assert(T::NLOC + T::NDYN == kKEndRow && kScAlt + T::NLOC <= 24, "K end classes");
        M.a[kKEndRow][kKEndVar] = M.a[2][2];                   // box row of an end steering variable
        M.a[T::NLOC + 0][kKEndVar] = M.a[T::NLOC + 0][2];      // its entry in the e_phi equation
        M.nr = kKEndRow + 1;
    }
    for (int v = 0; v < kNVC; ++v) { M.Pmax[v] = 0; M.cnt[v] = 0; }
    M.Pmax[0] = P.w_dev; M.Pmax[2] = (F == F_K) ? P.w_c + 2 * P.w_cr : P.w_c; M.Pmax[3] = P.w_s1;
    M.cnt[0] = N; M.cnt[1] = N; M.cnt[2] = N; M.cnt[3] = N;
    if constexpr (F == F_K) { M.Pmax[kKEndVar] = P.w_c + P.w_cr; M.cnt[kKEndVar] = N - 1 < 2 ? N - 1 : 2; M.cnt[2] = N - 1 - M.cnt[kKEndVar]; }
    if constexpr (F == F_KP) { M.Pmax[5] = P.w_u; M.cnt[5] = C; M.Pmax[7] = P.w_s1; M.cnt[7] = N; }
    if constexpr (F == F_KPC) { M.Pmax[4] = P.w_s2; M.cnt[4] = N; M.Pmax[5] = P.w_u; M.cnt[5] = C; M.Pmax[6] = P.w_su; M.cnt[6] = C; M.cnt[7] = N - C; }
    double Dv[kNVC], Er[kNRC], c = 1.0, ntot = 0;
    for (int v = 0; v < kNVC; ++v) { Dv[v] = 1.0; ntot += M.cnt[v]; }
    for (int r = 0; r < kNRC; ++r) Er[r] = 1.0;
    for (int pass = 0; pass < passes; ++pass) {
        double cn[kNVC], rn[kNRC];
        for (int v = 0; v < kNVC; ++v) cn[v] = fabs(c * M.Pmax[v] * Dv[v] * Dv[v]);
        for (int r = 0; r < M.nr; ++r) {
            double rmax = 0;
            for (int v = 0; v < kNVC; ++v) {
                const double a = fabs(Er[r] * M.a[r][v] * Dv[v]);
                if (a > rmax) rmax = a;
                if (a > cn[v]) cn[v] = a;
            }
            if (M.tgt[r] >= 0) {
                const double a = fabs(Er[r] * Dv[M.tgt[r]]);
                if (a > rmax) rmax = a;
                if (a > cn[M.tgt[r]]) cn[M.tgt[r]] = a;
            }
            rn[r] = rmax;
        }
        for (int v = 0; v < kNVC; ++v) Dv[v] *= 1.0 / sqrt(limit_scaling(cn[v]));
        for (int r = 0; r < M.nr; ++r) Er[r] *= 1.0 / sqrt(limit_scaling(rn[r]));
        double mean = 0;
        for (int v = 0; v < kNVC; ++v) mean += M.cnt[v] * fabs(c * M.Pmax[v] * Dv[v] * Dv[v]);
        mean /= ntot;
        double ct = mean > 1.0 ? mean : 1.0;  // ||q||_inf = 0 -> limit_scaling -> 1
        ct = 1.0 / limit_scaling(ct);
        c *= ct;
    }
    for (int i = 0; i < kScStride; ++i) out[i] = 0;
    for (int r = 0; r < kNRC; ++r) { out[kScW + r] = Er[r] * Er[r] / c; out[kScE + r] = Er[r]; }
    for (int v = 0; v < 7; ++v) { out[kScSig + v] = P.sigma / (c * Dv[v] * Dv[v]); out[kScCD + v] = c * Dv[v]; }
    if constexpr (F == F_K) {
        for (int r = 0; r < T::NLOC; ++r) { out[kScW + kScAlt + r] = out[kScW + (r == 2 ? kKEndRow : r)]; out[kScE + kScAlt + r] = out[kScE + (r == 2 ? kKEndRow : r)]; }
    }
    out[kScC] = c;
}

// one thread per path
template <int F> __global__ void scale_kernel(DevBatch in, DevParams P, int passes, double *sc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= in.B) return;
    const double *s = in.ref_s + (size_t)b * in.N;
    int N = in.N, C = in.C;
    if (in.n_points) { N = in.n_points[b]; N = N < 2 ? 2 : (N > in.N ? in.N : N); C = (F == F_K) ? N - 1 : (N + in.keep - 2) / in.keep; }
    double ds_nom = 0;  // solver.cpp:22-27
    for (int i = 1; i < N && i < 10; ++i) ds_nom = fmax(ds_nom, __dsub_rn(s[i], s[i - 1]));
    class_scaling<F>(P, N, in.keep, C, ds_nom, passes, sc + (size_t)b * kScStride);
}

}  // namespace po
