// po_kernels.hip — fused batched QP solve for gfx950 (MI355X): assembly + factorisation + ADMM + output
// map in ONE kernel, one 64-lane wavefront per path, all solver state resident in LDS.
//
// Replaces, per path, the work behind /root/reference/src/solver/solver.cpp:46-77:
//   setHessianMatrix / setConstraintMatrix   -> rows regenerated on the fly (po_device.hpp), never stored
//   osqp_setup (KKT + LDL')                   -> factor(): block LDL' of the reduced SPD system
//                                                M = P + sigma I + A' diag(rho) A in stage order
//   osqp_solve (ADMM loop)                    -> iterate(): rhs pass, chain solve, update pass,
//                                                residual pass every check_every iterations, adaptive rho
//   getOptimizedPath                          -> output pass
// The iterates are those of OSQP's ADMM (no Ruiz scaling) in exact arithmetic; see DESIGN.md §4 for the
// "v-form" state (v = z + y/rho; z = clip(v), y = rho (v - z)) that halves the per-row state.
#include <hip/hip_runtime.h>

#include "../../include/po_hip.h"
#include "po_device.hpp"

namespace po {

// -------------------------------------------------------------------------------------------------------
// LDS layout (doubles).  SoA [component][stage] for everything touched by the lane-parallel passes
// (conflict-free ds_read_b64), AoS per stage for the factor (broadcast reads in the chain).
// -------------------------------------------------------------------------------------------------------
template <int F> struct Layout {
    using T = FormTraits<F>;
    int N, C;
    int xz, xs, xu, xsu;        // primal x: node comps [3][N], slacks [NS][N], controls [C], control slack [C]
    int vloc, vend, vdyn, vctl; // v-state: [NLOC][N], [2], [NDYN][N], [NCTL][C]
    int bz, bu, gs, gsu;        // rhs / solution of the linear solve, slack rhs
    int sl, slu;                // slack elimination data [NS][4][N], control [2][C]
    int fac, facu;              // factor: [N][18], [C][4]
    int px, py;                 // output scratch [N] each (aliases bz)
    int total;
    __host__ __device__ Layout(int N_, int C_) : N(N_), C(C_) {
        int o = 0;
        xz = o; o += 3 * N;
        xs = o; o += T::NS * N;
        xu = o; o += C;
        xsu = o; o += T::HAS_SU ? C : 0;
        vloc = o; o += T::NLOC * N;
        vend = o; o += 2;
        vdyn = o; o += T::NDYN * N;
        vctl = o; o += T::NCTL * C;
        bz = o; o += 3 * N;
        bu = o; o += C;
        gs = o; o += T::NS * N;
        gsu = o; o += T::HAS_SU ? C : 0;
        sl = o; o += T::NS * 4 * N;
        slu = o; o += T::HAS_SU ? 2 * C : 0;
        fac = o; o += 18 * N;
        facu = o; o += 4 * C;
        px = bz; py = bz + N;
        total = o;
    }
};

// -------------------------------------------------------------------------------------------------------
// row functors
// -------------------------------------------------------------------------------------------------------
struct HessFn {  // H(5x5 sym, upper, row-major packed 15) += rho * a a'
    double H[15];
    double rho, rho_eq;
    __device__ HessFn(double r, double re) : rho(r), rho_eq(re) {
#pragma unroll
        for (int i = 0; i < 15; ++i) H[i] = 0;
    }
    static __device__ __forceinline__ constexpr int idx(int a, int b) { return a * 5 - a * (a - 1) / 2 + (b - a); }
    template <int MASK> __device__ __forceinline__ void row(int, double c0, double c1, double c2, double c3, double c4, double l, double u) {
        const double r = rho_of(l, u, rho, rho_eq);
        const double c[5] = {c0, c1, c2, c3, c4};
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = a; b < 5; ++b)
                if ((MASK >> a & 1) && (MASK >> b & 1)) H[idx(a, b)] += r * c[a] * c[b];
    }
};

// rhs pass: t = 2*zc - v (zc = clip(v), or 0 on the very first iteration); g += rho * t * a
struct RhsFn {
    double g[5];
    const double *v;  // LDS, element r at v[r*stride]
    int stride;
    double rho, rho_eq;
    bool first;
    template <int MASK> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, double l, double u) {
        const double rr = rho_of(l, u, rho, rho_eq);
        const double vv = v[r * stride];
        const double zc = first ? 0.0 : clipd(vv, l, u);
        const double t = rr * (2.0 * zc - vv);
        const double c[5] = {c0, c1, c2, c3, c4};
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) g[a] += t * c[a];
    }
};

// update pass: ztilde = a . xtilde ; v += alpha (ztilde - zc)
struct UpdFn {
    double xt[5];
    double *v;
    int stride;
    double alpha;
    bool first;
    template <int MASK> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, double l, double u) {
        const double c[5] = {c0, c1, c2, c3, c4};
        double zt = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) zt += c[a] * xt[a];
        const double vv = v[r * stride];
        const double zc = first ? 0.0 : clipd(vv, l, u);
        v[r * stride] = vv + alpha * (zt - zc);
    }
};

// residual pass: Ax, z = clip(v), y = rho (v - z)
struct ResFn {
    double x[5];
    double aty[5];
    const double *v;
    int stride;
    double rho, rho_eq;
    double rp, nAx, nz;
    template <int MASK> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, double l, double u) {
        const double rr = rho_of(l, u, rho, rho_eq);
        const double c[5] = {c0, c1, c2, c3, c4};
        double ax = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) ax += c[a] * x[a];
        const double vv = v[r * stride];
        const double z = clipd(vv, l, u);
        const double y = rr * (vv - z);
        rp = fmax(rp, fabs(ax - z));
        nAx = fmax(nAx, fabs(ax));
        nz = fmax(nz, fabs(z));
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) aty[a] += y * c[a];
    }
};

// rho change: keep z and y, re-express v = z + y/rho_new = zc + (rho_old/rho_new)(v - zc)
struct RescaleFn {
    double *v;
    int stride;
    double ratio;
    template <int MASK> __device__ __forceinline__ void row(int r, double, double, double, double, double, double l, double u) {
        const double vv = v[r * stride];
        const double zc = clipd(vv, l, u);
        const bool loose = (l < -kInfThresh && u > kInfThresh);
        v[r * stride] = loose ? vv : zc + ratio * (vv - zc);
    }
};

// diagnostic assembly: l,u into the reference row order
template <int F> struct AsmFn {
    double *l, *u;
    int j, N, C, kind;  // kind 0 local, 1 end, 2 ctl (j = control id)
    template <int MASK> __device__ __forceinline__ void row(int r, double, double, double, double, double, double lo, double hi) {
        const int rr = kind == 0 ? ref_row_local<F>(r, j, N, C) : (kind == 1 ? ref_row_end<F>(r, N, C) : ref_row_ctl<F>(r, j, N, C));
        l[rr] = lo;
        u[rr] = hi;
    }
};

// Control-local rows.  KP: the vacuous row -inf <= u_c <= inf (solver_kp_as_input.cpp:105-107,160-163).
// KPC: kpl / kpu / Skp >= 0 (solver_kp_as_input_constrained.cpp:119-125,178-187).  coefficient slots: (unused,unused,u,su,unused)
template <int F, class Fn> __device__ __forceinline__ void ctl_rows(double maxkp, Fn &fn) {
    if constexpr (F == F_KP) {
        fn.template row<M_C>(0, 0., 0., 1., 0., 0., -kInf, kInf);
    } else if constexpr (F == F_KPC) {
        fn.template row<M_C | M_S1>(0, 0., 0., 1., 1., 0., -maxkp, kInf);
        fn.template row<M_C | M_S1>(1, 0., 0., 1., -1., 0., -kInf, maxkp);
        fn.template row<M_S1>(2, 0., 0., 0., 1., 0., 0.0, kInf);
    }
}

// -------------------------------------------------------------------------------------------------------
// per-path context
// -------------------------------------------------------------------------------------------------------
template <int F> struct Ctx {
    using T = FormTraits<F>;
    const DevParams &P;
    const DevBatch &in;
    Layout<F> L;
    double *S;  // LDS base
    int N, C, keep, lane;
    size_t po;  // path offset (b*N)
    int b;
    double elo, ehi;

    __device__ Ctx(const DevParams &P_, const DevBatch &in_, double *S_, int b_)
        : P(P_), in(in_), L(in_.N, in_.C), S(S_), N(in_.N), C(in_.C), keep(in_.keep), lane(threadIdx.x), po((size_t)b_ * in_.N), b(b_) {
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

// -------------------------------------------------------------------------------------------------------
// factor(): step A (lane-parallel) builds per-stage blocks, step B (sequential) the block LDL'.
// fac[j] = { Ginv(6) | Ln(9) | lu(3) } ; before step B the same slots hold { G0(6) | E(9) | wsrc(3) },
// and facu[c] = { pu0, wtgt-scale... } see below.
// -------------------------------------------------------------------------------------------------------
template <int F> __device__ void factor(const Ctx<F> &cx, double rho) {
    using T = FormTraits<F>;
    const auto &L = cx.L;
    double *S = cx.S;
    const int N = cx.N, C = cx.C, keep = cx.keep;
    const DevParams &P = cx.P;
    const double rho_eq = kRhoEqOverIneq * rho;
    const double sg = P.sigma;

    // ---- step A: node-local reduced Hessian G0_j, coupling E_j, control couplings --------------------
    for (int j = cx.lane; j < N; j += 64) {
        const StageIn si = cx.stage_in(j);
        HessFn h(rho, rho_eq);
        local_rows<F>(si, P, h);
        if (T::NEND && si.last) end_rows<F>(si, P, h);
        // diagonal: sigma + P
        h.H[HessFn::idx(0, 0)] += sg + p_diag_node<F>(0, j, N, P);
        h.H[HessFn::idx(1, 1)] += sg + p_diag_node<F>(1, j, N, P);
        h.H[HessFn::idx(2, 2)] += sg + p_diag_node<F>(2, j, N, P);
        h.H[HessFn::idx(3, 3)] += sg + P.w_s1;
        h.H[HessFn::idx(4, 4)] += sg + P.w_s2;
        // eliminate local slacks (they touch no other stage): Hzz -= h h' / Hss
        double G[6] = {h.H[HessFn::idx(0, 0)], h.H[HessFn::idx(0, 1)], h.H[HessFn::idx(0, 2)],
                       h.H[HessFn::idx(1, 1)], h.H[HessFn::idx(1, 2)], h.H[HessFn::idx(2, 2)]};
#pragma unroll
        for (int s = 0; s < T::NS; ++s) {
            const double hss = h.H[HessFn::idx(3 + s, 3 + s)];
            const double hi = 1.0 / hss;
            const double hv[3] = {h.H[HessFn::idx(0, 3 + s)], h.H[HessFn::idx(1, 3 + s)], h.H[HessFn::idx(2, 3 + s)]};
            S[L.sl + (s * 4 + 0) * N + j] = hi;
            S[L.sl + (s * 4 + 1) * N + j] = hv[0];
            S[L.sl + (s * 4 + 2) * N + j] = hv[1];
            S[L.sl + (s * 4 + 3) * N + j] = hv[2];
            G[0] -= hv[0] * hv[0] * hi; G[1] -= hv[0] * hv[1] * hi; G[2] -= hv[0] * hv[2] * hi;
            G[3] -= hv[1] * hv[1] * hi; G[4] -= hv[1] * hv[2] * hi; G[5] -= hv[2] * hv[2] * hi;
        }
        // incoming dynamics rows (or the initial-state rows): -1 on component tau
#pragma unroll
        for (int r = 0; r < T::NDYN; ++r) {
            const int t = dyn_tau<F>(r);
            G[t == 0 ? 0 : (t == 1 ? 3 : 5)] += rho_eq;
        }
        double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double ws[3] = {0, 0, 0};
        if (j < N - 1) {  // outgoing transition j: source role
            const Dyn<F> d = cx.dyn(j);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) {
                const int t = dyn_tau<F>(r);
                G[0] += rho_eq * d.f[r][0] * d.f[r][0]; G[1] += rho_eq * d.f[r][0] * d.f[r][1]; G[2] += rho_eq * d.f[r][0] * d.f[r][2];
                G[3] += rho_eq * d.f[r][1] * d.f[r][1]; G[4] += rho_eq * d.f[r][1] * d.f[r][2]; G[5] += rho_eq * d.f[r][2] * d.f[r][2];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    E[t * 3 + c] += -rho_eq * d.f[r][c];
                    ws[c] += rho_eq * d.beta[r] * d.f[r][c];
                }
            }
            if constexpr (F == F_K) {
                if (j + 1 <= N - 2) E[8] += -P.w_cr;  // P's tri-diagonal R couples delta_j and delta_{j+1}
            }
        }
        double *f = S + L.fac + 18 * j;
#pragma unroll
        for (int q = 0; q < 6; ++q) f[q] = G[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) f[6 + q] = E[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) f[15 + q] = ws[q];
    }
    if constexpr (T::HAS_U) {
        for (int c = cx.lane; c < C; c += 64) {
            // control-local 2x2 on (u, su): sigma + P + control rows, then eliminate su
            HessFn h(rho, rho_eq);
            const double mkp = (F == F_KPC) ? cx.in.max_kp[cx.po + c] : 0.0;
            ctl_rows<F>(mkp, h);
            double huu = h.H[HessFn::idx(2, 2)] + sg + P.w_u;
            if constexpr (T::HAS_SU) {
                const double hss = h.H[HessFn::idx(3, 3)] + sg + P.w_su;
                const double hus = h.H[HessFn::idx(2, 3)];
                const double hi = 1.0 / hss;
                S[L.slu + c] = hi;
                S[L.slu + C + c] = hus;
                huu -= hus * hus * hi;
            }
            // + rho_eq * beta^2 of every transition that uses this control
            const int i0 = c * keep, i1 = min(i0 + keep, N - 1);
            for (int i = i0; i < i1; ++i) {
                const Dyn<F> d = cx.dyn(i);
#pragma unroll
                for (int r = 0; r < T::NDYN; ++r) huu += rho_eq * d.beta[r] * d.beta[r];
            }
            S[L.facu + 4 * c] = huu;
        }
    }
    __syncthreads();

    // ---- step B: sequential block LDL' (every lane runs the same scalar recursion on broadcast LDS reads;
    //      lane 0 stores) --------------------------------------------------------------------------------
    double G[6], w[3] = {0, 0, 0}, pu = 0;
    {
        const double *f0 = S + L.fac;
#pragma unroll
        for (int q = 0; q < 6; ++q) G[q] = f0[q];
        if constexpr (T::HAS_U) {
            w[0] = f0[15]; w[1] = f0[16]; w[2] = f0[17];
            pu = S[L.facu];
        }
    }
    int c = 0;
    for (int j = 0; j < N; ++j) {
        double *f = S + L.fac + 18 * j;
        double Gi[6];
        inv3_sym(G, Gi);
        if (j == N - 1) {
            if (cx.lane == 0) {
#pragma unroll
                for (int q = 0; q < 6; ++q) f[q] = Gi[q];
            }
            break;
        }
        double E[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) E[q] = f[6 + q];
        const double *fn = f + 18;
        double Gn[6], wsn[3];
#pragma unroll
        for (int q = 0; q < 6; ++q) Gn[q] = fn[q];
        wsn[0] = fn[15]; wsn[1] = fn[16]; wsn[2] = fn[17];
        // Ln = E * Ginv
        const double Gf[9] = {Gi[0], Gi[1], Gi[2], Gi[1], Gi[3], Gi[4], Gi[2], Gi[4], Gi[5]};
        double Ln[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) Ln[r * 3 + q] = E[r * 3 + 0] * Gf[0 * 3 + q] + E[r * 3 + 1] * Gf[1 * 3 + q] + E[r * 3 + 2] * Gf[2 * 3 + q];
        // Gnext = G0_{j+1} - Ln E'
        Gn[0] -= Ln[0] * E[0] + Ln[1] * E[1] + Ln[2] * E[2];
        Gn[1] -= Ln[0] * E[3] + Ln[1] * E[4] + Ln[2] * E[5];
        Gn[2] -= Ln[0] * E[6] + Ln[1] * E[7] + Ln[2] * E[8];
        Gn[3] -= Ln[3] * E[3] + Ln[4] * E[4] + Ln[5] * E[5];
        Gn[4] -= Ln[3] * E[6] + Ln[4] * E[7] + Ln[5] * E[8];
        Gn[5] -= Ln[6] * E[6] + Ln[7] * E[7] + Ln[8] * E[8];
        double lu[3] = {0, 0, 0};
        if constexpr (T::HAS_U) {
            // lu = Ginv w ; wn = wtgt_{j+1} - E lu ; pu -= lu.w
            lu[0] = Gf[0] * w[0] + Gf[1] * w[1] + Gf[2] * w[2];
            lu[1] = Gf[3] * w[0] + Gf[4] * w[1] + Gf[5] * w[2];
            lu[2] = Gf[6] * w[0] + Gf[7] * w[1] + Gf[8] * w[2];
            const Dyn<F> d = cx.dyn(j);
            double wn[3] = {0, 0, 0};
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) wn[dyn_tau<F>(r)] += -rho_eq * d.beta[r];
#pragma unroll
            for (int r = 0; r < 3; ++r) wn[r] -= E[r * 3 + 0] * lu[0] + E[r * 3 + 1] * lu[1] + E[r * 3 + 2] * lu[2];
            pu -= lu[0] * w[0] + lu[1] * w[1] + lu[2] * w[2];
            if (last_of_group(j, N, keep)) {
                const double pi = 1.0 / pu;
                const double lx[3] = {wn[0] * pi, wn[1] * pi, wn[2] * pi};
                Gn[0] -= lx[0] * wn[0]; Gn[1] -= lx[0] * wn[1]; Gn[2] -= lx[0] * wn[2];
                Gn[3] -= lx[1] * wn[1]; Gn[4] -= lx[1] * wn[2]; Gn[5] -= lx[2] * wn[2];
                if (cx.lane == 0) {
                    double *fu = S + L.facu + 4 * c;
                    fu[0] = pi; fu[1] = lx[0]; fu[2] = lx[1]; fu[3] = lx[2];
                }
                ++c;
                if (j + 1 <= N - 2) {
                    w[0] = wsn[0]; w[1] = wsn[1]; w[2] = wsn[2];
                    pu = S[L.facu + 4 * c];
                }
            } else {
                w[0] = wn[0] + wsn[0]; w[1] = wn[1] + wsn[1]; w[2] = wn[2] + wsn[2];
            }
        }
        __syncthreads();  // all lanes have read stage j / j+1 inputs before lane 0 overwrites stage j
        if (cx.lane == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) f[q] = Gi[q];
#pragma unroll
            for (int q = 0; q < 9; ++q) f[6 + q] = Ln[q];
            f[15] = lu[0]; f[16] = lu[1]; f[17] = lu[2];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) G[q] = Gn[q];
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------------
// chain solve: L D L' x = b, in place in bz/bu.  v1: every lane runs the same scalar recursion.
// -------------------------------------------------------------------------------------------------------
template <int F> __device__ void chain_solve(const Ctx<F> &cx) {
    using T = FormTraits<F>;
    const auto &L = cx.L;
    double *S = cx.S;
    const int N = cx.N, keep = cx.keep;
    double *bz = S + L.bz, *bu = S + L.bu;
    // forward
    double y0 = bz[0], y1 = bz[N], y2 = bz[2 * N];
    double acc_u = T::HAS_U ? bu[0] : 0.0;
    int c = 0;
    for (int j = 0; j < N - 1; ++j) {
        const double *f = S + L.fac + 18 * j;
        double n0 = bz[j + 1] - (f[6] * y0 + f[7] * y1 + f[8] * y2);
        double n1 = bz[N + j + 1] - (f[9] * y0 + f[10] * y1 + f[11] * y2);
        double n2 = bz[2 * N + j + 1] - (f[12] * y0 + f[13] * y1 + f[14] * y2);
        if constexpr (T::HAS_U) {
            acc_u -= f[15] * y0 + f[16] * y1 + f[17] * y2;
            if (last_of_group(j, N, keep)) {
                const double *fu = S + L.facu + 4 * c;
                n0 -= fu[1] * acc_u; n1 -= fu[2] * acc_u; n2 -= fu[3] * acc_u;
                if (cx.lane == 0) bu[c] = acc_u;
                ++c;
                acc_u = (j + 1 <= N - 2) ? bu[c] : 0.0;
            }
        }
        if (cx.lane == 0) { bz[j + 1] = n0; bz[N + j + 1] = n1; bz[2 * N + j + 1] = n2; }
        y0 = n0; y1 = n1; y2 = n2;
    }
    __syncthreads();
    // diagonal (lane-parallel)
    for (int j = cx.lane; j < N; j += 64) {
        const double *f = S + L.fac + 18 * j;
        const double a0 = bz[j], a1 = bz[N + j], a2 = bz[2 * N + j];
        bz[j] = f[0] * a0 + f[1] * a1 + f[2] * a2;
        bz[N + j] = f[1] * a0 + f[3] * a1 + f[4] * a2;
        bz[2 * N + j] = f[2] * a0 + f[4] * a1 + f[5] * a2;
    }
    if constexpr (T::HAS_U)
        for (int cc = cx.lane; cc < cx.C; cc += 64) bu[cc] *= S[L.facu + 4 * cc];
    __syncthreads();
    // backward
    double x0 = bz[N - 1], x1 = bz[2 * N - 1], x2 = bz[3 * N - 1];
    double xu = 0;
    c = cx.C;
    for (int j = N - 2; j >= 0; --j) {
        const double *f = S + L.fac + 18 * j;
        if constexpr (T::HAS_U) {
            if (last_of_group(j, N, keep)) {
                --c;
                const double *fu = S + L.facu + 4 * c;
                xu = bu[c] - (fu[1] * x0 + fu[2] * x1 + fu[3] * x2);
                if (cx.lane == 0) bu[c] = xu;
            }
        }
        double n0 = bz[j] - (f[6] * x0 + f[9] * x1 + f[12] * x2) - f[15] * xu;
        double n1 = bz[N + j] - (f[7] * x0 + f[10] * x1 + f[13] * x2) - f[16] * xu;
        double n2 = bz[2 * N + j] - (f[8] * x0 + f[11] * x1 + f[14] * x2) - f[17] * xu;
        if (cx.lane == 0) { bz[j] = n0; bz[N + j] = n1; bz[2 * N + j] = n2; }
        x0 = n0; x1 = n1; x2 = n2;
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------------
// rhs pass:  b = sigma x + A' diag(rho) (2 clip(v) - v), local slacks folded in
// -------------------------------------------------------------------------------------------------------
template <int F> __device__ void rhs_pass(const Ctx<F> &cx, double rho, bool first) {
    using T = FormTraits<F>;
    const auto &L = cx.L;
    double *S = cx.S;
    const int N = cx.N, C = cx.C, keep = cx.keep;
    const DevParams &P = cx.P;
    const double rho_eq = kRhoEqOverIneq * rho, sg = P.sigma;
    for (int j = cx.lane; j < N; j += 64) {
        const StageIn si = cx.stage_in(j);
        RhsFn fn;
        fn.rho = rho; fn.rho_eq = rho_eq; fn.first = first;
        fn.g[0] = sg * S[L.xz + j]; fn.g[1] = sg * S[L.xz + N + j]; fn.g[2] = sg * S[L.xz + 2 * N + j];
        fn.g[3] = sg * S[L.xs + j];
        fn.g[4] = T::NS > 1 ? sg * S[L.xs + N + j] : 0.0;
        fn.v = S + L.vloc + j; fn.stride = N;
        local_rows<F>(si, P, fn);
        if (T::NEND && si.last) { fn.v = S + L.vend; fn.stride = 1; end_rows<F>(si, P, fn); }
        // incoming dynamics rows (or initial-state rows): equality rows, coefficient -1 on component tau
        double bin[3];
        if (j == 0) cx.init_bounds(bin);
        else { const Dyn<F> d = cx.dyn(j - 1);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) bin[r] = d.b[r]; }
#pragma unroll
        for (int r = 0; r < T::NDYN; ++r) {
            const double vv = S[L.vdyn + r * N + j];
            const double t = first ? 0.0 : rho_eq * (2.0 * bin[r] - vv);
            fn.g[dyn_tau<F>(r)] -= t;
        }
        // outgoing transition rows (owned by stage j+1): coefficients f on Z_j
        if (j < N - 1) {
            const Dyn<F> d = cx.dyn(j);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) {
                const double vv = S[L.vdyn + r * N + j + 1];
                const double t = first ? 0.0 : rho_eq * (2.0 * d.b[r] - vv);
                fn.g[0] += t * d.f[r][0]; fn.g[1] += t * d.f[r][1]; fn.g[2] += t * d.f[r][2];
            }
        }
        // fold slacks:  gz -= h * gs / Hss ; keep gs for the recovery after the solve
#pragma unroll
        for (int s = 0; s < T::NS; ++s) {
            const double hi = S[L.sl + (s * 4 + 0) * N + j];
            const double q = fn.g[3 + s] * hi;
            fn.g[0] -= S[L.sl + (s * 4 + 1) * N + j] * q;
            fn.g[1] -= S[L.sl + (s * 4 + 2) * N + j] * q;
            fn.g[2] -= S[L.sl + (s * 4 + 3) * N + j] * q;
            S[L.gs + s * N + j] = fn.g[3 + s];
        }
        S[L.bz + j] = fn.g[0]; S[L.bz + N + j] = fn.g[1]; S[L.bz + 2 * N + j] = fn.g[2];
    }
    if constexpr (T::HAS_U) {
        for (int c = cx.lane; c < C; c += 64) {
            RhsFn fn;
            fn.rho = rho; fn.rho_eq = rho_eq; fn.first = first;
            fn.g[0] = fn.g[1] = fn.g[4] = 0;
            fn.g[2] = sg * S[L.xu + c];
            fn.g[3] = T::HAS_SU ? sg * S[L.xsu + c] : 0.0;
            fn.v = S + L.vctl + c; fn.stride = C;
            const double mkp = (F == F_KPC) ? cx.in.max_kp[cx.po + c] : 0.0;
            ctl_rows<F>(mkp, fn);
            const int i0 = c * keep, i1 = min(i0 + keep, N - 1);
            for (int i = i0; i < i1; ++i) {
                const Dyn<F> d = cx.dyn(i);
#pragma unroll
                for (int r = 0; r < T::NDYN; ++r) {
                    const double vv = S[L.vdyn + r * N + i + 1];
                    const double t = first ? 0.0 : rho_eq * (2.0 * d.b[r] - vv);
                    fn.g[2] += t * d.beta[r];
                }
            }
            if constexpr (T::HAS_SU) {
                const double q = fn.g[3] * S[L.slu + c];
                fn.g[2] -= S[L.slu + C + c] * q;
                S[L.gsu + c] = fn.g[3];
            }
            S[L.bu + c] = fn.g[2];
        }
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------------
// update pass:  recover slacks, ztilde = A xtilde, v += alpha (ztilde - zc), x += alpha (xtilde - x)
// -------------------------------------------------------------------------------------------------------
template <int F> __device__ void update_pass(const Ctx<F> &cx, bool first) {
    using T = FormTraits<F>;
    const auto &L = cx.L;
    double *S = cx.S;
    const int N = cx.N, C = cx.C, keep = cx.keep;
    const DevParams &P = cx.P;
    const double al = P.alpha;
    for (int j = cx.lane; j < N; j += 64) {
        const StageIn si = cx.stage_in(j);
        UpdFn fn;
        fn.alpha = al; fn.first = first;
        fn.xt[0] = S[L.bz + j]; fn.xt[1] = S[L.bz + N + j]; fn.xt[2] = S[L.bz + 2 * N + j];
        fn.xt[3] = fn.xt[4] = 0;
#pragma unroll
        for (int s = 0; s < T::NS; ++s) {
            const double hi = S[L.sl + (s * 4 + 0) * N + j];
            const double hz = S[L.sl + (s * 4 + 1) * N + j] * fn.xt[0] + S[L.sl + (s * 4 + 2) * N + j] * fn.xt[1] + S[L.sl + (s * 4 + 3) * N + j] * fn.xt[2];
            fn.xt[3 + s] = (S[L.gs + s * N + j] - hz) * hi;
        }
        fn.v = S + L.vloc + j; fn.stride = N;
        local_rows<F>(si, P, fn);
        if (T::NEND && si.last) { fn.v = S + L.vend; fn.stride = 1; end_rows<F>(si, P, fn); }
        // incoming dynamics rows
        if (j == 0) {
            double bin[3];
            cx.init_bounds(bin);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) {
                const double zt = -fn.xt[dyn_tau<F>(r)];
                const double vv = S[L.vdyn + r * N];
                S[L.vdyn + r * N] = vv + al * (zt - (first ? 0.0 : bin[r]));
            }
        } else {
            const Dyn<F> d = cx.dyn(j - 1);
            const double p0 = S[L.bz + j - 1], p1 = S[L.bz + N + j - 1], p2 = S[L.bz + 2 * N + j - 1];
            const double ut = T::HAS_U ? S[L.bu + ctl_index(j - 1, keep)] : 0.0;
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) {
                const double zt = d.f[r][0] * p0 + d.f[r][1] * p1 + d.f[r][2] * p2 + d.beta[r] * ut - fn.xt[dyn_tau<F>(r)];
                const double vv = S[L.vdyn + r * N + j];
                S[L.vdyn + r * N + j] = vv + al * (zt - (first ? 0.0 : d.b[r]));
            }
        }
        // x += alpha (xtilde - x)
#pragma unroll
        for (int q = 0; q < 3; ++q) { const double xo = S[L.xz + q * N + j]; S[L.xz + q * N + j] = xo + al * (fn.xt[q] - xo); }
#pragma unroll
        for (int s = 0; s < T::NS; ++s) { const double xo = S[L.xs + s * N + j]; S[L.xs + s * N + j] = xo + al * (fn.xt[3 + s] - xo); }
    }
    if constexpr (T::HAS_U) {
        for (int c = cx.lane; c < C; c += 64) {
            UpdFn fn;
            fn.alpha = al; fn.first = first;
            fn.xt[0] = fn.xt[1] = fn.xt[4] = 0;
            fn.xt[2] = S[L.bu + c];
            fn.xt[3] = 0;
            if constexpr (T::HAS_SU) fn.xt[3] = (S[L.gsu + c] - S[L.slu + C + c] * fn.xt[2]) * S[L.slu + c];
            fn.v = S + L.vctl + c; fn.stride = C;
            const double mkp = (F == F_KPC) ? cx.in.max_kp[cx.po + c] : 0.0;
            ctl_rows<F>(mkp, fn);
            { const double xo = S[L.xu + c]; S[L.xu + c] = xo + al * (fn.xt[2] - xo); }
            if constexpr (T::HAS_SU) { const double xo = S[L.xsu + c]; S[L.xsu + c] = xo + al * (fn.xt[3] - xo); }
        }
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------------
// residual pass (every check_every iterations): unscaled OSQP residuals and their normalisers
// -------------------------------------------------------------------------------------------------------
struct Resid { double rp, rd, nAx, nz, nPx, nAty; };

template <int F> __device__ Resid residual_pass(const Ctx<F> &cx, double rho) {
    using T = FormTraits<F>;
    const auto &L = cx.L;
    const double *S = cx.S;
    const int N = cx.N, C = cx.C, keep = cx.keep;
    const DevParams &P = cx.P;
    const double rho_eq = kRhoEqOverIneq * rho;
    Resid R = {0, 0, 0, 0, 0, 0};
    for (int j = cx.lane; j < N; j += 64) {
        const StageIn si = cx.stage_in(j);
        ResFn fn;
        fn.rho = rho; fn.rho_eq = rho_eq; fn.rp = fn.nAx = fn.nz = 0;
        fn.x[0] = S[L.xz + j]; fn.x[1] = S[L.xz + N + j]; fn.x[2] = S[L.xz + 2 * N + j];
        fn.x[3] = S[L.xs + j]; fn.x[4] = T::NS > 1 ? S[L.xs + N + j] : 0.0;
#pragma unroll
        for (int a = 0; a < 5; ++a) fn.aty[a] = 0;
        fn.v = S + L.vloc + j; fn.stride = N;
        local_rows<F>(si, P, fn);
        if (T::NEND && si.last) { fn.v = S + L.vend; fn.stride = 1; end_rows<F>(si, P, fn); }
        // incoming dynamics rows
        double bin[3], ax[3];
        if (j == 0) {
            cx.init_bounds(bin);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) ax[r] = -fn.x[dyn_tau<F>(r)];
        } else {
            const Dyn<F> d = cx.dyn(j - 1);
            const double p0 = S[L.xz + j - 1], p1 = S[L.xz + N + j - 1], p2 = S[L.xz + 2 * N + j - 1];
            const double uu = T::HAS_U ? S[L.xu + ctl_index(j - 1, keep)] : 0.0;
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) {
                bin[r] = d.b[r];
                ax[r] = d.f[r][0] * p0 + d.f[r][1] * p1 + d.f[r][2] * p2 + d.beta[r] * uu - fn.x[dyn_tau<F>(r)];
            }
        }
#pragma unroll
        for (int r = 0; r < T::NDYN; ++r) {
            const double vv = S[L.vdyn + r * N + j];
            const double y = rho_eq * (vv - bin[r]);
            fn.rp = fmax(fn.rp, fabs(ax[r] - bin[r]));
            fn.nAx = fmax(fn.nAx, fabs(ax[r]));
            fn.nz = fmax(fn.nz, fabs(bin[r]));
            fn.aty[dyn_tau<F>(r)] -= y;
        }
        if (j < N - 1) {
            const Dyn<F> d = cx.dyn(j);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) {
                const double y = rho_eq * (S[L.vdyn + r * N + j + 1] - d.b[r]);
                fn.aty[0] += y * d.f[r][0]; fn.aty[1] += y * d.f[r][1]; fn.aty[2] += y * d.f[r][2];
            }
        }
        // P x
        double px[5];
        px[0] = p_diag_node<F>(0, j, N, P) * fn.x[0];
        px[1] = 0.0;
        px[2] = p_diag_node<F>(2, j, N, P) * fn.x[2];
        if constexpr (F == F_K) {
            if (j <= N - 2) {
                if (j >= 1) px[2] -= P.w_cr * S[L.xz + 2 * N + j - 1];
                if (j + 1 <= N - 2) px[2] -= P.w_cr * S[L.xz + 2 * N + j + 1];
            }
        }
        px[3] = P.w_s1 * fn.x[3];
        px[4] = T::NS > 1 ? P.w_s2 * fn.x[4] : 0.0;
#pragma unroll
        for (int a = 0; a < 3 + T::NS; ++a) {
            R.rd = fmax(R.rd, fabs(px[a] + fn.aty[a]));
            R.nPx = fmax(R.nPx, fabs(px[a]));
            R.nAty = fmax(R.nAty, fabs(fn.aty[a]));
        }
        R.rp = fmax(R.rp, fn.rp); R.nAx = fmax(R.nAx, fn.nAx); R.nz = fmax(R.nz, fn.nz);
    }
    if constexpr (T::HAS_U) {
        for (int c = cx.lane; c < C; c += 64) {
            ResFn fn;
            fn.rho = rho; fn.rho_eq = rho_eq; fn.rp = fn.nAx = fn.nz = 0;
            fn.x[0] = fn.x[1] = fn.x[4] = 0;
            fn.x[2] = S[L.xu + c];
            fn.x[3] = T::HAS_SU ? S[L.xsu + c] : 0.0;
#pragma unroll
            for (int a = 0; a < 5; ++a) fn.aty[a] = 0;
            fn.v = S + L.vctl + c; fn.stride = C;
            const double mkp = (F == F_KPC) ? cx.in.max_kp[cx.po + c] : 0.0;
            ctl_rows<F>(mkp, fn);
            const int i0 = c * keep, i1 = min(i0 + keep, N - 1);
            for (int i = i0; i < i1; ++i) {
                const Dyn<F> d = cx.dyn(i);
#pragma unroll
                for (int r = 0; r < T::NDYN; ++r) fn.aty[2] += rho_eq * (S[L.vdyn + r * N + i + 1] - d.b[r]) * d.beta[r];
            }
            const double pxu = P.w_u * fn.x[2], pxs = T::HAS_SU ? P.w_su * fn.x[3] : 0.0;
            R.rd = fmax(R.rd, fabs(pxu + fn.aty[2]));
            R.nPx = fmax(R.nPx, fabs(pxu));
            R.nAty = fmax(R.nAty, fabs(fn.aty[2]));
            if constexpr (T::HAS_SU) {
                R.rd = fmax(R.rd, fabs(pxs + fn.aty[3]));
                R.nPx = fmax(R.nPx, fabs(pxs));
                R.nAty = fmax(R.nAty, fabs(fn.aty[3]));
            }
            R.rp = fmax(R.rp, fn.rp); R.nAx = fmax(R.nAx, fn.nAx); R.nz = fmax(R.nz, fn.nz);
        }
    }
    // wavefront shuffle reductions (64 lanes)
    R.rp = wave_max(R.rp); R.rd = wave_max(R.rd); R.nAx = wave_max(R.nAx);
    R.nz = wave_max(R.nz); R.nPx = wave_max(R.nPx); R.nAty = wave_max(R.nAty);
    return R;
}

template <int F> __device__ void rescale_pass(const Ctx<F> &cx, double ratio) {
    using T = FormTraits<F>;
    const auto &L = cx.L;
    double *S = cx.S;
    const int N = cx.N, C = cx.C;
    for (int j = cx.lane; j < N; j += 64) {
        const StageIn si = cx.stage_in(j);
        RescaleFn fn;
        fn.ratio = ratio;
        fn.v = S + L.vloc + j; fn.stride = N;
        local_rows<F>(si, cx.P, fn);
        if (T::NEND && si.last) { fn.v = S + L.vend; fn.stride = 1; end_rows<F>(si, cx.P, fn); }
        double bin[3];
        if (j == 0) cx.init_bounds(bin);
        else { const Dyn<F> d = cx.dyn(j - 1);
#pragma unroll
            for (int r = 0; r < T::NDYN; ++r) bin[r] = d.b[r]; }
#pragma unroll
        for (int r = 0; r < T::NDYN; ++r) {
            const double vv = S[L.vdyn + r * N + j];
            S[L.vdyn + r * N + j] = bin[r] + ratio * (vv - bin[r]);
        }
    }
    if constexpr (T::HAS_U) {
        for (int c = cx.lane; c < C; c += 64) {
            RescaleFn fn;
            fn.ratio = ratio;
            fn.v = S + L.vctl + c; fn.stride = C;
            const double mkp = (F == F_KPC) ? cx.in.max_kp[cx.po + c] : 0.0;
            ctl_rows<F>(mkp, fn);
        }
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------------
// output pass: getOptimizedPath (solver_kp_as_input.cpp:26-43, solver_k_as_input.cpp:22-44)
// -------------------------------------------------------------------------------------------------------
template <int F> __device__ void output_pass(const Ctx<F> &cx) {
    using T = FormTraits<F>;
    const auto &L = cx.L;
    double *S = cx.S;
    const int N = cx.N, C = cx.C;
    const DevBatch &in = cx.in;
    double *out = in.out_states + cx.po * 5;
    for (int j = cx.lane; j < N; j += 64) {
        const double ey = S[L.xz + j], ephi = S[L.xz + N + j];
        double k = S[L.xz + 2 * N + j];
        if constexpr (F == F_K) {
            if (j == N - 1) k = S[L.xz + 2 * N + N - 2];  // last point re-uses the last control (:33-38)
        }
        const double ang = in.ref_z[cx.po + j];
        const double na = wrap_angle(ang + kPi2);
        const double tx = in.ref_x[cx.po + j] + __dmul_rn(ey, cos(na));
        const double ty = in.ref_y[cx.po + j] + __dmul_rn(ey, sin(na));
        S[L.px + j] = tx;
        S[L.py + j] = ty;
        out[5 * j + 0] = tx;
        out[5 * j + 1] = ty;
        out[5 * j + 2] = ang + ephi;
        out[5 * j + 3] = k;
    }
    __syncthreads();
    // running Euclidean arc length: sequential like the reference so the rounding is identical
    if (cx.lane == 0) {
        double s = 0;
        out[4] = 0;
        for (int j = 1; j < N; ++j) {
            const double dx = S[L.px + j] - S[L.px + j - 1], dy = S[L.py + j] - S[L.py + j - 1];
            s += sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
            out[5 * j + 4] = s;
        }
    }
    if (in.out_x) {  // raw QP solution in the reference variable order (dead variables are 0)
        double *ox = in.out_x + (size_t)cx.b * in.n;
        for (int q = cx.lane; q < in.n; q += 64) ox[q] = 0.0;
        __syncthreads();
        for (int j = cx.lane; j < N; j += 64) {
            if constexpr (F == F_K) {
                ox[2 * j] = S[L.xz + N + j];
                ox[2 * j + 1] = S[L.xz + j];
                if (j < N - 1) ox[2 * N + j] = S[L.xz + 2 * N + j];
                ox[3 * N - 1 + j] = S[L.xs + j];
            } else {
                ox[3 * j] = S[L.xz + j];
                ox[3 * j + 1] = S[L.xz + N + j];
                ox[3 * j + 2] = S[L.xz + 2 * N + j];
                ox[3 * N + C + j] = S[L.xs + j];
                if constexpr (F == F_KPC) ox[3 * N + C + N + j] = S[L.xs + N + j];
            }
        }
        if constexpr (T::HAS_U)
            for (int c = cx.lane; c < C; c += 64) {
                ox[3 * N + c] = S[L.xu + c];
                if constexpr (T::HAS_SU) ox[3 * N + C + 2 * N + c] = S[L.xsu + c];
            }
    }
}

// -------------------------------------------------------------------------------------------------------
// the fused kernel: one wavefront (block of 64) per path
// -------------------------------------------------------------------------------------------------------
template <int F> __global__ __launch_bounds__(64) void solve_kernel(DevBatch in, DevParams P) {
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    Ctx<F> cx(P, in, smem, b);
    // cold start x = 0, v = z + y/rho = 0  (fresh OSQP workspace per call, solver.cpp:46-73)
    for (int q = cx.lane; q < cx.L.bz; q += 64) smem[q] = 0.0;
    __syncthreads();
    double rho = fmin(fmax(P.rho0, kRhoMin), kRhoMax);
    factor<F>(cx, rho);
    int status = PO_STATUS_UNSOLVED, it = 0, nref = 0;
    Resid R = {0, 0, 0, 0, 0, 0};
    long long tc[5] = {0, 0, 0, 0, 0};
    const bool prof = in.dbg_cycles != nullptr;
    for (it = 1; it <= P.max_iter; ++it) {
        const bool first = (it == 1);
        long long t0 = prof ? __builtin_readcyclecounter() : 0;
        rhs_pass<F>(cx, rho, first);
        long long t1 = prof ? __builtin_readcyclecounter() : 0;
        chain_solve<F>(cx);
        long long t2 = prof ? __builtin_readcyclecounter() : 0;
        update_pass<F>(cx, first);
        long long t3 = prof ? __builtin_readcyclecounter() : 0;
        tc[0] += t1 - t0; tc[1] += t2 - t1; tc[2] += t3 - t2;
        const bool can_check = P.check_every > 0 && (it % P.check_every == 0);
        const bool can_adapt = P.adapt_every > 0 && (it % P.adapt_every == 0);
        if (can_check || can_adapt || it == P.max_iter) {
            R = residual_pass<F>(cx, rho);
            if (can_check || it == P.max_iter) {
                const double eps_p = P.eps_abs + P.eps_rel * fmax(R.nAx, R.nz);
                const double eps_d = P.eps_abs + P.eps_rel * fmax(R.nPx, R.nAty);
                if (R.rp < eps_p && R.rd < eps_d) { status = PO_STATUS_SOLVED; break; }
            }
            if (can_adapt) {  // OSQP compute_rho_estimate / adapt_rho
                const double pr = R.rp / (fmax(R.nAx, R.nz) + 1e-10);
                const double dr = R.rd / (fmax(R.nPx, R.nAty) + 1e-10);
                double rn = rho * sqrt(pr / (dr + 1e-10));
                rn = fmin(fmax(rn, kRhoMin), kRhoMax);
                if (rn > rho * P.adapt_tol || rn < rho / P.adapt_tol) {
                    rescale_pass<F>(cx, rho / rn);
                    rho = rn;
                    factor<F>(cx, rho);
                    ++nref;
                }
            }
        }
    }
    if (it > P.max_iter) { it = P.max_iter; if (status == PO_STATUS_UNSOLVED) status = PO_STATUS_MAX_ITER; }
    output_pass<F>(cx);
    if (prof && cx.lane == 0) {
        long long *d = in.dbg_cycles + (size_t)b * 4;
        d[0] = tc[0]; d[1] = tc[1]; d[2] = tc[2]; d[3] = it;
    }
    if (cx.lane == 0) {
        po_info o;
        o.status = status; o.iters = it; o.n_refactor = nref; o.reserved = 0;
        o.r_prim = R.rp; o.r_dual = R.rd; o.rho = rho; o.obj = 0.0;
        in.out_info[b] = o;
    }
}

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

#include "po_fast.inc"

template <int F> size_t lds_bytes(int N, int C) { return sizeof(double) * (size_t)Layout<F>(N, C).total; }

}  // namespace po

// ---- launch wrappers used by the C ABI (po_capi.cpp) ----
namespace po {
template <class K> hipError_t launch1(K kern, const DevBatch *in, const DevParams *P, int nt, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(in->B), dim3(nt), lds, st, *in, *P);
    return hipGetLastError();
}
template <int F> hipError_t launch_form(const DevBatch *in, const DevParams *P, hipStream_t st, size_t *lds_out, int variant) {
    const int N = in->N, C = in->C;
    // variant: 0 = auto, 1 = force generic (v1), 2 = force fast
    const bool fast_ok = (N <= 512) && (C <= (N <= 256 ? 64 : 128));
    if (variant != 1 && fast_ok) {
        const size_t lds = lds_bytes_fast<F>(N, C);
        if (lds_out) *lds_out = lds;
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        if (N <= 128) return launch1(&solve_kernel_fast<F, 2, 64>, in, P, 64, lds, st);
        if (N <= 256) return launch1(&solve_kernel_fast<F, 4, 64>, in, P, 64, lds, st);
        return launch1(&solve_kernel_fast<F, 4, 128>, in, P, 128, lds, st);
    }
    if (variant == 2) return hipErrorInvalidValue;
    const size_t lds = lds_bytes<F>(N, C);
    if (lds_out) *lds_out = lds;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    return launch1(&solve_kernel<F>, in, P, 64, lds, st);
}
}  // namespace po

extern "C" hipError_t po_launch_solve(int form, const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out, int variant) {
    using namespace po;
    if (form == F_KP) return launch_form<F_KP>(in, P, st, lds_out, variant);
    if (form == F_KPC) return launch_form<F_KPC>(in, P, st, lds_out, variant);
    return launch_form<F_K>(in, P, st, lds_out, variant);
}

extern "C" hipError_t po_launch_assemble(int form, const po::DevBatch *in, const po::DevParams *P, double *l, double *u, double *dyn, hipStream_t st) {
    using namespace po;
    if (form == F_KP) hipLaunchKernelGGL(assemble_kernel<F_KP>, dim3(in->B), dim3(64), 0, st, *in, *P, l, u, dyn);
    else if (form == F_KPC) hipLaunchKernelGGL(assemble_kernel<F_KPC>, dim3(in->B), dim3(64), 0, st, *in, *P, l, u, dyn);
    else hipLaunchKernelGGL(assemble_kernel<F_K>, dim3(in->B), dim3(64), 0, st, *in, *P, l, u, dyn);
    return hipGetLastError();
}

extern "C" size_t po_lds_bytes(int form, int N, int C) {
    using namespace po;
    const bool fast_ok = (N <= 512) && (C <= (N <= 256 ? 64 : 128));
    if (fast_ok) return form == F_KP ? lds_bytes_fast<F_KP>(N, C) : (form == F_KPC ? lds_bytes_fast<F_KPC>(N, C) : lds_bytes_fast<F_K>(N, C));
    return form == F_KP ? lds_bytes<F_KP>(N, C) : (form == F_KPC ? lds_bytes<F_KPC>(N, C) : lds_bytes<F_K>(N, C));
}
