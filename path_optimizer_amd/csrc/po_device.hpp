// po_device.hpp — device-side model of the three QP formulations for gfx950.
//
// The reference assembles P and A into dense scratch matrices
//   KP  /root/reference/src/solver/solver_kp_as_input.cpp:45-203
//   KPC /root/reference/src/solver/solver_kp_as_input_constrained.cpp:45-221
//   K   /root/reference/src/solver/solver_k_as_input.cpp:46-207
// Here nothing is ever assembled: every constraint row is (re)generated on the fly from the few
// data-dependent numbers it contains, in a STAGE-INTERLEAVED layout:
//   node  Z_j = (e_y, e_phi, c)_j   c = curvature k (KP/KPC) or steering delta_j (K)
//   slacks s1_j, s2_j local to a stage, held controls u_c (+ its slack su_c in KPC) local to a group
// so that the reduced KKT matrix  M = P + sigma I + A' diag(rho) A  is block tri-diagonal in Z with
// one scalar u_c attached to `keep` consecutive nodes.  See DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/po_hip.h"

namespace po {

constexpr double kInf = 1e30;        // OsqpEigen::INFTY
constexpr double kInfThresh = 1e26;  // OSQP_INFTY * MIN_SCALING: "infinite" bound test
constexpr int kStatusDeferred = -100;  // po_info.status while a solve is in flight: left to the general launch of the round (-100) / handed back to round r (-100 - r)
constexpr double kRhoMin = 1e-6, kRhoMax = 1e6, kRhoEqOverIneq = 1e3, kRhoTol = 1e-4;
constexpr double kPi = 3.14159265358979323846;
constexpr double kPi2 = 1.57079632679489661923;

enum { F_KP = 0, F_KPC = 1, F_K = 2 };
enum { M_EY = 1, M_EPHI = 2, M_C = 4, M_S1 = 8, M_S2 = 16 };

struct DevParams {
    double d1, d2, d3, d4;
    double w_dev, w_c, w_cr, w_s1, w_s2, w_u, w_su;  // P diagonal per variable class (w_u, w_su already x keep)
    double margin, kmax, max_steer, wheel_base;
    double sigma, alpha, rho0, eps_abs, eps_rel, eps_pinf, adapt_tol;
    int max_iter, check_every, adapt_every, end_heading;
    double pol_delta;           // OSQP delta
    int polish, pol_refine;
    int refine, ref_rounds, ref_extra;  // po_params.refine (0 / 2), refine_rounds, refine_extra_rounds
    double ref_eps;
    double ref_nw_rho, ref_nw_rho_eq, ref_nw_rho_max, ref_nw_rho_eq_max, ref_ls_tol;  // po_params.refine = 2 (Newton refinement)
    int ref_ls_max, ref_nw_max, ref_nw_final, ref_nw_esc;
    int ref_nw_slice;  // sliced Newton launches: steps of the first launch (engine-internal scheduling, po_debug_set("newton_slice"); results do not depend on it)
};

struct DevBatch {
    int B, N, keep, C;
    const double *ref_x, *ref_y, *ref_z, *ref_k, *ref_s;
    const double *bounds, *x0, *goal_z, *max_k, *max_kp;
    double *out_states;
    po_info *out_info;
    double *out_x;
    const int *n_points;    // optional [B]: ragged batch (points of each path <= N); arrays keep stride N
    const int *order;       // optional [B]: workgroup -> path permutation supplied by the caller (po_batch_in.order), overrides perm_bits
    const double *scale;    // [B][64] per-path equilibration block (po_scale.hpp)
    long long *dbg_cycles;  // optional [B][4] per-phase shader-clock totals (dev tool), or nullptr
    int perm_bits;          // block -> path permutation: ceil(log2 B) bits of mixing (0 = blockIdx order), see solve_kernel_fast
    int only_deferred;      // set by the launcher for the second (general) launch of the two-level mapping: solve only the paths the first one deferred
    int n, m;
    int round;              // po_params.refine_rounds: which round this launch is (0: all paths; r > 0: the paths round r - 1 handed back; newton_fallback_kernel)
    double *pol_state;      // [B][pol_stride] per-lane ADMM state left by the solve kernels for newton_kernel / polish_kernel (or nullptr)
    int pol_stride;
    int *fb_list;           // po_params.refine = 2: work list of the paths newton_kernel hands back (count, then path ids; newton_fallback_kernel)
    // sliced Newton launches (newton_kernel): 0 one launch; 1 first launch (parks after DevParams::ref_nw_slice steps); 2 second launch (resumes the parked paths in list order)
    int nw_phase;
    double *nw_state;       // [B][nw_stride]: a parked path's lane state, factor classes and phase scalars (Fast::park_io)
    int nw_stride;
    int *nw_keys;           // [B]: -1 finished in the first launch, else the priority key (larger = longer expected)
    int *nw_list;           // count, then the parked path ids in launch order (nw_sort_kernel)
    // set by the engine on the warm-start launch of a split solve (po_params.refine = 2): newton_kernel follows on the same stream and writes the outputs of every path
    // the warm start reports SOLVED (from its own result, or — attempt not taken — from the state block), so the solve kernel skips its output pass for those
    int nw_follows;
};

template <int F> struct FormTraits;
template <> struct FormTraits<F_KP> {
    static constexpr int NLOC = 8, NEND = 2, NDYN = 3, NS = 1, HAS_U = 1, NCTL = 1, HAS_SU = 0;
};
template <> struct FormTraits<F_KPC> {
    static constexpr int NLOC = 9, NEND = 2, NDYN = 3, NS = 2, HAS_U = 1, NCTL = 3, HAS_SU = 1;
};
template <> struct FormTraits<F_K> {
    static constexpr int NLOC = 9, NEND = 0, NDYN = 2, NS = 1, HAS_U = 0, NCTL = 0, HAS_SU = 0;
};

__host__ __device__ __forceinline__ double clipd(double v, double l, double u) { return fmin(fmax(v, l), u); }
// One-sided and free rows: the generators pass these tags instead of the literal +-kInf, and the vacuous half of the clip is dropped at
// compile time (|v| never gets near 1e30): one v_max / v_min less per soft-corridor row and pass.  Everywhere else they read as +-kInf.
struct NegInfT { __host__ __device__ constexpr operator double() const { return -kInf; } };
struct PosInfT { __host__ __device__ constexpr operator double() const { return kInf; } };
__host__ __device__ __forceinline__ double clipd(double v, NegInfT, double u) { return fmin(v, u); }
__host__ __device__ __forceinline__ double clipd(double v, double l, PosInfT) { return fmax(v, l); }
__host__ __device__ __forceinline__ double clipd(double v, NegInfT, PosInfT) { return v; }

// OSQP set_rho_vec: loose rows (both bounds infinite) get RHO_MIN, equalities (u-l < 1e-4) 1e3*rho.
__device__ __forceinline__ double rho_of(double l, double u, double rho, double rho_eq) {
    if (l < -kInfThresh && u > kInfThresh) return kRhoMin;
    return (u - l < kRhoTol) ? rho_eq : rho;
}

// constraintAngle, /root/reference/include/path_optimizer/tools/tools.hpp:24-35
__device__ __forceinline__ double wrap_angle(double a) {
    for (int it = 0; it < 64; ++it) {
        if (a > kPi) a -= 2 * kPi;
        else if (a < -kPi) a += 2 * kPi;
        else break;
    }
    return a;
}

// Per-stage inputs a lane needs to regenerate the rows of stage j.
struct StageIn {
    double lb[4], ub[4];  // covering-circle clearances c0..c3
    double maxk;          // KPC
    double elo, ehi;      // end-heading window (only meaningful at j == N-1)
    int j, N, last;
};

// ---- stage-local rows.  fn.row<MASK>(r, c_ey, c_ephi, c_c, c_s1, c_s2, l, u) -------------------------
template <int F, class Fn> __device__ __forceinline__ void local_rows(const StageIn &s, const DevParams &P, Fn &fn) {
    if constexpr (F == F_KP) {
        // solver_kp_as_input.cpp:100-134 (rows) and :153-188 (bounds)
        fn.template row<M_C>(0, 0., 0., 1., 0., 0., -P.kmax, P.kmax);
        fn.template row<M_S1>(1, 0., 0., 0., 1., 0., 0.0, P.margin);
        fn.template row<M_EY | M_EPHI>(2, 1., P.d1, 0., 0., 0., s.lb[0], s.ub[0]);
        fn.template row<M_EY | M_EPHI>(3, 1., P.d3, 0., 0., 0., s.lb[2], s.ub[2]);
        fn.template row<M_EY | M_EPHI | M_S1>(4, 1., P.d4, 0., -1., 0., NegInfT{}, s.ub[3] - P.margin);
        fn.template row<M_EY | M_EPHI | M_S1>(5, 1., P.d4, 0., 1., 0., s.lb[3] + P.margin, PosInfT{});
        fn.template row<M_EY | M_EPHI | M_S1>(6, 1., P.d2, 0., -1., 0., NegInfT{}, s.ub[1] - P.margin);
        fn.template row<M_EY | M_EPHI | M_S1>(7, 1., P.d2, 0., 1., 0., s.lb[1] + P.margin, PosInfT{});
    } else if constexpr (F == F_KPC) {
        // solver_kp_as_input_constrained.cpp:110-143 (rows) and :165-205 (bounds)
        fn.template row<M_C | M_S2>(0, 0., 0., 1., 0., 1., -s.maxk, PosInfT{});
        fn.template row<M_C | M_S2>(1, 0., 0., 1., 0., -1., NegInfT{}, s.maxk);
        fn.template row<M_S1>(2, 0., 0., 0., 1., 0., 0.0, P.margin);
        fn.template row<M_S2>(3, 0., 0., 0., 0., 1., 0.0, fmax(P.kmax - s.maxk, 0.0));
        fn.template row<M_EY | M_EPHI>(4, 1., P.d1, 0., 0., 0., s.lb[0], s.ub[0]);
        fn.template row<M_EY | M_EPHI>(5, 1., P.d2, 0., 0., 0., s.lb[1], s.ub[1]);
        fn.template row<M_EY | M_EPHI>(6, 1., P.d4, 0., 0., 0., s.lb[3], s.ub[3]);
        fn.template row<M_EY | M_EPHI | M_S1>(7, 1., P.d3, 0., -1., 0., NegInfT{}, s.ub[2] - P.margin);
        fn.template row<M_EY | M_EPHI | M_S1>(8, 1., P.d3, 0., 1., 0., s.lb[2] + P.margin, PosInfT{});
    } else {
        // solver_k_as_input.cpp:123-147 (rows) and :167-206 (bounds); identity rows on every variable
        const bool win = s.last && (s.elo > -kInf);
        fn.template row<M_EPHI>(0, 0., 1., 0., 0., 0., win ? s.elo : -kInf, win ? s.ehi : kInf);
        fn.template row<M_EY>(1, 1., 0., 0., 0., 0., NegInfT{}, PosInfT{});
        if (!s.last) fn.template row<M_C>(2, 0., 0., 1., 0., 0., -P.max_steer, P.max_steer);
        fn.template row<M_S1>(3, 0., 0., 0., 1., 0., 0.0, P.margin);
        fn.template row<M_EY | M_EPHI>(4, 1., P.d1, 0., 0., 0., s.lb[0], s.ub[0]);
        fn.template row<M_EY | M_EPHI>(5, 1., P.d3, 0., 0., 0., s.lb[2], s.ub[2]);
        fn.template row<M_EY | M_EPHI>(6, 1., P.d4, 0., 0., 0., s.lb[3], s.ub[3]);
        fn.template row<M_EY | M_EPHI | M_S1>(7, 1., P.d2, 0., -1., 0., NegInfT{}, s.ub[1] - P.margin);
        fn.template row<M_EY | M_EPHI | M_S1>(8, 1., P.d2, 0., 1., 0., s.lb[1] + P.margin, PosInfT{});
    }
}

// End-state rows of KP (:135-137,191-202) / KPC (:146-147,209-220); K has none (its window is row 0 above).
template <int F, class Fn> __device__ __forceinline__ void end_rows(const StageIn &s, const DevParams &, Fn &fn) {
    if constexpr (F == F_KP) {
        fn.template row<M_EY>(0, 1., 0., 0., 0., 0., -1.0, 1.0);
        fn.template row<M_EPHI>(1, 0., 1., 0., 0., 0., s.elo, s.ehi);
    } else if constexpr (F == F_KPC) {
        fn.template row<M_EY>(0, 1., 0., 0., 0., 0., NegInfT{}, PosInfT{});
        fn.template row<M_EPHI>(1, 0., 1., 0., 0., 0., s.elo, s.ehi);
    }
}

// Reference row index of a stage-local row (for the diagnostic assembly output only).
template <int F> __device__ __forceinline__ int ref_row_local(int r, int j, int N, int C) {
    if constexpr (F == F_KP) {
        const int cb = 5 * N + C;
        switch (r) {
            case 0: return 3 * N + j;
            case 1: return 4 * N + C + j;
            case 2: return cb + 2 * j;
            case 3: return cb + 2 * j + 1;
            default: return cb + (r - 2) * N + j;  // r=4..7 -> cb+2N.. cb+5N
        }
    } else if constexpr (F == F_KPC) {
        const int sb = 5 * N + 2 * C, cb = 7 * N + 3 * C;
        switch (r) {
            case 0: return 3 * N + j;
            case 1: return 4 * N + j;
            case 2: return sb + j;
            case 3: return sb + N + j;
            case 4: case 5: case 6: return cb + 3 * j + (r - 4);
            case 7: return cb + 3 * N + j;
            default: return cb + 4 * N + j;
        }
    } else {
        switch (r) {
            case 0: return 2 * N + 2 * j;
            case 1: return 2 * N + 2 * j + 1;
            case 2: return 4 * N + j;
            case 3: return 5 * N - 1 + j;
            case 4: case 5: case 6: return 6 * N - 1 + 3 * j + (r - 4);
            case 7: return 9 * N - 1 + j;
            default: return 10 * N - 1 + j;
        }
    }
}
template <int F> __device__ __forceinline__ int ref_row_end(int r, int N, int C) {
    if constexpr (F == F_KP) return 11 * N + C + r;
    else if constexpr (F == F_KPC) return 12 * N + 3 * C + r;
    else return 0;
}
template <int F> __device__ __forceinline__ int ref_row_ctl(int r, int c, int N, int C) {
    if constexpr (F == F_KP) return 4 * N + c;
    else if constexpr (F == F_KPC) return r == 0 ? 5 * N + c : (r == 1 ? 5 * N + C + c : 5 * N + 2 * C + 2 * N + c);
    else return 0;
}

// ---- dynamics rows of transition i (from node i into node i+1) -----------------------------------------
// row r:  f[r] . Z_i + beta[r] * u_{c(i)} - Z_{i+1}[tau[r]] = b[r]     (equality: l = u = b)
template <int F> struct Dyn {
    double f[FormTraits<F>::NDYN][3];
    double beta[FormTraits<F>::NDYN];
    double b[FormTraits<F>::NDYN];
};
template <int F> __device__ __forceinline__ constexpr int dyn_tau(int r) {
    if constexpr (F == F_K) return r == 0 ? 1 : 0;  // K: row 0 is the e_phi equation, row 1 the e_y equation
    else return r;
}
template <int F> __device__ __forceinline__ Dyn<F> make_dyn(double k, double ds, const DevParams &P) {
    Dyn<F> d;
    if constexpr (F == F_K) {
        // setDynamicMatrix, solver_k_as_input.cpp:89-103 ; c_i :159-166.  pow(x,2) == x*x (gcc folds it).
        const double steer = atan(k * P.wheel_base);
        const double cs = cos(steer);
        const double c2 = __dmul_rn(cs, cs);
        d.f[0][0] = __dmul_rn(-ds, __dmul_rn(k, k));  // on e_y
        d.f[0][1] = 1.0;                             // on e_phi
        d.f[0][2] = ds / P.wheel_base / c2;          // on delta
        d.beta[0] = 0;
        d.b[0] = __dmul_rn(ds, steer) / P.wheel_base / c2;
        d.f[1][0] = 1.0;
        d.f[1][1] = ds;
        d.f[1][2] = 0;
        d.beta[1] = 0;
        d.b[1] = 0;
    } else {
        // A = a*ds + I, B = b*ds, bound = ds*k_ref   (solver_kp_as_input.cpp:78-98,148-151)
        d.f[0][0] = 1.0; d.f[0][1] = ds; d.f[0][2] = 0; d.beta[0] = 0; d.b[0] = 0;
        d.f[1][0] = __dmul_rn(-__dmul_rn(k, k), ds); d.f[1][1] = 1.0; d.f[1][2] = ds; d.beta[1] = 0;
        d.b[1] = __dmul_rn(ds, k);
        d.f[2][0] = 0; d.f[2][1] = 0; d.f[2][2] = 1.0; d.beta[2] = ds; d.b[2] = 0;
    }
    return d;
}

// K only: the two trigonometric coefficients of a transition, computed once per path at load time
//   tr[0] = ds / L / cos^2(atan(k L))   (coefficient on delta),   tr[1] = ds * atan(k L) / L / cos^2(...)   (rhs)
__device__ __forceinline__ void k_trig(double k, double ds, const DevParams &P, double tr[2]) {
    const double steer = atan(k * P.wheel_base);
    const double cs = cos(steer);
    const double c2 = __dmul_rn(cs, cs);
    tr[0] = ds / P.wheel_base / c2;
    tr[1] = __dmul_rn(ds, steer) / P.wheel_base / c2;
}
template <int F> __device__ __forceinline__ Dyn<F> make_dyn_tr(double k, double ds, const double tr[2], const DevParams &P) {
    if constexpr (F == F_K) {
        Dyn<F> d;
        d.f[0][0] = __dmul_rn(-ds, __dmul_rn(k, k)); d.f[0][1] = 1.0; d.f[0][2] = tr[0]; d.beta[0] = 0; d.b[0] = tr[1];
        d.f[1][0] = 1.0; d.f[1][1] = ds; d.f[1][2] = 0; d.beta[1] = 0; d.b[1] = 0;
        return d;
    } else {
        return make_dyn<F>(k, ds, P);
    }
}

// P diagonal of node component `comp` (0 e_y, 1 e_phi, 2 c) at stage j.
template <int F> __device__ __forceinline__ double p_diag_node(int comp, int j, int N, const DevParams &P) {
    if (comp == 0) return P.w_dev;
    if (comp == 1) return 0.0;
    if constexpr (F == F_K) {
        if (j >= N - 1) return 0.0;  // delta_{N-1} does not exist: padding variable
        return (j == 0 || j == N - 2) ? P.w_c + P.w_cr : P.w_cr * 2 + P.w_c;  // matrix_R, :62-76
    } else {
        return P.w_c;
    }
}

__device__ __forceinline__ int ctl_index(int i, int keep) { return keep == 4 ? (i >> 2) : (i / keep); }
__device__ __forceinline__ bool last_of_group(int i, int N, int keep) {
    return (i == N - 2) || (keep == 4 ? (((i + 1) & 3) == 0) : ((i + 1) % keep == 0));
}

// Bit-mixing bijection of a workgroup index onto [0, B): workgroup i runs on XCD i mod 8 (a static partition of every launch), so kernels
// whose work per instance varies take instance perm_index(blockIdx.x, ...) instead of blockIdx.x.  bits = ceil(log2 B), 0 = identity.
__device__ __forceinline__ int perm_index(unsigned i, int bits, int B) {
    if (bits <= 0) return (int)i;
    const unsigned mask = (1u << bits) - 1u, h = bits > 1 ? (unsigned)bits >> 1 : 1u;
    do {
        i ^= i >> h; i = (i * 0x9E3779B1u) & mask;
        i ^= i >> h; i = (i * 0x85EBCA6Bu) & mask;
        i ^= i >> h;
    } while (i >= (unsigned)B);
    return (int)i;
}

// Wave-wide reductions on DPP row shifts (VALU latency) instead of __shfl_xor (= ds_bpermute: an LDS round trip per level and value): Kogge-Stone inside the four
// rows of 16 lanes (row_shr 1, 2, 4, 8: lane 15 of a row ends up with the row's total), then the four row totals through v_readlane.  Every lane returns the same bits.
// Measured per kernel (A/B in one run, BASELINE config 3): newton_kernel 6.60 -> 5.92 ms per launch with the DPP form, the ADMM hot kernel 1.04 -> 1.10 ms (its
// residual passes run every 25 iterations only, and the allocation at the 512-register limit does not like the change): the DPP form in the Newton objects only.
#if defined(PO_REF) && PO_REF == 3 && !defined(PO_SHFL_REDUCE)
#define PO_DPP_REDUCE 1
#endif
#ifdef PO_DPP_REDUCE
template <int CTRL> __device__ __forceinline__ double dpp_shr_keep(double v) {  // lane t <- lane t - n of its row; lanes without a source keep their own value
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ double dpp_shr_zero(double v) {  // ... lanes without a source read 0
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_of(double v, int k) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_shr_zero<0x111>(v); v += dpp_shr_zero<0x112>(v); v += dpp_shr_zero<0x114>(v); v += dpp_shr_zero<0x118>(v);
    return (lane_of(v, 15) + lane_of(v, 31)) + (lane_of(v, 47) + lane_of(v, 63));
}
#else
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
#endif
// A WAVE-UNIFORM double into scalar registers.  Why it matters (round 6): a uniform value that VALU arithmetic produced lives in a VGPR, one copy per
// lane, and the register allocator may park such a copy (scratch / AccVGPR) INSIDE a region that only part of the wave executes — e.g. the `stage(q) < N` body of a slot the
// path's last lane does not own — and read it back under the full mask: the lanes that sat the region out get whatever their slot held.  Found on newton_kernel<KP,4,64,1,1>:
// rho of the warm start (loaded at kernel entry, used by the phase's last pass) came back 0 on the last lane of every path whose length is not a multiple of four, and the
// state handed to the fall-back rounds was Inf / NaN there.  SGPRs are spilled with v_writelane / v_readlane, which ignore the execution mask.
#ifndef PO_NO_UNI  // (-DPO_NO_UNI: the values stay where the compiler puts them — the build tools/poison_check.py's placement check is validated against)
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
__device__ __forceinline__ double uni(double v) { return v; }
__device__ __forceinline__ int uni(int v) { return v; }
#endif
// exponent all ones: Inf or NaN.  An integer test on purpose: the build uses -fno-honor-nans, under which `v != v` folds to false, and the
// residual norms are fmax-accumulated (fmax drops a NaN operand), so a non-finite iterate would otherwise read as "converged".
// The high word goes through an empty asm: otherwise the optimiser recognises the mask-and-compare as is.fpclass(v, inf | nan) and, the producing
// instruction carrying `nnan`, narrows it to an Inf test (measured: NaN iterates came back "solved").
__device__ __forceinline__ int nonfinite_bits(double v) {
    int hi = __double2hiint(v);
    asm volatile("" : "+v"(hi));
    return (hi & 0x7ff00000) == 0x7ff00000;
}

__device__ __forceinline__ int nan_bits(double v) {  // NaN only (an infinite clearance is a legitimate "no bound")
    int hi = __double2hiint(v), lo = __double2loint(v);
    asm volatile("" : "+v"(hi), "+v"(lo));
    return (hi & 0x7ff00000) == 0x7ff00000 && ((hi & 0x000fffff) | lo) != 0;
}

#ifdef PO_DPP_REDUCE
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_shr_keep<0x111>(v)); v = fmax(v, dpp_shr_keep<0x112>(v)); v = fmax(v, dpp_shr_keep<0x114>(v)); v = fmax(v, dpp_shr_keep<0x118>(v));
    return fmax(fmax(lane_of(v, 15), lane_of(v, 31)), fmax(lane_of(v, 47), lane_of(v, 63)));
}
#else
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
#endif

// -------------------------------------------------------------------------------------------------------
// row functors
// -------------------------------------------------------------------------------------------------------
// Every functor carries the per-path equilibration of the row kind it is visiting: W[r] = E_r^2 / c (step
// multiplier) and E[r] (bounds are classified on the SCALED problem, like OSQP's set_rho_vec).
__device__ __forceinline__ double row_rho(double l, double u, double e, double w, double rho, double rho_eq) {
    return rho_of(l * e, u * e, rho, rho_eq) * w;
}
// The class of a row (0 inequality, 1 equality, 2 free) depends only on its scaled bounds, i.e. it is fixed for the
// whole solve: it is computed once per path and packed 2 bits per row; the passes decode instead of re-deriving it.
__device__ __forceinline__ unsigned row_class(double l, double u, double e) {
    const double ls = l * e, us = u * e;
    if (ls < -kInfThresh && us > kInfThresh) return 2u;
    return (us - ls < kRhoTol) ? 1u : 0u;
}
__device__ __forceinline__ double class_rho(unsigned cls, int r, double w, double rho, double rho_eq) {
    const unsigned c = (cls >> (2 * r)) & 3u;
    return w * (c == 0u ? rho : (c == 1u ? rho_eq : kRhoMin));
}
struct ClassFn {  // packs the row classes of one stage / control
    unsigned cls;
    const double *E;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double, double, double, double, double, TL l, TU u) {
        cls |= row_class(l, u, E[r]) << (2 * r);
    }
};

struct HessFn {  // H(5x5 sym, upper, row-major packed 15) += rho * a a'
    double H[15];
    double rho, rho_eq;
    const double *W, *E;
    unsigned cls;
    __device__ HessFn(double r, double re, const double *W_, const double *E_, unsigned cls_) : rho(r), rho_eq(re), W(W_), E(E_), cls(cls_) {
#pragma unroll
        for (int i = 0; i < 15; ++i) H[i] = 0;
    }
    static __device__ __forceinline__ constexpr int idx(int a, int b) { return a * 5 - a * (a - 1) / 2 + (b - a); }
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int ri, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
        const double r = class_rho(cls, ri, W[ri], rho, rho_eq);
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
    const double *W, *E;
    unsigned cls;
    bool first;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
        const double rr = class_rho(cls, r, W[r], rho, rho_eq);
        const double vv = v[r * stride];
        const double zc = first ? 0.0 : clipd(vv, l, u);
        const double t = rr * (2.0 * zc - vv);
        const double c[5] = {c0, c1, c2, c3, c4};
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) g[a] += t * c[a];
    }
};

// Specialised rhs functor of the two-level hot loop.  UNI: every lane's rows have the same class pattern, so the per-row
// step rho~_r = W_r * {rho, rho_eq, rho_min}[class_r] is a wave-uniform number prepared once per pass (rr[]), instead of a
// 2-bit decode + two selects per row and stage.  FIRST: compile-time version of the cold-start special case (z^0 = 0).
template <bool UNI, bool FIRST, int NR, bool NWT = false> struct RhsFnX {  // NWT: Newton refinement (refine = 2), t = rho (clip(v) - v): minus the gradient
    double g[5];
    const double *v;
    int stride;
    double rho, rho_eq;
    const double *W;
    unsigned cls;
    double rr[NR];
    __device__ __forceinline__ void prepare(unsigned pat, int nrows) {
#pragma unroll
        for (int r = 0; r < NR; ++r) rr[r] = r < nrows ? class_rho(pat, r, W[r], rho, rho_eq) : 0.0;
    }
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
        const double rw = UNI ? rr[r] : class_rho(cls, r, W[r], rho, rho_eq);
        const double vv = v[r * stride];
        const double t = FIRST ? -(rw * vv) : (NWT ? rw * (clipd(vv, l, u) - vv) : rw * (2.0 * clipd(vv, l, u) - vv));
        const double c[5] = {c0, c1, c2, c3, c4};
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) g[a] += t * c[a];
    }
};
// ---- Newton refinement (po_params.refine = 2): v holds w = a.x + y / rho ----
// v <- a.x + ratio (v - clip(v)): the multiplier update (ratio = 1), a change of penalty (ratio = rho_old / rho_new), the entry from the ADMM state
struct NwReexFn {  // (v points into the lane state: copies in and out of the functor — ReclassFn's way — measured 9 % slower on the whole Newton launch)
    double x[5];
    double *v;
    double ratio, ratio_eq;  // inequality rows / rows that are equalities by TYPE (their penalty is fixed, po_params.refine_newton_rho_eq)
    unsigned cls_type;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
        const double c[5] = {c0, c1, c2, c3, c4};
        double ax = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) ax += c[a] * x[a];
        const double vv = v[r];
        v[r] = ax + (((cls_type >> (2 * r)) & 3u) == 1u ? ratio_eq : ratio) * (vv - clipd(vv, l, u));
    }
};
// the Newton direction d on one stage's rows, s = a.d.  MODE 0: the part of psi'(t) that is linear in t — rows that are equalities by TYPE:
// c0 += rho_eq (v - b) s, c1 += rho_eq s^2.  MODE 1: the step, v += t s.
template <int MODE> struct NwDirFn {
    double xt[5];
    double *v;
    double t, rho, rho_eq, c0, c1, f0;
    unsigned cls_type;
    const double *W;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double k0, double k1, double k2, double k3, double k4, TL l, TU u) {
        const double c[5] = {k0, k1, k2, k3, k4};
        double s = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) s += c[a] * xt[a];
        if constexpr (MODE == 0) {
            const unsigned k = (cls_type >> (2 * r)) & 3u;
            const double vv = v[r], dl = vv - clipd(vv, l, u);
            if (k == 1u) {
                const double rw = W[r] * rho_eq;
                c0 += rw * dl * s;
                c1 += rw * s * s;
            }
        } else v[r] += t * s;
    }
};
// one evaluation of the line search: the inequality rows (by TYPE) at x + t d:  f += rho (w - clip(w)) s,  fp += rho s^2 where w = v + t s is outside its bounds
struct NwLsFn {
    double xt[5];
    const double *v;
    double t, rho, f, fp;
    unsigned cls_type;
    const double *W;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double k0, double k1, double k2, double k3, double k4, TL l, TU u) {
        if (((cls_type >> (2 * r)) & 3u) != 0u) return;
        const double c[5] = {k0, k1, k2, k3, k4};
        double s = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) s += c[a] * xt[a];
        const double ww = v[r] + t * s, dl = ww - clipd(ww, l, u);
        const double rw = W[r] * rho;
        f += rw * dl * s;
        fp += dl != 0.0 ? rw * s * s : 0.0;
    }
};
// the first pass of the line search: psi'(0) AND psi'(1) with the slope at 1 in one walk over the rows (the first trial step is always t = 1; same numbers as two
// single evaluations — the row's s = a.d, its class test and its penalty are shared)
struct NwLs01Fn {
    double xt[5];
    const double *v;
    double rho, f0, f1, fp1;
    unsigned cls_type;
    const double *W;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double k0, double k1, double k2, double k3, double k4, TL l, TU u) {
        if (((cls_type >> (2 * r)) & 3u) != 0u) return;
        const double c[5] = {k0, k1, k2, k3, k4};
        double s = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) s += c[a] * xt[a];
        const double w0 = v[r], d0 = w0 - clipd(w0, l, u);
        const double w1 = w0 + 1.0 * s, d1 = w1 - clipd(w1, l, u);
        const double rw = W[r] * rho;
        f0 += rw * d0 * s;
        f1 += rw * d1 * s;
        fp1 += d1 != 0.0 ? rw * s * s : 0.0;
    }
};
template <bool FIRST> struct UpdFnX {
    double xt[5];
    double *v;
    int stride;
    double alpha;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
        const double c[5] = {c0, c1, c2, c3, c4};
        double zt = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) zt += c[a] * xt[a];
        const double vv = v[r * stride];
        v[r * stride] = vv + alpha * (FIRST ? zt : zt - clipd(vv, l, u));
    }
};

// update pass: ztilde = a . xtilde ; v += alpha (ztilde - zc)
struct UpdFn {
    double xt[5];
    double *v;
    int stride;
    double alpha;
    bool first;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
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

// update pass at a termination check: additionally accumulates OSQP's primal-infeasibility certificate on
// delta_y = y_new - y_old of THIS iteration (is_primal_infeasible): ||dy||, u.dy+ + l.dy-, and A' dy.
struct UpdCertFn {
    double xt[5];
    double *v;
    int stride;
    double alpha;
    bool first;
    double rho, rho_eq;
    const double *W, *E;
    unsigned cls;
    double ady[5];        // A' dy contribution to the local variables
    double ndy, sup;      // max |dy| (projected), sum of u*dy+ + l*dy-
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
        const double c[5] = {c0, c1, c2, c3, c4};
        double zt = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) zt += c[a] * xt[a];
        const double vv = v[r * stride];
        const double zc = first ? 0.0 : clipd(vv, l, u);
        const double vn = vv + alpha * (zt - zc);
        v[r * stride] = vn;
        const double rr = class_rho(cls, r, W[r], rho, rho_eq);
        double dy = rr * ((vn - clipd(vn, l, u)) - (vv - zc));
        const bool uinf = u * E[r] > kInfThresh, linf = l * E[r] < -kInfThresh;
        if (uinf) dy = linf ? 0.0 : fmin(dy, 0.0);
        else if (linf) dy = fmax(dy, 0.0);
        ndy = fmax(ndy, fabs(dy));
        sup += (uinf ? 0.0 : u * fmax(dy, 0.0)) + (linf ? 0.0 : l * fmin(dy, 0.0));
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) ady[a] += dy * c[a];
    }
};

// residual pass: Ax, z = clip(v), y = rho (v - z)
struct ResFn {
    double x[5];
    double aty[5];
    const double *v;
    int stride;
    double rho, rho_eq;
    const double *W, *E;
    unsigned cls;
    double rp, nAx, nz;     // unscaled (termination)
    double rps, nAxs, nzs;  // scaled by E (rho estimate)
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double c0, double c1, double c2, double c3, double c4, TL l, TU u) {
        const double rr = class_rho(cls, r, W[r], rho, rho_eq);
        const double c[5] = {c0, c1, c2, c3, c4};
        double ax = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
            if (MASK >> a & 1) ax += c[a] * x[a];
        const double vv = v[r * stride];
        const double z = clipd(vv, l, u);
        const double y = rr * (vv - z);
        const double e = E[r];
        rp = fmax(rp, fabs(ax - z)); nAx = fmax(nAx, fabs(ax)); nz = fmax(nz, fabs(z));
        rps = fmax(rps, e * fabs(ax - z)); nAxs = fmax(nAxs, e * fabs(ax)); nzs = fmax(nzs, e * fabs(z));
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
    unsigned cls;
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double, double, double, double, double, TL l, TU u) {
        const double vv = v[r * stride];
        const double zc = clipd(vv, l, u);
        const bool loose = ((cls >> (2 * r)) & 3u) == 2u;
        v[r * stride] = loose ? vv : zc + ratio * (vv - zc);
    }
};

// diagnostic assembly: l,u into the reference row order
template <int F> struct AsmFn {
    double *l, *u;
    int j, N, C, kind;  // kind 0 local, 1 end, 2 ctl (j = control id)
    template <int MASK, class TL, class TU> __device__ __forceinline__ void row(int r, double, double, double, double, double, TL lo, TU hi) {
        const int rr = kind == 0 ? ref_row_local<F>(r, j, N, C) : (kind == 1 ? ref_row_end<F>(r, N, C) : ref_row_ctl<F>(r, j, N, C));
        l[rr] = lo;
        u[rr] = hi;
    }
};

// Control-local rows.  KP: the vacuous row -inf <= u_c <= inf (solver_kp_as_input.cpp:105-107,160-163).
// KPC: kpl / kpu / Skp >= 0 (solver_kp_as_input_constrained.cpp:119-125,178-187).  coefficient slots: (unused,unused,u,su,unused)
template <int F, class Fn> __device__ __forceinline__ void ctl_rows(double maxkp, Fn &fn) {
    if constexpr (F == F_KP) {
        fn.template row<M_C>(0, 0., 0., 1., 0., 0., NegInfT{}, PosInfT{});
    } else if constexpr (F == F_KPC) {
        fn.template row<M_C | M_S1>(0, 0., 0., 1., 1., 0., -maxkp, PosInfT{});
        fn.template row<M_C | M_S1>(1, 0., 0., 1., -1., 0., NegInfT{}, maxkp);
        fn.template row<M_S1>(2, 0., 0., 0., 1., 0., 0.0, PosInfT{});
    }
}


}  // namespace po
