// po_smooth.hip — the reference-smoothing QPs on gfx950: one generic banded-QP ADMM engine, three assemblies.
//
// Reference (same OsqpEigen call pattern as the hot path: setHessianMatrix ... initSolver, solve):
//   TENSION2  TensionSmoother2::osqpSmooth       /root/reference/src/reference_path_smoother/tension_smoother_2.cpp:163-301
//   TENSION   TensionSmoother::osqpSmooth        /root/reference/src/reference_path_smoother/tension_smoother.cpp:186-314
//   POST      ReferencePathSmoother::postSmooth  /root/reference/src/reference_path_smoother/reference_path_smoother.cpp:534-650
//
// Mapping: one wavefront (one 64-thread block) per QP, whole solve on chip.
//   * variables are re-ordered point by point (x_i, y_i, theta_i, k_i | x_i, y_i, d_i | x_i, dx_i, ddx_i), so that the reduced
//     KKT matrix  M = P + sigma I + A' diag(rho) A  is BANDED with half-bandwidth W = 4 / 9 / 3;
//   * A lives in LDS as a row-wise ELL (<= 3 entries per row) plus a transposed index list per variable (<= 4 rows), built once;
//   * OSQP's Ruiz equilibration (10 passes over the KKT column norms + cost scaling) runs literally, lanes over columns/rows;
//   * banded LDL' of M: right-looking, the W(W+1)/2 trailing updates of a column on separate lanes;
//   * per iteration: rhs (lanes over variables), forward/backward substitution (one lane, register sliding window, factor column
//     read as 16-byte pairs), relaxation + projection + dual update in the one-number-per-row form v = z + y/rho (lanes over rows);
//   * termination / infeasibility certificates / adaptive rho exactly as OSQP (unscaled residuals every `check_every` iterations).
// The scaled P band, D and E (needed only at checks and refactorisations), the linear term q (touched by one fixed lane per entry) and
// the per-check dy scratch stay in an HBM block per QP: LDS decides how many QPs a CU holds, and the kernel is latency-bound.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>

#include "../../include/po_hip.h"
#define PO_MAP_DEVICE_CODE
#include "po_device.hpp"
#include "po_map.hpp"
#include "po_smooth.hpp"

namespace po {

template <int KIND> struct ST;
template <> struct ST<PO_SMOOTH_TENSION2> {
    static constexpr int W = 4, WP = 4, KA = 3, MINP = 3;
    static __host__ __device__ int n(int P) { return 4 * P - 1; }
    static __host__ __device__ int m(int P) { return 3 * (P - 1) + 2; }
};
template <> struct ST<PO_SMOOTH_TENSION> {
    static constexpr int W = 9, WP = 9, KA = 2, MINP = 3;
    static __host__ __device__ int n(int P) { return 3 * P; }
    static __host__ __device__ int m(int P) { return 3 * P; }
};
template <> struct ST<PO_SMOOTH_POST> {
    static constexpr int W = 3, WP = 0, KA = 3, MINP = 4;
    static __host__ __device__ int n(int P) { return 3 * P; }
    static __host__ __device__ int m(int P) { return 3 * P - 2; }
};
constexpr int kKT = 4;  // a variable appears in at most 4 rows in all three QPs

template <int W> __host__ __device__ constexpr int col_stride() { return (W + 2) & ~1; }  // dinv + W entries, padded to 16 bytes
template <int W> __host__ __device__ constexpr int col_pad() { return 4 * W; }            // zero columns past n: the sweeps need no guards
// Wide bands (W >= 8, TENSION): the factor is kept BLOCK-wise, W columns per block of blk_stride doubles — during the factorisation the W columns (LS doubles
// each) at the head of their block, afterwards the two dense W x W matrices of the blocked substitution (band_solve_blocks) in their place.
template <int W> __host__ __device__ constexpr int blk_stride() { return W >= 8 ? 2 * W * W : 0; }
template <int W> __host__ __device__ constexpr size_t factor_doubles(size_t np, bool blocked) {  // LDS doubles of the factor storage for np columns (natural layout: np columns of col_stride)
    const size_t nat = np * col_stride<W>();
    if constexpr (W >= 8) { const size_t blk = (np + W - 1) / W * blk_stride<W>(); return blocked && blk > nat ? blk : nat; }
    else return nat;
}
constexpr int kChunkPad = 72;  // doubles: one per chunk of the partitioned substitution (at most 64 chunks own rows, the zero padding columns reach a few further)
// Rows per lane of the partitioned substitution (narrow bands, W <= 4): a multiple of W; 0 when the QP is too small to give every chunk W rows inside the
// zero padding (the single-lane window then).  From 6 rows per lane on, the chunks are laid out bank-conflict-free (see Prob::cch); below that the natural
// layout's conflicts are mild and the extra LDS would cost a resident QP per CU.
template <int W> __host__ __device__ constexpr int scan_chunk(int n) {
    const int c = ((n + 63) / 64 + W - 1) / W * W;
    return c * ((n + c - 1) / c) <= n + col_pad<W>() ? c : 0;
}
__host__ __device__ constexpr bool scan_padded(int c) { return c >= 6; }


// LDS carve-up (doubles first, then 16-bit and 8-bit tables)
template <int KIND> struct Lds {
    using T = ST<KIND>;
    static constexpr int LS = col_stride<T::W>();
    static __host__ __device__ size_t bytes(int P, bool blocked) {  // blocked: wide bands in the block layout (band_solve_blocks), see po_smooth_blocked
        const size_t n = (size_t)T::n(P), m = (size_t)T::m(P), np = n + col_pad<T::W>();
        size_t d = factor_doubles<T::W>(np, blocked) + (size_t)T::KA * m + n + np + 3 * m;  // Lb, Av, x, wk, v, l, u  (q and the dy / scaling scratch tm live in the HBM block)
        if constexpr (T::W <= 4) {
            if (scan_padded(scan_chunk<T::W>((int)n))) d += kChunkPad + np + kChunkPad;  // chunk-padded factor, wkp (the partitioned substitution's right-hand side)
        }
        size_t b = d * 8 + ((size_t)T::KA * m + (size_t)kKT * n) * 2 + n * 4 + m + 8;
        return ((b + 15) & ~(size_t)15) + 64;  // + the kernel's static reduction scratch (8 doubles): what the capacity check must see
    }
};

__device__ __forceinline__ double lim_scaling(double v) {  // OSQP limit_scaling: MIN_SCALING 1e-4 -> 1, MAX_SCALING 1e4
    v = v < 1e-4 ? 1.0 : v;
    return v > 1e4 ? 1e4 : v;
}

template <int KIND> struct Prob {  // views into LDS + scratch for one QP
    using T = ST<KIND>;
    static constexpr int W = T::W, LS = col_stride<T::W>(), KA = T::KA;
    int n, m, np;
    double *Lb, *Av, *x, *q, *wk, *v, *l, *u, *tm;
    uint16_t *Ac, *Tl;
    int *Tc;
    uint8_t *cls;
    double *Pb, *Dv, *Ev;  // HBM scratch: Pb[d*n + j] = P[j][j+d] (scaled), D[n], E[m], then q[n] and tm[m]
    // Partitioned substitution (narrow bands): lane t owns the cch rows [t cch, (t + 1) cch).  In the natural layout the lanes of a wave would read the factor
    // cch LS doubles apart and the right-hand side cch doubles apart — even strides, 16- to 64-way LDS bank conflicts (measured: 60 % of the LDS-active
    // cycles).  So the factor carries one dead double per chunk (lane stride cch LS + 1: odd) and the substitution works on wkp, a copy of wk with one dead
    // double per chunk when cch is even.  cch == 0: natural layout (single-lane substitution).
    int cch, wpad;
    int bst;  // wide bands: block stride of the factor storage (blk_stride), 0 = natural layout (developer A/B: band_solve_lanes)
    float rcch;
    double *wkp;
    __device__ __forceinline__ int chunk_of(int j) const { return (int)(((float)j + 0.5f) * rcch); }  // j / cch, exact for j < 2^16
    __device__ __forceinline__ int lcol(int j) const {  // offset of factor column j in Lb
        if constexpr (W <= 4) return j * LS + (cch ? chunk_of(j) : 0);
        else return bst ? (j / W) * bst + (j % W) * LS : j * LS;
    }
    __device__ __forceinline__ int pidx(int j) const { return j + chunk_of(j) * wpad; }  // offset of row j in wkp (cch != 0)
    __device__ __forceinline__ void setA(int r, int s, int col, double val) { Av[s * m + r] = val; Ac[s * m + r] = (uint16_t)col; }
    __device__ __forceinline__ void setRow(int r, double lo, double hi) { l[r] = lo; u[r] = hi; }
    __device__ __forceinline__ void setP(int j, int d, double val) { Lb[j * LS + d] = val; }  // P band parked in the factor storage during setup
};

// ---- assemblies: rows in point order; values follow the reference expressions term by term ----------------------------------
template <int KIND> __device__ void assemble(const DevSmooth &a, Prob<KIND> &pb, int b, int P, int lane, int nts = 64) {  // lane: thread index, nts: threads of the block
    const size_t o = (size_t)b * a.P;
    if constexpr (KIND == PO_SMOOTH_TENSION2) {
        // tension_smoother_2.cpp:220-241 (Hessian), :288-299 (gradient), :243-286 (constraints)
        const double wd = a.w[0], wc = a.w[1], wr = a.w[2];
        for (int i = lane; i < P; i += nts) {
            const double xi = a.x[o + i], yi = a.y[o + i];
            pb.setP(4 * i, 0, wd * 2); pb.setP(4 * i + 1, 0, wd * 2); pb.setP(4 * i + 2, 0, 0.0);
            pb.q[4 * i] = -2 * wd * xi; pb.q[4 * i + 1] = -2 * wd * yi; pb.q[4 * i + 2] = 0;
            if (i < P - 1) {
                double dk = wc * 2;  // blocks i-1 and i of the curvature-rate term touch k_i (blocks run 0..P-3)
                if (i - 1 >= 0 && i - 1 <= P - 3) dk += 2 * wr;
                if (i <= P - 3) dk += 2 * wr;
                pb.setP(4 * i + 3, 0, dk);
                pb.setP(4 * i + 3, 4, i <= P - 3 ? -2 * wr : 0.0);
                pb.q[4 * i + 3] = 0;
                const double ds = a.s[o + i + 1] - a.s[o + i], ang = a.angle[o + i];
                const double sn = sin(ang), cs = cos(ang);
                pb.setA(3 * i, 0, 4 * i + 4, 1.0); pb.setA(3 * i, 1, 4 * i, -1.0); pb.setA(3 * i, 2, 4 * i + 2, ds * sn);
                pb.setRow(3 * i, ds * cs, ds * cs);
                pb.setA(3 * i + 1, 0, 4 * i + 5, 1.0); pb.setA(3 * i + 1, 1, 4 * i + 1, -1.0); pb.setA(3 * i + 1, 2, 4 * i + 2, -ds * cs);
                pb.setRow(3 * i + 1, ds * sn, ds * sn);
                pb.setA(3 * i + 2, 0, 4 * i + 6, 1.0); pb.setA(3 * i + 2, 1, 4 * i + 2, -1.0); pb.setA(3 * i + 2, 2, 4 * i + 3, -ds);
                const double bk = -ds * a.k[o + i];
                pb.setRow(3 * i + 2, bk, bk);
            }
        }
        if (lane == 0) {
            const int r = 3 * (P - 1);
            pb.setA(r, 0, 0, 1.0); pb.setA(r, 1, 0, 0.0); pb.setA(r, 2, 0, 0.0); pb.setRow(r, a.x[o], a.x[o]);
            pb.setA(r + 1, 0, 1, 1.0); pb.setA(r + 1, 1, 1, 0.0); pb.setA(r + 1, 2, 1, 0.0); pb.setRow(r + 1, a.y[o], a.y[o]);
        }
    } else if constexpr (KIND == PO_SMOOTH_TENSION) {
        // tension_smoother.cpp:238-261 (Hessian: [1 -2 1] and [-1 3 -3 1] stencils on x and on y), :263-314 (constraints)
        const double wc = a.w[3], wcr = a.w[4], wdev = a.w[5];
        const double v3[3] = {1, -2, 1}, v4[4] = {-1, 3, -3, 1};
        for (int i = lane; i < P; i += nts) {
            for (int e = 0; e <= 3; ++e) {  // H[i][i+e], accumulated block by block like the reference's loop
                double acc = 0;
                if (i + e < P)
                    for (int k = (i + e - 3 > 0 ? i + e - 3 : 0); k <= i && k <= P - 3; ++k) {
                        if (i + e - k <= 2) acc += v3[i - k] * v3[i + e - k] * wc;
                        if (k != P - 3) acc += v4[i - k] * v4[i + e - k] * wcr;
                    }
                pb.setP(3 * i, 3 * e, acc); pb.setP(3 * i + 1, 3 * e, acc);
            }
            pb.setP(3 * i + 2, 0, wdev);
            pb.q[3 * i] = 0; pb.q[3 * i + 1] = 0; pb.q[3 * i + 2] = 0;
            const double xi = a.x[o + i], yi = a.y[o + i], th = a.angle[o + i] + kPi2;
            pb.setA(3 * i, 0, 3 * i, 1.0); pb.setA(3 * i, 1, 3 * i + 2, -cos(th)); pb.setRow(3 * i, xi, xi);
            pb.setA(3 * i + 1, 0, 3 * i + 1, 1.0); pb.setA(3 * i + 1, 1, 3 * i + 2, -sin(th)); pb.setRow(3 * i + 1, yi, yi);
            pb.setA(3 * i + 2, 0, 3 * i + 2, 1.0); pb.setA(3 * i + 2, 1, 3 * i + 2, 0.0);
            double lo, hi;
            if (i == 0) { lo = 0; hi = 0; }
            else if (i == P - 1) { lo = -0.5; hi = 0.5; }
            else {
                double c = map_distance(a.map, xi, yi);  // Map::getObstacleDistance, tension_smoother.cpp:303
                c = c < 2.0 ? c : 2.0;
                lo = -c; hi = c;
            }
            pb.setRow(3 * i + 2, lo, hi);
        }
    } else {
        // reference_path_smoother.cpp:598-612 (Hessian), :614-650 (constraints)
        for (int i = lane; i < P; i += nts) {
            pb.setP(3 * i, 0, 1.0); pb.setP(3 * i + 1, 0, 100.0); pb.setP(3 * i + 2, 0, 1000.0);
            pb.q[3 * i] = 0; pb.q[3 * i + 1] = 0; pb.q[3 * i + 2] = 0;
            pb.setA(3 * i, 0, 3 * i, 1.0); pb.setA(3 * i, 1, 3 * i, 0.0); pb.setA(3 * i, 2, 3 * i, 0.0);
            if (i == 0) pb.setRow(0, a.l0[b], a.l0[b]);
            else pb.setRow(3 * i, a.lb[o + i], a.ub[o + i]);
            if (i < P - 1) {
                const double ds = a.s[o + i + 1] - a.s[o + i];
                pb.setA(3 * i + 1, 0, 3 * i + 3, 1.0); pb.setA(3 * i + 1, 1, 3 * i, -1.0); pb.setA(3 * i + 1, 2, 3 * i + 1, -ds);
                pb.setRow(3 * i + 1, 0.0, 0.0);
                pb.setA(3 * i + 2, 0, 3 * i + 4, 1.0); pb.setA(3 * i + 2, 1, 3 * i + 1, -1.0); pb.setA(3 * i + 2, 2, 3 * i + 2, -ds);
                pb.setRow(3 * i + 2, 0.0, 0.0);
            }
        }
    }
}

// device variable index -> reference variable index (raw output)
template <int KIND> __device__ __forceinline__ int ref_var(int j, int P) {
    if constexpr (KIND == PO_SMOOTH_TENSION2) return (j & 3) * P + (j >> 2);
    else return (j % 3) * P + j / 3;
}

template <int KIND> __device__ __forceinline__ double rho_row(const Prob<KIND> &pb, int r, double rho) {
    const unsigned c = pb.cls[r];
    return c == 0u ? rho : (c == 1u ? kRhoEqOverIneq * rho : kRhoMin);
}

// M = P + sigma I + A' diag(rho) A into the band storage, then right-looking banded LDL' (column j: dinv, L[j+1..j+W][j]).
// Barrier of the threads that work on one band: the block when it is one wave, wave 0 alone when the block has more (the pivot loop and the partitioned
// substitution run on wave 0 only; a wave's LDS operations complete in order, so draining its own counters is all the ordering they need).
template <int NWV> __device__ __forceinline__ void band_sync() {
    if constexpr (NWV == 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
}

template <int KIND, int NWV> __device__ void convert_blocks(Prob<KIND> &pb, int tid);
template <int KIND, int NWV = 1> __device__ void factorise(Prob<KIND> &pb, double rho, double sigma, int tid, int lane) {  // tid: thread of the block; lane: lane of the block's serial wave (outside 0 .. 63 on the others) — the pivot loop works on lanes 0 .. W (W + 1) / 2 - 1
    using T = ST<KIND>;
    constexpr int W = T::W, KA = T::KA, WP = T::WP;
    constexpr int nts = 64 * NWV;
    const int n = pb.n, m = pb.m;
    for (int j = tid; j < pb.np; j += nts) {
        double col[W + 1];
#pragma unroll
        for (int d = 0; d <= W; ++d) col[d] = 0;
        if (j < n) {
#pragma unroll
            for (int d = 0; d <= WP; ++d) col[d] = (j + d < n) ? pb.Pb[(size_t)d * n + j] : 0.0;
            col[0] += sigma;
            const int cnt = pb.Tc[j];
            for (int t = 0; t < cnt; ++t) {
                const int code = pb.Tl[j * kKT + t], r = code >> 2, s = code & 3;
                const double ra = rho_row(pb, r, rho) * pb.Av[s * m + r];
#pragma unroll
                for (int s2 = 0; s2 < KA; ++s2) {
                    const int c2 = pb.Ac[s2 * m + r];
                    const double a2 = pb.Av[s2 * m + r];
                    const int d = c2 - j;
                    if (d >= 0 && a2 != 0.0) {
#pragma unroll
                        for (int dd = 0; dd <= W; ++dd)
                            if (dd == d) col[dd] += ra * a2;
                    }
                }
            }
        }
        { const int bj = pb.lcol(j);
#pragma unroll
        for (int d = 0; d <= W; ++d) pb.Lb[bj + d] = col[d]; }
    }
    __syncthreads();
    if constexpr (W <= 4) {
        // Narrow bands: ONE lane carries the active window (the current column and the W columns it updates) in registers and takes one new column from
        // LDS per step, requested before the division it hides behind.  The multi-lane loop below pays two LDS round trips and two barriers per column
        // (650 cycles per column measured); here a column costs the reciprocal and two dependent operations.
        if (lane == 0) {
            double cur[W + 1], win[W][W + 1];
#pragma unroll
            for (int e = 0; e <= W; ++e) cur[e] = pb.Lb[pb.lcol(0) + e];
#pragma unroll
            for (int a = 0; a < W; ++a) {
                const int ba = pb.lcol(1 + a);
#pragma unroll
                for (int e = 0; e <= W; ++e) win[a][e] = pb.Lb[ba + e];
            }
            double out[W + 1];  // the finished column, stored one step later: behind the next step's loads, so that the wait for those does not include it
            int bo = pb.lcol(0);
#pragma unroll
            for (int e = 0; e <= W; ++e) out[e] = cur[e];  // (step 0 stores the column as it is; it is overwritten at step 1)
            for (int j = 0; j < n; ++j) {
                double nxt[W + 1];
                const int bn = pb.lcol(j + W + 1), bj = pb.lcol(j);  // (column j + W + 1 < np has not been touched yet: updates reach W columns ahead)
#pragma unroll
                for (int e = 0; e <= W; ++e) pb.Lb[bo + e] = out[e];
#pragma unroll
                for (int e = 0; e <= W; ++e) nxt[e] = pb.Lb[bn + e];
                // 1 / d by v_rcp_f64 and two Newton steps (within an ulp of the quotient): the IEEE division's scale / fix-up sequence is twice as long, and
                // this chain is what a column waits for
                double dinv = __builtin_amdgcn_rcp(cur[0]);
                dinv = fma(fma(-cur[0], dinv, 1.0), dinv, dinv);
                dinv = fma(fma(-cur[0], dinv, 1.0), dinv, dinv);
#pragma unroll
                for (int a = 1; a <= W; ++a) {
#pragma unroll
                    for (int b = a; b <= W; ++b) win[a - 1][b - a] -= cur[a] * dinv * cur[b];
                }
                out[0] = dinv;
#pragma unroll
                for (int e = 1; e <= W; ++e) out[e] = cur[e] * dinv;
                bo = bj;
#pragma unroll
                for (int e = 0; e <= W; ++e) {
                    cur[e] = win[0][e];
#pragma unroll
                    for (int a = 0; a + 1 < W; ++a) win[a][e] = win[a + 1][e];
                    win[W - 1][e] = nxt[e];
                }
            }
#pragma unroll
            for (int e = 0; e <= W; ++e) pb.Lb[bo + e] = out[e];
        }
    } else {
    // lane -> (a, b), 1 <= a <= b <= W
    constexpr int NPAIR = W * (W + 1) / 2;
    int pa = 0, pbb = 0;
    {
        int t = lane;
        for (int a = 1; a <= W; ++a) {
            const int len = W - a + 1;
            if (t >= 0 && t < len) { pa = a; pbb = a + t; }
            t -= len;
        }
    }
    if (NWV == 1 || (unsigned)lane < 64u)
    for (int j = 0; j < n; ++j) {
        const int bj = pb.lcol(j);
        const double d = pb.Lb[bj];
        const double dinv = 1.0 / d;
        double mine = 0;
        if (lane >= 1 && lane <= W) mine = pb.Lb[bj + lane];
        if (lane < NPAIR) {
            const double ca = pb.Lb[bj + pa], cb = pb.Lb[bj + pbb];
            pb.Lb[pb.lcol(j + pa) + (pbb - pa)] -= ca * dinv * cb;
        }
        band_sync<NWV>();
        if (lane == 0) pb.Lb[bj] = dinv;
        else if (lane <= W) pb.Lb[bj + lane] = mine * dinv;
        band_sync<NWV>();
    }
    }
    if constexpr (NWV > 1) __syncthreads();  // the trailing updates of the last columns land in the padding columns cleared below
    for (int j = n + tid; j < pb.np; j += nts) {  // padding columns: the trailing updates of the last columns spilled into them
#pragma unroll
        for (int d = 0; d <= W; ++d) pb.Lb[pb.lcol(j) + d] = 0;
    }
    __syncthreads();
    if constexpr (W >= 8) { if (pb.bst) convert_blocks<KIND, NWV>(pb, tid); }
}

// L D L' x = wk, in place, one lane.  Forward: column sweep with the W pending right-hand sides in registers; backward: row sweep.
// Both sweeps run block-wise over W columns with the NEXT block's factor columns (and pending right-hand sides) already in flight:
// a single wave per SIMD has nothing else to hide the LDS latency behind, so the loads are software-pipelined by hand (two register
// sets, ping-pong, no branch between issue and use so that the compiler's s_waitcnt can count instead of draining the queue).
template <int W, int LS> struct BandBlock {
    double c[W][W + 1];
    double b[W];
    __device__ __forceinline__ void load(const double *__restrict__ Lb, const double *__restrict__ wk, int jcol, int jb) {
#pragma unroll
        for (int uu = 0; uu < W; ++uu) {
            b[uu] = wk[jb + uu];
#pragma unroll
            for (int d = 0; d <= W; ++d) c[uu][d] = Lb[(jcol + uu) * LS + d];
        }
    }
};
template <int W, int LS> __device__ __forceinline__ void fwd_block(const BandBlock<W, LS> &k, double (&w)[W], double *__restrict__ wk, int j0) {
#pragma unroll
    for (int uu = 0; uu < W; ++uu) {
        const double yj = w[uu % W];
        wk[j0 + uu] = yj * k.c[uu][0];
#pragma unroll
        for (int d = 1; d < W; ++d) w[(uu + d) % W] -= k.c[uu][d] * yj;
        w[uu % W] = k.b[uu] - k.c[uu][W] * yj;
    }
}
template <int W, int LS> __device__ __forceinline__ void bwd_block(const BandBlock<W, LS> &k, double (&w)[W], double *__restrict__ wk, int j0) {
#pragma unroll
    for (int uu = W - 1; uu >= 0; --uu) {  // w[(j + d) % W] holds x_{j+d}, d = 1..W (slot j % W is the one being produced)
        double acc = k.b[uu];
#pragma unroll
        for (int d = W; d >= 2; --d) acc -= k.c[uu][d] * w[(uu + d) % W];
        acc -= k.c[uu][1] * w[(uu + 1) % W];
        wk[j0 + uu] = acc;
        w[uu % W] = acc;
    }
}
template <int KIND> __device__ __forceinline__ void band_solve(Prob<KIND> &pb) {
    using T = ST<KIND>;
    constexpr int W = T::W, LS = Prob<KIND>::LS;
    const int nsteps = (pb.n + 2 * W - 1) / (2 * W) * (2 * W);  // an even number of W-blocks; the zero padding (col_pad) covers the over-run
    const double *__restrict__ Lb = pb.Lb;
    double *__restrict__ wk = pb.wk;
    double w[W];
    BandBlock<W, LS> A, B;
#pragma unroll
    for (int d = 0; d < W; ++d) w[d] = wk[d];
    A.load(Lb, wk, 0, W);
    for (int j0 = 0; j0 < nsteps; j0 += 2 * W) {
        B.load(Lb, wk, j0 + W, j0 + 2 * W);
        fwd_block<W, LS>(A, w, wk, j0);
        A.load(Lb, wk, j0 + 2 * W, j0 + 3 * W);
        fwd_block<W, LS>(B, w, wk, j0 + W);
    }
#pragma unroll
    for (int d = 0; d < W; ++d) w[d] = 0;
    A.load(Lb, wk, nsteps - W, nsteps - W);
    for (int j0 = nsteps - W; j0 >= 0; j0 -= 2 * W) {
        B.load(Lb, wk, j0 - W, j0 - W);  // j0 - W >= 0: the block count is even
        bwd_block<W, LS>(A, w, wk, j0);
        const int jn = j0 - 2 * W > 0 ? j0 - 2 * W : 0;
        A.load(Lb, wk, jn, jn);
        bwd_block<W, LS>(B, w, wk, j0 - W);
    }
}

__device__ __forceinline__ double lane_bcast(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// L D L' x = wk, in place, on W cooperating lanes, column by column (wide bands; natural factor layout).  Since round 3 the DEVELOPER A/B path of TENSION
// (po_debug_set "smooth_seq" with one wave per QP; the product path is band_solve_blocks below) — the reference the blocked substitution is tested against.
// Lane l holds the pending sum of one row; at column j the owner lane's value is final, every lane takes it with a v_readlane broadcast and applies its entry
// of factor column j.  Round 2's form of this routine picked the entering row up in the same register — a v_cndmask pair between every two dependent FMAs —
// and guarded every load of the backward sweep: 81 k cycles per solve at P = 100 (136 per column and sweep).  This form: 32.6 k (54 per column):
//   * two registers per lane: wa = the row of the CURRENT block of W columns this lane owns (row j0 + l), wb = its row of the NEXT block (j0 + W + l).  At
//     in-block column uu a lane updates exactly one of them (wa if its row is still ahead of the column, wb otherwise): two FMAs on every lane, the lane mask
//     applied to the coefficient (see lane_fma2), no select between dependent FMAs; chain per column = v_readlane -> v_fma;
//   * one coefficient per lane and column at a per-lane offset (loop-invariant) from the block base, 1/d and the entering right-hand side once per block:
//     11 loads per block instead of 27, requested one whole block ahead into the other register set (ping-pong, no copies: the s_waitcnt counts);
//   * a lane's finished value stays in wa until the end of the block: one store per block;
//   * only the W working lanes run: an LDS access costs a quarter of the full-wave one.
// Arithmetic: the same FMAs in the same order as round 2's routine (bit-identical to it when it was replaced).
template <bool HASA> __device__ __forceinline__ void lane_fma2(double &wa, double &wb, double c, double y, bool ina) {
    // wa <- wa - c y on the lanes with `ina`, wb <- wb - c y on the others: the lane mask goes into the COEFFICIENT (off the chain; c - ca is exact), both FMAs run
    // on every lane.  (EXEC-masked FMAs — s_mov exec, v_fma, s_mov exec, v_fma, s_mov exec — measured 68 cycles per column: every instruction of the column
    // waits for the one before it.)  The empty assembler statement keeps the compiler from turning the zero coefficient back into a select of the result.
    if constexpr (HASA) {
        double ca = ina ? c : 0.0;
        asm volatile("" : "+v"(ca));
        const double cb = c - ca;
        wa = __builtin_fma(-ca, y, wa);
        wb = __builtin_fma(-cb, y, wb);
    } else {
        wb = __builtin_fma(-c, y, wb);
    }
}
template <int W> struct LaneSet { double c[W], dinv, bn; };
template <int W, int UU> __device__ __forceinline__ void lanes_fwd_step(double &wa, double &wb, const LaneSet<W> &s, int l) {
    lane_fma2<(UU < W - 1)>(wa, wb, s.c[UU], lane_bcast(wa, UU), l > UU);  // wa: lanes l > uu; wb: lanes l <= uu
}
template <int W, int UU> __device__ __forceinline__ void lanes_bwd_step(double &wa, double &wb, const LaneSet<W> &s, int l) {
    lane_fma2<(UU > 0)>(wa, wb, s.c[UU], lane_bcast(wa, UU), l < UU);  // wa: lanes l < uu; wb: lanes l >= uu
}
template <int W, int... U> __device__ __forceinline__ void lanes_fwd_block(double &wa, double &wb, const LaneSet<W> &s, int l, std::integer_sequence<int, U...>) {
    (lanes_fwd_step<W, U>(wa, wb, s, l), ...);
}
template <int W, int... U> __device__ __forceinline__ void lanes_bwd_block(double &wa, double &wb, const LaneSet<W> &s, int l, std::integer_sequence<int, U...>) {
    (lanes_bwd_step<W, W - 1 - U>(wa, wb, s, l), ...);
}
template <int KIND> __device__ __forceinline__ void band_solve_lanes(Prob<KIND> &pb, int lane) {
    using T = ST<KIND>;
    constexpr int W = T::W, LS = Prob<KIND>::LS;
    static_assert(W <= 16, "lane masks");
    if (lane >= W) return;
    const int l = lane;
    const int nblk = (pb.n + W - 1) / W;  // whole blocks of W columns; the zero padding (col_pad) covers the over-run of the look-ahead
    const double *Lb = pb.Lb;
    double *wk = pb.wk;
    constexpr auto seq = std::make_integer_sequence<int, W>{};
    LaneSet<W> A, B;
    // ---- forward: L y = b, stored as z = D^-1 y.  Column j0 + uu: row j0 + l takes L[j0 + l][j0 + uu] (l > uu), row j0 + W + l takes L[j0 + W + l][j0 + uu] (l <= uu) ----
    {
        int off[W];
#pragma unroll
        for (int uu = 0; uu < W; ++uu) off[uu] = uu * LS + (l - uu) + (l <= uu ? W : 0);
        auto load = [&](LaneSet<W> &s, int j0) {
#pragma unroll
            for (int uu = 0; uu < W; ++uu) s.c[uu] = Lb[j0 * LS + off[uu]];
            s.dinv = Lb[(j0 + l) * LS];
            s.bn = wk[j0 + 2 * W + l];
        };
        double wa = wk[l], wb = wk[W + l];
        load(A, 0);
        int k = 0;
        for (; k + 1 < nblk; k += 2) {
            load(B, (k + 1) * W);
            lanes_fwd_block<W>(wa, wb, A, l, seq);
            wk[k * W + l] = wa * A.dinv; wa = wb; wb = A.bn;
            load(A, (k + 2) * W);
            lanes_fwd_block<W>(wa, wb, B, l, seq);
            wk[(k + 1) * W + l] = wa * B.dinv; wa = wb; wb = B.bn;
        }
        if (k < nblk) {
            lanes_fwd_block<W>(wa, wb, A, l, seq);
            wk[k * W + l] = wa * A.dinv;
        }
    }
    // ---- backward: L' x = z, column-wise from the end.  Column j0 + uu: row j0 + l takes L[j0 + uu][j0 + l] (l < uu), row j0 - W + l takes L[j0 + uu][j0 - W + l] (l >= uu) ----
    {
        int off[W];
#pragma unroll
        for (int uu = 0; uu < W; ++uu) off[uu] = l < uu ? l * LS + (uu - l) : (l - W) * LS + (uu - l + W);
        auto load = [&]<bool CL>(LaneSet<W> &s, int j0) {  // CL: the blocks at the start of the band have no rows below them — clamped into the array, never used
#pragma unroll
            for (int uu = 0; uu < W; ++uu) { const int i = j0 * LS + off[uu]; s.c[uu] = Lb[CL ? (i > 0 ? i : 0) : i]; }
            const int r = j0 - 2 * W + l;
            s.bn = wk[CL ? (r > 0 ? r : 0) : r];
        };
        int j0 = (nblk - 1) * W;
        double wa = wk[j0 + l], wb = wk[j0 - W + l > 0 ? j0 - W + l : 0];
        load.template operator()<true>(A, j0);
        int k = nblk - 1;
        auto pair = [&]<bool CL>() {
            load.template operator()<CL>(B, (k - 1) * W);
            lanes_bwd_block<W>(wa, wb, A, l, seq);
            wk[k * W + l] = wa; wa = wb; wb = A.bn;
            load.template operator()<CL>(A, k >= 2 ? (k - 2) * W : 0);
            lanes_bwd_block<W>(wa, wb, B, l, seq);
            wk[(k - 1) * W + l] = wa; wa = wb; wb = B.bn;
        };
        for (; k >= 4; k -= 2) pair.template operator()<false>();  // (blocks k - 1 >= 3 and k - 2 >= 2: every index positive)
        for (; k >= 1; k -= 2) pair.template operator()<true>();
        if (k == 0) {
            lanes_bwd_block<W>(wa, wb, A, l, seq);
            wk[l] = wa;
        }
    }
}


// ---- wide bands, product path: BLOCKED substitution (round 3) ---------------------------------------------------------------------------------------------
// With the columns taken W at a time the factor is block bidiagonal: L = [T_0; C_0 T_1; C_1 T_2; ...], T_k unit lower triangular (the band inside block k),
// C_k upper triangular (rows of block k + 1 against the columns of block k), D = diag(D_k).  With r_k = T_k y_k,
//     forward   r_0 = b_0,  r_{k+1} = b_{k+1} - M_k r_k,      M_k = C_k T_k^-1                 (dense W x W)
//     backward  x_k = S_k r_k - M_k' x_{k+1},                  S_k = T_k^-T D_k^-1 T_k^-1       (applied in factored form Wm' (Wm r), Wm = D^-1/2 T^-1: round 4)
// M_k and S_k depend on the factor only: convert_blocks builds them after every (re-)factorisation, in place of the block's columns.  A block then costs one
// broadcast of W values (2 W v_readlane) and W FMAs on the chain instead of W dependent {v_readlane, v_readlane, v_fma} steps: a dependent v_fma_f64 ->
// v_readlane -> v_fma_f64 step is 30 cycles on gfx950 (tools/ubench/chain.hip), a broadcast whose source does not depend on the FMA before it 9 per value.
// g_k = S_k r_k rides on the forward broadcasts and is parked in wk; the backward sweep is the chain x_k = g_k - M_k' x_{k+1} alone.
// Numerically this applies explicit inverses of the W x W unit triangles T_k (entries of an SPD band factor: well conditioned) — iterates agree with
// band_solve_lanes to round-off, not bit for bit (tests/test_smooth.py bounds the difference and checks both against the oracle).
template <int KIND, int NWV> __device__ void convert_blocks(Prob<KIND> &pb, int tid) {
    using T = ST<KIND>;
    constexpr int W = T::W, LS = Prob<KIND>::LS, BS = blk_stride<W>(), nts = 64 * NWV, BPR = nts / W;  // BPR: blocks per round, one thread per (block, column)
    static_assert(W * LS + W * (W - 1) / 2 <= BS, "the inverse triangle is parked behind the block's columns");
    const int nblk = (pb.n + W - 1) / W;
    const int kl = tid / W, c = tid - kl * W;
    for (int k0 = 0; k0 < nblk; k0 += BPR) {
        const int k = k0 + kl;
        const bool act = kl < BPR && k < nblk;
        double *blk = pb.Lb + (size_t)(act ? k : 0) * BS;
        double t[W], mc[W], sc[W];
        // column c of T^-1 (zero above the diagonal): t_c = 1, t_i = - sum_{j < i} T[i][j] t_j
#pragma unroll
        for (int i = 0; i < W; ++i) {
            double acc = 0;
#pragma unroll
            for (int j = 0; j < i; ++j) acc -= blk[j * LS + (i - j)] * t[j];
            t[i] = i == c ? 1.0 : (i < c ? 0.0 : acc);
        }
        if (act) {
#pragma unroll
            for (int i = 1; i < W; ++i)
                if (i > c) blk[W * LS + i * (i - 1) / 2 + c] = t[i];
        }
        __syncthreads();
        // column c of M = C T^-1 (C[i][j] = L[W + i][j], j >= i) and of S = T^-T D^-1 T^-1
#pragma unroll
        for (int i = 0; i < W; ++i) {
            double acc = 0;
#pragma unroll
            for (int j = i; j < W; ++j) acc += blk[j * LS + (W + i - j)] * t[j];
            mc[i] = acc;
        }
        // column c of Wm = D^-1/2 T^-1 (lower triangular, stored DENSE with its zeros: the products below then need no masks).  S = Wm' Wm is applied in this
        // factored form, g = Wm' (Wm r) — see band_solve_blocks.
#pragma unroll
        for (int i = 0; i < W; ++i) sc[i] = i >= c ? sqrt(blk[i * LS]) * t[i] : 0.0;
        __syncthreads();
        if (act) {
#pragma unroll
            for (int i = 0; i < W; ++i) { blk[i * W + c] = -mc[i]; blk[W * W + i * W + c] = sc[i]; }  // (-M_k: the sweeps are pure multiply-adds)
        }
        __syncthreads();
    }
}

template <int W> struct BlkSet { double m[W], bn; };
// acc += (lane N of the caller's 16-lane row of v) * c in ONE instruction: gfx950 has DPP on 64-bit FMAC with row_newbcast (v_fmac_f64_dpp), so the broadcast of the block's
// vector costs no instruction of its own (the round-3 form: 2 v_readlane per value, 18 per block and sweep = a third of the substitution's VALU issue).  FIRST = the first
// use of v after the VALU instruction that wrote it (and after whatever touched EXEC): the DPP read needs wait states the compiler does not see inside inline assembly.
// The broadcast operand is declared read-write (it is not modified): the statements of one product are then ordered among themselves — the wait states sit in front of the FIRST
// one only — while loads and everything else may still move across them (with `asm volatile` instead: 13.8 instead of 13.4 ms per 4096 QPs).
template <int N, bool FIRST> __device__ __forceinline__ void fmac_bcast(double &acc, double &v, double c) {
    if constexpr (FIRST) asm("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(v) : "v"(c), "i"(N));
    else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(v) : "v"(c), "i"(N));
}
template <int W, int... U> __device__ __forceinline__ void blk_acc_bcast(double (&a)[3], const double (&c)[W], double &v, std::integer_sequence<int, U...>) {  // a[u % 3] += c[u] * v(lane u of the row)
    (fmac_bcast<U, U == 0>(a[U % 3], v, c[U]), ...);
}
template <int W, int... U> __device__ __forceinline__ double blk_dot_bcast(const double (&c)[W], double &v, double a0, std::integer_sequence<int, U...> sq) {  // a0 + sum_u c[u] * v(lane u of the row); three partial sums
    double a[3] = {a0, 0.0, 0.0};
    blk_acc_bcast<W>(a, c, v, sq);
    return a[0] + (a[1] + a[2]);
}
template <int W> __device__ __forceinline__ double blk_dot(const double (&c)[W], const double (&y)[W], double a0) {  // a0 + sum c y, three partial sums (a dependent v_fma_f64 is 10 cycles)
    double a[3] = {a0, 0.0, 0.0};
#pragma unroll
    for (int uu = 0; uu < W; ++uu) a[uu % 3] = __builtin_fma(c[uu], y[uu], a[uu % 3]);
    return a[0] + (a[1] + a[2]);
}
template <int KIND> __device__ __forceinline__ void band_solve_blocks(Prob<KIND> &pb, int lane) {
    using T = ST<KIND>;
    constexpr int W = T::W, BS = blk_stride<W>();
    static_assert(W <= 9 && 2 * (16 - W) >= W, "lane map below: W chain lanes + (16 - W) Wm rows per 16-lane DPP row, two rows");
    if (lane >= 32) return;
    // Lane map (two DPP rows of 16 lanes; row_newbcast broadcasts inside a row): positions 0 .. W-1 of BOTH rows run the chain (row 1 repeats row 0: the same instructions, it only
    // has to hold r_k for its own broadcasts), positions W .. 15 hold rows of Wm: row 0 the first 16 - W of them, row 1 the rest.
    const int pos = lane & 15, drow = lane >> 4;
    const bool chain = pos < W;
    const int wrow = (pos - W) + drow * (16 - W);   // Wm row of a non-chain lane
    const bool wm = !chain && wrow < W;
    const int l = chain ? pos : (wm ? W + wrow : 0);  // row of the block's stacked 2W x W storage [-M_k; Wm_k] this lane multiplies with
    const int nblk = (pb.n + W - 1) / W;
    const double *Mb = pb.Lb;
    double *wk = pb.wk;
    constexpr auto seq = std::make_integer_sequence<int, W>{};
    // ---- forward: r_{k+1} = b_{k+1} + (-M_k) r_k on the chain lanes (row l of -M_k), and IN THE SAME FMAs on the Wm lanes z_k = Wm_k r_k: both products take the broadcast
    //      r_k as their operand.  z_k is parked in wk where b_k was; S_k r_k = Wm_k' z_k is finished by the backward sweep.  The explicit product S = T^-T D^-1 T^-1 (round 3)
    //      loses the residual of the substitution (tools/tension_block_model.py: 1e-15 .. 5e-14 relative against 3e-16, exactly rounded entries included; the factored form gives
    //      3e-16 again), and the ADMM of a few instances amplifies a residual by 1e6: with the factored form the iterates are back within 1e-7 of the oracle's. ----
    {
        BlkSet<W> A, B;
        auto load = [&](BlkSet<W> &s, int k) {  // (k == nblk: the look-ahead reads the block behind the last one — inside the padding, never used)
            const double *mb = Mb + (size_t)k * BS + l * W;
#pragma unroll
            for (int uu = 0; uu < W; ++uu) s.m[uu] = mb[uu];
            const double bn = wk[(k + 1) * W + (chain ? pos : 0)];
            s.bn = chain ? bn : 0.0;
        };
        // One block: once this set's first value is known to have arrived, request the OTHER set (the block after this one), then the arithmetic.  The compiler's s_waitcnt
        // inside a loop waits for every outstanding LDS request at the first use of a loaded register; placed here, all that is outstanding is this block's own set,
        // requested a whole block earlier.
        auto step = [&](const BlkSet<W> &s, BlkSet<W> &other, int k, double &r, bool ahead) {
            asm volatile("" ::"v"(s.m[0]));
            __builtin_amdgcn_sched_barrier(0);
            if (ahead) load(other, k + 1);
            __builtin_amdgcn_sched_barrier(0);
            const double rn = blk_dot_bcast<W>(s.m, r, s.bn, seq);  // chain lanes: r_{k+1}; Wm lanes: z_k
            if (wm) wk[k * W + wrow] = rn;
            r = rn;
        };
        double r = wk[chain ? pos : 0];
        asm volatile("" : "+v"(r));
        load(A, 0);
        int k = 0;
        for (; k + 1 < nblk; k += 2) {
            step(A, B, k, r, true);
            step(B, A, k + 1, r, true);
        }
        if (k < nblk) step(A, B, k, r, false);
    }
    if (lane >= W) return;
    // ---- backward: x_k = g_k + (-M_k)' x_{k+1} (column l of -M_k on lane l, x_{k+1} broadcast inside the FMAs) with g_k = Wm_k' z_k (column l of Wm_k; z_k read back as an LDS
    //      broadcast: off the chain); the last block's x is its g ----
    {
        struct Bk { double m[W], c[W], z; };
        Bk A, B;
        const int lc = lane;
        auto load = [&](Bk &s, int k) {
            const int kk = k > 0 ? k : 0;
            const double *mb = Mb + (size_t)kk * BS + lc;
#pragma unroll
            for (int uu = 0; uu < W; ++uu) { s.m[uu] = mb[uu * W]; s.c[uu] = mb[W * W + uu * W]; }
            s.z = wk[kk * W + lc];  // z_k, one entry per lane: broadcast inside the FMAs like x
        };
        auto step = [&](Bk &s, Bk &other, int k, double &x, bool ahead) {
            asm volatile("" ::"v"(s.m[0]));
            __builtin_amdgcn_sched_barrier(0);
            if (ahead) load(other, k - 1);
            __builtin_amdgcn_sched_barrier(0);
            double a[3] = {0.0, 0.0, 0.0};
            blk_acc_bcast<W>(a, s.c, s.z, seq);  // g_k = Wm_k' z_k (does not wait for x)
            blk_acc_bcast<W>(a, s.m, x, seq);    // + (-M_k)' x_{k+1}
            x = a[0] + (a[1] + a[2]);
            wk[k * W + lc] = x;
        };
        double x;
        {   // the last block: x = g = Wm' z
            Bk L_;
            load(L_, nblk - 1);
            x = blk_dot_bcast<W>(L_.c, L_.z, 0.0, seq);
            wk[(nblk - 1) * W + lc] = x;
        }
        asm volatile("" : "+v"(x));
        int k = nblk - 2;
        load(A, k);
        for (; k >= 1; k -= 2) {
            step(A, B, k, x, true);
            step(B, A, k - 1, x, true);
        }
        if (k == 0) step(A, B, 0, x, false);
    }
}


// L D L' x = wk, in place, PARTITIONED over the lanes of the wave (narrow bands, W <= 4): lane t owns the c consecutive rows [t c, (t + 1) c), c a multiple of W.
// A banded substitution is a linear recurrence of order W: with W values as the state entering a chunk (forward sweep, column form: the W pending sums
// from the columns before it; backward sweep: the W solution values after it), the state leaving it is s_out = p + Phi s_in, where p comes from the chunk's
// own right-hand side with s_in = 0 and column k of Phi from a unit incoming state and a zero right-hand side.  So: (1) every lane runs the 1 + W recurrences over its own rows with rolling windows (no per-row storage), (2) the affine maps
// (Phi_t, p_t) are composed by a Kogge-Stone scan over the lanes, which hands every lane its true incoming state, (3) every lane runs the recurrence once
// more from that state and stores its rows.  The backward sweep (L' x = z) is the same recurrence from the far end (incoming state from the next
// lane).  2 x (W + 2) passes over c rows per lane instead of 2 x n dependent columns on one lane: the substitution was 2/3 of a TENSION2 solve.
// The factor is SPD-banded (P + sigma I + A' rho A): its homogeneous solutions decay, the products of the Phi_t stay bounded.
template <int W> struct AffW { double M[W][W], p[W]; };
template <int W> __device__ __forceinline__ AffW<W> aff_compose(const AffW<W> &first, const AffW<W> &second) {  // second o first
    // k outermost: W (W + 1) independent accumulators take one term each per round (measured: no faster than summing each output's W terms back to back —
    // the scan is bound by the ds_bpermute traffic and the loads, not by FMA latency — kept for the shorter dependent chains).
    AffW<W> o;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        o.p[i] = second.p[i];
#pragma unroll
        for (int j = 0; j < W; ++j) o.M[i][j] = 0;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
#pragma unroll
        for (int i = 0; i < W; ++i) {
            o.p[i] += second.M[i][k] * first.p[k];
#pragma unroll
            for (int j = 0; j < W; ++j) o.M[i][j] += second.M[i][k] * first.M[k][j];
        }
    }
    return o;
}
template <int W> __device__ __forceinline__ void aff_apply(AffW<W> &e, const double (&pin)[W]) {  // e.p <- e.p + e.M pin (the map applied to a state)
    double acc[W];
#pragma unroll
    for (int i = 0; i < W; ++i) acc[i] = e.p[i];
#pragma unroll
    for (int k = 0; k < W; ++k) {
#pragma unroll
        for (int i = 0; i < W; ++i) acc[i] += e.M[i][k] * pin[k];
    }
#pragma unroll
    for (int i = 0; i < W; ++i) e.p[i] = acc[i];
}
template <int W, bool UP> __device__ __forceinline__ AffW<W> aff_shift(const AffW<W> &e, int d) {
    AffW<W> o;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        o.p[i] = UP ? __shfl_up(e.p[i], d) : __shfl_down(e.p[i], d);
#pragma unroll
        for (int j = 0; j < W; ++j) o.M[i][j] = UP ? __shfl_up(e.M[i][j], d) : __shfl_down(e.M[i][j], d);
    }
    return o;
}
template <int KIND, int NWV = 1> __device__ __forceinline__ void band_solve_scan(Prob<KIND> &pb, int lane, int c) {
    using T = ST<KIND>;
    constexpr int W = T::W, LS = Prob<KIND>::LS;
    static_assert(W <= 4, "wide bands keep band_solve_lanes");
    const bool pad = pb.cch != 0;
    const int nl = (pb.n + c - 1) / c;  // lanes that own rows (the last one may own padding rows: zero factor columns, zero right-hand side)
    const int j0 = lane * c;
    const bool own = lane < nl;
    // Both sweeps read the lane's OWN factor columns only (column j: 1/d_j, then L[j+1][j] .. L[j+W][j]): the forward sweep in column form (a column's value
    // is final when the W pending sums ahead of it have taken the columns before), the backward sweep in row form on L'.  A block of W columns is loaded
    // in one go, then its W steps run from registers: one LDS round trip per block, not per row (a wave alone on its SIMD has nothing else to hide it).
    const double *__restrict__ Lc = pb.Lb + (size_t)j0 * LS + (pad ? lane : 0);                 // own columns (chunk-padded layout: see Prob::cch)
    double *wc = (pad ? pb.wkp + lane * pb.wpad : pb.wk) + j0;                                 // own rows: right-hand side in, z = D^-1 y in between
    double *wout = pb.wk + j0;                                                                 // the solution goes to the natural layout
    AffW<W> el;
    auto identity = [&]() {
#pragma unroll
        for (int i = 0; i < W; ++i) {
            el.p[i] = 0;
#pragma unroll
            for (int k = 0; k < W; ++k) el.M[i][k] = i == k ? 1.0 : 0.0;
        }
    };
    // ================= forward: L y = b, stored as z = D^-1 y =================
    // State of a chunk: the W pending sums u_k = - sum over the columns before the chunk of L[j0 + k][.] y[.], k = 0 .. W - 1.  The window up[k] holds
    // b + (pending sum) of row (current + k): row r's value is up[0] when its turn comes, and it waits for ONE FMA of row r - 1.
    identity();
    if (own) {
        double up[W], uh[W][W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            up[k] = wc[k];
#pragma unroll
            for (int i = 0; i < W; ++i) uh[k][i] = i == k ? 1.0 : 0.0;  // uh[k][i]: homogeneous solution for the unit incoming sum i
        }
        for (int jb = 0; jb < c; jb += W) {
            double L[W][W], bn[W];
            const bool more = jb + W < c;  // (the sums handed to the next chunk do not hold its right-hand side)
#pragma unroll
            for (int uu = 0; uu < W; ++uu) {
#pragma unroll
                for (int d = 1; d <= W; ++d) L[uu][d - 1] = Lc[(jb + uu) * LS + d];
                bn[uu] = more ? wc[jb + W + uu] : 0.0;
            }
#pragma unroll
            for (int uu = 0; uu < W; ++uu) {
                const double y = up[0];
                double yh[W];
#pragma unroll
                for (int i = 0; i < W; ++i) yh[i] = uh[0][i];
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    up[k] = (k + 1 < W ? up[k + 1 < W ? k + 1 : k] : bn[uu]) - L[uu][k] * y;
#pragma unroll
                    for (int i = 0; i < W; ++i) uh[k][i] = (k + 1 < W ? uh[k + 1 < W ? k + 1 : k][i] : 0.0) - L[uu][k] * yh[i];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            el.p[k] = up[k];
#pragma unroll
            for (int i = 0; i < W; ++i) el.M[k][i] = uh[k][i];
        }
    }
    // inclusive scan over the lanes: el_t <- el_t o el_{t-1} o ... o el_0
    for (int d = 1; d < nl; d <<= 1) {
        if (2 * d < nl) {
            const AffW<W> prev = aff_shift<W, true>(el, d);
            if (lane >= d) el = aff_compose<W>(prev, el);
        } else {  // last level: only the sums are read afterwards
            double pp[W];
#pragma unroll
            for (int k = 0; k < W; ++k) pp[k] = __shfl_up(el.p[k], d);
            if (lane >= d) aff_apply<W>(el, pp);
        }
    }
    {
        // incoming sums of lane t = outgoing sums of lane t - 1 from the zero state = p of its inclusive prefix
        double sin_[W];
#pragma unroll
        for (int i = 0; i < W; ++i) { const double v = __shfl_up(el.p[i], 1); sin_[i] = lane > 0 ? v : 0.0; }
        if (own) {
            double up[W];
#pragma unroll
            for (int k = 0; k < W; ++k) up[k] = sin_[k] + wc[k];
            for (int jb = 0; jb < c; jb += W) {
                double L[W][W], dinv[W], bn[W], z[W];
                const bool more = jb + W < c;
#pragma unroll
                for (int uu = 0; uu < W; ++uu) {
                    dinv[uu] = Lc[(jb + uu) * LS];
#pragma unroll
                    for (int d = 1; d <= W; ++d) L[uu][d - 1] = Lc[(jb + uu) * LS + d];
                    bn[uu] = more ? wc[jb + W + uu] : 0.0;
                }
#pragma unroll
                for (int uu = 0; uu < W; ++uu) {
                    const double y = up[0];
                    z[uu] = y * dinv[uu];
#pragma unroll
                    for (int k = 0; k < W; ++k) up[k] = (k + 1 < W ? up[k + 1 < W ? k + 1 : k] : bn[uu]) - L[uu][k] * y;
                }
#pragma unroll
                for (int uu = 0; uu < W; ++uu) wc[jb + uu] = z[uu];
            }
        }
    }
    band_sync<NWV>();
    // ================= backward: L' x = z,  x_j = z_j - sum_{d = 1..W} L[j + d][j] x_{j + d} =================
    // State of a chunk: the W values after it, x[j0 + c + k].  The window xw[k] holds x of row (current + 1 + k); the newest enters every sum last.
    identity();
    if (own) {
        double xw[W], xh[W][W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            xw[k] = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) xh[k][i] = i == k ? 1.0 : 0.0;
        }
        for (int jb = c - W; jb >= 0; jb -= W) {
            double L[W][W], z[W];
#pragma unroll
            for (int uu = 0; uu < W; ++uu) {
#pragma unroll
                for (int d = 1; d <= W; ++d) L[uu][d - 1] = Lc[(jb + uu) * LS + d];
                z[uu] = wc[jb + uu];
            }
#pragma unroll
            for (int uu = W - 1; uu >= 0; --uu) {
                double x = z[uu], xn[W];
#pragma unroll
                for (int i = 0; i < W; ++i) xn[i] = 0;
#pragma unroll
                for (int d = W; d >= 1; --d) {
                    x -= L[uu][d - 1] * xw[d - 1];
#pragma unroll
                    for (int i = 0; i < W; ++i) xn[i] -= L[uu][d - 1] * xh[d - 1][i];
                }
#pragma unroll
                for (int k = W - 1; k >= 1; --k) {
                    xw[k] = xw[k - 1];
#pragma unroll
                    for (int i = 0; i < W; ++i) xh[k][i] = xh[k - 1][i];
                }
                xw[0] = x;
#pragma unroll
                for (int i = 0; i < W; ++i) xh[0][i] = xn[i];
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            el.p[k] = xw[k];
#pragma unroll
            for (int i = 0; i < W; ++i) el.M[k][i] = xh[k][i];
        }
    }
    for (int d = 1; d < nl; d <<= 1) {  // el_t <- el_t o el_{t+1} o ... o el_{nl-1}
        if (2 * d < nl) {
            const AffW<W> nxt = aff_shift<W, false>(el, d);
            if (lane + d < nl) el = aff_compose<W>(nxt, el);
        } else {
            double pp[W];
#pragma unroll
            for (int k = 0; k < W; ++k) pp[k] = __shfl_down(el.p[k], d);
            if (lane + d < nl) aff_apply<W>(el, pp);
        }
    }
    {
        double sin_[W];
#pragma unroll
        for (int i = 0; i < W; ++i) { const double v = __shfl_down(el.p[i], 1); sin_[i] = lane + 1 < nl ? v : 0.0; }
        if (own) {
            double xw[W];
#pragma unroll
            for (int k = 0; k < W; ++k) xw[k] = sin_[k];
            for (int jb = c - W; jb >= 0; jb -= W) {
                double L[W][W], z[W];
#pragma unroll
                for (int uu = 0; uu < W; ++uu) {
#pragma unroll
                    for (int d = 1; d <= W; ++d) L[uu][d - 1] = Lc[(jb + uu) * LS + d];
                    z[uu] = wc[jb + uu];
                }
#pragma unroll
                for (int uu = W - 1; uu >= 0; --uu) {
                    double x = z[uu];
#pragma unroll
                    for (int d = W; d >= 1; --d) x -= L[uu][d - 1] * xw[d - 1];
#pragma unroll
                    for (int k = W - 1; k >= 1; --k) xw[k] = xw[k - 1];
                    xw[0] = x;
                    wout[jb + uu] = x;
                }
            }
        }
    }
}


// reductions over the block: NWV waves per QP (see smooth_kernel)
template <int NWV> __device__ __forceinline__ double blk_max(double v, double *red) {
    v = wave_max(v);
    if constexpr (NWV > 1) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        v = red[0];
#pragma unroll
        for (int k = 1; k < NWV; ++k) v = fmax(v, red[k]);
    }
    return v;
}
// sum over j < n of term(j), in an order that does not depend on the number of waves: the terms of class w = (j / 64) mod 8 are added up in ascending j per
// lane and reduced over the lanes (s_w), then s_0 + s_1 + ... + s_7 left to right.  A block of 1, 4 or 8 waves owns 8, 2 or 1 classes per thread, so a QP's
// iterates are bit-identical whichever variant the launcher picks for its batch.  term(j) is called exactly once per j, by thread j mod (64 NWV) (side
// effects — stores to row j, running maxima — are fine).
template <int NWV, class F> __device__ __forceinline__ double blk_sum_rows(int n, int tid, double *red, F term) {
    constexpr int NC = 8 / NWV;  // classes per thread
    static_assert(NWV == 1 || NWV == 2 || NWV == 4 || NWV == 8, "eight classes");
    const int lane = tid & 63, wv = tid >> 6;
    double t = 0;
    if constexpr (NWV > 1) __syncthreads();  // (earlier readers of red)
#pragma unroll 1
    for (int q = 0; q < NC; ++q) {  // one class at a time: the terms are inlined once, with one accumulator
        double acc = 0;
#pragma unroll 1
        for (int j = 64 * (wv + NWV * q) + lane; j < n; j += 512) acc += term(j);
        const double sw = wave_sum(acc);
        if constexpr (NWV == 1) t = q ? t + sw : sw;
        else if (lane == 0) red[wv + NWV * q] = sw;
    }
    if constexpr (NWV > 1) {
        __syncthreads();
        t = red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += red[w];
    }
    return t;
}

#define PO_TICK(slot)                                                   \
    do {                                                                \
        if (a.dbg_cycles) { const long long t_ = clock64(); acc_[slot] += t_ - tprev_; tprev_ = t_; } \
    } while (0)
// NWV waves per QP, WPE resident waves per SIMD the registers are allocated for.
//  <1, 1>: one wave per QP — many small QPs per CU (each SIMD holds several QPs' waves).
//  <4, 3>, <4, 2>, <8, 2> (narrow bands only): every row / variable loop of an iteration (right-hand side, update, residuals, scaling, assembly) runs on the
//  whole block; the substitution and the pivot loop stay on wave 0.  Taken whenever the LDS footprint leaves a CU with three QPs or fewer (one wave per QP
//  would idle the other SIMDs) and for batches that fit the chip at once — a planner's own call is ONE QP.
template <int KIND, int NWV, int WPE> __global__ __launch_bounds__(64 * NWV, WPE) void smooth_kernel(DevSmooth a) {
    constexpr int NTS = 64 * NWV;
    __shared__ double red_[8];
    long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = a.dbg_cycles ? clock64() : 0;
    using T = ST<KIND>;
    constexpr int W = T::W, LS = Prob<KIND>::LS, KA = T::KA, WP = T::WP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int b = perm_index(blockIdx.x, a.perm_bits, a.B), lane = threadIdx.x;  // (lane = thread index of the block: 0 .. NTS - 1)  // iteration counts vary 10x between instances (post QP)
    const int P = a.n_points ? a.n_points[b] : a.P;
    const size_t o = (size_t)b * a.P;
    po_info info{};
    info.status = PO_STATUS_UNSOLVED;
    if (P < T::MINP || P > a.P) {  // misuse inside a device-pointer batch: no abort, flagged per instance
        for (int i = lane; i < a.P; i += NTS) {
            a.out_x[o + i] = 0;
            if (a.out_y) a.out_y[o + i] = 0;
            if (a.out_s) a.out_s[o + i] = 0;
        }
        if (lane == 0) a.info[b] = info;
        return;
    }
    Prob<KIND> pb;
    const int n = pb.n = T::n(P), m = pb.m = T::m(P), np = pb.np = n + col_pad<W>();
    int cfull = 0;  // the LDS block is sized for a.P points
    if constexpr (W <= 4) cfull = scan_chunk<W>(T::n(a.P));
    const bool lds_pad = scan_padded(cfull);
    // the serial parts (pivot loop, substitution) run on wave 0 of the block (rotating the wave with the block index, to spread the QPs of a CU over the
    // SIMDs by hand, measured 10 % slower: the dispatcher already does it)
    {
        double *dp = reinterpret_cast<double *>(smem_raw);
        pb.Lb = dp; dp += factor_doubles<W>((size_t)np, a.blocked != 0) + (lds_pad ? kChunkPad : 0);
        pb.Av = dp; dp += KA * m;
        pb.x = dp; dp += n;
        pb.wk = dp; dp += np;
        pb.wkp = nullptr;
        if constexpr (W <= 4) { if (lds_pad) { pb.wkp = dp; dp += np + kChunkPad; } }
        pb.v = dp; dp += m;
        pb.l = dp; dp += m;
        pb.u = dp; dp += m;
        uint16_t *hp = reinterpret_cast<uint16_t *>(dp);
        pb.Ac = hp; hp += KA * m;
        pb.Tl = hp; hp += kKT * n;
        pb.Tc = reinterpret_cast<int *>(reinterpret_cast<uintptr_t>(hp + 1) & ~(uintptr_t)3);
        pb.cls = reinterpret_cast<uint8_t *>(pb.Tc + n);
        double *sc = a.scratch + (size_t)b * a.scratch_stride;
        pb.Pb = sc; pb.Dv = sc + (size_t)(WP + 1) * T::n(a.P); pb.Ev = pb.Dv + T::n(a.P);
        pb.q = pb.Ev + T::m(a.P); pb.tm = pb.q + T::n(a.P);  // q[j] is only ever touched by lane j % 64; tm crosses lanes behind __syncthreads
    }
    // narrow bands (W = 3, 4): the substitution is partitioned over the lanes of a wave (chunks of c rows, c a multiple of W, every row's read-ahead inside
    // the zero padding); QPs too small to give every chunk W rows keep the single-lane window
    int cscan = 0;
    pb.cch = 0; pb.wpad = 0; pb.rcch = 0; pb.bst = 0;
    if constexpr (W >= 8) pb.bst = a.blocked ? blk_stride<W>() : 0;
    if constexpr (W <= 4) {
        cscan = a.seq_band ? 0 : scan_chunk<W>(n);
        if (lds_pad && scan_padded(cscan) && !a.nopad) { pb.cch = cscan; pb.wpad = (cscan & 1) ? 0 : 1; pb.rcch = 1.0f / (float)cscan; }
    }
    for (int i = lane; i < (int)factor_doubles<W>((size_t)np, a.blocked != 0) + (lds_pad ? kChunkPad : 0); i += NTS) pb.Lb[i] = 0;  // (the P band is parked here in the natural layout until the first factorisation)
    for (int i = lane; i < np; i += NTS) pb.wk[i] = 0;
    if (pb.wkp) for (int i = lane; i < np + kChunkPad; i += NTS) pb.wkp[i] = 0;
    for (int i = lane; i < n; i += NTS) { pb.x[i] = 0; pb.Tc[i] = 0; }
    for (int i = lane; i < m; i += NTS) pb.v[i] = 0;
    __syncthreads();
    assemble<KIND>(a, pb, b, P, lane, NTS);
    __syncthreads();

    // ---- transposed index lists (variable -> rows), deterministic: count with LDS atomics, then sort each short list ----
    for (int e = lane; e < KA * m; e += NTS) {
        const int s = e / m, r = e - s * m;
        if (pb.Av[e] != 0.0) {
            const int c = pb.Ac[e];
            const int pos = atomicAdd(&pb.Tc[c], 1);
            if (pos < kKT) pb.Tl[c * kKT + pos] = (uint16_t)(r * 4 + s);
        }
    }
    __syncthreads();
    for (int j = lane; j < n; j += NTS) {
        const int cnt = pb.Tc[j] < kKT ? pb.Tc[j] : kKT;
        pb.Tc[j] = cnt;
        for (int i = 1; i < cnt; ++i) {
            const uint16_t key = pb.Tl[j * kKT + i];
            int k = i - 1;
            while (k >= 0 && pb.Tl[j * kKT + k] > key) { pb.Tl[j * kKT + k + 1] = pb.Tl[j * kKT + k]; --k; }
            pb.Tl[j * kKT + k + 1] = key;
        }
    }
    __syncthreads();

    // ---- data validation (OSQP validate_data: l <= u, else setup fails and the reference's initSolver() returns false) ----
    {
        // ... and a NaN anywhere in the assembled data (non-finite inputs): a NaN bound would vanish silently (fmin / fmax drop a NaN operand: the row would
        // act as unbounded and the QP come back "solved"); flagged PO_STATUS_NON_FINITE, like the path QP does
        double bad = 0, nonfin = 0;
        for (int r = lane; r < m; r += NTS) {
            bad = fmax(bad, pb.l[r] > pb.u[r] ? 1.0 : 0.0);
            int nf = nan_bits(pb.l[r]) | nan_bits(pb.u[r]);
#pragma unroll
            for (int s = 0; s < KA; ++s) nf |= nonfinite_bits(pb.Av[s * m + r]);
            if (nf) nonfin = 1.0;
        }
        for (int j = lane; j < n; j += NTS) {
            int nf = nonfinite_bits(pb.q[j]);
#pragma unroll
            for (int d = 0; d <= WP; ++d) nf |= nonfinite_bits(pb.Lb[j * LS + d]);
            if (nf) nonfin = 1.0;
        }
        bad = blk_max<NWV>(bad, red_);
        nonfin = blk_max<NWV>(nonfin, red_);
        if (bad > 0 || nonfin > 0) {
            for (int i = lane; i < a.P; i += NTS) {
                a.out_x[o + i] = 0;
                if (a.out_y) a.out_y[o + i] = 0;
                if (a.out_s) a.out_s[o + i] = 0;
            }
            if (a.raw) for (int j = lane; j < a.raw_stride; j += NTS) a.raw[(size_t)b * a.raw_stride + j] = 0;
            info.status = nonfin > 0 ? PO_STATUS_NON_FINITE : PO_STATUS_PRIMAL_INFEASIBLE;
            info.rho = a.rho0;
            if (lane == 0) a.info[b] = info;
            return;
        }
    }

    // ---- Ruiz equilibration (OSQP scale_data), P band parked in Lb, D in wk-free storage: Dv/Ev in scratch ----
    double cscale = 1.0;
    for (int j = lane; j < n; j += NTS) pb.Dv[j] = 1.0;
    for (int r = lane; r < m; r += NTS) pb.Ev[r] = 1.0;
    for (int pass = 0; pass < a.scaling; ++pass) {
        for (int j = lane; j < n; j += NTS) {  // column inf-norm of [P; A]
            double cn = 0;
#pragma unroll
            for (int d = 0; d <= WP; ++d) {
                cn = fmax(cn, fabs(pb.Lb[j * LS + d]));
                if (d > 0 && j - d >= 0) cn = fmax(cn, fabs(pb.Lb[(j - d) * LS + d]));
            }
            const int cnt = pb.Tc[j];
            for (int t = 0; t < cnt; ++t) {
                const int code = pb.Tl[j * kKT + t];
                cn = fmax(cn, fabs(pb.Av[(code & 3) * m + (code >> 2)]));
            }
            pb.wk[j] = 1.0 / sqrt(lim_scaling(cn));
        }
        for (int r = lane; r < m; r += NTS) {
            double rn = 0;
#pragma unroll
            for (int s = 0; s < KA; ++s) rn = fmax(rn, fabs(pb.Av[s * m + r]));
            pb.tm[r] = 1.0 / sqrt(lim_scaling(rn));
        }
        __syncthreads();
        for (int j = lane; j < n; j += NTS) {
            const double dj = pb.wk[j];
#pragma unroll
            for (int d = 0; d <= WP; ++d)
                if (j + d < n) pb.Lb[j * LS + d] *= dj * pb.wk[j + d];
            pb.q[j] *= dj;
            pb.Dv[j] *= dj;
        }
        for (int r = lane; r < m; r += NTS) {
            const double er = pb.tm[r];
#pragma unroll
            for (int s = 0; s < KA; ++s) pb.Av[s * m + r] *= pb.wk[pb.Ac[s * m + r]] * er;
            pb.Ev[r] *= er;
        }
        __syncthreads();
        double qn = 0;  // cost scaling: mean column norm of P vs ||q||_inf
        double csum = blk_sum_rows<NWV>(n, lane, red_, [&](int j) {
            double cn = 0;
#pragma unroll
            for (int d = 0; d <= WP; ++d) {
                cn = fmax(cn, fabs(pb.Lb[j * LS + d]));
                if (d > 0 && j - d >= 0) cn = fmax(cn, fabs(pb.Lb[(j - d) * LS + d]));
            }
            qn = fmax(qn, fabs(pb.q[j]));
            return cn;
        }) / n;
        qn = lim_scaling(blk_max<NWV>(qn, red_));
        double ct = csum > qn ? csum : qn;
        ct = 1.0 / lim_scaling(ct);
        __syncthreads();
        for (int j = lane; j < n; j += NTS) {
#pragma unroll
            for (int d = 0; d <= WP; ++d) pb.Lb[j * LS + d] *= ct;
            pb.q[j] *= ct;
        }
        cscale *= ct;
        __syncthreads();
    }
    const double cinv = 1.0 / cscale;
    for (int j = lane; j < n; j += NTS) {  // park the scaled P band in HBM; the band storage becomes the factor
#pragma unroll
        for (int d = 0; d <= WP; ++d) pb.Pb[(size_t)d * n + j] = pb.Lb[j * LS + d];
    }
    for (int r = lane; r < m; r += NTS) {
        const double e = pb.Ev[r];
        const double lo = pb.l[r] * e, hi = pb.u[r] * e;
        pb.l[r] = lo; pb.u[r] = hi;
        pb.cls[r] = (uint8_t)((lo < -kInfThresh && hi > kInfThresh) ? 2u : ((hi - lo < kRhoTol) ? 1u : 0u));  // set_rho_vec
    }
    for (int i = lane; i < np; i += NTS) pb.wk[i] = 0;
    __threadfence_block();
    __syncthreads();

    PO_TICK(0);
    double rho = fmin(fmax(a.rho0, kRhoMin), kRhoMax);
    factorise<KIND, NWV>(pb, rho, a.sigma, lane, lane);
    PO_TICK(1);

    // ---- ADMM (OSQP osqp_solve, cold start) ----
    int iter = 0, n_refactor = 0, status = PO_STATUS_UNSOLVED;
    double pri_res = 0, dua_res = 0;
    const double alpha = a.alpha, sigma = a.sigma;
    for (iter = 1; iter <= a.max_iter; ++iter) {
        const bool first = iter == 1;
        const bool can_check = a.check_every > 0 && (iter % a.check_every == 0);
        const bool can_adapt = a.adapt_every > 0 && (iter % a.adapt_every == 0);
        const bool last = iter == a.max_iter;
        const bool term = can_check || last;
        for (int j = lane; j < n; j += NTS) {  // rhs = sigma x - q + A' rho (2 z - v)
            double acc = sigma * pb.x[j] - pb.q[j];
            const int cnt = pb.Tc[j];
            for (int t = 0; t < cnt; ++t) {
                const int code = pb.Tl[j * kKT + t], r = code >> 2, s = code & 3;
                const double vv = pb.v[r];
                const double z = first ? 0.0 : clipd(vv, pb.l[r], pb.u[r]);
                acc += pb.Av[s * m + r] * (rho_row(pb, r, rho) * (2.0 * z - vv));
            }
            if (pb.cch) pb.wkp[pb.pidx(j)] = acc;
            else pb.wk[j] = acc;
        }
        __syncthreads();
        PO_TICK(2);
        // wide bands (TENSION, W = 9): one pending row per lane, one FMA per lane and column (35 instead of 40 ms per 4096 QPs); narrow bands (W = 3, 4): the
        // single-lane window is as fast or faster (measured: TENSION2 14.6 vs 17.2 ms) — the column-to-column latency, not the FMA count, bounds both
        if constexpr (ST<KIND>::W >= 8) {
            if (lane < 64) {
                if (pb.bst) band_solve_blocks<KIND>(pb, lane);
                else band_solve_lanes<KIND>(pb, lane);  // QPs whose block layout does not fit the LDS, and the developer A/B switch "smooth_seq"
            }
        }
        else {
            if (cscan) { if (NWV == 1 || lane < 64) band_solve_scan<KIND, NWV>(pb, lane, cscan); }
            else if (lane == 0) band_solve<KIND>(pb);
        }
        __syncthreads();
        PO_TICK(3);
        for (int r = lane; r < m; r += NTS) {  // ztilde = A xtilde ; v += alpha (ztilde - z)
            double zt = 0;
#pragma unroll
            for (int s = 0; s < KA; ++s) zt += pb.Av[s * m + r] * pb.wk[pb.Ac[s * m + r]];
            const double vv = pb.v[r];
            const double z = first ? 0.0 : clipd(vv, pb.l[r], pb.u[r]);
            const double vn = vv + alpha * (zt - z);
            pb.v[r] = vn;
            if (term) pb.tm[r] = rho_row(pb, r, rho) * ((vn - clipd(vn, pb.l[r], pb.u[r])) - (vv - z));  // delta y of this iteration
        }
        __syncthreads();
        for (int j = lane; j < n; j += NTS) {
            const double xo = pb.x[j];
            const double xn = xo + alpha * (pb.wk[j] - xo);
            pb.x[j] = xn;
            pb.wk[j] = xn - xo;  // delta x (dual-infeasibility certificate)
        }
        __syncthreads();
        PO_TICK(4);
        if (!(term || can_adapt)) continue;

        // ---- update_info: residuals, unscaled (termination) and scaled (rho estimate) ----
        double pr = 0, nz = 0, nAx = 0, prs = 0, nzs = 0, nAxs = 0;
        int bad = 0;  // a non-finite primal or dual residual term: fmax drops a NaN operand, so NaN iterates (non-finite inputs) would otherwise read as converged
        for (int r = lane; r < m; r += NTS) {
            double ax = 0;
#pragma unroll
            for (int s = 0; s < KA; ++s) ax += pb.Av[s * m + r] * pb.x[pb.Ac[s * m + r]];
            const double z = clipd(pb.v[r], pb.l[r], pb.u[r]);
            bad |= nonfinite_bits(ax - z);
            const double ei = 1.0 / pb.Ev[r];
            prs = fmax(prs, fabs(ax - z)); nzs = fmax(nzs, fabs(z)); nAxs = fmax(nAxs, fabs(ax));
            pr = fmax(pr, ei * fabs(ax - z)); nz = fmax(nz, ei * fabs(z)); nAx = fmax(nAx, ei * fabs(ax));
        }
        double du = 0, nq = 0, nAty = 0, nPx = 0, dus = 0, nqs = 0, nAtys = 0, nPxs = 0;
        for (int j = lane; j < n; j += NTS) {
            double px = 0;
#pragma unroll
            for (int d = 0; d <= WP; ++d) {
                if (j + d < n) px += pb.Pb[(size_t)d * n + j] * pb.x[j + d];
                if (d > 0 && j - d >= 0) px += pb.Pb[(size_t)d * n + j - d] * pb.x[j - d];
            }
            double aty = 0;
            const int cnt = pb.Tc[j];
            for (int t = 0; t < cnt; ++t) {
                const int code = pb.Tl[j * kKT + t], r = code >> 2, s = code & 3;
                const double vv = pb.v[r];
                aty += pb.Av[s * m + r] * (rho_row(pb, r, rho) * (vv - clipd(vv, pb.l[r], pb.u[r])));
            }
            const double qq = pb.q[j], di = 1.0 / pb.Dv[j];
            const double dres = px + qq + aty;
            bad |= nonfinite_bits(dres);
            dus = fmax(dus, fabs(dres)); nqs = fmax(nqs, fabs(qq)); nAtys = fmax(nAtys, fabs(aty)); nPxs = fmax(nPxs, fabs(px));
            du = fmax(du, di * fabs(dres)); nq = fmax(nq, di * fabs(qq)); nAty = fmax(nAty, di * fabs(aty)); nPx = fmax(nPx, di * fabs(px));
        }
        pr = blk_max<NWV>(pr, red_); nz = blk_max<NWV>(nz, red_); nAx = blk_max<NWV>(nAx, red_); prs = blk_max<NWV>(prs, red_); nzs = blk_max<NWV>(nzs, red_); nAxs = blk_max<NWV>(nAxs, red_);
        du = blk_max<NWV>(du, red_); nq = blk_max<NWV>(nq, red_); nAty = blk_max<NWV>(nAty, red_); nPx = blk_max<NWV>(nPx, red_);
        dus = blk_max<NWV>(dus, red_); nqs = blk_max<NWV>(nqs, red_); nAtys = blk_max<NWV>(nAtys, red_); nPxs = blk_max<NWV>(nPxs, red_);
        pri_res = pr;
        dua_res = cinv * du;
        if (blk_max<NWV>(bad ? 1.0 : 0.0, red_) != 0.0) { status = PO_STATUS_NON_FINITE; break; }
        if (term) {
            const double eps_prim = a.eps_abs + a.eps_rel * fmax(nz, nAx);
            const double eps_dual = a.eps_abs + a.eps_rel * cinv * fmax(fmax(nq, nAty), nPx);
            const bool prim_ok = pri_res < eps_prim, dual_ok = dua_res < eps_dual;  // strict, as OSQP
            bool prim_inf = false, dual_inf = false;
            if (!prim_ok) {  // is_primal_infeasible on delta y
                double ndy = 0;
                const double lhs = blk_sum_rows<NWV>(m, lane, red_, [&](int r) {
                    double d = pb.tm[r];
                    const bool uinf = pb.u[r] > kInfThresh, linf = pb.l[r] < -kInfThresh;
                    if (uinf) d = linf ? 0.0 : fmin(d, 0.0);
                    else if (linf) d = fmax(d, 0.0);
                    pb.tm[r] = d;
                    ndy = fmax(ndy, pb.Ev[r] * fabs(d));
                    return (uinf ? 0.0 : pb.u[r] * fmax(d, 0.0)) + (linf ? 0.0 : pb.l[r] * fmin(d, 0.0));
                });
                ndy = blk_max<NWV>(ndy, red_);
                __syncthreads();
                if (ndy > a.eps_pinf && lhs < -a.eps_pinf * ndy) {
                    double na = 0;
                    for (int j = lane; j < n; j += NTS) {
                        double s2 = 0;
                        const int cnt = pb.Tc[j];
                        for (int t = 0; t < cnt; ++t) {
                            const int code = pb.Tl[j * kKT + t];
                            s2 += pb.Av[(code & 3) * m + (code >> 2)] * pb.tm[code >> 2];
                        }
                        na = fmax(na, fabs(s2) / pb.Dv[j]);
                    }
                    prim_inf = blk_max<NWV>(na, red_) < a.eps_pinf * ndy;
                }
            }
            if (!dual_ok && !prim_inf) {  // is_dual_infeasible on delta x
                double ndx = 0;
                const double qdx = blk_sum_rows<NWV>(n, lane, red_, [&](int j) {
                    ndx = fmax(ndx, pb.Dv[j] * fabs(pb.wk[j]));
                    return pb.q[j] * pb.wk[j];
                });
                ndx = blk_max<NWV>(ndx, red_);
                if (ndx > a.eps_dinf && qdx < -cscale * a.eps_dinf * ndx) {
                    double npdx = 0;
                    for (int j = lane; j < n; j += NTS) {
                        double px = 0;
#pragma unroll
                        for (int d = 0; d <= WP; ++d) {
                            if (j + d < n) px += pb.Pb[(size_t)d * n + j] * pb.wk[j + d];
                            if (d > 0 && j - d >= 0) px += pb.Pb[(size_t)d * n + j - d] * pb.wk[j - d];
                        }
                        npdx = fmax(npdx, fabs(px) / pb.Dv[j]);
                    }
                    if (blk_max<NWV>(npdx, red_) < cscale * a.eps_dinf * ndx) {
                        double viol = 0;
                        for (int r = lane; r < m; r += NTS) {
                            double adx = 0;
#pragma unroll
                            for (int s = 0; s < KA; ++s) adx += pb.Av[s * m + r] * pb.wk[pb.Ac[s * m + r]];
                            adx /= pb.Ev[r];
                            if ((pb.u[r] < kInfThresh && adx > a.eps_dinf * ndx) || (pb.l[r] > -kInfThresh && adx < -a.eps_dinf * ndx)) viol = 1.0;
                        }
                        dual_inf = blk_max<NWV>(viol, red_) == 0.0;
                    }
                }
            }
            if (prim_ok && dual_ok) { status = PO_STATUS_SOLVED; break; }
            if (prim_inf) { status = PO_STATUS_PRIMAL_INFEASIBLE; break; }
            if (dual_inf) { status = PO_STATUS_DUAL_INFEASIBLE; break; }
        }
        PO_TICK(5);
        if (can_adapt) {  // compute_rho_estimate + adapt_rho on the scaled quantities
            const double prn = prs / (fmax(nzs, nAxs) + 1e-10);
            const double drn = dus / (fmax(fmax(nqs, nAtys), nPxs) + 1e-10);
            double rho_new = rho * sqrt(prn / (drn + 1e-10));
            rho_new = fmin(fmax(rho_new, kRhoMin), kRhoMax);
            if (rho_new > rho * a.adapt_tol || rho_new < rho / a.adapt_tol) {
                const double ratio = rho / rho_new;
                for (int r = lane; r < m; r += NTS) {  // keep z and y: v = z + y / rho_new
                    if (pb.cls[r] == 2u) continue;
                    const double vv = pb.v[r], z = clipd(vv, pb.l[r], pb.u[r]);
                    pb.v[r] = z + ratio * (vv - z);
                }
                rho = rho_new;
                __syncthreads();
                factorise<KIND, NWV>(pb, rho, sigma, lane, lane);
                ++n_refactor;
            }
            PO_TICK(1);
        }
    }
    if (iter > a.max_iter) {
        iter = a.max_iter;
        if (status == PO_STATUS_UNSOLVED) status = PO_STATUS_MAX_ITER;
    }
    __syncthreads();

    // ---- unscale, objective, outputs ----
    const double obj = cinv * blk_sum_rows<NWV>(n, lane, red_, [&](int j) {
        double px = 0;
#pragma unroll
        for (int d = 0; d <= WP; ++d) {
            if (j + d < n) px += pb.Pb[(size_t)d * n + j] * pb.x[j + d];
            if (d > 0 && j - d >= 0) px += pb.Pb[(size_t)d * n + j - d] * pb.x[j - d];
        }
        return (0.5 * px + pb.q[j]) * pb.x[j];
    });
    for (int j = lane; j < n; j += NTS) pb.wk[j] = pb.x[j] * pb.Dv[j];
    __syncthreads();
    if (a.raw) {
        double *rw = a.raw + (size_t)b * a.raw_stride;
        for (int j = lane; j < a.raw_stride; j += NTS) rw[j] = 0;
        __syncthreads();
        for (int j = lane; j < n; j += NTS) rw[ref_var<KIND>(j, P)] = pb.wk[j];
    }
    constexpr int NVV = KIND == PO_SMOOTH_TENSION2 ? 4 : 3;
    for (int i = lane; i < a.P; i += NTS) {
        a.out_x[o + i] = i < P ? pb.wk[NVV * i] : 0.0;
        if (a.out_y) a.out_y[o + i] = (KIND != PO_SMOOTH_POST && i < P) ? pb.wk[NVV * i + 1] : 0.0;
        if (KIND == PO_SMOOTH_POST && a.out_s) a.out_s[o + i] = 0.0;
    }
    if (KIND != PO_SMOOTH_POST && a.out_s && lane == 0) {  // running chord length, tension_smoother_2.cpp:208-216
        double tmp_s = 0;
        for (int i = 0; i < a.P; ++i) {
            if (i > 0 && i < P) {
                const double ddx = pb.wk[NVV * i] - pb.wk[NVV * (i - 1)], ddy = pb.wk[NVV * i + 1] - pb.wk[NVV * (i - 1) + 1];
                tmp_s += sqrt(ddx * ddx + ddy * ddy);
            }
            a.out_s[o + i] = i < P ? tmp_s : 0.0;
        }
    }
    PO_TICK(6);
    if (a.dbg_cycles && lane == 0)
        for (int i = 0; i < 8; ++i) a.dbg_cycles[(size_t)b * 8 + i] = acc_[i];
    if (lane == 0) {
        info.status = status; info.iters = iter; info.n_refactor = n_refactor;
        info.r_prim = pri_res; info.r_dual = dua_res; info.rho = rho; info.obj = obj;
        a.info[b] = info;
    }
}

template <int KIND, int NWV, int WPE> static hipError_t launch_kind_w(const DevSmooth &a, hipStream_t st, size_t lds) {
    const size_t dyn = lds - 64;  // lds = dynamic part + the kernel's 64 static bytes (Lds::bytes)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&smooth_kernel<KIND, NWV, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((smooth_kernel<KIND, NWV, WPE>), dim3(a.B), dim3(64 * NWV), dyn, st, a);
    return hipGetLastError();
}
template <int KIND> static hipError_t launch_kind(const DevSmooth &a, hipStream_t st, size_t lds) {
    {
        constexpr int kCUs = 256, kLdsPerCU = 160 * 1024;
        const int by_lds = (int)(kLdsPerCU / (lds ? lds : 1));  // QPs a CU holds
        // measured (tools/smooth_small.py, ms per batch, one wave -> four waves): 4096 QPs of P = 100 (three per CU) 6.3 -> 4.3, of P = 250 (one per CU)
        // 32 -> 17 (eight waves: 15); a single QP 1.03 -> 0.65 (P = 100; eight waves 0.60), 2.0 -> 1.09 (P = 250; eight waves 0.94).  Seven small QPs per
        // CU (post QP, P = 60): 5.5 -> 7.3, one wave per QP stays.
        int waves = a.waves;
        if (waves <= 0) {
            const bool alone = by_lds == 1 || a.B <= kCUs;
            waves = (alone && ST<KIND>::n(a.P) >= 256) ? 8 : ((by_lds <= 4 || a.B <= 3 * kCUs) ? 4 : 1);
        }
        if (waves == 8) return launch_kind_w<KIND, 8, 2>(a, st, lds);
        if (waves == 4) {
            // three QPs per CU need the 168-register allocation (a handful of spilled values); up to two per CU keep the full one
            if (by_lds >= 3 && a.B > 2 * kCUs) return launch_kind_w<KIND, 4, 3>(a, st, lds);
            return launch_kind_w<KIND, 4, 2>(a, st, lds);
        }
    }
    return launch_kind_w<KIND, 1, 1>(a, st, lds);
}

}  // namespace po

// TENSION (W = 9): the block layout of the factor (band_solve_blocks) while two QPs of it fit a CU, the natural layout with the column-by-column
// substitution beyond (P > 111: one QP per CU up to the 160 KB capacity).
extern "C" int po_smooth_blocked(int kind, int P) {
    return kind == PO_SMOOTH_TENSION && po::Lds<PO_SMOOTH_TENSION>::bytes(P, true) <= 80 * 1024 ? 1 : 0;
}
extern "C" size_t po_smooth_lds_bytes(int kind, int P) {
    switch (kind) {
        case PO_SMOOTH_TENSION2: return po::Lds<PO_SMOOTH_TENSION2>::bytes(P, false);
        case PO_SMOOTH_TENSION: return po::Lds<PO_SMOOTH_TENSION>::bytes(P, po_smooth_blocked(kind, P) != 0);
        case PO_SMOOTH_POST: return po::Lds<PO_SMOOTH_POST>::bytes(P, false);
    }
    return 0;
}
extern "C" size_t po_smooth_scratch_doubles(int kind, int P) {
    switch (kind) {
        case PO_SMOOTH_TENSION2: return (size_t)(4 + 3) * po::ST<PO_SMOOTH_TENSION2>::n(P) + 2 * po::ST<PO_SMOOTH_TENSION2>::m(P);
        case PO_SMOOTH_TENSION: return (size_t)(9 + 3) * po::ST<PO_SMOOTH_TENSION>::n(P) + 2 * po::ST<PO_SMOOTH_TENSION>::m(P);
        case PO_SMOOTH_POST: return (size_t)(0 + 3) * po::ST<PO_SMOOTH_POST>::n(P) + 2 * po::ST<PO_SMOOTH_POST>::m(P);
    }
    return 0;
}
extern "C" hipError_t po_launch_smooth(const po::DevSmooth *a0, hipStream_t st) {  // (po_smooth_lds_bytes counts the 64 static bytes too: the dynamic part is that minus 64)
    po::DevSmooth a1 = *a0;
    a1.blocked = (po_smooth_blocked(a1.kind, a1.P) && !a1.seq_band) ? 1 : 0;
    const po::DevSmooth *a = &a1;
    const size_t lds = a1.kind == PO_SMOOTH_TENSION ? po::Lds<PO_SMOOTH_TENSION>::bytes(a1.P, a1.blocked != 0) : po_smooth_lds_bytes(a1.kind, a1.P);
    switch (a->kind) {
        case PO_SMOOTH_TENSION2: return po::launch_kind<PO_SMOOTH_TENSION2>(*a, st, lds);
        case PO_SMOOTH_TENSION: return po::launch_kind<PO_SMOOTH_TENSION>(*a, st, lds);
        case PO_SMOOTH_POST: return po::launch_kind<PO_SMOOTH_POST>(*a, st, lds);
    }
    return hipErrorInvalidValue;
}
