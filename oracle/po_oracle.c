/*
 * po_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE, NOT PRODUCT
 * (see po_oracle.h for who may use it and for the parity-pinning statement).
 *
 * Part 1  assembly of (P, A, l, u) in the reference's variable/row order
 *           KP  : /root/reference/src/solver/solver_kp_as_input.cpp:13-24,45-203
 *           KPC : /root/reference/src/solver/solver_kp_as_input_constrained.cpp:13-24,45-221
 *           K   : /root/reference/src/solver/solver_k_as_input.cpp:14-20,46-207
 * Part 2  OSQP-style ADMM (sparse quasi-definite LDL' of the KKT system), restated from the
 *         published algorithm; call site /root/reference/src/solver/solver.cpp:48-74
 * Part 3  output map getOptimizedPath()  (solver_kp_as_input.cpp:26-43, solver_k_as_input.cpp:22-44)
 * Part 4  driver mirroring OsqpSolver::solve (solver.cpp:46-77) + KKT certificate
 *
 * Build with -ffp-contract=off: the reference is built without FMA contraction (x86-64 baseline).
 */
#include "po_oracle.h"
#include "../include/po_pmath.h"

/* Portable-math mode for the map stages.  Off (default): glibc's sin / cos / atan2 / pow — what the reference's own binaries call, so the
 * oracle stays bit-comparable with them.  On: the portable routines of include/po_pmath.h, the same IEEE operation sequence the HIP kernels execute, so that
 * device and oracle agree BIT FOR BIT on values and indices (tests/test_pmath.py also bounds the difference between the two modes: <= 1 ulp per call). */
static int g_portable_math = 0;
void po_oracle_set_portable_math(int on) { g_portable_math = on != 0; }
int po_oracle_get_portable_math(void) { return g_portable_math; }
#define PO_NW_STAGNATION 8 /* full Newton steps on an unchanged factorisation that do not halve the dual residual, in a row, before the refinement gives a path up (see `nstag`) */
static int g_refine_trace = 0; /* developer aid: one stderr line per refinement block and round (tools/refine_trace.py) */
void po_oracle_set_refine_trace(int on) { g_refine_trace = on != 0; }
static long long g_ls_evals = 0, g_nw_steps = 0; /* developer aid (tools/newton_eval.py): line-search evaluations and Newton steps since the last read; not thread-safe */
long long po_oracle_ls_evals(long long *steps) { const long long e = g_ls_evals; if (steps) *steps = g_nw_steps; g_ls_evals = 0; g_nw_steps = 0; return e; }
double po_oracle_psin(double x) { return po_psin(x); }
double po_oracle_pcos(double x) { return po_pcos(x); }
double po_oracle_patan2(double y, double x) { return po_patan2(y, x); }
#define MSIN(x) (g_portable_math ? po_psin(x) : sin(x))
#define MCOS(x) (g_portable_math ? po_pcos(x) : cos(x))
#define MATAN2(y, x) (g_portable_math ? po_patan2((y), (x)) : atan2((y), (x)))
#define MPOW15(x) (g_portable_math ? po_ppow15(x) : pow((x), 1.5))

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif

#define OSQP_MIN_SCALING 1e-4
#define OSQP_MAX_SCALING 1e4
#define OSQP_RHO_MIN 1e-6
#define OSQP_RHO_MAX 1e6
#define OSQP_RHO_EQ_OVER_INEQ 1e3
#define OSQP_RHO_TOL 1e-4

/* ------------------------------------------------------------------------------------------ */
/* defaults                                                                                     */
/* ------------------------------------------------------------------------------------------ */
void po_oracle_default_params(po_params *p) {
    /* planning_flags.cpp:18-43 (car geometry), :8-14 (updateConfig) */
    const double car_length = 4.9, rear_axle_to_center = 1.45;
    memset(p, 0, sizeof(*p));
    p->d[0] = -3.0 / 8.0 * car_length + rear_axle_to_center;
    p->d[1] = -1.0 / 8.0 * car_length + rear_axle_to_center;
    p->d[2] = 1.0 / 8.0 * car_length + rear_axle_to_center;
    p->d[3] = 3.0 / 8.0 * car_length + rear_axle_to_center;
    p->w_curv = 10;       /* :108 */
    p->w_curv_rate = 200; /* :110 */
    p->w_dev = 0;         /* :112 */
    p->w_slack = 3;       /* :114 */
    p->k_w_curv = 50;     /* :102 */
    p->k_w_curv_rate = 200;
    p->k_w_dev = 0;
    p->w_k_slack = 500;
    p->w_kp_slack = 25000;
    p->margin = 1.3; /* :116 */
    p->max_steer = 30.0 * M_PI / 180.0;
    p->wheel_base = 2.85;
    p->enable_collision_check = 1;
    p->car_width = 2.0; p->car_length = 4.9; p->rear_axle_to_center = 1.45; p->safety_margin = 0.0;  /* planning_flags.cpp:18-29 */
    p->constraint_end_heading = 1;
    p->scaling = 10; /* OSQP default; > 0: true Ruiz passes, < 0: class-level form (what the device runs) */
    p->eps_abs = 1e-4;
    p->eps_rel = 1e-4;
    p->eps_prim_inf = 1e-4;
    p->eps_dual_inf = 1e-4;
    p->rho0 = 0.1;
    p->sigma = 1e-6;
    p->alpha = 1.6;
    p->adapt_tol = 5.0;
    p->max_iter = 4000;
    p->check_every = 25;
    p->adapt_every = 100;
    /* planning_flags.cpp:76-86 (reference-smoothing QPs) */
    p->t2_w_dev = 0.005; p->t2_w_curv = 1; p->t2_w_curv_rate = 10;
    p->cart_w_curv = 1; p->cart_w_curv_rate = 50; p->cart_w_dev = 0.0;
    /* planning_flags.cpp:41-43,57-63,137 */
    p->mu = 0.4; p->max_curvature_rate = 0.1; p->search_lateral_range = 10.0; p->search_long_spacing = 1.5; p->search_lat_spacing = 0.6;
    p->enable_dynamic_segmentation = 1;
    p->enable_raw_output = 1; p->output_spacing = 0.3; /* planning_flags.cpp:127-129 */
    p->polish = 0; p->polish_delta = 1e-6; p->polish_refine_iter = 3; /* OSQP defaults (polish off) */
    p->refine = 0; p->refine_eps = 1e-7; p->refine_rounds = 1; p->refine_chain = 2 /* device scheduling only */; p->refine_extra_rounds = 0;
    p->refine_newton_rho = 100.0; p->refine_newton_rho_eq = 1e4; p->refine_newton_rho_max = 1e5; p->refine_ls_tol = 0.6; p->refine_ls_max = 30; p->refine_newton_max = 300; p->refine_newton_final = 3; p->refine_newton_escalate = 12; p->refine_newton_rho_eq_max = 1e6; /* refine = 2 */
}

/* tools.hpp:24-35 — recursive in the reference; same fixed point as this loop */
double po_oracle_wrap_angle(double a) {
    for (;;) {
        if (a > M_PI) a -= 2 * M_PI;
        else if (a < -M_PI) a += 2 * M_PI;
        else return a;
    }
}

int po_oracle_keep(int form, const double *ref_s, int N) {
    if (!ref_s || N < 2) return PO_ERR_INVALID;
    if (form == PO_KPC) return 4; /* solver_kp_as_input_constrained.cpp:17 */
    if (form == PO_K) return 1;   /* no held control */
    if (form != PO_KP) return PO_ERR_INVALID;
    double interval = 0; /* solver.cpp:19,22-27 */
    for (int i = 1; i < N && i < 10; ++i) {
        double d = ref_s[i] - ref_s[i - 1];
        if (d > interval) interval = d;
    }
    double q = 1.2 / interval; /* solver_kp_as_input.cpp:17 */
    int k = (q >= 2147483647.0 || q != q) ? 1 : (int)q;
    return k > 1 ? k : 1;
}

int po_oracle_dims(int form, int N, int keep, int *n, int *m, int *C) {
    if (N < 2) return PO_ERR_INVALID;
    int c = 0, nn = 0, mm = 0;
    if (form == PO_KP) {
        if (keep < 1) return PO_ERR_INVALID;
        c = (N + keep - 2) / keep;
        nn = 3 * N + c + 2 * N;
        mm = 11 * N + c + 2;
    } else if (form == PO_KPC) {
        if (keep != 4) return PO_ERR_INVALID;
        c = (N + keep - 2) / keep;
        nn = 3 * N + c + 3 * N;
        mm = 12 * N + 3 * c + 2;
    } else if (form == PO_K) {
        c = N - 1;
        nn = 4 * N - 1;
        mm = 11 * N - 1;
    } else {
        return PO_ERR_INVALID;
    }
    if (n) *n = nn;
    if (m) *m = mm;
    if (C) *C = c;
    return PO_OK;
}

int po_oracle_nnz_bound_A(int form, int N, int keep) {
    (void)keep;
    (void)form;
    return 40 * N + 64; /* >= the dense writes of any formulation except K's NxN identity blocks, handled sparsely */
}
int po_oracle_nnz_bound_P(int form, int N, int keep) {
    (void)keep;
    (void)form;
    return 8 * N + 16;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 1: assembly.  The reference writes into a dense zero matrix and calls sparseView();     */
/* we record the same writes as (row, col, value, set|add) and compress, dropping exact zeros.  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int r, c, seq, add;
    double v;
} trip_t;
typedef struct {
    trip_t *t;
    int n, cap;
} tripbuf;

static int tb_push(tripbuf *b, int r, int c, double v, int add) {
    if (b->n == b->cap) {
        int nc = b->cap ? 2 * b->cap : 1024;
        trip_t *nt = (trip_t *)realloc(b->t, (size_t)nc * sizeof(trip_t));
        if (!nt) return -1;
        b->t = nt;
        b->cap = nc;
    }
    trip_t *e = &b->t[b->n];
    e->r = r;
    e->c = c;
    e->v = v;
    e->add = add;
    e->seq = b->n;
    b->n++;
    return 0;
}
#define TSET(b, r, c, v) tb_push((b), (int)(r), (int)(c), (v), 0)
#define TADD(b, r, c, v) tb_push((b), (int)(r), (int)(c), (v), 1)

static int trip_cmp(const void *a, const void *b) {
    const trip_t *x = (const trip_t *)a, *y = (const trip_t *)b;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

/* Compress to CSC; upper_only keeps r<=c (OsqpEigen hands OSQP the upper triangle of P). */
static int tb_compress(tripbuf *b, int ncols, int upper_only, int *Cp, int *Ci, double *Cx) {
    qsort(b->t, (size_t)b->n, sizeof(trip_t), trip_cmp);
    int nz = 0, k = 0;
    for (int c = 0; c < ncols; ++c) {
        Cp[c] = nz;
        while (k < b->n && b->t[k].c == c) {
            int r = b->t[k].r;
            double v = 0;
            while (k < b->n && b->t[k].c == c && b->t[k].r == r) {
                v = b->t[k].add ? v + b->t[k].v : b->t[k].v;
                ++k;
            }
            if (v != 0 && (!upper_only || r <= c)) {
                Ci[nz] = r;
                Cx[nz] = v;
                ++nz;
            }
        }
    }
    Cp[ncols] = nz;
    return nz;
}

#define BND(i, c, lu) bounds[((i) * 4 + (c)) * 2 + (lu)] /* lu: 0 = lb, 1 = ub */

/* End-heading window shared by the three formulations
 * (solver_kp_as_input.cpp:193-202, ..._constrained.cpp:211-220, solver_k_as_input.cpp:171-177).
 * Note the SIGNED test `end_psi < 70 deg` (no fabs) — preserved. */
static void end_heading_window(const po_params *p, double goal_z, double ref_z_last, double *lo, double *hi) {
    *lo = -PO_ORACLE_INFTY;
    *hi = PO_ORACLE_INFTY;
    if (p->constraint_end_heading) {
        double end_psi = po_oracle_wrap_angle(goal_z - ref_z_last);
        if (end_psi < 70 * M_PI / 180) {
            *lo = end_psi - 5 * M_PI / 180;
            *hi = end_psi + 5 * M_PI / 180;
        }
    }
}

static int assemble_kp_like(int form, const po_params *p, int N, int keep, const double *ref_k,
                            const double *ref_s, double ref_z_last, const double *bounds,
                            const double *x0, double goal_z, const double *max_k,
                            const double *max_kp, tripbuf *TP, tripbuf *TA, double *l, double *u) {
    int n, m, C;
    po_oracle_dims(form, N, keep, &n, &m, &C);
    const int state_size = 3 * N, control_size = C;
    const double INF = PO_ORACLE_INFTY;
    const double kmax = tan(p->max_steer) / p->wheel_base;
    /* ---- Hessian ---- */
    if (form == PO_KP) { /* solver_kp_as_input.cpp:45-63 */
        for (int i = 0; i < N; ++i) {
            TADD(TP, 3 * i, 3 * i, p->w_dev);
            TADD(TP, 3 * i + 2, 3 * i + 2, p->w_curv);
            TADD(TP, state_size + control_size + i, state_size + control_size + i, p->w_slack);
            TADD(TP, state_size + control_size + N + i, state_size + control_size + N + i, p->w_slack);
        }
        for (int i = 0; i < C; ++i) TADD(TP, state_size + i, state_size + i, keep * p->w_curv_rate);
    } else { /* solver_kp_as_input_constrained.cpp:45-66 */
        for (int i = 0; i < N; ++i) {
            TADD(TP, 3 * i, 3 * i, p->w_dev);
            TADD(TP, 3 * i + 2, 3 * i + 2, p->w_curv);
            TADD(TP, state_size + control_size + i, state_size + control_size + i, p->w_slack);
            TADD(TP, state_size + control_size + N + i, state_size + control_size + N + i, p->w_k_slack);
        }
        for (int i = 0; i < C; ++i) {
            TADD(TP, state_size + i, state_size + i, keep * p->w_curv_rate);
            TADD(TP, state_size + control_size + 2 * N + i, state_size + control_size + 2 * N + i,
                 p->w_kp_slack * keep);
        }
    }
    /* ---- transition part, identical in KP (:75-98) and KPC (:81-104) ---- */
    for (int i = 0; i < state_size; ++i) TSET(TA, i, i, -1.0);
    for (int i = 0; i < m; ++i) l[i] = u[i] = 0.0;
    l[0] = u[0] = -x0[0]; /* :143-147 */
    l[1] = u[1] = -x0[1];
    l[2] = u[2] = -x0[2];
    for (int i = 0; i < N - 1; ++i) {
        const double k = ref_k[i];
        const double ds = ref_s[i + 1] - ref_s[i];
        const double a10 = -pow(k, 2);
        /* A = a*ds + I with a = [[0,1,0],[a10,0,1],[0,0,0]] */
        const double Am[3][3] = {{0 * ds + 1, 1 * ds + 0, 0 * ds + 0},
                                 {a10 * ds + 0, 0 * ds + 1, 1 * ds + 0},
                                 {0 * ds + 0, 0 * ds + 0, 0 * ds + 1}};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) TSET(TA, 3 * (i + 1) + r, 3 * i + c, Am[r][c]);
        const int ci = i / keep;
        TSET(TA, 3 * (i + 1) + 0, state_size + ci, 0 * ds);
        TSET(TA, 3 * (i + 1) + 1, state_size + ci, 0 * ds);
        TSET(TA, 3 * (i + 1) + 2, state_size + ci, 1 * ds);
        /* c_list[i] = ds*(c - a*ref_state - b*ref_kp) = ds*(0,-k,0); bounds = -c_list (:148-151) */
        const double c1 = ds * (0.0 - k);
        l[3 * (i + 1) + 0] = u[3 * (i + 1) + 0] = 0.0;
        l[3 * (i + 1) + 1] = u[3 * (i + 1) + 1] = -c1;
        l[3 * (i + 1) + 2] = u[3 * (i + 1) + 2] = 0.0;
    }
    double elo, ehi;
    end_heading_window(p, goal_z, ref_z_last, &elo, &ehi);
    if (form == PO_KP) {
        const int vars = 3 * N;
        const int coll = vars + 2 * N + C;
        const int endb = coll + 6 * N;
        for (int i = 0; i < N; ++i) { /* :100-104 */
            TSET(TA, vars + i, 3 * i + 2, 1.0);
            TSET(TA, vars + N + C + i, state_size + control_size + i, 1.0);
        }
        for (int i = 0; i < C; ++i) TSET(TA, vars + N + i, state_size + i, 1.0); /* :105-107 */
        for (int i = 0; i < N; ++i) { /* :109-134 */
            TSET(TA, coll + 2 * i, 3 * i, 1.0);
            TSET(TA, coll + 2 * i, 3 * i + 1, p->d[0]);
            TSET(TA, coll + 2 * i + 1, 3 * i, 1.0);
            TSET(TA, coll + 2 * i + 1, 3 * i + 1, p->d[2]);
            TSET(TA, coll + 2 * N + i, 3 * i, 1.0);
            TSET(TA, coll + 2 * N + i, 3 * i + 1, p->d[3]);
            TSET(TA, coll + 2 * N + i, state_size + control_size + i, -1.0);
            TSET(TA, coll + 3 * N + i, 3 * i, 1.0);
            TSET(TA, coll + 3 * N + i, 3 * i + 1, p->d[3]);
            TSET(TA, coll + 3 * N + i, state_size + control_size + i, 1.0);
            TSET(TA, coll + 4 * N + i, 3 * i, 1.0);
            TSET(TA, coll + 4 * N + i, 3 * i + 1, p->d[1]);
            TSET(TA, coll + 4 * N + i, state_size + control_size + i, -1.0);
            TSET(TA, coll + 5 * N + i, 3 * i, 1.0);
            TSET(TA, coll + 5 * N + i, 3 * i + 1, p->d[1]);
            TSET(TA, coll + 5 * N + i, state_size + control_size + i, 1.0);
        }
        TSET(TA, endb, state_size - 3, 1.0); /* :136-137 */
        TSET(TA, endb + 1, state_size - 2, 1.0);
        for (int i = 0; i < N; ++i) { /* :153-159 */
            l[vars + i] = -kmax;
            u[vars + i] = kmax;
            l[vars + N + C + i] = 0;
            u[vars + N + C + i] = p->margin;
        }
        for (int i = 0; i < C; ++i) { /* :160-163 */
            l[vars + N + i] = -INF;
            u[vars + N + i] = INF;
        }
        for (int i = 0; i < N; ++i) { /* :165-188 */
            l[coll + 2 * i] = BND(i, 0, 0);
            u[coll + 2 * i] = BND(i, 0, 1);
            l[coll + 2 * i + 1] = BND(i, 2, 0);
            u[coll + 2 * i + 1] = BND(i, 2, 1);
            u[coll + 2 * N + i] = BND(i, 3, 1) - p->margin;
            l[coll + 2 * N + i] = -INF;
            l[coll + 3 * N + i] = BND(i, 3, 0) + p->margin;
            u[coll + 3 * N + i] = INF;
            u[coll + 4 * N + i] = BND(i, 1, 1) - p->margin;
            l[coll + 4 * N + i] = -INF;
            l[coll + 5 * N + i] = BND(i, 1, 0) + p->margin;
            u[coll + 5 * N + i] = INF;
        }
        l[endb] = -1; /* :191-192 */
        u[endb] = 1;
        l[endb + 1] = elo;
        u[endb + 1] = ehi;
    } else { /* KPC, solver_kp_as_input_constrained.cpp:68-221 */
        if (!max_k || !max_kp) return PO_ERR_INVALID;
        const int kl = 3 * N, ku = kl + N, kpl = ku + N, kpu = kpl + C, sb = kpu + C;
        const int coll = sb + 2 * N + C, endb = coll + 5 * N;
        const int s0 = state_size + control_size;
        for (int i = 0; i < N; ++i) { /* :110-117 */
            TSET(TA, kl + i, 3 * i + 2, 1.0);
            TSET(TA, kl + i, s0 + N + i, 1.0);
            TSET(TA, ku + i, 3 * i + 2, 1.0);
            TSET(TA, ku + i, s0 + N + i, -1.0);
            TSET(TA, sb + i, s0 + i, 1.0);
            TSET(TA, sb + N + i, s0 + N + i, 1.0);
        }
        for (int i = 0; i < C; ++i) { /* :119-125 */
            TSET(TA, kpl + i, state_size + i, 1.0);
            TSET(TA, kpl + i, s0 + 2 * N + i, 1.0);
            TSET(TA, kpu + i, state_size + i, 1.0);
            TSET(TA, kpu + i, s0 + 2 * N + i, -1.0);
            TSET(TA, sb + 2 * N + i, s0 + 2 * N + i, 1.0);
        }
        for (int i = 0; i < N; ++i) { /* :128-143 */
            TSET(TA, coll + 3 * i, 3 * i, 1.0);
            TSET(TA, coll + 3 * i, 3 * i + 1, p->d[0]);
            TSET(TA, coll + 3 * i + 1, 3 * i, 1.0);
            TSET(TA, coll + 3 * i + 1, 3 * i + 1, p->d[1]);
            TSET(TA, coll + 3 * i + 2, 3 * i, 1.0);
            TSET(TA, coll + 3 * i + 2, 3 * i + 1, p->d[3]);
            TSET(TA, coll + 3 * N + i, 3 * i, 1.0);
            TSET(TA, coll + 3 * N + i, 3 * i + 1, p->d[2]);
            TSET(TA, coll + 3 * N + i, s0 + i, -1.0);
            TSET(TA, coll + 4 * N + i, 3 * i, 1.0);
            TSET(TA, coll + 4 * N + i, 3 * i + 1, p->d[2]);
            TSET(TA, coll + 4 * N + i, s0 + i, 1.0);
        }
        TSET(TA, endb, state_size - 3, 1.0);
        TSET(TA, endb + 1, state_size - 2, 1.0);
        for (int i = 0; i < N; ++i) { /* :165-177 */
            l[kl + i] = -max_k[i];
            u[kl + i] = INF;
            l[ku + i] = -INF;
            u[ku + i] = max_k[i];
            l[sb + i] = 0;
            u[sb + i] = p->margin;
            l[sb + N + i] = 0;
            double r = kmax - max_k[i];
            u[sb + N + i] = r > 0.0 ? r : 0.0;
        }
        for (int i = 0; i < C; ++i) { /* :178-187: max_kp_list is per POINT but indexed by control id */
            l[kpl + i] = -max_kp[i];
            u[kpl + i] = INF;
            l[kpu + i] = -INF;
            u[kpu + i] = max_kp[i];
            l[sb + 2 * N + i] = 0;
            u[sb + 2 * N + i] = INF;
        }
        for (int i = 0; i < N; ++i) { /* :190-205 */
            l[coll + 3 * i] = BND(i, 0, 0);
            u[coll + 3 * i] = BND(i, 0, 1);
            l[coll + 3 * i + 1] = BND(i, 1, 0);
            u[coll + 3 * i + 1] = BND(i, 1, 1);
            l[coll + 3 * i + 2] = BND(i, 3, 0);
            u[coll + 3 * i + 2] = BND(i, 3, 1);
            u[coll + 3 * N + i] = BND(i, 2, 1) - p->margin;
            l[coll + 3 * N + i] = -INF;
            l[coll + 4 * N + i] = BND(i, 2, 0) + p->margin;
            u[coll + 4 * N + i] = INF;
        }
        l[endb] = -INF; /* :209-210: end e_y is free in KPC */
        u[endb] = INF;
        l[endb + 1] = elo;
        u[endb + 1] = ehi;
    }
    return PO_OK;
}

static int assemble_k(const po_params *p, int N, const double *ref_k, const double *ref_s,
                      double ref_z_last, const double *bounds, const double *x0, double goal_z,
                      tripbuf *TP, tripbuf *TA, double *l, double *u) {
    const int m = 11 * N - 1;
    const int control_size = N - 1;
    const double INF = PO_ORACLE_INFTY;
    const double w_c = p->k_w_curv, w_cr = p->k_w_curv_rate, w_pq = p->k_w_dev, w_e = p->w_slack;
    /* Hessian, solver_k_as_input.cpp:46-87 */
    for (int i = 0; i < N; ++i) {
        TSET(TP, 2 * i, 2 * i, 0.0);
        TSET(TP, 2 * i + 1, 2 * i + 1, w_pq);
    }
    for (int i = 0; i < control_size; ++i) { /* matrix_R :62-76, only the tri-diagonal is ever inserted */
        if (i == 0 || i == control_size - 1) TSET(TP, 2 * N + i, 2 * N + i, w_c + w_cr);
        else TSET(TP, 2 * N + i, 2 * N + i, w_cr * 2 + w_c);
        if (i + 1 < control_size) {
            TSET(TP, 2 * N + i, 2 * N + i + 1, -w_cr);
            TSET(TP, 2 * N + i + 1, 2 * N + i, -w_cr);
        }
    }
    for (int i = 0; i < N; ++i) TSET(TP, 3 * N - 1 + i, 3 * N - 1 + i, 1.0 * w_e);
    /* constraints, :105-150 */
    for (int i = 0; i < 2 * N; ++i) TSET(TA, i, i, -1.0);
    for (int i = 0; i < m; ++i) l[i] = u[i] = 0.0;
    const double L = p->wheel_base;
    for (int i = 0; i < N - 1; ++i) { /* setDynamicMatrix :89-103 */
        const double k = ref_k[i];
        const double rs = ref_s[i + 1] - ref_s[i];
        const double delta = atan(k * L);
        TSET(TA, 2 * (i + 1) + 0, 2 * i + 0, 1.0);
        TSET(TA, 2 * (i + 1) + 0, 2 * i + 1, -rs * pow(k, 2));
        TSET(TA, 2 * (i + 1) + 1, 2 * i + 0, rs);
        TSET(TA, 2 * (i + 1) + 1, 2 * i + 1, 1.0);
        TSET(TA, 2 * (i + 1) + 0, 2 * N + i, rs / L / pow(cos(delta), 2));
        TSET(TA, 2 * (i + 1) + 1, 2 * N + i, 0.0);
    }
    for (int i = 0; i < 4 * N - 1; ++i) TSET(TA, 2 * N + i, i, 1.0);
    for (int i = 0; i < N; ++i) {
        TSET(TA, 6 * N - 1 + 3 * i + 0, 2 * i, p->d[0]);
        TSET(TA, 6 * N - 1 + 3 * i + 0, 2 * i + 1, 1.0);
        TSET(TA, 6 * N - 1 + 3 * i + 1, 2 * i, p->d[2]);
        TSET(TA, 6 * N - 1 + 3 * i + 1, 2 * i + 1, 1.0);
        TSET(TA, 6 * N - 1 + 3 * i + 2, 2 * i, p->d[3]);
        TSET(TA, 6 * N - 1 + 3 * i + 2, 2 * i + 1, 1.0);
        TSET(TA, 9 * N - 1 + i, 2 * i, p->d[1]);
        TSET(TA, 9 * N - 1 + i, 2 * i + 1, 1.0);
        TSET(TA, 10 * N - 1 + i, 2 * i, p->d[1]);
        TSET(TA, 10 * N - 1 + i, 2 * i + 1, 1.0);
        TSET(TA, 9 * N - 1 + i, 3 * N - 1 + i, -1.0);
        TSET(TA, 10 * N - 1 + i, 3 * N - 1 + i, 1.0);
    }
    /* bounds, :152-206 */
    l[0] = u[0] = -x0[1]; /* x0 << init_error[1], init_error[0] */
    l[1] = u[1] = -x0[0];
    for (int i = 0; i < N - 1; ++i) {
        const double ds = ref_s[i + 1] - ref_s[i];
        const double steer = atan(ref_k[i] * L);
        const double c0 = ds * steer / L / pow(cos(steer), 2);
        l[2 + 2 * i] = u[2 + 2 * i] = c0;
        l[2 + 2 * i + 1] = u[2 + 2 * i + 1] = 0;
    }
    for (int i = 0; i < 2 * N; ++i) {
        l[2 * N + i] = -INF;
        u[2 * N + i] = INF;
    }
    double elo, ehi;
    end_heading_window(p, goal_z, ref_z_last, &elo, &ehi);
    if (elo > -INF) { /* only overwritten when the window applies (:171-177) */
        l[2 * N + 2 * N - 2] = elo;
        u[2 * N + 2 * N - 2] = ehi;
    }
    for (int i = 0; i < N - 1; ++i) {
        l[4 * N + i] = -p->max_steer;
        u[4 * N + i] = p->max_steer;
    }
    for (int i = 0; i < N; ++i) {
        l[5 * N - 1 + i] = 0;
        u[5 * N - 1 + i] = p->margin;
    }
    for (int i = 0; i < N; ++i) {
        l[6 * N - 1 + 3 * i + 0] = BND(i, 0, 0);
        u[6 * N - 1 + 3 * i + 0] = BND(i, 0, 1);
        l[6 * N - 1 + 3 * i + 1] = BND(i, 2, 0);
        u[6 * N - 1 + 3 * i + 1] = BND(i, 2, 1);
        l[6 * N - 1 + 3 * i + 2] = BND(i, 3, 0);
        u[6 * N - 1 + 3 * i + 2] = BND(i, 3, 1);
        u[10 * N - 1 + i] = INF;
        l[9 * N - 1 + i] = -INF;
        u[9 * N - 1 + i] = BND(i, 1, 1) - p->margin;
        l[10 * N - 1 + i] = BND(i, 1, 0) + p->margin;
    }
    return PO_OK;
}

int po_oracle_assemble(int form, const po_params *p, int N, int keep, const double *ref_k,
                       const double *ref_s, const double *ref_z_last, const double *bounds,
                       const double *x0, double goal_z, const double *max_k, const double *max_kp,
                       int *Pp, int *Pi, double *Px, int *Ap, int *Ai, double *Ax, double *l,
                       double *u) {
    int n, m, C;
    int rc = po_oracle_dims(form, N, keep, &n, &m, &C);
    if (rc) return rc;
    if (!p || !ref_k || !ref_s || !ref_z_last || !bounds || !x0) return PO_ERR_INVALID;
    tripbuf TP = {0, 0, 0}, TA = {0, 0, 0};
    if (form == PO_K) rc = assemble_k(p, N, ref_k, ref_s, *ref_z_last, bounds, x0, goal_z, &TP, &TA, l, u);
    else rc = assemble_kp_like(form, p, N, keep, ref_k, ref_s, *ref_z_last, bounds, x0, goal_z, max_k, max_kp, &TP, &TA, l, u);
    if (rc == PO_OK) {
        tb_compress(&TP, n, 1, Pp, Pi, Px);
        tb_compress(&TA, n, 0, Ap, Ai, Ax);
    }
    free(TP.t);
    free(TA.t);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 2: OSQP-style ADMM                                                                      */
/* ------------------------------------------------------------------------------------------ */
static double vnorm_inf(const double *v, int n) {
    double r = 0;
    for (int i = 0; i < n; ++i) {
        double a = fabs(v[i]);
        if (a > r) r = a;
    }
    return r;
}
static double vnorm_inf_scaled(const double *s, const double *v, int n) {
    double r = 0;
    for (int i = 0; i < n; ++i) {
        double a = fabs(s[i] * v[i]);
        if (a > r) r = a;
    }
    return r;
}
/* y = A x (CSC) */
static void csc_mv(int ncol, int nrow, const int *Ap, const int *Ai, const double *Ax, const double *x, double *y) {
    for (int i = 0; i < nrow; ++i) y[i] = 0;
    for (int c = 0; c < ncol; ++c) {
        double xc = x[c];
        for (int k = Ap[c]; k < Ap[c + 1]; ++k) y[Ai[k]] += Ax[k] * xc;
    }
}
/* y = A' x */
static void csc_mtv(int ncol, const int *Ap, const int *Ai, const double *Ax, const double *x, double *y) {
    for (int c = 0; c < ncol; ++c) {
        double s = 0;
        for (int k = Ap[c]; k < Ap[c + 1]; ++k) s += Ax[k] * x[Ai[k]];
        y[c] = s;
    }
}
/* y = P x, P symmetric stored upper */
static void sym_mv(int n, const int *Pp, const int *Pi, const double *Px, const double *x, double *y) {
    for (int i = 0; i < n; ++i) y[i] = 0;
    for (int c = 0; c < n; ++c) {
        for (int k = Pp[c]; k < Pp[c + 1]; ++k) {
            int r = Pi[k];
            y[r] += Px[k] * x[c];
            if (r != c) y[c] += Px[k] * x[r];
        }
    }
}

static double limit_scaling(double v) {
    v = v < OSQP_MIN_SCALING ? 1.0 : v;
    v = v > OSQP_MAX_SCALING ? OSQP_MAX_SCALING : v;
    return v;
}

/* --- sparse LDL' of a symmetric (quasi-definite) matrix given as upper-triangular CSC ---
 * Elimination-tree based up-looking factorisation (the classic algorithm used by LDL/QDLDL). */
typedef struct {
    int n;
    int *parent, *Lp, *Li, *Lnz, *flag, *pattern;
    double *Lx, *D, *Dinv, *Y;
} ldl_t;

static void ldl_free(ldl_t *f) {
    free(f->parent);
    free(f->Lp);
    free(f->Li);
    free(f->Lnz);
    free(f->flag);
    free(f->pattern);
    free(f->Lx);
    free(f->D);
    free(f->Dinv);
    free(f->Y);
    memset(f, 0, sizeof(*f));
}

static int ldl_symbolic(ldl_t *f, int n, const int *Kp, const int *Ki) {
    memset(f, 0, sizeof(*f));
    f->n = n;
    f->parent = (int *)malloc(sizeof(int) * (size_t)n);
    f->Lp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    f->Lnz = (int *)malloc(sizeof(int) * (size_t)n);
    f->flag = (int *)malloc(sizeof(int) * (size_t)n);
    f->pattern = (int *)malloc(sizeof(int) * (size_t)n);
    f->D = (double *)malloc(sizeof(double) * (size_t)n);
    f->Dinv = (double *)malloc(sizeof(double) * (size_t)n);
    f->Y = (double *)malloc(sizeof(double) * (size_t)n);
    if (!f->parent || !f->Lp || !f->Lnz || !f->flag || !f->pattern || !f->D || !f->Dinv || !f->Y) return -1;
    for (int k = 0; k < n; ++k) {
        f->parent[k] = -1;
        f->flag[k] = k;
        f->Lnz[k] = 0;
        for (int p = Kp[k]; p < Kp[k + 1]; ++p) {
            int i = Ki[p];
            if (i > k) return -2; /* not upper triangular */
            for (; f->flag[i] != k; i = f->parent[i]) {
                if (f->parent[i] == -1) f->parent[i] = k;
                f->Lnz[i]++;
                f->flag[i] = k;
            }
        }
    }
    f->Lp[0] = 0;
    for (int k = 0; k < n; ++k) f->Lp[k + 1] = f->Lp[k] + f->Lnz[k];
    int lnz = f->Lp[n] > 0 ? f->Lp[n] : 1;
    f->Li = (int *)malloc(sizeof(int) * (size_t)lnz);
    f->Lx = (double *)malloc(sizeof(double) * (size_t)lnz);
    if (!f->Li || !f->Lx) return -1;
    return 0;
}

static int ldl_numeric(ldl_t *f, const int *Kp, const int *Ki, const double *Kx) {
    const int n = f->n;
    for (int k = 0; k < n; ++k) {
        f->Y[k] = 0;
        f->Lnz[k] = 0;
    }
    for (int k = 0; k < n; ++k) {
        int top = n;
        f->flag[k] = k;
        for (int p = Kp[k]; p < Kp[k + 1]; ++p) {
            int i = Ki[p];
            f->Y[i] += Kx[p];
            int len = 0;
            for (; f->flag[i] != k; i = f->parent[i]) {
                f->pattern[len++] = i;
                f->flag[i] = k;
            }
            while (len > 0) f->pattern[--top] = f->pattern[--len];
        }
        f->D[k] = f->Y[k];
        f->Y[k] = 0;
        for (; top < n; ++top) {
            int i = f->pattern[top];
            double yi = f->Y[i];
            f->Y[i] = 0;
            int p2 = f->Lp[i] + f->Lnz[i];
            for (int p = f->Lp[i]; p < p2; ++p) f->Y[f->Li[p]] -= f->Lx[p] * yi;
            double lki = yi * f->Dinv[i];
            f->D[k] -= lki * yi;
            f->Li[p2] = k;
            f->Lx[p2] = lki;
            f->Lnz[i]++;
        }
        if (f->D[k] == 0.0) return -1;
        f->Dinv[k] = 1.0 / f->D[k];
    }
    return 0;
}

static void ldl_solve(const ldl_t *f, double *x) {
    const int n = f->n;
    for (int j = 0; j < n; ++j) {
        double xj = x[j];
        for (int p = f->Lp[j]; p < f->Lp[j + 1]; ++p) x[f->Li[p]] -= f->Lx[p] * xj;
    }
    for (int j = 0; j < n; ++j) x[j] *= f->Dinv[j];
    for (int j = n - 1; j >= 0; --j) {
        double xj = x[j];
        for (int p = f->Lp[j]; p < f->Lp[j + 1]; ++p) xj -= f->Lx[p] * x[f->Li[p]];
        x[j] = xj;
    }
}

/* Greedy minimum-degree ordering on the pattern of a symmetric matrix (upper CSC).
 * Stand-in for OSQP's AMD: only the fill (speed of the CPU baseline) depends on it. */
static int min_degree_order(int n, const int *Kp, const int *Ki, int *perm) {
    int **adj = (int **)calloc((size_t)n, sizeof(int *));
    int *deg = (int *)calloc((size_t)n, sizeof(int));
    int *cap = (int *)calloc((size_t)n, sizeof(int));
    int *mark = (int *)malloc(sizeof(int) * (size_t)n);
    char *done = (char *)calloc((size_t)n, 1);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)n);
    if (!adj || !deg || !cap || !mark || !done || !tmp) return -1;
    for (int c = 0; c < n; ++c)
        for (int p = Kp[c]; p < Kp[c + 1]; ++p)
            if (Ki[p] != c) {
                deg[c]++;
                deg[Ki[p]]++;
            }
    for (int i = 0; i < n; ++i) {
        cap[i] = deg[i] + 8;
        adj[i] = (int *)malloc(sizeof(int) * (size_t)cap[i]);
        deg[i] = 0;
        mark[i] = -1;
    }
    for (int c = 0; c < n; ++c)
        for (int p = Kp[c]; p < Kp[c + 1]; ++p) {
            int r = Ki[p];
            if (r != c) {
                adj[c][deg[c]++] = r;
                adj[r][deg[r]++] = c;
            }
        }
    int stamp = 0;
    for (int step = 0; step < n; ++step) {
        int best = -1, bd = 1 << 30;
        for (int i = 0; i < n; ++i)
            if (!done[i] && deg[i] < bd) {
                bd = deg[i];
                best = i;
            }
        const int v = best;
        perm[step] = v;
        done[v] = 1;
        const int nv = deg[v];
        for (int a = 0; a < nv; ++a) {
            const int x = adj[v][a];
            /* new adj(x) = (adj(x) U adj(v)) \ {x, v}, deduplicated with a stamp */
            int cnt = 0;
            ++stamp;
            mark[x] = stamp;
            for (int q = 0; q < deg[x]; ++q) {
                int w = adj[x][q];
                if (w != v && !done[w] && mark[w] != stamp) {
                    mark[w] = stamp;
                    tmp[cnt++] = w;
                }
            }
            for (int b = 0; b < nv; ++b) {
                int w = adj[v][b];
                if (mark[w] != stamp) {
                    mark[w] = stamp;
                    tmp[cnt++] = w;
                }
            }
            if (cnt > cap[x]) {
                cap[x] = cnt + 8;
                free(adj[x]);
                adj[x] = (int *)malloc(sizeof(int) * (size_t)cap[x]);
            }
            memcpy(adj[x], tmp, sizeof(int) * (size_t)cnt);
            deg[x] = cnt;
        }
    }
    for (int i = 0; i < n; ++i) free(adj[i]);
    free(adj);
    free(deg);
    free(cap);
    free(mark);
    free(done);
    free(tmp);
    return 0;
}

/* Build the upper-triangular CSC pattern+values of K = [[P+sigma I, A'],[A, -diag(1/rho)]] under a
 * symmetric permutation (perm: new -> old).  rho_pos[j] = position of constraint j's diagonal. */
typedef struct {
    int nk, *Kp, *Ki, *rho_pos;
    double *Kx;
} kkt_t;

static int kkt_build(kkt_t *K, int n, int m, const int *Pp, const int *Pi, const double *Px,
                     const int *Ap, const int *Ai, const double *Ax, double sigma,
                     const double *rho_inv, const int *perm) {
    const int nk = n + m;
    int *pinv = (int *)malloc(sizeof(int) * (size_t)nk);
    tripbuf T = {0, 0, 0};
    if (!pinv) return -1;
    for (int i = 0; i < nk; ++i) pinv[perm ? perm[i] : i] = i;
#define KADD(i, j, v)                                    \
    do {                                                 \
        int a_ = pinv[i], b_ = pinv[j];                  \
        if (a_ <= b_) tb_push(&T, a_, b_, (v), 1);       \
        else tb_push(&T, b_, a_, (v), 1);                \
    } while (0)
    for (int c = 0; c < n; ++c) {
        int has_diag = 0;
        for (int k = Pp[c]; k < Pp[c + 1]; ++k) {
            if (Pi[k] == c) {
                KADD(c, c, Px[k] + sigma);
                has_diag = 1;
            } else {
                KADD(Pi[k], c, Px[k]);
            }
        }
        if (!has_diag) KADD(c, c, sigma);
        for (int k = Ap[c]; k < Ap[c + 1]; ++k) KADD(c, n + Ai[k], Ax[k]);
    }
    for (int j = 0; j < m; ++j) KADD(n + j, n + j, -rho_inv[j]);
#undef KADD
    K->nk = nk;
    K->Kp = (int *)malloc(sizeof(int) * (size_t)(nk + 1));
    K->Ki = (int *)malloc(sizeof(int) * (size_t)(T.n + 1));
    K->Kx = (double *)malloc(sizeof(double) * (size_t)(T.n + 1));
    K->rho_pos = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    if (!K->Kp || !K->Ki || !K->Kx || !K->rho_pos) return -1;
    /* like tb_compress but keeps explicit structure even if a value is 0 (pattern must be stable) */
    qsort(T.t, (size_t)T.n, sizeof(trip_t), trip_cmp);
    int nz = 0, k = 0;
    for (int c = 0; c < nk; ++c) {
        K->Kp[c] = nz;
        while (k < T.n && T.t[k].c == c) {
            int r = T.t[k].r;
            double v = 0;
            while (k < T.n && T.t[k].c == c && T.t[k].r == r) v += T.t[k++].v;
            K->Ki[nz] = r;
            K->Kx[nz] = v;
            ++nz;
        }
    }
    K->Kp[nk] = nz;
    for (int j = 0; j < m; ++j) {
        int c = pinv[n + j];
        K->rho_pos[j] = K->Kp[c + 1] - 1; /* diagonal is the last entry of an upper-triangular column */
    }
    free(T.t);
    free(pinv);
    return 0;
}
static void kkt_free(kkt_t *K) {
    free(K->Kp);
    free(K->Ki);
    free(K->Kx);
    free(K->rho_pos);
    memset(K, 0, sizeof(*K));
}

int po_oracle_qp_solve(int n, int m, const int *Pp0, const int *Pi0, const double *Px0,
                       const double *q0, const int *Ap0, const int *Ai0, const double *Ax0,
                       const double *l0, const double *u0, const po_params *prm, const int *perm_in,
                       double *x, double *y, double *z, po_info *info) {
    return po_oracle_qp_solve_ext(n, m, Pp0, Pi0, Px0, q0, Ap0, Ai0, Ax0, l0, u0, prm, perm_in, NULL, NULL, 1.0, x, y, z, info);
}

/* Same, with an externally supplied equilibration (Dext per variable, Eext per row, cost scale cext)
 * instead of the Ruiz passes when Dext != NULL. */
int po_oracle_qp_solve_ext(int n, int m, const int *Pp0, const int *Pi0, const double *Px0,
                           const double *q0, const int *Ap0, const int *Ai0, const double *Ax0,
                           const double *l0, const double *u0, const po_params *prm, const int *perm_in,
                           const double *Dext, const double *Eext, double cext,
                           double *x, double *y, double *z, po_info *info) {
    if (n <= 0 || m < 0 || !prm || !x || !y || !z || !info) return PO_ERR_INVALID;
    const int pnz = Pp0[n], anz = Ap0[n];
    int rc = PO_OK;
    double *entry = NULL; /* the point a round BELOW eps starts from (x, z, y): what the path returns if that round runs out of iterations on a worse one */
    /* working (scaled) copies */
    double *Px = (double *)malloc(sizeof(double) * (size_t)(pnz + 1));
    double *Ax = (double *)malloc(sizeof(double) * (size_t)(anz + 1));
    double *q = (double *)calloc((size_t)n, sizeof(double));
    double *l = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *u = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *D = (double *)malloc(sizeof(double) * (size_t)n), *Dinv = (double *)malloc(sizeof(double) * (size_t)n);
    double *E = (double *)malloc(sizeof(double) * (size_t)(m + 1)), *Einv = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *tn = (double *)malloc(sizeof(double) * (size_t)n), *tn2 = (double *)malloc(sizeof(double) * (size_t)n);
    double *tm = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *rho_vec = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *rho_inv = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    int *ctype = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    double *x_prev = (double *)calloc((size_t)n, sizeof(double));
    double *z_prev = (double *)calloc((size_t)(m + 1), sizeof(double));
    double *dx = (double *)calloc((size_t)n, sizeof(double));
    double *dy = (double *)calloc((size_t)(m + 1), sizeof(double));
    double *xz = (double *)malloc(sizeof(double) * (size_t)(n + m));
    double *rhs = (double *)malloc(sizeof(double) * (size_t)(n + m));
    double *Axv = (double *)calloc((size_t)(m + 1), sizeof(double));
    double *Pxv = (double *)calloc((size_t)n, sizeof(double));
    double *Aty = (double *)calloc((size_t)n, sizeof(double));
    int *perm = (int *)malloc(sizeof(int) * (size_t)(n + m));
    int *pinv = (int *)malloc(sizeof(int) * (size_t)(n + m));
    kkt_t K;
    ldl_t F;
    memset(&K, 0, sizeof(K));
    memset(&F, 0, sizeof(F));
    memcpy(Px, Px0, sizeof(double) * (size_t)pnz);
    memcpy(Ax, Ax0, sizeof(double) * (size_t)anz);
    if (q0) memcpy(q, q0, sizeof(double) * (size_t)n);
    memcpy(l, l0, sizeof(double) * (size_t)m);
    memcpy(u, u0, sizeof(double) * (size_t)m);
    double cscale = 1.0;
    for (int i = 0; i < n; ++i) D[i] = Dinv[i] = 1.0;
    for (int i = 0; i < m; ++i) E[i] = Einv[i] = 1.0;

    /* ---- Ruiz equilibration (OSQP scale_data); off (0 passes) on the device-matching path ---- */
    const int ruiz_passes = Dext ? 0 : (prm->scaling > 0 ? prm->scaling : 0);
    if (Dext) { /* external diagonal equilibration */
        for (int c = 0; c < n; ++c) {
            for (int k = Pp0[c]; k < Pp0[c + 1]; ++k) Px[k] *= Dext[c] * Dext[Pi0[k]] * cext;
            for (int k = Ap0[c]; k < Ap0[c + 1]; ++k) Ax[k] *= Dext[c] * Eext[Ai0[k]];
            q[c] *= Dext[c] * cext;
            D[c] = Dext[c];
        }
        for (int i = 0; i < m; ++i) E[i] = Eext[i];
        cscale = cext;
    }
    for (int pass = 0; pass < ruiz_passes; ++pass) {
        for (int i = 0; i < n; ++i) tn[i] = 0;
        for (int i = 0; i < m; ++i) tm[i] = 0;
        for (int c = 0; c < n; ++c) { /* column inf-norms of [P; A], rows of A */
            for (int k = Pp0[c]; k < Pp0[c + 1]; ++k) {
                double a = fabs(Px[k]);
                int r = Pi0[k];
                if (a > tn[c]) tn[c] = a;
                if (a > tn[r]) tn[r] = a;
            }
            for (int k = Ap0[c]; k < Ap0[c + 1]; ++k) {
                double a = fabs(Ax[k]);
                if (a > tn[c]) tn[c] = a;
                if (a > tm[Ai0[k]]) tm[Ai0[k]] = a;
            }
        }
        for (int i = 0; i < n; ++i) tn[i] = 1.0 / sqrt(limit_scaling(tn[i]));
        for (int i = 0; i < m; ++i) tm[i] = 1.0 / sqrt(limit_scaling(tm[i]));
        for (int c = 0; c < n; ++c) {
            for (int k = Pp0[c]; k < Pp0[c + 1]; ++k) Px[k] *= tn[c] * tn[Pi0[k]];
            for (int k = Ap0[c]; k < Ap0[c + 1]; ++k) Ax[k] *= tn[c] * tm[Ai0[k]];
            q[c] *= tn[c];
            D[c] *= tn[c];
        }
        for (int i = 0; i < m; ++i) E[i] *= tm[i];
        /* cost scaling */
        for (int i = 0; i < n; ++i) tn2[i] = 0;
        for (int c = 0; c < n; ++c)
            for (int k = Pp0[c]; k < Pp0[c + 1]; ++k) {
                double a = fabs(Px[k]);
                int r = Pi0[k];
                if (a > tn2[c]) tn2[c] = a;
                if (a > tn2[r]) tn2[r] = a;
            }
        double ct = 0;
        for (int i = 0; i < n; ++i) ct += tn2[i];
        ct /= n;
        double qn = limit_scaling(vnorm_inf(q, n));
        ct = ct > qn ? ct : qn;
        ct = 1.0 / limit_scaling(ct);
        for (int k = 0; k < pnz; ++k) Px[k] *= ct;
        for (int i = 0; i < n; ++i) q[i] *= ct;
        cscale *= ct;
    }
    if (ruiz_passes > 0 || Dext) {
        for (int i = 0; i < n; ++i) Dinv[i] = 1.0 / D[i];
        for (int i = 0; i < m; ++i) {
            Einv[i] = 1.0 / E[i];
            l[i] *= E[i];
            u[i] *= E[i];
        }
    }
    const double cinv = 1.0 / cscale;

    /* ---- data validation (OSQP validate_data: l <= u, else osqp_setup fails and the reference's
     * initSolver() returns false, solver.cpp:72) ---- */
    for (int i = 0; i < m; ++i)
        if (l[i] > u[i]) {
            memset(info, 0, sizeof(*info));
            info->status = PO_STATUS_PRIMAL_INFEASIBLE;
            info->rho = prm->rho0;
            for (int j = 0; j < n; ++j) x[j] = 0;
            for (int j = 0; j < m; ++j) y[j] = z[j] = 0;
            goto done;
        }

    /* ---- rho vector (OSQP set_rho_vec) ---- */
    double rho = prm->rho0;
    rho = rho < OSQP_RHO_MIN ? OSQP_RHO_MIN : (rho > OSQP_RHO_MAX ? OSQP_RHO_MAX : rho);
    for (int i = 0; i < m; ++i) {
        if (l[i] < -PO_ORACLE_INFTY * OSQP_MIN_SCALING && u[i] > PO_ORACLE_INFTY * OSQP_MIN_SCALING) {
            ctype[i] = -1;
            rho_vec[i] = OSQP_RHO_MIN;
        } else if (u[i] - l[i] < OSQP_RHO_TOL) {
            ctype[i] = 1;
            rho_vec[i] = OSQP_RHO_EQ_OVER_INEQ * rho;
        } else {
            ctype[i] = 0;
            rho_vec[i] = rho;
        }
        rho_inv[i] = 1.0 / rho_vec[i];
    }

    /* ---- KKT + ordering + factorisation ---- */
    if (perm_in) {
        memcpy(perm, perm_in, sizeof(int) * (size_t)(n + m));
    } else {
        kkt_t K0;
        memset(&K0, 0, sizeof(K0));
        if (kkt_build(&K0, n, m, Pp0, Pi0, Px, Ap0, Ai0, Ax, prm->sigma, rho_inv, NULL)) { rc = PO_ERR_NOMEM; goto done; }
        if (min_degree_order(K0.nk, K0.Kp, K0.Ki, perm)) { kkt_free(&K0); rc = PO_ERR_NOMEM; goto done; }
        kkt_free(&K0);
    }
    for (int i = 0; i < n + m; ++i) pinv[perm[i]] = i;
    if (kkt_build(&K, n, m, Pp0, Pi0, Px, Ap0, Ai0, Ax, prm->sigma, rho_inv, perm)) { rc = PO_ERR_NOMEM; goto done; }
    if (ldl_symbolic(&F, K.nk, K.Kp, K.Ki)) { rc = PO_ERR_NOMEM; goto done; }
    if (ldl_numeric(&F, K.Kp, K.Ki, K.Kx)) { rc = PO_ERR_INVALID; goto done; }

    /* ---- ADMM (OSQP osqp_solve), cold start ---- */
    for (int i = 0; i < n; ++i) x[i] = 0;
    for (int i = 0; i < m; ++i) y[i] = z[i] = 0;
    memset(info, 0, sizeof(*info));
    info->status = PO_STATUS_UNSOLVED;
    const double alpha = prm->alpha, sigma = prm->sigma;
    int iter = 0, n_refactor = 0, checked_this_iter = 0;
    double pri_res = 0, dua_res = 0;
    /* po_params.refine_rounds = R > 1: the type-based iteration first stops at 10^(R-1) x eps and hands over to the refinement; a path that one does not
     * certify comes back here (resume_main) at a 10 x tighter eps, down to eps itself */
    const int rounds = prm->refine ? (prm->refine_rounds > 1 ? prm->refine_rounds : 1) : 1;
    /* refine_extra_rounds = E: a path the LAST regular round does not certify goes on below eps — type-based iteration at eps / 10, refinement, eps / 100, ...
     * up to E more rounds (each with the full refinement budget) */
    const int rounds_total = rounds + (prm->refine && prm->refine_extra_rounds > 0 ? prm->refine_extra_rounds : 0);
    int round = 0, refine_its = 0, refine_fac = 0, exhausted = 0;
    double entry_res[3] = {0, 0, 0}; /* (r_prim, r_dual, rho of `entry`) */
    double eps_mul = 1.0;
    for (int r_ = 1; r_ < rounds; ++r_) eps_mul *= 10.0;
resume_main:
    for (iter = iter + 1; iter <= prm->max_iter; ++iter) {
        memcpy(x_prev, x, sizeof(double) * (size_t)n);
        memcpy(z_prev, z, sizeof(double) * (size_t)m);
        /* update_xz_tilde */
        for (int i = 0; i < n; ++i) rhs[pinv[i]] = sigma * x_prev[i] - q[i];
        for (int i = 0; i < m; ++i) rhs[pinv[n + i]] = z_prev[i] - rho_inv[i] * y[i];
        ldl_solve(&F, rhs);
        for (int i = 0; i < n; ++i) xz[i] = rhs[pinv[i]];
        for (int i = 0; i < m; ++i) xz[n + i] = z_prev[i] + rho_inv[i] * (rhs[pinv[n + i]] - y[i]);
        /* update_x, update_z, update_y */
        for (int i = 0; i < n; ++i) {
            x[i] = alpha * xz[i] + (1.0 - alpha) * x_prev[i];
            dx[i] = x[i] - x_prev[i];
        }
        for (int i = 0; i < m; ++i) {
            double v = alpha * xz[n + i] + (1.0 - alpha) * z_prev[i] + rho_inv[i] * y[i];
            z[i] = v < l[i] ? l[i] : (v > u[i] ? u[i] : v);
        }
        for (int i = 0; i < m; ++i) {
            dy[i] = rho_vec[i] * (alpha * xz[n + i] + (1.0 - alpha) * z_prev[i] - z[i]);
            y[i] += dy[i];
        }
        checked_this_iter = 0;
        const int can_check = prm->check_every > 0 && (iter % prm->check_every == 0);
        const int can_adapt = prm->adapt_every > 0 && (iter % prm->adapt_every == 0);
        double pri_norm_s = 0, dua_norm_s = 0, pri_res_s = 0, dua_res_s = 0;
        if (can_check || can_adapt || iter == prm->max_iter) {
            /* update_info: residuals (unscaled for termination, scaled for the rho estimate) */
            csc_mv(n, m, Ap0, Ai0, Ax, x, Axv);
            sym_mv(n, Pp0, Pi0, Px, x, Pxv);
            csc_mtv(n, Ap0, Ai0, Ax, y, Aty);
            for (int i = 0; i < m; ++i) tm[i] = Axv[i] - z[i];
            for (int i = 0; i < n; ++i) tn[i] = Pxv[i] + q[i] + Aty[i];
            pri_res_s = vnorm_inf(tm, m);
            dua_res_s = vnorm_inf(tn, n);
            pri_res = vnorm_inf_scaled(Einv, tm, m);
            dua_res = cinv * vnorm_inf_scaled(Dinv, tn, n);
            double nz_s = vnorm_inf(z, m), nAx_s = vnorm_inf(Axv, m);
            pri_norm_s = nz_s > nAx_s ? nz_s : nAx_s;
            double nq_s = vnorm_inf(q, n), nAty_s = vnorm_inf(Aty, n), nPx_s = vnorm_inf(Pxv, n);
            dua_norm_s = nq_s > nAty_s ? nq_s : nAty_s;
            dua_norm_s = dua_norm_s > nPx_s ? dua_norm_s : nPx_s;
            checked_this_iter = 1;
        }
        if (can_check || iter == prm->max_iter) {
            double nz = vnorm_inf_scaled(Einv, z, m), nAx = vnorm_inf_scaled(Einv, Axv, m);
            const double eps_prim_u = prm->eps_abs + prm->eps_rel * (nz > nAx ? nz : nAx); /* at the caller's eps (eps_mul = 1) */
            double eps_prim = eps_mul * eps_prim_u;
            double nq = vnorm_inf_scaled(Dinv, q, n), nAty = vnorm_inf_scaled(Dinv, Aty, n), nPx = vnorm_inf_scaled(Dinv, Pxv, n);
            double dn = nq > nAty ? nq : nAty;
            dn = dn > nPx ? dn : nPx;
            const double eps_dual_u = prm->eps_abs + prm->eps_rel * cinv * dn;
            double eps_dual = eps_mul * eps_dual_u;
            int prim_ok = pri_res < eps_prim, dual_ok = dua_res < eps_dual; /* strict, as OSQP */
            int prim_inf = 0, dual_inf = 0;
            if (!prim_ok) { /* is_primal_infeasible */
                for (int i = 0; i < m; ++i) {
                    double d = dy[i];
                    if (u[i] > PO_ORACLE_INFTY * OSQP_MIN_SCALING) {
                        if (l[i] < -PO_ORACLE_INFTY * OSQP_MIN_SCALING) d = 0;
                        else d = d < 0 ? d : 0;
                    } else if (l[i] < -PO_ORACLE_INFTY * OSQP_MIN_SCALING) {
                        d = d > 0 ? d : 0;
                    }
                    tm[i] = d;
                }
                double ndy = vnorm_inf_scaled(E, tm, m);
                if (ndy > prm->eps_prim_inf) {
                    double lhs = 0;
                    for (int i = 0; i < m; ++i) lhs += u[i] * (tm[i] > 0 ? tm[i] : 0) + l[i] * (tm[i] < 0 ? tm[i] : 0);
                    if (lhs < -prm->eps_prim_inf * ndy) {
                        csc_mtv(n, Ap0, Ai0, Ax, tm, tn);
                        prim_inf = vnorm_inf_scaled(Dinv, tn, n) < prm->eps_prim_inf * ndy;
                    }
                }
            }
            if (!dual_ok && !prim_inf) { /* is_dual_infeasible */
                double ndx = vnorm_inf_scaled(D, dx, n);
                if (ndx > prm->eps_dual_inf) {
                    double qdx = 0;
                    for (int i = 0; i < n; ++i) qdx += q[i] * dx[i];
                    if (qdx < -cscale * prm->eps_dual_inf * ndx) {
                        sym_mv(n, Pp0, Pi0, Px, dx, tn);
                        if (vnorm_inf_scaled(Dinv, tn, n) < cscale * prm->eps_dual_inf * ndx) {
                            csc_mv(n, m, Ap0, Ai0, Ax, dx, tm);
                            dual_inf = 1;
                            for (int i = 0; i < m; ++i) {
                                double a = Einv[i] * tm[i];
                                if ((u[i] < PO_ORACLE_INFTY * OSQP_MIN_SCALING && a > prm->eps_dual_inf * ndx) ||
                                    (l[i] > -PO_ORACLE_INFTY * OSQP_MIN_SCALING && a < -prm->eps_dual_inf * ndx)) {
                                    dual_inf = 0;
                                    break;
                                }
                            }
                        }
                    }
                }
            }
            if (prim_ok && dual_ok) { info->status = PO_STATUS_SOLVED; break; }
            if (prim_inf) { info->status = PO_STATUS_PRIMAL_INFEASIBLE; break; }
            if (dual_inf) { info->status = PO_STATUS_DUAL_INFEASIBLE; break; }
            /* a round BELOW eps (refine_extra_rounds) that runs out of iterations: the path met the caller's eps in the last regular round — if the iterate it ends
             * on still does, it is solved (not certified: status_refine stays -1), and no further refinement is attempted */
            if (iter == prm->max_iter && round >= rounds && pri_res < eps_prim_u && dua_res < eps_dual_u) { info->status = PO_STATUS_SOLVED; exhausted = 1; break; }
        }
        if (can_adapt) { /* compute_rho_estimate + adapt_rho (scaled-space quantities) */
            double pr = pri_res_s / (pri_norm_s + 1e-10);
            double dr = dua_res_s / (dua_norm_s + 1e-10);
            double rho_new = rho * sqrt(pr / (dr + 1e-10));
            rho_new = rho_new < OSQP_RHO_MIN ? OSQP_RHO_MIN : (rho_new > OSQP_RHO_MAX ? OSQP_RHO_MAX : rho_new);
            if (rho_new > rho * prm->adapt_tol || rho_new < rho / prm->adapt_tol) {
                rho = rho_new;
                for (int i = 0; i < m; ++i) {
                    if (ctype[i] == 0) rho_vec[i] = rho;
                    else if (ctype[i] == 1) rho_vec[i] = OSQP_RHO_EQ_OVER_INEQ * rho;
                    rho_inv[i] = 1.0 / rho_vec[i];
                    K.Kx[K.rho_pos[i]] = -rho_inv[i];
                }
                if (ldl_numeric(&F, K.Kp, K.Ki, K.Kx)) { rc = PO_ERR_INVALID; goto done; }
                ++n_refactor;
            }
        }
    }
    if (iter > prm->max_iter) {
        iter = prm->max_iter;
        if (info->status == PO_STATUS_UNSOLVED) {
            info->status = PO_STATUS_MAX_ITER;
            if (round >= rounds && entry) { /* a round below eps never unsolves a path: back to the point it started from (solved at eps in the round before), not certified */
                memcpy(x, entry, sizeof(double) * (size_t)n);
                memcpy(z, entry + n, sizeof(double) * (size_t)m);
                memcpy(y, entry + n + m, sizeof(double) * (size_t)m);
                pri_res = entry_res[0]; dua_res = entry_res[1]; rho = entry_res[2];
                info->status = PO_STATUS_SOLVED;
                exhausted = 1;
            }
        }
    }
    (void)checked_this_iter;
    info->iters = iter + refine_its;          /* (refinement iterations / refactorisations of earlier rounds: po_params.refine_rounds) */
    info->n_refactor = n_refactor + refine_fac;
    info->r_prim = pri_res;
    info->r_dual = dua_res;
    info->rho = rho;
    /* ---- refinement (po_params.refine = 2; extension, not OSQP): semismooth Newton on the augmented Lagrangian from the point the ADMM iteration stopped at (see po_hip.h;
     * the activity-weighted ADMM continuation refine = 1 of rounds 2 - 3 was removed in round 5) ---- */
    if (prm->refine == 2 && info->status == PO_STATUS_SOLVED && !exhausted) {
        int nfac = 0, it2 = 0, stop = 0;
        double *snap = (double *)malloc(sizeof(double) * (size_t)(n + 2 * m)); /* the solved point: kept if the phase does not end at least as well */
        const double pri0 = pri_res, dua0 = dua_res;
        memcpy(snap, x, sizeof(double) * (size_t)n);
        memcpy(snap + n, z, sizeof(double) * (size_t)m);
        memcpy(snap + n + m, y, sizeof(double) * (size_t)m);
        {
            /* ---- semismooth Newton on the augmented Lagrangian with an exact line search (po_hip.h).  State: x and w_i = a_i x + y_i / rho_i
             * (one number per row, like the engine's v); implied z = clip(w), y = rho (w - z).  rho_i: rb_in on inequality rows, rb_eq on equality rows (both grow on a stall, see below). ---- */
            double rn_ = prm->refine_newton_rho;
            rn_ = rn_ < OSQP_RHO_MIN ? OSQP_RHO_MIN : (rn_ > OSQP_RHO_MAX ? OSQP_RHO_MAX : rn_);
            /* equality rows: a FIXED penalty (not 1e3 x the inequality one): the gradient carries rho_eq x (a.x - b), a difference of O(1) numbers — at 1e6 and more
             * its rounding alone (1e-16 x 1e6 x the unscaling) sits above the dual tolerance */
            double rb_in = rn_, pri_outer = -1.0;
            double rb_eq = prm->refine_newton_rho_eq > 0 ? (prm->refine_newton_rho_eq < 1e8 ? prm->refine_newton_rho_eq : 1e8) : 1e4;
            /* the caps, defaulted and bounded exactly as the engine's make_dev_params does (csrc/po_capi.cpp): 0 in refine_newton_rho_eq_max is the documented "never grows" */
            const double in_cap = prm->refine_newton_rho_max > 0 ? (prm->refine_newton_rho_max < OSQP_RHO_MAX ? prm->refine_newton_rho_max : OSQP_RHO_MAX) : 1e5;
            const double eq_cap = prm->refine_newton_rho_eq_max > 0 ? (prm->refine_newton_rho_eq_max < 1e8 ? prm->refine_newton_rho_eq_max : 1e8) : (prm->refine_newton_rho_eq_max < 0 ? 1e6 : 0.0); /* (below rb_eq: never grows) */
            const int cap_nw = prm->refine_newton_max > 0 ? prm->refine_newton_max : 300;
            const int ls_max = prm->refine_ls_max > 0 ? prm->refine_ls_max : 30;
            double *w = (double *)malloc(sizeof(double) * (size_t)(m + 1)), *sv = (double *)malloc(sizeof(double) * (size_t)(m + 1));
            double *dv = (double *)malloc(sizeof(double) * (size_t)n), *Pd = (double *)malloc(sizeof(double) * (size_t)n);
            int first_fac = 1, nouter = 0, fail = 0, certified = 0, nfinal = 0;
            /* STAGNATION (round 5): `quiet` = the last step took the full step on an unchanged factorisation — the situation in which Newton's method converges quadratically; a dual
             * residual that does not even halve over PO_NW_STAGNATION such steps in a row sits on a floor the method cannot get under (the rounding of rho_eq (a.x - b) under extreme
             * weights, or a flat valley damped by the proximal terms): the ATTEMPT ends, uncertified, like one that ran out of steps — and the rounds go on as they do then (the
             * type-based iteration at a tighter eps, another attempt).  Without it such an attempt burns its whole budget (refine_newton_max = 300 steps) in every round: one such
             * path held its batch for ~80 ms. */
            int nstag = 0, quiet = 0;
            double rd_prev = -1.0;
            csc_mv(n, m, Ap0, Ai0, Ax, x, Axv);
            for (int i = 0; i < m; ++i) w[i] = Axv[i] + (ctype[i] == 0 ? y[i] / rb_in : (ctype[i] == 1 ? y[i] / rb_eq : 0.0));
            for (;;) {
                /* the point (x, z = clip(w), y = rho (w - z)) and OSQP's test on it at refine_eps; the dual residual IS the gradient of the merit */
                csc_mv(n, m, Ap0, Ai0, Ax, x, Axv);
                for (int i = 0; i < m; ++i) {
                    z[i] = w[i] < l[i] ? l[i] : (w[i] > u[i] ? u[i] : w[i]);
                    y[i] = (ctype[i] == 0 ? rb_in : (ctype[i] == 1 ? rb_eq : 0.0)) * (w[i] - z[i]);
                }
                sym_mv(n, Pp0, Pi0, Px, x, Pxv);
                csc_mtv(n, Ap0, Ai0, Ax, y, Aty);
                for (int i = 0; i < m; ++i) tm[i] = Axv[i] - z[i];
                for (int i = 0; i < n; ++i) tn[i] = Pxv[i] + q[i] + Aty[i];
                pri_res = vnorm_inf_scaled(Einv, tm, m);
                dua_res = cinv * vnorm_inf_scaled(Dinv, tn, n);
                const double nz = vnorm_inf_scaled(Einv, z, m), nAx = vnorm_inf_scaled(Einv, Axv, m);
                const double nAty = vnorm_inf_scaled(Dinv, Aty, n), nPx = vnorm_inf_scaled(Dinv, Pxv, n), nq = vnorm_inf_scaled(Dinv, q, n);
                double dn = nq > nAty ? nq : nAty;
                dn = dn > nPx ? dn : nPx;
                const double tol_d = prm->refine_eps + prm->refine_eps * cinv * dn;
                const int dual_ok = dua_res < tol_d;
                stop = dual_ok && pri_res < prm->refine_eps + prm->refine_eps * (nz > nAx ? nz : nAx);
                if (g_refine_trace) fprintf(stderr, "  newton round %d step %d outer %d nfac %d  r_prim %.3e r_dual %.3e%s\n", round, it2, nouter, nfac, pri_res, dua_res, stop ? "  CERTIFIED" : "");
                /* refine_newton_final: from the certified point Newton steps go on (tight line search; quadratic convergence once the active set is right) until the
                 * dual residual — the gradient of the merit — sits 100 x below its tolerance, at most that many steps */
                /* (round 5) the test is re-evaluated on EVERY point, the ones the correction steps produce included: `stop` is only ever true for a point that passes
                 * it, so status_refine = 1 always describes the point that is returned.  A correction step that leaves the test's region (a step that changes the
                 * active set can land with a larger dual residual) takes the path back to the regular iteration — multiplier updates and all — until it is certified again */
                if (quiet && rd_prev >= 0.0 && dua_res > 0.5 * rd_prev) ++nstag; else nstag = 0;
                rd_prev = dua_res;
                if (stop) {
                    if (prm->refine_newton_final <= 0 || dua_res < 1e-3 * tol_d || nfinal >= prm->refine_newton_final) break;
                    certified = 1;
                } else {
                    certified = 0;
                    if (it2 >= cap_nw) break;
                    if (nstag >= PO_NW_STAGNATION) break;
                }
                if (!certified && dual_ok) { /* the inner problem is solved: multiplier update, w <- A x + (w - clip(w)) */
                    if (++nouter > 50) break;
                    /* a multiplier update that did not cut the primal residual by 4: the penalty grows 10 x (the multipliers stay, w is re-expressed) — the inequality
                     * rows' up to refine_newton_rho_max, then the equality rows' up to refine_newton_rho_eq_max */
                    double ratio = 1.0, ratio_eq = 1.0;
                    /* a long stall (degenerate optima: the multipliers are not unique and the method of multipliers converges sublinearly at any fixed penalty): from the
                     * refine_newton_escalate-th update on both caps stand 10 x higher, from twice that on 100 x */
                    const int esc_n = prm->refine_newton_escalate;
                    const double esc = esc_n > 0 && nouter >= esc_n ? (nouter >= 2 * esc_n ? 100.0 : 10.0) : 1.0;
                    if (pri_outer >= 0.0 && pri_res > 0.25 * pri_outer) {
                        /* (the escalated caps stay inside OSQP's RHO_MAX for the inequality rows, 1e8 for the equality rows) */
                        if (rb_in * 10.0 <= (in_cap * esc < OSQP_RHO_MAX ? in_cap * esc : OSQP_RHO_MAX)) { ratio = 0.1; rb_in *= 10.0; first_fac = 1; }
                        else if (rb_eq * 10.0 <= (eq_cap * esc < 1e8 ? eq_cap * esc : 1e8)) { ratio_eq = 0.1; rb_eq *= 10.0; first_fac = 1; }
                    }
                    pri_outer = pri_res;
                    for (int i = 0; i < m; ++i) w[i] = Axv[i] + (ctype[i] == 1 ? ratio_eq : ratio) * (w[i] - z[i]);
                    quiet = 0;
                    continue;
                }
                /* Newton step: rows outside their bounds at rho_i, the others at RHO_MIN (the matrix of refine = 1) */
                int changed = first_fac;
                for (int i = 0; i < m; ++i) {
                    /* outside by more than rounding noise (a slack that sits on its bound comes out as +-1e-19: whether such a row counts as active must not
                     * depend on the last bit — it changes the Newton matrix, not the gradient) */
                    const double r = ctype[i] == -1 ? OSQP_RHO_MIN : (ctype[i] == 1 ? rb_eq : (fabs(w[i] - z[i]) > 1e-15 * (1.0 + fabs(z[i])) ? rb_in : OSQP_RHO_MIN));
                    if (r != rho_vec[i]) { rho_vec[i] = r; changed = 1; }
                }
                quiet = !changed;
                if (changed) {
                    for (int i = 0; i < m; ++i) { rho_inv[i] = 1.0 / rho_vec[i]; K.Kx[K.rho_pos[i]] = -rho_inv[i]; }
                    if (ldl_numeric(&F, K.Kp, K.Ki, K.Kx)) { free(w); free(sv); free(dv); free(Pd); free(snap); rc = PO_ERR_INVALID; goto done; }
                    ++nfac;
                    first_fac = 0;
                }
                for (int i = 0; i < n; ++i) rhs[pinv[i]] = -tn[i];
                for (int i = 0; i < m; ++i) rhs[pinv[n + i]] = 0.0;
                ldl_solve(&F, rhs);
                for (int i = 0; i < n; ++i) dv[i] = rhs[pinv[i]];
                csc_mv(n, m, Ap0, Ai0, Ax, dv, sv);
                sym_mv(n, Pp0, Pi0, Px, dv, Pd);
                /* psi'(t) = c0 + c1 t + sum over inequality rows of rb (w + t s - clip(w + t s)) s: the equality rows are linear in t */
                double c0 = 0, c1 = 0;
                for (int i = 0; i < n; ++i) { c0 += dv[i] * (Pxv[i] + q[i]); c1 += dv[i] * Pd[i]; }
                for (int i = 0; i < m; ++i)
                    if (ctype[i] == 1) { c0 += rb_eq * (w[i] - z[i]) * sv[i]; c1 += rb_eq * sv[i] * sv[i]; }
                double t = 1.0, lo = 0.0, hi = -1.0, f0 = 0.0;
                for (int ev = -1; ev < ls_max; ++ev) { /* ev = -1: psi'(0) */
                    const double tt = ev < 0 ? 0.0 : t;
                    double f = c0 + c1 * tt, fp = c1;
                    ++g_ls_evals;
                    for (int i = 0; i < m; ++i)
                        if (ctype[i] == 0) {
                            const double ww = w[i] + tt * sv[i];
                            if (ww < l[i]) { f += rb_in * (ww - l[i]) * sv[i]; fp += rb_in * sv[i] * sv[i]; }
                            else if (ww > u[i]) { f += rb_in * (ww - u[i]) * sv[i]; fp += rb_in * sv[i] * sv[i]; }
                        }
                    if (ev < 0) { f0 = f; if (!(f0 < 0.0)) { fail = 1; break; } continue; }
                    if (fabs(f) <= (certified && prm->refine_ls_tol > 1e-4 ? 1e-4 : prm->refine_ls_tol) * fabs(f0)) break; /* (the final correction step: always the tight search) */
                    if (f < 0) lo = t; else hi = t;
                    double tnx = fp > 0 ? t - f / fp : -1.0;
                    if (!(tnx > lo && (hi < 0 || tnx < hi))) tnx = hi < 0 ? 2.0 * t : 0.5 * (lo + hi);
                    t = tnx;
                }
                if (g_refine_trace) {
                    int na = 0, nb = 0;
                    for (int i = 0; i < m; ++i) { if (ctype[i] == 0) { if (rho_vec[i] == rb_in) ++na; else ++nb; } }
                    double sx = 0, sd = 0;
                    for (int i = 0; i < n; ++i) { sx += (D[i] * x[i]) * (D[i] * x[i]); sd += (D[i] * dv[i]) * (D[i] * dv[i]); }
                    fprintf(stderr, "      nact0 %d ninact %d |x|^2 %.9e |d|^2(all) %.9e\n", na, nb, sx, sd);
                    fprintf(stderr, "      c0 %.6e c1 %.6e f0 %.6e t %.9f\n", c0 / cscale, c1 / cscale, f0 / cscale, t);
                }
                if (fail) { stop = certified; break; } /* not a descent direction (rounding at the bottom of the merit): the attempt ends here — certified if it was */
                ++g_nw_steps;
                for (int i = 0; i < n; ++i) x[i] += t * dv[i];
                for (int i = 0; i < m; ++i) w[i] += t * sv[i];
                ++it2;
                if (certified) ++nfinal;
                quiet = quiet && t == 1.0;
            }
            free(w); free(sv); free(dv); free(Pd);
        }
        if (g_refine_trace) fprintf(stderr, "round %d: type-based iterations so far %d, refinement %d its %d refactorisations -> %s (r_prim %.3e r_dual %.3e; entered at %.3e %.3e)\n", round, iter, it2, nfac, stop ? "certified" : "not certified", pri_res, dua_res, pri0, dua0);
        refine_its += it2;
        refine_fac += nfac;
        info->iters = iter + refine_its;
        info->n_refactor = n_refactor + refine_fac;
        info->status_refine = stop ? 1 : -1; /* certified at refine_eps / out of budget (po_hip.h) */
        if (stop || (pri_res <= pri0 && dua_res <= dua0)) {
            info->r_prim = pri_res;
            info->r_dual = dua_res;
        } else { /* out of iterations and not better than the solved point: keep that one */
            memcpy(x, snap, sizeof(double) * (size_t)n);
            memcpy(z, snap + n, sizeof(double) * (size_t)m);
            memcpy(y, snap + n + m, sizeof(double) * (size_t)m);
            pri_res = pri0;
            dua_res = dua0;
        }
        free(snap);
        if (!stop && round + 1 < rounds_total) { /* not certified at refine_eps: back to the type-based iteration (its own rho) at a 10 x tighter eps, then again */
            ++round;
            if (round >= rounds) { /* a round below eps: remember where it starts (this point meets the caller's eps) */
                if (!entry) entry = (double *)malloc(sizeof(double) * (size_t)(n + 2 * m));
                memcpy(entry, x, sizeof(double) * (size_t)n);
                memcpy(entry + n, z, sizeof(double) * (size_t)m);
                memcpy(entry + n + m, y, sizeof(double) * (size_t)m);
                entry_res[0] = pri_res; entry_res[1] = dua_res; entry_res[2] = rho;
            }
            eps_mul *= 0.1;
            for (int i = 0; i < m; ++i) {
                rho_vec[i] = ctype[i] == -1 ? OSQP_RHO_MIN : (ctype[i] == 1 ? OSQP_RHO_EQ_OVER_INEQ * rho : rho);
                rho_inv[i] = 1.0 / rho_vec[i];
                K.Kx[K.rho_pos[i]] = -rho_inv[i];
            }
            if (ldl_numeric(&F, K.Kp, K.Ki, K.Kx)) { rc = PO_ERR_INVALID; goto done; }
            ++n_refactor;
            info->status = PO_STATUS_UNSOLVED;
            goto resume_main;
        }
    }
    /* ---- polish (OSQP polish.c, on the scaled problem like OSQP): reduced KKT system on the active set with the regularisation
     * +-delta, polish_refine_iter steps of iterative refinement, normal-cone projection, OSQP's acceptance rule. ---- */
    if (prm->polish && info->status == PO_STATUS_SOLVED) {
        const double delta = prm->polish_delta;
        const int passes = 1;
        int *act = (int *)malloc(sizeof(int) * (size_t)(m + 1)), *act_new = (int *)malloc(sizeof(int) * (size_t)(m + 1));
        int *ridx = (int *)malloc(sizeof(int) * (size_t)(m + 1));
        int *Rp = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *Ri = (int *)malloc(sizeof(int) * (size_t)(anz + 1));
        double *Rx = (double *)malloc(sizeof(double) * (size_t)(anz + 1));
        double *px = (double *)malloc(sizeof(double) * (size_t)n), *py = (double *)calloc((size_t)(m + 1), sizeof(double)), *pz = (double *)malloc(sizeof(double) * (size_t)(m + 1));
        double *sol = (double *)malloc(sizeof(double) * (size_t)(n + m)), *res = (double *)malloc(sizeof(double) * (size_t)(n + m));
        double *bred = (double *)malloc(sizeof(double) * (size_t)(m + 1)), *dlt = (double *)malloc(sizeof(double) * (size_t)(m + 1));
        int *prm2 = (int *)malloc(sizeof(int) * (size_t)(n + m)), *pin2 = (int *)malloc(sizeof(int) * (size_t)(n + m));
        int adopted = 0, self_consistent = 0;
        double pol_pri = 0, pol_dua = 0;
        /* form_Ared: 1 = active at the lower bound, 2 = at the upper bound */
        for (int i = 0; i < m; ++i) act[i] = (z[i] - l[i] < -y[i]) ? 1 : ((u[i] - z[i] < y[i]) ? 2 : 0);
        for (int pass = 0; pass < passes; ++pass) {
            int mred = 0;
            for (int i = 0; i < m; ++i) {
                ridx[i] = act[i] ? mred : -1;
                if (act[i]) { bred[mred] = act[i] == 1 ? l[i] : u[i]; dlt[mred] = delta; ++mred; }
            }
            int rz = 0;
            for (int c = 0; c < n; ++c) {
                Rp[c] = rz;
                for (int k = Ap0[c]; k < Ap0[c + 1]; ++k)
                    if (ridx[Ai0[k]] >= 0) { Ri[rz] = ridx[Ai0[k]]; Rx[rz] = Ax[k]; ++rz; }
            }
            Rp[n] = rz;
            kkt_t K0, K2;
            ldl_t F2;
            memset(&K0, 0, sizeof(K0)); memset(&K2, 0, sizeof(K2)); memset(&F2, 0, sizeof(F2));
            const int nk2 = n + mred;
            int bad = kkt_build(&K0, n, mred, Pp0, Pi0, Px, Rp, Ri, Rx, delta, dlt, NULL) || min_degree_order(K0.nk, K0.Kp, K0.Ki, prm2);
            kkt_free(&K0);
            if (!bad) {
                for (int i = 0; i < nk2; ++i) pin2[prm2[i]] = i;
                bad = kkt_build(&K2, n, mred, Pp0, Pi0, Px, Rp, Ri, Rx, delta, dlt, prm2) || ldl_symbolic(&F2, K2.nk, K2.Kp, K2.Ki) || ldl_numeric(&F2, K2.Kp, K2.Ki, K2.Kx);
            }
            if (bad) { ldl_free(&F2); kkt_free(&K2); break; }
            /* (K + dK) sol = [-q; b], then sol += (K + dK)^-1 ([-q; b] - K sol), polish_refine_iter times */
            for (int i = 0; i < n; ++i) res[pin2[i]] = -q[i];
            for (int i = 0; i < mred; ++i) res[pin2[n + i]] = bred[i];
            ldl_solve(&F2, res);
            for (int i = 0; i < nk2; ++i) sol[i] = res[pin2[i]];
            for (int it = 0; it < prm->polish_refine_iter; ++it) {
                sym_mv(n, Pp0, Pi0, Px, sol, tn);            /* P x */
                csc_mtv(n, Rp, Ri, Rx, sol + n, tn2);        /* Ared' y */
                for (int i = 0; i < mred; ++i) tm[i] = 0;
                csc_mv(n, mred, Rp, Ri, Rx, sol, tm);        /* Ared x */
                for (int i = 0; i < n; ++i) res[pin2[i]] = -q[i] - tn[i] - tn2[i];
                for (int i = 0; i < mred; ++i) res[pin2[n + i]] = bred[i] - tm[i];
                ldl_solve(&F2, res);
                for (int i = 0; i < nk2; ++i) sol[i] += res[pin2[i]];
            }
            ldl_free(&F2); kkt_free(&K2);
            for (int i = 0; i < n; ++i) px[i] = sol[i];
            for (int i = 0; i < m; ++i) py[i] = ridx[i] >= 0 ? sol[n + ridx[i]] : 0.0;
            for (int i = 0; i < m; ++i) pz[i] = 0;
            csc_mv(n, m, Ap0, Ai0, Ax, px, pz);              /* z = A x */
            /* the active set this point implies */
            int changes = 0;
            for (int i = 0; i < m; ++i) {
                act_new[i] = (pz[i] - l[i] < -py[i]) ? 1 : ((u[i] - pz[i] < py[i]) ? 2 : 0);
                if (u[i] - l[i] < OSQP_RHO_TOL && act_new[i]) act_new[i] = act[i] ? act[i] : act_new[i]; /* equalities: either label is the same row */
                changes += act_new[i] != act[i];
            }
            /* project_normalcone + residuals of the polished point (unscaled, like update_info with scaled_termination = 0) */
            memcpy(Axv, pz, sizeof(double) * (size_t)m);
            for (int i = 0; i < m; ++i) {
                const double t = pz[i] + py[i];
                pz[i] = t < l[i] ? l[i] : (t > u[i] ? u[i] : t);
                py[i] = t - pz[i];
                tm[i] = Axv[i] - pz[i];
            }
            sym_mv(n, Pp0, Pi0, Px, px, Pxv);
            csc_mtv(n, Ap0, Ai0, Ax, py, Aty);
            for (int i = 0; i < n; ++i) tn[i] = Pxv[i] + q[i] + Aty[i];
            pol_pri = vnorm_inf_scaled(Einv, tm, m);
            pol_dua = cinv * vnorm_inf_scaled(Dinv, tn, n);
            if (changes == 0) { self_consistent = 1; break; }
            if (pass + 1 < passes) memcpy(act, act_new, sizeof(int) * (size_t)m);
        }
        /* polish_successful (polish.c) — or, with the extension, a self-consistent active set whose point is feasible to round-off */
        adopted = (pol_pri < pri_res && pol_dua < dua_res) || (pol_pri < pri_res && dua_res < 1e-10) || (pol_dua < dua_res && pri_res < 1e-10);
        (void)self_consistent;
        if (adopted) {
            memcpy(x, px, sizeof(double) * (size_t)n);
            memcpy(y, py, sizeof(double) * (size_t)m);
            memcpy(z, pz, sizeof(double) * (size_t)m);
            info->r_prim = pol_pri;
            info->r_dual = pol_dua;
            info->status_polish = 1;
        } else {
            info->status_polish = -1;
        }
        free(act); free(act_new); free(ridx); free(Rp); free(Ri); free(Rx); free(px); free(py); free(pz); free(sol); free(res); free(bred); free(dlt);
        free(prm2); free(pin2);
    }
    /* unscale solution */
    for (int i = 0; i < n; ++i) x[i] *= D[i];
    for (int i = 0; i < m; ++i) {
        y[i] *= cinv * E[i];
        z[i] *= Einv[i];
    }
    {
        sym_mv(n, Pp0, Pi0, Px0, x, tn);
        double o = 0;
        for (int i = 0; i < n; ++i) o += 0.5 * x[i] * tn[i] + (q0 ? q0[i] * x[i] : 0.0);
        info->obj = o;
    }
done:
    free(entry);
    ldl_free(&F);
    kkt_free(&K);
    free(Px); free(Ax); free(q); free(l); free(u); free(D); free(Dinv); free(E); free(Einv);
    free(tn); free(tn2); free(tm); free(rho_vec); free(rho_inv); free(ctype); free(x_prev);
    free(z_prev); free(dx); free(dy); free(xz); free(rhs); free(Axv); free(Pxv); free(Aty);
    free(perm); free(pinv);
    return rc;
}


/* ------------------------------------------------------------------------------------------ */
/* Class-level equilibration ("structured Ruiz").                                               */
/* OSQP's Ruiz passes on these QPs return the SAME factor for every variable / row of one kind   */
/* (the pattern repeats along the path and the data-dependent entries never attain a column's    */
/* inf-norm), so the same iteration is run once on a one-stage template with the path's nominal */
/* arc-length step.  The device engine implements exactly this; mode scaling<0.   */
/* variable classes: 0 e_y, 1 e_phi, 2 c (k or delta), 3 s1, 4 s2, 5 u, 6 su (K: end delta), 7 dead */
/* ------------------------------------------------------------------------------------------ */
#define PO_NVC 8
#define PO_NRC 24
typedef struct {
    int nr;
    double Pmax[PO_NVC], cnt[PO_NVC];
    double a[PO_NRC][PO_NVC];
    int tgt[PO_NRC];
} class_model_t;

static void class_model_build(class_model_t *M, int form, const po_params *p, int N, int keep, double ds) {
    int n, m, C;
    po_oracle_dims(form, N, keep, &n, &m, &C);
    memset(M, 0, sizeof(*M));
    for (int r = 0; r < PO_NRC; ++r) M->tgt[r] = -1;
    const double d1 = p->d[0], d2 = p->d[1], d3 = p->d[2], d4 = p->d[3];
    int r = 0;
#define ROW2(dv) do { M->a[r][0] = 1; M->a[r][1] = (dv); ++r; } while (0)
#define ROW2S(dv, sg) do { M->a[r][0] = 1; M->a[r][1] = (dv); M->a[r][3] = (sg); ++r; } while (0)
    if (form == PO_KP) {
        M->a[r][2] = 1; ++r;            /* k box   */
        M->a[r][3] = 1; ++r;            /* S box   */
        ROW2(d1); ROW2(d3);             /* hard    */
        ROW2S(d4, -1); ROW2S(d4, 1); ROW2S(d2, -1); ROW2S(d2, 1);
        /* dyn rows */
        M->a[r][0] = 1; M->a[r][1] = ds; M->tgt[r] = 0; ++r;
        M->a[r][1] = 1; M->a[r][2] = ds; M->tgt[r] = 1; ++r;  /* a10 = -k^2 ds: template k = 0 */
        M->a[r][2] = 1; M->a[r][5] = ds; M->tgt[r] = 2; ++r;
        M->a[r][5] = 1; ++r;            /* U box   */
        M->a[r][0] = 1; ++r;            /* end e_y */
        M->a[r][1] = 1; ++r;            /* end e_phi */
        const double Pm[PO_NVC] = {p->w_dev, 0, p->w_curv, p->w_slack, 0, keep * p->w_curv_rate, 0, p->w_slack};
        const double cn[PO_NVC] = {N, N, N, N, 0, C, 0, N};
        memcpy(M->Pmax, Pm, sizeof(Pm)); memcpy(M->cnt, cn, sizeof(cn));
    } else if (form == PO_KPC) {
        M->a[r][2] = 1; M->a[r][4] = 1; ++r;   /* kl */
        M->a[r][2] = 1; M->a[r][4] = -1; ++r;  /* ku */
        M->a[r][3] = 1; ++r;                   /* Sc box */
        M->a[r][4] = 1; ++r;                   /* Sk box */
        ROW2(d1); ROW2(d2); ROW2(d4);
        ROW2S(d3, -1); ROW2S(d3, 1);
        M->a[r][0] = 1; M->a[r][1] = ds; M->tgt[r] = 0; ++r;
        M->a[r][1] = 1; M->a[r][2] = ds; M->tgt[r] = 1; ++r;
        M->a[r][2] = 1; M->a[r][5] = ds; M->tgt[r] = 2; ++r;
        M->a[r][5] = 1; M->a[r][6] = 1; ++r;   /* kpl */
        M->a[r][5] = 1; M->a[r][6] = -1; ++r;  /* kpu */
        M->a[r][6] = 1; ++r;                   /* Skp >= 0 */
        M->a[r][0] = 1; ++r;
        M->a[r][1] = 1; ++r;
        const double Pm[PO_NVC] = {p->w_dev, 0, p->w_curv, p->w_slack, p->w_k_slack, keep * p->w_curv_rate, p->w_kp_slack * keep, 0};
        const double cn[PO_NVC] = {N, N, N, N, N, C, C, N - C};
        memcpy(M->Pmax, Pm, sizeof(Pm)); memcpy(M->cnt, cn, sizeof(cn));
    } else {
        M->a[r][1] = 1; ++r;  /* identity e_phi */
        M->a[r][0] = 1; ++r;  /* identity e_y   */
        M->a[r][2] = 1; ++r;  /* delta box      */
        M->a[r][3] = 1; ++r;  /* S box          */
        ROW2(d1); ROW2(d3); ROW2(d4);
        ROW2S(d2, -1); ROW2S(d2, 1);
        M->a[r][1] = 1; M->a[r][2] = ds / p->wheel_base; M->a[r][6] = ds / p->wheel_base; M->tgt[r] = 1; ++r;  /* e_phi equation, template k = 0 */
        M->a[r][0] = 1; M->a[r][1] = ds; M->tgt[r] = 0; ++r;                   /* e_y equation */
        /* The first and the last steering variable carry w_c + w_cr on the diagonal of R (solver_k_as_input.cpp:62-76), the others
         * w_c + 2 w_cr: OSQP's Ruiz passes give those two columns - and their box rows - their own factors.  Class 6 / row 11. */
        M->a[r][6] = 1; ++r;  /* box row of an end steering variable */
        const int nend = N - 1 < 2 ? N - 1 : 2;
        const double Pm[PO_NVC] = {p->k_w_dev, 0, p->k_w_curv + 2 * p->k_w_curv_rate, p->w_slack, 0, 0, p->k_w_curv + p->k_w_curv_rate, 0};
        const double cn[PO_NVC] = {N, N, N - 1 - nend, N, 0, 0, nend, 0};
        memcpy(M->Pmax, Pm, sizeof(Pm)); memcpy(M->cnt, cn, sizeof(cn));
    }
#undef ROW2
#undef ROW2S
    M->nr = r;
}

/* Dv[8], Er[nr] (row classes in the order built above), *c. */
static void class_ruiz(const class_model_t *M, int passes, double *Dv, double *Er, double *cs) {
    double c = 1.0;
    for (int v = 0; v < PO_NVC; ++v) Dv[v] = 1.0;
    for (int r = 0; r < M->nr; ++r) Er[r] = 1.0;
    double ntot = 0;
    for (int v = 0; v < PO_NVC; ++v) ntot += M->cnt[v];
    for (int pass = 0; pass < passes; ++pass) {
        double cn[PO_NVC], rn[PO_NRC];
        for (int v = 0; v < PO_NVC; ++v) cn[v] = fabs(c * M->Pmax[v] * Dv[v] * Dv[v]);
        for (int r = 0; r < M->nr; ++r) {
            double rmax = 0;
            for (int v = 0; v < PO_NVC; ++v) {
                double a = fabs(Er[r] * M->a[r][v] * Dv[v]);
                if (a > rmax) rmax = a;
                if (a > cn[v]) cn[v] = a;
            }
            if (M->tgt[r] >= 0) {
                double a = fabs(Er[r] * Dv[M->tgt[r]]);
                if (a > rmax) rmax = a;
                if (a > cn[M->tgt[r]]) cn[M->tgt[r]] = a;
            }
            rn[r] = rmax;
        }
        for (int v = 0; v < PO_NVC; ++v) Dv[v] *= 1.0 / sqrt(limit_scaling(cn[v]));
        for (int r = 0; r < M->nr; ++r) Er[r] *= 1.0 / sqrt(limit_scaling(rn[r]));
        double mean = 0;
        for (int v = 0; v < PO_NVC; ++v) mean += M->cnt[v] * fabs(c * M->Pmax[v] * Dv[v] * Dv[v]);
        mean /= ntot;
        double ct = mean > 1.0 ? mean : 1.0; /* ||q||_inf = 0 -> limit_scaling -> 1 */
        ct = 1.0 / limit_scaling(ct);
        c *= ct;
    }
    *cs = c;
}

/* Expand class factors to the reference variable/row ordering. */
int po_oracle_class_scaling(int form, const po_params *p, int N, int keep, double ds_nom, int passes,
                            double *D, double *E, double *cs) {
    int n, m, C;
    int rc = po_oracle_dims(form, N, keep, &n, &m, &C);
    if (rc) return rc;
    class_model_t M;
    double Dv[PO_NVC], Er[PO_NRC];
    class_model_build(&M, form, p, N, keep, ds_nom);
    class_ruiz(&M, passes, Dv, Er, cs);
    if (form == PO_KP) {
        for (int i = 0; i < N; ++i) {
            D[3 * i] = Dv[0]; D[3 * i + 1] = Dv[1]; D[3 * i + 2] = Dv[2];
            D[3 * N + C + i] = Dv[3]; D[3 * N + C + N + i] = Dv[7];
            E[3 * i] = Er[8]; E[3 * i + 1] = Er[9]; E[3 * i + 2] = Er[10];
            E[3 * N + i] = Er[0]; E[4 * N + C + i] = Er[1];
            const int cb = 5 * N + C;
            E[cb + 2 * i] = Er[2]; E[cb + 2 * i + 1] = Er[3];
            E[cb + 2 * N + i] = Er[4]; E[cb + 3 * N + i] = Er[5]; E[cb + 4 * N + i] = Er[6]; E[cb + 5 * N + i] = Er[7];
        }
        for (int c = 0; c < C; ++c) { D[3 * N + c] = Dv[5]; E[4 * N + c] = Er[11]; }
        E[11 * N + C] = Er[12]; E[11 * N + C + 1] = Er[13];
    } else if (form == PO_KPC) {
        const int sb = 5 * N + 2 * C, cb = 7 * N + 3 * C, s0 = 3 * N + C;
        for (int i = 0; i < N; ++i) {
            D[3 * i] = Dv[0]; D[3 * i + 1] = Dv[1]; D[3 * i + 2] = Dv[2];
            D[s0 + i] = Dv[3]; D[s0 + N + i] = Dv[4];
            E[3 * i] = Er[9]; E[3 * i + 1] = Er[10]; E[3 * i + 2] = Er[11];
            E[3 * N + i] = Er[0]; E[4 * N + i] = Er[1]; E[sb + i] = Er[2]; E[sb + N + i] = Er[3];
            E[cb + 3 * i] = Er[4]; E[cb + 3 * i + 1] = Er[5]; E[cb + 3 * i + 2] = Er[6];
            E[cb + 3 * N + i] = Er[7]; E[cb + 4 * N + i] = Er[8];
        }
        for (int c = 0; c < C; ++c) {
            D[3 * N + c] = Dv[5]; D[s0 + 2 * N + c] = Dv[6];
            E[5 * N + c] = Er[12]; E[5 * N + C + c] = Er[13]; E[sb + 2 * N + c] = Er[14];
        }
        for (int i = 0; i < N - C; ++i) D[s0 + 2 * N + C + i] = Dv[7];
        E[cb + 5 * N] = Er[15]; E[cb + 5 * N + 1] = Er[16];
    } else {
        for (int i = 0; i < N; ++i) {
            D[2 * i] = Dv[1]; D[2 * i + 1] = Dv[0]; D[3 * N - 1 + i] = Dv[3];
            const int kend = (i == 0 || i == N - 2);
            if (i < N - 1) D[2 * N + i] = kend ? Dv[6] : Dv[2];
            E[2 * i] = Er[9]; E[2 * i + 1] = Er[10];
            E[2 * N + 2 * i] = Er[0]; E[2 * N + 2 * i + 1] = Er[1];
            if (i < N - 1) E[4 * N + i] = kend ? Er[11] : Er[2];
            E[5 * N - 1 + i] = Er[3];
            E[6 * N - 1 + 3 * i] = Er[4]; E[6 * N - 1 + 3 * i + 1] = Er[5]; E[6 * N - 1 + 3 * i + 2] = Er[6];
            E[9 * N - 1 + i] = Er[7]; E[10 * N - 1 + i] = Er[8];
        }
    }
    return PO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 3: output map                                                                           */
/* ------------------------------------------------------------------------------------------ */
int po_oracle_output(int form, int N, const double *xs, const double *ref_x, const double *ref_y,
                     const double *ref_z, double *out) {
    double tmp_s = 0, px = 0, py = 0;
    for (int i = 0; i < N; ++i) {
        double ey, ephi, k;
        if (form == PO_K) { /* solver_k_as_input.cpp:22-44 */
            ey = xs[2 * i + 1];
            ephi = xs[2 * i];
            k = (i != N - 1) ? xs[2 * N + i] : xs[3 * N - 2];
        } else { /* solver_kp_as_input.cpp:26-43 (KPC identical) */
            ey = xs[3 * i];
            ephi = xs[3 * i + 1];
            k = xs[3 * i + 2];
        }
        double angle = ref_z[i];
        double new_angle = po_oracle_wrap_angle(angle + M_PI_2);
        double tx = ref_x[i] + ey * cos(new_angle);
        double ty = ref_y[i] + ey * sin(new_angle);
        if (i != 0) tmp_s += sqrt(pow(tx - px, 2) + pow(ty - py, 2));
        out[5 * i + 0] = tx;
        out[5 * i + 1] = ty;
        out[5 * i + 2] = angle + ephi;
        out[5 * i + 3] = k;
        out[5 * i + 4] = tmp_s;
        px = tx;
        py = ty;
    }
    return PO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 4: drivers                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* ordering cache: the sparsity pattern depends only on (form, N, keep) and on which data-dependent
 * entries are exactly zero; we key on (form, N, keep, nnzA, nnzP) */
typedef struct {
    int form, N, keep, anz, pnz, len, *perm;
} perm_cache_t;
static perm_cache_t g_cache[8];
static int g_cache_next = 0;

int po_oracle_solve_path(int form, const po_params *p, int N, int keep, const double *ref_x,
                         const double *ref_y, const double *ref_z, const double *ref_k,
                         const double *ref_s, const double *bounds, const double *x0,
                         double goal_z, const double *max_k, const double *max_kp,
                         double *out_states, double *out_x, double *out_y, po_info *info) {
    int n, m, C;
    int rc = po_oracle_dims(form, N, keep, &n, &m, &C);
    if (rc) return rc;
    const int ab = po_oracle_nnz_bound_A(form, N, keep), pb = po_oracle_nnz_bound_P(form, N, keep);
    int *Pp = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *Pi = (int *)malloc(sizeof(int) * (size_t)pb);
    int *Ap = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *Ai = (int *)malloc(sizeof(int) * (size_t)ab);
    double *Px = (double *)malloc(sizeof(double) * (size_t)pb), *Ax = (double *)malloc(sizeof(double) * (size_t)ab);
    double *l = (double *)malloc(sizeof(double) * (size_t)m), *u = (double *)malloc(sizeof(double) * (size_t)m);
    double *x = (double *)malloc(sizeof(double) * (size_t)n), *y = (double *)malloc(sizeof(double) * (size_t)m);
    double *z = (double *)malloc(sizeof(double) * (size_t)m);
    po_info li;
    int ok = 0;
    rc = po_oracle_assemble(form, p, N, keep, ref_k, ref_s, &ref_z[N - 1], bounds, x0, goal_z, max_k, max_kp, Pp, Pi, Px, Ap, Ai, Ax, l, u);
    if (rc == PO_OK) {
        const int *perm = NULL;
        for (int i = 0; i < 8; ++i)
            if (g_cache[i].perm && g_cache[i].form == form && g_cache[i].N == N && g_cache[i].keep == keep &&
                g_cache[i].anz == Ap[n] && g_cache[i].pnz == Pp[n]) perm = g_cache[i].perm;
        if (!perm) {
            /* compute a minimum-degree ordering once for this structure */
            double *ri = (double *)malloc(sizeof(double) * (size_t)m);
            for (int i = 0; i < m; ++i) ri[i] = 1.0;
            kkt_t K0;
            memset(&K0, 0, sizeof(K0));
            int *pm = (int *)malloc(sizeof(int) * (size_t)(n + m));
            if (kkt_build(&K0, n, m, Pp, Pi, Px, Ap, Ai, Ax, 1.0, ri, NULL) == 0 && min_degree_order(K0.nk, K0.Kp, K0.Ki, pm) == 0) {
                perm_cache_t *c = &g_cache[g_cache_next++ % 8];
                free(c->perm);
                c->form = form; c->N = N; c->keep = keep; c->anz = Ap[n]; c->pnz = Pp[n]; c->len = n + m; c->perm = pm;
                perm = pm;
            } else {
                free(pm);
            }
            kkt_free(&K0);
            free(ri);
        }
        if (p->scaling < 0) { /* class-level equilibration with the path's nominal step (max of the first <= 9 gaps) */
            double ds_nom = 0;
            for (int i = 1; i < N && i < 10; ++i) { double d = ref_s[i] - ref_s[i - 1]; if (d > ds_nom) ds_nom = d; }
            double *Dx = (double *)malloc(sizeof(double) * (size_t)n), *Ex = (double *)malloc(sizeof(double) * (size_t)m), cs = 1.0;
            for (int i = 0; i < n; ++i) Dx[i] = 1.0;
            for (int i = 0; i < m; ++i) Ex[i] = 1.0;
            po_oracle_class_scaling(form, p, N, keep, ds_nom, -p->scaling, Dx, Ex, &cs);
            rc = po_oracle_qp_solve_ext(n, m, Pp, Pi, Px, NULL, Ap, Ai, Ax, l, u, p, perm, Dx, Ex, cs, x, y, z, &li);
            free(Dx); free(Ex);
        } else {
            rc = po_oracle_qp_solve(n, m, Pp, Pi, Px, NULL, Ap, Ai, Ax, l, u, p, perm, x, y, z, &li);
        }
    }
    if (rc == PO_OK) {
        ok = li.status == PO_STATUS_SOLVED;
        if (info) *info = li;
        if (out_x) memcpy(out_x, x, sizeof(double) * (size_t)n);
        if (out_y) memcpy(out_y, y, sizeof(double) * (size_t)m);
        if (out_states) po_oracle_output(form, N, x, ref_x, ref_y, ref_z, out_states);
    }
    free(Pp); free(Pi); free(Ap); free(Ai); free(Px); free(Ax); free(l); free(u); free(x); free(y); free(z);
    return rc == PO_OK ? ok : rc;
}

int po_oracle_solve_batch(const po_params *p, const po_batch_in *in, const po_batch_out *out) {
    if (!p || !in || !out) return PO_ERR_INVALID;
    int n, m, C;
    int rc = po_oracle_dims(in->formulation, in->N, in->keep, &n, &m, &C);
    if (rc) return rc;
    const int N = in->N;  /* stride of every array; a ragged batch gives each path its own length n_points[b] <= N */
    for (int b = 0; b < in->B; ++b) {
        const size_t o = (size_t)b * (size_t)N;
        const int Nb = in->n_points ? in->n_points[b] : N;
        if (Nb < 2 || Nb > N) return PO_ERR_INVALID;
        int nb, mb, Cb;
        po_oracle_dims(in->formulation, Nb, in->keep, &nb, &mb, &Cb);
        double *xb = out->x ? out->x + (size_t)b * (size_t)n : NULL;
        if (out->states) memset(out->states + o * 5, 0, sizeof(double) * (size_t)N * 5);
        if (xb) memset(xb, 0, sizeof(double) * (size_t)n);
        int r = po_oracle_solve_path(in->formulation, p, Nb, in->keep, in->ref_x + o, in->ref_y + o, in->ref_z + o,
                                     in->ref_k + o, in->ref_s + o, in->bounds + o * 8, in->x0 + (size_t)b * 3,
                                     in->goal_z[b], in->max_k ? in->max_k + o : NULL, in->max_kp ? in->max_kp + o : NULL,
                                     out->states ? out->states + o * 5 : NULL, xb, NULL, out->info ? &out->info[b] : NULL);
        if (r < 0) return r;
    }
    return PO_OK;
}

int po_oracle_kkt_check(int n, int m, const int *Pp, const int *Pi, const double *Px, const double *q,
                        const int *Ap, const int *Ai, const double *Ax, const double *l, const double *u,
                        const double *x, const double *y, double *res) {
    double *t = (double *)malloc(sizeof(double) * (size_t)n), *t2 = (double *)malloc(sizeof(double) * (size_t)n);
    double *ax = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    if (!t || !t2 || !ax) return PO_ERR_NOMEM;
    sym_mv(n, Pp, Pi, Px, x, t);
    csc_mtv(n, Ap, Ai, Ax, y, t2);
    double obj = 0, stat = 0;
    for (int i = 0; i < n; ++i) {
        obj += 0.5 * x[i] * t[i] + (q ? q[i] * x[i] : 0);
        double g = t[i] + (q ? q[i] : 0) + t2[i];
        if (fabs(g) > stat) stat = fabs(g);
    }
    csc_mv(n, m, Ap, Ai, Ax, x, ax);
    double viol = 0, comp = 0;
    for (int i = 0; i < m; ++i) {
        double v = 0;
        if (ax[i] < l[i]) v = l[i] - ax[i];
        if (ax[i] > u[i]) v = ax[i] - u[i];
        if (v > viol) viol = v;
        double c = 0;
        if (y[i] > 0) c = u[i] >= PO_ORACLE_INFTY * OSQP_MIN_SCALING ? fabs(y[i]) : fabs(y[i]) * fabs(u[i] - ax[i]);
        else if (y[i] < 0) c = l[i] <= -PO_ORACLE_INFTY * OSQP_MIN_SCALING ? fabs(y[i]) : fabs(y[i]) * fabs(ax[i] - l[i]);
        if (c > comp) comp = c;
    }
    res[0] = stat;
    res[1] = viol;
    res[2] = comp;
    res[3] = obj;
    free(t); free(t2); free(ax);
    return PO_OK;
}


/* =====================================================================================================
 * Post-solve step.  In-tree logic: Map.cpp:16-26, collision_checker.cpp:17-59, car_geometry.cpp:38-72,
 * tools.cpp:50-55, path_optimizer.cpp:183-200 — pinned against those files compiled through oracle/ref_shim
 * (tests/test_oracle_vs_reference.py).  grid_map_core (GridMap::isInside / atPosition(INTER_LINEAR), GridMapMath.cpp) is a
 * third-party dependency absent from /root/reference: restated from its published sources, "parity unpinned".
 * ===================================================================================================== */
int po_oracle_map_inside(const po_map *m, double x, double y) {
    const double lx = m->size_x * m->resolution, ly = m->size_y * m->resolution;
    const double tx = -((x - m->pos_x) - 0.5 * lx), ty = -((y - m->pos_y) - 0.5 * ly);
    return tx >= 0.0 && ty >= 0.0 && tx < lx && ty < ly;
}
static void map_index(const po_map *m, double x, double y, int *ix, int *iy) {
    const double lx = m->size_x * m->resolution, ly = m->size_y * m->resolution;
    *ix = (int)(-(((x - 0.5 * lx) - m->pos_x) / m->resolution));
    *iy = (int)(-(((y - 0.5 * ly) - m->pos_y) / m->resolution));
}
static int map_index_ok(const po_map *m, int ix, int iy) { return ix >= 0 && iy >= 0 && ix < m->size_x && iy < m->size_y; }
static void map_position(const po_map *m, int ix, int iy, double *x, double *y) {
    if (!map_index_ok(m, ix, iy)) return; /* getPositionFromIndex returns false and leaves the output untouched */
    const double lx = m->size_x * m->resolution, ly = m->size_y * m->resolution;
    *x = (m->pos_x + (0.5 * lx - 0.5 * m->resolution)) + m->resolution * (double)(-ix);
    *y = (m->pos_y + (0.5 * ly - 0.5 * m->resolution)) + m->resolution * (double)(-iy);
}
float po_oracle_map_at_linear(const po_map *m, double x, double y) {
    int i0x, i0y, ix[4], iy[4], sh[4], up;
    double ptx = 0, pty = 0;
    map_index(m, x, y, &i0x, &i0y);
    map_position(m, i0x, i0y, &ptx, &pty);
    ix[0] = i0x; iy[0] = i0y;
    if (x >= ptx) { ix[1] = i0x - 1; iy[1] = i0y; up = 1; } else { ix[1] = i0x + 1; iy[1] = i0y; up = 0; }
    if (y >= pty) {
        ix[2] = i0x; iy[2] = i0y - 1;
        if (up) { sh[0] = 0; sh[1] = 1; sh[2] = 2; sh[3] = 3; } else { sh[0] = 1; sh[1] = 0; sh[2] = 3; sh[3] = 2; }
    } else {
        ix[2] = i0x; iy[2] = i0y + 1;
        if (up) { sh[0] = 2; sh[1] = 3; sh[2] = 0; sh[3] = 1; } else { sh[0] = 3; sh[1] = 2; sh[2] = 1; sh[3] = 0; }
    }
    ix[3] = ix[1]; iy[3] = iy[2];
    const long long nbuf = (long long)m->size_x * m->size_y;
    float f[4];
    int ok = 1;
    for (int i = 0; i < 4; ++i) {
        const long long lin = (long long)iy[sh[i]] * m->size_x + ix[sh[i]];
        if (lin < 0 || lin > nbuf) ok = 0;
        f[i] = (lin >= 0 && lin < nbuf) ? m->distance[lin] : 0.0f;
    }
    if (ok) {
        map_position(m, ix[sh[0]], iy[sh[0]], &ptx, &pty);
        const double rx = (x - ptx) / m->resolution, ry = (y - pty) / m->resolution;
        const double fx = 1.0 - rx, fy = 1.0 - ry;
        const double v = f[0] * fx * fy + f[1] * rx * fy + f[2] * fx * ry + f[3] * rx * ry;
        return (float)v;
    }
    return map_index_ok(m, i0x, i0y) ? m->distance[(long long)i0y * m->size_x + i0x] : 0.0f;
}
double po_oracle_map_distance(const po_map *m, double x, double y) {
    return po_oracle_map_inside(m, x, y) ? (double)po_oracle_map_at_linear(m, x, y) : 0.0;
}
int po_oracle_collision_free(const po_params *p, const po_map *m, double x, double y, double z) {
    /* CollisionChecker ctor (collision_checker.cpp:9-15) + CarGeometry::setCircles (car_geometry.cpp:38-56) */
    const double width = p->car_width, back = p->car_length / 2.0 - p->rear_axle_to_center, front = p->car_length / 2.0 + p->rear_axle_to_center;
    const double length = front + back;
    const double bcx = (front - back) / 2.0, bcr = sqrt((length / 2) * (length / 2) + (width / 2) * (width / 2));
    const double shift = width / 4.0, small_r = sqrt(2 * (shift * shift));
    const double large_r = sqrt(width * width + ((length - width) / 2.0) * ((length - width) / 2.0)) / 2;
    const double cx[6] = {-back + shift, -back + shift, front - shift, front - shift, bcx + (length - width) / 4, bcx - (length - width) / 4};
    const double cy[6] = {-width / 2.0 + shift, width / 2.0 - shift, -width / 2.0 + shift, width / 2.0 - shift, 0, 0};
    const double cr[6] = {small_r, small_r, small_r, small_r, large_r, large_r};
    const double cz = MCOS(z), sz = MSIN(z);
    const double bx = bcx * cz - 0.0 * sz + x, by = bcx * sz + 0.0 * cz + y; /* local2Global, tools.cpp:50-55 */
    if (!po_oracle_map_inside(m, bx, by)) return 0;
    if (!(po_oracle_map_distance(m, bx, by) < bcr)) return 1;
    for (int k = 0; k < 6; ++k) {
        const double gx = cx[k] * cz - cy[k] * sz + x, gy = cx[k] * sz + cy[k] * cz + y;
        if (!po_oracle_map_inside(m, gx, gy)) return 0;
        if (po_oracle_map_distance(m, gx, gy) < cr[k]) return 0;
    }
    return 1;
}
int po_oracle_postcheck(const po_params *p, const po_map *m, int n, const double *states, int status, int *n_valid) {
    if (status != PO_STATUS_SOLVED) { *n_valid = 0; return 0; }          /* "QP failed." -> false (path_optimizer.cpp:183-186) */
    for (int i = 0; i < n; ++i) {
        /* s is the running length the output map already wrote (same recurrence, :193-195) */
        if (p->enable_collision_check && !po_oracle_collision_free(p, m, states[5 * i], states[5 * i + 1], states[5 * i + 2])) {
            *n_valid = i;
            return i > 0 && states[5 * (i - 1) + 4] >= 20.0;          /* (the reference calls back() on an empty vector when i == 0) */
        }
    }
    *n_valid = n;
    return 1;
}


/* =====================================================================================================
 * Corridor-bounds producer.  In-tree logic pinned against reference_path_impl.cpp / spline.cpp / tools.cpp
 * compiled through oracle/ref_shim (oracle/_ref/libpo_ref_bounds.so, tests/test_bounds.py).
 * ===================================================================================================== */
/* Portable-math mode: the device's own derivation of the same natural cubic spline (Thomas algorithm on the second derivatives, csrc/po_post.hip
 * spline_fit), operation for operation — so that device and oracle agree bit for bit downstream.  Equal to the reference's spline to a few ulp. */
static void spline_fit_thomas(int K, const double *x, const double *y, double *a, double *b, double *c) {
    double *cp = (double *)malloc(sizeof(double) * (size_t)K * 3), *dp = cp + K, *M = cp + 2 * K;
    cp[0] = 0.0; dp[0] = 0.0;
    for (int i = 1; i < K - 1; ++i) {
        const double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
        const double rhs = 6.0 * ((y[i + 1] - y[i]) / hr - (y[i] - y[i - 1]) / hl);
        const double piv = 2.0 * (hl + hr) - hl * cp[i - 1];
        cp[i] = hr / piv;
        dp[i] = (rhs - hl * dp[i - 1]) / piv;
    }
    M[K - 1] = 0.0;
    for (int i = K - 2; i >= 1; --i) M[i] = dp[i] - cp[i] * M[i + 1];
    M[0] = 0.0;
    for (int i = 0; i < K - 1; ++i) {
        const double h = x[i + 1] - x[i];
        b[i] = 0.5 * M[i];
        a[i] = (M[i + 1] - M[i]) / (6.0 * h);
        c[i] = (y[i + 1] - y[i]) / h - h * (2.0 * M[i] + M[i + 1]) / 6.0;
    }
    const double h = x[K - 1] - x[K - 2];
    b[K - 1] = 0.0;
    a[K - 1] = 0.0;
    c[K - 1] = 3.0 * a[K - 2] * h * h + 2.0 * b[K - 2] * h + c[K - 2];
    free(cp);
}
void po_oracle_spline_fit(int K, const double *x, const double *y, double *a, double *b, double *c) {
    if (g_portable_math) { spline_fit_thomas(K, x, y, a, b, c); return; }
    /* set_points (spline.cpp:168-250): tridiagonal system for b[], rows normalised to a unit diagonal, LU without pivoting
     * (band_matrix::lu_decompose, :70-101), then l_solve / r_solve (:103-130).  Scratch: lo, di, up, sd, rhs in the outputs. */
    double *lo = a, *up = c; /* reuse: a <- lower band, c <- upper band until the solve is done */
    double *di = (double *)malloc(sizeof(double) * (size_t)K * 3), *sd = di + K, *rh = di + 2 * K;
    for (int i = 1; i < K - 1; ++i) {
        lo[i] = 1.0 / 3.0 * (x[i] - x[i - 1]);
        di[i] = 2.0 / 3.0 * (x[i + 1] - x[i - 1]);
        up[i] = 1.0 / 3.0 * (x[i + 1] - x[i]);
        rh[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    }
    di[0] = 2.0; up[0] = 0.0; rh[0] = 0.0; lo[0] = 0.0;
    di[K - 1] = 2.0; lo[K - 1] = 0.0; rh[K - 1] = 0.0; up[K - 1] = 0.0;
    for (int i = 0; i < K; ++i) { /* preconditioning */
        sd[i] = 1.0 / di[i];
        if (i > 0) lo[i] *= sd[i];
        if (i < K - 1) up[i] *= sd[i];
        di[i] = 1.0;
    }
    for (int k = 0; k + 1 < K; ++k) { /* Gauss */
        const double xx = -lo[k + 1] / di[k];
        lo[k + 1] = -xx;
        di[k + 1] = di[k + 1] + xx * up[k];
    }
    for (int i = 0; i < K; ++i) rh[i] = (rh[i] * sd[i]) - (i > 0 ? lo[i] * rh[i - 1] : 0.0);           /* l_solve (in place) */
    for (int i = K - 1; i >= 0; --i) b[i] = (rh[i] - (i < K - 1 ? up[i] * b[i + 1] : 0.0)) / di[i];  /* r_solve */
    for (int i = 0; i < K - 1; ++i) {
        a[i] = 1.0 / 3.0 * (b[i + 1] - b[i]) / (x[i + 1] - x[i]);
        c[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - 1.0 / 3.0 * (2.0 * b[i] + b[i + 1]) * (x[i + 1] - x[i]);
    }
    const double h = x[K - 1] - x[K - 2];
    a[K - 1] = 0.0;
    c[K - 1] = 3.0 * a[K - 2] * h * h + 2.0 * b[K - 2] * h + c[K - 2];
    free(di);
}
double po_oracle_spline_eval(int K, const double *x, const double *y, const double *a, const double *b, const double *c, double at) {
    /* operator() (spline.cpp:251-271): idx = max(lower_bound(x, at) - 1, 0) */
    int lo = 0, hi = K;
    while (lo < hi) { const int mid = (lo + hi) / 2; if (x[mid] < at) lo = mid + 1; else hi = mid; }
    const int idx = lo - 1 > 0 ? lo - 1 : 0;
    const double h = at - x[idx];
    if (at < x[0]) return (b[0] * h + c[0]) * h + y[0];
    if (at > x[K - 1]) return (b[K - 1] * h + c[K - 1]) * h + y[K - 1];
    return ((a[idx] * h + b[idx]) * h + c[idx]) * h + y[idx];
}
/* getClearanceWithDirectionStrict (reference_path_impl.cpp:283-472), out[0] = left bound, out[1] = right bound */
static void clearance_strict(const po_map *m, double radius, double sx, double sy, double sz, double out[2]) {
    double left_bound = 0, right_bound = 0;
    const double delta_s = 0.5;
    const double la = po_oracle_wrap_angle(sz + M_PI_2), ra = po_oracle_wrap_angle(sz - M_PI_2);
    const int n = (int)(5.0 / delta_s);
    const double cl = MCOS(la), sl = MSIN(la), cr = MCOS(ra), sr = MSIN(ra);
    if (po_oracle_map_distance(m, sx, sy) > radius) { /* normal case */
        double right_s = 0, left_s = 0;
        for (int j = 0; j != n; ++j) { right_s += delta_s; if (po_oracle_map_distance(m, sx + right_s * cr, sy + right_s * sr) < radius) break; }
        for (int j = 0; j != n; ++j) { left_s += delta_s; if (po_oracle_map_distance(m, sx + left_s * cl, sy + left_s * sl) < radius) break; }
        right_bound = -(right_s - delta_s);
        left_bound = left_s - delta_s;
    } else { /* in collision already: expand both ways until free, pick the nearer side (:383-440) */
        double right_s = 0, left_s = 0;
        for (int j = 0; j != n; ++j) { right_s += delta_s; if (po_oracle_map_distance(m, sx + right_s * cr, sy + right_s * sr) > radius) break; }
        for (int j = 0; j != n; ++j) { left_s += delta_s; if (po_oracle_map_distance(m, sx + left_s * cl, sy + left_s * sl) > radius) break; }
        if (left_s < right_s) {
            right_bound = left_s;
            for (int j = 0; j != n; ++j) { left_s += delta_s; if (po_oracle_map_distance(m, sx + left_s * cl, sy + left_s * sl) < radius) break; }
            left_bound = left_s - delta_s;
        } else {
            left_bound = -right_s;
            for (int j = 0; j != n; ++j) { right_s += delta_s; if (po_oracle_map_distance(m, sx + right_s * cr, sy + right_s * sr) < radius) break; }
            right_bound = -(right_s - delta_s);
        }
    }
    const double smaller_ds = 0.1;
    const int nf = (int)(delta_s / smaller_ds);
    for (int i = 1; i != nf; ++i) {
        left_bound += smaller_ds;
        if (po_oracle_map_distance(m, sx + left_bound * cl, sy + left_bound * sl) < radius) { left_bound -= smaller_ds; break; }
    }
    for (int i = 1; i != nf; ++i) {
        right_bound -= smaller_ds;
        if (po_oracle_map_distance(m, sx + right_bound * cr, sy + right_bound * sr) < radius) { right_bound += smaller_ds; break; }
    }
    out[0] = left_bound; out[1] = right_bound;
}
int po_oracle_bounds_path(const po_params *p, const po_map *m, int N, const double *ref_x, const double *ref_y, const double *ref_z,
                          const double *ref_s, int K, const double *ks, const double *kx, const double *ky, double *bounds) {
    double *co = (double *)malloc(sizeof(double) * (size_t)K * 6);
    double *xa = co, *xb = co + K, *xc = co + 2 * K, *ya = co + 3 * K, *yb = co + 4 * K, *yc = co + 5 * K;
    po_oracle_spline_fit(K, ks, kx, xa, xb, xc);
    po_oracle_spline_fit(K, ks, ky, ya, yb, yc);
    /* FLAGS_circle_radius as updateConfig() computes it (planning_flags.cpp:9) */
    const double radius = sqrt((p->car_length / 8) * (p->car_length / 8) + (p->car_width / 2) * (p->car_width / 2)) + p->safety_margin;
    int kept = 0;
    for (int i = 0; i < N; ++i) {
        const double cz = MCOS(ref_z[i]), sz = MSIN(ref_z[i]);
        double cl[4][2];
        int blocked = 0;
        for (int j = 0; j < 4; ++j) {
            const double len = p->d[j];
            const double cx = ref_x[i] + len * cz, cy = ref_y[i] + len * sz;            /* circle centre on the tangent (:150-161) */
            /* getApproxState (:121-140) */
            const double px = po_oracle_spline_eval(K, ks, kx, xa, xb, xc, ref_s[i] + len), py = po_oracle_spline_eval(K, ks, ky, ya, yb, yc, ref_s[i] + len);
            const double v1x = cx - ref_x[i], v1y = cy - ref_y[i], v2x = px - ref_x[i], v2y = py - ref_y[i];
            const double nrm = sqrt(v1x * v1x + v1y * v1y);
            const double proj = (v1x * v2x + v1y * v2y) / (0.001 > nrm ? 0.001 : nrm);
            const double move = fabs(len) - proj;
            const int sign = len >= 0 ? 1 : -1;
            const double ax = px + sign * move * cz, ay = py + sign * move * sz;
            double c2[2];
            clearance_strict(m, radius, ax, ay, ref_z[i], c2);
            /* offset = global2Local(c_j, c_jj).y (tools.cpp:57-64) */
            const double dx = ax - cx, dy = ay - cy;
            const double off = -dx * sz + dy * cz;
            cl[j][0] = c2[0] + off; cl[j][1] = c2[1] + off;
            if (fabs(cl[j][0] - cl[j][1]) < 1e-6) blocked = 1;                        /* isEqual, FLAGS_epsilon (tools.cpp:28-30) */
        }
        if (blocked) break;
        for (int j = 0; j < 4; ++j) { bounds[(i * 4 + j) * 2] = cl[j][1]; bounds[(i * 4 + j) * 2 + 1] = cl[j][0]; }
        ++kept;
    }
    free(co);
    return kept;
}


/* ------------------------------------------------------------------------------------------ */
/* Part 6: the reference-smoothing QPs.  Same OsqpEigen call pattern as the   */
/* hot path; assembly restated in the reference's variable / row order with its dense-scratch   */
/* + sparseView() semantics, solved by po_oracle_qp_solve (true Ruiz passes, scaling = 10).     */
/*   TENSION2 : src/reference_path_smoother/tension_smoother_2.cpp:163-301                      */
/*   TENSION  : src/reference_path_smoother/tension_smoother.cpp:186-314                        */
/*   POST     : src/reference_path_smoother/reference_path_smoother.cpp:534-644                 */
/* ------------------------------------------------------------------------------------------ */
int po_oracle_smooth_dims(int kind, int P, int *n, int *m) {
    int nn, mm;
    if (kind == PO_SMOOTH_TENSION2) { if (P < 3) return PO_ERR_INVALID; nn = 4 * P - 1; mm = 3 * (P - 1) + 2; }   /* tension_smoother_2.cpp:177-178 */
    else if (kind == PO_SMOOTH_TENSION) { if (P < 3) return PO_ERR_INVALID; nn = 3 * P; mm = 3 * P; }             /* tension_smoother.cpp:201-202 */
    else if (kind == PO_SMOOTH_POST) { if (P < 4) return PO_ERR_INVALID; nn = 3 * P; mm = 3 * P - 2; }            /* reference_path_smoother.cpp:536,544-545 */
    else return PO_ERR_INVALID;
    if (n) *n = nn;
    if (m) *m = mm;
    return PO_OK;
}

int po_oracle_smooth_assemble(int kind, const po_params *p, const po_map *map, int P, const double *x, const double *y,
                              const double *angle, const double *k, const double *s, const double *lb, const double *ub, double l0,
                              int *Pp, int *Pi, double *Px, double *q, int *Ap, int *Ai, double *Ax, double *l, double *u) {
    int n, m;
    if (po_oracle_smooth_dims(kind, P, &n, &m)) return PO_ERR_INVALID;
    tripbuf H = {0, 0, 0}, A = {0, 0, 0};
    int rc = PO_OK;
    for (int i = 0; i < n; ++i) q[i] = 0;
    for (int i = 0; i < m; ++i) l[i] = u[i] = 0;
    if (kind == PO_SMOOTH_TENSION2) {
        const int xs = 0, ys = P, ts = 2 * P, ks = 3 * P;
        /* setHessianMatrix, tension_smoother_2.cpp:220-241 */
        for (int i = 0; i < P; ++i) {
            TSET(&H, xs + i, xs + i, p->t2_w_dev * 2);
            TSET(&H, ys + i, ys + i, p->t2_w_dev * 2);
            if (i != P - 1) TSET(&H, ks + i, ks + i, p->t2_w_curv * 2);
        }
        for (int i = 0; i < P - 2; ++i) { /* hessian.block(k+i, k+i, 2, 2) += 2 * w * [1 -1; -1 1] */
            const double w2 = 2 * p->t2_w_curv_rate;
            TADD(&H, ks + i, ks + i, w2 * 1.0);
            TADD(&H, ks + i + 1, ks + i, w2 * -1.0);
            TADD(&H, ks + i, ks + i + 1, w2 * -1.0);
            TADD(&H, ks + i + 1, ks + i + 1, w2 * 1.0);
        }
        /* setGradient, :288-299 */
        for (int i = 0; i < P; ++i) {
            q[xs + i] = -2 * p->t2_w_dev * x[i];
            q[ys + i] = -2 * p->t2_w_dev * y[i];
        }
        /* setConstraintMatrix, :243-286 */
        const int cx = 0, cy = P - 1, ct = 2 * (P - 1), c0x = 3 * (P - 1), c0y = 3 * (P - 1) + 1;
        for (int i = 0; i < P - 1; ++i) {
            const double ds = s[i + 1] - s[i];
            TSET(&A, cx + i, xs + i + 1, 1.0); TSET(&A, cy + i, ys + i + 1, 1.0); TSET(&A, ct + i, ts + i + 1, 1.0);
            TSET(&A, cx + i, xs + i, -1.0); TSET(&A, cy + i, ys + i, -1.0); TSET(&A, ct + i, ts + i, -1.0);
            TSET(&A, cx + i, ts + i, ds * sin(angle[i]));
            TSET(&A, cy + i, ts + i, -ds * cos(angle[i]));
            TSET(&A, ct + i, ks + i, -ds);
            l[cx + i] = u[cx + i] = ds * cos(angle[i]);
            l[cy + i] = u[cy + i] = ds * sin(angle[i]);
            l[ct + i] = u[ct + i] = -ds * k[i];
        }
        TSET(&A, c0x, xs, 1.0); TSET(&A, c0y, ys, 1.0);
        l[c0x] = u[c0x] = x[0];
        l[c0y] = u[c0y] = y[0];
    } else if (kind == PO_SMOOTH_TENSION) {
        const int xs = 0, ys = P, dsi = 2 * P;
        /* setHessianMatrix, tension_smoother.cpp:238-261: dds = [1 -2 1]'[1 -2 1] * w_c, ddds = [-1 3 -3 1]'[-1 3 -3 1] * w_cr */
        static const double v3[3] = {1, -2, 1}, v4[4] = {-1, 3, -3, 1};
        for (int i = 0; i < P - 2; ++i) {
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    TADD(&H, xs + i + a, xs + i + b, v3[a] * v3[b] * p->cart_w_curv);
                    TADD(&H, ys + i + a, ys + i + b, v3[a] * v3[b] * p->cart_w_curv);
                }
            if (i != P - 3)
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        TADD(&H, xs + i + a, xs + i + b, v4[a] * v4[b] * p->cart_w_curv_rate);
                        TADD(&H, ys + i + a, ys + i + b, v4[a] * v4[b] * p->cart_w_curv_rate);
                    }
        }
        for (int i = 0; i < P; ++i) TSET(&H, dsi + i, dsi + i, p->cart_w_dev);
        /* setConstraintMatrix, :263-314 */
        for (int i = 0; i < P; ++i) {
            TSET(&A, xs + i, xs + i, 1.0); TSET(&A, ys + i, ys + i, 1.0);
            const double theta = angle[i] + M_PI_2;
            TSET(&A, xs + i, dsi + i, -cos(theta));
            TSET(&A, ys + i, dsi + i, -sin(theta));
            TSET(&A, dsi + i, dsi + i, 1.0);
            l[xs + i] = u[xs + i] = x[i];
            l[ys + i] = u[ys + i] = y[i];
        }
        l[dsi] = u[dsi] = 0;
        l[dsi + P - 1] = -0.5; u[dsi + P - 1] = 0.5;
        for (int i = 1; i < P - 1; ++i) {
            double clearance = map ? po_oracle_map_distance(map, x[i], y[i]) : 2.0;
            clearance = clearance < 2.0 ? clearance : 2.0; /* std::min(clearance, default_clearance) */
            l[dsi + i] = -clearance; u[dsi + i] = clearance;
        }
    } else { /* POST: setPostHessianMatrix :598-612, setPostConstraintMatrix :614-650 */
        const int L = P, xi = 0, dxi = L, ddxi = 2 * L, cdx = L, cddx = 2 * L - 1;
        for (int i = 0; i < L; ++i) { TSET(&H, i, i, 1.0); TSET(&H, L + i, L + i, 100.0); TSET(&H, 2 * L + i, 2 * L + i, 1000.0); }
        for (int i = 0; i < L; ++i) TSET(&A, i, i, 1.0);
        for (int i = 0; i < L - 1; ++i) {
            TSET(&A, cdx + i, xi + i + 1, 1.0); TSET(&A, cdx + i, xi + i, -1.0); TSET(&A, cdx + i, dxi + i, -(s[i + 1] - s[i]));
        }
        for (int i = 0; i < L - 1; ++i) {
            TSET(&A, cddx + i, dxi + i + 1, 1.0); TSET(&A, cddx + i, dxi + i, -1.0); TSET(&A, cddx + i, ddxi + i, -(s[i + 1] - s[i]));
        }
        l[0] = u[0] = l0;
        for (int i = 1; i < L; ++i) { l[i] = lb[i]; u[i] = ub[i]; }
    }
    if (!H.t || !A.t) rc = PO_ERR_NOMEM;
    if (!rc) { tb_compress(&H, n, 1, Pp, Pi, Px); tb_compress(&A, n, 0, Ap, Ai, Ax); }
    free(H.t); free(A.t);
    return rc;
}

/* osqpSmooth / postSmooth for one instance: assemble, OSQP-style solve, and the output loop
 * (tension_smoother_2.cpp:204-217 == tension_smoother.cpp:222-235: result lists + running chord length). Returns 1 iff "solved". */
int po_oracle_smooth_solve(int kind, const po_params *p, const po_map *map, int P, const double *x, const double *y, const double *angle,
                           const double *k, const double *s, const double *lb, const double *ub, double l0,
                           double *out_x, double *out_y, double *out_s, double *raw, po_info *info) {
    int n, m;
    if (po_oracle_smooth_dims(kind, P, &n, &m)) return PO_ERR_INVALID;
    const int pcap = 16 * P + 16, acap = 4 * m + 16;
    int *Pp = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *Pi = (int *)malloc(sizeof(int) * (size_t)pcap);
    int *Ap = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *Ai = (int *)malloc(sizeof(int) * (size_t)acap);
    double *Px = (double *)malloc(sizeof(double) * (size_t)pcap), *Ax = (double *)malloc(sizeof(double) * (size_t)acap);
    double *q = (double *)malloc(sizeof(double) * (size_t)n), *l = (double *)malloc(sizeof(double) * (size_t)m), *u = (double *)malloc(sizeof(double) * (size_t)m);
    double *xs = (double *)malloc(sizeof(double) * (size_t)n), *ys = (double *)malloc(sizeof(double) * (size_t)m), *zs = (double *)malloc(sizeof(double) * (size_t)m);
    int rc = po_oracle_smooth_assemble(kind, p, map, P, x, y, angle, k, s, lb, ub, l0, Pp, Pi, Px, q, Ap, Ai, Ax, l, u);
    if (!rc) rc = po_oracle_qp_solve(n, m, Pp, Pi, Px, q, Ap, Ai, Ax, l, u, p, NULL, xs, ys, zs, info);
    int ok = 0;
    if (!rc) {
        ok = info->status == PO_STATUS_SOLVED;
        if (raw) memcpy(raw, xs, sizeof(double) * (size_t)n);
        if (kind == PO_SMOOTH_POST) {
            for (int i = 0; i < P; ++i) out_x[i] = xs[i];
        } else {
            double tmp_s = 0;
            for (int i = 0; i < P; ++i) {
                out_x[i] = xs[i];
                out_y[i] = xs[P + i];
                if (i != 0) {
                    const double ddx = out_x[i] - out_x[i - 1], ddy = out_y[i] - out_y[i - 1];
                    tmp_s += sqrt(ddx * ddx + ddy * ddy);
                }
                out_s[i] = tmp_s;
            }
        }
    }
    free(Pp); free(Pi); free(Ap); free(Ai); free(Px); free(Ax); free(q); free(l); free(u); free(xs); free(ys); free(zs);
    return rc ? rc : ok;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 7: reference re-sampling, limits and the DP lattice search.           */
/*   tk::spline::deriv             src/tools/spline.cpp:273-318                                 */
/*   getHeading / getCurvature / global2Local / findClosestPoint   src/tools/tools.cpp:34-112   */
/*   buildReferenceFromSpline / updateLimits   src/data_struct/reference_path_impl.cpp:474-499 / :203-235 */
/*   graphSearchDp / calculateCostAt   src/reference_path_smoother/reference_path_smoother.cpp:147-300 / :110-145 */
/* ------------------------------------------------------------------------------------------ */
#include <float.h>
double po_oracle_spline_deriv(int K, const double *x, const double *y, const double *a, const double *b, const double *c, int order, double at) {
    (void)y;
    int lo = 0, hi = K;
    while (lo < hi) { const int mid = (lo + hi) / 2; if (x[mid] < at) lo = mid + 1; else hi = mid; }
    const int idx = lo - 1 > 0 ? lo - 1 : 0;
    const double h = at - x[idx];
    if (at < x[0]) return order == 1 ? 2.0 * b[0] * h + c[0] : (order == 2 ? 2.0 * b[0] * h : 0.0); /* sic: the reference's order-2 left branch keeps the h */
    if (at > x[K - 1]) return order == 1 ? 2.0 * b[K - 1] * h + c[K - 1] : (order == 2 ? 2.0 * b[K - 1] : 0.0);
    if (order == 1) return (3.0 * a[idx] * h + 2.0 * b[idx]) * h + c[idx];
    if (order == 2) return 6.0 * a[idx] * h + 2.0 * b[idx];
    return order == 3 ? 6.0 * a[idx] : 0.0;
}

typedef struct { int K; const double *s, *vx, *vy; double *ax, *bx, *cx, *ay, *by, *cy; } spl2_t;
static int spl2_init(spl2_t *S, int K, const double *ks, const double *kx, const double *ky) {
    S->K = K; S->s = ks; S->vx = kx; S->vy = ky;
    S->ax = (double *)malloc(sizeof(double) * (size_t)K * 6);
    if (!S->ax) return -1;
    S->bx = S->ax + K; S->cx = S->ax + 2 * K; S->ay = S->ax + 3 * K; S->by = S->ax + 4 * K; S->cy = S->ax + 5 * K;
    po_oracle_spline_fit(K, ks, kx, S->ax, S->bx, S->cx);
    po_oracle_spline_fit(K, ks, ky, S->ay, S->by, S->cy);
    return 0;
}
static double spl2_x(const spl2_t *S, double at) { return po_oracle_spline_eval(S->K, S->s, S->vx, S->ax, S->bx, S->cx, at); }
static double spl2_y(const spl2_t *S, double at) { return po_oracle_spline_eval(S->K, S->s, S->vy, S->ay, S->by, S->cy, at); }
static double spl2_dx(const spl2_t *S, int o, double at) { return po_oracle_spline_deriv(S->K, S->s, S->vx, S->ax, S->bx, S->cx, o, at); }
static double spl2_dy(const spl2_t *S, int o, double at) { return po_oracle_spline_deriv(S->K, S->s, S->vy, S->ay, S->by, S->cy, o, at); }
static double spl2_heading(const spl2_t *S, double at) { return MATAN2(spl2_dy(S, 1, at), spl2_dx(S, 1, at)); } /* tools.cpp:34-38 */
static double spl2_curvature(const spl2_t *S, double at) {                                                      /* tools.cpp:40-46 */
    const double x1 = spl2_dx(S, 1, at), y1 = spl2_dy(S, 1, at), x2 = spl2_dx(S, 2, at), y2 = spl2_dy(S, 2, at);
    return (x1 * y2 - y1 * x2) / MPOW15(x1 * x1 + y1 * y1);
}
/* findClosestPoint(xs, ys, x, y, max_s, start_s = 0): the arc length it settles on (tools.cpp:71-112) */
static double spl2_closest_s(const spl2_t *S, double x, double y, double max_s) {
    const double start_s = 0.0;
    if (max_s <= start_s) return 0.0; /* State{xs(start_s), ys(start_s)}: s stays 0 */
    double tmp_s = start_s, min_dis_s = start_s, min_dis = DBL_MAX;
    while (tmp_s <= max_s) {
        const double ddx = spl2_x(S, tmp_s) - x, ddy = spl2_y(S, tmp_s) - y;
        const double d = sqrt(ddx * ddx + ddy * ddy);
        if (d < min_dis) { min_dis = d; min_dis_s = tmp_s; }
        tmp_s += 0.5;
    }
    double cur_s = min_dis_s, prev_s = min_dis_s;
    for (int i = 0; i < 20; ++i) { /* Newton on the squared distance */
        const double px = spl2_x(S, cur_s), py = spl2_y(S, cur_s), dx = spl2_dx(S, 1, cur_s), dy = spl2_dy(S, 1, cur_s);
        const double ddx = spl2_dx(S, 2, cur_s), ddy = spl2_dy(S, 2, cur_s);
        const double j = (px - x) * dx + (py - y) * dy;
        const double hh = dx * dx + (px - x) * ddx + dy * dy + (py - y) * ddy;
        cur_s -= j / hh;
        if (fabs(cur_s - prev_s) < 1e-5) break;
        prev_s = cur_s;
    }
    return cur_s < max_s ? cur_s : max_s;
}

int po_oracle_resample(const po_params *p, int K, const double *ks, const double *kx, const double *ky, double max_s, double ds_smaller,
                       double ds_larger, int N, double *ox, double *oy, double *oz, double *ok, double *os) {
    if (max_s <= 0) return -1; /* "Cannot build reference line from spline!" */
    spl2_t S;
    if (spl2_init(&S, K, ks, kx, ky)) return PO_ERR_NOMEM;
    const double large_k = 0.2, small_k = 0.08;
    double tmp_s = 0;
    int n = 0;
    while (tmp_s <= max_s) {
        if (n == N) { n = -2; break; }
        const double k = spl2_curvature(&S, tmp_s);
        ox[n] = spl2_x(&S, tmp_s); oy[n] = spl2_y(&S, tmp_s); oz[n] = spl2_heading(&S, tmp_s); ok[n] = k; os[n] = tmp_s;
        ++n;
        if (p->enable_dynamic_segmentation) {
            const double k_share = fabs(k) > large_k ? 1 : fabs(k) < small_k ? 0 : (fabs(k) - small_k) / (large_k - small_k);
            tmp_s += ds_larger - k_share * (ds_larger - ds_smaller);
        } else tmp_s += ds_larger;
    }
    free(S.ax);
    return n;
}

void po_oracle_limits(const po_params *p, int N, const double *v, const double *a, double *max_k, double *max_kp) {
    for (int i = 0; i < N; ++i) {
        const double mg = p->mu * 9.8;
        const double ay_allowed = sqrt(mg * mg - a[i] * a[i]);
        max_k[i] = v[i] > 0.0001 ? ay_allowed / (v[i] * v[i]) : DBL_MAX;
        max_kp[i] = v[i] > 0.0001 ? p->max_curvature_rate / v[i] : DBL_MAX;
    }
}

#define PO_DP_MAXLAT 64
int po_oracle_dp_search(const po_params *p, const po_map *map, int K, const double *ks, const double *kx, const double *ky, double length,
                        const double *start /*x,y,heading*/, int Lcap, double *layer_s, double *lb, double *ub, double *l0) {
    spl2_t S;
    if (spl2_init(&S, K, ks, kx, ky)) return PO_ERR_NOMEM;
    const double range = p->search_lateral_range, spacing = p->search_lat_spacing;
    const double search_threshold = 1.45;
    int rc = 0;
    /* layers */
    double tmp_s = spl2_closest_s(&S, start[0], start[1], length);
    const double search_ds = length > 6 ? p->search_long_spacing : 0.5;
    int L = 0;
    double *ls = (double *)malloc(sizeof(double) * (size_t)(Lcap + 1));
    while (tmp_s < length) {
        if (L >= Lcap) { rc = -2; goto done0; }
        ls[L++] = tmp_s;
        tmp_s += search_ds;
    }
    if (L >= Lcap) { rc = -2; goto done0; }
    ls[L++] = length;
    {
        const double vs = ls[0], pxr = spl2_x(&S, vs), pyr = spl2_y(&S, vs), pz = spl2_heading(&S, vs);
        const double dx = start[0] - pxr, dy = start[1] - pyr;
        const double vl = -dx * MSIN(pz) + dy * MCOS(pz); /* global2Local(proj_point, start_state).y */
        *l0 = vl;
        if (fabs(vl) > range) { rc = -1; goto done0; }
        const int start_idx = (int)((range + vl) / spacing);
        /* lateral offsets by the reference's running sum */
        double lat[PO_DP_MAXLAT];
        int nlat = 0;
        for (double cur_l = -range; cur_l <= range && nlat < PO_DP_MAXLAT; cur_l += spacing) lat[nlat++] = cur_l;
        typedef struct { double x, y, heading, s, l, cost, dir, dis, rlo, rhi; int parent, feas; } node_t;
        node_t *nd = (node_t *)malloc(sizeof(node_t) * (size_t)L * (size_t)nlat);
        for (int i = 0; i < L; ++i) {
            const double cur_s = ls[i], rx = spl2_x(&S, cur_s), ry = spl2_y(&S, cur_s), rh = spl2_heading(&S, cur_s);
            const double rk = spl2_curvature(&S, cur_s), rr = 1 / rk;
            node_t *row = nd + (size_t)i * nlat;
            for (int j = 0; j < nlat; ++j) {
                node_t *q = row + j;
                const double cur_l = lat[j];
                q->x = rx + cur_l * MCOS(rh + M_PI_2); q->y = ry + cur_l * MSIN(rh + M_PI_2);
                q->heading = rh; q->s = cur_s; q->l = cur_l; q->cost = DBL_MAX; q->dir = 0; q->parent = -1; q->feas = 1;
                q->dis = po_oracle_map_inside(map, q->x, q->y) ? po_oracle_map_distance(map, q->x, q->y) : -1;
                if ((rk < 0 && cur_l < rr) || (rk > 0 && cur_l > rr) || q->dis < search_threshold) q->feas = 0;
                if (i == 0 && j != start_idx) q->feas = 0;
                if (i == 0 && j == start_idx) { q->feas = 1; q->dir = start[2]; q->cost = 0.0; }
            }
            for (int j = 0; j < nlat; ++j) row[j].rlo = (j == 0 || !row[j - 1].feas || !row[j].feas) ? row[j].l : row[j - 1].rlo;
            for (int j = nlat - 1; j >= 0; --j) row[j].rhi = (j == nlat - 1 || !row[j + 1].feas || !row[j].feas) ? row[j].l : row[j + 1].rhi;
        }
        /* cost recursion (calculateCostAt) */
        int max_layer = 0;
        for (int i = 0; i < L; ++i) {
            int any = 0;
            node_t *row = nd + (size_t)i * nlat;
            for (int j = 0; j < nlat && i > 0; ++j) {
                node_t *q = row + j;
                if (!q->feas) continue;
                double self = 0;
                if (q->dis < 3.0) self += (3.0 - q->dis) / 3.0 * 0.5;
                self += fabs(q->l) / range * 1.0;
                double min_cost = DBL_MAX;
                const node_t *prow = row - nlat;
                for (int k = 0; k < nlat; ++k) {
                    const node_t *pp = prow + k;
                    if (!pp->feas) continue;
                    if (fabs(pp->l - q->l) > (q->s - pp->s)) continue;
                    const double direction = MATAN2(q->y - pp->y, q->x - pp->x);
                    const double edge = fabs(po_oracle_wrap_angle(direction - pp->dir)) / M_PI_2 * 16.0 +
                                        fabs(po_oracle_wrap_angle(direction - q->heading)) / M_PI_2 * 0.5;
                    const double total = self + edge + pp->cost;
                    if (total < min_cost) { min_cost = total; q->parent = k; q->dir = direction; }
                }
                if (q->parent >= 0) { q->cost = min_cost; any = 1; }
            }
            if (i != 0 && !any) break;
            max_layer = i;
        }
        /* cheapest node of the last reachable layer, then walk back */
        int best = -1;
        double min_cost = DBL_MAX;
        for (int j = 0; j < nlat; ++j)
            if (nd[(size_t)max_layer * nlat + j].cost < min_cost) { best = j; min_cost = nd[(size_t)max_layer * nlat + j].cost; }
        int count = 0;
        for (int i = max_layer, j = best; j >= 0 && i >= 0; --i) {
            const node_t *q = nd + (size_t)i * nlat + j;
            double lo, hi;
            if (i == 0) { lo = -10; hi = 10; }
            else {
                const double check_s = 0.2, check_limit = 6.0;
                hi = check_s + q->rhi; lo = -check_s + q->rlo;
                const double rx = spl2_x(&S, q->s), ry = spl2_y(&S, q->s);
                while (hi < check_limit) {
                    const double px2 = rx + hi * MCOS(q->heading + M_PI_2), py2 = ry + hi * MSIN(q->heading + M_PI_2);
                    if (po_oracle_map_inside(map, px2, py2) && po_oracle_map_distance(map, px2, py2) > search_threshold) hi += check_s;
                    else { hi -= check_s; break; }
                }
                while (lo > -check_limit) {
                    const double px2 = rx + lo * MCOS(q->heading + M_PI_2), py2 = ry + lo * MSIN(q->heading + M_PI_2);
                    if (po_oracle_map_inside(map, px2, py2) && po_oracle_map_distance(map, px2, py2) > search_threshold) lo -= check_s;
                    else { lo += check_s; break; }
                }
            }
            lb[i] = lo; ub[i] = hi; /* index == layer: the chain is contiguous from max_layer down to 0 (std::reverse) */
            ++count;
            j = q->parent;
            if (i == 0) break;
        }
        for (int i = 0; i < count; ++i) layer_s[i] = ls[i]; /* layers_s_list_.resize(layers_bounds_.size()) */
        rc = count;
        free(nd);
    }
done0:
    free(ls);
    free(S.ax);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 8: the remaining glue stages of PathOptimizer::solve (path_optimizer.cpp:40-178):       */
/*   ReferencePathSmoother::bSpline        reference_path_smoother.cpp:495-532                  */
/*     tinyspline (ROS package tinyspline_ros) is NOT in /root/reference and un-pinned: its      */
/*     clamped B-spline (uniform interior knots, de Boor evaluation) is restated from the        */
/*     published algorithm — "parity unpinned" for that library.                                 */
/*   ReferencePathSmoother::segmentRawReference   :50-91                                        */
/*   the tail of ReferencePathSmoother::postSmooth :568-590 (re-projection of the QP offsets)   */
/*   PathOptimizer::segmentSmoothedPath    path_optimizer.cpp:119-169 (initial errors, goal trim) */
/* ------------------------------------------------------------------------------------------ */
/* clamped B-spline of degree deg with n control points (dim 2) on [0,1]; knots[i] = 0 (i <= deg), (i - deg) / (n - deg), 1 (i >= n) */
static double bs_knot(int i, int n, int deg) {
    if (i <= deg) return 0.0;
    if (i >= n) return 1.0;
    const double fac = (1.0 - 0.0) / (double)(n + deg + 1 - 2 * deg - 1);
    return fac * (double)(i - deg) + 0.0;
}
int po_oracle_bspline_eval(int n, int deg, const double *cx, const double *cy, double u, double *ox, double *oy) {
    if (deg < 1 || n <= deg || deg > 7) return PO_ERR_INVALID;
    if (u >= 1.0) { *ox = cx[n - 1]; *oy = cy[n - 1]; return PO_OK; } /* u at the last knot (multiplicity = order): the last control point */
    if (u <= 0.0) { *ox = cx[0]; *oy = cy[0]; return PO_OK; }
    int k = deg; /* span: knots[k] <= u < knots[k+1] */
    while (k + 1 < n && bs_knot(k + 1, n, deg) <= u) ++k;
    double dx[8], dy[8];
    for (int j = 0; j <= deg; ++j) { dx[j] = cx[k - deg + j]; dy[j] = cy[k - deg + j]; }
    for (int r = 1; r <= deg; ++r)
        for (int j = deg; j >= r; --j) {
            const int i = k - deg + j;
            const double ki = bs_knot(i, n, deg), kj = bs_knot(i + deg - r + 1, n, deg);
            const double a = (u - ki) / (kj - ki);
            dx[j] = (1.0 - a) * dx[j - 1] + a * dx[j];
            dy[j] = (1.0 - a) * dy[j - 1] + a * dy[j];
        }
    *ox = dx[deg]; *oy = dy[deg];
    return PO_OK;
}
/* bSpline(): returns the number of samples written to x, y, s (x_list_, y_list_, s_list_), < 0 on error / cap too small (-2) */
int po_oracle_bspline_sample(int n, const double *px, const double *py, int cap, double *x, double *y, double *s) {
    if (n < 4) return -1; /* "Few reference points." (ReferencePathSmoother::solve) */
    double length = 0;
    for (int i = 0; i + 1 < n; ++i) {
        const double ddx = px[i] - px[i + 1], ddy = py[i] - py[i + 1];
        length += sqrt(ddx * ddx + ddy * ddy);
    }
    const double average_length = length / (n - 1);
    const int degree = average_length > 10 ? 3 : (average_length > 5 ? 4 : 5);
    if (n <= degree) return -1; /* tinyspline throws (fewer control points than the order) */
    const double delta_t = 1.0 / length;
    double tmp_t = 0;
    int m = 0;
    while (tmp_t < 1) {
        if (m >= cap - 1) return -2;
        po_oracle_bspline_eval(n, degree, px, py, tmp_t, &x[m], &y[m]);
        ++m;
        tmp_t += delta_t;
    }
    po_oracle_bspline_eval(n, degree, px, py, 1.0, &x[m], &y[m]);
    ++m;
    s[0] = 0;
    for (int i = 1; i < m; ++i) {
        const double ddx = x[i] - x[i - 1], ddy = y[i] - y[i - 1];
        s[i] = s[i - 1] + sqrt(ddx * ddx + ddy * ddy);
    }
    return m;
}

/* segmentRawReference: spline through the dense raw lists, 1 m stations (the last one may lie beyond max_s), heading and curvature */
int po_oracle_segment_raw(int K, const double *ks, const double *kx, const double *ky, int cap, double *x, double *y, double *s, double *angle, double *k) {
    spl2_t S;
    if (K < 3) return -1;
    if (spl2_init(&S, K, ks, kx, ky)) return PO_ERR_NOMEM;
    const double max_s = ks[K - 1], delta_s = 1.0;
    int n = 0, rc = 0;
    s[n++] = 0;
    while (s[n - 1] < max_s) {
        if (n >= cap) { rc = -2; break; }
        s[n] = s[n - 1] + delta_s;
        ++n;
    }
    /* `if (max_s - s_list->back() > 1)` can never hold after the loop */
    for (int i = 0; i < n && !rc; ++i) {
        const double at = s[i], dx = spl2_dx(&S, 1, at), dy = spl2_dy(&S, 1, at), ddx = spl2_dx(&S, 2, at), ddy = spl2_dy(&S, 2, at);
        angle[i] = MATAN2(dy, dx);
        k[i] = (dx * ddy - dy * ddx) / MPOW15(dx * dx + dy * dy);
        x[i] = spl2_x(&S, at); y[i] = spl2_y(&S, at);
    }
    free(S.ax);
    return rc ? rc : n;
}

/* postSmooth's tail: x = xs(s) + l MCOS(dir + pi/2), y = ys(s) + l MSIN(dir + pi/2), running chord length */
int po_oracle_post_project(int K, const double *ks, const double *kx, const double *ky, int L, const double *layer_s, const double *offsets,
                           double *x, double *y, double *s) {
    spl2_t S;
    if (K < 3 || L < 1) return PO_ERR_INVALID;
    if (spl2_init(&S, K, ks, kx, ky)) return PO_ERR_NOMEM;
    double acc = 0;
    for (int i = 0; i < L; ++i) {
        const double ref_s = layer_s[i], ref_dir = spl2_heading(&S, ref_s);
        x[i] = spl2_x(&S, ref_s) + offsets[i] * MCOS(ref_dir + M_PI_2);
        y[i] = spl2_y(&S, ref_s) + offsets[i] * MSIN(ref_dir + M_PI_2);
        if (i > 0) {
            const double ddx = x[i] - x[i - 1], ddy = y[i] - y[i - 1];
            acc += sqrt(ddx * ddx + ddy * ddy);
        }
        s[i] = acc;
    }
    free(S.ax);
    return PO_OK;
}

/* segmentSmoothedPath up to (not including) the re-sampling: returns 1 (go on) or 0 (the reference returns false);
 * out[0] = initial_offset, out[1] = initial_heading_error, out[2] = the (possibly trimmed) length */
int po_oracle_segment_init(int K, const double *ks, const double *kx, const double *ky, double length, const double *start /*x,y,z*/,
                           const double *goal /*x,y*/, int exact_position, double *out) {
    spl2_t S;
    out[0] = out[1] = 0; out[2] = length;
    if (length == 0) return 0; /* "Smoothed path is empty!" */
    if (K < 3) return 0;
    if (spl2_init(&S, K, ks, kx, ky)) return PO_ERR_NOMEM;
    int ok = 1;
    const double fx = spl2_x(&S, 0), fy = spl2_y(&S, 0), fz = spl2_heading(&S, 0);
    const double dx = fx - start[0], dy = fy - start[1];
    const double local_y = -dx * MSIN(start[2]) + dy * MCOS(start[2]); /* global2Local(start_state, first_point).y */
    const double min_distance = sqrt((start[0] - fx) * (start[0] - fx) + (start[1] - fy) * (start[1] - fy));
    out[0] = local_y < 0 ? min_distance : -min_distance;
    out[1] = po_oracle_wrap_angle(start[2] - fz);
    if (fabs(out[1]) > 75 * M_PI / 180) ok = 0; /* "Initial psi error is larger than 75 deg" */
    if (ok) {
        const double ex = goal[0] - spl2_x(&S, length), ey = goal[1] - spl2_y(&S, length);
        const double end_distance = sqrt(ex * ex + ey * ey);
        if (!(fabs(end_distance - 0) < 1e-6)) { /* isEqual(end_distance, 0), FLAGS_epsilon */
            const double search_delta_s = exact_position ? 0.1 : 0.5;
            double tmp_s = length - search_delta_s, min_dis_to_goal = end_distance, min_dis_s = length;
            while (tmp_s > 0) {
                const double px = spl2_x(&S, tmp_s), py = spl2_y(&S, tmp_s);
                const double tmp_dis = sqrt((px - goal[0]) * (px - goal[0]) + (py - goal[1]) * (py - goal[1]));
                if (tmp_dis < min_dis_to_goal) { min_dis_to_goal = tmp_dis; min_dis_s = tmp_s; }
                if (tmp_dis > 8 && min_dis_to_goal < 8) break;
                tmp_s -= search_delta_s;
            }
            out[2] = min_dis_s;
        }
    }
    free(S.ax);
    return ok;
}


/* optimizePath's densifying output branch (path_optimizer.cpp:201-226, FLAGS_enable_raw_output = false).  states [n][5] as solved;
 * out [cap][5]; returns ok, *n_out = samples kept (-2: cap too small). */
int po_oracle_densify(const po_params *p, const po_map *m, int n, const double *states, int status, int cap, double *out, int *n_out) {
    *n_out = 0;
    if (status != PO_STATUS_SOLVED || n < 3) return 0;
    double *rx = (double *)malloc(sizeof(double) * (size_t)n * 3), *ry = rx + n, *rs = rx + 2 * n;
    for (int i = 0; i < n; ++i) { rx[i] = states[5 * i]; ry[i] = states[5 * i + 1]; rs[i] = states[5 * i + 4]; }
    spl2_t S;
    if (spl2_init(&S, n, rs, rx, ry)) { free(rx); return 0; }
    const double delta_s = p->output_spacing;
    int ok = 1, cnt = 0;
    for (int i = 0; i * delta_s <= rs[n - 1]; ++i) {
        const double tmp_s = i * delta_s;
        const double x = spl2_x(&S, tmp_s), y = spl2_y(&S, tmp_s), z = spl2_heading(&S, tmp_s), k = spl2_curvature(&S, tmp_s);
        if (p->enable_collision_check && !po_oracle_collision_free(p, m, x, y, z)) {
            ok = cnt > 0 && out[5 * (cnt - 1) + 4] >= 20.0;
            break;
        }
        if (cnt >= cap) { cnt = -2; ok = 0; break; }
        out[5 * cnt] = x; out[5 * cnt + 1] = y; out[5 * cnt + 2] = z; out[5 * cnt + 3] = k; out[5 * cnt + 4] = tmp_s;
        ++cnt;
    }
    *n_out = cnt;
    free(S.ax); free(rx);
    return ok;
}
