/*
 * po_oracle.h — CPU oracle for the QP hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libpo_hip.so) never links, loads or calls anything in oracle/.
 *
 * PARITY PINNING STATUS
 *   - assembly (P, A, l, u) and the output map: pinned against the reference's own source files
 *     compiled from /root/reference via oracle/ref_shim (see oracle/Makefile target `ref`,
 *     tests/test_oracle_vs_reference.py, tests/golden/ref_assembly_*.npz).
 *   - the ADMM arithmetic: the reference delegates it to OSQP + OsqpEigen, which are NOT in
 *     /root/reference (cloned un-pinned from upstream master, scripts/install_deps.sh:102,116) and
 *     are not installed here.  This file restates the published OSQP algorithm (Stellato et al.,
 *     "OSQP: an operator splitting solver for quadratic programs", Math. Prog. Comp. 2020; OSQP
 *     ~0.6 defaults).  => "parity unpinned" for the solver arithmetic: the reference holds no golden
 *     vectors and no test asserting a numeric output.  Correctness is anchored on solver-independent
 *     KKT certificates and an interior-point cross-check instead (tests/test_oracle.py).
 */
#ifndef PO_ORACLE_H_
#define PO_ORACLE_H_

#include "../include/po_hip.h" /* shared plain-C types only: po_params, po_info, enums */

#ifdef __cplusplus
extern "C" {
#endif

#define PO_ORACLE_INFTY 1e30 /* OsqpEigen::INFTY == OSQP_INFTY */

void po_oracle_default_params(po_params *p);
int  po_oracle_dims(int form, int N, int keep, int *n, int *m, int *C);
int  po_oracle_keep(int form, const double *ref_s, int N);
/* constraintAngle (include/path_optimizer/tools/tools.hpp:24-35) */
double po_oracle_wrap_angle(double a);

/* Upper bounds on nnz so callers can size CSC buffers. */
int po_oracle_nnz_bound_A(int form, int N, int keep);
int po_oracle_nnz_bound_P(int form, int N, int keep);

/* Assemble one path's QP in the REFERENCE variable/row order (App. A of SURVEY.md):
 * P upper-triangular CSC (exact zeros dropped, like Eigen's sparseView()), A CSC, l, u. */
int po_oracle_assemble(int form, const po_params *p, int N, int keep,
                       const double *ref_k, const double *ref_s, const double *ref_z_last,
                       const double *bounds /*[N][4][2]*/, const double *x0 /*[3]*/, double goal_z,
                       const double *max_k, const double *max_kp,
                       int *Pp, int *Pi, double *Px, int *Ap, int *Ai, double *Ax,
                       double *l, double *u);

/* OSQP-style ADMM on a general QP  min 0.5 x'Px + q'x  s.t. l <= Ax <= u.
 * perm: optional fill-reducing permutation of the (n+m) KKT nodes (new -> old), or NULL.
 * x,y,z are outputs (cold start x=z=y=0, as the reference's fresh solver per call). */
int po_oracle_qp_solve(int n, int m, const int *Pp, const int *Pi, const double *Px, const double *q,
                       const int *Ap, const int *Ai, const double *Ax, const double *l, const double *u,
                       const po_params *p, const int *perm, double *x, double *y, double *z,
                       po_info *info);

/* As po_oracle_qp_solve but with an externally supplied diagonal equilibration (D per variable, E per row,
 * cost scale c) replacing the Ruiz passes when D != NULL. */
int po_oracle_qp_solve_ext(int n, int m, const int *Pp, const int *Pi, const double *Px, const double *q,
                           const int *Ap, const int *Ai, const double *Ax, const double *l, const double *u,
                           const po_params *p, const int *perm, const double *D, const double *E, double c,
                           double *x, double *y, double *z, po_info *info);

/* Class-level ("structured") Ruiz equilibration expanded to the reference ordering; used when
 * po_params.scaling < 0 (|scaling| passes).  This is what the device engine implements. */
int po_oracle_class_scaling(int form, const po_params *p, int N, int keep, double ds_nom, int passes,
                            double *D, double *E, double *c);

/* Stage-interleaved KKT permutation for a formulation (keeps LDL' fill O(n)). perm has n+m entries. */
int po_oracle_kkt_perm(int form, int N, int keep, int *perm);

/* getOptimizedPath (solver_kp_as_input.cpp:26-43 etc.): out [N][5] = x,y,heading,k,s. */
int po_oracle_output(int form, int N, const double *xsol, const double *ref_x, const double *ref_y,
                     const double *ref_z, double *out);

/* OsqpSolver::solve for one path (solver.cpp:46-77): returns 1 (true) iff status == solved. */
int po_oracle_solve_path(int form, const po_params *p, int N, int keep,
                         const double *ref_x, const double *ref_y, const double *ref_z,
                         const double *ref_k, const double *ref_s,
                         const double *bounds, const double *x0, double goal_z,
                         const double *max_k, const double *max_kp,
                         double *out_states /*[N][5]*/, double *out_x /*[n] or NULL*/,
                         double *out_y /*[m] or NULL*/, po_info *info);

/* Batch driver over po_oracle_solve_path (sequential; used by tests and the CPU baseline). */
int po_oracle_solve_batch(const po_params *p, const po_batch_in *in, const po_batch_out *out);

/* Solver-independent KKT certificate of (x,y) for the assembled QP:
 * res[0]=||Px+q+A'y||_inf  res[1]=max bound violation of Ax  res[2]=max complementarity violation
 * (y_i>0 needs (Ax)_i at u_i, y_i<0 at l_i: |y_i|*dist) res[3]=objective 0.5x'Px+q'x */
int po_oracle_kkt_check(int n, int m, const int *Pp, const int *Pi, const double *Px, const double *q,
                        const int *Ap, const int *Ai, const double *Ax, const double *l, const double *u,
                        const double *x, const double *y, double *res);

/* ---- post-solve step ----
 * Map::isInside / Map::getObstacleDistance (src/tools/Map.cpp:16-26) over the grid_map conventions restated in
 * include/po_hip.h (grid_map_core itself is absent from /root/reference: that part is "parity unpinned"). */
int    po_oracle_map_inside(const po_map *m, double x, double y);
double po_oracle_map_distance(const po_map *m, double x, double y);
/* grid_map::GridMap::atPosition("distance", p, INTER_LINEAR) for a position inside the map (float result) */
float  po_oracle_map_at_linear(const po_map *m, double x, double y);
/* CollisionChecker::isSingleStateCollisionFreeImproved (src/tools/collision_checker.cpp:42-59) */
int    po_oracle_collision_free(const po_params *p, const po_map *m, double x, double y, double heading);
/* the output loop of PathOptimizer::optimizePath (src/path_optimizer/path_optimizer.cpp:183-200), raw-output branch:
 * returns ok, writes the number of states kept.  states [n][5] = x,y,heading,k,s as produced by the solve. */
int    po_oracle_postcheck(const po_params *p, const po_map *m, int n, const double *states, int status, int *n_valid);

/* ---- corridor-bounds producer ----
 * tk::spline (src/tools/spline.cpp:154-271, natural boundary conditions): coefficients a,b,c [K] from knots (ks, kv). */
void   po_oracle_spline_fit(int K, const double *ks, const double *kv, double *a, double *b, double *c);
double po_oracle_spline_eval(int K, const double *ks, const double *kv, const double *a, const double *b, const double *c, double at);
/* ReferencePathImpl::updateBoundsImproved (src/data_struct/reference_path_impl.cpp:142-201) with getApproxState (:121-140) and
 * getClearanceWithDirectionStrict (:283-472, FLAGS_enable_simple_boundary_decision = true as shipped) for one path.
 * bounds [N][4][2] = (lb, ub) of circles c0..c3; returns the number of states kept (the loop stops at the first blocked one). */
int    po_oracle_bounds_path(const po_params *p, const po_map *m, int N, const double *ref_x, const double *ref_y, const double *ref_z,
                             const double *ref_s, int K, const double *ks, const double *kx, const double *ky, double *bounds);

/* ---- reference-smoothing QPs (kinds PO_SMOOTH_* of include/po_hip.h) ----
 * TensionSmoother2::osqpSmooth (tension_smoother_2.cpp:163-301), TensionSmoother::osqpSmooth (tension_smoother.cpp:186-314),
 * ReferencePathSmoother::postSmooth's QP (reference_path_smoother.cpp:534-650): assembly in the reference's variable/row order. */
int po_oracle_smooth_dims(int kind, int P, int *n, int *m);
int po_oracle_smooth_assemble(int kind, const po_params *p, const po_map *map, int P, const double *x, const double *y,
                              const double *angle, const double *k, const double *s, const double *lb, const double *ub, double l0,
                              int *Pp, int *Pi, double *Px, double *q, int *Ap, int *Ai, double *Ax, double *l, double *u);
/* one instance end to end (assemble, po_oracle_qp_solve, output lists); returns 1 iff solved, 0 otherwise, < 0 on misuse */
int po_oracle_smooth_solve(int kind, const po_params *p, const po_map *map, int P, const double *x, const double *y, const double *angle,
                           const double *k, const double *s, const double *lb, const double *ub, double l0,
                           double *out_x, double *out_y, double *out_s, double *raw, po_info *info);

/* ---- reference re-sampling, limits and the DP lattice search ---- */
double po_oracle_spline_deriv(int K, const double *ks, const double *kv, const double *a, const double *b, const double *c, int order, double at);
/* ReferencePathImpl::buildReferenceFromSpline: returns the number of states (<= N), -1 if the reference returns false, -2 if N is too small */
int po_oracle_resample(const po_params *p, int K, const double *ks, const double *kx, const double *ky, double max_s, double ds_smaller,
                       double ds_larger, int N, double *ox, double *oy, double *oz, double *ok, double *os);
/* ReferencePathImpl::updateLimits, states-given-directly branch */
void po_oracle_limits(const po_params *p, int N, const double *v, const double *a, double *max_k, double *max_kp);
/* ReferencePathSmoother::graphSearchDp: returns layers kept (layer_s, lb, ub filled), -1 if the reference returns false, -2 if Lcap is too small */
int po_oracle_dp_search(const po_params *p, const po_map *map, int K, const double *ks, const double *kx, const double *ky, double length,
                        const double *start, int Lcap, double *layer_s, double *lb, double *ub, double *l0);

/* ---- the remaining glue stages of PathOptimizer::solve (path_optimizer.cpp:40-178) ---- */
/* tinyspline's clamped B-spline (library absent, un-pinned: restated from the published algorithm, parity unpinned) */
int po_oracle_bspline_eval(int n, int deg, const double *cx, const double *cy, double u, double *ox, double *oy);
/* ReferencePathSmoother::bSpline: x_list_, y_list_, s_list_; returns their length, -1 (too few points), -2 (cap too small) */
int po_oracle_bspline_sample(int n, const double *px, const double *py, int cap, double *x, double *y, double *s);
/* ReferencePathSmoother::segmentRawReference: returns the number of 1 m stations, -1 / -2 as above */
int po_oracle_segment_raw(int K, const double *ks, const double *kx, const double *ky, int cap, double *x, double *y, double *s, double *angle, double *k);
/* the tail of postSmooth: the QP offsets re-projected onto the spline -> knots of the new spline */
int po_oracle_post_project(int K, const double *ks, const double *kx, const double *ky, int L, const double *layer_s, const double *offsets,
                           double *x, double *y, double *s);
/* PathOptimizer::segmentSmoothedPath before the re-sampling: 1 / 0 as the reference's return value so far; out = offset, heading error, length */
int po_oracle_segment_init(int K, const double *ks, const double *kx, const double *ky, double length, const double *start, const double *goal,
                           int exact_position, double *out);

/* optimizePath's densifying output branch (path_optimizer.cpp:201-226) */
int po_oracle_densify(const po_params *p, const po_map *m, int n, const double *states, int status, int cap, double *out, int *n_out);

#ifdef __cplusplus
}
#endif
#endif
