/*
 * po_hip.h — C ABI of libpo_hip.so, the MI355X (gfx950) batched QP path-optimisation engine.
 *
 * This is the drop-in boundary for the reference's OSQP wrapper.  The reference has no FFI layer:
 * its lower edge is the OsqpEigen::Solver member used in
 *     /root/reference/src/solver/solver.cpp:46-77            (OsqpSolver::solve)
 * and its upper edge is
 *     /root/reference/include/path_optimizer/solver/solver.hpp:31-36
 *         OsqpSolver::create(type, ReferencePath&, VehicleState&, horizon) / solve(std::vector<State>*)
 * called only from /root/reference/src/path_optimizer/path_optimizer.cpp:182-183.
 * Everything between those two edges (Hessian/constraint/bound assembly, the ADMM iteration, the
 * Frenet->Cartesian output map) runs behind the entry points below.  Plain pointers and sizes only.
 *
 * All floating point data is IEEE double ("f64"), as in the reference.
 */
#ifndef PO_HIP_H_
#define PO_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---- formulations: the three subclasses OsqpSolver::create() can return (solver.cpp:30-44) ---- */
enum {
    PO_KP  = 0, /* "KP"  SolverKpAsInput            src/solver/solver_kp_as_input.cpp             */
    PO_KPC = 1, /* "KPC" SolverKpAsInputConstrained src/solver/solver_kp_as_input_constrained.cpp */
    PO_K   = 2  /* "K"   SolverKAsInput             src/solver/solver_k_as_input.cpp              */
};

/* ---- API return codes (never abort; the reference's CHECK_* aborts become codes) ---- */
enum {
    PO_OK              = 0,
    PO_ERR_INVALID     = -1, /* bad argument (null pointer, N < 2, unknown formulation, ...) */
    PO_ERR_HIP         = -2, /* a HIP runtime call failed; po_last_hip_error() has the text  */
    PO_ERR_UNSUPPORTED = -3, /* problem does not fit the device tile (N too large)           */
    PO_ERR_NOMEM       = -4
};

/* ---- per-path solver status; values follow OSQP's so that "solved" == 1 maps to the
 *      reference's `solver_.solve()` returning true (OsqpEigen returns true only for OSQP_SOLVED) ---- */
enum {
    PO_STATUS_SOLVED            = 1,
    PO_STATUS_MAX_ITER          = -2,
    PO_STATUS_PRIMAL_INFEASIBLE = -3,
    PO_STATUS_DUAL_INFEASIBLE   = -4,
    PO_STATUS_NON_FINITE        = -8, /* not an OSQP value: a NaN / Inf appeared in the iterate (non-finite inputs); never reported as solved */
    PO_STATUS_UNSOLVED          = -10
};

/* Parameter block.  Replaces the gflags globals the reference reads during assembly
 * (src/config/planning_flags.cpp:8-14,18-43,102-119) plus the OSQP settings
 * (reference touches only verbosity/warm_start, src/solver/solver.cpp:48-49; the rest are
 * OSQP defaults, except eps which this project's metric fixes at 1e-4). Immutable after po_create. */
typedef struct po_params {
    double d[4];            /* FLAGS_d1..d4: rear-axle -> covering-circle centre offsets          */
    double w_curv;          /* FLAGS_KP_curvature_weight       (10)                                */
    double w_curv_rate;     /* FLAGS_KP_curvature_rate_weight  (200)                               */
    double w_dev;           /* FLAGS_KP_deviation_weight       (0)                                 */
    double w_slack;         /* FLAGS_KP_slack_weight           (3)  (also used by K, k_as_input.cpp:54) */
    double k_w_curv;        /* FLAGS_K_curvature_weight        (50)                                */
    double k_w_curv_rate;   /* FLAGS_K_curvature_rate_weight   (200)                               */
    double k_w_dev;         /* FLAGS_K_deviation_weight        (0)                                 */
    double w_k_slack;       /* KPC literal 500    (solver_kp_as_input_constrained.cpp:52)          */
    double w_kp_slack;      /* KPC literal 25000  (solver_kp_as_input_constrained.cpp:53)          */
    double margin;          /* FLAGS_expected_safety_margin    (1.3)                               */
    double max_steer;       /* FLAGS_max_steering_angle        (30 deg)                            */
    double wheel_base;      /* FLAGS_wheel_base                (2.85)                              */
    int    constraint_end_heading; /* FLAGS_constraint_end_heading (true)                          */
    int    scaling;         /* equilibration passes: 10 = OSQP default (the reference never changes it);
                               the engine runs the class-level form of Ruiz; 0 = off  */
    /* ADMM (OSQP names) */
    double eps_abs, eps_rel;            /* 1e-4, 1e-4 (project metric; OSQP default is 1e-3)       */
    double eps_prim_inf, eps_dual_inf;  /* 1e-4, 1e-4                                              */
    double rho0, sigma, alpha;          /* 0.1, 1e-6, 1.6                                          */
    double adapt_tol;                   /* adaptive_rho_tolerance 5                                */
    int    max_iter;                    /* 4000                                                    */
    int    check_every;                 /* check_termination 25                                    */
    int    adapt_every;                 /* adaptive-rho interval in iterations (100 = OSQP's
                                           non-profiling default 4*check_termination); 0 = off     */
    int    enable_collision_check;      /* FLAGS_enable_collision_check (true), post-solve step only */
    /* vehicle footprint, read by the post-solve collision check (planning_flags.cpp:18-29) */
    double car_width;           /* FLAGS_car_width           (2.0)  */
    double car_length;          /* FLAGS_car_length          (4.9)  */
    double rear_axle_to_center; /* FLAGS_rear_axle_to_center (1.45) */
    double safety_margin;       /* FLAGS_safety_margin       (0.0): circle_radius = sqrt((L/8)^2+(W/2)^2) + safety_margin */
    /* reference-smoothing QPs, planning_flags.cpp:76-86 */
    double t2_w_dev;            /* FLAGS_tension_2_deviation_weight       (0.005) */
    double t2_w_curv;           /* FLAGS_tension_2_curvature_weight       (1)     */
    double t2_w_curv_rate;      /* FLAGS_tension_2_curvature_rate_weight  (10)    */
    double cart_w_curv;         /* FLAGS_cartesian_curvature_weight       (1)     */
    double cart_w_curv_rate;    /* FLAGS_cartesian_curvature_rate_weight  (50)    */
    double cart_w_dev;          /* FLAGS_cartesian_deviation_weight       (0)     */
    /* re-sampling, limits and the DP lattice search, planning_flags.cpp:41-43,57-63,137 */
    double mu;                  /* FLAGS_mu                          (0.4)  */
    double max_curvature_rate;  /* FLAGS_max_curvature_rate          (0.1)  */
    double search_lateral_range;   /* FLAGS_search_lateral_range        (10.0) */
    double search_long_spacing;    /* FLAGS_search_longitudial_spacing  (1.5)  */
    double search_lat_spacing;     /* FLAGS_search_lateral_spacing      (0.6)  */
    int    enable_dynamic_segmentation; /* FLAGS_enable_dynamic_segmentation (true) */
    int    enable_raw_output;           /* FLAGS_enable_raw_output (true): output the QP states directly; false: densify through a spline */
    double output_spacing;              /* FLAGS_output_spacing (0.3) */
    /* po_plan_batch only: which smoother QP and which path formulation the chain uses */
    int    smoothing_method;            /* FLAGS_smoothing_method: PO_SMOOTH_TENSION2 (default "TENSION2") or PO_SMOOTH_TENSION ("TENSION") */
    int    optimization_method;         /* FLAGS_optimization_method: PO_KP (default "KP"), PO_K ("K") or PO_KPC */
    int    enable_exact_position;       /* FLAGS_enable_exact_position (false): goal-trim search step 0.1 m instead of 0.5 m (path_optimizer.cpp:151) */
    int    polish;                      /* OSQP `polish` (0 = off: OSQP's default, which the reference never changes).  1: after a path is solved, the reduced
                                           KKT system on the active set is solved with the regularisation `polish_delta` and `polish_refine_iter` steps of
                                           iterative refinement (OSQP polish.c); the result replaces the ADMM solution when OSQP's acceptance rule holds.  Shapes
                                           without a polish kernel (KP keep 6 .. 16, and the single-level mapping): po_info.status_polish = PO_NOT_AVAILABLE */
    double polish_delta;                /* OSQP `delta` (1e-6) */
    int    polish_refine_iter;          /* OSQP `polish_refine_iter` (3) */
    /* Refinement (extension, off by default; runs after a path is solved and before the polish).  refine = 2: semismooth Newton on the augmented Lagrangian
     *     phi_y(x) = 1/2 x'Px + sum_i rho_i / 2 dist^2(a_i x + y_i / rho_i, [l_i, u_i])
     * (rho_i = refine_newton_rho on inequality rows, refine_newton_rho_eq on equality rows, scaled problem) with a line search on the piecewise-quadratic merit (safeguarded
     * Newton on its piecewise-linear derivative, run to refine_ls_tol), and a multiplier update  y <- rho (w - clip(w))  whenever the inner problem is solved to the dual
     * tolerance.  Each Newton step is ONE factorisation of the same block-tridiagonal matrix the ADMM iteration uses, with the rows outside their bounds at rho_i and the
     * others at OSQP's RHO_MIN, one solve, and a few row passes for the line search; the method is monotone in phi and terminates finitely.  It runs from the point the
     * OSQP-faithful ADMM iteration stops at (refine_rounds: how early) until OSQP's termination test holds at refine_eps (po_info.status_refine = 1: certified).
     * po_info.iters counts a Newton step as one iteration.  On BASELINE config 3 every path ends within 1e-4 m of the exact optimum (39 % at eps 1e-4 without).
     * Values: 0 off (the library default: OSQP-faithful), 2 on.  (ABI 5 removed refine = 1 — the activity-weighted ADMM continuation of rounds 2 - 3 —, its knobs
     * refine_every / refine_max_iter / refine_max_refactor / refine_rho / refine_adapt, the chained-rounds scheduling refine_chain 0 / 1 with refine_speculate, probe_iters
     * and polish_passes: the Newton phase in its own launches is faster on every shape, profiles/README.md; po_create returns PO_ERR_INVALID for refine = 1.) */
    int    refine;
    double refine_eps;                  /* 1e-7: eps_abs = eps_rel of the termination test of this phase (what po_info.status_refine = 1 certifies).  Measured on all 4096
                                           paths of BASELINE config 3 against their exact optima: certified at 1e-7 -> max e_y RMS error 1e-5 m; at 1e-6 seven certified
                                           paths were 1e-4 .. 4e-4 m away (these QPs are flat: e_y is the double integral of the curvature) */
    int    refine_rounds;               /* 1.  R > 1: the solve first stops at 10^(R-1) x (eps_abs, eps_rel) and is refined from there; a path the refinement does not
                                           certify at refine_eps goes back to the type-based iteration at a 10 x tighter eps and is refined again, down to eps itself.
                                           Every path returned satisfies OSQP's test at eps_abs / eps_rel or at refine_eps.  The headline setting uses 5: the ADMM
                                           iteration is then only a 25-iteration warm start (its first termination check passes at 1e4 x eps) */
    int    refine_chain;                /* 2 (default) or 3; scheduling only, results bit-identical.  The solve is three launches on one stream: the plain solve kernels as the
                                           warm start, the Newton refinement of round 0 as a launch of its own, and a fallback launch that takes the paths it did not
                                           certify (rare) through their later rounds.  2: the fallback launch is issued only when a path needs it — the engine reads a
                                           4-byte count back, so the device-pointer entry returns when the Newton launch has FINISHED (it blocks; the launch it saves
                                           costs 0.5 ms; a stream that is being captured is detected and treated as 3); 3: always issued, fully asynchronous. */
    int    refine_extra_rounds;         /* 0.  E > 0: a path that the last regular round does not certify at refine_eps continues BELOW eps — type-based iteration at
                                           eps / 10, refinement again, eps / 100, ... — for up to E more rounds.  A path returned after them is certified, or satisfies
                                           OSQP's test at eps / 10^E — or ran out of max_iter in one of these rounds: then the point it ends on is tested against OSQP's
                                           criteria at the caller's eps once more — passes: that point, PO_STATUS_SOLVED with status_refine -1; fails (ADMM residuals are
                                           not monotone): the point that round STARTED from (it met eps in the round before), PO_STATUS_SOLVED with status_refine -1.  A
                                           round below eps never turns a solved path into MAX_ITER. */
    double refine_newton_rho;           /* 100 (scaled problem): penalty of the inequality rows at the start of an attempt; it grows 10 x whenever a multiplier update does not cut the
                                           primal residual by 4.  Measured on the whole BASELINE batches (oracle): 100 needs the fewest steps AND has the shortest tail (config 3:
                                           mean 14.8 / max 40 Newton steps; 1e3: 16.1 / 52; 1e4: 18.4 / 240; 30: 16.3 / 40 but 4 of 512 narrow-corridor paths uncertified) */
    double refine_newton_rho_eq;        /* 1e4: penalty of the equality rows at the start of an attempt (NOT 1e3 x the inequality one as in OSQP's step vector: the merit's
                                           gradient carries rho_eq x (a.x - b), a difference of O(1) numbers, whose rounding at rho_eq >= 1e6 sits near the dual tolerance);
                                           raised only for a path whose multiplier updates stall, see refine_newton_rho_eq_max */
    double refine_newton_rho_max;       /* 1e5: a multiplier update that does not cut the primal residual by 4 raises the penalty 10 x, up to this (the slow
                                           case: active rows that are nearly dependent through the heavily weighted curvature-rate variables).  <= 0: the default;
                                           at most OSQP's RHO_MAX = 1e6, which also bounds what refine_newton_escalate raises it to */
    double refine_ls_tol;               /* 0.6: the line search stops at |psi'(t)| <= tol |psi'(0)| (the search is a safeguarded Newton iteration on the piecewise-linear psi', so
                                           what it accepts is close to the root anyway).  Measured on the whole BASELINE batches, every path certified at every setting: 1e-4 -> 0.3
                                           3 - 7 % fewer Newton steps and 14 % fewer evaluations per step (round 4); 0.3 -> 0.6 another 1 % fewer steps, the first trial step
                                           accepted more often (config 3 5.19 -> 4.98 ms, config 5 21.3 -> 19.6 ms) and the SAME largest distance from the exact optimum on every
                                           golden batch (2.1e-5 m); 0.9 is 3 % faster again but lands 3 paths of config 5 (KPC, flat directions) at 5 - 6.5e-5 m, half the margin
                                           to the 1e-4 m bar (round 5).  The final correction steps always search to 1e-4. */
    int    refine_ls_max;               /* 30: evaluations of psi' per line search at most */
    int    refine_newton_max;           /* 300: Newton steps per attempt (every round).  BASELINE config 3: mean 16, max 60; config 5 (KPC): mean 28, max ~210 */
    int    refine_newton_final;         /* 3: once the point is certified at refine_eps, Newton steps go on (tight line search, no multiplier update) until the dual residual — the
                                           gradient of the merit — sits 1e3 x below its tolerance, at most this many (none when it already does at certification).  Newton converges
                                           quadratically once the active set is right, so this is normally ONE step to the rounding floor: the accuracy a tighter refine_eps would
                                           buy, without asking the termination test for tolerances below what fp64 delivers on the KPC rows weighted 1e5 (measured: refine_eps 3e-9
                                           leaves 93 of 4096 KPC paths stalling at r_dual 5e-9 until the step budget is spent).  Exactly one step is not enough: a step that changes
                                           the active set can land with a LARGER dual residual than the certified point's (2 of 4096 KPC paths ended 1.06e-4 m from their optimum
                                           that way; the offset e_y carries no cost of its own, so a gradient of 1e-7 is 1e-4 m there).  0: stop at the certified point. */
    int    refine_newton_escalate;      /* 12: from this many multiplier updates of an attempt on, refine_newton_rho_max and refine_newton_rho_eq_max stand 10 x higher, from twice as
                                           many on 100 x.  For DEGENERATE optima (linearly dependent active rows — corridors narrower than the soft margin produce them): the
                                           multipliers are not unique, the method of multipliers converges sublinearly at any fixed penalty (measured: the primal residual falls
                                           0.4 % per update at 1.2 x its tolerance), and only a larger penalty shortens it.  On the narrow-corridor batch of `host_test bench`: 2 of
                                           4096 paths uncertified and 15 - 18 through the fall-back rounds (12 - 27 ms) without it, none with it; no BASELINE path gets that far.  0: off. */
    double refine_newton_rho_eq_max;    /* 1e6: once the inequality penalty sits at refine_newton_rho_max and a multiplier update still does not cut the primal residual by 4,
                                           the EQUALITY rows' penalty grows 10 x instead, up to this.  The case: the primal residual left on the dynamics rows, whose multipliers
                                           converge at H / (H + rho_eq) per update when the active inequality rows beside them carry 10 x their penalty (wide corridors with a
                                           large initial offset: 24 of 4096 paths of `host_test bench` ran out of updates uncertified without it; config 5: hardest path 222 -> 129).
                                           0: the equality penalty never grows; < 0: the default; at most 1e8, which also bounds what refine_newton_escalate raises it to */
} po_params;

#define PO_NOT_AVAILABLE (-2) /* po_info.status_refine / status_polish: asked for, but the batch's shape has no kernel for it (see the fields) */
typedef struct po_info {
    int    status;      /* PO_STATUS_*                                  */
    int    iters;       /* ADMM iterations run (with po_params.refine: the solve's and the refinement's together) */
    int    n_refactor;  /* numeric refactorisations after the first (the refinement's included) */
    int    status_polish; /* OSQP info.status_polish: 0 not attempted (polish off, or the path was not solved), 1 polished solution adopted, -1 polish unsuccessful
                             (ADMM solution kept); PO_NOT_AVAILABLE (-2): po_params.polish was set but the SHAPE of this batch has no polish kernel — the role-split
                             mapping (KP, keep_control_steps_ 6 .. 16) and the single-level mapping (see status_refine) — the solve itself is unaffected */
    double r_prim;      /* ||Ax - z||_inf   at exit (unscaled)          */
    double r_dual;      /* ||Px + q + A'y||_inf at exit                 */
    double rho;         /* final rho                                    */
    double obj;         /* 0.5 x'Px at exit                             */
    int    status_refine; /* po_params.refine: 0 the refinement did not run on this path (refine off, or the path was not solved); 1 CERTIFIED: OSQP's
                             termination test holds ON THE RETURNED POINT at refine_eps (default 1e-7; re-evaluated after every step, the final correction steps included:
                             ABI 5), i.e. the point is the QP's optimum to that tolerance; -1 the
                             refinement ran out of its budget before that: the returned point satisfies OSQP's test at eps_abs / eps_rel only (it is the refined
                             point when its residuals are no worse than the solved point's, else the solved point) — a caller that needs the <= 1e-4 m accuracy
                             clause per path treats -1 as "not certified";
                             PO_NOT_AVAILABLE (-2, ABI 6): po_params.refine = 2 was set but the SHAPE of this batch has no Newton kernel, so the call ran the plain solve at
                             eps_abs / eps_rel on every path (status is what that solve reports).  The shapes WITH one — every case the reference's own pipeline produces
                             (keep_control_steps_ 1 .. 8 from 0.15 .. 1.0 m spacing, path_optimizer.cpp:171-172; the reference itself accepts any horizon_,
                             solver_kp_as_input.cpp:13-24): KP keep 1 .. 4 up to N = 512, keep 5 up to N = 320, keep 6 .. 8 up to N = 64 keep, keep 9 .. 16 up to
                             N = 32 keep; KPC (keep 4) and K up to N = 512; in every case only while the two-level tables fit 160 KB of LDS.  Without one (the single-level
                             chain): keep > 16, and longer paths than those limits. */
    int    reserved;
} po_info;

/* One homogeneous batch: B independent paths, each with N points and the same `keep`
 * (keep_control_steps_, solver_kp_as_input.cpp:17; ignored for PO_K; must be 4 for PO_KPC,
 * solver_kp_as_input_constrained.cpp:17).  All arrays are row-major, path-major.
 * Replaces the reads of ReferencePath::{getReferenceStates,getBounds,getMaxKList,getMaxKpList}
 * (include/path_optimizer/data_struct/reference_path.hpp:34-37) and
 * VehicleState::{getInitError,getStartState,getEndState}
 * (include/path_optimizer/data_struct/vehicle_state_frenet.hpp:18-23). */
typedef struct po_batch_in {
    int formulation, B, N, keep;
    const double *ref_x, *ref_y, *ref_z, *ref_k, *ref_s; /* [B][N]  State.{x,y,z,k,s}              */
    const double *bounds;   /* [B][N][4][2]: circles c0..c3 x {lb (right, <=0), ub (left, >=0)}    */
    const double *x0;       /* [B][3]: init_offset, init_heading_error, start_state.k               */
    const double *goal_z;   /* [B]: end_state.z                                                     */
    const double *max_k;    /* [B][N] KPC only (else NULL)                                          */
    const double *max_kp;   /* [B][N] KPC only; indexed by CONTROL id like the reference (:179-183) */
    const int    *n_points; /* optional [B]: points of each path, 2 <= n_points[b] <= N (ragged batch); NULL = all N.
                               Every array keeps its stride N; outputs beyond n_points[b] are zero.              */
    const int    *order;    /* optional [B]: a PERMUTATION of 0..B-1; workgroup i solves path order[i] (workgroups start in index order).  A scheduling
                               hint only — results do not depend on it: a launch ends with its longest path, so a caller that re-solves similar
                               problems every planning cycle passes the paths sorted by the previous cycle's po_info.iters, longest first.
                               NULL = the engine's own XCD-aware mixing of the path order. */
} po_batch_in;

typedef struct po_batch_out {
    double  *states; /* [B][N][5]: x, y, heading, k, s  (State.v = State.a = 0 in the reference)   */
    po_info *info;   /* [B]                                                                         */
    double  *x;      /* optional [B][n]: raw QP solution in the REFERENCE variable order, or NULL   */
} po_batch_out;

/* Obstacle-distance map: what the reference reads through PathOptimizationNS::Map::getObstacleDistance / isInside
 * (src/tools/Map.cpp:16-26), i.e. layer "distance" of a grid_map::GridMap sampled with
 * atPosition(INTER_LINEAR).  grid_map is a third-party dependency that is NOT in /root/reference (ROS package
 * grid_map_core, un-pinned): its geometry conventions are restated in csrc/po_map.hpp / oracle/po_oracle.c:
 * `distance` is the layer's Eigen::MatrixXf, column-major, size_x rows (along x) by size_y columns (along y);
 * cell (0,0) is the corner of LARGEST x and y; cell (i,j) is centred at
 *   pos + 0.5*len - (idx + 0.5)*resolution,  len = size*resolution;  the circular-buffer start index is (0,0). */
typedef struct po_map {
    const float *distance;  /* [size_y][size_x] in memory (column-major), metres to the nearest obstacle */
    int    size_x, size_y;
    double resolution;      /* metres per cell */
    double pos_x, pos_y;    /* position of the map centre in the world frame */
} po_map;

typedef struct po_handle_s *po_handle;

/* Fill `p` with the reference defaults (planning_flags.cpp) and the project's ADMM settings. */
void po_default_params(po_params *p);

/* QP dimensions exactly as the reference constructors compute them
 * (solver_kp_as_input.cpp:13-24, solver_kp_as_input_constrained.cpp:13-24, solver_k_as_input.cpp:14-20). */
int po_problem_dims(int formulation, int N, int keep, int *n, int *m, int *C);

/* keep_control_steps_ from the first <=9 arc-length gaps, with the reference's truncation
 * (solver.cpp:22-27 + solver_kp_as_input.cpp:17). Returns keep (>=1) or PO_ERR_INVALID. */
int po_keep_control_steps(int formulation, const double *ref_s, int N);

/* Number of HIP devices visible to the process (0 when there is none, or no driver): one handle per device is how a batch is spread over the GPUs of a
 * node (host/include/path_optimizer_amd/solver.hpp: OsqpSolver::solveBatch over several PoEngine). */
int po_device_count(void);
/* One handle = one HIP device + one stream; calls on a handle are serialised; distinct handles (same device or not) may be used concurrently
 * from different host threads. */
int po_create(int device, const po_params *params, po_handle *out);
int po_destroy(po_handle h);
/* Use an existing hipStream_t (e.g. torch's current stream); NULL = the handle's own stream. */
int po_set_stream(po_handle h, void *hip_stream);
/* Developer switches for A/B measurements and tests (the library reads NO environment variable; results never depend on these).  Keys: "identity_order" (workgroup i
 * solves path i instead of the XCD-aware mixing), "debug_cycles" (per-phase shader clocks of path 0 on stderr; makes the solve entry synchronous), "host_threads",
 * "smooth_seq", "smooth_waves", "smooth_nopad", "smooth_debug" (smoothing-QP engine variants), "dp_one_wave" (DP lattice search on one wave per instance whatever
 * the batch size), "newton_slice" (refine = 2: Newton steps of the FIRST of the two Newton launches, default 8 — every path runs that many, what is unfinished is parked
 * and the second launch takes the parked paths in order of expected remaining work, longest first; 0: one launch; left alone the engine slices batches of
 * at least two rounds of the device's wave slots (2 x 4 x CUs = 2048 paths) on shapes that run one wave per path, a value set here applies to every batch.  Scheduling only: the same operations in the same
 * order — statuses and certificates do not depend on it, the solutions agree to round-off (the kernels of the two launches are compiled separately; identical bit for bit on the
 * BASELINE KP / K shapes, <= 1e-10 elsewhere); BASELINE config 3: 4.76 -> 4.06 ms).  Unknown key: PO_ERR_INVALID. */
int po_debug_set(po_handle h, const char *key, int value);
/* Developer read-back (synchronises the stream): "fallback_paths" = how many paths the Newton launch of the last solve with refine = 2 did not certify and handed to
 * the fallback launch; "newton_parked" = how many went on into the second of the sliced Newton launches (-1: the last solve was not sliced). */
int po_debug_get(po_handle h, const char *key, long long *value);

/* Host-pointer entry: H2D, solve, D2H, synchronous. */
int po_solve_batch(po_handle h, const po_batch_in *in, const po_batch_out *out);
/* Device-pointer entry: all pointers in `in`/`out` are device pointers; asynchronous on the stream. */
int po_solve_batch_device(po_handle h, const po_batch_in *in, const po_batch_out *out);
/* ---- post-solve step: PathOptimizer::optimizePath, src/path_optimizer/path_optimizer.cpp:183-200 ----
 * Upload the obstacle-distance layer (host pointer in `map->distance`) to the handle's device; kept until replaced. */
int po_set_map(po_handle h, const po_map *map);
/* For every path: walk the optimised states in order and stop at the first state that fails
 * CollisionChecker::isSingleStateCollisionFreeImproved (src/tools/collision_checker.cpp:42-59: bounding circle, then the
 * six footprint circles of src/tools/car_geometry.cpp:38-72; outside the map = collision).
 *   n_valid[b] = number of states kept (the reference erases from the colliding state on);
 *   ok[b]      = what optimizePath returns: 0 if the QP was not solved (info[b].status != PO_STATUS_SOLVED),
 *                1 if no state collides, else (s of the last kept state >= 20 m); 0 if the first state collides.
 * `states`/`info` are the outputs of po_solve_batch* ([B][N][5], [B]); n_points as in po_batch_in (or NULL).
 * With params.enable_collision_check == 0 every solved path is kept whole.  Host-pointer and device-pointer entries. */
int po_postcheck_batch(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info,
                       int *n_valid, int *ok);
int po_postcheck_batch_device(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info,
                              int *n_valid, int *ok);
/* The other output branch of optimizePath (FLAGS_enable_raw_output = false, path_optimizer.cpp:201-226): x(s), y(s) splines through the
 * solved states, sampled every FLAGS_output_spacing with heading and curvature from the spline, each sample collision-checked; the walk
 * stops at the first colliding sample.  out_states [B][M][5]; n_out[b] samples kept; ok[b] as po_postcheck_batch (0 for an unsolved QP;
 * the reference dereferences back() of an empty vector when the very first sample collides: reported as ok = 0, n_out = 0);
 * n_out[b] = -2 when M is too small. */
int po_densify_batch(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info, int M, double *out_states, int *n_out,
                     int *ok);
int po_densify_batch_device(po_handle h, int B, int N, const int *n_points, const double *states, const po_info *info, int M, double *out_states,
                            int *n_out, int *ok);
/* ---- corridor-bounds producer: ReferencePath::updateBounds -> ReferencePathImpl::updateBoundsImproved,
 * src/data_struct/reference_path_impl.cpp:142-201 (+ getApproxState :121-140, getClearanceWithDirectionStrict :283-472 with
 * FLAGS_enable_simple_boundary_decision = true as shipped, tk::spline src/tools/spline.cpp).  Needs po_set_map.
 * FLAGS_enable_simple_boundary_decision (src/config/planning_flags.cpp:84): no po_params field, because it has no effect in the reference either — the branch it
 * selects in getClearanceWithDirectionStrict (:322) also requires is_original_spline_set, and ReferencePath::setOriginalSpline (reference_path.cpp:95) has no caller
 * anywhere in the reference: the reference-compiled producer returns bit-identical bounds for both values (tests/test_bounds.py).  Only a caller that patches the
 * reference to CALL setOriginalSpline and clears the flag gets bounds this entry does not produce; such a caller keeps its CPU producer and hands po_batch_in.bounds over.
 * Inputs per path: the N reference states (x, y, heading, s) and the K knots (s, x, y) the path's x(s) / y(s) splines were set
 * from (tk::spline::set_points, natural boundary conditions — what ReferencePath::setSpline receives).
 * Outputs: bounds [B][N][4][2] = (lb, ub) of the four covering circles, directly consumable as po_batch_in.bounds, and
 * n_valid[b] = number of states kept (the reference stops at the first blocked state and truncates the reference there);
 * rows >= n_valid[b] are zero.  n_valid is directly consumable as po_batch_in.n_points (when >= 2). */
typedef struct po_bounds_in {
    int B, N, K;                                   /* paths, reference states per path (stride), knots per path (stride) */
    const double *ref_x, *ref_y, *ref_z, *ref_s;   /* [B][N] */
    const int    *n_points;                        /* optional [B]: states of each path (<= N) */
    const double *knot_s, *knot_x, *knot_y;        /* [B][K], knot_s strictly increasing */
    const int    *n_knots;                         /* optional [B]: knots of each path (3 <= n_knots[b] <= K) */
} po_bounds_in;
int po_bounds_batch(po_handle h, const po_bounds_in *in, double *bounds, int *n_valid);        /* host pointers, synchronous  */
int po_bounds_batch_device(po_handle h, const po_bounds_in *in, double *bounds, int *n_valid); /* device pointers, on the stream */

/* ---- reference-smoothing QPs: the three other places where the reference hands a chain-structured QP to
 * OsqpEigen with the same call pattern as the hot path (setHessianMatrix ... initSolver, solve):
 *   PO_SMOOTH_TENSION2  TensionSmoother2::osqpSmooth       src/reference_path_smoother/tension_smoother_2.cpp:163-301
 *                       (FLAGS_smoothing_method = "TENSION2", FLAGS_tension_solver = "OSQP": the shipped defaults)
 *   PO_SMOOTH_TENSION   TensionSmoother::osqpSmooth        src/reference_path_smoother/tension_smoother.cpp:186-314  (needs po_set_map)
 *   PO_SMOOTH_POST      ReferencePathSmoother::postSmooth  src/reference_path_smoother/reference_path_smoother.cpp:534-644 (the QP; the
 *                       re-projection onto the spline that follows it stays with the caller)
 * One batch = B independent instances with up to P points each (arrays [B][P], ragged through n_points).
 *   TENSION2 / TENSION inputs: x, y, angle, k, s = the five lists segmentRawReference produces (reference_path_smoother.cpp:50-91);
 *       outputs: out_x, out_y, out_s = result_x_list, result_y_list, result_s_list (s = running chord length of the result).
 *   POST inputs: s = layers_s_list_, lb / ub = layers_bounds_ (.first / .second), l0[b] = vehicle_l_wrt_smoothed_ref_;
 *       x, y, angle, k unused (may be NULL); output: out_x[b][i] = QPSolution(i), the lateral offset of layer i; out_y / out_s unused.
 * info[b].status == PO_STATUS_SOLVED <=> the reference's `solver.solve()` returned true; raw (optional) = the whole QP solution in
 * the REFERENCE variable order, [B][n_max] with n_max = po_smooth_dims(kind, P).  OSQP settings come from the handle's po_params
 * (the reference runs the smoothers at OSQP's default eps_abs = eps_rel = 1e-3; create the handle accordingly to mirror that).
 * Fewer points than the reference accepts (TENSION*: 3, POST: 4, reference_path_smoother.cpp:536) -> PO_ERR_INVALID. */
enum { PO_SMOOTH_TENSION2 = 0, PO_SMOOTH_TENSION = 1, PO_SMOOTH_POST = 2 };
typedef struct po_smooth_in {
    int kind, B, P;
    const int    *n_points;            /* optional [B] */
    const double *x, *y, *angle, *k;   /* [B][P] */
    const double *s;                   /* [B][P] */
    const double *lb, *ub;             /* [B][P]  POST only */
    const double *l0;                  /* [B]     POST only */
} po_smooth_in;
typedef struct po_smooth_out {
    double  *x, *y, *s;   /* [B][P] (y, s may be NULL for POST) */
    po_info *info;        /* [B] */
    double  *raw;         /* optional [B][n_max] */
} po_smooth_out;
/* QP dimensions as the reference sets them (tension_smoother_2.cpp:177-178, tension_smoother.cpp:201-202, reference_path_smoother.cpp:544-545) */
int po_smooth_dims(int kind, int P, int *n, int *m);
int po_smooth_batch(po_handle h, const po_smooth_in *in, const po_smooth_out *out);        /* host pointers, synchronous  */
int po_smooth_batch_device(po_handle h, const po_smooth_in *in, const po_smooth_out *out); /* device pointers, on the stream */

/* ---- reference re-sampling, limits and the DP lattice search ----
 * A batch of planar splines x(s), y(s) (tk::spline, natural boundary conditions) given by the knots they were set from, plus the arc
 * length the reference attaches to them (ReferencePath::getLength(), i.e. max_s_). */
typedef struct po_spline_in {
    int B, K;                                /* instances, knots per instance (stride) */
    const double *knot_s, *knot_x, *knot_y;  /* [B][K], knot_s strictly increasing */
    const int    *n_knots;                   /* optional [B]: 3 <= n_knots[b] <= K */
    const double *length;                    /* [B]: max_s_ */
} po_spline_in;
/* ReferencePathImpl::buildReferenceFromSpline (src/data_struct/reference_path_impl.cpp:474-499): walk the spline from s = 0 while
 * s <= length with the curvature-adaptive step (FLAGS_enable_dynamic_segmentation; |k| >= 0.2 -> delta_s_smaller, |k| <= 0.08 ->
 * delta_s_larger, linear in between) and emit State{x, y, heading, k, s} (getHeading / getCurvature, src/tools/tools.cpp:34-47).
 * Outputs [B][N] (directly consumable as po_batch_in.ref_* / po_bounds_in.ref_*) and n_points[b]; n_points[b] = -1 when the reference
 * returns false (length <= 0) and -2 when more than N states would be produced (raise N); rows beyond n_points are zero. */
int po_resample_batch(po_handle h, const po_spline_in *in, double delta_s_smaller, double delta_s_larger, int N, double *ref_x, double *ref_y,
                      double *ref_z, double *ref_k, double *ref_s, int *n_points);
int po_resample_batch_device(po_handle h, const po_spline_in *in, double delta_s_smaller, double delta_s_larger, int N, double *ref_x,
                             double *ref_y, double *ref_z, double *ref_k, double *ref_s, int *n_points);
/* ReferencePathImpl::updateLimits (reference_path_impl.cpp:203-235), the branch the KPC formulation uses (reference states given
 * directly, with speed v and acceleration a): max_k = sqrt((mu g)^2 - a^2) / v^2, max_kp = max_curvature_rate / v, DBL_MAX for
 * v <= 1e-4.  Outputs [B][N] are directly consumable as po_batch_in.max_k / max_kp. */
int po_limits_batch(po_handle h, int B, int N, const int *n_points, const double *v, const double *a, double *max_k, double *max_kp);
int po_limits_batch_device(po_handle h, int B, int N, const int *n_points, const double *v, const double *a, double *max_k, double *max_kp);
/* ReferencePathSmoother::graphSearchDp (src/reference_path_smoother/reference_path_smoother.cpp:147-300, cost :110-145): project the
 * vehicle onto the spline (findClosestPoint, tools.cpp:71-112), lay out layers every FLAGS_search_longitudial_spacing, sample each at
 * lateral offsets -range..range every FLAGS_search_lateral_spacing against the obstacle map (po_set_map), run the layer-by-layer
 * min-cost recursion and walk back from the cheapest node of the last reachable layer widening each node's rough corridor in 0.2 m
 * steps.  start [B][3] = start_state (x, y, heading).  Outputs, [B][L]: layer_s = layers_s_list_, lb / ub = layers_bounds_
 * (.first / .second); l0[b] = vehicle_l_wrt_smoothed_ref_; n_layers[b] = layers kept, -1 when the reference returns false (vehicle
 * further than the lateral range from the spline), -2 when more than L layers are needed.  These are the inputs of PO_SMOOTH_POST. */
int po_dp_search_batch(po_handle h, const po_spline_in *in, const double *start, int L, double *layer_s, double *lb, double *ub, double *l0,
                       int *n_layers);
int po_dp_search_batch_device(po_handle h, const po_spline_in *in, const double *start, int L, double *layer_s, double *lb, double *ub,
                              double *l0, int *n_layers);

/* ---- the remaining glue stages of PathOptimizer::solve, device-pointer entries (each is one kernel on the handle's stream) ----
 * ReferencePathSmoother::bSpline (reference_path_smoother.cpp:495-532): dense samples x_list_, y_list_, s_list_ [B][M] of the clamped
 * B-spline whose control points are the input points (tinyspline, a library that is NOT in /root/reference: restated, parity unpinned);
 * n_samples[b] = -1 for fewer than 4 points ("Few reference points."), -2 if M is too small. */
int po_bspline_batch_device(po_handle h, int B, int W, const int *n_way, const double *way_x, const double *way_y, int M, double *x, double *y, double *s,
                            int *n_samples);
/* ReferencePathSmoother::segmentRawReference (:50-91): the spline through the dense lists (raw->knot_*, raw->n_knots; raw->length unused)
 * sampled at 1 m stations: x, y, s, angle, k [B][P] = the five inputs of PO_SMOOTH_TENSION2 / PO_SMOOTH_TENSION; n_points as above. */
int po_segment_raw_batch_device(po_handle h, const po_spline_in *raw, int P, double *x, double *y, double *s, double *angle, double *k, int *n_points);
/* The tail of ReferencePathSmoother::postSmooth (:568-590): layer i moves to xs(s_i) + l_i (cos, sin)(heading + pi/2); x, y, s [B][L] are the
 * knots of the re-fitted spline (s = running chord length), length[b] = s.back() (may be NULL). */
int po_post_project_batch_device(po_handle h, const po_spline_in *spline, int L, const int *n_layers, const double *layer_s, const double *offsets, double *x,
                                 double *y, double *s, double *length);
/* PathOptimizer::segmentSmoothedPath up to the re-sampling (path_optimizer.cpp:119-169): init [B][3] = initial_offset,
 * initial_heading_error, length after the goal trim; ok[b] = 0 where the reference returns false (empty path, heading error > 75 deg).
 * start [B][start_stride] = x, y, heading, ...; goal [B][goal_stride] = x, y, ... */
int po_segment_init_batch_device(po_handle h, const po_spline_in *spline, const double *start, int start_stride, const double *goal, int goal_stride,
                                 double *init, int *ok);

/* ---- PathOptimizer::solve for a batch of planning instances (path_optimizer.cpp:40-85): bSpline -> TensionSmoother2 -> graphSearchDp ->
 * postSmooth -> segmentSmoothedPath (goal trim, buildReferenceFromSpline(0.15, FLAGS_output_spacing = 0.3), updateBounds) -> KP QP ->
 * collision check, every stage on the device, chained through HBM (one D2H of 3 ints per instance in the middle to group the QPs by
 * keep_control_steps_).  Needs po_set_map.  The shipped flag defaults are assumed (smoothing_method TENSION2, tension_solver OSQP,
 * optimization_method KP, enable_raw_output, enable_collision_check from po_params); OSQP settings from po_params for all three QPs.
 *   way_x / way_y [B][W]: reference_points (solve() reads only x and y); start [B][4] = start_state x, y, heading, k; goal [B][3].
 *   states [B][N][5] = final_path (x, y, heading, k, s), n_states[b] its size, ok[b] = the bool solve() returns;
 *   stage[b] (optional) = 0 or the stage that made it return false: 1 too few points / B-spline, 2 tension smoothing QP, 3 graph search,
 *   4 post smoothing, 5 segmentation (heading error > 75 deg), 6 reference blocked at its start, 7 path QP, 8 collision check (states
 *   hold the truncated path), 9 capacity (N / max_length too small or the QP does not fit the on-chip tile).
 *   max_length: upper bound of the waypoint polyline lengths (sizes the intermediate buffers); <= 0: computed from the waypoints
 *   (host-pointer entry only).
 *   Limits (the reference has none): max_length up to about 540 m (the spline stages keep 120 x knots bytes in LDS, knots = ceil(max_length) + 6 <= 546;
 *   beyond that the call returns PO_ERR_UNSUPPORTED for the whole batch); the path QP of an instance must fit the on-chip tile — at the default 0.15 ... 0.3 m
 *   re-sampling that is a route of roughly 75 m (keep_control_steps_ <= 4: N <= 512) to 150 m; longer instances come back with stage 9, the others are unaffected. */
typedef struct po_plan_in {
    int B, W;
    const int    *n_way;          /* optional [B] */
    const double *way_x, *way_y;  /* [B][W] */
    const double *start, *goal;   /* [B][4], [B][3] */
    double max_length;
    int N;                        /* rows of `states` per instance */
} po_plan_in;
typedef struct po_plan_out {
    double  *states;   /* [B][N][5] */
    int     *n_states; /* [B] */
    int     *ok;       /* [B] */
    int     *stage;    /* optional [B] */
    po_info *info;     /* optional [B]: the path QP */
} po_plan_out;
int po_plan_batch(po_handle h, const po_plan_in *in, const po_plan_out *out);         /* host pointers, synchronous */
int po_plan_batch_device(po_handle h, const po_plan_in *in, const po_plan_out *out);  /* device pointers; synchronises the stream once mid-way */

/* Test/diagnostic entry: Map::getObstacleDistance at `n` world positions xy[n][2] (host pointers); inside[n] = Map::isInside. */
int po_map_sample(po_handle h, int n, const double *xy, double *dist, int *inside);

/* Test/diagnostic entry: run only the device assembly and return the QP data in the REFERENCE
 * row order: l,u [B][m]; dyn [B][N-1][3] = per-transition data-dependent A entries
 * (KP/KPC: ds, -k^2*ds, ds ; K: -ds*k^2, ds, ds/L/cos^2) ; host pointers. */
int po_assemble_batch(po_handle h, const po_batch_in *in, double *l, double *u, double *dyn);

/* Test/diagnostic entry: the per-path equilibration block [B][64] the solve kernel consumes
 * (layout in csrc/po_scale.hpp: W = E^2/c per row class, E, sigma/(c D^2) and c*D per variable class, c). */
int po_scaling_batch(po_handle h, const po_batch_in *in, double *out);

/* Kernel time (ms) of the last timed launch sequence on this handle, measured with hipEvents on the handle's stream (valid after the
 * stream has been synchronised): po_solve_batch* (equilibration + solve launches + polish) or po_smooth_batch*.  After po_plan_batch*, which
 * issues several of them, it is the LAST such sequence of the chain (the path QP of the last keep-group), not the whole call. */
int po_last_kernel_ms(po_handle h, float *ms);
/* Where the last solve spent its time, ms8[8] (hipEvents on the handle's stream + host wall clock; valid after the call returned / the stream was synchronised):
 *   after po_solve_batch (host pointers: the caller's arrays are packed into a pinned staging block on several host threads while the slices already packed travel
 *   over PCIe, one D2H into a pinned block, threaded unpack): [0] pack + H2D, [1] the solve (= po_last_kernel_ms), [2] D2H, [3] host pack alone, [4] host unpack;
 *   with po_params.refine = 2 (either entry): [5] equilibration + warm-start launches, [6] the Newton launch, [7] what follows it: the fallback launch when a path needs it (refine_chain = 3: always), status sweep. */
int po_last_phase_ms(po_handle h, float *ms8);

const char *po_strerror(int code);
const char *po_last_hip_error(void);
/* "po_hip <abi> (gfx950)"; PO_ABI_VERSION is bumped whenever a struct layout, an entry point or the meaning of a field changes (5: round 5, see po_params.refine;
 * 6: round 6, po_info.status_refine / status_polish may be PO_NOT_AVAILABLE; po_create refuses refine_rounds + refine_extra_rounds >= 32).  A binding should compare
 * the number in po_version() with the PO_ABI_VERSION it was written against before it passes a struct (path_optimizer_amd/binding.py does). */
#define PO_ABI_VERSION 6
const char *po_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PO_HIP_H_ */
