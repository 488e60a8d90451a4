"""ctypes mirror of include/po_hip.h (struct layouts and enums only; no behaviour)."""
PO_ABI_VERSION = 6  # include/po_hip.h
PO_NOT_AVAILABLE = -2  # po_info.status_refine / status_polish: asked for, no kernel for this shape
import ctypes as C

PO_KP, PO_KPC, PO_K = 0, 1, 2
PO_OK, PO_ERR_INVALID, PO_ERR_HIP, PO_ERR_UNSUPPORTED, PO_ERR_NOMEM = 0, -1, -2, -3, -4
PO_STATUS_SOLVED, PO_STATUS_MAX_ITER = 1, -2
PO_STATUS_PRIMAL_INFEASIBLE, PO_STATUS_DUAL_INFEASIBLE, PO_STATUS_UNSOLVED = -3, -4, -10
PO_STATUS_NON_FINITE = -8
FORM_BY_NAME = {"KP": PO_KP, "KPC": PO_KPC, "K": PO_K}  # OsqpSolver::create strings (solver.cpp:34-38)

_dp = C.POINTER(C.c_double)


class PoParams(C.Structure):
    _fields_ = [
        ("d", C.c_double * 4),
        ("w_curv", C.c_double), ("w_curv_rate", C.c_double), ("w_dev", C.c_double), ("w_slack", C.c_double),
        ("k_w_curv", C.c_double), ("k_w_curv_rate", C.c_double), ("k_w_dev", C.c_double),
        ("w_k_slack", C.c_double), ("w_kp_slack", C.c_double),
        ("margin", C.c_double), ("max_steer", C.c_double), ("wheel_base", C.c_double),
        ("constraint_end_heading", C.c_int), ("scaling", C.c_int),
        ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
        ("rho0", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("adapt_tol", C.c_double),
        ("max_iter", C.c_int), ("check_every", C.c_int), ("adapt_every", C.c_int), ("enable_collision_check", C.c_int),
        ("car_width", C.c_double), ("car_length", C.c_double), ("rear_axle_to_center", C.c_double), ("safety_margin", C.c_double),
        ("t2_w_dev", C.c_double), ("t2_w_curv", C.c_double), ("t2_w_curv_rate", C.c_double),
        ("cart_w_curv", C.c_double), ("cart_w_curv_rate", C.c_double), ("cart_w_dev", C.c_double),
        ("mu", C.c_double), ("max_curvature_rate", C.c_double), ("search_lateral_range", C.c_double),
        ("search_long_spacing", C.c_double), ("search_lat_spacing", C.c_double), ("enable_dynamic_segmentation", C.c_int), ("enable_raw_output", C.c_int), ("output_spacing", C.c_double), ("smoothing_method", C.c_int), ("optimization_method", C.c_int), ("enable_exact_position", C.c_int), ("polish", C.c_int),
        ("polish_delta", C.c_double), ("polish_refine_iter", C.c_int),
        ("refine", C.c_int), ("refine_eps", C.c_double), ("refine_rounds", C.c_int), ("refine_chain", C.c_int), ("refine_extra_rounds", C.c_int),
        ("refine_newton_rho", C.c_double), ("refine_newton_rho_eq", C.c_double), ("refine_newton_rho_max", C.c_double), ("refine_ls_tol", C.c_double), ("refine_ls_max", C.c_int), ("refine_newton_max", C.c_int), ("refine_newton_final", C.c_int), ("refine_newton_escalate", C.c_int), ("refine_newton_rho_eq_max", C.c_double),
    ]


class PoMap(C.Structure):
    _fields_ = [("distance", C.c_void_p), ("size_x", C.c_int), ("size_y", C.c_int), ("resolution", C.c_double),
                ("pos_x", C.c_double), ("pos_y", C.c_double)]


class PoInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int), ("n_refactor", C.c_int), ("status_polish", C.c_int),
                ("r_prim", C.c_double), ("r_dual", C.c_double), ("rho", C.c_double), ("obj", C.c_double),
                ("status_refine", C.c_int), ("reserved", C.c_int)]


class PoBatchIn(C.Structure):
    _fields_ = [("formulation", C.c_int), ("B", C.c_int), ("N", C.c_int), ("keep", C.c_int),
                ("ref_x", C.c_void_p), ("ref_y", C.c_void_p), ("ref_z", C.c_void_p), ("ref_k", C.c_void_p),
                ("ref_s", C.c_void_p), ("bounds", C.c_void_p), ("x0", C.c_void_p), ("goal_z", C.c_void_p),
                ("max_k", C.c_void_p), ("max_kp", C.c_void_p), ("n_points", C.c_void_p), ("order", C.c_void_p)]


class PoBatchOut(C.Structure):
    _fields_ = [("states", C.c_void_p), ("info", C.c_void_p), ("x", C.c_void_p)]


INFO_DTYPE = [("status", "<i4"), ("iters", "<i4"), ("n_refactor", "<i4"), ("status_polish", "<i4"),
              ("r_prim", "<f8"), ("r_dual", "<f8"), ("rho", "<f8"), ("obj", "<f8"), ("status_refine", "<i4"), ("reserved", "<i4")]
INFO_BYTES = 56  # sizeof(po_info)


class PoBoundsIn(C.Structure):
    _fields_ = [("B", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("ref_x", C.c_void_p), ("ref_y", C.c_void_p), ("ref_z", C.c_void_p), ("ref_s", C.c_void_p), ("n_points", C.c_void_p),
                ("knot_s", C.c_void_p), ("knot_x", C.c_void_p), ("knot_y", C.c_void_p), ("n_knots", C.c_void_p)]


PO_SMOOTH_TENSION2, PO_SMOOTH_TENSION, PO_SMOOTH_POST = 0, 1, 2


class PoSmoothIn(C.Structure):
    _fields_ = [("kind", C.c_int), ("B", C.c_int), ("P", C.c_int), ("n_points", C.c_void_p),
                ("x", C.c_void_p), ("y", C.c_void_p), ("angle", C.c_void_p), ("k", C.c_void_p), ("s", C.c_void_p),
                ("lb", C.c_void_p), ("ub", C.c_void_p), ("l0", C.c_void_p)]


class PoSmoothOut(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("s", C.c_void_p), ("info", C.c_void_p), ("raw", C.c_void_p)]


class PoSplineIn(C.Structure):
    _fields_ = [("B", C.c_int), ("K", C.c_int), ("knot_s", C.c_void_p), ("knot_x", C.c_void_p), ("knot_y", C.c_void_p),
                ("n_knots", C.c_void_p), ("length", C.c_void_p)]


class PoPlanIn(C.Structure):
    _fields_ = [("B", C.c_int), ("W", C.c_int), ("n_way", C.c_void_p), ("way_x", C.c_void_p), ("way_y", C.c_void_p), ("start", C.c_void_p),
                ("goal", C.c_void_p), ("max_length", C.c_double), ("N", C.c_int)]


class PoPlanOut(C.Structure):
    _fields_ = [("states", C.c_void_p), ("n_states", C.c_void_p), ("ok", C.c_void_p), ("stage", C.c_void_p), ("info", C.c_void_p)]
