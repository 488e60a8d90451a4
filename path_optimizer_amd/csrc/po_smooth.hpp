// po_smooth.hpp — launch arguments of the reference-smoothing QP engine (po_smooth.hip); shared with po_capi.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/po_hip.h"
#include "po_map.hpp"

namespace po {
struct DevSmooth {
    int kind, B, P;  // P = stride of every [B][P] array = max points per instance
    const int *n_points;
    const double *x, *y, *angle, *k, *s, *lb, *ub, *l0;
    double *out_x, *out_y, *out_s;
    po_info *info;
    double *raw;
    int raw_stride;
    double w[6];  // t2_w_dev, t2_w_curv, t2_w_curv_rate, cart_w_curv, cart_w_curv_rate, cart_w_dev
    double sigma, alpha, rho0, eps_abs, eps_rel, eps_pinf, eps_dinf, adapt_tol;
    int max_iter, check_every, adapt_every, scaling;
    double *scratch;        // [B][scratch_stride]: scaled P band, D, E
    size_t scratch_stride;  // doubles
    DevMap map;             // TENSION only
    int perm_bits;          // block -> instance mixing (po_device.hpp perm_index), 0 = blockIdx order
    long long *dbg_cycles;  // optional [B][8] per-phase shader-clock totals (dev tool: PO_SMOOTH_DEBUG=1), or nullptr
    int seq_band;           // dev (po_debug_set "smooth_seq"): narrow-band substitutions on one lane (the round-1 path) instead of partitioned over the wave
    int waves;              // dev (PO_SMOOTH_WAVES=1|4|8): waves per QP of the narrow-band kinds (0: chosen from the LDS footprint and the batch size)
    int blocked;            // set by po_launch_smooth: TENSION in the block layout of the factor (po_smooth_blocked; off with seq_band)
    int nopad;              // dev (PO_SMOOTH_NOPAD=1): the partitioned substitution on the natural LDS layout (bank conflicts) for A/B
};
}  // namespace po
