// Storage the OsqpEigen stand-in (oracle/ref_shim/OsqpEigen/OsqpEigen.h) needs in the drop-in test binary: it serves the reference's SMOOTHER QPs only
// (tension_smoother*.cpp, reference_path_smoother.cpp call OsqpEigen directly; OSQP is absent from this image).  The path QP does not pass through here.
#include "OsqpEigen/OsqpEigen.h"

namespace OsqpEigen {
static Captured g_cap;
static po_params g_params = [] { po_params p; po_oracle_default_params(&p); return p; }();
Captured &last_captured() { return g_cap; }
const po_params &shim_params() { return g_params; }
}  // namespace OsqpEigen
