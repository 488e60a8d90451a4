// po_solve_form.hip — the solve_kernel_fast instantiations of ONE formulation (compiled three times: -DPO_FORM=0 KP, 1 KPC, 2 K),
// so that the three sets build in parallel.  -DPO_DEV_HEADLINE (dev builds only) keeps just the BASELINE config-3 variant.
#include "po_solve_common.hpp"

#ifndef PO_FORM
#error "compile with -DPO_FORM=0|1|2"
#endif
#if PO_FORM == 0
#define PO_ENTRY po_launch_solve_kp
#elif PO_FORM == 1
#define PO_ENTRY po_launch_solve_kpc
#else
#define PO_ENTRY po_launch_solve_k
#endif

extern "C" hipError_t PO_ENTRY(const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out) {
    return po::launch_form<PO_FORM>(in, P, st, lds_out);
}
