// po_solve_form.hip — the solve_kernel_fast instantiations of ONE formulation and ONE loop variant (-DPO_FORM=0 KP, 1 KPC, 2 K;
// -DPO_UNI=1 uniform row classes, 0 general), five objects that build in parallel (K has no uniform variant).  -DPO_DEV_HEADLINE (dev builds only) keeps just the BASELINE config-3 variant.
#include "po_solve_common.hpp"

#if !defined(PO_FORM) || !defined(PO_UNI)
#error "compile with -DPO_FORM=0|1|2 -DPO_UNI=0|1"
#endif
#define PO_CAT2(a, b) a##b
#define PO_CAT(a, b) PO_CAT2(a, b)
#if PO_FORM == 0
#define PO_ENTRY_BASE po_launch_solve_kp
#elif PO_FORM == 1
#define PO_ENTRY_BASE po_launch_solve_kpc
#else
#define PO_ENTRY_BASE po_launch_solve_k
#endif
#ifndef PO_REF
#define PO_REF 0
#endif
#if PO_UNI && PO_REF == 2  // the variants with the Newton refinement phase (po_params.refine = 2): their own objects
#define PO_ENTRY PO_CAT(PO_ENTRY_BASE, _uni_nw)
#elif PO_REF == 2
#define PO_ENTRY PO_CAT(PO_ENTRY_BASE, _nw)
#elif PO_UNI && PO_REF  // the variants with the refinement phase (po_params.refine = 1): their own objects
#define PO_ENTRY PO_CAT(PO_ENTRY_BASE, _uni_ref)
#elif PO_UNI
#define PO_ENTRY PO_CAT(PO_ENTRY_BASE, _uni)
#elif PO_REF
#define PO_ENTRY PO_CAT(PO_ENTRY_BASE, _ref)
#else
#define PO_ENTRY PO_ENTRY_BASE
#endif

#if PO_REF == 3  // the Newton refinement as its own kernels (po_params.refine = 2, refine_chain = 2): one object per formulation, nothing else in it
#if PO_FORM == 0
#define PO_NEWTON_ENTRY po_launch_newton_kp
#elif PO_FORM == 1
#define PO_NEWTON_ENTRY po_launch_newton_kpc
#else
#define PO_NEWTON_ENTRY po_launch_newton_k
#endif
extern "C" hipError_t PO_NEWTON_ENTRY(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_newton<PO_FORM, false>(in, P, st); }
extern "C" hipError_t PO_CAT(PO_NEWTON_ENTRY, _fb)(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_newton<PO_FORM, true>(in, P, st); }
#else
extern "C" hipError_t PO_ENTRY(const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out) {
    return po::launch_form<PO_FORM, PO_UNI != 0, PO_REF>(in, P, st, lds_out);
}
#endif

#if !PO_UNI && !PO_REF  // the polish kernels of this formulation build with the general-variant object
#if PO_FORM == 0
#define PO_POLISH_ENTRY po_launch_polish_kp
#define PO_POLISH_SIZE po_polish_state_doubles_kp
#elif PO_FORM == 1
#define PO_POLISH_ENTRY po_launch_polish_kpc
#define PO_POLISH_SIZE po_polish_state_doubles_kpc
#else
#define PO_POLISH_ENTRY po_launch_polish_k
#define PO_POLISH_SIZE po_polish_state_doubles_k
#endif
extern "C" hipError_t PO_POLISH_ENTRY(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_polish<PO_FORM>(in, P, st); }
extern "C" int PO_POLISH_SIZE(int N, int C, int keep) { return po::polish_state_doubles<PO_FORM>(N, C, keep); }
#endif
