// po_solve_form.hip — the kernel instantiations of ONE formulation (-DPO_FORM=0 KP, 1 KPC, 2 K) and ONE kind:
//   -DPO_UNI=1 / 0            the solve kernels, uniform-row-class / general loop variant (the general object also holds the polish kernels);
//   -DPO_UNI=0 -DPO_REF=3     the Newton refinement (po_params.refine = 2): newton_kernel + newton_fallback_kernel, nothing else.
// Eleven objects that build in parallel (the Newton refinement of KP is three: its shapes split in groups).  -DPO_DEV_HEADLINE (dev builds only) keeps just one shape (po_solve_common.hpp).
#include "po_solve_common.hpp"

#if !defined(PO_FORM) || !defined(PO_UNI)
#error "compile with -DPO_FORM=0|1|2 -DPO_UNI=0|1"
#endif
#define PO_CAT2(a, b) a##b
#define PO_CAT(a, b) PO_CAT2(a, b)
#if PO_FORM == 0
#define PO_ENTRY_BASE po_launch_solve_kp
#define PO_NEWTON_ENTRY po_launch_newton_kp
#define PO_POLISH_ENTRY po_launch_polish_kp
#define PO_POLISH_SIZE po_polish_state_doubles_kp
#define PO_POLISH_HAS po_has_polish_kernel_kp
#elif PO_FORM == 1
#define PO_ENTRY_BASE po_launch_solve_kpc
#define PO_NEWTON_ENTRY po_launch_newton_kpc
#define PO_POLISH_ENTRY po_launch_polish_kpc
#define PO_POLISH_SIZE po_polish_state_doubles_kpc
#define PO_POLISH_HAS po_has_polish_kernel_kpc
#else
#define PO_ENTRY_BASE po_launch_solve_k
#define PO_NEWTON_ENTRY po_launch_newton_k
#define PO_POLISH_ENTRY po_launch_polish_k
#define PO_POLISH_SIZE po_polish_state_doubles_k
#define PO_POLISH_HAS po_has_polish_kernel_k
#endif
#ifndef PO_REF
#define PO_REF 0
#endif

#if PO_REF == 3  // the Newton refinement as its own kernels: one object per formulation (KP: three, -DPO_SHAPE_GROUP=1 / 2 / 3 — the entries of the second and third end in _b / _c)
#if PO_SHAPE_GROUP == 2
#define PO_NEWTON_ENTRY_G PO_CAT(PO_NEWTON_ENTRY, _b)
#elif PO_SHAPE_GROUP == 3
#define PO_NEWTON_ENTRY_G PO_CAT(PO_NEWTON_ENTRY, _c)
#else
#define PO_NEWTON_ENTRY_G PO_NEWTON_ENTRY
#endif
extern "C" hipError_t PO_NEWTON_ENTRY_G(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_newton<PO_FORM, false>(in, P, st); }
extern "C" hipError_t PO_CAT(PO_NEWTON_ENTRY_G, _fb)(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_newton<PO_FORM, true>(in, P, st); }
#if defined(PO_DEV_HEADLINE) && PO_FORM == 0  // dev builds hold one shape in one Newton object: the second object's entries answer "not mine"
extern "C" hipError_t po_launch_newton_kp_b(const po::DevBatch *, const po::DevParams *, hipStream_t) { return hipErrorInvalidValue; }
extern "C" hipError_t po_launch_newton_kp_b_fb(const po::DevBatch *, const po::DevParams *, hipStream_t) { return hipErrorInvalidValue; }
extern "C" hipError_t po_launch_newton_kp_c(const po::DevBatch *, const po::DevParams *, hipStream_t) { return hipErrorInvalidValue; }
extern "C" hipError_t po_launch_newton_kp_c_fb(const po::DevBatch *, const po::DevParams *, hipStream_t) { return hipErrorInvalidValue; }
#endif
#elif PO_UNI
extern "C" hipError_t PO_CAT(PO_ENTRY_BASE, _uni)(const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out) {
    return po::launch_form<PO_FORM, true>(in, P, st, lds_out);
}
#else
extern "C" hipError_t PO_ENTRY_BASE(const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out) {
    return po::launch_form<PO_FORM, false>(in, P, st, lds_out);
}
// the polish kernels of this formulation build with the general-variant object
extern "C" hipError_t PO_POLISH_ENTRY(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_polish<PO_FORM>(in, P, st); }
extern "C" int PO_POLISH_SIZE(int N, int C, int keep) { return po::polish_state_doubles<PO_FORM>(N, C, keep); }
extern "C" int PO_CAT(PO_POLISH_SIZE, _park)(int N, int C, int keep) { return po::newton_park_doubles<PO_FORM>(N, C, keep); }
extern "C" int PO_POLISH_HAS(int N, int C, int keep) { return po::has_polish_kernel<PO_FORM>(N, C, keep) ? 1 : 0; }
#endif
