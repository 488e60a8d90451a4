// po_solve_form.hip — the kernel instantiations of ONE formulation (-DPO_FORM=0 KP, 1 KPC, 2 K) and ONE kind:
//   -DPO_UNI=1 / 0            the solve kernels, uniform-row-class / general loop variant (the general object also holds the polish kernels);
//   -DPO_UNI=0 -DPO_REF=3     the Newton refinement (po_params.refine = 2): newton_kernel + newton_fallback_kernel, nothing else.
// Sixteen objects that build in parallel (KP: its shapes in groups — keep 1 .. 8 / the wide role-split shapes of keep 9 .. 16 — and the Newton refinement of each group by kind of shape).  -DPO_DEV_HEADLINE (dev builds only) keeps just one shape (po_solve_common.hpp).
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
// entries of an object that holds only a group of KP's shapes (-DPO_SHAPE_GROUP, po_solve_common.hpp) carry the group's suffix
#if PO_SHAPE_GROUP == 2
#define PO_G(name) PO_CAT(name, _b)
#elif PO_SHAPE_GROUP == 3
#define PO_G(name) PO_CAT(name, _c)
#elif PO_SHAPE_GROUP == 4
#define PO_G(name) PO_CAT(name, _w1)
#elif PO_SHAPE_GROUP == 5
#define PO_G(name) PO_CAT(name, _w2)
#elif PO_SHAPE_GROUP == 6
#define PO_G(name) PO_CAT(name, _w3)
#elif PO_SHAPE_GROUP == 7
#define PO_G(name) PO_CAT(name, _w)
#else
#define PO_G(name) name
#endif

#if PO_REF == 3  // the Newton refinement as its own kernels: one object per formulation (KP: three + three for the wide shapes, by shape group)
extern "C" hipError_t PO_G(PO_NEWTON_ENTRY)(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_newton<PO_FORM, false>(in, P, st); }
extern "C" hipError_t PO_CAT(PO_G(PO_NEWTON_ENTRY), _fb)(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_newton<PO_FORM, true>(in, P, st); }
#if defined(PO_DEV_HEADLINE) && PO_FORM == 0  // dev builds hold one shape in one object of each kind: the entries of the other groups answer "not mine"
#define PO_STUB(name) extern "C" hipError_t name(const po::DevBatch *, const po::DevParams *, hipStream_t) { return po::kNotMyShape; }
PO_STUB(po_launch_newton_kp_b) PO_STUB(po_launch_newton_kp_b_fb) PO_STUB(po_launch_newton_kp_c) PO_STUB(po_launch_newton_kp_c_fb)
PO_STUB(po_launch_newton_kp_w1) PO_STUB(po_launch_newton_kp_w1_fb) PO_STUB(po_launch_newton_kp_w2) PO_STUB(po_launch_newton_kp_w2_fb) PO_STUB(po_launch_newton_kp_w3) PO_STUB(po_launch_newton_kp_w3_fb)
#undef PO_STUB
#endif
#elif PO_UNI
extern "C" hipError_t PO_CAT(PO_G(PO_ENTRY_BASE), _uni)(const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out) {
    return po::launch_form<PO_FORM, true>(in, P, st, lds_out);
}
#if defined(PO_DEV_HEADLINE) && PO_FORM == 0
extern "C" hipError_t po_launch_solve_kp_w_uni(const po::DevBatch *, const po::DevParams *, hipStream_t, size_t *) { return po::kNotMyShape; }
#endif
#else
extern "C" hipError_t PO_G(PO_ENTRY_BASE)(const po::DevBatch *in, const po::DevParams *P, hipStream_t st, size_t *lds_out) {
    return po::launch_form<PO_FORM, false>(in, P, st, lds_out);
}
extern "C" int PO_G(PO_POLISH_SIZE)(int N, int C, int keep) { return po::polish_state_doubles<PO_FORM>(N, C, keep); }
extern "C" int PO_CAT(PO_G(PO_POLISH_SIZE), _park)(int N, int C, int keep) { return po::newton_park_doubles<PO_FORM>(N, C, keep); }
#if PO_SHAPE_GROUP == 0
// the polish kernels of this formulation build with the general-variant object (the wide role-split shapes have none)
extern "C" hipError_t PO_POLISH_ENTRY(const po::DevBatch *in, const po::DevParams *P, hipStream_t st) { return po::launch_polish<PO_FORM>(in, P, st); }
extern "C" int PO_POLISH_HAS(int N, int C, int keep) { return po::has_polish_kernel<PO_FORM>(N, C, keep) ? 1 : 0; }
#endif
#if defined(PO_DEV_HEADLINE) && PO_FORM == 0
extern "C" hipError_t po_launch_solve_kp_w(const po::DevBatch *, const po::DevParams *, hipStream_t, size_t *) { return po::kNotMyShape; }
extern "C" int po_polish_state_doubles_kp_w(int, int, int) { return 0; }
extern "C" int po_polish_state_doubles_kp_w_park(int, int, int) { return 0; }
#endif
#endif
