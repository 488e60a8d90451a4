"""Dev tool: static instruction mix of the headline solve kernel per phase of one ADMM iteration.
  hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -fno-signed-zeros -fno-honor-nans -DPO_DEV_HEADLINE -DPO_FORM=0 -DPO_UNI=1 -DPO_MARKS \
        -S --cuda-device-only -o /tmp/kdev.s path_optimizer_amd/csrc/po_solve_form.hip ; python tools/isa_phase_hist.py [/tmp/kdev.s]
The PO_MARK comments (po_fast.inc) delimit the phases; loops (scan row carries) are counted once."""
import re,collections,sys
L=open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/kdev.s').read().split('\n')
marks=[(i,l.split('PO_MARK ')[1].strip()) for i,l in enumerate(L) if 'PO_MARK' in l]
def isins(l): return l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')
def cls(op):
    if 'f64' in op: return 'fp64'
    if 'accvgpr' in op: return 'agpr'
    if op.startswith('v_cndmask'): return 'cndmask'
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
    if op.startswith('v_mov'): return 'vmov'
    if op.startswith('ds_bpermute'): return 'bperm'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch'): return 'scratch'
    if op.startswith('v_cmp'): return 'vcmp'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_'): return 'salu'
    return 'other'
for (a,na),(b,nb) in zip(marks,marks[1:]):
    ins=[l.split()[0] for l in L[a:b] if isins(l)]
    c=collections.Counter(cls(o) for o in ins)
    print(f"{na:14s}->{nb:14s} n={len(ins):5d} "+' '.join(f"{k}={v}" for k,v in sorted(c.items(),key=lambda x:-x[1])))
