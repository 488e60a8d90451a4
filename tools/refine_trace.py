"""Dev tool (CPU): the refinement rounds of single paths of BASELINE config 3 in the oracle, block by block (po_oracle_set_refine_trace).  python tools/refine_trace.py 2410 [setting k=v ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as O
from path_optimizer_amd import synth
ids = [int(a) for a in sys.argv[1:] if "=" not in a] or [2410]
kw = dict(refine=1, refine_rounds=3, refine_extra_rounds=2)
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
full = synth.make_batch(3, B=4096)
p = O.device_equivalent_params()
for k, v in kw.items(): setattr(p, k, v)
O.lib().po_oracle_set_refine_trace(1)
for i in ids:
    print("== path", i, kw, file=sys.stderr)
    _, info, _ = O.solve_batch(full.slice(i, i + 1), p, want_x=True)
    print("   ->", info, file=sys.stderr)
