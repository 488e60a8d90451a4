"""Dev tool (GPU box): one path of a golden set through a PO_NW_TRACE build of the Newton kernel (csrc: `make dev DEVFLAGS=-DPO_NW_TRACE TAG=_tr`,
then PO_LIB=path_optimizer_amd/libpo_hip_dev_tr.so python tools/newton_trace.py c3 <path> <batch>): prints the per-step trace and the cycle counters of path 0; compare with the oracle's
po_oracle_set_refine_trace(1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from make_tight_full import batch_of
from path_optimizer_amd import binding
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
b = batch_of(sys.argv[1], B, int(sys.argv[2]))
p = binding.default_params(); p.refine=2; p.refine_rounds=5; p.refine_extra_rounds=2; p.refine_eps=1e-8; p.refine_chain=2
st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
print(info[:1])
