"""Dev tool (GPU box): TENSION raw iterates of the blocked substitution and of the column-by-column one ("smooth_seq") against the oracle, per instance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as O
from path_optimizer_amd import binding, synth
kind = 1
d = synth.make_distance_map(3)
om = O.make_map(*d[:4])
eng = binding.Engine(0); eng.set_map(*d[:4])
inp = synth.make_smooth_inputs(22, 40, P=70, kind=kind, ragged=True, jitter_ds=True)
inp["n_points"][:3] = [3, 4, 70]
orc = O.smooth_batch(kind, O.default_params(), inp, m_map=om, want_raw=True)
res = {}
for tag, seq in (("blocked", 0), ("lanes", 1)):
    eng.debug_set("smooth_seq", seq)
    res[tag] = eng.smooth_batch(kind, inp, want_raw=True)
eng.debug_set("smooth_seq", 0)
for b in range(40):
    n = int(inp["n_points"][b])
    print(b, "P", n, "iters orc/blk/lanes", orc[3]["iters"][b], res["blocked"][3]["iters"][b], res["lanes"][3]["iters"][b],
          "refac", orc[3]["n_refactor"][b], "rho %.3g" % orc[3]["rho"][b],
          "err blk %.2e lanes %.2e  blk-lanes %.2e" % (np.abs(res["blocked"][4][b] - orc[4][b]).max(), np.abs(res["lanes"][4][b] - orc[4][b]).max(), np.abs(res["blocked"][4][b] - res["lanes"][4][b]).max()),
          "|x| %.1f" % np.abs(orc[4][b]).max())
