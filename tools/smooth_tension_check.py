"""Dev tool (GPU box): TENSION raw iterates of the blocked substitution (seq 0) and of the column-by-column one (seq 1) against the oracle at eps 1e-3 and 1e-4, three batches of 100-point QPs."""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle import oracle_py as O
from path_optimizer_amd import binding, synth
dist, res, px, py, _ = synth.make_distance_map(3)
om = O.make_map(dist, res, px, py)
for eps in (1e-3, 1e-4):
    for seed, B in ((21, 48), (22, 40), (5, 64)):
        inp = synth.make_smooth_inputs(seed, B, P=100, kind=1)
        p = binding.default_params(); p.eps_abs = p.eps_rel = eps
        op = O.default_params(); op.eps_abs = op.eps_rel = eps
        orc = O.smooth_batch(1, op, inp, m_map=om, want_raw=True)
        for seq in (0, 1):
            e = binding.Engine(0, p); e.set_map(dist, res, px, py); e.debug_set("smooth_seq", seq)
            dev = e.smooth_batch(1, inp, want_raw=True)
            same = dev[3]["iters"] == orc[3]["iters"]
            per = np.abs(dev[4] - orc[4]).reshape(B, -1).max(axis=1)[same]
            print("eps", eps, "seed", seed, "seq", seq, "same", same.mean(), "max %.2e" % per.max(), "n>1e-7", (per > 1e-7).sum(), "n>3e-7", (per > 3e-7).sum(), "iters max", orc[3]["iters"].max())
