"""Throughput with consecutive batches issued round-robin on S handles/streams (the stragglers of batch k drain while batch k+1 fills the CUs).  Dev tool (GPU box)."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth
base = synth.make_batch(3, B=4096)
db0 = binding.DeviceBatch(base)
def clone_outputs(db):
    d = copy.copy(db)
    d.out_states = torch.zeros_like(db.out_states); d.out_info = torch.zeros_like(db.out_info)
    return d
for S in (1, 2, 3, 4):
    engs, dbs = [], []
    for i in range(S):
        e = binding.Engine(0); st = torch.cuda.Stream(); e.set_stream(st.cuda_stream); e._st = st
        engs.append(e); dbs.append(clone_outputs(db0))
    for K in (5, 10, 20):
        for i in range(S): engs[i].solve_batch_device(dbs[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K): engs[k % S].solve_batch_device(dbs[k % S])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"streams {S} steps {K}: {dt/K*1e3:7.2f} ms/step  {4096*K/dt:9.0f} paths/s", flush=True)
    it = [d.info_numpy()["iters"] for d in dbs]
    assert all(np.array_equal(it[0], x) for x in it)
