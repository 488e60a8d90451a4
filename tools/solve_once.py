"""Dev tool: BASELINE config 3 solved a few times with the po_params given as a JSON object (argv[1]); wall ms per solve and iteration statistics.  GPU box."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
from path_optimizer_amd import binding, synth
import torch
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
full = synth.make_batch(cfg)
kw = json.loads(sys.argv[1])
p = binding.default_params()
for k, v in kw.items(): setattr(p, k, v)
eng = binding.Engine(0, p)
dev = binding.DeviceBatch(full)
for _ in range(2): eng.solve_batch_device(dev)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); eng.solve_batch_device(dev); torch.cuda.synchronize(); print((time.perf_counter() - t0) * 1e3)
info = dev.info_numpy()
print('iters mean', info['iters'].mean(), 'p50/p90/p99', np.percentile(info['iters'], [50, 90, 99]), 'max', info['iters'].max(), 'nfac mean', info['n_refactor'].mean(), 'max', info['n_refactor'].max(), 'status', np.unique(info['status'], return_counts=True))
