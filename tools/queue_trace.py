"""Dev tool (GPU box): item timeline of one chained-rounds solve of BASELINE config 3 at the headline setting -> gpurun_out/queue_trace.npy + a summary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth
pol = int(sys.argv[1]) if len(sys.argv) > 1 else 0
spec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
full = synth.make_batch(3)
db = binding.DeviceBatch(full)
p = binding.default_params(); p.refine, p.refine_rounds, p.refine_extra_rounds, p.refine_speculate = 1, 3, 2, spec
eng = binding.Engine(0, p); eng.debug_set("queue_policy", pol); eng.debug_set("queue_trace", 1)
eng.solve_batch_device(db); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.solve_batch_device(db); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
tr = eng.debug_trace_read()
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/queue_trace_p%d_s%d.npy" % (pol, spec), tr)
key = tr[:, 0]; b = key & 0xffffffff; rnd = (key >> 32) & 0xff; sp = (key >> 40) & 1; oc = (key >> 48) & 0xff
ts = (tr[:, 1] - tr[:, 1].min()) / 100.0e3; te = (tr[:, 2] - tr[:, 1].min()) / 100.0e3  # ms
print("policy %d spec %d: %.2f ms wall; %d items; last end %.2f ms" % (pol, spec, ms, len(tr), te.max()))
for r in range(5):
    m = rnd == r
    if m.any():
        print(" round %d: %4d items (%4d speculative), start %.2f..%.2f ms, end max %.2f, duration mean %.3f max %.3f ms; outcomes %s" % (r, m.sum(), (m & (sp == 1)).sum(), ts[m].min(), ts[m].max(), te[m].max(), (te - ts)[m].mean(), (te - ts)[m].max(), np.bincount(oc[m], minlength=5).tolist()))
# the paths that end last: their chains
last = np.argsort(-te)[:3]
for i in last:
    pb = b[i]; m = b == pb
    o = np.argsort(ts[m])
    print(" path %d chain:" % pb, [("r%d%s" % (rnd[m][j], "s" if sp[m][j] else ""), round(float(ts[m][j]), 2), round(float(te[m][j]), 2), int(oc[m][j])) for j in o])
busy = np.zeros(400)
for a, c in zip(ts, te):
    busy[int(a * 20):int(c * 20) + 1] += 1
print(" resident working items per 0.5 ms:", [int(busy[k * 10:(k + 1) * 10].mean()) for k in range(int(te.max() * 2) + 1)])
