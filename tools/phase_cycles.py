"""Per-phase shader-clock breakdown of path 0 (PO_DEBUG_CYCLES hook).  Dev tool (GPU box).
usage: phase_cycles.py B cfg            (BASELINE config)      |      phase_cycles.py B keep N   (KP, random corridors, spacing 1.2/keep)"""
import os, sys
os.environ["PO_DEBUG_CYCLES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from path_optimizer_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) > 3:
    import np_twin as T
    keep, N = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(keep)
    insts = [T.random_instance(rng, N, ds=1.2 / keep * 0.999) for _ in range(64)]
    st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    base = synth.Batch(0, 64, N, keep, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"), st("x0"), np.array([i["goal_z"] for i in insts]))
else:
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    base = synth.make_batch(cfg, B=min(B, 64))
batch = synth.replicate(base, B)
db = binding.DeviceBatch(batch)
p = binding.default_params(); p.max_iter = 100; p.check_every = 0; p.adapt_every = 0
eng = binding.Engine(0, p)
for _ in range(2):
    eng.solve_batch_device(db)
torch.cuda.synchronize()
print("kernel_ms", eng.last_kernel_ms())
