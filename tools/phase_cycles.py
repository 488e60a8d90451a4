"""Per-phase shader-clock breakdown of path 0 (PO_DEBUG_CYCLES hook).  Dev tool (GPU box)."""
import os, sys
os.environ["PO_DEBUG_CYCLES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_optimizer_amd import binding, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
batch = synth.replicate(synth.make_batch(cfg, B=min(B, 64)), B)
db = binding.DeviceBatch(batch)
p = binding.default_params(); p.max_iter = 100; p.check_every = 0; p.adapt_every = 0
eng = binding.Engine(0, p)
for _ in range(2):
    eng.solve_batch_device(db)
torch.cuda.synchronize()
print("kernel_ms", eng.last_kernel_ms())
