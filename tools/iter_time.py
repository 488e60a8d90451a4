"""Micro-timing of the fused kernel at a FIXED iteration count (no termination test): per-iteration cost and
how it scales with the number of resident paths.  Dev tool (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth

base = synth.make_batch(3, B=64)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for B in (64, 256, 512, 1024, 4096):
    batch = synth.replicate(base, B)
    db = binding.DeviceBatch(batch)
    for iters in (0, 100, 200):
        p = binding.default_params(); p.max_iter = max(iters, 1); p.check_every = 0; p.adapt_every = 0
        eng = binding.Engine(0, p)
        s = torch.cuda.Stream()
        eng.set_stream(s.cuda_stream)
        eng.solve_batch_device(db); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.solve_batch_device(db)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"B={B} iters={p.max_iter}: {dt*1e3:.2f} ms  kernel_ms={eng.last_kernel_ms():.2f}", flush=True)
