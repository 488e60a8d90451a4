"""Batch-size / path-length sweeps and the PCIe-inclusive rate of the host-pointer entry.  Dev tool (GPU box)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth

out = {"batch_sweep": [], "length_sweep": [], "pcie": {}}
base = synth.make_batch(3, B=4096)
eng = binding.Engine(0)
s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
def timed(db, reps=3):
    eng.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.solve_batch_device(db)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for B in (64, 256, 1024, 2048, 4096, 8192, 16384, 32768):
    batch = base.slice(0, B) if B <= 4096 else synth.replicate(base, B)
    db = binding.DeviceBatch(batch)
    dt = timed(db)
    it = db.info_numpy()["iters"]
    out["batch_sweep"].append({"B": B, "ms": dt * 1e3, "paths_per_s": B / dt, "path_iters_per_s": float(it.sum()) / dt})
    print(f"B={B:6d}: {dt*1e3:8.2f} ms  {B/dt:10.0f} paths/s  {it.sum()/dt:.3e} path-iters/s", flush=True)
for N in (40, 80, 120, 160, 200, 256, 320, 400, 512):
    batch = synth.make_batch(3, B=512, N=N)
    batch = synth.replicate(batch, 4096)
    db = binding.DeviceBatch(batch)
    dt = timed(db)
    it = db.info_numpy()["iters"]
    out["length_sweep"].append({"N": N, "ms": dt * 1e3, "paths_per_s": 4096 / dt, "iters_mean": float(it.mean())})
    print(f"N={N:4d}: {dt*1e3:8.2f} ms  {4096/dt:10.0f} paths/s  iters mean {it.mean():.0f}", flush=True)
# host-pointer entry: H2D + solve + D2H
eng2 = binding.Engine(0)
eng2.solve_batch(base)
t0 = time.perf_counter()
for _ in range(3): eng2.solve_batch(base)
dt = (time.perf_counter() - t0) / 3
out["pcie"] = {"B": 4096, "ms": dt * 1e3, "paths_per_s": 4096 / dt, "bytes_in": int(4096 * (200 * 13 + 4) * 8), "bytes_out": int(4096 * 200 * 5 * 8)}
print(f"host-pointer entry (PCIe inclusive): {dt*1e3:.2f} ms  {4096/dt:.0f} paths/s")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/sweep.json", "w"), indent=1)
