"""Accuracy clause under the refinement / polish settings: share of BASELINE config-3 paths whose e_y lies within 1e-4 m RMS of the exact optimum
(tests/golden/tight_c3.npz, the first 256 paths of the config) and the throughput of each setting on the full config (B = 4096).  GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_amd import binding, synth

gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tight_c3.npz"))["e_y"]
small = synth.make_batch(3, B=len(gold))
full = synth.make_batch(3)
N = small.N
rows = []
settings = [dict(), dict(polish=1), dict(polish=1, polish_passes=6), dict(refine=1), dict(refine=1, polish=1), dict(refine=1, polish=1, polish_passes=6)]
for a in sys.argv[1:]:
    settings.append(json.loads(a))
for kw in settings:
    p = binding.default_params()
    for k, v in kw.items(): setattr(p, k, v)
    eng = binding.Engine(0, p)
    st, info, xs = eng.solve_batch(small, want_x=True)
    rms = np.sqrt(np.mean((xs[:, 0:3 * N:3] - gold) ** 2, axis=1))
    import torch
    dev = binding.DeviceBatch(full)
    for _ in range(2): eng.solve_batch_device(dev)
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.solve_batch_device(dev); torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.median(ms))
    row = dict(setting=kw, frac_le_1e4=float((rms <= 1e-4).mean()), frac_le_1e3=float((rms <= 1e-3).mean()), max_rms=float(rms.max()), median_rms=float(np.median(rms)),
               iters_mean=float(info["iters"].mean()), refactor_mean=float(info["n_refactor"].mean()), polished=float((info["status_polish"] == 1).mean()),
               solved=float((info["status"] == 1).mean()), ms=ms, paths_per_s=full.B / ms * 1e3)
    rows.append(row)
    print(json.dumps(row), flush=True)
json.dump(rows, open(os.path.join("gpurun_out", "refine_eval.json"), "w"), indent=1)
