"""Dev tool (GPU box): the headline solve of BASELINE config 3 repeated `n` times on one handle (sliced Newton launches, the sort, the read-back): every repetition must reproduce the
first one bit for bit — statuses, iteration counts, solutions — and certify every path.  `python tools/soak.py [n] [B]`"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from path_optimizer_amd import binding, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = synth.make_batch(3, B=B)
db = binding.DeviceBatch(b, want_x=True)
p = binding.default_params()
for k, v in dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2).items(): setattr(p, k, v)
eng = binding.Engine(0, p); s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
ref = None; bad = 0; t0 = time.time()
for i in range(n):
    eng.solve_batch_device(db); torch.cuda.synchronize()
    info = db.info_numpy().copy(); x = db.out_x.cpu().numpy().copy(); st = db.out_states.cpu().numpy().copy()
    if ref is None: ref = (info, x, st)
    same = info.tobytes() == ref[0].tobytes() and np.array_equal(x, ref[1]) and np.array_equal(st, ref[2])
    if not same or (info["status_refine"] != 1).any():
        bad += 1
        print("repetition", i, "differs:", int((info["iters"] != ref[0]["iters"]).sum()), "iteration counts,", float(np.abs(x - ref[1]).max()), "max |dx|, uncertified", int((info["status_refine"] != 1).sum()), flush=True)
print("soak:", n, "solves of", B, "paths,", bad, "differing,", round(time.time() - t0, 1), "s; parked in the last", eng.debug_get("newton_parked"))
