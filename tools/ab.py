"""Dev tool (GPU box): A/B of solve-kernel builds.  `python tools/ab.py lib1.so lib2.so ...` runs, per library (PO_LIB, own process):
the bare iteration rate (BASELINE config 3, 4096 paths, 200 fixed iterations), one real config-3 launch (eps 1e-4) and a digest of the
solution so that variants can be compared for equality."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    import torch

    from path_optimizer_amd import binding, synth

    cfg = int(os.environ.get("AB_CFG", "3"))
    batch = synth.make_batch(cfg, B=4096)
    db = binding.DeviceBatch(batch, want_x=True)
    out = {"lib": os.path.basename(binding.LIB_PATH)}
    p = binding.default_params(); p.max_iter = 200; p.check_every = 0; p.adapt_every = 0
    eng = binding.Engine(0, p)
    eng.solve_batch_device(db); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng.solve_batch_device(db); torch.cuda.synchronize(); ts.append(eng.last_kernel_ms())
    out["bare_ms"] = float(np.median(ts)); out["bare_Mit_s"] = 4096 * 200 / out["bare_ms"] / 1e3
    x200 = db.out_x.cpu().numpy().copy()
    eng2 = binding.Engine(0)
    eng2.solve_batch_device(db); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng2.solve_batch_device(db); torch.cuda.synchronize(); ts.append(eng2.last_kernel_ms())
    info = db.info_numpy()
    out["real_ms"] = float(np.median(ts)); out["paths_s"] = 4096 / out["real_ms"] * 1e3
    out["iters_mean"] = float(info["iters"].mean()); out["unsolved"] = int((info["status"] != 1).sum())
    out["real_Mit_s"] = float(info["iters"].sum()) / out["real_ms"] / 1e3
    x = db.out_x.cpu().numpy()
    np.save(os.environ["AB_DUMP"], np.concatenate([x200.ravel(), x.ravel(), info["iters"].astype(np.float64)]))
    print("AB " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        one()
        sys.exit(0)
    import numpy as np
    ref = None
    for i, lib in enumerate(sys.argv[1:]):
        dump = f"/tmp/ab_{i}.npy"
        env = dict(os.environ, PO_LIB=os.path.abspath(lib), AB_CHILD="1", AB_DUMP=dump)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        if not line:
            print(lib, "FAILED", r.stdout[-400:], r.stderr[-800:]); continue
        d = json.loads(line[0][3:])
        v = np.load(dump)
        if ref is None:
            ref = v
        d["max_abs_diff_vs_first"] = float(np.abs(v - ref).max())
        print(json.dumps(d), flush=True)
