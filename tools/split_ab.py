"""Dev tool (GPU box): the stage-split two-wave mapping (PO_SPLIT=1, default) against the one-wave mapping (PO_SPLIT=0) of the keep-4 kernel:
bare iteration rate (fixed 200 iterations), one real config-3 launch, the order-hinted launch, 3-stream pipelined throughput; result equality."""
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
    out = {"split": os.environ.get("PO_SPLIT")}
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
    x = db.out_x.cpu().numpy().copy()
    db.set_order(np.argsort(-info["iters"].astype(np.int64), kind="stable"))
    eng2.solve_batch_device(db); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng2.solve_batch_device(db); torch.cuda.synchronize(); ts.append(eng2.last_kernel_ms())
    out["hinted_ms"] = float(np.median(ts)); out["hinted_paths_s"] = 4096 / out["hinted_ms"] * 1e3
    db.set_order(None)
    # 3 streams
    engs = [binding.Engine(0) for _ in range(3)]; sts = [torch.cuda.Stream() for _ in range(3)]; dbs = [db.clone_outputs() for _ in range(3)]
    for e, s_ in zip(engs, sts): e.set_stream(s_.cuda_stream)
    for k in range(3): engs[k].solve_batch_device(dbs[k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(12): engs[k % 3].solve_batch_device(dbs[k % 3])
    torch.cuda.synchronize(); out["pipelined_paths_s"] = 12 * 4096 / (time.perf_counter() - t0)
    np.save(os.environ["AB_DUMP"], np.concatenate([x200.ravel(), x.ravel(), info["iters"].astype(np.float64)]))
    print("AB " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        one(); sys.exit(0)
    import numpy as np
    ref = None
    for i, sp in enumerate(("0", "1")):
        dump = f"/tmp/sab_{i}.npy"
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PO_SPLIT=sp, AB_CHILD="1", AB_DUMP=dump), capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        if not line:
            print(sp, "FAILED", r.stdout[-400:], r.stderr[-800:]); continue
        d = json.loads(line[0][3:]); v = np.load(dump)
        if ref is None: ref = v
        d["max_abs_diff_vs_first"] = float(np.abs(v - ref).max())
        print(json.dumps(d), flush=True)
