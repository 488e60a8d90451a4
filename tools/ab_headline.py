"""Dev tool (GPU box): A/B of builds at the HEADLINE setting.  `python tools/ab_headline.py lib1.so lib2.so ...` runs, per library (PO_LIB, own process), BASELINE config 3
(4096 paths; AB_KEEP=k: the random keep-k batch of tools/keep_sweep.py instead) at the headline setting: median / min launch time, the phases (warm start, Newton launch),
iteration statistics, certificates, and a digest of the solution so that variants can be compared for equality.  Libraries come from `make -C path_optimizer_amd/csrc dev TAG=x`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HEADLINE = dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2)


def one():
    import numpy as np
    import torch

    from path_optimizer_amd import binding, synth

    keep = int(os.environ.get("AB_KEEP", "0"))
    if keep:
        import np_twin as T
        rng = np.random.default_rng(keep)
        insts = [T.random_instance(rng, 200, ds=1.2 / keep * 0.999) for _ in range(256)]
        st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
        batch = synth.replicate(synth.Batch(0, 256, 200, keep, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"), st("x0"), np.array([i["goal_z"] for i in insts])), 4096)
    else:
        batch = synth.make_batch(int(os.environ.get("AB_CFG", "3")), B=4096)
    db = binding.DeviceBatch(batch, want_x=True)
    out = {"lib": os.path.basename(binding.LIB_PATH)}
    p = binding.default_params()
    for k, v in HEADLINE.items():
        setattr(p, k, v)
    eng = binding.Engine(0, p)
    s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
    for _ in range(3):
        eng.solve_batch_device(db); torch.cuda.synchronize()
    ts, ph = [], []
    for _ in range(9):
        eng.solve_batch_device(db); torch.cuda.synchronize(); ts.append(eng.last_kernel_ms()); ph.append(eng.last_phase_ms())
    info = db.info_numpy()
    out["ms_median"] = float(np.median(ts)); out["ms_min"] = float(np.min(ts)); out["paths_s"] = 4096 / out["ms_median"] * 1e3
    out["warm_ms"] = float(np.median([q["warm_start"] for q in ph])); out["newton_ms"] = float(np.median([q["newton"] for q in ph]))
    out["iters_mean"] = float(info["iters"].mean()); out["iters_max"] = int(info["iters"].max()); out["refactor_mean"] = float(info["n_refactor"].mean())
    out["certified"] = int((info["status_refine"] == 1).sum()); out["unsolved"] = int((info["status"] != 1).sum())
    p0 = binding.default_params()
    e0 = binding.Engine(0, p0); e0.set_stream(s.cuda_stream)
    e0.solve_batch_device(db); torch.cuda.synchronize()
    t0 = []
    for _ in range(3):
        e0.solve_batch_device(db); torch.cuda.synchronize(); t0.append(e0.last_kernel_ms())
    i0 = db.info_numpy()
    out["plain_ms"] = float(np.median(t0)); out["plain_Mit_s"] = float(i0["iters"].sum()) / out["plain_ms"] / 1e3
    eng.solve_batch_device(db); torch.cuda.synchronize()
    np.save(os.environ["AB_DUMP"], np.concatenate([db.out_x.cpu().numpy().ravel(), db.info_numpy()["iters"].astype(np.float64)]))
    print("AB " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        one()
        sys.exit(0)
    import numpy as np
    ref = None
    for i, lib in enumerate(sys.argv[1:]):
        dump = f"/tmp/abh_{i}.npy"
        env = dict(os.environ, PO_LIB=os.path.abspath(lib), AB_CHILD="1", AB_DUMP=dump)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        if not line:
            print(lib, "FAILED", r.stdout[-400:], r.stderr[-1200:]); continue
        d = json.loads(line[0][3:])
        v = np.load(dump)
        if ref is None:
            ref = v
        d["max_abs_diff_vs_first"] = float(np.abs(v - ref).max()) if v.shape == ref.shape else None
        print(json.dumps(d), flush=True)
