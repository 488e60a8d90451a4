"""Dev tool (GPU box): the Newton refinement (refine = 2) on the device against the oracle and the exact optima, and timing against the other settings.

    python tools/newton_dev.py [set] [B] [key=value ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_tight_full import SETS, batch_of, e_y_of  # noqa: E402


def main():
    import torch
    from oracle import oracle_py as O
    from path_optimizer_amd import binding

    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    kv = dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8)
    for a in sys.argv[3:]:
        k, v = a.split("="); kv[k] = float(v)
    b = batch_of(name, B)
    p = binding.default_params()
    for k, v in kv.items():
        setattr(p, k, type(getattr(p, k))(v))
    eng = binding.Engine(0, p)
    st, info, xs = eng.solve_batch(b, want_x=True)
    try:
        print("  paths handed to the fallback launch:", eng.debug_get("fallback_paths"))
    except Exception as e:
        print("  (no fallback count:", e, ")")
    t0 = time.time()
    ost, oinfo, oxs = O.solve_batch(b, O.device_equivalent_params(p))
    print(f"oracle {time.time() - t0:.1f} s")
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"tight_full_{name}.npz"))["e_y"].astype(np.float64)
    nb = min(B, len(gold))
    ey = np.stack([e_y_of(b.formulation, b.N, xs[i]) for i in range(nb)])
    rms = np.sqrt(np.mean((ey - gold[:nb]) ** 2, axis=1))
    same = (info["iters"] == oinfo["iters"]) & (info["n_refactor"] == oinfo["n_refactor"])
    dx = np.abs(xs - oxs).max(axis=1)
    print(f"{name} B={B} {kv}")
    print(f"  device: status==1 {(info['status'] == 1).sum()} certified {(info['status_refine'] == 1).sum()} iters mean {info['iters'].mean():.1f} max {info['iters'].max()}  nref mean {info['n_refactor'].mean():.1f}")
    print(f"  oracle: status==1 {(oinfo['status'] == 1).sum()} certified {(oinfo['status_refine'] == 1).sum()} iters mean {oinfo['iters'].mean():.1f} max {oinfo['iters'].max()}  nref mean {oinfo['n_refactor'].mean():.1f}")
    print(f"  equal (iters, nref) {same.mean():.3f}; max |dx| {dx.max():.2e} median {np.median(dx):.2e}; rms vs exact max {rms.max():.2e} n>1e-4 {(rms > 1e-4).sum()}")
    di = np.abs(info["iters"].astype(int) - oinfo["iters"].astype(int)); dn = np.abs(info["n_refactor"].astype(int) - oinfo["n_refactor"].astype(int))
    print(f"  |d iters| <= 1: {(di <= 1).mean():.3f}  <= 2: {(di <= 2).mean():.3f}  <= 3: {(di <= 3).mean():.3f}; |d nref| <= 1: {(dn <= 1).mean():.3f} <= 2: {(dn <= 2).mean():.3f}; iters equal {(di == 0).mean():.3f}")
    bad = np.where(~same)[0][:10]
    for i in bad:
        print(f"   path {i}: dev it {info['iters'][i]} nref {info['n_refactor'][i]} sref {info['status_refine'][i]} rp {info['r_prim'][i]:.2e} rd {info['r_dual'][i]:.2e} | "
              f"oracle it {oinfo['iters'][i]} nref {oinfo['n_refactor'][i]} sref {oinfo['status_refine'][i]} rp {oinfo['r_prim'][i]:.2e} rd {oinfo['r_dual'][i]:.2e} | dx {dx[i]:.2e}")
    # timing: a few solves of the device-resident batch
    dev = binding.DeviceBatch(b, want_x=False)
    for label, e in (("this setting", eng),):
        for _ in range(2):
            e.solve_batch_device(dev)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e.solve_batch_device(dev); torch.cuda.synchronize(); ts.append(e.last_kernel_ms())
        print(f"  {label}: {np.median(ts):.3f} ms per batch of {B} ({B / np.median(ts):.0f} k paths/s)")


if __name__ == "__main__":
    main()
