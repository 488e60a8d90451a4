"""Dev tool (GPU box): the refinement on the other BASELINE configs / formulations — solved share, iteration counts, throughput, and device vs oracle on a sample."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_amd import binding, synth
from oracle import oracle_py as O
import torch
for name, cfg, form in (("c2", 2, None), ("c5 (KPC)", 5, None), ("K", 3, 2), ("c1", 1, None)):
    full = synth.make_batch(cfg) if form is None else synth.make_batch(cfg, formulation=form)
    small = synth.make_batch(cfg, B=24) if form is None else synth.make_batch(cfg, B=24, formulation=form)
    for kw in (dict(), dict(refine=1), dict(refine=1, eps_abs=3e-4, eps_rel=3e-4), dict(refine=1, refine_rounds=3)):
        p = binding.default_params()
        for k, v in kw.items(): setattr(p, k, v)
        eng = binding.Engine(0, p)
        st, info, xs = eng.solve_batch(small, want_x=True)
        ost, oinfo, oxs = O.solve_batch(small, O.device_equivalent_params(p))
        dev = binding.DeviceBatch(full)
        for _ in range(2): eng.solve_batch_device(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): eng.solve_batch_device(dev)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        fi = dev.info_numpy()
        same = info["iters"] == oinfo["iters"]
        print(f"{name:9s} {json.dumps(kw):55s} B {full.B} {ms:7.2f} ms {full.B / ms:7.1f} k paths/s | solved {(fi['status'] == 1).mean():.4f} its mean {fi['iters'].mean():6.1f} max {fi['iters'].max():5d} "
              f"| sample: status equal {np.array_equal(info['status'], oinfo['status'])} same iters {same.mean():.2f} max|dx| same {np.abs(xs - oxs)[same].max() if same.any() else -1:.1e} all {np.abs(st - ost)[..., :3].max():.1e} r_prim max {info['r_prim'].max():.1e}", flush=True)
