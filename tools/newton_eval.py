"""Dev tool (CPU): the oracle at a refinement setting on a whole golden set — iteration statistics, certified count, distance to the exact optima.

    python tools/newton_eval.py <set> [B] key=value ...     (keys = po_params fields; default refine=2 refine_rounds=3 refine_extra_rounds=2)
"""
import os
import sys
import time
import multiprocessing as mp

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def _work(arg):
    name, lo, hi, kv = arg
    from make_tight_full import batch_of
    from oracle import oracle_py as O
    batch = batch_of(name, hi - lo, lo)
    p = O.device_equivalent_params()
    for k, v in kv.items():
        setattr(p, k, type(getattr(p, k))(v))
    st, info, xs = O.solve_batch(batch, p)
    return lo, info, xs


def main():
    from make_tight_full import SETS, e_y_of, batch_of
    name = sys.argv[1]
    rest = sys.argv[2:]
    B = SETS[name][2]
    if rest and "=" not in rest[0]:
        B = int(rest[0]); rest = rest[1:]
    kv = dict(refine=2, refine_rounds=3, refine_extra_rounds=2)
    for a in rest:
        k, v = a.split("="); kv[k] = float(v)
    nproc = int(os.environ.get("PROCS", "4"))
    step = max(16, -(-B // (4 * nproc)))
    jobs = [(name, lo, min(B, lo + step), kv) for lo in range(0, B, step)]
    t = time.time()
    with mp.get_context("spawn").Pool(nproc) as pool:
        parts = sorted(pool.map(_work, jobs, chunksize=1), key=lambda r: r[0])
    info = np.concatenate([r[1] for r in parts]); xs = np.concatenate([r[2] for r in parts])
    b0 = batch_of(name, 1)
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"tight_full_{name}.npz"))["e_y"].astype(np.float64)
    nb = min(B, len(gold))
    ey = np.stack([e_y_of(b0.formulation, b0.N, xs[b]) for b in range(nb)])
    rms = np.sqrt(np.mean((ey - gold[:nb]) ** 2, axis=1))
    it = info["iters"]
    order = np.argsort(-it)
    print(f"{name} B={B} {kv}  [{time.time() - t:.1f} s]")
    print(f"  status solved {(info['status'] == 1).sum()}  certified {(info['status_refine'] == 1).sum()}  iters mean {it.mean():.1f} p95 {np.percentile(it, 95):.0f} max {it.max()}  "
          f"refactor mean {info['n_refactor'].mean():.1f} max {info['n_refactor'].max()}")
    print(f"  e_y rms vs exact ({nb} paths): n>1e-4 {(rms > 1e-4).sum()}  max {rms.max():.2e}  p99 {np.percentile(rms, 99):.2e}")
    print("  hardest:", [(int(b), int(it[b]), int(info['n_refactor'][b]), int(info['status_refine'][b])) for b in order[:12]])


if __name__ == "__main__":
    main()
