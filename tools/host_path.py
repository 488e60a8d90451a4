"""Dev tool (GPU box): the host-pointer entry po_solve_batch — wall clock host-to-host and its parts (po_last_phase_ms) on a BASELINE batch.

    python tools/host_path.py [set] [B] [threads ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_tight_full import batch_of  # noqa: E402


def main():
    from path_optimizer_amd import binding

    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    threads = [int(a) for a in sys.argv[3:]] or [0]
    b = batch_of(name, B)
    p = binding.default_params()
    p.refine, p.refine_rounds, p.refine_extra_rounds, p.refine_eps, p.refine_chain = 2, 5, 2, 1e-8, 2
    for nt in threads:
        eng = binding.Engine(0, p)
        eng.debug_set("host_threads", nt)
        eng.solve_batch(b)
        ts, ph = [], []
        for _ in range(7):
            t0 = time.perf_counter(); eng.solve_batch(b); ts.append((time.perf_counter() - t0) * 1e3); ph.append(eng.last_phase_ms())
        med = {k: float(np.median([q[k] for q in ph])) for k in ph[0]}
        print(f"{name} B={B} host_threads={nt or 'auto'}: host-to-host {np.median(ts):.2f} ms (min {np.min(ts):.2f})  " + "  ".join(f"{k} {v:.2f}" for k, v in med.items()))
        eng.close()


if __name__ == "__main__":
    main()
