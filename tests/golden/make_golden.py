"""Generates tests/golden/ref_{KP,KPC,K}.npz from the reference-compiled library oracle/_ref/libpo_ref.so
(only possible where /root/reference exists).  The fixtures hold, per case, the QP the REFERENCE's own
setHessianMatrix/setConstraintMatrix produced (CSC), its bounds, one solution vector and the reference's
getOptimizedPath output for it.  Inputs are regenerated deterministically by the tests (np_twin.random_instance)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import np_twin as T  # noqa: E402
from oracle import oracle_py, ref_py  # noqa: E402

CASES = [(2, 0.25), (9, 0.3), (40, 0.25), (64, 0.5), (130, 0.25)]
for form, name in ((0, "KP"), (1, "KPC"), (2, "K")):
    out = {"ncases": len(CASES)}
    for k, (N, ds) in enumerate(CASES):
        inst = T.random_instance(np.random.default_rng(1000 * form + N), N, ds=ds)
        p = oracle_py.default_params()
        p.scaling = 0
        p.max_iter = 200
        r = ref_py.solve(name, inst, p)
        assert r["rc"] in (0, 1)
        P, A = r["P"], r["A"]
        out.update({f"N_{k}": N, f"ds_{k}": ds, f"Pp_{k}": P.indptr, f"Pi_{k}": P.indices, f"Px_{k}": P.data,
                    f"Ap_{k}": A.indptr, f"Ai_{k}": A.indices, f"Ax_{k}": A.data, f"l_{k}": r["l"], f"u_{k}": r["u"], f"x_{k}": r["x"]})
        # the reference's getOptimizedPath on that x (solve() returns it only when solved: re-run the map through the oracle
        # ONLY if the reference did not produce states; otherwise keep the reference's own output)
        if r["rc"] == 1:
            out[f"states_{k}"] = r["states"]
        else:
            q = oracle_py.default_params(); q.scaling = 0; q.max_iter = 4000
            r2 = ref_py.solve(name, inst, q)
            assert r2["rc"] == 1
            out[f"x_{k}"] = r2["x"]; out[f"states_{k}"] = r2["states"]
    np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), **out)
    print("wrote", name)
