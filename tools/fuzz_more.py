"""Dev tool (GPU box): the two sweeps of tests/test_gpu_fuzz.py on further seeds — `python tools/fuzz_more.py LO HI` runs seeds LO .. HI-1 through both tests and lists what fails
("Beyond the committed seeds")."""
import sys, os, traceback
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import test_gpu_fuzz as F
from oracle import oracle_py
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    for fn in (F.test_random_case_matches_oracle, F.test_random_case_newton_matches_oracle):
        try:
            fn(oracle_py, seed)
        except Exception as e:
            bad.append((seed, fn.__name__, repr(e)[:300]))
            print("FAIL", seed, fn.__name__, repr(e)[:300], flush=True)
print("seeds", lo, hi, "failures", len(bad))
