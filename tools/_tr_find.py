import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from make_tight_full import batch_of
from path_optimizer_amd import binding
name = sys.argv[1]; B = int(sys.argv[2])
b = batch_of(name, B)
p = binding.default_params(); p.refine=2; p.refine_rounds=5; p.refine_extra_rounds=2; p.refine_eps=1e-8
st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
bad = np.where(info["status_refine"] != 1)[0]
print("uncertified:", bad.tolist(), info["iters"][bad].tolist())
for pid in [0] + bad[:2].tolist():
    print("==== trace of path", pid, flush=True)
    env = dict(os.environ, PO_LIB=os.path.join(os.path.dirname(binding.__file__), "libpo_hip_devnw_tr.so"))
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tr_dev.py"), name, str(pid)], env=env)
