"""Dev tool: the synthetic batch of `host_test bench` (path_optimizer_amd/host/test/host_test.cpp) rebuilt in numpy, solved by the oracle (CPU) or the device at the headline setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from path_optimizer_amd import synth

def host_batch(B=4096, N=200, formulation=0):
    rx = np.zeros((B, N)); ry = np.zeros((B, N)); rz = np.zeros((B, N)); rk = np.zeros((B, N)); rs = np.zeros((B, N)); bd = np.zeros((B, N, 4, 2)); x0 = np.zeros((B, 3)); gz = np.zeros(B)
    for b in range(B):
        z = 0.3 * (b % 17); x = 0.0; y = 0.0
        for i in range(N):
            s = 0.25 * i; k = 0.04 * np.sin(0.2 * s + (b % 31))
            rx[b, i] = x; ry[b, i] = y; rz[b, i] = z; rk[b, i] = k; rs[b, i] = s
            x += np.cos(z) * 0.25; y += np.sin(z) * 0.25; z += k * 0.25
            w = 1.6 + 0.4 * np.sin(0.1 * i + (b % 13))
            bd[b, i, :, 0] = -w; bd[b, i, :, 1] = w
        x0[b] = (0.2 - 0.01 * (b % 29), 0.03, rk[b, 0])
        gz[b] = rz[b, N - 1] + 0.02
    return synth.Batch(formulation, B, N, 4, rx, ry, rz, rk, rs, bd, x0, gz)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    which = sys.argv[2] if len(sys.argv) > 2 else "oracle"
    b = host_batch(B)
    kw = dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2)
    if which == "oracle":
        from oracle import oracle_py
        p = oracle_py.device_equivalent_params()
        for k, v in kw.items(): setattr(p, k, v)
        t0 = time.time(); st, info, xs = oracle_py.solve_batch(b, p); print("oracle s", time.time() - t0)
    else:
        from path_optimizer_amd import binding
        p = binding.default_params()
        for k, v in kw.items(): setattr(p, k, v)
        eng = binding.Engine(0, p)
        st, info, xs = eng.solve_batch(b, want_x=True)
        print("fallback paths", eng.debug_get("fallback_paths"), "phases", eng.last_phase_ms())
        np.save("gpurun_out/host_workload_iters.npy", info["iters"]); np.save("gpurun_out/host_workload_cert.npy", info["status_refine"])
    it = info["iters"]; bad = np.nonzero(info["status_refine"] != 1)[0]
    print("iters mean", it.mean(), "max", it.max(), "uncertified", len(bad), bad[:40], "solved", (info["status"] == 1).sum())
    print("iters of bad", it[bad][:40], "r_prim", info["r_prim"][bad][:8], "r_dual", info["r_dual"][bad][:8])
    print("hist", np.histogram(it, bins=[0, 40, 60, 80, 100, 150, 200, 400, 1000, 10000])[0])
