import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, np_twin as T
from oracle import oracle_py as O
from path_optimizer_amd import binding, synth
def rand(keep, N, B, seed):
    rng = np.random.default_rng(seed)
    insts = [T.random_instance(rng, N, ds=1.2 / keep * 0.999) for _ in range(B)]
    st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
    return synth.Batch(0, B, N, keep, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"), st("x0"), np.array([i["goal_z"] for i in insts]))
keep = int(sys.argv[1])
for N in (2 * keep + 1, 100, 200, 32 * keep - 3):
    b = rand(keep, N, 6, 3)
    b.n_points = np.array([N, N - 1, N - keep, max(3, N // 2), max(3, N // 3), N], np.int32)
    p = binding.default_params(); p.max_iter, p.check_every, p.adapt_every = 50, 0, 0
    st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
    ost, oinfo, oxs = O.solve_batch(b, O.device_equivalent_params(p))
    d1 = np.abs(xs - oxs).max()
    p = binding.default_params()
    for k, v in dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8).items(): setattr(p, k, v)
    st, info, xs = binding.Engine(0, p).solve_batch(b, want_x=True)
    ost, oinfo, oxs = O.solve_batch(b, O.device_equivalent_params(p), want_x=True)
    print("keep", keep, "N", N, "fixed50 diff %.2e" % d1, "headline diff %.2e" % np.abs(xs - oxs).max(), "status_refine", info["status_refine"], oinfo["status_refine"], "iters", info["iters"], flush=True)
b = synth.replicate(rand(keep, 200, 256, keep), 4096)
db = binding.DeviceBatch(b)
for label, kw in (("plain", {}), ("headline", dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2))):
    p = binding.default_params()
    for k, v in kw.items(): setattr(p, k, v)
    eng = binding.Engine(0, p); s = torch.cuda.Stream(); eng.set_stream(s.cuda_stream)
    eng.solve_batch_device(db); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.solve_batch_device(db)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    info = db.info_numpy()
    print("keep", keep, label, "%.2f ms" % (dt * 1e3), "%.0f k paths/s" % (4096 / dt / 1e3), "iters mean %.1f" % info["iters"].mean(), "certified", int((info["status_refine"] == 1).sum()), "solved", int((info["status"] == 1).sum()))
