"""Dev tool: po_plan_batch_device from T host threads, one handle (= one HIP stream) each: consecutive batches of planning instances
overlap (the QP tail of one batch drains while the next batch's stages fill the CUs), like bench.py does for the headline."""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_optimizer_amd import binding, synth  # noqa: E402
from path_optimizer_amd.abi import INFO_BYTES

B = 4096
scn = synth.make_planning_scenes(2, 64)
rs = -(-B // 64)
perm = np.random.default_rng(5).permutation(64 * rs)[:B] % 64  # shuffled replication: a period-64 pattern would pin scenes to XCDs
tp = {k: torch.from_numpy(np.ascontiguousarray(scn[k][perm])).cuda() for k in ("way_x", "way_y", "start", "goal")}
Np = 320
way_len = float(np.hypot(np.diff(scn["way_x"], axis=1), np.diff(scn["way_y"], axis=1)).sum(axis=1).max())
for T in (1, 2, 3, 4):
    engs, outs = [], []
    for _ in range(T):
        e = binding.Engine(0); e.set_map(*scn["map"]); engs.append(e)
        outs.append(dict(states=torch.zeros((B, Np, 5), dtype=torch.float64, device="cuda"), n_states=torch.zeros(B, dtype=torch.int32, device="cuda"),
                         ok=torch.zeros(B, dtype=torch.int32, device="cuda"), stage=torch.zeros(B, dtype=torch.int32, device="cuda"), info=torch.zeros((B, INFO_BYTES), dtype=torch.uint8, device="cuda")))
        e.plan_batch_device(tp, outs[-1], Np, way_len)
    torch.cuda.synchronize()
    K = 3  # batches per thread

    def work(i):
        for _ in range(K):
            engs[i].plan_batch_device(tp, outs[i], Np, way_len)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("threads/handles %d: %.1f ms per batch of %d -> %.0f planning instances/s (ok %.3f)" % (T, dt / (K * T) * 1e3, B, B * K * T / dt, outs[0]["ok"].double().mean().item()))
