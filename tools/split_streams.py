"""Dev tool (GPU box): would an ENGINE-INTERNAL split of one batch over several HIP streams shorten a headline solve?
One launch sequence leaves wave slots idle at the end of every kernel (PH 1: 17 %, PH 2: 27 % of the slot time by rocprofv3 SQ_WAVE_CYCLES against the launch length,
profiles/r6a); sub-batches on their own streams fill each other's tails.  Emulated here with one engine per sub-batch (own stream, refine_chain = 3 so that no call blocks
the host), all issued back to back, one synchronise at the end; compared with the single engine at the same setting.  Shares: fractions of the 4096 paths per stream."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from path_optimizer_amd import binding, synth

CFG = int(os.environ.get("AB_CFG", "3"))
B = 4096
HEAD = dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8)


def params(chain):
    p = binding.default_params()
    for k, v in HEAD.items():
        setattr(p, k, v)
    p.refine_chain = chain
    return p


def run(shares, chain=3, reps=9):
    cuts = np.round(np.cumsum([0] + list(shares)) / sum(shares) * B).astype(int)
    dbs = [binding.DeviceBatch(synth.make_batch(CFG, B=int(hi - lo), first_path=int(lo))) for lo, hi in zip(cuts[:-1], cuts[1:])]
    engs = []
    for _ in dbs:
        e = binding.Engine(0, params(chain)); s = torch.cuda.Stream(); e.set_stream(s.cuda_stream); engs.append((e, s))
        if os.environ.get("SPLIT_FORCE_SLICE"):
            e.debug_set("newton_slice", int(os.environ["SPLIT_FORCE_SLICE"]))
    ts = []
    for r in range(3 + reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for (e, s), db in zip(engs, dbs):
            e.solve_batch_device(db)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    infos = np.concatenate([db.info_numpy() for db in dbs])
    return {"shares": list(shares), "chain": chain, "ms_median": float(np.median(ts[3:])), "ms_min": float(np.min(ts[3:])), "certified": int((infos["status_refine"] == 1).sum()),
            "iters_mean": float(infos["iters"].mean())}


if __name__ == "__main__":
    for sh, ch in (((1,), 2), ((1,), 3), ((1, 1), 3), ((1, 2), 3), ((2, 1), 3), ((1, 1, 1), 3), ((1, 2, 3), 3), ((1, 1, 1, 1), 3), ((3, 1), 3), ((1, 3), 3), ((7, 1), 3), ((1, 7), 3), ((4, 2, 1, 1), 3), ((1, 1, 2, 4), 3)):
        print(json.dumps(run(sh, ch)), flush=True)
