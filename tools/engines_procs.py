"""Dev tool (GPU box): where do E concurrent engines on ONE device serialise?
bench.py's `engines_on_pinned_threads` leg: E handles on E host THREADS, a 64-path batch each (the GPU side of one call is one short critical path on 64 of the 1 024
wave slots, so E <= 8 calls fit the chip side by side): 1.22 / 1.24 / 2.43 / 3.66 ms per call for E = 1 / 2 / 4 / 8.  This tool runs the same leg with
  * E PROCESSES (own HIP runtime, own hardware queues each):   flat in E  =>  the serialisation is inside one process (runtime locks / its hardware-queue pool),
                                                                 not flat   =>  the command processor / the device;
  * E threads under GPU_MAX_HW_QUEUES = 2 / 4 / 8 (ROCclr maps a process's streams onto that many hardware queues, default 4: streams that share a hardware queue run
    their kernels one after the other);
  * refine_chain = 3 (no 4-byte read-back inside the call) and the plain solve (three launches instead of seven).
    python tools/engines_procs.py            -> one JSON line per configuration"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CALLS = 40


def _params(binding, mode):
    p = binding.default_params()
    if mode != "plain":
        p.refine = 2; p.refine_rounds = 5; p.refine_extra_rounds = 2; p.refine_eps = 1e-8; p.refine_chain = 3 if mode == "chain3" else 2
    return p


def child(k, E, mode, gate_dir):
    import numpy as np
    import torch

    from path_optimizer_amd import binding, synth

    cores = sorted(os.sched_getaffinity(0))
    try:
        os.sched_setaffinity(0, {cores[k % len(cores)]})
    except OSError:
        pass
    eng = binding.Engine(0, _params(binding, mode))
    st = torch.cuda.Stream(); eng.set_stream(st.cuda_stream)
    db = binding.DeviceBatch(synth.make_batch(3, B=64))
    for _ in range(5):
        eng.solve_batch_device(db); st.synchronize()
    open(os.path.join(gate_dir, f"ready_{k}"), "w").close()
    t_end = time.time() + 60
    while not os.path.exists(os.path.join(gate_dir, "go")) and time.time() < t_end:
        time.sleep(0.0005)
    ts = []
    for _ in range(CALLS):
        t0 = time.perf_counter(); eng.solve_batch_device(db); st.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("CHILD " + json.dumps({"k": k, "median_ms": float(np.median(ts)), "min_ms": float(np.min(ts))}), flush=True)


def procs(E, mode, env=None):
    import tempfile

    with tempfile.TemporaryDirectory() as gate:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(k), str(E), mode, gate], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                               env=dict(os.environ, **(env or {}))) for k in range(E)]
        t_end = time.time() + 240
        while sum(os.path.exists(os.path.join(gate, f"ready_{k}")) for k in range(E)) < E and time.time() < t_end:
            time.sleep(0.01)
        open(os.path.join(gate, "go"), "w").close()
        res = []
        for p in ps:
            out, _ = p.communicate(timeout=120)
            for l in out.splitlines():
                if l.startswith("CHILD "):
                    res.append(json.loads(l[6:])["median_ms"])
    return res


def threads(E, mode):
    import threading

    import numpy as np
    import torch

    from path_optimizer_amd import binding, synth

    small = synth.make_batch(3, B=64)
    cores = sorted(os.sched_getaffinity(0))
    engs, strs, dbs = [], [], []
    for _ in range(E):
        e_ = binding.Engine(0, _params(binding, mode)); s_ = torch.cuda.Stream(); e_.set_stream(s_.cuda_stream)
        engs.append(e_); strs.append(s_); dbs.append(binding.DeviceBatch(small))
    res = [None] * E
    gate = threading.Barrier(E)

    def work(k):
        try:
            os.sched_setaffinity(threading.get_native_id(), {cores[k % len(cores)]})
        except OSError:
            pass
        for _ in range(5):
            engs[k].solve_batch_device(dbs[k]); strs[k].synchronize()
        gate.wait()
        ts = []
        for _ in range(CALLS):
            t0 = time.perf_counter(); engs[k].solve_batch_device(dbs[k]); strs[k].synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res[k] = float(np.median(ts))

    th = [threading.Thread(target=work, args=(k,)) for k in range(E)]
    [t.start() for t in th]; [t.join() for t in th]
    [e_.close() for e_ in engs]
    return res


def threads_host(E, mode):
    """E engines on E threads through the HOST-pointer entry (po_solve_batch: pack + H2D + solve + D2H), no torch.cuda call in this process: the library's own first HIP
    call starts the runtime, i.e. po_create's GPU_MAX_HW_QUEUES default is what the runtime sees."""
    import threading

    import numpy as np

    from path_optimizer_amd import binding, synth

    small = synth.make_batch(3, B=64)
    engs = [binding.Engine(0, _params(binding, mode)) for _ in range(E)]
    res = [None] * E
    gate = threading.Barrier(E)

    def work(k):
        for _ in range(5):
            engs[k].solve_batch(small)
        gate.wait()
        ts = []
        for _ in range(CALLS):
            t0 = time.perf_counter(); engs[k].solve_batch(small); ts.append((time.perf_counter() - t0) * 1e3)
        res[k] = float(np.median(ts))

    th = [threading.Thread(target=work, args=(k,)) for k in range(E)]
    [t.start() for t in th]; [t.join() for t in th]
    [e_.close() for e_ in engs]
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--threads-host":
        import numpy as np
        out = {}
        for E in (1, 2, 4, 8):
            out[str(E)] = round(float(np.median(threads_host(E, sys.argv[2]))), 3)
        print("THREADS " + json.dumps(out), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--threads":  # (own process: GPU_MAX_HW_QUEUES is read when the runtime starts)
        import numpy as np
        out = {}
        for E in (1, 2, 4, 8):
            r = threads(E, sys.argv[2])
            out[str(E)] = round(float(np.median(r)), 3)
        print("THREADS " + json.dumps(out), flush=True)
        sys.exit(0)
    import numpy as np
    for q in (None, "4"):  # the library's own default (16, set before its first HIP call) against the runtime's (4)
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        if q:
            env["GPU_MAX_HW_QUEUES"] = q
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--threads-host", "chain2"], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("THREADS ")]
        print(json.dumps({"what": "E threads, host-pointer entry, no torch.cuda in the process", "GPU_MAX_HW_QUEUES": q or "unset (po_create sets 16)", "per_call_ms": json.loads(line[0][8:]) if line else r.stderr[-300:]}), flush=True)
    if os.environ.get("ENGINES_QUICK"):
        sys.exit(0)
    for mode in ("chain2", "chain3", "plain"):
        row = {}
        for E in (1, 2, 4, 8):
            r = procs(E, mode)
            row[str(E)] = round(float(np.median(r)), 3) if r else None
        print(json.dumps({"what": "E processes", "mode": mode, "per_call_ms": row}), flush=True)
    for q in (None, "2", "4", "8", "16"):
        for mode in ("chain2", "chain3"):
            env = dict(os.environ)
            if q:
                env["GPU_MAX_HW_QUEUES"] = q
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--threads", mode], capture_output=True, text=True, env=env, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("THREADS ")]
            print(json.dumps({"what": "E threads", "GPU_MAX_HW_QUEUES": q or "default", "mode": mode, "per_call_ms": json.loads(line[0][8:]) if line else r.stderr[-300:]}), flush=True)
