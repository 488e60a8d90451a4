"""Dev tool (GPU box), the profiled command of tools/stall_pmc.sh: `python tools/stall_child.py <cfg> <B> [reps]` solves the first B paths of a BASELINE
config at the HEADLINE setting `reps` times (after one warm-up) with the Newton launch ALWAYS sliced in two (po_debug_set newton_slice 8: so that the
kernels under the counters are newton_kernel<..., 1> and <..., 2> whatever B is — left alone the engine slices only from 2 048 paths up).
B = 1: the path alone on the device; 256: about one path per CU; 1024: one wave per SIMD everywhere, no queueing; 4096: BASELINE config 3 as benchmarked."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from path_optimizer_amd import binding, synth

    cfg_s = sys.argv[1]; B = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cfg = int(cfg_s.rstrip("k"))  # "3k": config 3's generator with the K formulation (what bench.py's `k_formulation` leg runs)
    batch = synth.make_batch(cfg, B=B, **({"formulation": 2} if cfg_s.endswith("k") else {}))  # (path ids seed the generator: the first B paths of the config)
    db = binding.DeviceBatch(batch)
    p = binding.default_params()
    p.refine = 2; p.refine_rounds = 5; p.refine_extra_rounds = 2; p.refine_eps = 1e-8; p.refine_chain = 2
    eng = binding.Engine(0, p)
    if os.environ.get("STALL_NO_FORCE_SLICE") is None:
        eng.debug_set("newton_slice", 8)
    for _ in range(1 + reps):
        eng.solve_batch_device(db); torch.cuda.synchronize()
    info = db.info_numpy()
    print(f"STALL cfg {cfg_s} B {B} reps {reps} ms {eng.last_kernel_ms():.4f} phases {eng.last_phase_ms()} iters mean {info['iters'].mean():.2f} max {int(info['iters'].max())} "
          f"certified {int((info['status_refine'] == 1).sum())}", flush=True)


if __name__ == "__main__":
    main()
