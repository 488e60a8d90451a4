"""Condense a tools/profile.sh output directory into profiles/<tag>/{kernel_stats.csv,pmc_summary.json,bench_under_rocprof.json}
and refresh profiles/traffic_latest.json (HBM bytes per launch of the solve kernel = 2*FETCH_SIZE + WRITE_SIZE, KB units,
gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md).  Usage: python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<tag>"""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_trace.json")):
    shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_f64"):
    for f in glob.glob(os.path.join(src, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
out = {k: {c: {"sum": v[0], "dispatches": len(v[1]), "per_dispatch": v[0] / max(len(v[1]), 1)} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
# ---- round 4: one solve at the headline setting is several kernels (warm-start launches, newton_kernel, newton_fallback_kernel, scale, status sweep): per-SOLVE totals ----
def _is_solve(k):
    return "solve_kernel_fast" in k or "newton_" in k or "scale_kernel" in k or "finalize_status" in k
nk = next((v for k, v in out.items() if "newton_kernel" in k), None)
n_solves = nk["SQ_INSTS_VALU"]["dispatches"] if nk and "SQ_INSTS_VALU" in nk else (next(iter(nk.values()))["dispatches"] if nk else 0)
def per_solve(counter):
    return sum(d[counter]["sum"] for k, d in out.items() if _is_solve(k) and counter in d) / max(n_solves, 1)
if n_solves and any("FETCH_SIZE" in d for d in out.values()):
    fetch, write = per_solve("FETCH_SIZE") * 1024.0, per_solve("WRITE_SIZE") * 1024.0
    t = {"hbm_bytes_per_launch": 2 * fetch + write, "fetch_bytes_raw": fetch,
         "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)",
         "write_bytes": write, "write_note": "32.8 MB of outputs + the state block handed from the warm-start launch to newton_kernel + register-spill scratch write-backs",
         "per": "one solve of BASELINE config 3 at the headline setting = every kernel of the solve summed (warm-start launches + newton_kernel + fallback + scale + status sweep)",
         "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
         "compulsory_io_bytes": 4096 * 8 * (18 * 200 + 8)}
    json.dump(t, open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic_latest.json"), "w"), indent=1)
    print(t)
if n_solves and nk and "SQ_INSTS_VALU_FMA_F64" in nk:
    flop = lambda get: 64.0 * (get("SQ_INSTS_VALU_ADD_F64") + get("SQ_INSTS_VALU_MUL_F64") + 2 * get("SQ_INSTS_VALU_FMA_F64") + get("SQ_INSTS_VALU_TRANS_F64"))
    v = {"fp64_flop_per_solve_headline_c3_b4096": flop(per_solve),
         "valu_wave_instr_per_solve": per_solve("SQ_INSTS_VALU"),
         "fp64_wave_instr_per_solve": sum(per_solve(c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")),
         "newton_kernel": {"fp64_flop_per_launch": flop(lambda c: nk[c]["per_dispatch"]), "valu_wave_instr_per_launch": nk["SQ_INSTS_VALU"]["per_dispatch"],
                           "sq_wave_cycles_per_launch_x4": nk.get("SQ_WAVE_CYCLES", {}).get("per_dispatch", 0) * 4},
         "per_path": {"newton_steps_mean_c3": 16.4, "admm_iterations": 25, "valu_wave_instr_per_newton_step": nk["SQ_INSTS_VALU"]["per_dispatch"] / (4096 * 16.4)},
         "occupancy_waves_per_simd": 1,
         "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc SQ_INSTS_VALU ... , separate passes) and the ISA histogram of the same tree (profiles/<tag>/phase_breakdown.txt)"}
    json.dump(v, open(os.path.join(os.path.dirname(dst.rstrip("/")), "valu_latest.json"), "w"), indent=1)
    print(v)
for k, d in out.items():
    if _is_solve(k) and "SQ_INSTS_VALU" in d:
        print(k[:60], "VALU wave-instr per launch", d["SQ_INSTS_VALU"]["per_dispatch"], "LDS instr", d.get("SQ_INSTS_LDS", {}).get("per_dispatch"),
              "bank-conflict cycles", d.get("SQ_LDS_BANK_CONFLICT", {}).get("per_dispatch"), "busy", d.get("SQ_BUSY_CYCLES", {}).get("per_dispatch"))
