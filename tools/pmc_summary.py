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
solve = (next((v for k, v in out.items() if "solve_kernel_fast<0, 4, 64, true, true, true" in k), None)  # the headline: uniform-row-class launch with the refinement phase (chained rounds)
         or next((v for k, v in out.items() if "solve_kernel_fast<0, 4, 64, true, true" in k), None)  # uniform-row-class launch: the one that does the work
         or next((v for k, v in out.items() if "solve_kernel_fast<0, 4, 64" in k), None) or next((v for k, v in out.items() if "solve_kernel" in k), None))  # the headline kernel (KP, SPL 4, one wave), not the pipeline legs' <0, 3, 128>
if solve and "FETCH_SIZE" in solve and "WRITE_SIZE" in solve:
    fetch = solve["FETCH_SIZE"]["per_dispatch"] * 1024.0
    write = solve["WRITE_SIZE"]["per_dispatch"] * 1024.0
    t = {"hbm_bytes_per_launch": 2 * fetch + write, "fetch_bytes_raw": fetch,
         "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)",
         "write_bytes": write, "write_note": "32.8 MB of outputs + register-spill scratch write-backs",
         "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
         "compulsory_io_bytes": 4096 * 8 * (18 * 200 + 8)}
    json.dump(t, open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic_latest.json"), "w"), indent=1)
    print(t)
if solve and "SQ_INSTS_VALU" in solve:
    # VALU wave-instructions per path-iteration: BASELINE config 3 = 4096 paths x 339.734 iterations per launch (po_info.iters, deterministic).
    # fp64 wave-instructions per path-iteration: static count of one plain iteration of the headline kernel (tools/isa_phase_hist.py; argv[3] if given).
    ref = any("solve_kernel_fast<0, 4, 64, true, true, true" in k for k in out)
    it_sum = 4096 * (171.25 if ref else 339.73388671875)  # headline setting (3 + 2 refinement rounds): 171.25 iterations per path, refinement iterations included
    f64 = {c: solve[c]["per_dispatch"] for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64") if c in solve}
    v = {"valu_wave_instr_per_path_iter": solve["SQ_INSTS_VALU"]["per_dispatch"] / it_sum,
         "fp64_wave_instr_per_path_iter": (sum(f64.values()) / it_sum) if f64 else (float(sys.argv[3]) if len(sys.argv) > 3 else None),
         "fp64_flop_per_launch": 64.0 * (f64["SQ_INSTS_VALU_ADD_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] + 2 * f64["SQ_INSTS_VALU_FMA_F64"] + f64["SQ_INSTS_VALU_TRANS_F64"]) if len(f64) == 4 else None,
         "kernel": "solve_kernel_fast<0, 4, 64, true, true, " + ("true>" if ref else "false>"),
         "occupancy_waves_per_simd": 1,
         "sq_wave_cycles_per_path_iter": solve.get("SQ_WAVE_CYCLES", {}).get("per_dispatch", 0) * 4 / it_sum,
         "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc SQ_INSTS_VALU ...) and the ISA histogram of the same tree (profiles/<tag>/phase_breakdown.txt)"}
    json.dump(v, open(os.path.join(os.path.dirname(dst.rstrip("/")), "valu_latest.json"), "w"), indent=1)
    print(v)
for k, d in out.items():
    if "solve_kernel" in k and "SQ_INSTS_VALU" in d:
        print("VALU wave-instr per launch", d["SQ_INSTS_VALU"]["per_dispatch"], "LDS instr", d["SQ_INSTS_LDS"]["per_dispatch"],
              "bank-conflict cycles", d["SQ_LDS_BANK_CONFLICT"]["per_dispatch"], "busy", d["SQ_BUSY_CYCLES"]["per_dispatch"])
