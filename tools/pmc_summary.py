"""Condense a tools/profile.sh output directory into profiles/<tag>/{kernel_stats.csv,pmc_summary.json,bench_under_rocprof.json}
and refresh profiles/traffic_latest.json / profiles/valu_latest.json (the fallbacks bench.py uses when its live rocprofv3 passes are unavailable).

HBM bytes per solve = 2 x FETCH_SIZE + WRITE_SIZE (KB units; gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md).

Round 5: the profiled command runs the SAME kernel name at two settings — `solve_kernel_fast<0,4,64,true,true,0>` is the 25-iteration warm start of a headline solve
AND the whole OSQP-faithful solve of the `osqp_default` leg (340 iterations) — so launches are attributed to the SOLVE they belong to, not to their name: the
dispatches of one pass are walked in order, every `scale_kernel` starts a solve, and a solve that contains a `newton_kernel` launch is a headline solve.  Per-solve
figures are sums over the headline solves only (round 4's tool averaged both settings: 9.3e10 flop per solve instead of the 4.5e10 the run measures).
Usage: python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<tag>"""
import collections, csv, glob, json, os, shutil, sys


def short(name):
    return name.split("(")[0][:70]


def solves_of(rows):
    """rows: one pass's counter rows.  Returns (per_dispatch {id: (kernel, {counter: value})}, groups [[ids]], headline flags)."""
    disp = {}
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), [short(r["Kernel_Name"]), collections.defaultdict(float)])
        d[1][r["Counter_Name"]] += float(r["Counter_Value"])
    groups, cur = [], None
    for i in sorted(disp):
        k = disp[i][0]
        if "scale_kernel" in k:
            cur = []
            groups.append(cur)
        if cur is not None and ("solve_kernel" in k or "newton_" in k or "scale_kernel" in k or "finalize_status" in k or "polish_kernel" in k):
            cur.append(i)
    head = [any("newton_kernel" in disp[i][0] for i in g) for g in groups]
    return disp, groups, head


def summarise(src):
    """{setting: {kernel: {counter: {sum, dispatches, per_dispatch}}}}, {setting: n_solves} over every PMC pass under src."""
    agg = {"headline": collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0])), "other": collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))}
    n_solves = {"headline": {}, "other": {}}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_f64"):
        for f in glob.glob(os.path.join(src, sub, "*counter_collection.csv")):
            disp, groups, head = solves_of(list(csv.DictReader(open(f))))
            for g, h in zip(groups, head):
                tag = "headline" if h else "other"
                n_solves[tag][sub] = n_solves[tag].get(sub, 0) + 1
                for i in g:
                    k, cs = disp[i]
                    for c, v in cs.items():
                        a = agg[tag][k][c]
                        a[0] += v; a[1] += 1
    out = {tag: {k: {c: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for c, v in d.items()} for k, d in a.items()} for tag, a in agg.items()}
    return out, n_solves


def per_solve(out, n_solves, tag, counter, sub):
    n = n_solves[tag].get(sub, 0)
    return sum(d[counter]["sum"] for d in out[tag].values() if counter in d) / n if n else 0.0


F64 = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")


def flop_of(get):
    return 64.0 * (get("SQ_INSTS_VALU_ADD_F64") + get("SQ_INSTS_VALU_MUL_F64") + 2 * get("SQ_INSTS_VALU_FMA_F64") + get("SQ_INSTS_VALU_TRANS_F64"))


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "bench_trace.json")):
        shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))
    out, n_solves = summarise(src)
    json.dump({"attribution": "launches grouped by the solve they belong to (scale_kernel starts one; a solve with a newton_kernel launch is a headline solve)",
               "solves_per_pass": n_solves, **out}, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    top = os.path.dirname(dst.rstrip("/"))
    H = "headline"
    # the Newton refinement of one solve = its newton_kernel launches together (round 5: two, the sliced pair <..., 1> and <..., 2>): per-launch figures summed over the variants
    nk = None
    for k, v in out[H].items():
        if "newton_kernel" in k:
            nk = nk or collections.defaultdict(lambda: {"per_dispatch": 0.0})
            for c, d in v.items():
                nk[c]["per_dispatch"] += d["per_dispatch"]
    if n_solves[H].get("pmc_fetch") and n_solves[H].get("pmc_write"):
        fetch, write = per_solve(out, n_solves, H, "FETCH_SIZE", "pmc_fetch") * 1024.0, per_solve(out, n_solves, H, "WRITE_SIZE", "pmc_write") * 1024.0
        t = {"hbm_bytes_per_launch": 2 * fetch + write, "fetch_bytes_raw": fetch,
             "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)",
             "write_bytes": write, "write_note": "32.8 MB of outputs + the state block handed from the warm-start launch to newton_kernel (147 MB) + the parked paths' blocks between the two Newton launches (~250 MB) + register-spill scratch write-backs",
             "per": "one solve of BASELINE config 3 at the headline setting = every kernel of that solve summed (scale + warm-start launches + newton_kernel + fallback + status sweep); launches of the osqp_default leg excluded",
             "headline_solves_averaged": n_solves[H]["pmc_fetch"],
             "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
             "compulsory_io_bytes": 4096 * 8 * (18 * 200 + 8)}
        json.dump(t, open(os.path.join(top, "traffic_latest.json"), "w"), indent=1)
        print(t)
    if nk and n_solves[H].get("pmc_f64") and "SQ_INSTS_VALU_FMA_F64" in nk:
        g64 = lambda c: per_solve(out, n_solves, H, c, "pmc_f64")
        gsq = lambda c: per_solve(out, n_solves, H, c, "pmc_sq")
        v = {"fp64_flop_per_solve_headline_c3_b4096": flop_of(g64),
             "valu_wave_instr_per_solve": gsq("SQ_INSTS_VALU"),
             "fp64_wave_instr_per_solve": sum(g64(c) for c in F64),
             "headline_solves_averaged": n_solves[H]["pmc_f64"],
             "newton_kernel": {"launches_per_solve": sum(1 for k in out[H] if "newton_kernel" in k), "fp64_flop_per_launch": flop_of(lambda c: nk[c]["per_dispatch"]), "valu_wave_instr_per_launch": nk.get("SQ_INSTS_VALU", {}).get("per_dispatch", 0),
                               "sq_wave_cycles_per_launch_x4": nk.get("SQ_WAVE_CYCLES", {}).get("per_dispatch", 0) * 4,
                               "lds_bank_conflict_over_busy": (nk["SQ_LDS_BANK_CONFLICT"]["per_dispatch"] / nk["SQ_BUSY_CYCLES"]["per_dispatch"]) if "SQ_LDS_BANK_CONFLICT" in nk and nk.get("SQ_BUSY_CYCLES", {}).get("per_dispatch") else None},
             "osqp_default_solve": {"fp64_flop_per_solve": flop_of(lambda c: per_solve(out, n_solves, "other", c, "pmc_f64")), "solves_averaged": n_solves["other"].get("pmc_f64", 0)},
             "occupancy_waves_per_simd": 1,
             "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc, separate passes; launches attributed to the solve they belong to)"}
        json.dump(v, open(os.path.join(top, "valu_latest.json"), "w"), indent=1)
        print(v)
    for tag in out:
        for k, d in out[tag].items():
            if "SQ_INSTS_VALU" in d:
                print(tag, k[:60], "VALU wave-instr per launch", d["SQ_INSTS_VALU"]["per_dispatch"], "LDS instr", d.get("SQ_INSTS_LDS", {}).get("per_dispatch"),
                      "bank-conflict cycles", d.get("SQ_LDS_BANK_CONFLICT", {}).get("per_dispatch"), "busy", d.get("SQ_BUSY_CYCLES", {}).get("per_dispatch"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
