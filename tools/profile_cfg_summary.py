"""Condense a tools/profile_cfg.sh directory: kernel_stats.csv (copied), pmc_summary.json = per kernel the mean per-launch counters, and a roofline block for the
Newton kernels of that config (fp64 flop from the instruction counters / the launches' time from the kernel trace; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE in KB, the
gfx950 correction of MI355X_MICROARCH.md).  Usage: python tools/profile_cfg_summary.py gpurun_out/prof_<tag>/c<cfg> <cfg> <B>"""
import collections
import csv
import glob
import json
import os
import sys

PEAK_TF = 78.6


def short(n):
    return n.split("(")[0].replace("void ", "")[:80]


def main(src, cfg, B):
    ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    stats = {}
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open(os.path.join(src, "kernel_stats.csv"), "w") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
            for r in rows:
                r["Name"] = short(r["Name"]); w.writerow(r)
        for r in rows:
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1].add((f, r["Dispatch_Id"]))
    pmc = {k: {c: v[0] / max(len(v[1]), 1) for c, v in d.items()} for k, d in agg.items() if any(s in k for s in ("newton", "solve_kernel", "scale_kernel", "nw_sort", "finalize"))}
    out = {"config": cfg, "B": B, "what": "headline setting, tools/stall_child.py (1 warm-up + 5 solves per pass); counters are means per launch", "kernel_stats": {k: v for k, v in stats.items() if k in pmc},
           "pmc_mean_per_launch": pmc}
    nk = [k for k in pmc if "newton_kernel" in k]
    flop = sum(64.0 * (pmc[k].get("SQ_INSTS_VALU_ADD_F64", 0) + pmc[k].get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * pmc[k].get("SQ_INSTS_VALU_FMA_F64", 0) + pmc[k].get("SQ_INSTS_VALU_TRANS_F64", 0)) for k in nk)
    t_ms = sum(stats[k]["avg_us"] for k in nk if k in stats) / 1e3
    hbm = sum(2e3 * pmc[k].get("FETCH_SIZE", 0) + 1e3 * pmc[k].get("WRITE_SIZE", 0) for k in nk)
    if nk and t_ms > 0:
        out["roofline_newton_kernels"] = {"kernels": nk, "bound": "fp64_valu", "fp64_flop_per_solve": flop, "ms_per_solve": t_ms, "achieved_tflops": flop / t_ms / 1e9, "peak_tflops": PEAK_TF,
                                          "frac": flop / t_ms / 1e9 / PEAK_TF, "hbm_bytes_per_solve": hbm, "hbm_gbs": hbm / t_ms / 1e6}
        for k in nk:
            c = pmc[k]
            if c.get("SQ_WAVE_CYCLES"):
                out.setdefault("resident_time_split", {})[k] = {"issuing": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], "wait_any": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"],
                                                                "wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], "cycles_per_inst": 4 * c["SQ_WAVE_CYCLES"] / max(c.get("SQ_INSTS", 1), 1),
                                                                "slot_busy_of_launch": (4 * c["SQ_WAVE_CYCLES"] / 1024.0) / (c.get("GRBM_GUI_ACTIVE", 0) / 8.0) if c.get("GRBM_GUI_ACTIVE") else None}
    all_flop = sum(64.0 * (c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0)) * stats.get(k, {}).get("calls", 0) for k, c in pmc.items())
    out["plain_run"] = [l for l in open(os.path.join(src, "plain.txt")).read().splitlines() if l.startswith("STALL")][-1:] if os.path.exists(os.path.join(src, "plain.txt")) else []
    json.dump(out, open(os.path.join(src, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: out.get(k) for k in ("config", "roofline_newton_kernels", "resident_time_split", "plain_run")}, indent=1))
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
        print(f"  {k:70s} calls {v['calls']:4d} avg {v['avg_us']:10.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
