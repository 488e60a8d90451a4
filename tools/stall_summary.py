"""Condense a tools/stall_pmc.sh output directory into one stall_attribution.json.
Per occupancy (B) and kernel: the mean per-launch value of every counter collected, and the derived split of the resident wave time:

  SQ_WAVE_CYCLES (quad-cycles a wave is resident) ~= SQ_ACTIVE_INST_ANY (issuing) + SQ_WAIT_INST_ANY (has an instruction, cannot issue it: pipe busy / dependency)
                                                   + SQ_WAIT_ANY (parked: s_waitcnt, barrier, instruction fetch)          [MI355X_MICROARCH.md, "rocprofv3 PMC slots"]

plus the instruction-fetch side (SQ_IFETCH requests, SQC_ICACHE hit / miss, instruction requests to L2), the busy time per instruction type and the
instruction counts.  Usage: python tools/stall_summary.py gpurun_out/stall_<tag> profiles/<tag>/stall_attribution.json"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.split("(")[0]
    name = name.replace("void ", "")
    return name[:80]


def load(src):
    out = {}
    for bdir in sorted(glob.glob(os.path.join(src, "B*")), key=lambda p: int(os.path.basename(p)[1:])):
        B = int(os.path.basename(bdir)[1:])
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
        for f in glob.glob(os.path.join(bdir, "*", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1].add((f, r["Dispatch_Id"]))
        plain = ""
        pf = os.path.join(bdir, "plain.txt")
        if os.path.exists(pf):
            plain = [l for l in open(pf).read().splitlines() if l.startswith("STALL")]
            plain = plain[-1] if plain else ""
        out[B] = {"unprofiled_run": plain, "kernels": {k: {c: v[0] / max(len(v[1]), 1) for c, v in d.items()} for k, d in agg.items()}}
    return out


def derive(c):
    g = lambda k: c.get(k, 0.0)
    d = {}
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        d["resident_quad_cycles_per_wave"] = wc / max(g("SQ_WAVES"), 1.0)
        d["frac_issuing"] = g("SQ_ACTIVE_INST_ANY") / wc
        d["frac_wait_inst_any"] = g("SQ_WAIT_INST_ANY") / wc
        d["frac_wait_any"] = g("SQ_WAIT_ANY") / wc
        d["frac_wait_inst_lds"] = g("SQ_WAIT_INST_LDS") / wc
        d["insts_per_wave"] = g("SQ_INSTS") / max(g("SQ_WAVES"), 1.0)
        d["cycles_per_inst"] = 4.0 * wc / max(g("SQ_INSTS"), 1.0)
        for k in ("VALU", "SCA", "LDS", "VMEM", "FLAT", "MISC"):
            if ("SQ_ACTIVE_INST_" + k) in c:
                d["frac_active_" + k.lower()] = g("SQ_ACTIVE_INST_" + k) / wc
    if g("SQC_ICACHE_REQ"):
        d["icache_miss_rate"] = g("SQC_ICACHE_MISSES") / g("SQC_ICACHE_REQ")
        d["icache_miss_dup_rate"] = g("SQC_ICACHE_MISSES_DUPLICATE") / g("SQC_ICACHE_REQ")
        if g("SQ_IFETCH"):
            d["ifetch_mean_inflight_quad_cycles"] = g("SQ_IFETCH_LEVEL") / g("SQ_IFETCH")
    if g("SQC_DCACHE_REQ"):
        d["scalar_cache_miss_rate"] = g("SQC_DCACHE_MISSES") / g("SQC_DCACHE_REQ")
    if g("TCC_REQ_sum"):
        d["l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)
    if g("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_of_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    return d


def main(src, dst):
    data = load(src)
    for B, e in data.items():
        e["derived"] = {k: derive(c) for k, c in e["kernels"].items()}
    doc = {"what": "rocprofv3 --pmc passes (counters only, one group per pass) over tools/stall_child.py: headline setting, Newton launch forced into its two slices; "
                   "values are MEANS PER LAUNCH over the 4 solves of a pass; SQ_* time counters are quad-cycles summed over waves",
           "by_batch_size": {str(B): e for B, e in data.items()}}
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    json.dump(doc, open(dst, "w"), indent=1, sort_keys=True)
    for B, e in data.items():
        print(f"== B = {B}   {e['unprofiled_run']}")
        for k, d in sorted(e["derived"].items()):
            if not re.search("newton_kernel|solve_kernel_fast", k) or not d:
                continue
            print("  ", k)
            print("     ", " ".join(f"{a}={b:.4g}" for a, b in sorted(d.items())))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
