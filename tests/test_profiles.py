"""The committed profile summaries are self-consistent (CPU): the fallback bench.py would use when its live rocprofv3 passes fail
(profiles/valu_latest.json, profiles/traffic_latest.json — written by tools/pmc_summary.py) equals what the newest committed driver-style run measured live."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_live_bench():
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default*.json"))):
        try:
            b = json.load(open(f))
        except ValueError:
            continue
        rf = b.get("roofline") or {}
        if "measured in this run" in (rf.get("achieved_def") or "") and rf.get("achieved"):
            best = (f, b)
    return best


def test_committed_roofline_fallback_matches_the_live_measurement():
    f, b = _newest_live_bench()
    rf = b["roofline"]
    live_flop = rf["achieved"] * 1e12 * b["single_batch"]["median_ms"] * 1e-3  # fp64 flop per solve as the run's own PMC passes measured it
    v = json.load(open(os.path.join(ROOT, "profiles", "valu_latest.json")))
    fb = v["fp64_flop_per_solve_headline_c3_b4096"]
    assert abs(fb - live_flop) <= 0.10 * live_flop, (f, fb, live_flop)
    # the dominant kernel's share must be a share: the Newton launch cannot hold more flop than the whole solve
    assert 0.3 * fb < v["newton_kernel"]["fp64_flop_per_launch"] < fb
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert abs(t["hbm_bytes_per_launch"] - rf["traffic"]) <= 0.10 * rf["traffic"], (f, t["hbm_bytes_per_launch"], rf["traffic"])


def test_pmc_summary_attributes_launches_to_their_solve(tmp_path):
    """tools/pmc_summary.py on a synthetic counter file: the same kernel name at two settings (warm start of a headline solve / whole osqp_default solve) is split by
    the solve the launch belongs to."""
    import csv
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as S

    src = tmp_path / "prof"
    (src / "pmc_f64").mkdir(parents=True)
    rows, did = [], 0
    def launch(name, fma):
        nonlocal did
        did += 1
        for c, val in (("SQ_INSTS_VALU_ADD_F64", 0), ("SQ_INSTS_VALU_MUL_F64", 0), ("SQ_INSTS_VALU_FMA_F64", fma), ("SQ_INSTS_VALU_TRANS_F64", 0)):
            rows.append({"Dispatch_Id": did, "Kernel_Name": name, "Counter_Name": c, "Counter_Value": val})
    for _ in range(3):  # headline solves: warm start 10, Newton 30
        launch("void po::scale_kernel<0>(a)", 0); launch("void po::solve_kernel_fast<0, 4, 64, true, true, 0>(a)", 10); launch("void po::newton_kernel<0, 4, 64>(a)", 30); launch("po::finalize_status_kernel(a)", 0)
    for _ in range(2):  # osqp_default solves: the same kernel name, 100
        launch("void po::scale_kernel<0>(a)", 0); launch("void po::solve_kernel_fast<0, 4, 64, true, true, 0>(a)", 100)
    with open(src / "pmc_f64" / "x_counter_collection.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader(); w.writerows(rows)
    out, n = S.summarise(str(src))
    assert n["headline"]["pmc_f64"] == 3 and n["other"]["pmc_f64"] == 2
    assert S.flop_of(lambda c: S.per_solve(out, n, "headline", c, "pmc_f64")) == 64.0 * 2 * 40
    assert S.flop_of(lambda c: S.per_solve(out, n, "other", c, "pmc_f64")) == 64.0 * 2 * 100


def test_committed_kernel_resources_describe_the_objects_that_ship():
    """profiles/r5c/kernel_resources.txt listed 131 kernels of twelve objects the Makefile had stopped building (a glob over a box with leftovers).
    tools/kernel_resources.py now takes the object list from the Makefile's link line; the newest committed listing names exactly objects on that line, and every
    solve / Newton object of the line appears in it."""
    import re
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as K

    linked = set(K.linked_objects())
    assert {"po_kernels.o", "po_newton_kp.o", "po_solve_kp_uni.o", "po_capi.o"} <= linked
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r6*", "kernel_resources.txt")))
    assert files, "no round-6 kernel_resources.txt committed"
    rows = [l.split()[0] for l in open(files[-1]).read().splitlines()[1:] if l.strip()]
    named = set(rows)
    assert named <= linked, sorted(named - linked)
    mk = open(os.path.join(ROOT, "path_optimizer_amd", "csrc", "Makefile")).read()
    solve_objs = set(re.findall(r"\$\(BUILD\)/(po_(?:solve|newton)_\w+\.o)", re.search(r"^SOLVE_OBJS :=(.*?)\n\n", mk, flags=re.S | re.M).group(1)))
    assert solve_objs and solve_objs <= named, sorted(solve_objs - named)
