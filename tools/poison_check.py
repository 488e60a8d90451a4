"""Dev tool (GPU box): do the solve kernels read a REGISTER (or an LDS word) they never wrote?
VGPRs, AccVGPRs and LDS are not cleared between waves / workgroups, so such a read normally sees harmless leftovers of the same kernel and survives every parity test — until a
one-line change moves the register allocation (round 6: a condition that is never true put NaNs into the fall-back rounds of the ragged test batch).
tools/ubench/poison.hip fills every VGPR / AccVGPR of every lane and all 160 KB of LDS on every CU with a pattern; this tool solves each case after a NaN-payload poison and after
a zero poison (and once more after the NaN poison): results must be BITWISE equal.  Cases: the headline shape (sliced, unsliced), the ragged batch whose infeasible paths go
through the fall-back rounds, KPC, K, role-split and multi-group shapes, the OSQP-faithful solve, the polish.
What it can and cannot see (round 6, second session): the poison reaches the FIRST kernel of a solve only — every later kernel sees the leftovers of the kernels
before it — so for the Newton / fall-back kernels the three runs differ by which hardware wave slot a path lands on (the poison launch shifts the dispatcher), not by the pattern.
That was enough to expose the bug of section 13 (a wave-uniform scalar lost on the last lane of paths with n_points % 4 != 0: run 2 differed from runs 0 and 1), but a "SAME" is a
sample, not a proof.  POISON_ONLY=<substring> runs the matching cases only; a DIFFER line says which output differs between which runs, and where.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/.bin/libpoison.so tools/ubench/poison.hip ; python tools/poison_check.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

NAN_PATTERN = 0x7FF8DEAD7FF8BEEF  # both halves of every 64-bit pair are quiet-NaN bit patterns as doubles AND as floats
HEAD = dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8, refine_chain=2)


def cases():
    import np_twin as T
    from path_optimizer_amd import synth

    def rand(keep, N, B, seed):
        rng = np.random.default_rng(seed)
        insts = [T.random_instance(rng, N, ds=1.2 / keep * 0.999) for _ in range(B)]
        st = lambda k: np.ascontiguousarray(np.stack([i[k] for i in insts]))
        return synth.Batch(0, B, N, keep, st("ref_x"), st("ref_y"), st("ref_z"), st("ref_k"), st("ref_s"), st("bounds"), st("x0"), np.array([i["goal_z"] for i in insts]))

    rag = synth.make_batch(3, B=333)
    rag.n_points = np.random.default_rng(5).integers(60, 201, size=333).astype(np.int32)
    yield "c3 headline, engine's slicing (B 2048)", synth.make_batch(3, B=2048), HEAD, None
    yield "c3 headline, unsliced", synth.make_batch(3, B=700), HEAD, 0
    yield "c3 headline, sliced 8", synth.make_batch(3, B=700), HEAD, 8
    yield "c3 ragged (14 infeasible paths -> fall-back rounds), unsliced", rag, HEAD, 0
    yield "c3 ragged, sliced 3", rag, HEAD, 3
    yield "c3 OSQP-faithful", synth.make_batch(3, B=512), {}, None
    yield "c3 OSQP-faithful + polish", synth.make_batch(3, B=256), dict(polish=1), None
    yield "c5 KPC headline", synth.make_batch(5, B=192), HEAD, None
    yield "c5 KPC sliced 8", synth.make_batch(5, B=96), HEAD, 8
    yield "K headline sliced 8", synth.make_batch(3, B=300, formulation=2), HEAD, 8
    yield "K OSQP-faithful", synth.make_batch(3, B=200, formulation=2), {}, None
    yield "c2 headline", synth.make_batch(2, B=512), HEAD, None
    for keep in (1, 2, 3, 5, 6, 7, 8, 10, 12, 15):
        yield f"keep {keep} headline sliced 8", rand(keep, 150, 64, 7 + keep), HEAD, 8
    yield "keep 12 headline unsliced", rand(12, 150, 64, 19), HEAD, 0
    yield "keep 17 (single-level chain) OSQP-faithful", rand(17, 120, 16, 40), {}, None


def main():
    import torch  # noqa: F401  (one HIP runtime for both libraries)

    from path_optimizer_amd import binding

    P = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", ".bin", "libpoison.so"))
    P.po_poison_what.argtypes = [ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_int]
    what = int(os.environ.get("POISON_WHAT", "3"))  # 1 registers only, 2 LDS only, 3 both
    bad = 0
    only = os.environ.get("POISON_ONLY", "")
    for name, b, kw, sl in cases():
        if only and only not in name:
            continue
        res = []
        for pattern in (NAN_PATTERN, 0, NAN_PATTERN):
            p = binding.default_params()
            for k, v in kw.items():
                setattr(p, k, v)
            e = binding.Engine(0, p)
            if sl is not None:
                e.debug_set("newton_slice", sl)
            assert P.po_poison_what(pattern, None, what) == 0
            st, info, xs = e.solve_batch(b, want_x=True)
            res.append((st.copy(), info.copy(), xs.copy()))
            e.close()
        same = all(np.array_equal(res[0][j].view(np.uint8), res[i][j].view(np.uint8)) for i in (1, 2) for j in (0, 2)) and all(
            res[0][1].tobytes() == res[i][1].tobytes() for i in (1, 2))
        diffs = ""
        if not same:
            bad += 1
            d = np.flatnonzero([(not np.array_equal(res[0][2][q].view(np.uint64), res[1][2][q].view(np.uint64))) or res[0][1][q].tobytes() != res[1][1][q].tobytes() for q in range(b.B)])
            diffs = f" paths that differ (NaN poison vs zero poison): {d.tolist()[:12]} statuses {res[0][1]['status'][d][:8].tolist()} vs {res[1][1]['status'][d][:8].tolist()}"
            for i in (1, 2):  # which output differs between which runs, and where
                for j, nm in ((0, "states"), (2, "x")):
                    a, c = res[0][j].view(np.uint64).reshape(b.B, -1), res[i][j].view(np.uint64).reshape(b.B, -1)
                    rows = np.flatnonzero((a != c).any(axis=1))
                    if len(rows):
                        q = int(rows[0]); cols = np.flatnonzero(a[q] != c[q])
                        diffs += f" | run 0 vs {i}: {nm} differ on {len(rows)} paths, first {rows[:8].tolist()}; path {q} (n_points {int(b.n_points[q]) if getattr(b, 'n_points', None) is not None else b.N}, status {int(res[0][1]['status'][q])}, refine {int(res[0][1]['status_refine'][q])}): words {cols[:10].tolist()} of {a.shape[1]}, e.g. {res[0][j].reshape(b.B, -1)[q][cols[:3]].tolist()} vs {res[i][j].reshape(b.B, -1)[q][cols[:3]].tolist()}"
                if res[0][1].tobytes() != res[i][1].tobytes():
                    diffs += f" | run 0 vs {i}: info differs"
        print(("SAME   " if same else "DIFFER ") + name + diffs, flush=True)
    print("cases that differ:", bad)
    return bad


def placement():
    """The same solve on different PARTS of the chip (HIP streams with a CU mask: all CUs, two complementary halves, every fourth CU) and after an unrelated batch has run: a
    path then lands on other wave slots, behind other leftovers in registers, LDS and scratch.  Results must be bitwise equal.  (The bug of DESIGN.md section 13 was of this kind:
    what a scratch slot of the path's last lane held decided the status.)"""
    import torch  # noqa: F401

    from path_optimizer_amd import binding, synth

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32

    def masked_stream(bits):
        m = (ctypes.c_uint32 * words)(*[sum(1 << k for k in range(32) if 32 * w + k < ncu and bits(32 * w + k)) for w in range(words)])
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, m)
        assert rc == 0 and st.value, rc
        return st

    masks = [("all CUs", None), ("first half", lambda c: c < ncu // 2), ("second half", lambda c: c >= ncu // 2), ("every fourth CU", lambda c: c % 4 == 1), ("all CUs, after another batch", None)]
    other = synth.make_batch(3, B=1500)
    bad = 0
    wanted = ("c3 ragged", "c3 headline, unsliced", "c5 KPC headline", "K headline", "keep 3 ", "keep 6 ", "keep 12 headline sliced", "c3 OSQP-faithful")
    only = os.environ.get("POISON_ONLY", "")
    for name, b, kw, sl in cases():
        if not name.startswith(wanted) or (only and only not in name):
            continue
        res = []
        for label, bits in masks:
            p = binding.default_params()
            for k, v in kw.items():
                setattr(p, k, v)
            e = binding.Engine(0, p)
            if sl is not None:
                e.debug_set("newton_slice", sl)
            st_ = None
            if bits is not None:
                st_ = masked_stream(bits)
                e.set_stream(st_.value)
            if label.endswith("another batch"):
                e.solve_batch(other)
            st, info, xs = e.solve_batch(b, want_x=True)
            res.append((label, st.copy(), info.copy(), xs.copy()))
            e.close()
            if st_ is not None:
                hip.hipStreamDestroy(st_)
        diff = [r[0] for r in res[1:] if not (np.array_equal(r[1].view(np.uint64), res[0][1].view(np.uint64)) and np.array_equal(r[3].view(np.uint64), res[0][3].view(np.uint64)) and r[2].tobytes() == res[0][2].tobytes())]
        bad += bool(diff)
        extra = ""
        if diff:
            r = next(r for r in res[1:] if r[0] == diff[0])
            rows = np.flatnonzero((r[3].view(np.uint64) != res[0][3].view(np.uint64)).any(axis=1) | (r[2]["status"] != res[0][2]["status"]))
            extra = f" differs on: {diff}; first paths {rows[:8].tolist()} statuses {res[0][2]['status'][rows[:8]].tolist()} vs {r[2]['status'][rows[:8]].tolist()}"
        print(("SAME   " if not diff else "DIFFER ") + "placement: " + name + extra, flush=True)
    print("placement cases that differ:", bad)
    return bad


if __name__ == "__main__":
    rc = main()
    if os.environ.get("POISON_PLACEMENT", "1") != "0":
        rc += placement()
    sys.exit(1 if rc else 0)
