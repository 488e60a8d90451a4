"""Dev tool (GPU box): how often do the INDEX outputs of the map stages differ between device and oracle?  
bounds producer: n_valid per path, bound values; DP search: n_layers, corridors; post-check: n_valid / ok."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_bounds_golden as GB  # noqa: E402
import make_post_golden as GP  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from path_optimizer_amd import binding, synth  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
O.set_portable_math(len(sys.argv) > 2 and sys.argv[2] == "portable")
print("oracle portable math:", O.lib().po_oracle_get_portable_math())
d, res, px, py, _ = synth.make_distance_map(**GP.MAP_ARGS)
m = O.make_map(d, res, px, py)
eng = binding.Engine(0)
eng.set_map(d, res, px, py)
P = synth.make_spline_paths(GB.SEED + 1, nb, GB.N)
bd, nv = eng.bounds_batch(P)
p = O.default_params()
onv = np.zeros(nb, np.int32); ob = np.zeros_like(bd)
for b in range(nb):
    ob[b], onv[b] = O.bounds_path(p, m, *[P[k][b] for k in GB.KEYS])
same = nv == onv
diff = np.abs(bd - ob)
print(f"bounds: paths {nb}  n_valid equal {same.sum()}  | entries bit-identical {np.mean(bd[same] == ob[same]):.6f}  max|d| {diff[same].max():.3e}  entries > 1e-9: {(diff[same] > 1e-9).sum()} of {diff[same].size}")
print("   n_valid mismatches (dev, oracle):", list(zip(nv[~same].tolist(), onv[~same].tolist()))[:20])
MAP_KW = dict(size_x=600, size_y=600, resolution=0.2, pos=(1.0, -2.0), n_obstacles=40, r_range=(0.5, 2.0))
dm = synth.make_distance_map(3, **MAP_KW)
om = O.make_map(*dm[:4]); eng.set_map(*dm[:4])
sp, length, start = synth.make_search_inputs(16, nb)
ls, lb, ub, l0, nl = eng.dp_search_batch(sp, length, start, 64)
ident = nlsame = 0; bad = []
for b in range(nb):
    n, ols, olb, oub, ol0 = O.dp_search(p, om, sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b], length[b], start[b], cap=64)
    nlsame += nl[b] == n
    if n < 0 or nl[b] != n:
        ident += nl[b] == n
        continue
    e = max(np.abs(lb[b, :n] - olb).max(), np.abs(ub[b, :n] - oub).max(), np.abs(ls[b, :n] - ols).max())
    bit = np.array_equal(lb[b, :n], olb) and np.array_equal(ub[b, :n], oub) and np.array_equal(ls[b, :n], ols) and l0[b] == ol0
    nbit = nbit + bit if "nbit" in dir() else int(bit)
    ident += e < 1e-9
    if e >= 1e-9: bad.append((b, float(e)))
print(f"dp_search: paths {nb}  n_layers equal {nlsame}  corridors within 1e-9: {ident}  bit-identical (layers, corridor, offset): {nbit}  differing: {bad[:10]}")
out = eng.resample_batch(sp, length, 0.15, 0.3, 320)
rb = 0
for b in range(nb):
    n, oo = O.resample(p, sp["knot_s"][b], sp["knot_x"][b], sp["knot_y"][b], length[b], 0.15, 0.3, cap=320)
    rb += out["n_points"][b] == n and all(np.array_equal(out[k][b, :n], ov) for k, ov in zip(("ref_x", "ref_y", "ref_z", "ref_k", "ref_s"), oo))
print(f"resample: paths {nb}  bit-identical states: {rb}")
