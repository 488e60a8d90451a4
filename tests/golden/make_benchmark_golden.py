"""Generator of tests/golden/benchmark_scene.npz: BASELINE config 1 on the reference's REAL benchmark scene — the only input fixture the reference
holds (/root/reference/src/test/path_optimizer_benchmark.cpp).

  * map: obstacles_for_benchmark.png -> grid map exactly as :28-44 does it.  cv::imread(CV_8UC1) (the image holds 0 / 255 only, so the grey
    conversion is the identity), GridMapCvConverter::initializeFromImage(img, 0.2, map, Position::Zero()) (length = rows x cols x 0.2 m, centre 0),
    addLayerFromImage(OCCUPY = 0, FREE = 255): layer(i, j) = image(i, j); cv::distanceTransform(L2, MASK_PRECISE) = the exact Euclidean distance
    to the nearest zero pixel in float32 (scipy.ndimage.distance_transform_edt, cast to float32), then `*= resolution` in float32.
  * 100 way points :47-66, start / goal states :75-82 (typed in below: they are the benchmark's inputs).
  * outputs: the reference-compiled PathOptimizer (oracle/_ref/libpo_ref_smooth.so = the reference's own path_optimizer.cpp, smoothers, ReferencePath,
    solver, collision checker; OSQP stood in by the oracle's ADMM, tinyspline by the restated clamped B-spline) running BM_optimizePath's solve() and
    BM_optimizePathWithoutSmoothing's solveWithoutSmoothing() on the same object, at OSQP's default eps 1e-3 (what the reference runs) and at this
    project's 1e-4.

    python tests/golden/make_benchmark_golden.py        (build container only: needs /root/reference)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

X_LIST = [36.933, 35.664, 34.5232, 33.5006, 32.5863, 31.7711, 31.0461, 30.4029, 29.8334, 29.33, 28.8857, 28.4938,
          28.1478, 27.8421, 27.5711, 27.3299, 27.1139, 26.919, 26.7415, 26.5781, 26.4261, 26.283, 26.1468, 26.016,
          25.8895, 25.7666, 25.6471, 25.5308, 25.4176, 25.3073, 25.1998, 25.0951, 24.9929, 24.8933, 24.7961, 24.7011,
          24.6084, 24.5178, 24.4292, 24.3425, 24.2578, 24.1748, 24.0936, 24.0141, 23.9361, 23.8597, 23.7848, 23.7114,
          23.6394, 23.5687, 23.4994, 23.4314, 23.3647, 23.2992, 23.235, 23.172, 23.1101, 23.0493, 22.9897, 22.9312,
          22.8738, 22.8174, 22.762, 22.7076, 22.6542, 22.6018, 22.5504, 22.4998, 22.4502, 22.4015, 22.3536, 22.3066,
          22.2605, 22.2151, 22.1707, 22.127, 22.0841, 22.042, 22.0007, 21.9603, 21.9208, 21.8821, 21.8445, 21.8079,
          21.7724, 21.7381, 21.7051, 21.6736, 21.6436, 21.6153, 21.5888, 21.5642, 21.5418, 21.5217, 21.5042, 21.4893,
          21.4773, 21.4685, 21.463, 21.4611]
Y_LIST = [33.6609, 30.1924, 27.1101, 24.3825, 21.9795, 19.8724, 18.0336, 16.437, 15.0581, 13.8733, 12.8606, 11.9994,
          11.2702, 10.6552, 10.1376, 9.70216, 9.3349, 9.02324, 8.7559, 8.52298, 8.31592, 8.1275, 7.95186, 7.78447,
          7.62217, 7.46313, 7.30673, 7.15283, 7.00127, 6.85193, 6.70466, 6.55933, 6.41578, 6.27389, 6.13352, 5.99451,
          5.85674, 5.72006, 5.58434, 5.44943, 5.31518, 5.18147, 5.04815, 4.91508, 4.78211, 4.64912, 4.51595, 4.38246,
          4.24852, 4.11398, 3.9787, 3.84254, 3.70538, 3.5671, 3.4276, 3.28681, 3.14465, 3.00106, 2.85602, 2.70948,
          2.56145, 2.41193, 2.26093, 2.10849, 1.95465, 1.79949, 1.64306, 1.48548, 1.32684, 1.16726, 1.00687,
          0.845838, 0.684314, 0.522481, 0.360532, 0.198675, 0.0371402, -0.123809, -0.283872, -0.442713, -0.599958,
          -0.755201, -0.907996, -1.05786, -1.20428, -1.3467, -1.48454, -1.61716, -1.7439, -1.86408, -1.97694,
          -2.08173, -2.17764, -2.26383, -2.33941, -2.40347, -2.45507, -2.49321, -2.51688, -2.52501]
START = [36.933, 33.6609, -1.36375, 0.0]  # x, y, z, k
GOAL = [21.4611, -2.52501, -1.30825]
RESOLUTION = 0.2


def benchmark_map(png="/root/reference/obstacles_for_benchmark.png"):
    """(distance [rows, cols] float32, resolution, pos_x, pos_y) in the layout of po_map / synth.make_distance_map."""
    from PIL import Image
    from scipy.ndimage import distance_transform_edt

    img = np.array(Image.open(png).convert("L"))  # 495 rows x 497 columns, values 0 / 255
    assert set(np.unique(img)) <= {0, 255}
    edt = distance_transform_edt(img != 0).astype(np.float32)  # cv::distanceTransform(L2, PRECISE): float32 distance to the nearest zero pixel
    dist = edt * np.float32(RESOLUTION)                        # grid_map.get("distance") *= resolution  (MatrixXf: float arithmetic)
    return dist.astype(np.float32), RESOLUTION, 0.0, 0.0


def main():
    from oracle import oracle_py as o, ref_py as r

    dist, res, px, py = benchmark_map()
    mp = o.make_map(dist, res, px, py)
    out = dict(distance=dist, resolution=res, pos=np.array([px, py]), way_x=np.array(X_LIST), way_y=np.array(Y_LIST), start=np.array(START), goal=np.array(GOAL))
    for tag, eps in (("e3", 1e-3), ("e4", 1e-4)):
        p = o.default_params()
        p.eps_abs = p.eps_rel = eps
        g = r.benchmark_scene(mp, p, X_LIST, Y_LIST, START, GOAL)
        assert g["ok1"] and g["ok2"], (tag, g["ok1"], g["ok2"])
        for k in ("path1", "path2", "knot_s", "knot_x", "knot_y"):
            out[f"{k}_{tag}"] = g[k]
        out[f"max_s_{tag}"] = g["max_s"]
        out[f"qp_{tag}"] = np.array([[g[q]["status"], g[q]["iters"], g[q]["n_refactor"]] for q in ("qp1", "qp2")])
        print(tag, "solve: n", len(g["path1"]), g["qp1"], "| without smoothing: n", len(g["path2"]), g["qp2"], "| knots", len(g["knot_s"]), "max_s", g["max_s"])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "benchmark_scene.npz"), **out)


if __name__ == "__main__":
    main()
