"""Golden vectors for the radius query and the on-disk format of Metric_Cache, generated with
the reference's own dependency: sklearn.neighbors.BallTree, built and queried exactly as the
reference does --

    position_tree = BallTree(positions, metric = metric)              cache.pyx:276 ("minkowski")
    indices, distances = pos_tree.query_radius(numpy.array((x,y)), r = radius,
                                               return_distance=True,
                                               sort_results=sort_results)   cache.pyx:182-185
    numpy.savez(... position_tree = pickle.dumps(tree), size = (w, h))     cache.pyx:199-210

(sklearn >= 0.2x wants the single query point as a [1, 2] array; that is the only change).
Run in the BUILD container (needs scikit-learn; nothing of the reference is imported or
copied -- sklearn is a third-party package):

    python tests/golden/make_radius_golden.py

Writes  tests/golden/radius_golden.json   query -> (indices, distances) in sklearn's order
        tests/golden/radius_metric_golden.json   the same for metric = chebyshev / manhattan / euclidean
        tests/golden/metric_cache_npz/<ripemd160(path)>.npz, ..._thumb.npz
                                          a Metric_Cache file pair in the reference's layout
                                          (pickled BallTree bytes in `position_tree`).
"""
import json
import os
import pickle
import sys

import numpy as np
from sklearn.neighbors import BallTree
import sklearn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def position_sets():
    rng = np.random.default_rng(20260001)
    sets = {}
    # SIFT-like sub-pixel positions, several keypoints per pixel position (orientations)
    p = rng.uniform(0, [800, 640], (1500, 2))
    dup = rng.choice(1500, 200, replace=False)
    p = np.concatenate([p, p[dup], p[dup[:50]]])          # coincident keypoints (2x and 3x)
    sets["subpixel_800x640"] = p
    # integer lattice: many points at exactly the radius (3-4-5, 6-8-10, 5-12-13 triangles) and
    # many equal distances
    xs, ys = np.meshgrid(np.arange(0, 60), np.arange(0, 40))
    sets["lattice_60x40"] = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.float64)
    # float32-rounded positions as cv2 KeyPoint.pt delivers them, widened to float64
    sets["float32_pts"] = rng.uniform(0, [300, 200], (700, 2)).astype(np.float32).astype(np.float64)
    # degenerate: one point, two coincident points
    sets["single"] = np.array([[10.5, 20.25]])
    sets["pair_same"] = np.array([[7.0, 7.0], [7.0, 7.0]])
    return sets


def queries(name, pos, rng):
    w, h = pos[:, 0].max() + 1, pos[:, 1].max() + 1
    out = []
    radii = [0, 1, 5, 5, 10, 13] if name == "lattice_60x40" else [0, 3, 25, 50, 100, 100]
    if name != "lattice_60x40":
        out.append((int(w / 2), int(h / 2), 1000))       # everything, sorted
    for r in radii:
        for _ in range(4):
            # Metric_Cache.get takes C ints (cache.pyx:173): integer query points; some outside the image
            x = int(rng.integers(-20, int(w) + 20))
            y = int(rng.integers(-20, int(h) + 20))
            out.append((x, y, r))
    # centred exactly on keypoints (distance 0 entries, ties at 0 for coincident keypoints)
    for i in rng.choice(len(pos), min(5, len(pos)), replace=False):
        out.append((int(pos[i, 0]), int(pos[i, 1]), 25))
    return out


def main():
    rng = np.random.default_rng(20260002)
    golden = {"sklearn_version": sklearn.__version__, "metric": "minkowski", "sets": {}}
    for name, pos in position_sets().items():
        tree = BallTree(pos, metric="minkowski")
        qs = []
        for (x, y, r) in queries(name, pos, rng):
            ind, dist = tree.query_radius(np.array((x, y)).reshape(1, -1), r=r, return_distance=True,
                                          sort_results=True)
            qs.append({"x": x, "y": y, "r": r, "indices": [int(v) for v in ind[0]],
                       "distances": [float(v) for v in dist[0]]})   # json repr round-trips float64
        golden["sets"][name] = {"positions": [[float(a), float(b)] for a, b in pos], "queries": qs}
    with open(os.path.join(HERE, "radius_golden.json"), "w") as f:
        json.dump(golden, f, separators=(",", ":"))

    # ---- the other metrics options["metric"] can name (cache.pyx:160 -> BallTree(positions, metric = metric)) ----
    rng = np.random.default_rng(20260004)
    mg = {"sklearn_version": sklearn.__version__, "sets": {}}
    sets = position_sets()
    for metric in ("chebyshev", "manhattan", "euclidean"):
        for name in ("subpixel_800x640", "lattice_60x40", "float32_pts"):
            pos = sets[name][:900]
            tree = BallTree(pos, metric=metric)
            qs = []
            for (x, y, r) in queries(name, pos, rng)[:24]:
                ind, dist = tree.query_radius(np.array((x, y)).reshape(1, -1), r=r, return_distance=True, sort_results=True)
                qs.append({"x": x, "y": y, "r": r, "indices": [int(v) for v in ind[0]], "distances": [float(v) for v in dist[0]]})
            mg["sets"]["%s/%s" % (metric, name)] = {"metric": metric, "positions": [[float(a), float(b)] for a, b in pos], "queries": qs}
    with open(os.path.join(HERE, "radius_metric_golden.json"), "w") as f:
        json.dump(mg, f, separators=(",", ":"))

    # ---- a Metric_Cache file pair written the way cache.pyx:199-210 writes it -------------
    from fastmatch_amd import synth
    from fastmatch_amd.cache import _ripemd160
    rng = np.random.default_rng(20260003)
    n, nth = 48, 20
    desc = synth.synth_sift(n, rng).astype(np.float32)               # cv2 SIFT emits float32
    pos = rng.uniform(0, [800, 640], (n, 2)).astype(np.float32).astype(np.float64)
    tdesc = synth.synth_sift(nth, rng).astype(np.float32)
    tpos = rng.uniform(0, [600, 480], (nth, 2)).astype(np.float32).astype(np.float64)

    def self_distances(d):          # r[1].distance of knnMatch(d, d, k=2): exact for integer data
        d2 = ((d[:, None, :].astype(np.float64) - d[None, :, :]) ** 2).sum(-1)
        return np.sqrt(np.sort(d2, axis=1)[:, 1].astype(np.float32)).astype(np.float64)

    path = b"images/graf/img4.ppm"
    out_dir = os.path.join(HERE, "metric_cache_npz")
    os.makedirs(out_dir, exist_ok=True)
    data_path = _ripemd160(path)
    tree = BallTree(pos, metric="minkowski")
    np.savez("%s/%s" % (out_dir, data_path),
             descriptors=desc, positions=pos, distances=self_distances(desc),
             position_tree=pickle.dumps(tree), size=(800, 640))
    np.savez("%s/%s_thumb" % (out_dir, data_path),
             positions=tpos, descriptors=tdesc, distances=self_distances(tdesc), size=(600, 480))
    print("wrote radius_golden.json and metric_cache_npz/%s{,_thumb}.npz (sklearn %s)" % (data_path, sklearn.__version__))


if __name__ == "__main__":
    main()
