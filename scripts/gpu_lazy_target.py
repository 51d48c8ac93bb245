"""Pixel target (cells computed on demand, the reference's own mode) through the device loop (r04: park / resume) and through
the host-driven loop, with the cells' features memoised so that what is timed is the LOOP, not the stand-in extractor:
wall per threshold, rounds, cells computed, launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import cache, fastmatch, imaging, standin
from imagegen import texture, warp

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 640)
ctx = fm.Context(0)
img1 = texture(W, H, seed=1)
img4 = warp(img1, np.array([[1.0, 0.01, 18.0], [-0.008, 1.0, -11.0], [1e-5, -5e-6, 1.0]]))
feat = standin.standin_features
kq, dq = feat(img4)
tq = imaging.get_thumbnail(img4, (600, 600))
ktq, dtq = feat(tq)
pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
mc = cache.Metric_Cache.from_arrays(dq, pos(kq), (W, H), dtq, pos(ktq), (tq.shape[1], tq.shape[0]), options={"context": ctx})
memo = {}


def cached(data):
    k = (data.shape, data[::7, ::7].tobytes())
    if k not in memo:
        memo[k] = feat(data)
    return memo[k]


for name, opts in (("device loop, cells on demand", {}), ("host-driven loop", {"device_loop": False})):
    for rep in range(2):                   # second pass: every crop's features come from the memo
        st = {}
        get = fastmatch.match(mc, img1, dict(opts, context=ctx, feature_function=cached, stats=st))
        t0 = time.perf_counter()
        m = get(0.9)
        dt = time.perf_counter() - t0
    print("%-30s %d keypoints, %d rounds, %d matches, %d cells computed: %.1f ms (%.1f us per round)"
          % (name, len(dq), st["rounds"], len(m), st.get("lazy_cells", -1), 1e3 * dt, 1e6 * dt / max(st["rounds"], 1)), flush=True)
