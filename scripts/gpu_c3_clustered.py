"""GPU: the clustered configs[2] pair of bench.py's `expand_c3.clustered` leg on its own (300k keypoints per side, half of
them in 12 Gaussian blobs): device loop with and without delegation of the big rounds' cross-checks; FM_EXPAND_DEBUG=1
prints where the host's time in the park / resume cycle goes.

  python scripts/gpu_c3_clustered.py [reps]          (FM_ROOTSIFT=1: float32 descriptors)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = fm.Context(0)
q, t = synth.image_pair((6000, 4000), 300000, 20250004, n_thumb=2000, p=0.15, clusters=12, cluster_sigma=60.0, cluster_frac=0.5)
if os.environ.get("FM_ROOTSIFT"):              # RootSIFT-style float32 descriptors: the float32 route (K8 / the float32 round)
    for side in (q, t):
        for k in ("descriptors", "thumb_descriptors"):
            d = side[k].astype(np.float32)
            side[k] = np.sqrt(d / np.maximum(d.sum(axis=1, keepdims=True), 1.0)).astype(np.float32)
mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"], q["thumb_positions"],
                                    q["thumb_size"], options={"context": ctx})
fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
stats = {}
get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "return_arrays": True})
get(0.7)
for d in (ctx.get_option("expand_delegate"), 0):
    ctx.set_option("expand_delegate", d)
    for _ in range(reps if d else 1):
        stats.clear()
        t0 = time.perf_counter()
        index, pos, ratio = get(0.7)
        wall = time.perf_counter() - t0
        print("expand_delegate %8d: %d rounds, %d matches, %.4f s, fallbacks %d" % (d, stats.get("rounds", 0), len(index), wall,
                                                                                  stats.get("device_fallbacks", 0)), flush=True)
