import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch
ctx = fm.Context(0)
q, t = synth.image_pair((1000, 1000), 12500, 20250100)
mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"], q["thumb_positions"], q["thumb_size"], options={"context": ctx})
fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
stats = {}
get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "return_arrays": True})
for _ in range(3):
    stats.clear(); t0 = time.perf_counter(); m = get(0.7); dt = time.perf_counter() - t0
    print("%d matches %d rounds %.4f s -> %.1f us/round" % (len(m[0]), stats["rounds"], dt, 1e6 * dt / stats["rounds"]), flush=True)
