"""Where the HOST time of a dataset of small image pairs goes (BASELINE configs[3]: 1000 x 1000 pairs, 12.5k keypoints a side):
Metric_Cache builds, then the first fastmatch.match_many (grids, seeding, expanders) under cProfile -- per pair, next to the
26 ms the device loop takes for all 64.  python scripts/gpu_pair_setup_prof.py [pairs (32)]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

ctx = fm.Context(0)
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 32
raw = [synth.image_pair((1000, 1000), 12500, 20250100 + i, n_thumb=600) for i in range(NP)]
# warm every code path once
q, t = raw[0]
mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"], q["thumb_positions"], q["thumb_size"], options={"context": ctx})
fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
fastmatch.match_many([(mc, fi)], 0.7, {"context": ctx, "return_arrays": True})


def top(pr, n=18):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_stats(n)
    return "\n".join(l[:150] for l in s.getvalue().splitlines()[4:])


pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
pairs = []
for q, t in raw:
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"], q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
    pairs.append((mc, fi))
pr.disable()
print("caches: %.3f ms per pair" % ((time.perf_counter() - t0) * 1e3 / NP))
print(top(pr))
pr = cProfile.Profile()
prepared = []
t0 = time.perf_counter()
pr.enable()
fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared_out": prepared, "return_arrays": True})
pr.disable()
print("first match_many (grids, seeding, expanders, launch): %.3f ms per pair" % ((time.perf_counter() - t0) * 1e3 / NP))
print(top(pr, 26))
t0 = time.perf_counter()
fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared": prepared, "return_arrays": True})
print("second match_many (the device loop + fetch): %.3f ms per pair" % ((time.perf_counter() - t0) * 1e3 / NP))
