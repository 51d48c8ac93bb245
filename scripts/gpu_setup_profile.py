"""GPU: where the per-pair SET-UP of the device-resident loop goes (not the loop itself).

config 4 shape (1 MP pairs, 12.5k keypoints per side): the loop takes ~0.4 ms per pair inside a 64-pair launch, the
preparation (Metric_Cache, Grid_Cache, thumbnail seeding, expander, first run state) several ms -- this prints the
preparation's phases per pair and a cProfile of fastmatch.match_many's preparing call.

  python scripts/gpu_setup_profile.py [n_pairs=16] [config3=0|1]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
c3 = len(sys.argv) > 2 and sys.argv[2] == "1"
ctx = fm.Context(0)
size, n, nth = ((6000, 4000), 300000, 2000) if c3 else ((1000, 1000), 12500, 600)
if c3:
    n_pairs = 1

raw = [synth.image_pair(size, n, 20250100 + i, n_thumb=nth) for i in range(n_pairs + 1)]


def caches(q, t):
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    return mc, fi


# warm everything once (module load, first allocations)
w = caches(*raw[-1])
fastmatch.match_many([w], 0.7, {"context": ctx})
ctx.sync()

t0 = time.perf_counter()
pairs = [caches(q, t) for q, t in raw[:n_pairs]]
ctx.sync()
t_cache = time.perf_counter() - t0

prepared = []
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared_out": prepared, "return_arrays": True})
pr.disable()
t_first = time.perf_counter() - t0

t0 = time.perf_counter()
res2 = fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared": prepared, "return_arrays": True})
t_again = time.perf_counter() - t0
assert all(np.array_equal(a[0], b[0]) for a, b in zip(res, res2))

print("pairs %d (%s): caches %.2f ms per pair | first match_many (grids + seeding + expanders + run states + loop) %.2f ms per pair "
      "| prepared match_many (the loop + fetch) %.3f ms per pair" % (n_pairs, "config 3" if c3 else "config 4 shape",
                                                                     1e3 * t_cache / n_pairs, 1e3 * t_first / n_pairs, 1e3 * t_again / n_pairs))
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
