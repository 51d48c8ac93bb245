import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch
ctx = fm.Context(0)
q, t = synth.image_pair((1000, 1000), 12500, 20250100)
mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"], q["thumb_positions"], q["thumb_size"], options={"context": ctx})
fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
get = fastmatch.match(mc, fi, {"context": ctx})
get(0.7)
pr = cProfile.Profile(); pr.enable(); m = get(0.7); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
