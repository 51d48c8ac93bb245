import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch
ctx = fm.Context(0)
q, t = synth.image_pair((1000, 1000), 12500, 20250100)
mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"], q["thumb_positions"], q["thumb_size"], options={"context": ctx})
fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
st = {}
get = fastmatch.match(mc, fi, {"context": ctx, "stats": st})
get(0.7)
for rep in range(3):                       # un-instrumented first: the timers themselves cost ~0.2 us per round
    st.clear(); ctx.reset_stats()
    get(0.7)
    print("no timers: rounds", st["rounds"], "us/round %.2f" % (ctx.stats()["kernel_ms"] * 1e3 / st["rounds"]))
ctx.set_option("expand_prof", 1)
st.clear(); ctx.reset_stats()
t0 = time.perf_counter(); m = get(0.7); w = time.perf_counter() - t0
print("rounds", st["rounds"], "wall ms %.2f" % (w * 1e3), "us/round %.2f" % (ctx.stats()["kernel_ms"] * 1e3 / st["rounds"]))
