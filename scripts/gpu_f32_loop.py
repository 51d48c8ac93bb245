"""Device-resident expansion loop on float32 (RootSIFT-style, not integer valued) descriptors:
rounds/s of the float32 round (round_body_f32.h) vs the int8 round on the same geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

ctx = fm.Context(0)


def root(d):
    d = d.astype(np.float32)
    return np.sqrt(d / np.maximum(d.sum(1, keepdims=True), 1)).astype(np.float32)


for name, conv in (("int8", lambda d: d), ("float32 rootsift", root),
                   ("float32 sift+noise", lambda d: d.astype(np.float32) + np.random.default_rng(1).uniform(-.5, .5, d.shape).astype(np.float32))):
    q, t = synth.image_pair((1000, 1000), 12500, 20250100)
    mc = cache.Metric_Cache.from_arrays(conv(q["descriptors"]), q["positions"], q["size"], conv(q["thumb_descriptors"]),
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], conv(t["descriptors"]), t["thumb_positions"],
                             conv(t["thumb_descriptors"]), t["thumb_size"])
    stats = {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "return_arrays": True})
    tau = 0.7 if name == "int8" else 0.85
    get(tau)
    stats.clear()
    ctx.reset_stats()
    t0 = time.perf_counter()
    m = get(tau)
    w = time.perf_counter() - t0
    k = ctx.stats()["kernel_ms"]
    print("%-20s rounds %6d matches %6d kernel %.2f ms -> %.1f us/round (wall %.2f ms)"
          % (name, stats["rounds"], len(m[0]), k, 1e3 * k / max(stats["rounds"], 1), w * 1e3), flush=True)
    hs = {}
    t0 = time.perf_counter()
    fastmatch.match(mc, fi, {"context": ctx, "stats": hs, "device_loop": False})(tau)
    w = time.perf_counter() - t0
    print("%-20s host-driven loop: %.1f us/round" % (name, 1e6 * w / max(hs["rounds"], 1)), flush=True)
