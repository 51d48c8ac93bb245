"""BASELINE.json config 3 (single 24-MP pair, ~300k descriptors/side, Grid_Cache expansion
schedule) and one config-4 pair (1 MP, 12.5k/side) through fastmatch.match()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

ctx = fm.Context(0)
for name, size, n, seed, n_thumb in (("C4 pair", (1000, 1000), 12500, 20250100, 600), ("C3", (6000, 4000), 300000, 20250003, 2000)):
    t0 = time.perf_counter()
    q, t = synth.image_pair(size, n, seed, n_thumb=n_thumb)
    t1 = time.perf_counter()
    ctx.reset_stats()
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    t2 = time.perf_counter()
    s = ctx.stats()
    print("%s: synth %.2fs, Metric_Cache build %.3fs (self-2NN kernel %.2f ms)" % (name, t1 - t0, t2 - t1, s["kernel_ms"]), flush=True)
    stats = {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats})
    t3 = time.perf_counter()
    print("  match() setup + seeding %.3fs" % (t3 - t2), flush=True)
    for tau in (0.7, 0.7):
        stats.clear(); ctx.reset_stats()
        t4 = time.perf_counter()
        m = get(tau)
        t5 = time.perf_counter()
        s = ctx.stats()
        print("  tau %.1f: %d matches, %d rounds, %.3e pairs in %.3fs -> %.0f rounds/s, %.3e pairs/s, %.0f matches/s | device calls %.3fs kernels %.3fs" % (
            tau, len(m), stats["rounds"], stats["pairs"], t5 - t4, stats["rounds"] / (t5 - t4), stats["pairs"] / (t5 - t4),
            len(m) / (t5 - t4), s["total_ms"] / 1e3, s["kernel_ms"] / 1e3), flush=True)
