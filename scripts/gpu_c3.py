"""BASELINE.json config 3 (single 24-MP pair, ~300k descriptors/side, Grid_Cache expansion
schedule) and config 4 (batch of 1-MP pairs, 12.5k/side) through fastmatch.match() /
match_many(): device-resident loop vs host-driven loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

ctx = fm.Context(0)

def build(size, n, seed, n_thumb):
    q, t = synth.image_pair(size, n, seed, n_thumb=n_thumb)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    return mc, fi

for name, size, n, seed, n_thumb in (("C4 pair", (1000, 1000), 12500, 20250100, 600), ("C3", (6000, 4000), 300000, 20250003, 2000)):
    t0 = time.perf_counter()
    mc, fi = build(size, n, seed, n_thumb)
    print("%s: synth + Metric_Cache build %.2fs" % (name, time.perf_counter() - t0), flush=True)
    for mode in ("device", "host"):
        if mode == "host" and name == "C3" and os.environ.get("SKIP_HOST_C3"):
            continue
        stats = {}
        t2 = time.perf_counter()
        get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "device_loop": mode == "device", "return_arrays": True})
        t3 = time.perf_counter()
        for rep in range(3 if mode == "device" else 1):
            stats.clear(); ctx.reset_stats()
            t4 = time.perf_counter()
            m = get(0.7)
            t5 = time.perf_counter()
            s = ctx.stats()
            nm = len(m[0]) if isinstance(m, tuple) else len(m)
            print("  %-6s loop (seeding %.3fs): %d matches, %d rounds, %.3e pairs in %.4fs -> %.0f rounds/s, %.3e pairs/s, %.0f matches/s | kernels %.4fs" % (
                mode, t3 - t2, nm, stats["rounds"], stats["pairs"], t5 - t4, stats["rounds"] / (t5 - t4), stats["pairs"] / (t5 - t4),
                nm / (t5 - t4), s["kernel_ms"] / 1e3), flush=True)

# config 4 batch: 64 pairs on one GPU (8 per GPU on an 8-GPU node)
for npairs in (8, 64):
    pairs = [build((1000, 1000), 12500, 20250100 + i, 600) for i in range(npairs)]
    prepared, stats = [], {}
    t0 = time.perf_counter()
    fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared_out": prepared, "return_arrays": True})
    t1 = time.perf_counter()
    for rep in range(2):
        stats.clear(); ctx.reset_stats()
        t2 = time.perf_counter()
        res = fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared": prepared, "stats": stats, "return_arrays": True})
        t3 = time.perf_counter()
        s = ctx.stats()
        nm = sum(len(r[0]) for r in res)
        print("C4 batch of %d pairs: first call (pack cells + seeding + run) %.2fs; run %.4fs: %d matches, %d rounds, %.3e pairs -> %.0f rounds/s, %.3e pairs/s, %.0f matches/s | kernel %.4fs" % (
            npairs, t1 - t0, t3 - t2, nm, stats["rounds"], stats["pairs"], stats["rounds"] / (t3 - t2), stats["pairs"] / (t3 - t2), nm / (t3 - t2), s["kernel_ms"] / 1e3), flush=True)
