"""Repeatability of the device-resident expansion loop: the same batch of image pairs (int8 and float32
descriptors mixed) run N times in one launch each; every run must return the identical match lists."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ctx = fm.Context(0)
rng = np.random.default_rng(5)
pairs = []
for k in range(npairs):
    q, t = synth.image_pair((1000, 1000), 12500, 20250100 + k)
    conv = (lambda d: d) if k % 4 else (lambda d: d.astype(np.float32) + rng.uniform(-0.4, 0.4, d.shape).astype(np.float32))
    mc = cache.Metric_Cache.from_arrays(conv(q["descriptors"]), q["positions"], q["size"], conv(q["thumb_descriptors"]),
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], conv(t["descriptors"]), t["thumb_positions"],
                             conv(t["thumb_descriptors"]), t["thumb_size"])
    pairs.append((mc, fi))
prepared = []
ref = fastmatch.match_many(pairs, 0.75, {"context": ctx, "prepared_out": prepared, "return_arrays": True})
assert all(p["expander"] not in (None, False) for p in prepared)
t0 = time.perf_counter()
for rep in range(reps):
    got = fastmatch.match_many(pairs, 0.75, {"context": ctx, "prepared": prepared, "return_arrays": True})
    for k, (a, b) in enumerate(zip(got, ref)):
        if not all(np.array_equal(x, y) for x, y in zip(a, b)):
            raise SystemExit("MISMATCH rep %d pair %d" % (rep, k))
print("expand repeatability ok: %d runs x %d pairs (%d matches), %.2f s" % (reps, npairs, sum(len(r[0]) for r in ref), time.perf_counter() - t0))
