"""Metric_Cache builds of images that all differ in size: the first self-distance call of every NEW bank size pays for the
triangular sweep's plan (host arithmetic + table upload); wall time of first calls against second calls, and against the
masked full sweep (no plan).  python scripts/gpu_tri_newsizes.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth
ctx = fm.Context(0)
rng = np.random.default_rng(3)
big = synth.synth_sift(120000, rng)
sizes = [int(x) for x in rng.integers(60000, 120000, 48)]
banks = [ctx.bank(big[:n]) for n in sizes]
ctx.self_dist(ctx.bank(big[:50000])); ctx.sync()
for tri in (1, 0):
    ctx.set_option("self_tri", tri)
    first, second = [], []
    for b in banks:
        t0 = time.perf_counter(); ctx.self_dist(b); first.append(time.perf_counter() - t0)
    for b in banks:
        t0 = time.perf_counter(); ctx.self_dist(b); second.append(time.perf_counter() - t0)
    print("self_tri %d: %d new sizes, first call %.3f ms mean (max %.3f), second call %.3f ms mean" %
          (tri, len(sizes), 1e3 * np.mean(first), 1e3 * np.max(first), 1e3 * np.mean(second)), flush=True)
