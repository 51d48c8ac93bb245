"""Triangular self sweep against the masked full sweep on banks beyond the bench's 100k rows: bit-equality, first-call
wall time (the plan is computed and its table uploaded once per size) and steady kernel time.  python scripts/gpu_tri_big.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth
ctx = fm.Context(0)
for n in (131072, 250001, 500000, 1000003):
    D = synth.synth_sift(n, np.random.default_rng(n))
    D[10] = D[n - 5]; D[300] = 0
    b = ctx.bank(D)
    out = {}
    for name, tri in (("full", 0), ("tri", 1)):
        ctx.set_option("self_tri", tri)
        t0 = time.time(); r = ctx.self_dist(b); first = time.time() - t0
        ctx.reset_stats()
        t0 = time.time()
        for _ in range(3):
            r = ctx.self_dist(b)
        wall = (time.time() - t0) / 3
        s = ctx.stats()
        out[name] = (r, first * 1e3, wall * 1e3, s["kernel_ms"] / 3)
    eq = np.array_equal(out["full"][0].view(np.uint64), out["tri"][0].view(np.uint64))
    print(n, "equal" if eq else "MISMATCH", " | ".join("%s first %.1f wall %.1f kernel %.2f ms" % ((k,) + out[k][1:]) for k in out), flush=True)
    del b
