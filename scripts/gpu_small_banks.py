"""Metric_Cache builds of a dataset of SMALL images (BASELINE configs[3]: 12.5k keypoints each; and 3k / 30k): fm_self_dist_batch
over 64 banks, per bank, under the default rule (triangular sweep from 32768 padded rows on) and with the triangular sweep on
every size ("self_tri" 2: batched launches of up to 16 banks of any sizes).  python scripts/gpu_small_banks.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)
ctx.set_option("batch_group", 16)
rng = np.random.default_rng(11)
base = synth.synth_sift(40000, rng)
for lo, hi in ((2500, 3500), (11000, 14000), (28000, 34000)):
    sizes = [int(x) for x in rng.integers(lo, hi, 64)]
    banks = [ctx.bank(base[:n]) for n in sizes]
    ref = None
    for tri in (1, 2, 0):
        ctx.set_option("self_tri", tri)
        ctx.self_dist_batch(banks, want_host=False); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.self_dist_batch(banks, want_host=False)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 5 / len(banks)
        got = ctx.self_dist_batch(banks)
        if ref is None:
            ref = got
        ok = all(np.array_equal(a.view(np.uint64), b.view(np.uint64)) for a, b in zip(ref, got))
        work = np.mean([n * n for n in sizes])
        print("banks of %5d .. %5d rows, self_tri %d: %7.1f us per bank (%.2e pairs/s) %s" % (lo, hi, tri, dt * 1e6, work / dt, "same" if ok else "DIFFERENT"), flush=True)
