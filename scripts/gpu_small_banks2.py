"""Where the triangular self sweep starts to pay: single calls and 64-bank batched calls, full sweep ("self_tri" 0) against the
triangular sweep on every size (2), by bank size.  python scripts/gpu_small_banks2.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)
ctx.set_option("batch_group", 16)
rng = np.random.default_rng(12)
base = synth.synth_sift(40000, rng)
print("rows   | single call us: full  tri | 64-bank batch us per bank: full  tri")
for n in (300, 700, 1500, 3000, 6000, 12500, 20000, 30000, 36000):
    sizes = [int(x) for x in rng.integers(int(n * 0.9), int(n * 1.1) + 1, 64)]
    banks = [ctx.bank(base[:m]) for m in sizes]
    out = []
    for tri in (0, 2):
        ctx.set_option("self_tri", tri)
        for b in banks[:8]:
            ctx.self_dist_batch([b], want_host=False)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            for b in banks[:16]:
                ctx.self_dist_batch([b], want_host=False)
        ctx.sync()
        single = (time.perf_counter() - t0) / 48
        ctx.self_dist_batch(banks, want_host=False); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.self_dist_batch(banks, want_host=False)
        ctx.sync()
        out += [single * 1e6, (time.perf_counter() - t0) / 5 / 64 * 1e6]
    print("%6d | %7.1f %7.1f | %7.1f %7.1f" % (n, out[0], out[2], out[1], out[3]), flush=True)
