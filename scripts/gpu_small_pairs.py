"""X1 + R1 over a dataset of SMALL image pairs (BASELINE configs[3] sizes: ~12.5k keypoints a side, every pair another
size): fm_match_accepted_batch per pair under the planner's default shapes (4-wave workgroups below 32768 train rows:
one launch per pair) and with the 8-wave shape forced (options nb = 4, nw = 8: batched launches).
python scripts/gpu_small_pairs.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)
ctx.set_option("batch_group", 16)
rng = np.random.default_rng(21)
base, base2 = synth.synth_sift(40000, rng), synth.synth_sift(40000, rng)
for lo, hi in ((2500, 3500), (11000, 14000), (28000, 34000)):
    sizes = [(int(a), int(b)) for a, b in zip(rng.integers(lo, hi, 64), rng.integers(lo, hi, 64))]
    pairs = [(ctx.bank(base[:a]), ctx.bank(base2[:b])) for a, b in sizes]
    ctx.self_dist_batch([q for q, _ in pairs], want_host=False); ctx.sync()
    cap = hi
    outs = [(ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.float32),
             ctx.pinned_empty(cap, np.float64)) for _ in pairs]
    cnts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    blk = ctx.prepare_batch(pairs, outs, cnts)
    ref = None
    for name, nb, nw in (("default", 0, 0), ("nb4 nw8", 4, 8)):
        ctx.set_option("nb", nb); ctx.set_option("nw", nw)
        ctx.match_accepted_batch(blk, 0.9); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.match_accepted_batch(blk, 0.9)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 5 / len(pairs)
        got = [(int(c[0]), o[0][:int(c[0])].copy(), o[1][:int(c[0])].copy()) for c, o in zip(cnts, outs)]
        if ref is None:
            ref = got
        same = all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(ref, got))
        work = np.mean([a * b for a, b in sizes])
        print("pairs of %5d .. %5d rows, %-8s: %7.1f us per pair (%.2e pairs/s) %s" % (lo, hi, name, dt * 1e6, work / dt, "same" if same else "DIFFERENT"), flush=True)
    ctx.set_option("nb", 0); ctx.set_option("nw", 0)
