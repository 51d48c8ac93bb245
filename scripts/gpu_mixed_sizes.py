"""X1 + R1 over a batch of image pairs whose banks all differ in size (a dataset's images do) against a batch of equal
pairs with the same total work: ms per pair and per 1e10 descriptor pairs.  python scripts/gpu_mixed_sizes.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)
rng = np.random.default_rng(5)
NP = 12
base = synth.synth_sift(115000, rng)
base2 = synth.synth_sift(115000, rng)


def make(sizes):
    pairs = []
    for nq, nt in sizes:
        q, t = ctx.bank(base[:nq]), ctx.bank(base2[:nt])
        pairs.append((q, t))
    ctx.self_dist_batch([q for q, _ in pairs], want_host=False)
    ctx.sync()
    return pairs


def run(pairs, reps=6):
    cap = max(q.n for q, _ in pairs)
    outs = [(ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.float32),
             ctx.pinned_empty(cap, np.float64)) for _ in pairs]
    cnts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    blk = ctx.prepare_batch(pairs, outs, cnts)
    ctx.match_accepted_batch(blk, 0.7); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.match_accepted_batch(blk, 0.7)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    work = sum(q.n * t.n for q, t in pairs)
    return dt * 1e3 / len(pairs), dt * 1e3 / (work / 1e10), [int(c[0]) for c in cnts][:3]


equal = make([(100000, 100000)] * NP)
print("equal 100k x 100k          : %.3f ms per pair, %.3f ms per 1e10 pairs %s" % run(equal), flush=True)
sizes = [(int(a), int(b)) for a, b in zip(rng.integers(85000, 115000, NP), rng.integers(85000, 115000, NP))]
mixed = make(sizes)
print("mixed 85k..115k (all differ): %.3f ms per pair, %.3f ms per 1e10 pairs %s" % run(mixed), flush=True)
pad = make([(((a + 4095) // 4096) * 4096, ((b + 4095) // 4096) * 4096) for a, b in sizes])
print("sizes rounded up to 4096    : %.3f ms per pair, %.3f ms per 1e10 pairs %s" % run(pad), flush=True)
