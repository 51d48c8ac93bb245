"""Triangular self sweep (rowreduce.hip, TRI) against the masked full sweep: results bit for bit on a range of sizes
(the full sweep is the suite's oracle-checked form), then kernel time A/B in one process, interleaved, for several
piece lengths.  Usage: python scripts/gpu_tri_check.py [rows for the timing (100000)]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)


def self_dist(D, tri, stages=None):
    ctx.set_option("self_tri", tri)
    if stages:
        ctx.set_option("tri_stages", stages)
    return ctx.self_dist(ctx.bank(D))


bad = 0
for n in (1, 2, 3, 127, 128, 129, 511, 512, 513, 640, 1000, 1025, 4097, 9000, 33000, 70001):
    D = synth.synth_sift(n, np.random.default_rng(n))
    if n >= 600:
        D[10] = D[500]
        D[300] = 0
    a = self_dist(D, 0)
    for st in (4, 7, 32):
        b = self_dist(D, 2, st)
        ok = np.array_equal(a.view(np.uint64), b.view(np.uint64))
        if not ok:
            bad += 1
            w = np.flatnonzero(a.view(np.uint64) != b.view(np.uint64))
            print("MISMATCH n", n, "stages", st, "rows", w[:8], a[w[:4]], b[w[:4]], len(w))
print("parity:", "OK" if bad == 0 else "%d FAILED" % bad)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = synth.synth_sift(N, np.random.default_rng(7))
bank = ctx.bank(D)
ref = None
res = {}
configs = [("full", 0, 32)] + [("tri%d" % s, 1, s) for s in (0, 16, 24, 32, 48, 61)]
for rep in range(4):
    for name, tri, st in configs:
        ctx.set_option("self_tri", tri)
        ctx.set_option("tri_stages", st)
        ctx.self_dist(bank)            # warm (table upload, clocks)
        ctx.reset_stats()
        for _ in range(6):
            sd = ctx.self_dist(bank)
        s = ctx.stats()
        res.setdefault(name, []).append(s["kernel_ms"] / max(1, s["kernel_launches"]))
        if ref is None:
            ref = sd
        elif not np.array_equal(ref.view(np.uint64), sd.view(np.uint64)):
            print("MISMATCH at", N, name)
print(json.dumps({k: [round(x, 4) for x in v] for k, v in res.items()}))
# batched (12 banks in one launch, the bench's self_2nn / fresh_pair shape)
banks = [ctx.bank(synth.synth_sift(N, np.random.default_rng(100 + i))) for i in range(12)]
ctx.set_option("batch_group", 16)
for name, tri, st in [("full", 0, 32), ("tri0", 1, 0), ("tri24", 1, 24), ("tri48", 1, 48)]:
    ctx.set_option("self_tri", tri)
    ctx.set_option("tri_stages", st)
    ctx.self_dist_batch(banks, want_host=False); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.self_dist_batch(banks, want_host=False)
    ctx.sync()
    print("batch12", name, "ms per bank %.4f" % ((time.perf_counter() - t0) * 1e3 / 60))
