"""A/B of K1 / K2 launch knobs in ONE process on one box (run-to-run and box-to-box variance is
+-3 %): kernel time of fm_xcheck1 (K1) and fm_knn2 (K2) on the 100k x 100k pair, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

ctx = fm.Context(0)
DEFAULTS = {k: ctx.get_option(k) for k in ("nbuf", "prio", "nsplit", "nw", "nb")}
Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
qb, tb = ctx.bank(Q), ctx.bank(T)
# variants: comma separated option=value lists (fm_ctx_set_option names), e.g.  nbuf=2  nbuf=3,prio=0
variants = [dict(s.split("=") for s in v.split(",") if s) for v in (sys.argv[1:] or ["nbuf=2", "nbuf=3"])]
res = {i: {"k1": [], "k2": [], "stream": []} for i in range(len(variants))}
import time
qb.set_selfdist(ctx.self_dist(qb))
outs = [(ctx.pinned_empty(100000, np.int32), ctx.pinned_empty(100000, np.int32), ctx.pinned_empty(100000, np.float32),
         ctx.pinned_empty(100000, np.float64)) for _ in range(10)]
cnts = [ctx.pinned_empty(1, np.int64) for _ in range(10)]


def stream_ms():
    """ms per pair of a batch of 10 async calls (the bench step), best of 5 batches"""
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        for j in range(10):
            ctx.match_accepted_async(qb, tb, 0.7, outs[j], cnts[j])
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 100.0)
    return best

ref = None
for rep in range(6):
    for i, env in enumerate(variants):
        for k, v in DEFAULTS.items():
            ctx.set_option(k, v)
        for k, v in env.items():
            ctx.set_option(k, int(v))
        for name, fn in (("k1", lambda: ctx.xcheck1(qb, tb)), ("k2", lambda: ctx.knn2(qb, tb))):
            fn()
            ctx.reset_stats()
            for _ in range(10):
                out = fn()
            st = ctx.stats()
            res[i][name].append(st["kernel_ms"] / st["kernel_launches"])
        res[i]["stream"].append(stream_ms())
        if ref is None:
            ref = (ctx.xcheck1(qb, tb), ctx.knn2(qb, tb))
        else:
            a, b = ctx.xcheck1(qb, tb), ctx.knn2(qb, tb)
            assert all(np.array_equal(x, y) for x, y in zip(a, ref[0])) and all(np.array_equal(x, y) for x, y in zip(b, ref[1])), "results differ"
for i, env in enumerate(variants):
    print("%-40s K1 min %.4f med %.4f ms | K2 min %.4f med %.4f ms | async stream min %.4f med %.4f ms/pair"
          % (env, min(res[i]["k1"]), float(np.median(res[i]["k1"])), min(res[i]["k2"]), float(np.median(res[i]["k2"])),
             min(res[i]["stream"]), float(np.median(res[i]["stream"]))), flush=True)
