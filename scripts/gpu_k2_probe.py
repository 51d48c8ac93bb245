import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth
N = 100000
Q, T, _ = synth.planted_pair(N, N, 20250002)
R = synth.synth_sift(N, np.random.default_rng(5))
for coop in ("1", "0"):
    os.environ["FM_COOP"] = coop
    ctx = fm.Context(0)
    qb, tb, rb = ctx.bank(Q), ctx.bank(T), ctx.bank(R)
    for name, a, b in (("knn2(Q,T)", qb, tb), ("knn2(T,Q)", tb, qb), ("knn2(Q,Q)", qb, qb), ("knn2(T,T)", tb, tb), ("knn2(R,T)", rb, tb)):
        ts = []
        for _ in range(4):
            ctx.reset_stats(); ctx.knn2(a, b); ts.append(ctx.stats()["kernel_ms"])
        print("coop=%s %-10s min %.3f ms" % (coop, name, min(ts)), flush=True)
    ctx.close()
