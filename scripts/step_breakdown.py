import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, sharding
ctx = fm.Context(0)
Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
qb, tb = ctx.bank(Q), ctx.bank(T)
qb.set_selfdist(ctx.self_dist(qb))
out = (ctx.pinned_empty(100000, np.int32), ctx.pinned_empty(100000, np.float32), ctx.pinned_empty(100000, np.float64), ctx.pinned_empty(100000, np.uint8))
for mode in ("pinned", "pageable"):
    for _ in range(3):
        ctx.match_ratio(qb, tb, 0.7, out=out if mode == "pinned" else None)
    ctx.reset_stats()
    t_call = t_nz = t_pack = 0.0
    K = 30
    for _ in range(K):
        t0 = time.perf_counter()
        tidx, d, ratio, passed, npass = ctx.match_ratio(qb, tb, 0.7, out=out if mode == "pinned" else None)
        t1 = time.perf_counter()
        q_acc = np.nonzero(passed)[0]
        t2 = time.perf_counter()
        packed = sharding.pack_matches(q_acc, tidx[q_acc], d[q_acc])
        t3 = time.perf_counter()
        t_call += t1 - t0; t_nz += t2 - t1; t_pack += t3 - t2
    s = ctx.stats()
    print(mode, "python call %.3f ms | device call %.3f | K1 %.3f | nonzero %.3f | pack %.3f" % (
        1e3 * t_call / K, s["total_ms"] / s["calls"], s["kernel_ms"] / s["kernel_launches"], 1e3 * t_nz / K, 1e3 * t_pack / K))
