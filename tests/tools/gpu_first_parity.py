"""First-light GPU script: HIP path vs oracle on a ladder of sizes, both staging modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
import fastmatch_amd as fm
from fastmatch_amd import synth

def run(nq, nt, seed, ctx):
    Q, T, _ = synth.planted_pair(nq, nt, seed)
    t0 = time.time()
    qb, tb = ctx.bank(Q), ctx.bank(T)
    tidx, dist = ctx.xcheck1(qb, tb)
    t1 = time.time()
    ot, od = oracle.bf_xcheck1(Q, T)
    ok1 = np.array_equal(tidx, ot) and np.array_equal(dist, od)
    idx, d2 = ctx.knn2(qb, tb)
    oi, od2 = oracle.bf_knn(Q, T, 2)
    ok2 = np.array_equal(idx, oi) and np.array_equal(d2, od2)
    sd = ctx.self_dist(qb)
    osd = oracle.self_dist(Q)
    ok3 = np.array_equal(sd, osd)
    print("nq=%d nt=%d xcheck=%s knn2=%s self=%s matched=%d gpu_s=%.3f" % (nq, nt, ok1, ok2, ok3, (tidx >= 0).sum(), t1 - t0), flush=True)
    if not ok1:
        bad = np.nonzero((tidx != ot) | (dist != od))[0]
        print("  xcheck mismatches:", len(bad), bad[:5], tidx[bad[:5]], ot[bad[:5]], dist[bad[:5]], od[bad[:5]])
    if not ok2:
        bad = np.nonzero((idx != oi).any(1) | (d2 != od2).any(1))[0]
        print("  knn2 mismatches:", len(bad), bad[:5], idx[bad[:5]], oi[bad[:5]], d2[bad[:5]], od2[bad[:5]])
    return ok1 and ok2 and ok3

if __name__ == "__main__":
    allok = True
    for glds in ("1", "0"):
        os.environ["FM_GLDS"] = glds
        ctx = fm.Context(0)
        print("device:", ctx.device_name(), "glds=", glds, flush=True)
        for (nq, nt) in [(1, 1), (3, 2), (33, 17), (128, 128), (300, 200), (393, 125), (1000, 1500), (4096, 5000)]:
            allok &= run(nq, nt, 1234 + nq, ctx)
        for nb in ("1", "2", "4"):
            os.environ["FM_NB"] = nb
            c2 = fm.Context(0)
            allok &= run(2000, 3000, 77, c2)
            c2.close()
        os.environ.pop("FM_NB")
        ctx.close()
    # a timing at scale
    ctx = fm.Context(0)
    Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    for it in range(3):
        ctx.reset_stats()
        t0 = time.time(); tidx, dist = ctx.xcheck1(qb, tb); t1 = time.time()
        s = ctx.stats()
        print("100k x 100k xcheck: wall %.4f s kernel %.3f ms total %.3f ms -> %.3e pairs/s (kernel)" % (t1 - t0, s["kernel_ms"], s["total_ms"], 1e10 / (s["kernel_ms"] * 1e-3)), flush=True)
    for it in range(2):
        ctx.reset_stats()
        idx, d2 = ctx.knn2(qb, tb)
        s = ctx.stats()
        print("100k x 100k knn2: kernel %.3f ms total %.3f ms -> %.3e pairs/s" % (s["kernel_ms"], s["total_ms"], 1e10 / (s["kernel_ms"] * 1e-3)), flush=True)
    # spot-check rows of the big problem against the oracle
    rows = np.random.default_rng(1).choice(100000, 64, replace=False)
    oi, od = oracle.bf_knn(Q[rows], T, 2)
    print("big knn2 spot-check:", np.array_equal(idx[rows], oi) and np.array_equal(d2[rows], od))
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
