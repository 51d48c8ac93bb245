"""Kernel-time sweep of the dense path over tuning knobs (env: FM_NB, FM_NSPLIT, FM_COOP,
FM_GLDS); prints HIP-event kernel time for X1 and K2 at N x N.  Each variant also checks
results (X1 against the first variant, K2 rows against the oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
import fastmatch_amd as fm
from fastmatch_amd import synth

N = int(os.environ.get("SWEEP_N", "100000"))
Q, T, _ = synth.planted_pair(N, N, 20250002)
rows = np.random.default_rng(1).choice(N, 48, replace=False)
oi, od = oracle.bf_knn(Q[rows], T, 2)
ref = None


def run(env):
    global ref
    for k, v in env.items():
        os.environ[k] = str(v)
    ctx = fm.Context(0)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    res = {}
    for name, fn in (("x1", lambda: ctx.xcheck1(qb, tb)), ("k2", lambda: ctx.knn2(qb, tb))):
        out = fn()
        ts = []
        for _ in range(5):
            ctx.reset_stats()
            out = fn()
            ts.append(ctx.stats()["kernel_ms"])
        res[name] = (min(ts), float(np.median(ts)))
        if name == "k2":
            ok = np.array_equal(out[0][rows], oi) and np.array_equal(out[1][rows], od)
        else:
            if ref is None:
                ref = out
            ok = np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
        res[name + "_ok"] = ok
    ctx.close()
    for k in env:
        os.environ.pop(k)
    print(" ".join("%s=%s" % kv for kv in sorted(env.items())),
          "| x1 min %.3f med %.3f ms (%.2e pairs/s) ok=%s | k2 min %.3f med %.3f ms ok=%s" % (
              res["x1"][0], res["x1"][1], N * N / res["x1"][0] * 1e3, res["x1_ok"],
              res["k2"][0], res["k2"][1], res["k2_ok"]), flush=True)


variants = [dict(FM_COOP=c, FM_NSPLIT=s, FM_NB=nb, FM_NW=nw) for c, s, nb, nw in
            [(1, 0, 0, 0), (0, 0, 0, 0), (1, 8, 4, 8), (1, 16, 4, 8), (1, 32, 4, 8), (1, 16, 4, 4),
             (1, 16, 4, 16), (1, 16, 8, 8), (0, 1, 4, 8)]]
if len(sys.argv) > 1:
    variants = [eval("dict(%s)" % a) for a in sys.argv[1:]]
for v in variants:
    run(v)
