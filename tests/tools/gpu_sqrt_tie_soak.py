"""Soak of the float32-root tie repair at sizes with several splits and workgroups: random tie-range banks
(tests/kat.far_banks) with random shapes, option settings and planted {n, n + 1} pairs; 2-NN, cross-check both
ways and the accepted-match path against the oracle.  python tests/tools/gpu_sqrt_tie_soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fastmatch_amd as fm
import oracle
from kat import far_banks, row_with_sumsq, SQRT_TIE_MIN

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = fm.Context(0)
rng = np.random.default_rng(int(time.time()))
t0, it, rows_fixed = time.time(), 0, 0
eq = lambda a, b: a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
while time.time() - t0 < budget:
    nq, nt = int(rng.integers(1, 9000)), int(rng.integers(2, 40000))
    Q, T = far_banks(nq, nt, rng, int(rng.integers(1, 14)), int(rng.integers(1, 4)))
    for _ in range(int(rng.integers(0, 6))):                       # planted pairs from a zero query row
        n = int(rng.integers(SQRT_TIE_MIN, 6400000))
        q = int(rng.integers(0, nq)); Q[q] = 0
        i, j = rng.choice(nt, 2, replace=False)
        T[i], T[j] = row_with_sumsq(n + 1), row_with_sumsq(n)
    for k, v in (("nsplit", int(rng.choice([0, 1, 3, 8]))), ("nbuf", int(rng.choice([0, 2, 3]))), ("coop", int(rng.integers(0, 2)))):
        ctx.set_option(k, v)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    idx, d = ctx.knn2(qb, tb)
    oi, od = oracle.bf_knn(Q, T, 2)
    assert eq(idx, oi) and eq(d, od), ("knn2", nq, nt)
    for a, b, A, B in ((qb, tb, Q, T), (tb, qb, T, Q)):
        t, x = ctx.xcheck1(a, b)
        ot, ox = oracle.bf_xcheck1(A, B)
        assert eq(t, ot) and eq(x, ox), ("xcheck", nq, nt)
    it += 1
print("sqrt-tie soak ok: %d random problems in %.0f s" % (it, time.time() - t0))
