"""Float32 route: bf16x3 filter (K8) vs all-pairs kernel (K5) vs oracle -- parity and timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fastmatch_amd
from fastmatch_amd import _ffi, synth
import oracle


def banks(nq, nt, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "sift":                      # SIFT-like bytes, RootSIFT-style normalisation -> non-integer floats
        Q = synth.synth_sift(nq, rng).astype(np.float32)
        T = synth.synth_sift(nt, rng).astype(np.float32)
        m = min(nq, nt) // 2
        T[:m] = Q[:m] + rng.normal(0, 6, size=(m, 128)).astype(np.float32)
        Q = np.sqrt(Q / np.maximum(Q.sum(1, keepdims=True), 1)).astype(np.float32)
        T = np.sqrt(np.abs(T) / np.maximum(np.abs(T).sum(1, keepdims=True), 1)).astype(np.float32)
    elif kind == "gauss":
        Q = rng.normal(0, 1, size=(nq, 128)).astype(np.float32)
        T = rng.normal(0, 1, size=(nt, 128)).astype(np.float32)
    else:                                   # near-duplicates: many rows within a hair of each other
        base = rng.normal(0, 1, size=(8, 128)).astype(np.float32)
        Q = (base[rng.integers(0, 8, nq)] + rng.normal(0, 1e-4, size=(nq, 128))).astype(np.float32)
        T = (base[rng.integers(0, 8, nt)] + rng.normal(0, 1e-4, size=(nt, 128))).astype(np.float32)
    return Q, T


def run(mode, Q, T, reps=3):
    os.environ["FM_F32_FILTER"] = str(mode)
    ctx = _ffi.Context(0)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    out = {}
    for name, fn in (("knn2", lambda: ctx.knn2(qb, tb)), ("xcheck1", lambda: ctx.xcheck1(qb, tb))):
        fn()
        ctx.reset_stats()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        dt = (time.perf_counter() - t0) / reps
        st = ctx.stats()
        out[name] = (r, dt, st["kernel_ms"] / max(st["kernel_launches"], 1))
    out["filter"] = ctx.f32_filter_stats()
    ctx.close()
    return out


if __name__ == "__main__":
    cases = [(300, 500, "gauss"), (3000, 5000, "sift"), (2000, 3000, "dup"), (10000, 20000, "gauss"), (20000, 100000, "sift")]
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        cases = [(10000, 1000000, "sift"), (100000, 100000, "sift")]
    for nq, nt, kind in cases:
        Q, T = banks(nq, nt, 7, kind)
        a = run(2, Q, T)
        b = run(0, Q, T)
        line = "%-6s %7d x %7d  " % (kind, nq, nt)
        for name in ("knn2", "xcheck1"):
            ra, rb = a[name][0], b[name][0]
            same = all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
                       for x, y in zip(ra, rb))
            line += "%s: same=%s K8 %.3f ms K5 %.3f ms (%.2e pairs/s)  " % (name, same, a[name][2], b[name][2], nq * nt / (a[name][2] * 1e-3))
        line += "filter(launches, fallbacks)=%s" % (a["filter"],)
        if nq * nt <= 2e8:
            oi, od = oracle.bf_knn(Q, T, 2, order=1)
            ok = np.array_equal(oi, a["knn2"][0][0]) and np.array_equal(od.view(np.uint32), a["knn2"][0][1].view(np.uint32))
            ti, td = oracle.bf_xcheck1(Q, T, order=1)
            ok2 = np.array_equal(ti, a["xcheck1"][0][0]) and np.array_equal(td.view(np.uint32), a["xcheck1"][0][1].view(np.uint32))
            line += " oracle: knn2=%s xcheck=%s" % (ok, ok2)
        print(line, flush=True)
