"""Reproduces the BASELINE.md section-4 table on one MI355X: every BASELINE.json config that
runs without SIFT (2, 3, 4, 5), one JSON line each.  Not the driver's bench (that is
bench.py = config 2); run with  /usr/local/graft/bin/gpurun -- python scripts/run_configs.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch

ctx = fm.Context(0)
only = set(sys.argv[1:])


def emit(**kw):
    print(json.dumps(kw), flush=True)


def best_of(fn, n=5):
    ts = []
    for _ in range(n):
        ctx.reset_stats()
        t0 = time.perf_counter()
        out = fn()
        ts.append((time.perf_counter() - t0, ctx.stats()["kernel_ms"]))
    return out, min(t[0] for t in ts), min(t[1] for t in ts)


def config2():
    Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    sd, w, k = best_of(lambda: ctx.self_dist(qb))
    emit(config=2, op="self 2-NN (Metric_Cache build)", pairs_per_s=1e10 / (k * 1e-3), kernel_ms=k, wall_ms=w * 1e3)
    qb.set_selfdist(sd)
    (qa, ta, da, ra), w, k = best_of(lambda: ctx.match_accepted(qb, tb, 0.7), 10)
    emit(config=2, op="X1 + ratio 0.7 (fm_match_accepted)", pairs_per_s=1e10 / w, matches_per_s=len(qa) / w,
         kernel_ms=k, wall_ms=w * 1e3, frac_int8_mfma_peak=1e10 * 256 / (k * 1e-3) / 5e15)
    _, w, k = best_of(lambda: ctx.knn2(qb, tb))
    emit(config=2, op="K2 2-NN (Classic Ratio-Match core)", pairs_per_s=1e10 / w, kernel_ms=k, wall_ms=w * 1e3,
         frac_int8_mfma_peak=1e10 * 256 / (k * 1e-3) / 5e15)


def build_pair(size, n, seed, n_thumb):
    q, t = synth.image_pair(size, n, seed, n_thumb=n_thumb)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    return mc, fi


def config3():
    mc, fi = build_pair((6000, 4000), 300000, 20250003, 2000)
    stats = {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "return_arrays": True})
    get(0.7)
    stats.clear()
    m, w, k = best_of(lambda: get(0.7), 3)
    rounds, pairs = stats["rounds"] // 3, stats["pairs"] // 3
    emit(config=3, op="fastmatch.match device loop, 24 MP pair, 300k/side", matches=len(m[0]), rounds=rounds,
         wall_s=w, rounds_per_s=rounds / w, pairs_per_s=pairs / w, matches_per_s=len(m[0]) / w)


def config3_thresholds():
    """configs[2] at 15 thresholds in ONE launch (key "3t"): the reference's driver asks a pair for a list of
    thresholds (turntable.py:59-60); the runs are independent, one workgroup and one run state each."""
    mc, fi = build_pair((6000, 4000), 300000, 20250003, 2000)
    stats = {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats, "return_arrays": True})
    taus = [float(t) for t in np.linspace(0.5, 1.0, 15)]
    get(taus)
    stats.clear()
    m, w, k = best_of(lambda: get(taus), 2)
    rounds, pairs = stats["rounds"] // 2, stats["pairs"] // 2
    emit(config=3, op="15 thresholds linspace(0.5, 1.0, 15) of the 24 MP pair in one launch", runs=len(taus),
         matches=int(sum(len(r[0]) for r in m)), rounds=rounds, wall_s=w, kernel_ms=k, rounds_per_s=rounds / w, pairs_per_s=pairs / w)


def config4():
    pairs = [build_pair((1000, 1000), 12500, 20250100 + i, 600) for i in range(64)]
    for n in (8, 64):
        prepared, stats = [], {}
        fastmatch.match_many(pairs[:n], 0.7, {"context": ctx, "prepared_out": prepared, "return_arrays": True})
        res, w, k = best_of(lambda: fastmatch.match_many(pairs[:n], 0.7, {"context": ctx, "prepared": prepared,
                                                                     "stats": stats, "return_arrays": True}), 3)
        nm = sum(len(r[0]) for r in res)
        emit(config=4, op="match_many, %d x 1 MP pairs in one launch" % n, matches=nm, rounds=stats["rounds"] // 3,
             wall_s=w, rounds_per_s=stats["rounds"] / 3 / w, pairs_per_s=stats["pairs"] / 3 / w, matches_per_s=nm / w)


def config5():
    rng = np.random.default_rng(20250005)
    NT, NQ = 1000000, 10000
    Ti = synth.synth_sift(NT, rng)
    T = Ti.astype(np.float32) + rng.uniform(-0.5, 0.5, (NT, 128)).astype(np.float32)
    tb = ctx.bank(T)
    Qi = synth.synth_sift(NQ, rng)
    qb = ctx.bank(Qi.astype(np.float32) + rng.uniform(-0.5, 0.5, (NQ, 128)).astype(np.float32))
    for name, fn in (("K2", lambda: ctx.knn2(qb, tb)), ("X1", lambda: ctx.xcheck1(qb, tb))):
        _, w, k = best_of(fn, 3)
        emit(config=5, op="%s float32 non-integer (K5), 10k x 1M" % name, pairs_per_s=1e10 / (k * 1e-3), kernel_ms=k,
             frac_fp32_valu_bound=1e10 / (k * 1e-3) / 3.07e11, hbm_bytes_algorithmic=NT * 512 + NQ * 512,
             frac_hbm=(NT * 512 + NQ * 512) / (k * 1e-3) / 8e12)
    tbi, qbi = ctx.bank(Ti), ctx.bank(Qi)
    for name, fn in (("K2", lambda: ctx.knn2(qbi, tbi)), ("X1", lambda: ctx.xcheck1(qbi, tbi))):
        _, w, k = best_of(fn, 3)
        emit(config=5, op="%s same shape, integer valued -> int8 route" % name, pairs_per_s=1e10 / (k * 1e-3), kernel_ms=k,
             frac_int8_mfma_peak=1e10 * 256 / (k * 1e-3) / 5e15)


for name, fn in (("2", config2), ("3", config3), ("3t", config3_thresholds), ("4", config4), ("5", config5)):
    if (not only and name != "3t") or name in only:
        fn()
