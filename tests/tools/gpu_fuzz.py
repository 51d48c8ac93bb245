"""Time-boxed differential fuzz of the C-ABI operators and of fastmatch.match() against the oracle.
Random shapes (1 row .. tens of thousands, ragged, duplicates, tie-range rows), integer and float32
routes, random per-context options, random grid / radius / metric / threshold options of the
expansion loop.  Stops at the first difference and prints the seed that reproduces it.

    python tests/tools/gpu_fuzz.py [seconds] [seed]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, cache, fastmatch
import oracle
from oracle import fastmatch_oracle as fo
from kat import far_banks, far_image_pair

ctx = None                      # set by run()
ONLY = os.environ.get("FM_FUZZ_ONLY", "")      # "match": image pairs through fastmatch.match() only
eq = lambda a, b: a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def size(rng, big):
    k = rng.integers(0, 4)
    return int([rng.integers(1, 8), rng.integers(1, 200), rng.integers(1, 3000), rng.integers(1, big)][k])


def banks(rng):
    nq, nt = size(rng, 30000), size(rng, 60000)
    kind = rng.choice(["sift", "planted", "dups", "far", "nonint", "lowrange"])
    if kind == "far":
        Q, T = far_banks(nq, max(nt, 2), rng, int(rng.integers(1, 14)), int(rng.integers(1, 4)))
    elif kind == "planted":
        Q, T, _ = synth.planted_pair(nq, nt, int(rng.integers(1 << 30)))
    else:
        Q, T = synth.synth_sift(nq, rng), synth.synth_sift(nt, rng)
    if kind == "dups" and nt > 1:                       # equal rows: index tie-breaks, self distance 0
        for _ in range(int(rng.integers(1, 20))):
            i, j = rng.integers(0, nt, 2)
            T[i] = T[j]
            Q[rng.integers(0, nq)] = T[j]
    if kind == "lowrange":
        Q, T = Q // 64, T // 64                          # many exact d2 ties
    if kind == "nonint":
        Q = (Q.astype(np.float32) + rng.uniform(-0.5, 0.5, Q.shape).astype(np.float32)) * np.float32(rng.choice([1.0, 1 / 512.0]))
        T = (T.astype(np.float32) + rng.uniform(-0.5, 0.5, T.shape).astype(np.float32)) * np.float32(rng.choice([1.0, 1 / 512.0]))
    elif rng.integers(0, 3) == 0:
        Q, T = Q.astype(np.float32), T.astype(np.float32)   # integer valued float32: int8 route
    return kind, Q, T


def fuzz_operators(rng):
    kind, Q, T = banks(rng)
    for k, v in (("nsplit", int(rng.choice([0, 0, 1, 3, 8, 13]))), ("nbuf", int(rng.choice([0, 2, 3]))),
                 ("coop", int(rng.integers(0, 2))), ("f32_filter", int(rng.choice([0, 0, 1, 2]))),
                 ("k1_order", int(rng.choice([0, 0, 1, 2]))), ("bound_every", int(rng.choice([1, 2, 16, 16, 64, 1024]))),
                 # r05: the triangular self sweep on every size (2 = also below its 32768-row default), random piece lengths
                 ("self_tri", int(rng.choice([0, 1, 2, 2]))), ("tri_stages", int(rng.choice([0, 0, 4, 5, 17, 40])))):
        ctx.set_option(k, v)
    if rng.integers(0, 4) == 0 and kind != "far":         # the train bank as a device-side gather of uploaded rows
        m = rng.integers(0, len(T), size=size(rng, 40000)).astype(np.int32)
        qb, tb, T = ctx.bank(Q), ctx.bank_gather(T, m), T[m]
    else:
        qb, tb = ctx.bank(Q), ctx.bank(T)
    order = 1 if kind == "nonint" else 0                 # the device's fixed float32 accumulation order
    what = rng.choice(["knn2", "xcheck", "accepted", "selfdist", "batch", "knnk"])
    tag = (what, kind, Q.shape[0], T.shape[0], str(Q.dtype))
    if what == "knn2":
        idx, d = ctx.knn2(qb, tb)
        oi, od = oracle.bf_knn(Q, T, 2, order=order)
        assert eq(idx, oi) and eq(d, od), tag
    elif what == "knnk":                                     # r06: fm_knn, the rest of bf_match's signature (k up to 8)
        k = int(rng.integers(1, 9))
        nq = min(Q.shape[0], 3000)                           # (the k > 2 kernels run on the vector ALU: keep the problem small)
        nt = min(T.shape[0], 6000)
        qs, ts = ctx.bank(Q[:nq]), ctx.bank(T[:nt])
        idx, d = ctx.knn(qs, ts, k)
        oi, od = oracle.bf_knn(Q[:nq], T[:nt], k, order=order)
        assert eq(idx, oi) and eq(d, od), tag + (k,)
        qs.close(); ts.close()
    elif what == "xcheck":
        for a, b, A, B in ((qb, tb, Q, T), (tb, qb, T, Q)):
            t, x = ctx.xcheck1(a, b)
            ot, ox = oracle.bf_xcheck1(A, B, order=order)
            assert eq(t, ot) and eq(x, ox), tag
    elif what == "selfdist":
        assert eq(ctx.self_dist(qb), oracle.self_dist(Q, order=order)), tag
        # r04: the batched form (both banks + a copy of the first: banks of one size share a launch) and a refill
        qc = ctx.bank(Q)
        for g, M in zip(ctx.self_dist_batch([qb, tb, qc]), (Q, T, Q)):
            assert eq(g, oracle.self_dist(M, order=order)), tag
        if Q.dtype == np.uint8 and Q.shape[0] > 1:
            m = int(rng.integers(1, Q.shape[0] + 1))
            src = ctx.pinned_empty((m, Q.shape[1]), np.uint8)
            src[:] = Q[rng.permutation(Q.shape[0])[:m]]
            qc.refill_async(src)
            ctx.upload_fence()
            (g,) = ctx.self_dist_batch([qc])
            assert eq(g, oracle.self_dist(np.array(src), order=order)), tag + ("refill", m)
    else:
        sd = oracle.self_dist(Q, order=order)
        qb.set_selfdist(sd)
        tau = float(rng.choice([0.5, 0.7, 0.9, 1.0, 1.5, 1e9]))
        ot, ox = oracle.bf_xcheck1(Q, T, order=order)
        rows = np.nonzero(ot >= 0)[0]
        ratio, passed = oracle.ratio_filter(ox[rows], sd, tau, qrows=rows)
        exp = (rows[passed].astype(np.int32), ot[rows][passed].astype(np.int32), ox[rows][passed], ratio[passed])
        if what == "accepted":
            got = ctx.match_accepted(qb, tb, tau)
        else:
            n, cap = int(rng.integers(1, 5)), Q.shape[0]
            plist = [(qb, tb)] * n
            if kind != "nonint" and Q.dtype == np.uint8 and rng.integers(0, 2):
                # r05: pairs of OTHER sizes between them (one batched launch takes pairs of any sizes: leading rows of the
                # same banks, 30 .. 100 % of them, self distances of their own)
                for _ in range(int(rng.integers(1, 4))):
                    nq2, nt2 = max(1, int(Q.shape[0] * rng.uniform(0.3, 1.0))), max(1, int(T.shape[0] * rng.uniform(0.3, 1.0)))
                    q2, t2 = ctx.bank(Q[:nq2]), ctx.bank(T[:nt2])
                    q2.set_selfdist(oracle.self_dist(Q[:nq2], order=order))
                    plist.insert(int(rng.integers(0, len(plist) + 1)), (q2, t2))
            outs = [(ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.float32),
                     ctx.pinned_empty(cap, np.float64)) for _ in plist]
            cnts = [ctx.pinned_empty(1, np.int64) for _ in plist]
            ctx.match_accepted_batch(plist, tau, outs, cnts)
            ctx.sync()
            k = int(rng.choice([i for i, pr in enumerate(plist) if pr[0] is qb]))
            got = tuple(a[:int(cnts[k][0])] for a in outs[k])
        for g, e in zip(got, exp):
            assert eq(np.asarray(g), np.asarray(e)), tag
    return tag


def fuzz_match(rng):
    w, h = int(rng.integers(120, 1100)), int(rng.integers(120, 1100))
    n = int(rng.integers(20, 9000))
    opts = {}
    if rng.integers(0, 2):
        opts["grid_size"] = (int(rng.integers(30, 120)), int(rng.integers(30, 120)))
    if rng.integers(0, 2):
        opts["grid_margin"] = int(rng.integers(0, 45))
    if rng.integers(0, 2):
        opts["radius"] = int(rng.integers(20, 160))
    elif rng.integers(0, 4) == 0:
        opts["radius"] = int(rng.integers(160, 400))          # subsets beyond the LDS tables: the 4096-row and the chunked kernel
    if rng.integers(0, 3) == 0:
        opts["metric"] = str(rng.choice(["euclidean", "chebyshev", "manhattan"]))
    seed = int(rng.integers(1 << 30))
    kw = {}
    if rng.integers(0, 3) == 0:                               # r04: clustered keypoints
        kw = {"clusters": int(rng.integers(1, 5)), "cluster_sigma": float(rng.uniform(8.0, 60.0)), "cluster_frac": float(rng.uniform(0.2, 0.9))}
    q, t = synth.image_pair((w, h), n, seed, n_thumb=min(600, n), **kw)
    if rng.integers(0, 10) == 0:                              # r05: a share of the query keypoints at ONE position (one sort key:
        k = int(len(q["positions"]) * rng.uniform(0.2, 0.8))  # the chunked round's split by rank), sometimes of the target's too
        q["positions"] = q["positions"].copy()
        q["positions"][rng.choice(len(q["positions"]), k, replace=False)] = (float(rng.uniform(0, w - 1)), float(rng.uniform(0, h - 1)))
        if rng.integers(0, 3) == 0:
            t["positions"] = t["positions"].copy()
            t["positions"][rng.choice(len(t["positions"]), len(t["positions"]) // 3, replace=False)] = (float(rng.uniform(0, w - 1)), float(rng.uniform(0, h - 1)))
    rootsift = rng.integers(0, 5) == 0                        # r04: float32 descriptors (the float32 round, its chunks, its delegation)
    if rootsift:
        for side in (q, t):
            for k in ("descriptors", "thumb_descriptors"):
                d = side[k].astype(np.float32)
                side[k] = np.sqrt(d / np.maximum(d.sum(axis=1, keepdims=True), 1.0)).astype(np.float32)
    far = (not rootsift) and rng.integers(0, 8) == 0          # r05: every d2 in the float32-root tie range (kat.far_image_pair)
    # r06: an integer-valued query bank against a target whose descriptors are NOT all integer valued (a share of its rows
    # shifted by a fraction): the pair runs on the float32 route through the query bank's float32 twin
    mixed = (not rootsift) and (not far) and rng.integers(0, 8) == 0
    if mixed:
        for k in ("descriptors", "thumb_descriptors"):
            d = t[k].astype(np.float32)
            rows = rng.random(len(d)) < rng.uniform(0.05, 0.9)
            d[rows] += np.float32(rng.choice([0.25, 0.5, 0.125]))
            t[k] = d
    fo.FLOAT_ORDER = 1 if (rootsift or mixed) else 0          # (the oracle in the device's accumulation order for those)
    if far:
        q, t = far_image_pair(q, t, seed=seed)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options=dict(opts, context=ctx))
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    # (the oracle wants both sides in one dtype: for the mixed case it gets the query's integers as float32 -- the same numbers)
    oqd = (lambda a: a.astype(np.float32)) if mixed else (lambda a: a)
    oq = fo.OQuery(oqd(q["descriptors"]), q["positions"], q["size"],
                   thumb={"descriptors": oqd(q["thumb_descriptors"]), "positions": q["thumb_positions"], "size": q["thumb_size"]},
                   **({"metric": opts["metric"]} if "metric" in opts else {}))
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": t["descriptors"],
          "thumb": {"descriptors": t["thumb_descriptors"], "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    stats = {}
    # r04: big rounds park for a dense cross-check from this many descriptor pairs on (1 = every chunked round, 0 = never)
    ctx.set_option("expand_delegate", int(rng.choice([0, 1, 200000, 1500000])))
    # r05: every other problem asks for the per-round log (the kernel writes it; fastmatch.pyx:79-80, 172-180)
    log, olog = ([], []) if seed % 2 == 0 else (None, None)
    get = fastmatch.match(mc, fi, dict(opts, context=ctx, stats=stats, device_loop=bool(rng.integers(0, 4)), log=log))
    oget = fo.o_match(oq, ot, dict(opts, log=olog))
    taus = sorted(float(x) for x in rng.choice([0.3, 0.5, 0.6, 0.7, 0.8, 0.9, 0.97], int(rng.integers(1, 4)), replace=False))
    if far:                                                   # (ratios there are ~2550 / self distance: 1041, 1140, 1275, 1472 ...)
        taus = sorted(float(x) for x in rng.choice([900.0, 1045.0, 1100.0, 1150.0, 1300.0], int(rng.integers(1, 3)), replace=False))
        if n > 1200:
            taus = [x for x in taus if x < 1140.0] or [1100.0]         # (above that nearly every row is accepted: minutes of oracle)
    tag = ("match", (w, h), n, seed, sorted(opts.items()), taus, sorted(kw.items()), "rootsift" if rootsift else ("far" if far else ("mixed" if mixed else "u8")))
    if rng.integers(0, 2) and len(taus) > 1:
        got_all = get(taus)
    else:
        got_all = [get(x) for x in taus]
    for x, got in zip(taus, got_all):
        exp = oget(x)
        assert len(got) == len(exp), tag + (x, len(got), len(exp))
        for (ia, da), (ib, db) in zip(got, exp):
            assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"]), tag + (x,)
    if log is not None:
        assert len(log) == len(olog), tag + ("log", len(log), len(olog))
        for a, b in zip(log, olog):
            assert np.array_equal(a["query_pos"], b["query_pos"]) and np.array_equal(a["target_pos"], b["target_pos"]), tag + ("log",)
            assert a["target_grid"] == b["target_grid"] and a["radius"] == b["radius"] and a["margin"] == b["margin"], tag + ("log",)
            assert np.array_equal(a["matches"], b["matches"]) and np.array_equal(a["ratios"], b["ratios"]), tag + ("log",)
    return tag


def run(budget, seed0, max_problems=None, context=None):
    """Random problems seed0, seed0 + 1, ... until the time budget or the problem count is used up.
    Returns (problems, counts by kind); raises AssertionError at the first difference."""
    global ctx
    ctx = context if context is not None else fm.Context(0)
    saved = {k: ctx.get_option(k) for k in ("nsplit", "nbuf", "coop", "f32_filter", "k1_order", "bound_every", "expand_delegate", "self_tri", "tri_stages")}
    counts = {}
    t0, it = time.time(), 0
    try:
        while time.time() - t0 < budget and (max_problems is None or it < max_problems):
            rng = np.random.default_rng(seed0 + it)
            fn = fuzz_match if (it % 4 == 3 or ONLY == "match") else fuzz_operators
            try:
                tag = fn(rng)
            except Exception:
                print("FUZZ FAILURE at seed %d (iteration %d of base %d): python tests/tools/gpu_fuzz.py 1 %d"
                      % (seed0 + it, it, seed0, seed0 + it), flush=True)
                raise
            counts[str(tag[0])] = counts.get(str(tag[0]), 0) + 1
            it += 1
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
        fo.FLOAT_ORDER = 0
    return it, counts


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    t0 = time.time()
    n, counts = run(budget, seed0)
    print("fuzz ok: %d problems in %.0f s, base seed %d: %s" % (n, time.time() - t0, seed0, sorted(counts.items())))
