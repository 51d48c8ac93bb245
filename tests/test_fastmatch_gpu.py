"""GPU: the product fastmatch.match()/Metric_Cache/Grid_Cache surface (HIP rounds, host
expansion loop) against the oracle's restatement of fastmatch.pyx on synthetic image pairs:
identical match lists (indices, positions, ratios), round counts and logs."""
import os

import numpy as np
import pytest

from fastmatch_amd import fastmatch, cache, synth
from oracle import fastmatch_oracle as fo

pytestmark = pytest.mark.gpu


def _build(size, n, seed, ctx, **kw):
    q, t = synth.image_pair(size, n, seed, **kw)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    oq = fo.OQuery(q["descriptors"], q["positions"], q["size"],
                   thumb={"descriptors": q["thumb_descriptors"], "positions": q["thumb_positions"],
                          "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": t["descriptors"],
          "thumb": {"descriptors": t["thumb_descriptors"], "positions": t["thumb_positions"],
                    "size": t["thumb_size"]}}
    return mc, fi, oq, ot


def _same_matches(a, b):
    assert len(a) == len(b)
    for (ia, da), (ib, db) in zip(a, b):
        assert ia == ib and da["ratio"] == db["ratio"]
        assert np.array_equal(da["positions"], db["positions"])


@pytest.mark.parametrize("size,n,opts", [
    ((800, 640), 3000, {}),                                                    # graf-sized, default options
    ((800, 640), 3000, {"grid_size": (75, 75), "grid_margin": 30, "radius": 50}),  # Fast Matching.ipynb options
    ((1000, 1000), 12500, {}),                                                 # one C4 pair
    ((611, 389), 2500, {"grid_size": (64, 48), "grid_margin": 0, "radius": 75, "thumb_strategy": lambda t: t * 1.2}),
])
def test_match_equals_oracle(ctx, size, n, opts):
    mc, fi, oq, ot = _build(size, n, seed=size[0] + n, ctx=ctx)
    assert np.array_equal(mc.original["distances"], oq.distances)
    assert np.array_equal(mc.thumb["distances"], oq.thumb["distances"])
    log, olog, stats = [], [], {}
    o = dict(opts, log=log, context=ctx, stats=stats)
    get = fastmatch.match(mc, fi, o)
    oget = fo.o_match(oq, ot, dict(opts, log=olog))
    for tau in (0.7, 0.9, 0.5):
        del log[:], olog[:]
        stats.clear()
        got, exp = get(tau), oget(tau)
        _same_matches(got, exp)
        assert stats["rounds"] == oget.rounds == len(log) == len(olog)
        for a, b in zip(log, olog):
            assert np.array_equal(a["query_pos"], b["query_pos"]) and np.array_equal(a["target_pos"], b["target_pos"])
            assert a["target_grid"] == b["target_grid"] and a["radius"] == b["radius"] and a["margin"] == b["margin"]
            assert np.array_equal(a["matches"], b["matches"]) and np.array_equal(a["ratios"], b["ratios"])
    assert len(got) > 0


def test_seeds_equal_oracle(ctx):
    mc, fi, oq, ot = _build((800, 640), 3000, seed=5, ctx=ctx)
    pos, ratios = fastmatch.match_thumbs(fi, mc, context=ctx)
    opos, oratios = fo.o_match_thumbs(oq, ot["thumb"], ot["size"])
    assert np.array_equal(pos, opos) and np.array_equal(ratios, oratios)
    assert np.all(np.diff(ratios) >= 0) and len(ratios) > 50


def test_match_position_round(ctx):
    mc, fi, oq, ot = _build((800, 640), 3000, seed=6, ctx=ctx)
    grid = cache.Grid_Cache(fi, (50, 50), fi, margin=25)
    ogrid = fo.OGrid(ot["size"], (50, 50), 25, ot["positions"], ot["descriptors"])
    for pos in [((400.7, 300.2), (420.9, 280.1)), ((10.0, 10.0), (0.0, 0.0)), ((799.0, 639.0), (800.0, 640.0))]:
        p, r, i = fastmatch.match_position(pos, mc, grid, radius=100, context=ctx)
        op, orr, oi = fo.o_match_position(pos, oq, ogrid, 100)
        assert np.array_equal(p, op) and np.array_equal(r, orr) and np.array_equal(i, oi)


# ---- device-resident expansion loop (K7) --------------------------------------------------
@pytest.mark.parametrize("size,n,opts", [
    ((800, 640), 3000, {}),
    ((800, 640), 3000, {"grid_size": (75, 75), "grid_margin": 30, "radius": 50}),
    ((1000, 1000), 12500, {}),
    ((611, 389), 2500, {"grid_size": (64, 48), "grid_margin": 0, "radius": 75, "thumb_strategy": lambda t: t * 1.2}),
    ((300, 200), 40, {}),                                  # nearly empty cells, tiny subsets
])
def test_device_loop_equals_oracle_and_host_loop(ctx, size, n, opts, monkeypatch):
    mc, fi, oq, ot = _build(size, n, seed=7 * size[0] + n, ctx=ctx, n_thumb=min(600, n))
    stats, hstats = {}, {}
    ran = []
    orig = fastmatch.run_device_loops

    def spy(*a, **k):
        r = orig(*a, **k)
        ran.append([x is not None for x in r])
        return r
    monkeypatch.setattr(fastmatch, "run_device_loops", spy)
    get = fastmatch.match(mc, fi, dict(opts, context=ctx, stats=stats))
    hget = fastmatch.match(mc, fi, dict(opts, context=ctx, stats=hstats, device_loop=False))
    oget = fo.o_match(oq, ot, dict(opts))
    for tau in (0.7, 0.95, 0.4):
        stats.clear()
        hstats.clear()
        got, host, exp = get(tau), hget(tau), oget(tau)
        assert ran and ran[-1] == [True], "device loop did not run"
        _same_matches(got, exp)
        _same_matches(host, exp)
        assert stats["rounds"] == hstats["rounds"] == oget.rounds
        assert stats["pairs"] == hstats["pairs"]


def test_device_and_host_loop_in_the_sqrt_tie_range(ctx):
    """Descriptors whose every query/target d2 is >= 4 197 200 (kat.far_image_pair): the election of
    each round is decided by OpenCV's float32-root order.  Device loop (K7), host loop (K4 rounds) and
    oracle agree on every match and on the round count."""
    from kat import far_image_pair
    q, t = synth.image_pair((400, 320), 800, 5, n_thumb=300)
    q, t = far_image_pair(q, t)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    oq = fo.OQuery(q["descriptors"], q["positions"], q["size"],
                   thumb={"descriptors": q["thumb_descriptors"], "positions": q["thumb_positions"], "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": t["descriptors"],
          "thumb": {"descriptors": t["thumb_descriptors"], "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    assert np.array_equal(mc.original["distances"], oq.distances)
    stats, hstats = {}, {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats})
    hget = fastmatch.match(mc, fi, {"context": ctx, "stats": hstats, "device_loop": False})
    oget = fo.o_match(oq, ot, {})
    for tau, on_device in ((970.0, True), (1045.0, False)):       # (the second exceeds the device's result capacity of 4 nq)
        stats.clear()
        hstats.clear()
        got, host, exp = get(tau), hget(tau), oget(tau)
        assert len(exp) > 500
        _same_matches(got, exp)
        _same_matches(host, exp)
        assert stats["rounds"] == hstats["rounds"] == oget.rounds
        if on_device:
            assert stats.get("device_loops", 0) == 1


@pytest.mark.parametrize("metric", ["chebyshev", "manhattan"])
def test_metric_option_reaches_the_radius_query_on_host_and_device(ctx, metric):
    """options["metric"] of the query cache (cache.pyx:160 -> BallTree(positions, metric), cache.pyx:276):
    the device loop's radius query, the host loop's and the oracle's select the same keypoints in the
    same order, and not the Euclidean ones."""
    q, t = synth.image_pair((800, 640), 3000, seed=31)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx, "metric": metric})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    oq = fo.OQuery(q["descriptors"], q["positions"], q["size"], metric=metric,
                   thumb={"descriptors": q["thumb_descriptors"], "positions": q["thumb_positions"], "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": t["descriptors"],
          "thumb": {"descriptors": t["thumb_descriptors"], "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    stats, hstats = {}, {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats})
    hget = fastmatch.match(mc, fi, {"context": ctx, "stats": hstats, "device_loop": False})
    oget = fo.o_match(oq, ot, {})
    exp = oget(0.7)
    _same_matches(get(0.7), exp)
    _same_matches(hget(0.7), exp)
    assert stats.get("device_loops", 0) == 1 and stats["rounds"] == hstats["rounds"] == oget.rounds
    assert stats["pairs"] == hstats["pairs"]
    # Euclidean discs hold other keypoints: the number of descriptor pairs differs
    mc2 = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                         q["thumb_positions"], q["thumb_size"], distances=mc.original["distances"],
                                         thumb_distances=mc.thumb["distances"], options={"context": ctx})
    s2 = {}
    fastmatch.match(mc2, fi, {"context": ctx, "stats": s2})(0.7)
    assert s2["pairs"] != stats["pairs"]


def test_device_loop_return_arrays_and_many_pairs(ctx):
    pairs, oracles = [], []
    for k in range(5):
        mc, fi, oq, ot = _build((640, 480), 2000 + 300 * k, seed=900 + k, ctx=ctx)
        pairs.append((mc, fi))
        oracles.append(fo.o_match(oq, ot, {}))
    prepared, stats = [], {}
    res = fastmatch.match_many(pairs, 0.7, {"context": ctx, "prepared_out": prepared, "stats": stats})
    assert len(res) == 5 and all(p["expander"] not in (None, False) for p in prepared)
    rounds = 0
    for r, og in zip(res, oracles):
        _same_matches(r, og(0.7))
        rounds += og.rounds
    assert stats["rounds"] == rounds
    # same prepared state, other threshold, array output
    res2 = fastmatch.match_many(pairs, 0.9, {"context": ctx, "prepared": prepared, "return_arrays": True})
    for (index, pos, ratio), og in zip(res2, oracles):
        exp = og(0.9)
        assert index.tolist() == [e[0] for e in exp]
        assert np.array_equal(ratio, np.array([e[1]["ratio"] for e in exp]))
        assert np.array_equal(pos, np.array([e[1]["positions"] for e in exp]).reshape(-1, 2, 2))


def test_device_loop_falls_back_when_it_cannot_run(ctx):
    # a radius that makes the query subset exceed the LDS tables of both kernels (4096 rows) with the chunked
    # variant switched off: the device reports it and match() silently replays the loop on the host, same results
    mc, fi, oq, ot = _build((400, 300), 6000, seed=77, ctx=ctx)
    stats = {}
    ctx.set_option("expand_huge", 0)
    try:
        got = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": stats})(0.7)
    finally:
        ctx.set_option("expand_huge", 1)
    exp = fo.o_match(oq, ot, {"radius": 200})(0.7)
    _same_matches(got, exp)
    assert stats.get("device_fallbacks") == 1 and "device_loops" not in stats


def test_device_loop_chunks_radius_subsets_of_any_size(ctx):
    """The reference's radius query has no size limit (cache.pyx:173-188).  Subsets beyond 4096 rows (here: nearly
    the whole 6000-keypoint image inside radius 200) stay on the device: the round takes the subset in chunks of the
    sort-key range and merges the per-train-row minimum across them (expand.hip, HUGE)."""
    mc, fi, oq, ot = _build((400, 300), 6000, seed=77, ctx=ctx)
    pos = mc.original["positions"]
    assert ((pos[:, 0] - 200.0) ** 2 + (pos[:, 1] - 150.0) ** 2 <= 200.0 ** 2).sum() > 4096
    for metric in ("minkowski", "manhattan", "chebyshev"):
        q = cache.Metric_Cache.from_arrays(mc.original["descriptors"], pos, mc.original["size"], mc.thumb["descriptors"],
                                           mc.thumb["positions"], mc.thumb["size"], options={"context": ctx, "metric": metric})
        oqm = fo.OQuery(mc.original["descriptors"], pos, mc.original["size"],
                        thumb={"descriptors": mc.thumb["descriptors"], "positions": mc.thumb["positions"], "size": mc.thumb["size"]},
                        metric=metric)
        for tau in (0.7, 0.95):
            stats, hs = {}, {}
            got = fastmatch.match(q, fi, {"context": ctx, "radius": 200, "stats": stats})(tau)
            assert stats.get("device_loops") == 1 and "device_fallbacks" not in stats
            _same_matches(got, fo.o_match(oqm, ot, {"radius": 200})(tau))
            host = fastmatch.match(q, fi, {"context": ctx, "radius": 200, "stats": hs, "device_loop": False})(tau)
            _same_matches(got, host)
            assert stats["rounds"] == hs["rounds"] and stats["pairs"] == hs["pairs"] and len(got) > 300


def test_device_loop_chunks_radius_subsets_in_the_sqrt_tie_range(ctx):
    """r05: radius subsets beyond 4096 rows of a pair under the float32-root guard (every query/target d2 >= 4 197 200,
    kat.far_image_pair) stay on the device too: the chunked round merges (d2, slot) minima per train row, and its election
    repairs the rows whose best d2 shares its float32 root with d2 + 1 (the lowest slot at either wins, as in
    cv::batchDistance's float32 comparison) -- before r05 such a pair gave the whole run back to the host loop."""
    from kat import far_image_pair
    q, t = synth.image_pair((400, 300), 6000, 79)
    q, t = far_image_pair(q, t)
    pos = q["positions"]
    assert ((pos[:, 0] - 200.0) ** 2 + (pos[:, 1] - 150.0) ** 2 <= 200.0 ** 2).sum() > 4096
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], pos, q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    oq = fo.OQuery(q["descriptors"], pos, q["size"],
                   thumb={"descriptors": q["thumb_descriptors"], "positions": q["thumb_positions"], "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": t["descriptors"],
          "thumb": {"descriptors": t["thumb_descriptors"], "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    for tau in (1100.0, 1138.0):       # (ratios are d / self distance = 2550 / 2.449 .. 2550 / 1: 1041, 1140, 1275 ...)
        stats, hs = {}, {}
        exp = fo.o_match(oq, ot, {"radius": 200})(tau)
        got = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": stats})(tau)
        assert stats.get("device_loops") == 1 and "device_fallbacks" not in stats, stats
        _same_matches(got, exp)
        host = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": hs, "device_loop": False})(tau)
        _same_matches(host, exp)
        assert stats["rounds"] == hs["rounds"] and stats["pairs"] == hs["pairs"]
        print("tie-range chunked rounds: tau %g, %d matches, %d rounds" % (tau, len(got), stats["rounds"]))
    assert len(got) > 100


@pytest.mark.parametrize("rootsift", [False, True])
def test_device_loop_chunks_by_rank_when_thousands_of_keypoints_share_one_sort_key(ctx, monkeypatch, rootsift):
    """r05: 3000 of 6000 query keypoints sit at ONE position, so every radius subset that holds them has 3000 entries with
    the same sort key (one histogram bucket larger than a 2048-row chunk; inside it the order is by keypoint index).  The
    chunked round then splits the subset by RANK in (key bits, index) order (expand.hip, huge_rank_pivots) instead of
    giving the run back to the host loop: device == host loop == oracle, no fallback; uint8 and RootSIFT-style float32."""
    from fastmatch_amd import _ffi
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1 if rootsift else 0)
    q, t = synth.image_pair((400, 300), 6000, 81)
    rng = np.random.default_rng(81)
    same = rng.choice(6000, 3000, replace=False)
    q["positions"] = q["positions"].copy()
    q["positions"][same] = (203.25, 148.5)

    def conv(d):
        if not rootsift:
            return d
        d = d.astype(np.float32)
        return np.sqrt(d / np.maximum(d.sum(1, keepdims=True), 1)).astype(np.float32)

    qd, td, qtd, ttd = conv(q["descriptors"]), conv(t["descriptors"]), conv(q["thumb_descriptors"]), conv(t["thumb_descriptors"])
    mc = cache.Metric_Cache.from_arrays(qd, q["positions"], q["size"], qtd, q["thumb_positions"], q["thumb_size"],
                                        options={"context": ctx})
    assert mc.bank(ctx).kind == (_ffi.FM_BANK_F32 if rootsift else _ffi.FM_BANK_I8)
    fi = cache.Feature_Image(t["size"], t["positions"], td, t["thumb_positions"], ttd, t["thumb_size"])
    oq = fo.OQuery(qd, q["positions"], q["size"],
                   thumb={"descriptors": qtd, "positions": q["thumb_positions"], "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": td,
          "thumb": {"descriptors": ttd, "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    for dmin in ((0,) if rootsift else (0, 1)):          # every cross-check in the run's own workgroup / every one delegated
        ctx.set_option("expand_delegate", dmin)
        try:
            for tau in (0.8, 0.95):
                stats, hs = {}, {}
                exp = fo.o_match(oq, ot, {"radius": 200})(tau)
                got = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": stats})(tau)
                assert stats.get("device_loops") == 1 and "device_fallbacks" not in stats, stats
                _same_matches(got, exp)
                host = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": hs, "device_loop": False})(tau)
                _same_matches(host, exp)
                assert stats["rounds"] == hs["rounds"] and stats["pairs"] == hs["pairs"]
                print("one sort key: rootsift %s, delegate %d, tau %g: %d matches, %d rounds" % (rootsift, dmin, tau, len(got), stats["rounds"]))
        finally:
            ctx.set_option("expand_delegate", 1500000)
    assert len(got) > 50


def test_device_loop_chunks_radius_subsets_of_float32_banks(ctx, monkeypatch):
    """RootSIFT-style float32 banks with radius subsets far beyond the float32 round's 2048 rows: the chunked variant of
    the float32 kernel (fp16 filter + exact chain per chunk, per-train-row (distance bits, slot) minimum merged across the
    chunks) == the host loop == the oracle in the device's accumulation order; with the variant off: host loop."""
    from fastmatch_amd import _ffi
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    q, t = synth.image_pair((400, 300), 6000, 78)

    def root(d):
        d = d.astype(np.float32)
        return np.sqrt(d / np.maximum(d.sum(1, keepdims=True), 1)).astype(np.float32)

    qd, td, qtd, ttd = root(q["descriptors"]), root(t["descriptors"]), root(q["thumb_descriptors"]), root(t["thumb_descriptors"])
    mc = cache.Metric_Cache.from_arrays(qd, q["positions"], q["size"], qtd, q["thumb_positions"], q["thumb_size"],
                                        options={"context": ctx})
    assert mc.bank(ctx).kind == _ffi.FM_BANK_F32
    fi = cache.Feature_Image(t["size"], t["positions"], td, t["thumb_positions"], ttd, t["thumb_size"])
    oq = fo.OQuery(qd, q["positions"], q["size"],
                   thumb={"descriptors": qtd, "positions": q["thumb_positions"], "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": td,
          "thumb": {"descriptors": ttd, "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    exp = fo.o_match(oq, ot, {"radius": 200})(0.9)
    stats, hs = {}, {}
    ctx.set_option("delegated_rounds", 0)
    got = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": stats})(0.9)
    assert stats.get("device_loops") == 1 and "device_fallbacks" not in stats
    _same_matches(got, exp)
    # (rounds of >= 1.5e6 descriptor pairs had their cross-check DELEGATED: the gathered subset in the float32 layout,
    # the float32 route's dense cross-check with the cell's rows as output rows, the run resumed at steps 4 / 5)
    assert ctx.get_option("delegated_rounds") > 0
    host = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": hs, "device_loop": False})(0.9)
    _same_matches(host, exp)
    assert stats["rounds"] == hs["rounds"] and stats["pairs"] == hs["pairs"] and len(got) > 100
    # every cross-check in the run's own workgroup (chunks), and every one delegated (small rounds take the all-pairs
    # float32 kernel behind the gather, big ones the fp16 filter): the same lists
    for dmin in (0, 1):
        ctx.set_option("expand_delegate", dmin)
        ctx.set_option("delegated_rounds", 0)
        try:
            st2 = {}
            again = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": st2})(0.9)
        finally:
            ctx.set_option("expand_delegate", 1500000)
        assert st2.get("device_loops") == 1 and "device_fallbacks" not in st2 and st2["rounds"] == stats["rounds"]
        assert (ctx.get_option("delegated_rounds") > 0) == (dmin == 1)
        _same_matches(again, exp)
    ctx.set_option("expand_huge", 0)
    try:
        fb = {}
        again = fastmatch.match(mc, fi, {"context": ctx, "radius": 200, "stats": fb})(0.9)
    finally:
        ctx.set_option("expand_huge", 1)
    assert fb.get("device_fallbacks") == 1
    _same_matches(again, exp)


def test_device_loop_on_clustered_keypoints(ctx):
    """Real SIFT keypoints crowd on texture.  A Gaussian-mixture image pair (three blobs holding half of 120k keypoints,
    peak density > 20x the mean): the largest radius subset holds more than 10 000 rows, the largest cell thousands --
    device loop == host loop == oracle, no run handed back to the host."""
    q, t = synth.image_pair((2000, 1500), 120000, seed=4100, p=0.15, n_thumb=1200, clusters=3, cluster_sigma=60.0,
                            cluster_frac=0.5)
    qpos = q["positions"]
    # (a keypoint near the densest point: the count inside its radius-100 disc)
    dens = max(int((((qpos - c) ** 2).sum(axis=1) <= 100.0 ** 2).sum()) for c in qpos[::997])
    assert dens >= 10000
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], qpos, q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    ds, hs = {}, {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": ds, "return_arrays": True})
    ctx.set_option("delegated_rounds", 0)
    index, pos, ratio = get(0.7)
    assert ds.get("device_loops") == 1 and "device_fallbacks" not in ds
    n_deleg = ctx.get_option("delegated_rounds")
    assert n_deleg > 100
    # the big rounds' cross-checks were DELEGATED to the dense kernels (option expand_delegate: rounds of >= 1.5e6 descriptor
    # pairs park the run; K1 + the election on the whole GPU; resume at steps 4 / 5); without delegation the round's own
    # workgroup does them in chunks -- same lists either way
    assert ctx.get_option("expand_delegate") == 1500000
    first = dict(ds)
    ds.clear()
    ctx.set_option("expand_delegate", 0)
    try:
        i2, p2, r2 = get(0.7)
    finally:
        ctx.set_option("expand_delegate", 1500000)
    assert ctx.get_option("delegated_rounds") == n_deleg       # (none in the second run)
    assert np.array_equal(index, i2) and np.array_equal(pos, p2) and np.array_equal(ratio, r2) and ds == first
    host = fastmatch.match(mc, fi, {"context": ctx, "stats": hs, "device_loop": False})(0.7)
    assert ds["rounds"] == hs["rounds"] and ds["pairs"] == hs["pairs"] and len(host) == len(index) > 3000
    assert index.tolist() == [e[0] for e in host]
    assert np.array_equal(ratio, np.array([e[1]["ratio"] for e in host]))
    assert np.array_equal(pos, np.array([e[1]["positions"] for e in host]).reshape(-1, 2, 2))
    oq = fo.OQuery(q["descriptors"], qpos, q["size"],
                   thumb={"descriptors": q["thumb_descriptors"], "positions": q["thumb_positions"], "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": t["descriptors"],
          "thumb": {"descriptors": t["thumb_descriptors"], "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    # (the oracle replays a prefix of the run by default -- the suite's time budget; the rounds beyond it
    # are covered by device loop == host loop above.  FM_CLUSTERED_ORACLE_ROUNDS=0: all of them.)
    cap = int(os.environ.get("FM_CLUSTERED_ORACLE_ROUNDS", "800"))
    oget = fo.o_match(oq, ot, {"max_rounds": cap} if cap else {})
    exp = oget(0.7)
    if cap and oget.rounds >= cap:
        assert len(exp) > 100
        _same_matches(host[:len(exp)], exp)
    else:
        _same_matches(host, exp)


def test_device_loop_reruns_in_the_large_capacity_kernel(ctx):
    """Radius subsets between 2049 and 4096 rows: the 2048-row kernel gives up, fm_expand_run runs
    the pair again in the 4096-row variant (256-row gather steps) -- still on the device, same
    matches as the oracle and as the host-driven loop."""
    mc, fi, oq, ot = _build((700, 500), 9000, seed=78, ctx=ctx)
    pos = mc.original["positions"]
    inside = ((pos[:, 0] - 350.0) ** 2 + (pos[:, 1] - 250.0) ** 2 <= 180.0 ** 2).sum()
    assert 2048 < inside <= 4096                     # (the interior rounds are in the big kernel's range)
    stats, hs = {}, {}
    got = fastmatch.match(mc, fi, {"context": ctx, "radius": 180, "stats": stats})(0.7)
    assert stats.get("device_loops") == 1 and "device_fallbacks" not in stats
    host = fastmatch.match(mc, fi, {"context": ctx, "radius": 180, "stats": hs, "device_loop": False})(0.7)
    _same_matches(got, host)
    assert stats["rounds"] == hs["rounds"] and stats["pairs"] == hs["pairs"]
    _same_matches(got, fo.o_match(oq, ot, {"radius": 180})(0.7))
    assert len(got) > 200


def test_device_loop_equals_host_loop_at_config3_scale(ctx):
    """BASELINE config 3 geometry at reduced keypoint count (6000 x 4000 image, 9801 cells,
    60k keypoints/side): tens of thousands of rounds, device loop == host-driven loop."""
    q, t = synth.image_pair((6000, 4000), 60000, seed=20250003, n_thumb=1500)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    ds, hs = {}, {}
    dev = fastmatch.match(mc, fi, {"context": ctx, "stats": ds})(0.7)
    host = fastmatch.match(mc, fi, {"context": ctx, "stats": hs, "device_loop": False})(0.7)
    assert ds["rounds"] == hs["rounds"] > 3000 and ds["pairs"] == hs["pairs"]
    _same_matches(dev, host)
    assert len(dev) > 5000


def test_evaluate_many_thresholds_reuses_state(ctx):
    """turntable.evaluate-style driver: precision table over thresholds, device loop and host
    loop agree, precision against the planted ground truth is high at tau 0.7."""
    from fastmatch_amd import evaluate
    pairs, scorers = [], []
    for k in range(3):
        q, t = synth.image_pair((640, 480), 2500, seed=300 + k)
        mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                            q["thumb_positions"], q["thumb_size"], options={"context": ctx})
        fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                                 t["thumb_descriptors"], t["thumb_size"])
        pairs.append((mc, fi))
        scorers.append(evaluate.planted_scorer(q["planted"], t["positions"]))
    taus = [0.5, 0.6, 0.7, 0.8, 0.9]
    dev = evaluate.evaluate(pairs, taus, scorers, {"context": ctx})
    host = evaluate.evaluate(pairs, taus, scorers, {"context": ctx, "device_loop": False})
    assert dev == host
    totals = [r["total"] for r in dev]
    assert totals == sorted(totals) and totals[0] > 0
    assert dev[2]["precision"] > 0.7


def test_many_runs_are_cut_into_launches_that_fit_the_memory(ctx, monkeypatch):
    """every (pair, threshold) run of a launch gets a run state of its own and keeps it; when the states a
    launch would create exceed the free memory the run list is cut into several launches that re-use the slots
    (fastmatch._launch_plan), an FM_ENOMEM launch is halved, and slots can be given back (Expander.trim)."""
    mcs = [_build((640, 480), 2500, seed=610 + k, ctx=ctx) for k in range(2)]
    gets = [fastmatch.match(mc, fi, {"context": ctx, "return_arrays": True}) for mc, fi, _, _ in mcs]
    taus = [0.5, 0.6, 0.7, 0.8, 0.9, 1.0]
    exs = [g.expander() for g in gets for _ in taus]
    sds = [g.seeds_for(t) for g in gets for t in taus]
    tts = [t for _ in gets for t in taus]
    st0 = {}
    ref = fastmatch.run_device_loops(ctx, exs, sds, tts, stats=st0, as_arrays=True)
    assert "device_launches" not in st0 and st0["device_loops"] == 12
    state_bytes, slots = exs[0].info()
    assert slots == 6 and state_bytes > 1 << 20
    for ex in set(exs):
        ex.trim(1)
    assert exs[0].info()[1] == 1
    real = ctx.mem_info()
    assert real[0] > 1 << 30 and real[1] >= real[0]
    monkeypatch.setattr(ctx, "mem_info", lambda: (int(4.5 * state_bytes), real[1]))     # room for two new states per launch
    st1 = {}
    got = fastmatch.run_device_loops(ctx, exs, sds, tts, stats=st1, as_arrays=True)
    assert st1.get("device_launches", 0) >= 3 and st1["device_loops"] == 12 and st1["rounds"] == st0["rounds"]
    for a, b in zip(got, ref):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert max(ex.info()[1] for ex in set(exs)) <= 3


def test_many_thresholds_of_one_pair_in_one_launch(ctx, monkeypatch):
    """get_matches([taus]) runs every threshold of the pair in ONE launch of the device loop (one
    workgroup and one run state each): each list equals the single-threshold call and the oracle, in
    any order of the thresholds, repeated (run states are reused), with a threshold twice."""
    mc, fi, oq, ot = _build((800, 640), 3000, seed=4242, ctx=ctx)
    calls = []
    orig = fastmatch.run_device_loops

    def spy(context, expanders, seeds, taus, **k):
        calls.append(len(expanders))
        return orig(context, expanders, seeds, taus, **k)
    monkeypatch.setattr(fastmatch, "run_device_loops", spy)
    stats = {}
    get = fastmatch.match(mc, fi, {"context": ctx, "stats": stats})
    taus = [0.9, 0.5, 0.7, 1.0, 0.7, 0.6, 0.8, 0.95]
    oget = fo.o_match(oq, ot, {})
    exp = {t: oget(t) for t in set(taus)}
    for rep in range(2):
        del calls[:]
        stats.clear()
        many = get(taus)
        assert calls == [len(taus)], "the thresholds did not share one launch"
        assert stats["device_loops"] == len(taus)
        for t, got in zip(taus, many):
            _same_matches(got, exp[t])
    for t in (0.7, 1.0):
        _same_matches(get(t), exp[t])                      # the single-threshold call (run slot 0) still works
    assert len(exp[1.0]) > len(exp[0.7]) >= len(exp[0.5]) > 0
    # numpy array of thresholds, empty list
    _same_matches(get(np.array([0.7]))[0], exp[0.7])
    assert get([]) == []


def test_many_runs_of_several_pairs_in_one_launch_any_order(ctx):
    """40 runs over three pairs in random order with random thresholds (some repeated) in ONE launch: every
    run equals the single-run launch of the same (pair, threshold); run states are created on demand and
    reused by the next launch."""
    from fastmatch_amd.cache import Grid_Cache
    rng = np.random.default_rng(3)
    pairs = []
    for k in range(3):
        mc, fi, _, _ = _build((640, 480), 1500 + 400 * k, seed=50 + k, ctx=ctx)
        pos, ratios = fastmatch.match_thumbs(fi, mc, context=ctx)
        grid = Grid_Cache(fi, (50, 50), fi, margin=25)
        pairs.append((fastmatch.make_expander(mc, grid, 100, ctx), pos, ratios))
    taus = [0.5, 0.6, 0.7, 0.8, 0.9, 1.0]
    single = {}
    for pi, (ex, pos, ratios) in enumerate(pairs):
        for t in taus:
            single[(pi, t)] = fastmatch.run_device_loops(ctx, [ex], [pos[ratios < t]], [t])[0]
            assert single[(pi, t)] is not None
    for rep in range(2):
        runs = [(int(rng.integers(0, 3)), float(rng.choice(taus))) for _ in range(40)]
        got = fastmatch.run_device_loops(ctx, [pairs[pi][0] for pi, _ in runs],
                                         [pairs[pi][1][pairs[pi][2] < t] for pi, t in runs], [t for _, t in runs])
        for (pi, t), g in zip(runs, got):
            _same_matches(g, single[(pi, t)])
    assert ctx.expand_slots([pairs[0][0], pairs[1][0], pairs[0][0], pairs[0][0]]) == [0, 0, 1, 2]


def test_run_states_grow_when_a_run_fills_them(ctx):
    """A run that fills its result list or pending stack is repeated by fm_expand_run in a state four
    times as large (option expand_grow, default twice): tiny first capacities give the results of the
    default ones; with growth switched off the status reaches the caller."""
    import fastmatch_amd
    from fastmatch_amd.cache import Grid_Cache
    mc, fi, oq, ot = _build((640, 480), 2000, seed=77, ctx=ctx)
    pos, ratios = fastmatch.match_thumbs(fi, mc, context=ctx)
    seeds = pos[ratios < 0.7]
    grid = Grid_Cache(fi, (50, 50), fi, margin=25)
    ref = fastmatch.run_device_loops(ctx, [fastmatch.make_expander(mc, grid, 100, ctx)], [seeds], [0.7])[0]
    assert ref is not None and len(ref) > 300
    c = fastmatch_amd.Context(0)
    c.set_option("expand_grow", 4)                          # four steps in all: 64 -> 1024 results, 2048 -> 8192+ stack entries
    mc2 = cache.Metric_Cache.from_arrays(mc.original["descriptors"], mc.original["positions"], mc.original["size"],
                                         mc.thumb["descriptors"], mc.thumb["positions"], mc.thumb["size"], options={"context": c})
    small = fastmatch.make_expander(mc2, grid, 100, c, match_cap=64, stack_cap=2048)
    got = fastmatch.run_device_loops(c, [small, small], [seeds, seeds], [0.7, 0.7])
    _same_matches(got[0], ref)
    _same_matches(got[1], ref)
    _same_matches(fastmatch.run_device_loops(c, [small], [seeds], [0.7])[0], ref)      # (the grown state is kept)
    c.set_option("expand_grow", 0)
    tiny = fastmatch.make_expander(mc2, grid, 100, c, match_cap=64, stack_cap=65536)
    res = c.expand_run([tiny], [seeds], [0.7])
    assert res[0][3] == 4                                   # FM_EXPAND_MATCH_FULL
    c.close()


def test_evaluate_puts_all_pairs_and_thresholds_into_one_launch(ctx, monkeypatch):
    from fastmatch_amd import evaluate
    pairs, scorers = [], []
    for k in range(3):
        q, t = synth.image_pair((640, 480), 2500, seed=300 + k)
        mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                            q["thumb_positions"], q["thumb_size"], options={"context": ctx})
        fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                                 t["thumb_descriptors"], t["thumb_size"])
        pairs.append((mc, fi))
        scorers.append(evaluate.planted_scorer(q["planted"], t["positions"]))
    calls = []
    orig = fastmatch.run_device_loops

    def spy(context, expanders, seeds, taus, **k):
        calls.append((len(expanders), len(set(id(e) for e in expanders))))
        return orig(context, expanders, seeds, taus, **k)
    monkeypatch.setattr(fastmatch, "run_device_loops", spy)
    taus = list(np.linspace(0.5, 1.0, 15))                  # 15 thresholds, the reference's count (Evaluate Turntable.ipynb)
    dev = evaluate.evaluate(pairs, taus, scorers, {"context": ctx})
    assert calls == [(45, 3)]                               # 3 pairs x 15 thresholds, one launch
    # the sequential driver (one launch per pair and threshold) gives the same table
    getters = [fastmatch.match(q, t, {"context": ctx, "return_arrays": True}) for q, t in pairs]
    for row, tau in zip(dev, taus):
        correct = total = 0
        for get, score in zip(getters, scorers):
            index, positions, ratio = get(float(tau))
            correct += int(score(index, positions, ratio).sum())
            total += len(index)
        assert (row["correct"], row["total"]) == (correct, total)
    assert dev[4]["precision"] > 0.7


def test_match_on_non_integer_descriptors_takes_the_float_route(ctx, monkeypatch):
    """RootSIFT-style (non-integer float32) descriptors: fastmatch.match() runs the
    device-resident loop with the float32 round, the host loop runs one float32 round launch
    per round, and both equal the oracle's transcription with the device's accumulation order."""
    from fastmatch_amd import _ffi
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    q, t = synth.image_pair((640, 480), 2500, 777)

    def root(d):
        d = d.astype(np.float32)
        return np.sqrt(d / np.maximum(d.sum(1, keepdims=True), 1)).astype(np.float32)

    qd, td, qtd, ttd = root(q["descriptors"]), root(t["descriptors"]), root(q["thumb_descriptors"]), root(t["thumb_descriptors"])
    mc = cache.Metric_Cache.from_arrays(qd, q["positions"], q["size"], qtd, q["thumb_positions"], q["thumb_size"],
                                        options={"context": ctx})
    assert mc.bank(ctx).kind == _ffi.FM_BANK_F32
    fi = cache.Feature_Image(t["size"], t["positions"], td, t["thumb_positions"], ttd, t["thumb_size"])
    oq = fo.OQuery(qd, q["positions"], q["size"],
                   thumb={"descriptors": qtd, "positions": q["thumb_positions"], "size": q["thumb_size"]})
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": td,
          "thumb": {"descriptors": ttd, "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    assert np.array_equal(mc.original["distances"], oq.distances)
    ran, batched = [], []
    orig = fastmatch.run_device_loops
    monkeypatch.setattr(fastmatch, "run_device_loops",
                        lambda *a, **k: (lambda r: (ran.append([x is not None for x in r]), r)[1])(orig(*a, **k)))
    orig_b = type(ctx).xcheck1_batched
    monkeypatch.setattr(type(ctx), "xcheck1_batched",
                        lambda self, *a, **k: (lambda r: (batched.append(int(r[0][0]) if len(r[0]) else 0), r)[1])(orig_b(self, *a, **k)))
    stats, hstats = {}, {}
    oget = fo.o_match(oq, ot, {})
    for tau in (0.8, 0.95):
        stats.clear()
        hstats.clear()
        del ran[:], batched[:]
        got = fastmatch.match(mc, fi, {"context": ctx, "stats": stats})(tau)
        assert ran == [[True]], "the device loop did not run on the float32 banks"
        host = fastmatch.match(mc, fi, {"context": ctx, "stats": hstats, "device_loop": False})(tau)
        assert len(batched) == hstats["rounds"] > 0 and -2 not in batched      # every host round = one float32 round launch
        exp = oget(tau)
        _same_matches(got, exp)
        _same_matches(host, exp)
        assert len(got) > 20 and stats["rounds"] == hstats["rounds"] == oget.rounds


def test_device_loop_float32_banks_with_an_integer_valued_side(ctx, monkeypatch):
    """Query descriptors integer valued, target descriptors not (and the reverse): the pair is
    matched on the float32 route, device loop included, and equals the oracle."""
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    q, t = synth.image_pair((500, 400), 1800, 4242)
    rng = np.random.default_rng(1)
    for noisy_query in (False, True):
        qd = q["descriptors"].astype(np.float32)
        td = t["descriptors"].astype(np.float32)
        qtd, ttd = q["thumb_descriptors"].astype(np.float32), t["thumb_descriptors"].astype(np.float32)
        if noisy_query:
            qd = qd + rng.uniform(-0.4, 0.4, qd.shape).astype(np.float32)
            qtd = qtd + rng.uniform(-0.4, 0.4, qtd.shape).astype(np.float32)
        else:
            td = td + rng.uniform(-0.4, 0.4, td.shape).astype(np.float32)
            ttd = ttd + rng.uniform(-0.4, 0.4, ttd.shape).astype(np.float32)
        mc = cache.Metric_Cache.from_arrays(qd, q["positions"], q["size"], qtd, q["thumb_positions"], q["thumb_size"],
                                            options={"context": ctx})
        fi = cache.Feature_Image(t["size"], t["positions"], td, t["thumb_positions"], ttd, t["thumb_size"])
        oq = fo.OQuery(qd, q["positions"], q["size"],
                       thumb={"descriptors": qtd, "positions": q["thumb_positions"], "size": q["thumb_size"]})
        ot = {"size": t["size"], "positions": t["positions"], "descriptors": td,
              "thumb": {"descriptors": ttd, "positions": t["thumb_positions"], "size": t["thumb_size"]}}
        stats = {}
        got = fastmatch.match(mc, fi, {"context": ctx, "stats": stats})(0.8)
        host = fastmatch.match(mc, fi, {"context": ctx, "device_loop": False})(0.8)
        oget = fo.o_match(oq, ot, {})
        exp = oget(0.8)
        _same_matches(got, exp)
        _same_matches(host, exp)
        assert len(got) > 20 and stats["rounds"] == oget.rounds


def test_match_many_with_integer_and_float32_pairs_in_one_call(ctx, monkeypatch):
    """One fm_expand_run over pairs of both descriptor kinds (two kernels behind one call): every
    pair equals its own single-pair run and the oracle."""
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    rng = np.random.default_rng(9)
    pairs, oracles = [], []
    for k in range(4):
        q, t = synth.image_pair((640, 480), 2200 + 200 * k, seed=1300 + k)
        conv = (lambda d: d) if k % 2 == 0 else (lambda d: d.astype(np.float32) + rng.uniform(-0.4, 0.4, d.shape).astype(np.float32))
        qd, td, qtd, ttd = conv(q["descriptors"]), conv(t["descriptors"]), conv(q["thumb_descriptors"]), conv(t["thumb_descriptors"])
        mc = cache.Metric_Cache.from_arrays(qd, q["positions"], q["size"], qtd, q["thumb_positions"], q["thumb_size"],
                                            options={"context": ctx})
        fi = cache.Feature_Image(t["size"], t["positions"], td, t["thumb_positions"], ttd, t["thumb_size"])
        pairs.append((mc, fi))
        oq = fo.OQuery(qd, q["positions"], q["size"],
                       thumb={"descriptors": qtd, "positions": q["thumb_positions"], "size": q["thumb_size"]})
        ot = {"size": t["size"], "positions": t["positions"], "descriptors": td,
              "thumb": {"descriptors": ttd, "positions": t["thumb_positions"], "size": t["thumb_size"]}}
        oracles.append(fo.o_match(oq, ot, {}))
    prepared = []
    res = fastmatch.match_many(pairs, 0.8, {"context": ctx, "prepared_out": prepared})
    assert all(p["expander"] not in (None, False) for p in prepared)
    for (mc, fi), r, og in zip(pairs, res, oracles):
        _same_matches(r, og(0.8))
        _same_matches(r, fastmatch.match(mc, fi, {"context": ctx})(0.8))
        assert len(r) > 20


@pytest.mark.parametrize("f32_filter", [0, 1])
def test_float32_delegated_round_on_the_last_cell_of_the_bank(ctx, monkeypatch, f32_filter):
    """a delegated round's cross-check runs the float32 route on a VIEW of the target bank (one cell's
    rows from an arbitrary first row, clipped to the bank's allocation), so the view's padded size is not a multiple
    of the all-pairs kernel's 64-row tile: on the LAST non-empty cell K5 read up to 63 rows past the array.  A
    RootSIFT-style pair whose keypoints crowd into the image's last cell (620 x 430: the last column and row of cells
    are 20 and 30 px wide), all-pairs kernel alone (f32_filter 0) and behind the fp16 filter: device loop with
    delegated rounds == host loop, nothing faults."""
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    q, t = synth.image_pair((620, 430), 9000, 9105, p=0.25)
    rng = np.random.default_rng(9105)
    crowd = rng.choice(9000, 6000, replace=False)
    for side, sh in ((t, (0.0, 0.0)), (q, (-3.0, 2.0))):
        side["positions"][crowd] = np.stack([rng.uniform(585, 619, 6000), rng.uniform(385, 429, 6000)], axis=1) + np.array(sh)

    def root(d):
        d = d.astype(np.float32)
        return np.sqrt(d / np.maximum(d.sum(1, keepdims=True), 1)).astype(np.float32)

    mc = cache.Metric_Cache.from_arrays(root(q["descriptors"]), q["positions"], q["size"], root(q["thumb_descriptors"]),
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], root(t["descriptors"]), t["thumb_positions"],
                             root(t["thumb_descriptors"]), t["thumb_size"])
    old = ctx.get_option("f32_filter")
    ctx.set_option("f32_filter", f32_filter)
    ctx.set_option("delegated_rounds", 0)
    try:
        ds, hs = {}, {}
        got = fastmatch.match(mc, fi, {"context": ctx, "stats": ds})(0.8)
        assert ds.get("device_loops") == 1 and "device_fallbacks" not in ds
        assert ctx.get_option("delegated_rounds") > 0
        host = fastmatch.match(mc, fi, {"context": ctx, "stats": hs, "device_loop": False})(0.8)
    finally:
        ctx.set_option("f32_filter", old)
    _same_matches(got, host)
    assert ds["rounds"] == hs["rounds"] and len(got) > 100


def _same_log(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x["query_pos"], y["query_pos"]) and np.array_equal(x["target_pos"], y["target_pos"])
        assert x["target_grid"] == y["target_grid"] and x["radius"] == y["radius"] and x["margin"] == y["margin"]
        assert np.array_equal(x["matches"], y["matches"]) and np.array_equal(x["ratios"], y["ratios"])
        assert np.asarray(x["matches"]).shape == np.asarray(y["matches"]).shape


@pytest.mark.parametrize("size,n,opts", [
    ((800, 640), 3000, {}),
    ((611, 389), 2500, {"grid_size": (64, 48), "grid_margin": 0, "radius": 75}),
    ((1000, 1000), 12500, {"radius": 140}),                   # rounds beyond 512 slots: the compacted accepted list
])
def test_log_is_written_by_the_device_loop(ctx, size, n, opts):
    """options["log"] (the README's flow, README.md:47-49; fastmatch.pyx:79-80, 172-180) no longer sends a run to the
    host loop: the kernel records every processed round and the records rebuilt from it equal the host loop's and the
    oracle's entry by entry -- including Grid_Cache.last going stale once every visited cell is cached (second threshold
    on the same closure) and rounds on cells without features."""
    mc, fi, oq, ot = _build(size, n, seed=size[1] + n, ctx=ctx)
    dlog, hlog, olog, ds, hs = [], [], [], {}, {}
    get = fastmatch.match(mc, fi, dict(opts, log=dlog, context=ctx, stats=ds))
    hget = fastmatch.match(mc, fi, dict(opts, log=hlog, context=ctx, stats=hs, device_loop=False))
    oget = fo.o_match(oq, ot, dict(opts, log=olog))
    for tau in (0.8, 0.6):
        got, host, exp = get(tau), hget(tau), oget(tau)
        _same_matches(got, host)
        _same_matches(got, exp)
        assert "device_fallbacks" not in ds
    assert ds["device_loops"] == 2 and len(dlog) == ds["rounds"] == hs["rounds"] > 20
    _same_log(dlog, hlog)
    _same_log(dlog, olog)
    assert any(len(e["ratios"]) > 0 for e in dlog)


def test_log_of_several_thresholds_in_one_launch_and_growing_log_arrays(ctx):
    """A list of thresholds = several runs of one pair in ONE launch, each with a log of its own, appended in the list's
    order; log arrays that start far too small (8 records) grow until the run fits."""
    mc, fi, oq, ot = _build((800, 640), 3000, seed=77, ctx=ctx)
    dlog, hlog, ds = [], [], {}
    get = fastmatch.match(mc, fi, {"log": dlog, "context": ctx, "stats": ds})
    get.expander().set_log(True, first_capacity=8)
    hget = fastmatch.match(mc, fi, {"log": hlog, "context": ctx, "device_loop": False})
    taus = [0.9, 0.5, 0.7]
    got = get(taus)
    host = [hget(t) for t in taus]
    for g, h in zip(got, host):
        _same_matches(g, h)
    assert ds["device_loops"] == 3 and "device_fallbacks" not in ds
    _same_log(dlog, hlog)


def test_log_on_clustered_keypoints_with_delegated_rounds(ctx):
    """Chunked rounds (subsets beyond the LDS tables) and rounds parked for the dense kernels keep the log in step:
    the records of a clustered pair equal the host loop's."""
    q, t = synth.image_pair((1200, 900), 40000, seed=4242, p=0.15, n_thumb=900, clusters=2, cluster_sigma=50.0, cluster_frac=0.5)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
    dlog, hlog, ds, hs = [], [], {}, {}
    ctx.set_option("delegated_rounds", 0)
    got = fastmatch.match(mc, fi, {"log": dlog, "context": ctx, "stats": ds})(0.7)
    assert ds.get("device_loops") == 1 and "device_fallbacks" not in ds and ctx.get_option("delegated_rounds") > 0
    host = fastmatch.match(mc, fi, {"log": hlog, "context": ctx, "stats": hs, "device_loop": False})(0.7)
    _same_matches(got, host)
    _same_log(dlog, hlog)
    assert max(len(e["ratios"]) for e in dlog) > 512


@pytest.mark.parametrize("kind", ["u8", "rootsift"])
def test_round_with_more_accepted_matches_than_the_lds_lists_hold(ctx, monkeypatch, kind):
    """r05: a chunked round that ACCEPTS more than 2048 matches (until r04: FM_EXPAND_SUBSET_FULL -> host loop).  Thousands of
    keypoints of both images crowd into one cell and the threshold accepts every cross-checked match (tau 50), so the first
    rounds on that cell keep 3000+ matches: taken in blocks of the slot range, pushed in slot order and reversed once, results
    and log == host loop, nothing handed back."""
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1 if kind == "rootsift" else 0)
    q, t = synth.image_pair((620, 430), 9000, 9107, p=0.6, sigma=3.0)
    rng = np.random.default_rng(9107)
    crowd = np.flatnonzero(q["planted"] >= 0)[:4500]
    tp = np.stack([rng.uniform(585, 619, len(crowd)), rng.uniform(385, 429, len(crowd))], axis=1)
    t["positions"][q["planted"][crowd]] = tp
    q["positions"][crowd] = tp + np.array([-3.0, 2.0])
    conv = (lambda d: d) if kind == "u8" else (lambda d: np.sqrt(d.astype(np.float32) / np.maximum(d.astype(np.float32).sum(1, keepdims=True), 1)).astype(np.float32))
    mc = cache.Metric_Cache.from_arrays(conv(q["descriptors"]), q["positions"], q["size"], conv(q["thumb_descriptors"]),
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], conv(t["descriptors"]), t["thumb_positions"], conv(t["thumb_descriptors"]), t["thumb_size"])
    dlog, hlog, ds, hs = [], [], {}, {}
    got = fastmatch.match(mc, fi, {"context": ctx, "stats": ds, "log": dlog})(50.0)
    assert ds.get("device_loops") == 1 and "device_fallbacks" not in ds
    host = fastmatch.match(mc, fi, {"context": ctx, "stats": hs, "log": hlog, "device_loop": False})(50.0)
    _same_matches(got, host)
    assert ds["rounds"] == hs["rounds"]
    _same_log(dlog, hlog)
    assert max(len(e["ratios"]) for e in dlog) > 2048
