"""GPU: Metric_Cache self distances as a masked-diagonal top-1 (fm_self_dist, fm_self_dist_batch), the refill of
an existing bank (fm_bank_refill_u8_async) and the K1 workgroup orders -- against the oracle, which keeps the
reference's literal form: bf_match(d, d, k = 2) then r[1].distance (cache.pyx:250-252, 271-273)."""
import numpy as np
import pytest

import oracle
from fastmatch_amd import synth, _ffi
from kat import far_banks

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 2], ids=["masked-full-sweep", "triangular-sweep"])
def sweep(ctx, request):
    """Both forms of the integer route's self sweep on every case: option "self_tri" 0 = the masked full sweep,
    2 = the triangular one whatever the size (the default, 1, takes it from 32768 padded rows on)."""
    old = ctx.get_option("self_tri")
    ctx.set_option("self_tri", request.param)
    yield request.param
    ctx.set_option("self_tri", old)


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 64, 65, 127, 128, 129, 511, 512, 513, 1000, 4097, 33000])
def test_self_dist_sizes_u8(ctx, sweep, n):
    """Every position of the diagonal inside a wave's two 32-row units, banks with and without padding rows,
    the 4-wave and the 8-wave kernel (33 000 rows), one row -> +inf."""
    D = synth.synth_sift(n, np.random.default_rng(n))
    sd = ctx.self_dist(ctx.bank(D))
    assert _eq(sd, oracle.self_dist(D))
    if n == 1:
        assert np.isinf(sd[0])


def test_self_dist_duplicates_and_zero_rows(ctx, sweep):
    """A duplicate's 0 is the value, whichever index it has; all-equal banks; all-zero rows (the padding rows'
    twin: a zero row must still beat them)."""
    rng = np.random.default_rng(5)
    D = synth.synth_sift(700, rng)
    D[10] = D[500]            # duplicate with a higher index
    D[650] = D[3]             # ... with a lower one
    D[64:67] = D[63]          # a run across a 64-row wave boundary
    D[200] = 0
    D[420] = 0
    sd = ctx.self_dist(ctx.bank(D))
    assert _eq(sd, oracle.self_dist(D))
    assert sd[10] == 0 and sd[500] == 0 and sd[3] == 0 and sd[650] == 0 and sd[200] == 0 and np.all(sd[63:67] == 0)
    same = np.repeat(synth.synth_sift(1, rng), 300, axis=0)
    assert np.all(ctx.self_dist(ctx.bank(same)) == 0)
    zeros = np.zeros((130, 128), np.uint8)
    assert np.all(ctx.self_dist(ctx.bank(zeros)) == 0)
    two = np.zeros((2, 128), np.uint8)
    two[1, 0] = 3
    assert ctx.self_dist(ctx.bank(two)).tolist() == [3.0, 3.0]


def test_self_dist_short_dim_and_extremes(ctx, sweep):
    rng = np.random.default_rng(6)
    for dim in (1, 7, 64, 127):
        D = rng.integers(0, 256, (300, dim), dtype=np.uint8)
        assert _eq(ctx.self_dist(ctx.bank(D)), oracle.self_dist(D)), dim
    D = rng.choice(np.array([0, 255], np.uint8), (520, 128))
    assert _eq(ctx.self_dist(ctx.bank(D)), oracle.self_dist(D))


def test_self_dist_in_the_sqrt_tie_range(ctx, sweep):
    """Only the VALUE is kept: sqrtf is monotone, so the smallest float32 root is the root of the smallest d2
    and no tie repair is needed -- banks whose every distance lies where two d2 share a root."""
    rng = np.random.default_rng(11)
    Q, T = far_banks(300, 200, rng)
    for D in (Q, T, np.concatenate([Q[:1], T]), np.concatenate([T[:1], Q])):
        assert _eq(ctx.self_dist(ctx.bank(D)), oracle.self_dist(D))
    # a handful of rows that are ALL far from each other: nearest other row at d2 ~ 4.2e6 .. 8e6
    hit = 0
    for seed in range(120):
        r = np.random.default_rng(seed)
        n = 2 + seed % 3
        D = (r.integers(0, 2, (n, 128)) * 254 + r.integers(0, 2, (n, 128))).astype(np.uint8)
        sd = ctx.self_dist(ctx.bank(D))
        assert _eq(sd, oracle.self_dist(D))
        hit += int((sd * sd >= 4197200).sum())
    assert hit > 20


@pytest.mark.parametrize("n", [1, 2, 100, 700, 2100])
def test_self_dist_float32_route(ctx, n):
    """Non-integer float32 banks: the float32 route with the diagonal masked (K5 for small banks, the fp16 filter +
    exact rescoring from 4e6 pairs on), bit-identical to the oracle's order-1 chain."""
    rng = np.random.default_rng(40 + n)
    D = (synth.synth_sift(n, rng).astype(np.float32) + rng.uniform(-0.4, 0.4, (n, 128)).astype(np.float32))
    if n >= 100:
        D[7] = D[60]
        D[n - 1] = D[n // 2]
    sd = ctx.self_dist(ctx.bank(D))
    assert _eq(sd, oracle.self_dist(D, order=1))


@pytest.mark.parametrize("n,stages", [(2, 0), (130, 0), (512, 0), (513, 4), (700, 0), (1023, 5), (1537, 0), (4100, 17), (9000, 0)])
def test_self_dist_float32_triangular_sweep_small_sizes(n, stages):
    """r06: the float32 route's triangular self sweep -- every tile above the diagonal once, the
    column direction as a filter against the streamed rows' bounds + per-row candidate lists, exact rescoring -- forced on
    ("self_tri" 2) for sizes around the chunk (512) and stage (128) boundaries and several piece lengths: the oracle's
    order-1 chain bit for bit, duplicates (distance 0) included, and the masked full sweep's bits."""
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    c.set_option("f32_filter", 2)
    rng = np.random.default_rng(600 + n)
    D = (synth.synth_sift(n, rng).astype(np.float32) + rng.uniform(-0.4, 0.4, (n, 128)).astype(np.float32))
    if n >= 100:
        D[7] = D[60]
        D[n - 1] = D[n // 2]
    b = c.bank(D)
    before = c.f32_filter_stats()[0]
    c.set_option("self_tri", 2)
    c.set_option("tri_stages", stages)
    tri = c.self_dist(b)
    c.set_option("self_tri", 0)
    full = c.self_dist(b)
    assert c.f32_filter_stats()[0] == before + 2
    assert _eq(tri, oracle.self_dist(D, order=1)) and _eq(tri, full)
    c.close()


def test_self_dist_float32_triangular_sweep_at_size_and_on_near_duplicates():
    """The sizes the rule sends there by itself (from 65536 padded rows): 70 001 rows against the masked sweep (all rows) and
    the oracle (a row sample); a bank of near duplicates, where everything is inside the margin and the lists overflow into
    rescans or the whole call into K5 -- same bits either way."""
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    rng = np.random.default_rng(701)
    n = 70001
    D = (synth.synth_sift(n, rng).astype(np.float32) + rng.uniform(-0.4, 0.4, (n, 128)).astype(np.float32))
    D[11] = D[40000]; D[n - 1] = D[123]
    b = c.bank(D)
    assert c.get_option("self_tri") == 1
    tri = c.self_dist(b)                                    # (the default rule: triangular at this size)
    c.set_option("self_tri", 0)
    full = c.self_dist(b)
    assert _eq(tri, full) and tri[11] == 0.0 and tri[n - 1] == 0.0
    rows = np.sort(rng.choice(n, 600, replace=False))
    oi, od = oracle.bf_knn(D[rows], D, 2, order=1)
    assert _eq(tri[rows], od[:, 1].astype(np.float64))
    assert c.f32_filter_stats()[1] == 0                     # neither call needed the all-pairs kernel
    base = synth.synth_sift(40, rng).astype(np.float32)
    E = np.repeat(base, 60, axis=0) + rng.normal(0, 1e-3, (2400, 128)).astype(np.float32)
    c.set_option("f32_filter", 2)
    c.set_option("self_tri", 2)
    assert _eq(c.self_dist(c.bank(E)), oracle.self_dist(E, order=1))
    c.close()


def test_self_dist_float32_filter_forced_and_rescans(ctx):
    """The fp16 filter on a bank of near duplicates (many rows inside the margin -> rescans) with the option that
    sends every call through it."""
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    c.set_option("f32_filter", 2)
    rng = np.random.default_rng(77)
    base = synth.synth_sift(40, rng).astype(np.float32)
    D = np.repeat(base, 30, axis=0) + rng.normal(0, 1e-3, (1200, 128)).astype(np.float32)
    D[5] = D[900]
    assert _eq(c.self_dist(c.bank(D)), oracle.self_dist(D, order=1))
    E = rng.normal(0, 1, (3000, 128)).astype(np.float32)
    assert _eq(c.self_dist(c.bank(E)), oracle.self_dist(E, order=1))
    c.close()


def test_self_dist_batch_of_banks_that_all_differ_in_size(ctx, sweep):
    """r05: the banks of a batched triangular launch need not share a padded size -- every bank sweeps under the plan of
    its own size, in its own block range of the two launches.  Five banks of five sizes: the oracle's values, and (triangular
    sweep) two distance-kernel launches for all of them."""
    rng = np.random.default_rng(19)
    mats = [synth.synth_sift(n, rng) for n in (33000, 35000, 40001, 33500, 36111)]
    mats[2][17] = mats[2][39000]
    banks = [ctx.bank(m) for m in mats]
    ctx.sync()
    ctx.reset_stats()
    got = ctx.self_dist_batch(banks)
    for m, g in zip(mats, got):
        assert _eq(g, oracle.self_dist(m, order=1)), m.shape
    assert all(b.has_selfdist for b in banks)


def test_metric_caches_of_a_dataset_in_batched_launches(ctx):
    """Metric_Cache.from_arrays_many: the caches of a dataset of small images (every image another size) with ALL self
    distances -- originals and thumbnails -- from batched triangular launches: the same caches as from_arrays image by
    image (values, attached banks), in a handful of distance-kernel launches instead of two per image."""
    from fastmatch_amd import cache
    images = []
    for k, n in enumerate((2500, 3100, 1800, 4000, 2999, 3500)):
        q, _ = synth.image_pair((500, 400), n, 300 + k, n_thumb=200 + 10 * k)
        images.append({"descriptors": q["descriptors"], "positions": q["positions"], "size": q["size"],
                       "thumb_descriptors": q["thumb_descriptors"], "thumb_positions": q["thumb_positions"],
                       "thumb_size": q["thumb_size"]})
    one_by_one = [cache.Metric_Cache.from_arrays(options={"context": ctx}, **im) for im in images]
    ctx.sync()
    ctx.reset_stats()
    many = cache.Metric_Cache.from_arrays_many(images, {"context": ctx})
    launches = ctx.stats()["kernel_launches"]
    for a, b, im in zip(one_by_one, many, images):
        assert _eq(a.original["distances"], b.original["distances"]) and _eq(a.thumb["distances"], b.thumb["distances"])
        assert _eq(b.original["distances"], oracle.self_dist(im["descriptors"], order=1))
        assert b.bank(ctx).has_selfdist and b.thumb_bank(ctx).has_selfdist and b.bank(ctx).n == len(im["descriptors"])
    assert launches <= 2, launches          # twelve banks: one group of the triangular sweep (timed as one launch pair)


def test_self_dist_batch_attaches_and_matches(ctx, sweep):
    """Several Metric_Cache builds in one call: banks of one size share a launch, others (another size, float32,
    empty, one row) run beside them; the values are attached on the device (match_ratio uses them) and equal the
    oracle's."""
    rng = np.random.default_rng(9)
    # (33 000-row banks take the 8-wave kernel: the batched launch; the 3 000-row ones the single launch)
    mats = [synth.synth_sift(33000, rng), synth.synth_sift(33000, rng), synth.synth_sift(32900, rng),
            synth.synth_sift(3000, rng), synth.synth_sift(3000, rng),
            (synth.synth_sift(900, rng).astype(np.float32) + 0.25), np.zeros((0, 128), np.uint8),
            synth.synth_sift(1, rng), synth.synth_sift(33000, rng)]
    mats[1][17] = mats[1][30000]
    banks = [ctx.bank(m) for m in mats]
    got = ctx.self_dist_batch(banks)
    for m, b, g in zip(mats, banks, got):
        exp = oracle.self_dist(m, order=1) if m.shape[0] else np.zeros(0)
        assert _eq(g, exp), m.shape
        assert b.has_selfdist
    # the attached values are what the ratio test divides by
    T = synth.synth_sift(2500, rng)
    tb = ctx.bank(T)
    tidx, dist, ratio, passed, npass = ctx.match_ratio(banks[3], tb, 0.9)
    otidx, odist = oracle.bf_xcheck1(mats[3], T)
    m = otidx >= 0
    oratio, opass = oracle.ratio_filter(odist[m], oracle.self_dist(mats[3]), 0.9, qrows=np.nonzero(m)[0])
    assert _eq(tidx, otidx) and _eq(ratio[m], oratio) and np.array_equal(passed[m], opass)
    # enqueue-only form: same values after a sync
    fresh = [ctx.bank(m) for m in mats[:5]]
    assert ctx.self_dist_batch(fresh, want_host=False) is None
    ctx.sync()
    for m, b in zip(mats[:5], fresh):
        tidx2, _, ratio2, _, _ = ctx.match_ratio(b, b, 2.0)      # ratio = d / selfdist, d = 0 on the diagonal
        assert np.all(tidx2 >= -1)
    again = ctx.self_dist_batch(fresh)
    for m, g in zip(mats[:5], again):
        assert _eq(g, oracle.self_dist(m))
    with pytest.raises(_ffi.FastMatchHipError):
        ctx.self_dist_batch([banks[0], banks[0]])
    # launches of at most two banks (option batch_group), float32 banks in the enqueue-only form
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    c.set_option("batch_group", 2)
    five = [c.bank(m) for m in (mats[0], mats[1], mats[8], mats[0], mats[1])] + [c.bank(mats[5])]
    assert c.self_dist_batch(five, want_host=False) is None
    c.sync()
    for m, g in zip((mats[0], mats[1], mats[8], mats[0], mats[1], mats[5]), c.self_dist_batch(five)):
        assert _eq(g, oracle.self_dist(m, order=1))
    tq, _, _, _, _ = c.match_ratio(five[5], five[5], 2.0)          # (the float32 bank's attached values are in use)
    assert tq.shape == (900,)
    c.close()


def test_refill_gives_what_a_fresh_bank_gives(ctx):
    """fm_bank_refill_u8_async + fm_upload_fence + fm_self_dist_batch + the batched match: a pipeline over a
    stream of images re-uses the device arrays; the results are those of banks created from scratch."""
    rng = np.random.default_rng(21)
    n0 = 33000
    Q0, T0, _ = synth.planted_pair(n0, n0, seed=1)
    qb, tb = ctx.bank(Q0), ctx.bank(T0)
    ctx.self_dist_batch([qb], want_host=False)
    ctx.sync()
    for step, (nq, nt) in enumerate([(n0, n0), (32800, 33000), (5000, 7000)]):
        Q, T, _ = synth.planted_pair(nq, nt, seed=50 + step)
        pq = ctx.pinned_empty((nq, 128), np.uint8)
        pt = ctx.pinned_empty((nt, 128), np.uint8)
        pq[:] = Q
        pt[:] = T
        qb.refill_async(pq)
        tb.refill_async(pt)
        ctx.upload_fence()
        ctx.self_dist_batch([qb], want_host=False)
        out = tuple(ctx.pinned_empty(nq, dt) for dt in (np.int32, np.int32, np.float32, np.float64))
        cnt = ctx.pinned_empty(1, np.int64)
        ctx.match_accepted_batch([(qb, tb)], 0.7, [out], [cnt])
        ctx.sync()
        m = int(cnt[0])
        fq, ft = ctx.bank(Q), ctx.bank(T)
        sd = ctx.self_dist(fq)
        assert _eq(sd, oracle.self_dist(Q))
        fq.set_selfdist(sd)
        eq_, et, ed, er = ctx.match_accepted(fq, ft, 0.7)
        assert m == len(eq_) and m > 100
        assert _eq(out[0][:m], eq_) and _eq(out[1][:m], et) and _eq(out[2][:m], ed) and _eq(out[3][:m], er)
    with pytest.raises(_ffi.FastMatchHipError):
        qb.refill_async(ctx.pinned_empty((n0 + 200, 128), np.uint8))      # beyond the bank's first size
    with pytest.raises(ValueError):
        qb.refill_async(np.zeros((10, 64), np.uint8))


_ORDERS_ORACLE = {}       # (what, seed) -> the oracle's answer: the nine parameter sets below share their inputs


def _memo(key, fn):
    if key not in _ORDERS_ORACLE:
        _ORDERS_ORACLE[key] = fn()
    return _ORDERS_ORACLE[key]


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("nsplit", [0, 3, 13])
def test_k1_workgroup_orders_give_identical_results(order, nsplit):
    """Option "k1_order": the three workgroup -> (chunk, split) mappings cover every (chunk, split) exactly once
    (padded grids, chunk counts that are and are not multiples of 8, single and batched launches)."""
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    c.set_option("k1_order", order)
    c.set_option("nsplit", nsplit)
    for nq, nt, seed in ((3000, 2600, 77), (33000, 36000, 78), (700, 40000, 79)):
        Q, T, _ = synth.planted_pair(nq, nt, seed=seed)
        qb, tb = c.bank(Q), c.bank(T)
        tidx, dist = c.xcheck1(qb, tb)
        otidx, odist = _memo(("x1", seed), lambda: oracle.bf_xcheck1(Q, T))
        assert _eq(tidx, otidx) and _eq(dist, odist)
        idx, d2 = c.knn2(qb, tb)
        oidx, od2 = _memo(("knn2", seed), lambda: oracle.bf_knn(Q, T, 2))
        assert _eq(idx, oidx) and _eq(d2, od2)
        assert _eq(c.self_dist(qb), _memo(("sd", seed), lambda: oracle.self_dist(Q)))
    # batched launches: three pairs of one shape + the self distances of three banks in one launch each
    mats = [synth.planted_pair(33000, 33000, seed=90 + i)[:2] for i in range(3)]
    qbs = [c.bank(q) for q, _ in mats]
    tbs = [c.bank(t) for _, t in mats]
    sds = c.self_dist_batch(qbs)
    outs = [tuple(c.pinned_empty(33000, dt) for dt in (np.int32, np.int32, np.float32, np.float64)) for _ in mats]
    cnts = [c.pinned_empty(1, np.int64) for _ in mats]
    c.set_option("batch_tail", 0)
    c.match_accepted_batch(list(zip(qbs, tbs)), 0.7, outs, cnts)
    c.sync()
    for i, ((Q, T), sd, out, cnt) in enumerate(zip(mats, sds, outs, cnts)):
        osd = _memo(("sd", 90 + i), lambda: oracle.self_dist(Q))
        assert _eq(sd, osd)
        otidx, odist = _memo(("x1", 90 + i), lambda: oracle.bf_xcheck1(Q, T))
        mm = otidx >= 0
        oratio, opass = oracle.ratio_filter(odist[mm], osd, 0.7, qrows=np.nonzero(mm)[0])
        rows = np.nonzero(mm)[0][opass]
        m = int(cnt[0])
        assert m == len(rows) and _eq(out[0][:m], rows.astype(np.int32)) and _eq(out[1][:m], otidx[rows])
        assert _eq(out[2][:m], odist[rows]) and _eq(out[3][:m], oratio[opass])
    c.close()


def test_stats_carry_algorithmic_bytes(ctx):
    Q, T, _ = synth.planted_pair(3000, 2600, seed=3)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    ctx.reset_stats()
    ctx.xcheck1(qb, tb)
    s = ctx.stats()
    assert s["pairs"] == 3000 * 2600 and s["bytes_moved"] == (3000 + 2600) * 128
    ctx.self_dist(qb)
    assert ctx.stats()["bytes_moved"] == (3000 + 2600) * 128 + 3000 * 128


def test_self_dist_300k_rows_against_oracle_rows(ctx, sweep):
    """configs[2]'s bank size: 300k rows through the masked top-1 sweep; a row sample against the oracle's 2-NN of
    those rows over the whole bank, and size-independent properties for all rows."""
    rng = np.random.default_rng(300)
    D = synth.synth_sift(300000, rng)
    D[123456] = D[7]
    sd = ctx.self_dist(ctx.bank(D))
    assert sd.shape == (300000,) and sd[123456] == 0 and sd[7] == 0 and np.all(sd >= 0) and np.all(np.isfinite(sd))
    rows = np.concatenate([rng.choice(300000, 300, replace=False), [0, 7, 63, 64, 123456, 299999]])
    _, od = oracle.bf_knn(D[rows], D, 2)
    assert _eq(sd[rows], od[:, 1].astype(np.float64))


@pytest.mark.parametrize("stages", [0, 4, 9, 40])
def test_triangular_sweep_piece_lengths_and_fuzz(ctx, stages):
    """The triangular sweep under several piece lengths ("tri_stages") on random sizes around the chunk (512) and stage
    (128) boundaries, duplicates and zero rows included: bit for bit the oracle's values."""
    old = ctx.get_option("self_tri"), ctx.get_option("tri_stages")
    ctx.set_option("self_tri", 2)
    ctx.set_option("tri_stages", stages)
    try:
        rng = np.random.default_rng(500 + stages)
        for _ in range(14):
            n = int(rng.choice([rng.integers(1, 300), rng.integers(500, 530), rng.integers(1000, 1040), rng.integers(2040, 2600),
                                rng.integers(5000, 9000)]))
            D = synth.synth_sift(n, rng)
            if n > 40:
                D[rng.integers(0, n)] = D[rng.integers(0, n)]
                D[rng.integers(0, n)] = 0
            assert _eq(ctx.self_dist(ctx.bank(D)), oracle.self_dist(D)), (n, stages)
    finally:
        ctx.set_option("self_tri", old[0])
        ctx.set_option("tri_stages", old[1])


def test_default_takes_the_triangular_sweep_from_32768_rows(ctx):
    """The default ("self_tri" 1): 70 001 rows through the triangular sweep == the masked full sweep == the oracle on a
    row sample; each distance is computed once, the statistics still count the n x n pairs the caller asked for."""
    assert ctx.get_option("self_tri") == 1
    rng = np.random.default_rng(70001)
    D = synth.synth_sift(70001, rng)
    D[69999] = D[12]
    b = ctx.bank(D)
    ctx.reset_stats()
    tri = ctx.self_dist(b)
    assert ctx.stats()["pairs"] == 70001 * 70001
    ctx.set_option("self_tri", 0)
    try:
        full = ctx.self_dist(b)
    finally:
        ctx.set_option("self_tri", 1)
    assert _eq(tri, full) and tri[69999] == 0 and tri[12] == 0
    rows = np.concatenate([rng.choice(70001, 200, replace=False), [0, 511, 512, 69999, 70000]])
    _, od = oracle.bf_knn(D[rows], D, 2)
    assert _eq(tri[rows], od[:, 1].astype(np.float64))
