"""GPU: the HIP path (through the C-ABI) against the oracle -- bit-exact indices,
distances (float32 bits) and ratio pass/fail on the same seeded inputs -- plus
size-independent properties at BASELINE.json's full 100k x 100k size."""
import numpy as np
import pytest

import oracle
from fastmatch_amd import synth, matchutil, sharding, _ffi
from kat import (xcheck_cases, knn2_cases, selfdist_case, sqrt_tie_knn2_cases, sqrt_tie_xcheck_cases,
                 SQRT_TIE_MIN, row_with_sumsq, far_banks)

pytestmark = pytest.mark.gpu


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("case", xcheck_cases() + sqrt_tie_xcheck_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("as_f32", [False, True])
def test_xcheck_kat(ctx, case, as_f32):
    _, Q, T, etidx, edist = case
    if as_f32:
        Q, T = Q.astype(np.float32), T.astype(np.float32)
    tidx, dist = ctx.xcheck1(ctx.bank(Q), ctx.bank(T))
    assert tidx.tolist() == etidx
    assert _eq(dist, np.array(edist, dtype=np.float32))


@pytest.mark.parametrize("case", knn2_cases() + sqrt_tie_knn2_cases(), ids=lambda c: c[0])
def test_knn2_kat(ctx, case):
    _, Q, T, eidx, edist = case
    idx, dist = ctx.knn2(ctx.bank(Q), ctx.bank(T))
    assert idx.tolist() == eidx
    assert _eq(dist, np.array(edist, dtype=np.float32))


def test_selfdist_kat(ctx):
    D, exp = selfdist_case()
    assert ctx.self_dist(ctx.bank(D)).tolist() == exp


def test_empty_inputs(ctx):
    e = ctx.bank(np.zeros((0, 128), np.uint8))
    t = ctx.bank(np.ones((5, 128), np.uint8))
    tidx, dist = ctx.xcheck1(e, t)
    assert tidx.shape == (0,) and dist.shape == (0,)
    idx, d = ctx.knn2(e, t)
    assert idx.shape == (0, 2)
    tidx, dist = ctx.xcheck1(t, e)                     # empty train: all unmatched
    assert tidx.tolist() == [-1] * 5 and np.all(np.isinf(dist))
    idx, d = ctx.knn2(t, e)
    assert idx.tolist() == [[-1, -1]] * 5 and np.all(np.isinf(d))
    assert ctx.self_dist(e).shape == (0,)


SIZES = [(1, 1), (2, 1), (1, 2), (31, 33), (32, 32), (33, 31), (127, 129), (128, 128), (129, 127),
         (393, 125), (125, 393), (1000, 1), (1, 1000), (640, 513), (2049, 1027), (5000, 4096)]


@pytest.mark.parametrize("nq,nt", SIZES)
def test_dense_parity_u8(ctx, nq, nt):
    Q, T, _ = synth.planted_pair(nq, nt, seed=1000 + 7 * nq + nt)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    tidx, dist = ctx.xcheck1(qb, tb)
    otidx, odist = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(dist, odist)
    idx, d2 = ctx.knn2(qb, tb)
    oidx, od2 = oracle.bf_knn(Q, T, 2)
    assert _eq(idx, oidx) and _eq(d2, od2)
    assert _eq(ctx.self_dist(qb), oracle.self_dist(Q))


def test_dense_parity_integer_valued_f32_routes_to_int8(ctx):
    Q, T, _ = synth.planted_pair(700, 900, seed=5)
    qb, tb = ctx.bank(Q.astype(np.float32)), ctx.bank(T.astype(np.float32))
    assert qb.kind == _ffi.FM_BANK_I8 and tb.kind == _ffi.FM_BANK_I8
    tidx, dist = ctx.xcheck1(qb, tb)
    otidx, odist = oracle.bf_xcheck1(Q.astype(np.float32), T.astype(np.float32))
    assert _eq(tidx, otidx) and _eq(dist, odist)


def test_many_ties_and_duplicates(ctx):
    # low-entropy descriptors: masses of exact distance ties and duplicate rows exercise
    # every lowest-index tie-break (in-lane order, lane halves, waves, splits)
    rng = np.random.default_rng(42)
    Q = rng.integers(0, 2, (3000, 128), dtype=np.uint8) * 255
    T = rng.integers(0, 2, (2500, 128), dtype=np.uint8) * 255
    Q[:, 8:] = 0
    T[:, 8:] = 0                                         # only 256 distinct rows
    qb, tb = ctx.bank(Q), ctx.bank(T)
    tidx, dist = ctx.xcheck1(qb, tb)
    otidx, odist = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(dist, odist)
    idx, d2 = ctx.knn2(qb, tb)
    oidx, od2 = oracle.bf_knn(Q, T, 2)
    assert _eq(idx, oidx) and _eq(d2, od2)


def test_extreme_values_and_short_dim(ctx):
    rng = np.random.default_rng(8)
    Q = rng.choice(np.array([0, 255], dtype=np.uint8), (300, 128))
    T = rng.choice(np.array([0, 255], dtype=np.uint8), (200, 128))
    tidx, dist = ctx.xcheck1(ctx.bank(Q), ctx.bank(T))
    otidx, odist = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(dist, odist)
    # dim < 128 (zero padded on the device)
    Q64 = rng.integers(0, 256, (150, 64), dtype=np.uint8)
    T64 = rng.integers(0, 256, (170, 64), dtype=np.uint8)
    idx, d = ctx.knn2(ctx.bank(Q64), ctx.bank(T64))
    oidx, od = oracle.bf_knn(Q64, T64, 2)
    assert _eq(idx, oidx) and _eq(d, od)


# ---- float32 square-root ties (d2 >= 4 197 200: OpenCV orders by the float32 root, then index) -----
@pytest.mark.parametrize("nq,nt,small", [(300, 200, (10, 1)), (393, 125, (6, 2)), (2500, 3100, (10, 1)), (129, 4000, (8, 1))])
def test_sqrt_tie_range_dense_parity(ctx, nq, nt, small):
    rng = np.random.default_rng(nq + nt)
    Q, T = far_banks(nq, nt, rng, *small)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    idx, d2 = ctx.knn2(qb, tb)
    oidx, od2 = oracle.bf_knn(Q, T, 2)
    assert _eq(idx, oidx) and _eq(d2, od2)
    tidx, dist = ctx.xcheck1(qb, tb)
    otidx, odist = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(dist, odist)
    assert _eq(ctx.self_dist(tb), oracle.self_dist(T))            # (train rows are 0 apart in the far dims: no ties here)
    # the same banks as integer-valued float32 input (int8 route) and on the float32 route
    qf, tf = ctx.bank(Q.astype(np.float32)), ctx.bank(T.astype(np.float32))
    assert qf.kind == _ffi.FM_BANK_I8
    t2, x2 = ctx.xcheck1(qf, tf)
    assert _eq(t2, otidx) and _eq(x2, odist)
    qr, tr = ctx.bank(Q, float_route=True), ctx.bank(T, float_route=True)
    t3, x3 = ctx.xcheck1(qr, tr)
    i3, k3 = ctx.knn2(qr, tr)
    assert _eq(t3, otidx) and _eq(x3, odist) and _eq(i3, oidx) and _eq(k3, od2)


@pytest.mark.parametrize("nsplit", [1, 2, 5, 13])
def test_sqrt_tie_across_split_boundaries(nsplit):
    """The two members of a tie group sit in different splits of the reduction range (different
    workgroups, merged by the election / the top-2 merge): filler rows are farther than both."""
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    c.set_option("nsplit", nsplit)
    assert c.get_option("nsplit") == nsplit
    n = SQRT_TIE_MIN + 2122                                       # 4 199 322 / 4 199 323 share a root too
    assert np.sqrt(np.float32(n)) == np.sqrt(np.float32(n + 1))
    lo, hi = row_with_sumsq(n), row_with_sumsq(n + 1)
    filler = np.zeros(128, np.uint8)
    filler[:66] = 255                                             # d2 = 4 291 650 from the zero row
    rows = 13 * 1024 + 300
    T = np.tile(filler, (rows, 1))
    for a, b in ((100, rows - 50), (rows // 2 - 1, rows // 2 + 700), (5, 6)):
        T2 = T.copy()
        T2[a], T2[b] = hi, lo                                     # the LARGER d2 has the lower index
        z = np.zeros((3, 128), np.uint8)
        z[1, 127] = 200                                           # other queries, far from everything
        z[2, 126] = 201
        qb, tb = c.bank(z), c.bank(T2)
        idx, d = c.knn2(qb, tb)
        oidx, od = oracle.bf_knn(z, T2, 2)
        assert _eq(idx, oidx) and _eq(d, od) and idx[0].tolist() == [a, b]
        tidx, xd = c.xcheck1(qb, tb)
        otidx, oxd = oracle.bf_xcheck1(z, T2)
        assert _eq(tidx, otidx) and _eq(xd, oxd)
        # the election side: many query rows, one train row (the zero row): the lower query index wins
        tz = c.bank(np.zeros((1, 128), np.uint8))
        qq = c.bank(T2)
        tidx, xd = c.xcheck1(qq, tz)
        otidx, oxd = oracle.bf_xcheck1(T2, np.zeros((1, 128), np.uint8))
        assert _eq(tidx, otidx) and _eq(xd, oxd) and tidx[a] == 0 and tidx[b] == -1
    c.close()


def test_sqrt_tie_rounds_and_keys(ctx):
    """K4 rounds (fm_xcheck1_batched) and the sharded election keys on tie-range banks."""
    rng = np.random.default_rng(99)
    Q, T = far_banks(3000, 2000, rng)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    sizes = [(393, 125), (1, 1), (130, 257), (1000, 33), (2048, 100), (37, 300)]
    q_rows, q_off, t_off = [], [0], [0]
    for nq, nt in sizes:
        q_rows.append(rng.choice(3000, nq, replace=False))
        q_off.append(q_off[-1] + nq)
        t_off.append(t_off[-1] + nt)
    q_rows = np.concatenate(q_rows).astype(np.int32)
    tidx, dist, _ = ctx.xcheck1_batched(qb, q_rows, q_off, tb, t_off)
    differs = 0
    for b, (nq, nt) in enumerate(sizes):
        rows = q_rows[q_off[b]:q_off[b + 1]]
        ot, od = oracle.bf_xcheck1(Q[rows], T[t_off[b]:t_off[b + 1]])
        sl = slice(q_off[b], q_off[b + 1])
        assert _eq(tidx[sl], ot) and _eq(dist[sl], od), "round %d" % b
        d2 = ((Q[rows][:, None].astype(np.int64) - T[t_off[b]:t_off[b + 1]][None].astype(np.int64)) ** 2).sum(-1)
        differs += int((np.argmin(d2, axis=0) != np.argmin(np.sqrt(d2.astype(np.float32)), axis=0)).sum())
    assert differs > 10                                           # (the rounds do contain elections the root decides)
    full_t, full_d = ctx.xcheck1(qb, tb)
    ot, od = oracle.bf_xcheck1(Q, T)
    assert _eq(full_t, ot) and _eq(full_d, od)
    for world in (2, 3):
        keys = None
        for r in range(world):
            lo, hi = sharding.shard_rows(T.shape[0], r, world)
            k = ctx.xcheck1_keys(qb, ctx.bank(T[lo:hi]), lo)
            keys = k if keys is None else np.minimum(keys, k)
        t, d = sharding.decode_keys(keys)
        assert _eq(t, ot) and _eq(d, od)


def test_sqrt_tie_accepted_paths(ctx):
    """fm_match_accepted (sync), _async and _batch on tie-range banks == oracle."""
    rng = np.random.default_rng(5)
    pairs, exp = [], []
    for k in range(3):
        Q, T = far_banks(2000, 2300, rng)
        Q[:, 111:128] = rng.integers(0, 2, (2000, 17), dtype=np.uint8)     # distinct query rows: non-zero self distances
        qb, tb = ctx.bank(Q), ctx.bank(T)
        sd = oracle.self_dist(Q)
        assert _eq(ctx.self_dist(qb), sd)
        qb.set_selfdist(sd)
        ot, od = oracle.bf_xcheck1(Q, T)
        m = ot >= 0
        tau = float(np.median(od[m] / sd[m]))
        orat, opass = oracle.ratio_filter(od[m], sd, tau, qrows=np.nonzero(m)[0].astype(np.int32))
        acc = np.nonzero(m)[0][opass]
        pairs.append((qb, tb, tau))
        exp.append((acc.astype(np.int32), ot[acc], od[acc], orat[opass]))
        got = ctx.match_accepted(qb, tb, tau)
        assert all(_eq(a, b) for a, b in zip(got, exp[-1]))
    outs = [tuple(ctx.pinned_empty(2000, dt) for dt in (np.int32, np.int32, np.float32, np.float64)) for _ in pairs]
    cnts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    for (qb, tb, tau), o, c in zip(pairs, outs, cnts):
        ctx.match_accepted_async(qb, tb, tau, o, c)
    ctx.sync()
    for e, o, c in zip(exp, outs, cnts):
        m = int(c[0])
        assert m == len(e[0]) and all(_eq(a[:m], b) for a, b in zip(o, e))
    # one batched launch (same shapes, same tau for all: use pair 0's threshold)
    tau0 = pairs[0][2]
    ctx.match_accepted_batch([(q, t) for q, t, _ in pairs], tau0, outs, cnts)
    ctx.sync()
    for (qb, tb, _), o, c in zip(pairs, outs, cnts):
        e = ctx.match_accepted(qb, tb, tau0)
        m = int(c[0])
        assert m == len(e[0]) and all(_eq(a[:m], b) for a, b in zip(o, e))


def test_ctx_options_api(ctx):
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    assert c.get_option("batch_group") == 8 and c.get_option("batch_tail") == 2
    c.set_option("batch_group", 16)
    c.set_option("batch_tail", 0)
    assert c.get_option("batch_group") == 16 and c.get_option("batch_tail") == 0
    assert ctx.get_option("batch_group") == 8                     # per context, not per process
    for bad in (("batch_group", 0), ("batch_group", 17), ("nbuf", 5), ("no_such_option", 1)):
        with pytest.raises(_ffi.FastMatchHipError):
            c.set_option(*bad)
    with pytest.raises(_ffi.FastMatchHipError):
        c.get_option("no_such_option")
    c.close()


@pytest.mark.parametrize("nsplit,nb", [(1, 4), (1, 8), (3, 4), (8, 8), (16, 4)])
def test_split_and_tile_shapes_give_identical_results(nsplit, nb, monkeypatch):
    import fastmatch_amd
    monkeypatch.setenv("FM_NSPLIT", str(nsplit))
    monkeypatch.setenv("FM_NB", str(nb))
    c = fastmatch_amd.Context(0)
    Q, T, _ = synth.planted_pair(3000, 2600, seed=77)
    qb, tb = c.bank(Q), c.bank(T)
    tidx, dist = c.xcheck1(qb, tb)
    otidx, odist = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(dist, odist)
    idx, d2 = c.knn2(qb, tb)
    oidx, od2 = oracle.bf_knn(Q, T, 2)
    assert _eq(idx, oidx) and _eq(d2, od2)
    c.close()


def test_ratio_filter_and_fused_match_ratio(ctx):
    Q, T, _ = synth.planted_pair(2000, 2400, seed=13)
    Q[5] = Q[6]                                           # duplicate query rows: selfdist 0
    qb, tb = ctx.bank(Q), ctx.bank(T)
    sd = ctx.self_dist(qb)
    osd = oracle.self_dist(Q)
    assert _eq(sd, osd) and sd[5] == 0.0 and sd[6] == 0.0
    qb.set_selfdist(sd)
    tidx, dist, ratio, passed, npass = ctx.match_ratio(qb, tb, 0.7)
    otidx, odist = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(dist, odist)
    m = otidx >= 0
    oratio, opass = oracle.ratio_filter(odist[m], osd, 0.7, qrows=np.nonzero(m)[0])
    assert _eq(ratio[m], oratio) and np.array_equal(passed[m], opass)
    assert np.all(np.isnan(ratio[~m])) and not passed[~m].any()
    assert npass == int(opass.sum()) and npass > 100
    # stand-alone R1 on host arrays, with and without row indirection; x/0 and 0/0 rejected
    d = np.array([3.0, 0.0, 2.0, 7.0], dtype=np.float32)
    s = np.array([4.0, 0.0, 0.0, 10.0])
    r, p, n = ctx.ratio_filter(d, s, 0.75)
    orr, opp = oracle.ratio_filter(d, s, 0.75)
    assert _eq(r, orr) and np.array_equal(p, opp) and n == 1
    rows = np.nonzero(m)[0].astype(np.int32)
    r, p, n = ctx.ratio_filter(odist[m], osd, 0.7, qrows=rows)
    assert _eq(r, oratio) and np.array_equal(p, opass) and n == int(opass.sum())


def test_batched_rounds_match_oracle_per_round(ctx):
    rng = np.random.default_rng(21)
    Q, T, _ = synth.planted_pair(6000, 5000, seed=31)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    sd = oracle.self_dist(Q)
    qb.set_selfdist(sd)
    q_rows, q_off, t_off_pairs = [], [0], []
    sizes = [(393, 125), (1, 1), (0, 40), (50, 0), (130, 257), (1000, 33), (4096, 100), (37, 300)]
    t_off = [0]
    for nq, nt in sizes:
        q_rows.append(rng.choice(6000, nq, replace=False))
        q_off.append(q_off[-1] + nq)
        t_off.append(t_off[-1] + nt)                     # cells packed back to back
    q_rows = np.concatenate(q_rows).astype(np.int32)
    tidx, dist, ratio = ctx.xcheck1_batched(qb, q_rows, q_off, tb, t_off)
    for b, (nq, nt) in enumerate(sizes):
        rows = q_rows[q_off[b]:q_off[b + 1]]
        ot, od = oracle.bf_xcheck1(Q[rows], T[t_off[b]:t_off[b + 1]])
        sl = slice(q_off[b], q_off[b + 1])
        assert _eq(tidx[sl], ot) and _eq(dist[sl], od), "round %d" % b
        m = ot >= 0
        orat, _ = oracle.ratio_filter(od[m], sd, 0.7, qrows=rows[m])
        assert _eq(ratio[sl][m], orat) and np.all(np.isnan(ratio[sl][~m]))
    # overlapping / repeated rounds on the same cell are independent
    tidx2, dist2, _ = ctx.xcheck1_batched(qb, np.concatenate([q_rows[:393], q_rows[:393]]), [0, 393, 786],
                                          tb, [0, 125, 125 + 0], )
    assert _eq(tidx2[:393], tidx[:393]) and np.all(tidx2[393:] == -1)


@pytest.mark.parametrize("kind", ["noisy_sift", "rootsift", "duplicates", "tiny_scale"])
def test_batched_rounds_float32_route_match_oracle_per_round(ctx, kind):
    """fm_xcheck1_batched on banks that are not integer valued: the float32 round (fp16 MFMA
    filter + exact float32 chain) must equal the oracle's order-1 chain bit for bit, per round."""
    rng = np.random.default_rng(77)
    base_q = synth.synth_sift(5000, rng).astype(np.float32)
    base_t = synth.synth_sift(4000, rng).astype(np.float32)
    base_q[:1500] = base_t[:1500] + rng.normal(0, 4, (1500, 128)).astype(np.float32)     # planted near pairs
    if kind == "noisy_sift":
        Q = base_q + rng.uniform(-0.5, 0.5, base_q.shape).astype(np.float32)
        T = base_t + rng.uniform(-0.5, 0.5, base_t.shape).astype(np.float32)
    elif kind == "rootsift":
        f = lambda d: np.sqrt(np.abs(d) / np.maximum(np.abs(d).sum(1, keepdims=True), 1)).astype(np.float32)
        Q, T = f(base_q), f(base_t)
    elif kind == "duplicates":                       # ties: equal rows on both sides -> lowest index wins
        Q = (base_q + 0.25).astype(np.float32)
        T = (base_t + 0.25).astype(np.float32)
        Q[100:140] = Q[100]
        Q[2000:2100] = T[7]
        T[50:60] = T[50]
        T[300:340] = Q[3]
    else:
        Q = ((base_q + 0.25) * 1e-4).astype(np.float32)
        T = ((base_t + 0.25) * 1e-4).astype(np.float32)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    assert qb.kind == _ffi.FM_BANK_F32 and tb.kind == _ffi.FM_BANK_F32
    sd = oracle.self_dist(Q, order=1)
    qb.set_selfdist(sd)
    sizes = [(393, 125), (1, 1), (0, 40), (50, 0), (130, 257), (1000, 33), (4096, 100), (37, 300), (256, 128), (257, 129)]
    q_rows, q_off, t_off = [], [0], [0]
    for nq, nt in sizes:
        rows = rng.choice(5000, nq, replace=False)
        if kind == "duplicates" and nq >= 130:
            rows[:60] = np.arange(100, 160)          # duplicate query rows inside one round
            rows[60:100] = np.arange(2000, 2040)
        q_rows.append(rows)
        q_off.append(q_off[-1] + nq)
        t_off.append(t_off[-1] + nt)
    q_rows = np.concatenate(q_rows).astype(np.int32)
    tidx, dist, ratio = ctx.xcheck1_batched(qb, q_rows, q_off, tb, t_off)
    for b, (nq, nt) in enumerate(sizes):
        rows = q_rows[q_off[b]:q_off[b + 1]]
        ot, od = oracle.bf_xcheck1(Q[rows], T[t_off[b]:t_off[b + 1]], order=1)
        sl = slice(q_off[b], q_off[b + 1])
        assert _eq(tidx[sl], ot) and _eq(dist[sl], od), "round %d" % b
        m = ot >= 0
        orat, _ = oracle.ratio_filter(od[m], sd, 0.7, qrows=rows[m])
        assert _eq(ratio[sl][m], orat) and np.all(np.isnan(ratio[sl][~m]))
    # and the dense float32 route gives the same answers on the gathered sub-matrices
    rows = q_rows[:393]
    sub = ctx.bank(Q[rows], float_route=True)
    cell = ctx.bank(T[:125], float_route=True)
    dt, dd = ctx.xcheck1(sub, cell)
    assert _eq(dt, tidx[:393]) and _eq(dd, dist[:393])


def test_matchutil_surface(ctx):
    Q, T, _ = synth.planted_pair(300, 280, seed=3)
    opts = {"context": ctx}
    m = matchutil.bf_match(Q, T, k=1, options={"crossCheck": True, "context": ctx})
    otidx, odist = oracle.bf_xcheck1(Q, T)
    assert len(m) == 300
    for qi, row in enumerate(m):
        if otidx[qi] < 0:
            assert row == []
        else:
            assert len(row) == 1 and row[0].queryIdx == qi and row[0].trainIdx == otidx[qi]
            assert row[0].distance == float(odist[qi]) and row[0].imgIdx == 0
    # crossCheck is ignored unless k == 1 (matchutil.py:41)
    m2 = matchutil.bf_match(Q, T, k=2, options={"crossCheck": True, "context": ctx})
    oidx, od = oracle.bf_knn(Q, T, 2)
    assert [[d.trainIdx for d in r] for r in m2] == oidx.tolist()
    assert [[d.distance for d in r] for r in m2] == od.astype(np.float64).tolist()
    m1 = matchutil.bf_match(Q, T, k=1, options=opts)
    assert [r[0].trainIdx for r in m1] == oidx[:, 0].tolist()
    mf = matchutil.flann_match(Q, Q, k=2, options=opts)          # exact substitute for FLANN
    assert [r[1].distance for r in mf] == oracle.self_dist(Q).tolist()
    with pytest.raises(ValueError):
        matchutil.bf_match(Q, T, k=9, options=opts)            # (r06: k up to 8 is served, test_bf_match_for_any_k_up_to_8)
    with pytest.raises(_ffi.FastMatchHipError):                   # width mismatch (cv2.error there)
        matchutil.bf_match(Q, T[:, :64], k=2, options=opts)


def test_matchutil_mixed_integer_and_non_integer_float32(ctx):
    """cv2.BFMatcher accepts any two float32 arrays; one of them being integer valued (here a
    one-row all-zero bank and a bank of rounded values) must not fail on the bank-kind pairing:
    the pair runs on the float32 route and equals the oracle's float32 chain."""
    rng = np.random.default_rng(12)
    F = (synth.synth_sift(200, rng).astype(np.float32) / 512.0).astype(np.float32)      # non-integer
    I = synth.synth_sift(150, rng).astype(np.float32)                                    # integer valued
    Z = np.zeros((1, 128), dtype=np.float32)
    opts = {"context": ctx}
    for A, B in ((Z, F), (F, Z), (I, F), (F, I)):
        for k in (1, 2):
            idx, dist = matchutil.bf_match_arrays(A, B, k=k, options=opts)
            oidx, odist = oracle.bf_knn(A, B, k, order=1)
            assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        tidx, d = matchutil.bf_match_arrays(A, B, k=1, options={"crossCheck": True, "context": ctx})
        otidx, od = oracle.bf_xcheck1(A, B, order=1)
        assert np.array_equal(tidx, otidx) and np.array_equal(d, od)
        q, t, dd, r = matchutil.ratio_match_arrays(A, B, 0.9, options=opts)
        assert len(q) == len(t) == len(dd) == len(r)
    # integer valued on both sides still takes the exact int8 route and agrees with it
    idx, dist = matchutil.bf_match_arrays(I, I[:70], k=2, options=opts)
    oidx, odist = oracle.bf_knn(I, I[:70], 2)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)


def test_deterministic_replay(ctx):
    Q, T, _ = synth.planted_pair(4000, 4000, seed=99)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    a = ctx.xcheck1(qb, tb)
    b = ctx.xcheck1(qb, tb)
    assert _eq(a[0], b[0]) and _eq(a[1], b[1])


@pytest.fixture(scope="module")
def full_size(ctx):
    Q, T, planted = synth.planted_pair(100000, 100000, seed=20250002)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    return Q, T, planted, qb, tb


def test_full_size_properties_100k(ctx, full_size):
    """BASELINE.json config 2 (100k x 100k): the cross-check against the oracle IN FULL (all 10^10 pairs:
    seconds on the GPU box's host cores), the 2-NN lists against the oracle on a row sample (rows are
    independent), plus size-independent properties."""
    Q, T, planted, qb, tb = full_size
    idx, dist = ctx.knn2(qb, tb)
    tidx, xd = ctx.xcheck1(qb, tb)
    # (a) sortedness and validity of the 2-NN lists
    assert idx.min() >= 0 and idx.max() < 100000
    assert np.all(dist[:, 0] <= dist[:, 1]) and np.all(idx[:, 0] != idx[:, 1])
    tie = dist[:, 0] == dist[:, 1]
    assert np.all(idx[tie, 0] < idx[tie, 1])
    # (b) reported distances are the true distances of the reported pairs (exact int math)
    rows = np.random.default_rng(0).choice(100000, 4000, replace=False)
    for col in (0, 1):
        d2 = ((Q[rows].astype(np.int64) - T[idx[rows, col]].astype(np.int64)) ** 2).sum(1)
        assert _eq(dist[rows, col], np.sqrt(d2.astype(np.float32)))
    # (c) oracle agreement on a row sample of K2 (each row is independent of the others)
    srows = rows[:2000]
    oidx, od = oracle.bf_knn(Q[srows], T, 2)
    assert _eq(idx[srows], oidx) and _eq(dist[srows], od)
    # (c') the whole cross-check (election over all query rows, scatter-min over all train rows)
    otidx, oxd = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(xd, oxd)
    # (d) cross-check structure: matched train rows are distinct, each match is the
    #     reverse nearest neighbour (t elects q), and a cross-checked match can never be
    #     closer than q's own nearest train row
    mt = tidx[tidx >= 0]
    assert len(np.unique(mt)) == len(mt)
    mq = np.nonzero(tidx >= 0)[0]
    assert np.all(xd[mq] >= dist[mq, 0])
    # transposed problem: reverse NN of T over Q from the K2 kernel with roles swapped
    ridx, rdist = ctx.knn2(tb, qb)
    assert np.array_equal(ridx[mt, 0], mq) and _eq(rdist[mt, 0], xd[mq])
    # scatter-min replay on the host from the reverse-NN table == device result
    exp = np.full(100000, -1, dtype=np.int32)
    best = np.full(100000, np.inf, dtype=np.float32)
    order = np.lexsort((np.arange(100000), rdist[:, 0]))          # ascending (dist, t)
    q_of = ridx[order, 0]
    first = np.unique(q_of, return_index=True)[1]
    exp[q_of[first]] = order[first]
    best[q_of[first]] = rdist[order[first], 0]
    assert _eq(tidx, exp) and _eq(xd, best)
    # (e) planted pairs are overwhelmingly recovered and survive the ratio test at 0.7
    sd = ctx.self_dist(qb)
    qb.set_selfdist(sd)
    t2, d2_, ratio, passed, npass = ctx.match_ratio(qb, tb, 0.7)
    assert _eq(t2, tidx) and _eq(d2_, xd)
    pl = planted >= 0
    assert (tidx[pl] == planted[pl]).mean() > 0.99
    assert npass == int(passed.sum()) and passed[pl].mean() > 0.9 and passed[~pl].mean() < 0.01
    # (f) idempotence / determinism
    t3, d3 = ctx.xcheck1(qb, tb)
    assert _eq(t3, tidx) and _eq(d3, xd)


# ---- K5: float32 route (non-integer descriptors) -------------------------------------------
def _nonint(n, seed):
    rng = np.random.default_rng(seed)
    base = synth.synth_sift(n, rng).astype(np.float32)
    return (base + rng.uniform(-0.5, 0.5, base.shape).astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("nq,nt", [(1, 1), (3, 2), (63, 65), (64, 64), (130, 257), (1000, 777), (3000, 4100)])
def test_f32_route_parity(ctx, nq, nt):
    Q, T = _nonint(nq, 100 + nq), _nonint(nt, 200 + nt)
    if nq > 10 and nt > 10:
        T[7] = T[3]                                       # duplicate train rows: index tie-break
        Q[5] = T[3]                                       # exact zero distance
    qb, tb = ctx.bank(Q), ctx.bank(T)
    assert qb.kind == _ffi.FM_BANK_F32 and tb.kind == _ffi.FM_BANK_F32
    idx, dist = ctx.knn2(qb, tb)
    oidx, odist = oracle.bf_knn(Q, T, 2, order=1)          # the fixed fma-chain order
    assert _eq(idx, oidx) and _eq(dist, odist)             # bit-exact (tolerance 0 ulp <= 1 ulp)
    tidx, xd = ctx.xcheck1(qb, tb)
    otidx, oxd = oracle.bf_xcheck1(Q, T, order=1)
    assert _eq(tidx, otidx) and _eq(xd, oxd)
    sd = ctx.self_dist(qb)
    assert _eq(sd, oracle.self_dist(Q, order=1))
    if nq > 10:
        qb.set_selfdist(sd)
        t2, d2, ratio, passed, npass = ctx.match_ratio(qb, tb, 0.7)
        m = otidx >= 0
        orat, opass = oracle.ratio_filter(oxd[m], sd, 0.7, qrows=np.nonzero(m)[0])
        assert _eq(t2, otidx) and _eq(ratio[m], orat) and np.array_equal(passed[m], opass)


# ---- K8: bf16x3 MFMA filter + exact rescoring, forced on for every size --------------------
def _f32_kind(kind, nq, nt, seed):
    rng = np.random.default_rng(seed)
    if kind == "rootsift":                                 # unit-norm rows, planted near twins
        Q = synth.synth_sift(nq, rng).astype(np.float32)
        T = synth.synth_sift(nt, rng).astype(np.float32)
        m = min(nq, nt) // 2
        T[:m] = np.abs(Q[:m] + rng.normal(0, 6, size=(m, 128)).astype(np.float32))
        Q = np.sqrt(Q / np.maximum(Q.sum(1, keepdims=True), 1)).astype(np.float32)
        T = np.sqrt(T / np.maximum(T.sum(1, keepdims=True), 1)).astype(np.float32)
    elif kind == "gauss":
        Q = rng.normal(0, 1, size=(nq, 128)).astype(np.float32)
        T = rng.normal(0, 1, size=(nt, 128)).astype(np.float32)
    elif kind == "offset":                                 # huge norms, tiny distances: everything is inside the margin
        Q = (1000 + rng.normal(0, 1, size=(nq, 128))).astype(np.float32)
        T = (1000 + rng.normal(0, 1, size=(nt, 128))).astype(np.float32)
    elif kind == "neartie":                                # a few rows per query within a hair of each other
        T = rng.normal(0, 1, size=(nt, 128)).astype(np.float32)
        Q = rng.normal(0, 1, size=(nq, 128)).astype(np.float32)
        for i in range(0, min(nq, 40)):
            for k in range(6):                             # six train rows at almost the same distance from Q[i]
                T[(37 * i + 4 * k) % nt] = Q[i] + np.float32(0.05) * np.roll(np.eye(128, dtype=np.float32)[0], k)
    elif kind == "scaled":
        Q = (rng.normal(0, 1, size=(nq, 128)) * 1e-4).astype(np.float32)
        T = (rng.normal(0, 1, size=(nt, 128)) * 1e-4).astype(np.float32)
    else:                                                  # "dup": eight distinct rows, heavy exact and near ties
        base = rng.normal(0, 1, size=(8, 128)).astype(np.float32)
        Q = (base[rng.integers(0, 8, nq)] + rng.normal(0, 1e-4, size=(nq, 128))).astype(np.float32)
        T = (base[rng.integers(0, 8, nt)] + rng.normal(0, 1e-4, size=(nt, 128))).astype(np.float32)
    return Q, T


@pytest.fixture
def filter_ctx(monkeypatch):
    import fastmatch_amd
    monkeypatch.setenv("FM_F32_FILTER", "2")               # filter every float32 call, whatever its size
    c = fastmatch_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("kind,nq,nt", [
    ("gauss", 1, 1), ("gauss", 3, 2), ("gauss", 63, 65), ("gauss", 700, 450), ("gauss", 5000, 9000),
    ("rootsift", 130, 257), ("rootsift", 3000, 4100), ("rootsift", 9000, 20000),
    ("neartie", 600, 3000), ("offset", 500, 800), ("scaled", 400, 1500), ("dup", 900, 1300)])
def test_f32_filter_route_parity(filter_ctx, kind, nq, nt):
    """K8 gives the bits K5 and the oracle give, whichever of its three endings a call takes
    (rescoring alone, per-row rescan, whole-call redo by K5)."""
    c = filter_ctx
    Q, T = _f32_kind(kind, nq, nt, 1000 + nq + nt)
    qb, tb = c.bank(Q), c.bank(T)
    assert qb.kind == _ffi.FM_BANK_F32
    idx, dist = c.knn2(qb, tb)
    oidx, odist = oracle.bf_knn(Q, T, 2, order=1)
    assert _eq(idx, oidx) and _eq(dist, odist)
    tidx, xd = c.xcheck1(qb, tb)
    otidx, oxd = oracle.bf_xcheck1(Q, T, order=1)
    assert _eq(tidx, otidx) and _eq(xd, oxd)
    sd = c.self_dist(qb)
    assert _eq(sd, oracle.self_dist(Q, order=1))
    launches, redone = c.f32_filter_stats()
    assert launches >= 3
    if kind in ("gauss", "rootsift"):
        assert redone == 0                                 # well separated data never needs K5
    if kind in ("dup", "offset") and nq * nt > 100000:
        assert redone > 0                                  # everything inside the margin: K5 redoes the call


def test_f32_filter_handles_dim_below_128_and_empty_banks(filter_ctx):
    c = filter_ctx
    rng = np.random.default_rng(5)
    Q, T = rng.normal(0, 1, (300, 64)).astype(np.float32), rng.normal(0, 1, (500, 64)).astype(np.float32)
    idx, dist = c.knn2(c.bank(Q), c.bank(T))
    oidx, odist = oracle.bf_knn(Q, T, 2, order=1)
    assert _eq(idx, oidx) and _eq(dist, odist)
    E = c.bank(np.zeros((0, 64), np.float32))               # an empty bank pairs with either kind
    tidx, d = c.xcheck1(c.bank(Q), E)
    assert (tidx == -1).all() and np.isinf(d).all()
    idx, dist = c.knn2(c.bank(Q), E)
    assert (idx == -1).all() and np.isinf(dist).all()
    tidx, d = c.xcheck1(E, c.bank(T))
    assert tidx.shape == (0,)


def test_f32_filter_scales_each_bank_by_its_own_power_of_two(filter_ctx):
    """Magnitudes far outside fp16's range still take the filter (banks are stored scaled);
    banks whose scales differ by more than 2^40, or that hold a non-finite value, stay on K5."""
    c = filter_ctx
    rng = np.random.default_rng(6)
    Q = (rng.normal(0, 1, (200, 128)) * 1e18).astype(np.float32)
    T = (rng.normal(0, 1, (300, 128)) * 1e18).astype(np.float32)
    before = c.f32_filter_stats()[0]
    idx, dist = c.knn2(c.bank(Q), c.bank(T))
    assert c.f32_filter_stats()[0] == before + 1
    oidx, odist = oracle.bf_knn(Q, T, 2, order=1)
    assert _eq(idx, oidx) and _eq(dist, odist)
    Ts = (T * np.float32(1e-30)).astype(np.float32)        # 2^160 apart: K5
    idx, dist = c.knn2(c.bank(Q), c.bank(Ts))
    assert c.f32_filter_stats()[0] == before + 1
    oidx, odist = oracle.bf_knn(Q, Ts, 2, order=1)
    assert _eq(idx, oidx) and _eq(dist, odist)
    Tn = rng.normal(0, 1, (300, 128)).astype(np.float32)
    Qn = rng.normal(0, 1, (200, 128)).astype(np.float32)
    Tn[17, 5] = np.inf
    idx, dist = c.knn2(c.bank(Qn), c.bank(Tn))
    assert c.f32_filter_stats()[0] == before + 1
    oidx, odist = oracle.bf_knn(Qn, Tn, 2, order=1)
    keep = np.arange(200)
    assert np.array_equal(idx[:, 0], oidx[:, 0]) and np.array_equal(dist[:, 0], odist[:, 0])   # (the inf row is never nearest)


def test_f32_route_close_to_opencv_order(ctx):
    # The device accumulates the float32 chain in ONE fixed order (fma chain, oracle order 1:
    # bit-exact, tested above).  OpenCV's own order depends on the build -- generic unrolled-by-4
    # (order 0), 2.4.x SSE2 lanes (order 2), 4.x 128-bit SIMD lanes (order 3) -- and differs from
    # the chain only in rounding.  Measured on BASELINE config 5 data (tests/tools/f32_ulp_report.py,
    # profiles/r02_f32_ulp_vs_opencv_orders.json): same neighbours in every row, distances
    # 0 ulp 34-36 %, <= 1 ulp 74-78 %, <= 2 ulp 95-97 %, max 5 ulp.  So the "within 1 ulp" of the
    # north star holds against the oracle's restatement of the device order, NOT against an
    # arbitrary OpenCV build; this test pins what is actually achieved.
    Q, T = _nonint(500, 1), _nonint(600, 2)
    idx, dist = ctx.knn2(ctx.bank(Q), ctx.bank(T))
    for order in (0, 2, 3):
        oidx, odist = oracle.bf_knn(Q, T, 2, order=order)
        same = idx == oidx
        assert same.mean() > 0.999, order
        ulp = np.abs(dist.view(np.int32).astype(np.int64) - odist.view(np.int32).astype(np.int64))[same]
        assert ulp.max() <= 6, (order, ulp.max())
        assert (ulp <= 1).mean() > 0.65 and (ulp <= 2).mean() > 0.90, (order, (ulp <= 1).mean(), (ulp <= 2).mean())   # (1000 values here)


def test_mixed_kind_pair_is_rejected(ctx):
    Q = _nonint(10, 3)
    T = synth.synth_sift(10, np.random.default_rng(4))
    with pytest.raises(_ffi.FastMatchHipError):
        ctx.xcheck1(ctx.bank(Q), ctx.bank(T))


def test_match_accepted_is_the_compacted_match_ratio(ctx):
    Q, T, _ = synth.planted_pair(5000, 4300, seed=61)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    qb.set_selfdist(ctx.self_dist(qb))
    tidx, dist, ratio, passed, npass = ctx.match_ratio(qb, tb, 0.7)
    qa, ta, da, ra = ctx.match_accepted(qb, tb, 0.7)
    sel = np.nonzero(passed)[0]
    assert len(qa) == npass and np.array_equal(qa, sel.astype(np.int32))      # ascending query index
    assert _eq(ta, tidx[sel]) and _eq(da, dist[sel]) and _eq(ra, ratio[sel])
    # caller-owned (pinned) buffers with a capacity below the count: truncated, order kept
    out = (ctx.pinned_empty(100, np.int32), ctx.pinned_empty(100, np.int32),
           ctx.pinned_empty(100, np.float32), ctx.pinned_empty(100, np.float64))
    qa2, ta2, da2, ra2 = ctx.match_accepted(qb, tb, 0.7, out=out)
    assert len(qa2) == 100 and np.array_equal(qa2, qa[:100]) and _eq(ra2, ra[:100])
    # nothing accepted
    qa3, _, _, _ = ctx.match_accepted(qb, tb, 0.0)
    assert len(qa3) == 0


def test_classic_ratio_match_knn2_ratio(ctx):
    """CR row: 2-NN + d1/d2 (float64) < tau, against the oracle's restatement of the notebook."""
    Q, T, _ = synth.planted_pair(3000, 2600, seed=17)
    T[10] = T[11]                                          # some query may see d1 == d2
    Q[4] = T[10]                                           # d1 = d2 = 0 -> nan -> rejected
    qb, tb = ctx.bank(Q), ctx.bank(T)
    qa, ta, da, ra = ctx.knn2_ratio(qb, tb, 0.7)
    oidx, od = oracle.bf_knn(Q, T, 2)
    orat = oracle.lowe_ratio(od)
    sel = np.nonzero(orat < 0.7)[0]
    assert len(sel) > 100 and 4 not in sel
    assert np.array_equal(qa, sel.astype(np.int32)) and _eq(ta, oidx[sel, 0]) and _eq(da, od[sel, 0]) and _eq(ra, orat[sel])
    q2, t2, d2, r2 = matchutil.ratio_match_arrays(Q, T, 0.7, {"context": ctx})
    assert np.array_equal(q2, qa) and _eq(r2, ra)
    # fewer than two train rows: nothing can pass
    assert len(ctx.knn2_ratio(qb, ctx.bank(T[:1]), 0.99)[0]) == 0


def test_train_set_beyond_opencv_18bit_limit(ctx):
    """OpenCV packs the train index in 18 bits and asserts nt < 262144;
    this path has no such limit: a 300k-row bank (config-3 size) against the oracle."""
    rng = np.random.default_rng(33)
    T = synth.synth_sift(300000, rng)
    Q = synth.synth_sift(512, rng)
    sel = rng.choice(300000, 200, replace=False)
    Q[:200] = np.clip(T[sel].astype(np.int32) + rng.integers(-4, 5, (200, 128)), 0, 255).astype(np.uint8)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    idx, dist = ctx.knn2(qb, tb)
    oidx, odist = oracle.bf_knn(Q, T, 2)
    assert _eq(idx, oidx) and _eq(dist, odist) and idx.max() >= 262144
    tidx, xd = ctx.xcheck1(qb, tb)
    otidx, oxd = oracle.bf_xcheck1(Q, T)
    assert _eq(tidx, otidx) and _eq(xd, oxd)
    # and the transposed problem (big query side)
    t2, d2 = ctx.xcheck1(tb, qb)
    ot2, od2 = oracle.bf_xcheck1(T, Q)
    assert _eq(t2, ot2) and _eq(d2, od2)


# ---- ONE large problem sharded over ranks: the per-shard pieces -----------
@pytest.mark.parametrize("as_f32", [False, True])
def test_train_sharded_crosscheck_keys_reduce_to_the_unsharded_result(ctx, as_f32):
    """fm_xcheck1_keys of each train-row shard, element-wise min (what the all-reduce does),
    decode == fm_xcheck1 on the whole train set, incl. ties across the shard boundary."""
    from fastmatch_amd import sharding
    rng = np.random.default_rng(11)
    if as_f32:
        Q = rng.normal(0, 1, (700, 128)).astype(np.float32)
        T = rng.normal(0, 1, (5000, 128)).astype(np.float32)
        T[4000] = T[10]; Q[5] = T[10]
    else:
        Q, T, _ = synth.planted_pair(700, 5000, seed=12)
        T[4000] = T[10]                                    # equal rows in different shards: lower global index wins
        Q[5] = T[10]
    qb = ctx.bank(Q)
    full_t, full_d = ctx.xcheck1(qb, ctx.bank(T))
    for world in (2, 3):
        keys = None
        for r in range(world):
            lo, hi = sharding.shard_rows(T.shape[0], r, world)
            k = ctx.xcheck1_keys(qb, ctx.bank(T[lo:hi]), lo)
            keys = k if keys is None else np.minimum(keys, k)
        t, d = sharding.decode_keys(keys)
        assert _eq(t, full_t) and _eq(d, full_d)
    # single process: the helper is the plain call
    t, d = sharding.xcheck1_sharded(ctx, qb, ctx.bank(T), 0)
    assert _eq(t, full_t) and _eq(d, full_d)
    # the device path: keys left in HBM (fm_xcheck1_keys_dev), reduced there (torch.minimum stands in
    # for the all-reduce of several ranks), none-keys mapped on the device
    import torch
    dev = torch.device("cuda", 0)
    for world in (1, 2):
        acc = None
        for r in range(world):
            lo, hi = sharding.shard_rows(T.shape[0], r, world)
            kt = torch.empty(qb.n, dtype=torch.int64, device=dev)
            ctx.xcheck1_keys_dev(qb, ctx.bank(T[lo:hi]), lo, kt.data_ptr())
            assert np.array_equal(kt.cpu().numpy().view(np.uint64), ctx.xcheck1_keys(qb, ctx.bank(T[lo:hi]), lo))
            kt.masked_fill_(kt == -1, torch.iinfo(torch.int64).max)
            acc = kt if acc is None else torch.minimum(acc, kt)
        acc.masked_fill_(acc == torch.iinfo(torch.int64).max, -1)
        t, d = sharding.decode_keys(sharding.reduce_keys_device(acc))
        assert _eq(t, full_t) and _eq(d, full_d)
    t, d = sharding.xcheck1_sharded(ctx, qb, ctx.bank(T), 0, device=dev)
    assert _eq(t, full_t) and _eq(d, full_d)
    with pytest.raises(Exception):
        ctx.xcheck1_keys_dev(qb, ctx.bank(T), 0, np.zeros(qb.n, np.uint64).ctypes.data)      # host memory
    lo, hi = sharding.shard_rows(700, 0, 1)
    i2, d2 = sharding.knn2_sharded(ctx, qb, ctx.bank(T), 700)
    oi, od = ctx.knn2(qb, ctx.bank(T))
    assert _eq(i2, oi) and _eq(d2, od)


def test_results_into_page_locked_buffers_and_offsets_into_them(ctx):
    """Outputs in page-locked caller memory are written by the library's copy kernel through
    the buffer's device alias -- also when the pointer is an offset into the allocation."""
    Q, T, _ = synth.planted_pair(3000, 3500, seed=91)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    qb.set_selfdist(ctx.self_dist(qb))
    ref = ctx.match_ratio(qb, tb, 0.7)
    n = qb.n
    big = (ctx.pinned_empty(n + 9, np.int32), ctx.pinned_empty(n + 9, np.float32),
           ctx.pinned_empty(n + 9, np.float64), ctx.pinned_empty(n + 13, np.uint8))
    for a in big:
        a[...] = 0x55 if a.dtype != np.float64 else -7.0
    out = (big[0][9:], big[1][9:], big[2][9:], big[3][13:])
    got = ctx.match_ratio(qb, tb, 0.7, out=out)
    assert _eq(got[0], ref[0]) and _eq(got[1], ref[1]) and np.array_equal(got[3].astype(bool), ref[3]) and got[4] == ref[4]
    m = ref[0] >= 0
    assert _eq(got[2][m], ref[2][m])
    assert (big[0][:9] == 0x55).all() and (big[3][:13] == 0x55).all() and (big[2][:9] == -7.0).all()   # nothing before the offset touched


@pytest.mark.parametrize("lpc", ["1", "16", "64"])
def test_f32_filter_rescoring_layouts_agree(filter_ctx, monkeypatch, lpc):
    """The three lane layouts of the rescoring kernel (picked by shape in production) give the
    same bits."""
    monkeypatch.setenv("FM_F32_LPC", lpc)
    Q, T = _f32_kind("rootsift", 2500, 3100, 77)
    Q[7] = T[3]; T[9] = T[3]                                # exact zero distance, duplicate train rows
    qb, tb = filter_ctx.bank(Q), filter_ctx.bank(T)
    idx, dist = filter_ctx.knn2(qb, tb)
    oidx, odist = oracle.bf_knn(Q, T, 2, order=1)
    assert _eq(idx, oidx) and _eq(dist, odist)
    tidx, xd = filter_ctx.xcheck1(qb, tb)
    otidx, oxd = oracle.bf_xcheck1(Q, T, order=1)
    assert _eq(tidx, otidx) and _eq(xd, oxd)


def test_full_size_float32_route_config5(monkeypatch):
    """BASELINE config 5 at full size (10k queries vs a 1M-row non-integer float32 bank): the
    fp16-MFMA filter route (K8) and the all-pairs kernel (K5) return the same bits for 2-NN and
    cross-check, 64 random query rows agree with the oracle, and the size-independent properties
    of a k-NN result hold (sorted pairs, indices in range, idempotent)."""
    import fastmatch_amd
    rng = np.random.default_rng(20250005)
    NT, NQ = 1000000, 10000
    T = synth.synth_sift(NT, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (NT, 128)).astype(np.float32)
    Q = synth.synth_sift(NQ, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (NQ, 128)).astype(np.float32)
    Q[:2000] = T[rng.choice(NT, 2000, replace=False)] + rng.normal(0, 3, (2000, 128)).astype(np.float32)   # planted near twins
    res = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("FM_F32_FILTER", mode)
        c = fastmatch_amd.Context(0)
        qb, tb = c.bank(Q), c.bank(T)
        res[mode] = (c.knn2(qb, tb), c.xcheck1(qb, tb))
        if mode == "2":
            again = c.knn2(qb, tb)
            assert _eq(again[0], res[mode][0][0]) and _eq(again[1], res[mode][0][1])          # idempotent
            launches, redone = c.f32_filter_stats()
            assert launches == 3 and redone == 0
        c.close()
    (i8, d8), (t8, x8) = res["2"]
    (i5, d5), (t5, x5) = res["0"]
    assert _eq(i8, i5) and _eq(d8, d5) and _eq(t8, t5) and _eq(x8, x5)
    assert i8.min() >= 0 and i8.max() < NT and np.all(d8[:, 0] <= d8[:, 1])
    assert (t8 >= 0).sum() > 1500                                                              # the planted twins cross-check
    rows = rng.choice(NQ, 64, replace=False)
    oi, od = oracle.bf_knn(Q[rows], T, 2, order=1)
    assert _eq(i8[rows], oi) and _eq(d8[rows], od)


def test_float32_route_results_do_not_depend_on_the_options(filter_ctx):
    """The K8 launch-shape / tuning options (waves per workgroup, splits, fused or separate rescoring, lanes per output row,
    how often the shared bounds are re-read) leave the 2-NN lists, the cross-check and the self distances bit-identical."""
    c = filter_ctx
    Q, T = _f32_kind("rootsift", 2300, 9100, 4242)
    Q[5] = T[11]; T[12] = T[11]                              # an exact zero distance and a duplicate train row
    qb, tb = c.bank(Q), c.bank(T)

    def run():
        return c.knn2(qb, tb) + c.xcheck1(qb, tb) + c.knn2(tb, qb) + (c.self_dist(tb),)
    ref = run()
    oi, od = oracle.bf_knn(Q, T, 2, order=1)
    assert _eq(ref[0], oi) and _eq(ref[1], od)
    domains = {"f32_nw": [0, 4, 8], "f32_nsplit": [0, 1, 2, 7], "f32_fused": [-1, 0, 1], "f32_lpc": [0, 1, 16, 64],
               "f32_bound_every": [1, 2, 4, 16, 64]}
    defaults = {k: c.get_option(k) for k in domains}
    rng = np.random.default_rng(11)
    settings = [{k: v} for k, vs in domains.items() for v in vs]
    settings += [{k: int(rng.choice(vs)) for k, vs in domains.items()} for _ in range(8)]
    for s in settings:
        for k, v in defaults.items():
            c.set_option(k, v)
        for k, v in s.items():
            c.set_option(k, v)
        got = run()
        assert all(_eq(a, b) for a, b in zip(got, ref)), s
    assert c.f32_filter_stats()[1] == 0                     # never through the all-pairs kernel


def test_results_do_not_depend_on_the_options():
    """fm_ctx_set_option: every launch-shape / tuning option, alone and in random combinations, leaves the
    2-NN lists, the cross-check and the accepted matches bit-identical (the header promises it)."""
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    Q, T, _ = synth.planted_pair(3000, 33000, seed=91)
    qb, tb = c.bank(Q), c.bank(T)
    qb.set_selfdist(c.self_dist(qb))

    def run():
        return c.knn2(qb, tb) + c.xcheck1(qb, tb) + c.knn2(tb, qb) + c.match_accepted(qb, tb, 0.8)
    ref = run()
    oi, od = oracle.bf_knn(Q, T, 2)
    assert _eq(ref[0], oi) and _eq(ref[1], od)
    domains = {"nsplit": [0, 1, 2, 5, 12], "nb": [0, 4, 8], "nw": [0, 4, 8], "nbuf": [0, 2, 3], "prio": [0, 1],
               "glds": [0, 1], "coop": [0, 1], "async_time_every": [0, 1, 4], "bound_every": [1, 2, 8, 64, 1024], "k1_order": [0, 1, 2]}
    defaults = {k: c.get_option(k) for k in domains}
    rng = np.random.default_rng(7)
    settings = [{k: v} for k, vs in domains.items() for v in vs]
    settings += [{k: int(rng.choice(vs)) for k, vs in domains.items()} for _ in range(12)]
    for s in settings:
        for k, v in defaults.items():
            c.set_option(k, v)
        for k, v in s.items():
            c.set_option(k, v)
        got = run()
        assert all(_eq(a, b) for a, b in zip(got, ref)), s
    c.close()


# ---- knnMatch for k > 2 (fm_knn) ---------------------------------------------------------
@pytest.mark.parametrize("k", [1, 2, 3, 5, 8])
def test_bf_match_for_any_k_up_to_8(ctx, k):
    """matchutil.bf_match(dt1, dt2, k) (matchutil.py:39-43: `k` is any int) == oracle.bf_knn on SIFT-like banks, on banks of
    duplicates and ties (lower train index first), in the float32-root tie range, on non-integer float32 banks, and on
    train sets with fewer than k rows (OpenCV returns shorter inner lists: -1 / inf here)."""
    rng = np.random.default_rng(900 + k)
    Q, T, _ = synth.planted_pair(700, 5300, seed=31 + k)
    T[100:140] = T[60]                                       # 41 identical train rows: ties in ascending index
    Q[5] = T[60]
    cases = [(Q, T), (Q[:3], T[:max(1, k - 1)]), (Q[:65], T[:k])]
    far_q, far_t = far_banks(120, 900, rng)                  # d2 up to the float32-root tie range
    cases.append((far_q, far_t))
    Qf = (Q[:300] + rng.uniform(-0.5, 0.5, (300, 128))).astype(np.float32)
    Tf = (T[:2100] + rng.uniform(-0.5, 0.5, (2100, 128))).astype(np.float32)
    Tf[700:704] = Tf[3]
    cases.append((Qf, Tf))
    for q, t in cases:
        order = 1 if q.dtype == np.float32 else 0
        oi, od = oracle.bf_knn(q, t, k, order=order)
        idx, dist = matchutil.bf_match_arrays(q, t, k=k, options={"context": ctx})
        assert _eq(idx, oi) and _eq(dist, od), (k, q.shape, t.shape, q.dtype)
    # the C-ABI entry point itself for every k (k = 1, 2 take the matrix-core path inside it); 704 query rows: a multiple of 64
    qb, tb = ctx.bank(Q[:704] if len(Q) >= 704 else Q), ctx.bank(T)
    oi, od = oracle.bf_knn(Q[:704] if len(Q) >= 704 else Q, T, k)
    idx, dist = ctx.knn(qb, tb, k)
    assert _eq(idx, oi) and _eq(dist, od)
    lists = matchutil.bf_match(Q[:4], T[:k - 1] if k > 1 else T[:1], k=k, options={"context": ctx})
    assert [len(m) for m in lists] == [max(1, k - 1) if k > 1 else 1] * 4      # shorter inner lists, as cv2 returns them
    if k == 3:
        with pytest.raises(ValueError):
            matchutil.bf_match_arrays(Q, T, k=9, options={"context": ctx})
        with pytest.raises(_ffi.FastMatchHipError) as e:
            ctx.knn(ctx.bank(Q), ctx.bank(T), 9)
        assert e.value.code == -4                            # FM_EUNSUPPORTED
        big_q, big_t, _ = synth.planted_pair(9000, 70000, seed=77)      # several splits of the train range
        oi, od = oracle.bf_knn(big_q, big_t, 4)
        idx, dist = ctx.knn(ctx.bank(big_q), ctx.bank(big_t), 4)
        assert _eq(idx, oi) and _eq(dist, od)
