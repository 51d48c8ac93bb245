"""GPU: banks made by gathering the caller's rows on the device (fm_bank_create_u8_gather / _f32_gather) and the
packed target bank of a pre-extracted image built that way (fm_grid_pack_cells + gather) -- against the bank of the
same rows gathered on the host, and through it against the oracle."""
import numpy as np
import pytest

import oracle
from fastmatch_amd import synth, cache, fastmatch, _ffi

pytestmark = pytest.mark.gpu


def _same_answers(ctx, q, a, b):
    ia, da = ctx.knn2(q, a)
    ib, db = ctx.knn2(q, b)
    assert np.array_equal(ia, ib) and np.array_equal(da.view(np.uint32), db.view(np.uint32))
    ta, xa = ctx.xcheck1(q, a)
    tb, xb = ctx.xcheck1(q, b)
    assert np.array_equal(ta, tb) and np.array_equal(xa.view(np.uint32), xb.view(np.uint32))
    return ia, da


@pytest.mark.parametrize("n_src,n,dim", [(1, 1, 128), (40, 33, 128), (300, 1000, 128), (5000, 20011, 128), (257, 700, 64), (64, 0, 128)])
def test_gathered_u8_bank_equals_host_gather(ctx, n_src, n, dim):
    rng = np.random.default_rng(n_src + n)
    src = synth.synth_sift(n_src, rng)[:, :dim].copy()
    m = rng.integers(0, n_src, size=n).astype(np.int32)
    Q = synth.synth_sift(200, rng)[:, :dim].copy()
    g = ctx.bank_gather(src, m)
    assert (g.n, g.dim, g.kind) == (n, dim, _ffi.FM_BANK_I8)
    if n == 0:
        return
    idx, dist = _same_answers(ctx, ctx.bank(Q), g, ctx.bank(src[m]))
    oi, od = oracle.bf_knn(Q, src[m], k=2)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    # as the output side of the self distances too (norms and aux words of the gathered rows)
    assert np.array_equal(ctx.self_dist(g), ctx.self_dist(ctx.bank(src[m])))


def test_gathered_float_banks(ctx):
    rng = np.random.default_rng(8)
    src_i = synth.synth_sift(500, rng)
    m = rng.integers(0, 500, size=1500).astype(np.int32)
    Q = synth.synth_sift(150, rng)
    # float32 input with integer values: integer route, like fm_bank_create_f32
    g = ctx.bank_gather(src_i.astype(np.float32), m)
    assert g.kind == _ffi.FM_BANK_I8
    _same_answers(ctx, ctx.bank(Q), g, ctx.bank(src_i[m]))
    # ... kept on the float32 route on request
    gf = ctx.bank_gather(src_i, m, float_route=True)
    assert gf.kind == _ffi.FM_BANK_F32
    _same_answers(ctx, ctx.bank(Q, float_route=True), gf, ctx.bank(src_i[m], float_route=True))
    # values that are not integers (RootSIFT style)
    src_f = np.sqrt(src_i.astype(np.float32) / np.maximum(src_i.sum(axis=1, keepdims=True), 1).astype(np.float32))
    Qf = np.sqrt(Q.astype(np.float32) / np.maximum(Q.sum(axis=1, keepdims=True), 1).astype(np.float32))
    gr = ctx.bank_gather(src_f, m)
    assert gr.kind == _ffi.FM_BANK_F32
    _same_answers(ctx, ctx.bank(Qf), gr, ctx.bank(src_f[m]))


def test_gather_rejects_rows_outside_the_source(ctx):
    src = synth.synth_sift(10, np.random.default_rng(1))
    for bad in ([0, 10], [-1], [3, 2 ** 31 - 1]):
        with pytest.raises(_ffi.FastMatchHipError) as e:
            ctx.bank_gather(src, np.array(bad, dtype=np.int32))
        assert e.value.code == -1 and "src_row" in str(e.value)
    with pytest.raises(ValueError):
        ctx.bank_gather(src.reshape(-1), np.zeros(1, np.int32))


@pytest.mark.parametrize("float_route", [False, True])
def test_expander_on_a_gathered_target_equals_the_host_packed_one(ctx, float_route, monkeypatch):
    """make_expander's target bank (plan on the host, gather on the device) against Grid_Cache.pack_cells' matrix
    uploaded as it is: same matches; the device loop == the oracle (float32 descriptors: in the device's accumulation order)."""
    from oracle import fastmatch_oracle as fo
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    q, t = synth.image_pair((640, 480), 5000, seed=77, n_thumb=400)
    if float_route:
        for side in (q, t):
            for k in ("descriptors", "thumb_descriptors"):
                d = side[k].astype(np.float32)
                side[k] = np.sqrt(d / np.maximum(d.sum(axis=1, keepdims=True), 1.0)).astype(np.float32)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"], q["thumb_positions"],
                                        q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
    st = {}
    got = fastmatch.match(mc, fi, {"context": ctx, "stats": st})(0.8)
    assert st.get("device_loops", 0) == 1 and len(got) > 50
    # the same pair with the packed matrix built on the host and uploaded whole
    grid = cache.Grid_Cache(fi, (50, 50), fi, margin=25)
    descs, t_pos, cell_off = grid.pack_cells()
    q_bank = mc.bank(ctx)
    t_bank = ctx.bank(descs, float_route=(q_bank.kind == _ffi.FM_BANK_F32))
    assert t_bank.kind == q_bank.kind
    ex = _ffi.Expander(ctx, q_bank, mc.original["positions"], mc.original["position_tree"], t_bank, cell_off, t_pos,
                       {"width": grid.width, "height": grid.height, "cell_w": 50, "cell_h": 50, "rows": grid.rows, "cols": grid.cols,
                        "margin": 25}, 100)
    get = fastmatch.match(mc, fi, {"context": ctx})
    seeds = get.seeds_for(0.8)
    res = fastmatch.run_device_loops(ctx, [ex], [seeds], [0.8])[0]
    assert len(res) == len(got)
    for (ia, da), (ib, db) in zip(got, res):
        assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    thumb = {"descriptors": q["thumb_descriptors"], "positions": q["thumb_positions"], "size": q["thumb_size"]}
    oq = fo.OQuery(q["descriptors"], q["positions"], q["size"], thumb=thumb)
    ot = {"size": t["size"], "positions": t["positions"], "descriptors": t["descriptors"],
          "thumb": {"descriptors": t["thumb_descriptors"], "positions": t["thumb_positions"], "size": t["thumb_size"]}}
    exp = fo.o_match(oq, ot, {})(0.8)
    assert len(exp) == len(got)
    for (ia, da), (ib, db) in zip(got, exp):
        assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
