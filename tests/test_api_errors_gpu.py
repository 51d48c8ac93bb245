"""GPU: error behaviour and robustness of the C-ABI / Python surface (the analogue of the
cv2.error cases of the reference's operator boundary)."""
import threading

import numpy as np
import pytest

import fastmatch_amd
import oracle
from fastmatch_amd import _ffi, synth, cache

pytestmark = pytest.mark.gpu
Err = _ffi.FastMatchHipError


def test_bad_arguments_raise_and_context_survives(ctx):
    Q, T, _ = synth.planted_pair(50, 60, seed=1)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    with pytest.raises(Err):                                   # dim > 128
        ctx.bank(np.zeros((4, 129), np.uint8))
    with pytest.raises(ValueError):
        ctx.bank(np.zeros(128, np.uint8))                      # not 2-D
    with pytest.raises(Err):                                   # width mismatch (cv2.error in the reference)
        ctx.xcheck1(ctx.bank(Q[:, :64].copy()), tb)
    with pytest.raises(Err):                                   # ratio test without self distances
        ctx.match_ratio(qb, tb, 0.7)
    with pytest.raises(ValueError):
        qb.set_selfdist(np.ones(3))
    with pytest.raises(Err):                                   # q_rows out of range
        ctx.xcheck1_batched(qb, np.array([0, 50], np.int32), [0, 2], tb, [0, 10])
    with pytest.raises(Err):                                   # cell range beyond the bank
        ctx.xcheck1_batched(qb, np.array([0, 1], np.int32), [0, 2], tb, [0, 61])
    with pytest.raises(ValueError):
        ctx.xcheck1_batched(qb, np.array([0, 1], np.int32), [0, 2], tb, [0, 10, 20])
    with pytest.raises(Err):                                   # more than 4096 rows in one round
        big = ctx.bank(np.zeros((5000, 128), np.uint8))
        ctx.xcheck1_batched(big, np.arange(5000, dtype=np.int32), [0, 5000], tb, [0, 10])
    # the context still works after every failure
    tidx, dist = ctx.xcheck1(qb, tb)
    ot, od = oracle.bf_xcheck1(Q, T)
    assert np.array_equal(tidx, ot) and np.array_equal(dist, od)


def test_closed_bank_and_context_fail_cleanly():
    c = fastmatch_amd.Context(0)
    b = c.bank(np.ones((3, 128), np.uint8))
    name = c.device_name()
    assert name.startswith("gfx950")
    b.close()
    b.close()                                                  # idempotent
    with pytest.raises(Err):
        c.knn2(b, b)                                           # NULL bank handle
    c.close()
    c.close()
    with pytest.raises(Exception):
        c.bank(np.ones((3, 128), np.uint8))


def test_float_dtypes_are_converted_like_float32(ctx):
    Q, T, _ = synth.planted_pair(120, 90, seed=4)
    a = ctx.xcheck1(ctx.bank(Q.astype(np.float64)), ctx.bank(T.astype(np.int32)))
    b = oracle.bf_xcheck1(Q, T)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # non-contiguous input views are handled (copied) on the way in
    Qs = np.asfortranarray(Q)
    c = ctx.knn2(ctx.bank(Qs), ctx.bank(T[::-1][::-1]))
    d = oracle.bf_knn(Q, T, 2)
    assert np.array_equal(c[0], d[0]) and np.array_equal(c[1], d[1])


def test_two_contexts_on_two_threads(ctx):
    """Distinct contexts are independent (one stream each): concurrent use from two host
    threads gives the single-threaded answers."""
    Q, T, _ = synth.planted_pair(3000, 2500, seed=9)
    exp = oracle.bf_knn(Q, T, 2)
    out, errs = {}, []

    def work(k):
        try:
            c = fastmatch_amd.Context(0)
            qb, tb = c.bank(Q), c.bank(T)
            for _ in range(5):
                out[k] = c.knn2(qb, tb)
            c.close()
        except Exception as e:       # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    for k in range(2):
        assert np.array_equal(out[k][0], exp[0]) and np.array_equal(out[k][1], exp[1])


def test_stats_account_pairs_and_kernel_time(ctx):
    Q, T, _ = synth.planted_pair(700, 900, seed=2)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    ctx.reset_stats()
    ctx.xcheck1(qb, tb)
    ctx.knn2(qb, tb)
    s = ctx.stats()
    assert s["pairs"] == 2 * 700 * 900 and s["kernel_launches"] == 2 and s["calls"] == 2
    assert 0 < s["kernel_ms"] <= s["total_ms"]


def test_metric_cache_from_arrays_computes_exact_self_distances(ctx, tmp_path):
    q, _ = synth.image_pair((300, 200), 400, seed=8, n_thumb=80)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], path=b"q.ppm",
                                        options={"context": ctx})
    assert np.array_equal(mc.original["distances"], oracle.self_dist(q["descriptors"]))
    assert np.array_equal(mc.thumb["distances"], oracle.self_dist(q["thumb_descriptors"]))
    mc.save(str(tmp_path))
    mc2 = cache.Metric_Cache(None, {"context": ctx})
    mc2.path = b"q.ppm"
    assert mc2.load(str(tmp_path))
    ds, ps, dis, idx = mc2.get(150, 100, 60)
    eds, eps, edis, eidx = mc.get(150, 100, 60)
    assert np.array_equal(idx, eidx) and np.array_equal(dis, edis)
    assert mc2.bank(ctx).n == 400                            # lazily re-created device bank


def test_batch_calls_reject_bad_arguments_and_the_context_survives(ctx):
    """fm_match_accepted_batch / _dev_batch: pageable outputs, missing self distances, float32-route
    banks, host pointers for device outputs -> error with a message; an empty batch is a no-op; the
    context keeps working afterwards."""
    import ctypes
    from fastmatch_amd import _ffi
    Q, T, _ = synth.planted_pair(600, 500, seed=3)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    outs = [(ctx.pinned_empty(600, np.int32), ctx.pinned_empty(600, np.int32), ctx.pinned_empty(600, np.float32),
             ctx.pinned_empty(600, np.float64))]
    cnt = [ctx.pinned_empty(1, np.int64)]
    with pytest.raises(_ffi.FastMatchHipError):                         # no self distances on the query bank
        ctx.match_accepted_batch([(qb, tb)], 0.7, outs, cnt)
    ctx.sync()
    qb.set_selfdist(ctx.self_dist(qb))
    ctx.match_accepted_batch([], 0.7, [], [])                           # empty batch
    with pytest.raises(ValueError):                                     # lists of different lengths
        ctx.match_accepted_batch([(qb, tb)], 0.7, outs, [])
    qf = ctx.bank((Q.astype(np.float32) + 0.25))
    qf.set_selfdist(ctx.self_dist(qf))
    tf = ctx.bank(T.astype(np.float32) + 0.25)                          # float32 route: runs synchronously in place
    ctx.match_accepted_batch([(qf, tf)], 0.7, outs, cnt)
    fq, ft, fd, fr = ctx.match_accepted(qf, tf, 0.7)
    m = int(cnt[0][0])
    assert m == len(fq) and np.array_equal(outs[0][0][:m], fq) and np.array_equal(outs[0][3][:m], fr)
    ctx.sync()
    ctx.match_accepted_batch([(qb, tb)], 0.7, outs, cnt)                # and it still works
    ctx.sync()
    qa, ta, da, ra = ctx.match_accepted(qb, tb, 0.7)
    m = int(cnt[0][0])
    assert m == len(qa) and np.array_equal(outs[0][0][:m], qa) and np.array_equal(outs[0][3][:m], ra)
    with pytest.raises(_ffi.FastMatchHipError):                         # device outputs must be device memory
        ctx.match_accepted_dev_batch([(qb, tb)], 0.7, np.zeros((600, 3), np.int32).ctypes.data,
                                     np.zeros(1, np.int64).ctypes.data, 600)
    ctx.sync()


def test_growing_banks_refills_and_lazy_pairs_reject_misuse(ctx):
    """r04 entry points: fm_bank_create_u8_cap / fm_bank_append_u8 (rows land on 32-row boundaries, the appended bank matches a
    bank made of the same rows), fm_bank_refill_u8_async, fm_expand_set_cell / fm_expand_run_lazy argument checks."""
    rng = np.random.default_rng(8)
    A, B, C = (synth.synth_sift(n, rng) for n in (100, 37, 64))
    T = synth.synth_sift(500, rng)
    tb = ctx.bank(T)
    g = ctx.bank_with_capacity(A, 400)
    assert g.n == 100 and g.append(B) == 128 and g.n == 165 and g.append(C) == 192 and g.n == 256
    assert g.append(np.zeros((0, 128), np.uint8)) == 256 and g.n == 256
    # the rows between the pieces are padding rows: as the reduced (query) side of a cross-check they are never elected and
    # come back unmatched, and the real rows behave like the same rows in a bank of their own
    whole = np.zeros((256, 128), np.uint8)
    whole[:100], whole[128:165], whole[192:256] = A, B, C
    real = np.r_[0:100, 128:165, 192:256]
    gap = np.setdiff1d(np.arange(256), real)
    t1, d1 = ctx.xcheck1(g, tb)
    t2, d2 = ctx.xcheck1(ctx.bank(whole[real]), tb)
    assert np.array_equal(t1[real], t2) and np.array_equal(d1[real], d2) and np.all(t1[gap] == -1) and (t2 >= 0).sum() > 50
    with pytest.raises(Err):                                   # (a capacity of 400 rows is rounded up to 512)
        ctx.bank_with_capacity(A, 400).append(synth.synth_sift(600, rng))
    with pytest.raises(ValueError):
        g.append(np.zeros((3, 64), np.uint8))
    f = ctx.bank(A.astype(np.float32) + 0.5)
    with pytest.raises(Err):                                   # float32-route banks neither grow nor refill
        ctx._check(ctx.lib.fm_bank_append_u8(ctx.handle, f.handle, A.ctypes.data, 10, None))
    with pytest.raises(Err):
        ctx._check(ctx.lib.fm_bank_refill_u8_async(ctx.handle, f.handle, A.ctypes.data, 10))
    # lazy pairs: a non-lazy expander takes neither set_cell nor run_lazy; a lazy one does not take fm_expand_run
    q, t = synth.image_pair((300, 200), 400, seed=5)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    from fastmatch_amd import fastmatch
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"], t["thumb_descriptors"], t["thumb_size"])
    eager = fastmatch.match(mc, fi, {"context": ctx}).expander()
    with pytest.raises(Err):
        eager.set_cell(0, 0, np.zeros((0, 2)))
    with pytest.raises(Err):
        eager.run_lazy(np.zeros((0, 2, 2)), 0.7, False)
    grid = fastmatch.Grid_Cache(np.zeros((200, 300, 3), np.uint8), (50, 50), None, margin=25)
    lazy, tbank = fastmatch.make_lazy_expander(mc, grid, 100, ctx)
    with pytest.raises(Err):
        ctx.expand_run([lazy], [np.zeros((0, 2, 2))], [0.7])
    with pytest.raises(Err):
        lazy.set_cell(10 ** 6, 0, np.zeros((0, 2)))             # no such cell
    with pytest.raises(Err):
        lazy.set_cell(0, 0, np.zeros((5, 2)))                   # rows the target bank does not hold
    assert lazy.run_lazy(np.zeros((0, 2, 2)), 0.7, False)[3] == 0        # no seeds: done at once


def test_r05_entry_points_reject_misuse(ctx):
    """r05: a resume of a run that is not parked (it would restore loop state nothing had saved), the growing
    float32-route bank (scale source, range check, capacity), the log calls on a pair that writes none, fm_self_dist_plan."""
    from fastmatch_amd import fastmatch, _ffi
    rng = np.random.default_rng(9)
    q, t = synth.image_pair((300, 200), 400, seed=6)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    grid = fastmatch.Grid_Cache(np.zeros((200, 300, 3), np.uint8), (50, 50), None, margin=25)
    lazy, tbank = fastmatch.make_lazy_expander(mc, grid, 100, ctx)
    seeds = np.array([[[150.0, 100.0], [150.0, 100.0]]])
    with pytest.raises(Err):
        lazy.run_lazy(seeds, 0.7, True)                         # nothing is parked yet
    nm, nr, npairs, status, need = lazy.run_lazy(seeds, 0.7, False)
    assert status == _ffi.FM_EXPAND_NEED_CELL and need >= 0     # parked at the seed's cell
    lazy.set_cell(need, 0, np.zeros((0, 2)))
    assert lazy.run_lazy(seeds, 0.7, True)[3] == 0              # resumed and done (the cell has no features)
    with pytest.raises(Err):
        lazy.run_lazy(seeds, 0.7, True)                         # ... and a finished run cannot be resumed again
    # log calls on a pair that writes none
    with pytest.raises(Err):
        lazy.fetch_log()
    # growing float32-route bank
    F = synth.synth_sift(200, rng).astype(np.float32) + 0.25
    fq = ctx.bank(F)
    with pytest.raises(Err):
        ctx.bank_f32_with_capacity(128, 1000, ctx.bank(synth.synth_sift(10, rng)))      # the scale source must be a float32-route bank
    g = ctx.bank_f32_with_capacity(128, 300, fq)
    assert g.kind == _ffi.FM_BANK_F32 and g.n == 0
    assert g.append(F[:50]) == 0 and g.append(F[50:87]) == 64 and g.n == 101
    t1, d1 = ctx.xcheck1(fq, g)
    whole = ctx.bank(np.concatenate([F[:50], F[50:87]]), float_route=True)
    t2, d2 = ctx.xcheck1(fq, whole)
    remap = np.r_[0:50, 64:101]
    assert np.array_equal(np.where(t2 >= 0, remap[np.maximum(t2, 0)], -1), t1) and np.array_equal(d1, d2)
    with pytest.raises(Err):
        g.append(F[:10] * 1000.0)                               # leaves fp16's range under the bank's scale
    assert g.n == 101
    with pytest.raises(Err):
        g.append(np.full((3, 128), np.inf, np.float32))
    with pytest.raises(Err):
        g.append(np.tile(F, (3, 1)))                            # capacity (300 -> 384 rows) used up
    assert _ffi.self_dist_plan(1024, 5)[1] == 2
