"""GPU: the library gives back what it allocates.

Every byte of device memory behind the C-ABI belongs to an opaque handle -- banks (with their refill staging and
float32-route planes), expanders (run states, log arrays that grow, chunked-round tables, park / resume state), the
context's workspaces, plan and timer caches.  This soak cycles create / use / destroy of each of them, with forced failures
in between (a log of capacity 1 that must grow, a result list that is too small, a capacity request the device cannot
serve, misuse that is refused half way into a call), and asserts that fm_mem_info's free bytes come back to where they
were (to within one allocation granule of the runtime) and that no HIP error is left pending for the next call."""
import numpy as np
import pytest

import fastmatch_amd
from fastmatch_amd import cache, fastmatch, synth, _ffi

pytestmark = pytest.mark.gpu

GRANULE = 4 << 20                # the runtime hands memory back in 2 MiB pieces; two of them.  The checks are one-sided: what
                                 # must not happen is free memory going DOWN from cycle to cycle


def _free(c):
    c.sync()
    return c.mem_info()[0]


def _pair(ctx, size, n, seed):
    q, t = synth.image_pair(size, n, seed=seed)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], options={"context": ctx})
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"], t["thumb_positions"],
                             t["thumb_descriptors"], t["thumb_size"])
    return mc, fi


def test_bank_cycles_give_the_memory_back():
    """2000 cycles of bank create / refill / self distances / match / destroy, integer and float32 route, growing banks."""
    c = fastmatch_amd.Context(0)
    rng = np.random.default_rng(1)
    Q = synth.synth_sift(1500, rng)
    T = synth.synth_sift(2600, rng)
    Tf = (T + rng.uniform(-0.5, 0.5, T.shape)).astype(np.float32)
    src = c.pinned_empty((2600, 128), np.uint8)
    src[:] = T

    def cycle(k):
        qb, tb = c.bank(Q), c.bank(T)
        qb.set_selfdist(c.self_dist(qb))
        n = c.match_accepted(qb, tb, 0.8)[0].shape[0]
        if k % 4 == 0:                                     # refill in place, then the batched self distances
            tb.refill_async(src[:2000 + (k % 600)])
            c.upload_fence()
            c.self_dist_batch([tb, qb], want_host=False)
        if k % 5 == 0:                                     # float32 route: fp16 planes, K8's workspace
            fb = c.bank(Tf[:1200 + (k % 300)])
            c.knn2(fb, fb)
            fb.close()
        if k % 7 == 0:                                     # a growing bank and a gathered one
            gb = c.bank_with_capacity(T[:100], 4096)
            gb.append(T[100:900])
            c.knn2(qb, gb)
            gb.close()
            hb = c.bank_gather(T, np.arange(0, 2600, 3, dtype=np.int32))
            c.xcheck1(qb, hb)
            hb.close()
        qb.close(); tb.close()
        return n
    ref = cycle(0)
    for k in range(1, 40):                                 # warm-up: workspaces, plans and timer pools reach their sizes
        cycle(k)
    base = _free(c)
    for k in range(2000):
        n = cycle(k)
        if k % 20 == 0:
            assert n == ref if k % 4 else n >= 0
    after = _free(c)
    assert base - after <= GRANULE, (base, after)        # (more free than before: the runtime trimmed a pool of its own)
    c.close()


def test_error_paths_leave_no_memory_and_no_pending_error():
    """Refused calls and calls that fail half way (FM_ENOMEM from a capacity the device cannot serve, misuse of growing
    banks and lazy pairs) leave the free memory where it was and the context usable."""
    c = fastmatch_amd.Context(0)
    rng = np.random.default_rng(2)
    Q, T = synth.synth_sift(900, rng), synth.synth_sift(1400, rng)
    qb, tb = c.bank(Q), c.bank(T)
    ref = c.knn2(qb, tb)
    total = c.mem_info()[1]
    base = _free(c)
    for k in range(60):
        if k == 0:
            # a bank of 2.1e9 rows (269 GB + side arrays) fits this device ONCE: the second one fails in the middle of its
            # allocations (FM_ENOMEM), whatever it had got by then is given back, and so is the first one's memory
            big = c.bank_with_capacity(T[:10], 2100000000)
            with pytest.raises(_ffi.FastMatchHipError) as e:
                c.bank_with_capacity(T[:10], 2100000000)
            assert e.value.code == -3                                          # FM_ENOMEM
            with pytest.raises(_ffi.FastMatchHipError) as e:
                c.bank_f32_with_capacity(128, 400000000, c.bank(T.astype(np.float32) + 0.25))    # 300 GB of planes
            assert e.value.code in (-3, -4)
            big.close()
        with pytest.raises((_ffi.FastMatchHipError, ValueError)):
            tb.append(T[:5])                                              # not a growing bank
        with pytest.raises((_ffi.FastMatchHipError, ValueError)):
            c.knn2(qb, c.bank(T[:, :64]))                                 # dim mismatch, refused before any launch
        got = c.knn2(qb, tb)                                              # ... and the next call is clean
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
    import gc
    gc.collect()
    after = _free(c)
    assert base - after <= GRANULE, (base, after)        # (more free than before: the runtime trimmed a pool of its own)
    qb.close(); tb.close()
    c.close()


def test_expander_cycles_with_growth_parks_and_trims_give_the_memory_back():
    """Expander create -> runs whose log starts at capacity 8 and whose result list / stack start too small (growth of
    the run state and of the log arrays, FM_EXPAND_LOG_FULL retries inside the library) -> several thresholds in one
    launch (run slots) -> trim -> destroy; a pixel target whose cells are computed on demand (park / resume, a growing
    target bank).  Results stay identical from cycle to cycle."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from imagegen import texture, warp
    from fastmatch_amd import standin, imaging
    c = fastmatch_amd.Context(0)
    mc, fi = _pair(c, (640, 480), 2500, seed=11)
    seeds_ref = None

    def cycle(k):
        nonlocal seeds_ref
        grid = cache.Grid_Cache(fi, (50, 50), fi, margin=25)
        ex = fastmatch.make_expander(mc, grid, 100, c, match_cap=64 if k % 2 else 0, stack_cap=2048 if k % 3 == 0 else 0)
        ex.set_log(True, first_capacity=8 if k % 2 == 0 else 0)
        pos, ratios = fastmatch.match_thumbs(fi, mc, context=c)
        seeds = pos[ratios < 0.8]
        taus = [0.8, 0.6, 0.9] if k % 4 == 0 else [0.8]
        logs = [([], grid, 100) for _ in taus]
        out = fastmatch.run_device_loops(c, [ex] * len(taus), [seeds] * len(taus), taus, logs=logs)
        assert all(o is not None for o in out)
        ex.trim(1)
        n = (len(out[0]), len(logs[0][0]))
        ex.close()
        return n
    ref = cycle(0)
    for k in range(1, 6):
        cycle(k)
    base = _free(c)
    for k in range(120):
        n = cycle(k)
        assert n == ref
    after = _free(c)
    assert base - after <= GRANULE, (base, after)        # (more free than before: the runtime trimmed a pool of its own)

    # pixel target: lazy pair with park / resume, a log that grows, a target bank that grows
    img1 = texture(400, 320, seed=3)
    img4 = warp(img1, np.array([[1.0, 0.01, 9.0], [-0.008, 1.0, -6.0], [1e-5, -5e-6, 1.0]]))
    feat = standin.standin_features
    kq, dq = feat(img4)
    thumb_q = imaging.get_thumbnail(img4, (300, 300))
    ktq, dtq = feat(thumb_q)
    P = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    mcp = cache.Metric_Cache.from_arrays(dq, P(kq), (400, 320), dtq, P(ktq), (thumb_q.shape[1], thumb_q.shape[0]), options={"context": c})
    memo = {}

    def feat_memo(data):
        key = (data.shape, data[::7, ::7].tobytes())
        if key not in memo:
            memo[key] = feat(data)
        return memo[key]

    def lazy_cycle(k):
        log, st = [], {}
        get = fastmatch.match(mcp, img1, {"context": c, "feature_function": feat_memo, "stats": st, "log": log,
                                          "log_first_capacity": 8 if k % 2 == 0 else 0, "grid_size": (40, 40)})
        got = get(0.8)
        assert st.get("device_loops") == 1 and "device_fallbacks" not in st
        n = (len(got), len(log))
        del get
        return n
    lref = lazy_cycle(0)
    for k in range(1, 4):
        lazy_cycle(k)
    import gc
    gc.collect()
    base = _free(c)
    for k in range(40):
        assert lazy_cycle(k) == lref
        gc.collect()
    after = _free(c)
    assert base - after <= GRANULE, (base, after)        # (more free than before: the runtime trimmed a pool of its own)
    c.close()


def test_context_cycles_give_the_memory_back():
    """Context create -> every kind of call once -> destroy, 60 times: the device's free memory as seen by a context that
    outlives them all."""
    watcher = fastmatch_amd.Context(0)
    rng = np.random.default_rng(3)
    Q, T = synth.synth_sift(700, rng), synth.synth_sift(33000, rng)
    Tf = (T[:3000] + rng.uniform(-0.5, 0.5, (3000, 128))).astype(np.float32)

    def cycle():
        c = fastmatch_amd.Context(0)
        qb, tb, fb = c.bank(Q), c.bank(T), c.bank(Tf)
        qb.set_selfdist(c.self_dist(qb))
        c.self_dist(tb)                                    # the triangular sweep (33k rows)
        c.match_accepted(qb, tb, 0.8)
        c.knn2(qb, tb)
        c.knn2(fb, fb)
        c.xcheck1(fb, fb)
        c.close()                                          # with its banks still open: their handles die with it
    for _ in range(3):
        cycle()
    base = _free(watcher)
    for _ in range(60):
        cycle()
    after = _free(watcher)
    assert base - after <= GRANULE, (base, after)        # (more free than before: the runtime trimmed a pool of its own)
    watcher.close()
