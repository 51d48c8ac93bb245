"""BASELINE config 1 (images/graf img1 <-> img4, ratio 0.7) as PLUMBING: the README flow
(reference README.md:34-55) -- Metric_Cache(query_path), target pixel array,
fastmatch.match(query_cache, target_img, {'log': log})(0.7) -- on real pixel arrays through
the thumbnail path and the lazy Grid_Cache.cache() crop -> features -> per-cell bank path.

cv2 (SIFT) is installed neither here nor on the GPU box and the graf images do not travel,
so the pixels are procedural (tests/imagegen.py) and the features come from the labelled
stand-in extractor (fast-match_amd/standin.py: NOT SIFT).  What this pins is the plumbing
and its parity with the oracle on identical features, not matching quality on graf."""
import os

import numpy as np
import pytest

from fastmatch_amd import imaging, standin, evaluate
from imagegen import texture, warp, H1TO4P


# ---- host-side pieces (no GPU) ---------------------------------------------------------------
@pytest.mark.parametrize("wh,box", [((800, 640), (400, 400)), ((800, 640), (600, 600)), ((640, 800), (400, 400)),
                                    ((500, 500), (200, 200)), ((1000, 300), (400, 400)), ((300, 200), (400, 400))])
def test_get_thumbnail_sizes_like_the_reference(wh, box):
    """reference imaging.py:28-36,49-55: the longer side takes the box extent, the other one is
    int(aspect * it); PIL's thumbnail() then never enlarges and rounds the short side itself."""
    from PIL import Image
    w, h = wh
    img = texture(w, h, seed=w + h)
    th = imaging.get_thumbnail(img, box)
    # the same two calls on a PIL image, written out independently
    im = Image.fromarray(img)
    if w > h:
        new = (box[0], int((box[0] / float(w)) * h))
    else:
        new = (int((box[1] / float(h)) * w), box[1])
    im.thumbnail((new[0] * 2, new[1] * 2))
    im.thumbnail(new, Image.LANCZOS)
    assert th.dtype == np.uint8 and th.shape == (im.size[1], im.size[0], 3)
    assert np.array_equal(th, np.asarray(im))
    assert th.shape[1] <= max(new[0], 1) and th.shape[0] <= max(new[1], 1) or (w <= new[0] and h <= new[1])
    assert imaging.get_size(img) == (w, h)


def test_open_img_and_get_size_on_files(tmp_path):
    from PIL import Image
    img = texture(320, 200, seed=9)
    img[:, :, 0] //= 2                                   # make the channels differ
    path = str(tmp_path / "img1.ppm")
    Image.fromarray(img[:, :, ::-1]).save(path)          # file holds RGB; cv2.imread would return BGR
    assert imaging.get_size(path) == (320, 200)
    assert np.array_equal(imaging.open_img(path), img)
    assert np.array_equal(imaging.open_img(path, -1), img)
    assert imaging.open_img(path, (160, 160)).shape == (100, 160, 3)
    assert imaging.open_img(path, 160).shape == (100, 160, 3)        # Metric_Cache's max_size is one number (cache.pyx:158)
    th = imaging.get_thumbnail(path, (100, 100))         # target (100, 62); PIL's own aspect rounding may give 99
    assert th.shape[0] == 62 and th.shape[1] in (99, 100) and th.shape[2] == 3


def test_homography_scorer_on_the_graf_homography():
    """H1to4p maps img1 (target in the README flow) to img4 (query) coordinates."""
    rng = np.random.default_rng(4)
    p1 = rng.uniform([50, 50], [750, 590], (400, 2))
    q = np.concatenate([p1, np.ones((400, 1))], axis=1) @ H1TO4P.T
    p4 = q[:, :2] / q[:, 2:3]
    noise = rng.normal(0, 0.3, (400, 2))      # measured in the target image after H^-1 (scale ~1.5)
    wrong = rng.random(400) < 0.25
    p4n = p4 + noise + wrong[:, None] * rng.choice([-40.0, 40.0], (400, 2))
    positions = np.stack([p4n, p1], axis=1)              # (query = img4, target = img1), as do_iter returns them
    score = evaluate.homography_scorer(H1TO4P, distance_threshold=5.0, query_is_source=False)
    good = score(np.arange(400), positions, np.zeros(400))
    assert np.array_equal(good, ~wrong)
    # the file format of images/graf/H1toNp (three whitespace-separated rows)
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix="H1to4p", delete=False) as f:
        for row in H1TO4P:
            f.write("   " + "   ".join("%.7e" % v for v in row) + "\n")
    assert np.array_equal(evaluate.load_homography(f.name), H1TO4P)
    os.unlink(f.name)


def test_standin_features_format():
    img = texture(400, 300, seed=2)
    kp, d = standin.standin_features(img)
    assert len(kp) == len(d) > 300 and d.dtype == np.float32 and d.shape[1] == 128
    assert np.array_equal(d, np.rint(d)) and d.min() >= 0 and d.max() <= 255       # SIFT's value range
    assert all(8 <= k.pt[0] < 392 and 8 <= k.pt[1] < 292 for k in kp)
    kp2, d2 = standin.standin_features(img)
    assert [k.pt for k in kp] == [k.pt for k in kp2] and np.array_equal(d, d2)      # deterministic
    assert standin.standin_features(np.zeros((50, 50, 3), np.uint8)) == ([], None)
    assert standin.standin_features(img[:10, :10]) == ([], None)


# ---- the README flow on the device --------------------------------------------------------------
@pytest.mark.gpu
def test_readme_flow_on_pixels_equals_oracle(ctx, tmp_path, monkeypatch):
    from PIL import Image
    from fastmatch_amd import cache, fastmatch
    import oracle
    from oracle import fastmatch_oracle as fo
    monkeypatch.chdir(tmp_path)
    calls = []

    def feat(data):
        calls.append(data.shape)
        return standin.standin_features(data)
    img1 = texture(800, 640, seed=1)                                       # "img1": target
    mild = np.array([[1.0, 0.01, 18.0], [-0.008, 1.0, -11.0], [1e-5, -5e-6, 1.0]])   # the stand-in is not rotation invariant
    img4 = warp(img1, mild)                                                # "img4": query
    os.makedirs("images/graf")
    Image.fromarray(img1[:, :, ::-1]).save("images/graf/img1.ppm")
    Image.fromarray(img4[:, :, ::-1]).save("images/graf/img4.ppm")

    # README.md:41-50
    target_path, query_path = "images/graf/img1.ppm", "images/graf/img4.ppm"
    opts = {"context": ctx, "feature_function": feat}
    query_cache = cache.Metric_Cache(query_path, opts)                    # features + self 2-NN on the device + save()
    target_img = imaging.open_img(target_path)
    log, stats = [], {}
    match_fun = fastmatch.match(query_cache, target_img, dict(opts, log=log, stats=stats))
    matches = list(match_fun(0.7))
    # query image + query thumbnail + target thumbnail + one call per grid cell the expansion
    # reached (lazy: cache.pyx:102-106), far fewer than the 17 x 13 cells of the grid
    n_cells = len(calls) - 3
    assert calls[0] == (480, 600, 3) and calls[1] == (640, 800, 3) and 10 < n_cells < 17 * 13   # thumbnail, then full image (cache.pyx:168-169)
    assert all(c[0] <= 100 and c[1] <= 100 for c in calls[3:])
    assert os.path.isfile("data/image_data/%s.npz" % cache._ripemd160(query_path.encode()))
    again = cache.Metric_Cache(query_path, {"context": ctx})              # second time: load() hit, no features needed
    assert np.array_equal(again.original["descriptors"], query_cache.original["descriptors"])
    assert np.array_equal(again.original["distances"], query_cache.original["distances"])

    # the oracle on the same pixels / features
    feat = standin.standin_features
    kq, dq = feat(imaging.open_img(query_path))
    thumb_q = imaging.get_thumbnail(query_path, (600, 600))
    ktq, dtq = feat(thumb_q)
    pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    assert np.array_equal(query_cache.original["descriptors"], dq) and np.array_equal(query_cache.thumb["descriptors"], dtq)
    oq = fo.OQuery(dq, pos(kq), (800, 640),
                   thumb={"descriptors": dtq, "positions": pos(ktq), "size": (thumb_q.shape[1], thumb_q.shape[0])})
    assert np.array_equal(query_cache.original["distances"], oq.distances)
    assert np.array_equal(query_cache.thumb["distances"], oq.thumb["distances"])
    thumb_t = imaging.get_thumbnail(target_img, (400, 400))
    ktt, dtt = feat(thumb_t)
    ot = {"size": (800, 640), "image": target_img, "feature_function": feat,
          "thumb": {"descriptors": dtt, "positions": pos(ktt), "size": (thumb_t.shape[1], thumb_t.shape[0])}}
    olog = []
    oget = fo.o_match(oq, ot, {"log": olog})
    exp = oget(0.7)
    assert len(matches) == len(exp) > 50
    for (ia, da), (ib, db) in zip(matches, exp):
        assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    assert stats["rounds"] == oget.rounds == len(log) == len(olog) > 10
    for a, b in zip(log, olog):
        assert a["target_grid"] == b["target_grid"] and np.array_equal(a["matches"], b["matches"])
    # plumbing-level sanity against the known warp: most accepted matches obey it
    good = evaluate.homography_scorer(mild, 4.0, query_is_source=False)(
        np.array([m[0] for m in matches]), np.array([m[1]["positions"] for m in matches]), None)
    assert good.mean() > 0.8


@pytest.mark.gpu
def test_pixel_target_runs_the_loop_on_the_device_with_cells_computed_on_demand(ctx):
    """The reference's own mode -- fastmatch.match(query_cache, PIXELS): a grid cell's features are computed when the loop
    first reaches it (cache.pyx:102-106, 124-138).  r04: the loop still runs on the device; a round that needs a missing cell
    parks, the host computes the cell (one call of the feature function), registers it and resumes.  Same matches, rounds and
    computed cells as the host-driven loop and the oracle; later thresholds re-use the cells; a target bank that is too
    small hands the run to the host loop."""
    from fastmatch_amd import cache, fastmatch
    from oracle import fastmatch_oracle as fo
    img1 = texture(800, 640, seed=1)
    mild = np.array([[1.0, 0.01, 18.0], [-0.008, 1.0, -11.0], [1e-5, -5e-6, 1.0]])
    img4 = warp(img1, mild)
    feat = standin.standin_features
    kq, dq = feat(img4)
    thumb_q = imaging.get_thumbnail(img4, (600, 600))
    ktq, dtq = feat(thumb_q)
    pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    mc = cache.Metric_Cache.from_arrays(dq, pos(kq), (800, 640), dtq, pos(ktq), (thumb_q.shape[1], thumb_q.shape[0]),
                                        options={"context": ctx})
    calls = {"dev": [], "host": []}

    def counting(which):
        def f(data):
            calls[which].append(data.shape)
            return feat(data)
        return f
    ds, hs = {}, {}
    dev = fastmatch.match(mc, img1, {"context": ctx, "feature_function": counting("dev"), "stats": ds})
    host = fastmatch.match(mc, img1, {"context": ctx, "feature_function": counting("host"), "stats": hs, "device_loop": False})
    oq = fo.OQuery(dq, pos(kq), (800, 640), thumb={"descriptors": dtq, "positions": pos(ktq), "size": (thumb_q.shape[1], thumb_q.shape[0])})
    thumb_t = imaging.get_thumbnail(img1, (400, 400))
    ktt, dtt = feat(thumb_t)
    ot = {"size": (800, 640), "image": img1, "feature_function": feat,
          "thumb": {"descriptors": dtt, "positions": pos(ktt), "size": (thumb_t.shape[1], thumb_t.shape[0])}}
    oget = fo.o_match(oq, ot, {})
    cells_before = 0
    for tau in (0.7, 0.9, 0.6):
        ds.clear(); hs.clear()
        got, ref, exp = dev(tau), host(tau), oget(tau)
        assert ds.get("device_loops") == 1 and "device_fallbacks" not in ds
        assert len(got) == len(ref) == len(exp) > 20
        for (ia, da), (ib, db), (ic, dc) in zip(got, ref, exp):
            assert ia == ib == ic and da["ratio"] == db["ratio"] == dc["ratio"]
            assert np.array_equal(da["positions"], db["positions"]) and np.array_equal(da["positions"], dc["positions"])
        assert ds["rounds"] == hs["rounds"] == oget.rounds and ds["pairs"] == hs["pairs"]
        # the same cells were computed, each once, and only the cells the loop reached
        assert sorted(calls["dev"]) == sorted(calls["host"]) and 10 < len(calls["dev"]) - 1 < 17 * 13
        assert ds.get("lazy_cells", 0) == len(calls["dev"]) - 1 - cells_before
        cells_before = len(calls["dev"]) - 1
    many = dev([0.9, 0.7])
    assert [len(m) for m in many] == [len(oget(0.9)), len(oget(0.7))]
    # radius 300: subsets of thousands of rows (beyond the LDS tables) -- the lazy kernel is the chunked one
    big = {}
    wide = fastmatch.match(mc, img1, {"context": ctx, "feature_function": feat, "stats": big, "radius": 300})(0.7)
    wexp = fo.o_match(oq, ot, {"radius": 300})(0.7)
    assert big.get("device_loops") == 1 and "device_fallbacks" not in big and len(wide) == len(wexp) > 20
    assert big["pairs"] > 50 * 2048 * big["rounds"] / 4            # (subsets really are that large)
    for (ia, da), (ib, db) in zip(wide, wexp):
        assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    # ... and with EVERY round's cross-check handed to the dense kernels (expand_delegate = 1: a lazy run parks for them as it
    # parks for cells; both kinds of park in one run)
    keep = ctx.get_option("expand_delegate")
    ctx.set_option("expand_delegate", 1)
    ctx.set_option("delegated_rounds", 0)
    exp09 = oget(0.9)
    oget(0.7)                                                   # (oget.rounds is the last call's: the check further down wants 0.7's)
    try:
        for opts, want in (({"radius": 300}, wexp), ({}, exp09)):
            dl = {}
            o = {"context": ctx, "feature_function": feat, "stats": dl}
            o.update(opts)
            got = fastmatch.match(mc, img1, o)(0.7 if opts else 0.9)
            assert dl.get("device_loops") == 1 and "device_fallbacks" not in dl and len(got) == len(want)
            for (ia, da), (ib, db) in zip(got, want):
                assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    finally:
        ctx.set_option("expand_delegate", keep)
    assert ctx.get_option("delegated_rounds") > 500            # (nearly every round of the two runs)
    # a target bank with room for 64 rows: the second cell does not fit -> host loop, same result
    fb = {}
    small = fastmatch.match(mc, img1, {"context": ctx, "feature_function": feat, "stats": fb, "lazy_capacity": 64})(0.7)
    assert "device_loops" not in fb and fb["rounds"] == oget.rounds and len(small) == len(oget(0.7))


@pytest.mark.gpu
def test_readme_flow_on_the_graf_pixels(ctx, tmp_path, monkeypatch, capsys):
    """BASELINE.json configs[0] on the reference's OWN pixels (tests/golden/graf: images/graf/img1 and img4,
    re-encoded as PNG; H1to4p): README.md:41-50 with the stand-in extractor in place of cv2 SIFT (absent on
    both boxes; standin.py says what it is).  Pins the plumbing on real image content -- file -> thumbnail ->
    Metric_Cache -> lazy Grid_Cache cells -> rounds on the device -- against the oracle on the same features,
    and prints the homography precision as a STAND-IN number (it says nothing about SIFT matching quality)."""
    from PIL import Image
    from fastmatch_amd import cache, fastmatch
    from oracle import fastmatch_oracle as fo
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graf")
    monkeypatch.chdir(tmp_path)
    os.makedirs("images/graf")
    for n in ("img1", "img4"):                                             # the README's paths, as .ppm files
        Image.open(os.path.join(here, n + ".png")).save("images/graf/%s.ppm" % n)
    H = evaluate.load_homography(os.path.join(here, "H1to4p"))
    feat = standin.standin_features
    target_path, query_path = "images/graf/img1.ppm", "images/graf/img4.ppm"
    opts = {"context": ctx, "feature_function": feat}
    query_cache = cache.Metric_Cache(query_path, opts)
    target_img = imaging.open_img(target_path)
    assert target_img.shape == (640, 800, 3) and query_cache.original["size"] == (800, 640)
    log, stats = [], {}
    match_fun = fastmatch.match(query_cache, target_img, dict(opts, log=log, stats=stats))
    results = {tau: list(match_fun(tau)) for tau in (0.7, 0.9)}

    kq, dq = feat(imaging.open_img(query_path))
    thumb_q = imaging.get_thumbnail(query_path, (600, 600))
    ktq, dtq = feat(thumb_q)
    pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    oq = fo.OQuery(dq, pos(kq), (800, 640),
                   thumb={"descriptors": dtq, "positions": pos(ktq), "size": (thumb_q.shape[1], thumb_q.shape[0])})
    assert np.array_equal(query_cache.original["distances"], oq.distances)
    thumb_t = imaging.get_thumbnail(target_img, (400, 400))
    ktt, dtt = feat(thumb_t)
    ot = {"size": (800, 640), "image": target_img, "feature_function": feat,
          "thumb": {"descriptors": dtt, "positions": pos(ktt), "size": (thumb_t.shape[1], thumb_t.shape[0])}}
    oget = fo.o_match(oq, ot, {})
    for tau, matches in results.items():
        exp = oget(tau)
        assert len(matches) == len(exp)
        for (ia, da), (ib, db) in zip(matches, exp):
            assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    # query = img4, target = img1; H1to4p maps img1 -> img4 coordinates, i.e. target -> query
    score = evaluate.homography_scorer(H, 5.0, query_is_source=False)
    with capsys.disabled():
        for tau, matches in results.items():
            idx = np.array([m[0] for m in matches])
            p = np.array([m[1]["positions"] for m in matches]).reshape(-1, 2, 2)
            good = score(idx, p, None)
            print("\n[graf img1-img4, stand-in features (NOT SIFT), tau %.1f] %d keypoints/query image, %d matches, %d within 5 px of H1to4p"
                  % (tau, len(dq), len(matches), int(good.sum())))


@pytest.mark.gpu
def test_pixel_target_with_a_log_stays_on_the_device(ctx):
    """README.md:47-49 passes options = {'log': log}: until r05 that alone sent the run to the host-driven loop.  The lazy
    device loop now writes the records itself: log of the device run == log of the host loop (device_loop False), entry by
    entry, over two thresholds on one closure (the second finds every cell cached, so target_grid goes stale exactly as
    Grid_Cache.last does, cache.pyx:102-106), and the arrays' growth path (first capacity 8)."""
    from fastmatch_amd import cache, fastmatch
    img1 = texture(800, 640, seed=1)
    mild = np.array([[1.0, 0.01, 18.0], [-0.008, 1.0, -11.0], [1e-5, -5e-6, 1.0]])
    img4 = warp(img1, mild)
    feat = standin.standin_features
    kq, dq = feat(img4)
    thumb_q = imaging.get_thumbnail(img4, (600, 600))
    ktq, dtq = feat(thumb_q)
    pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    mc = cache.Metric_Cache.from_arrays(dq, pos(kq), (800, 640), dtq, pos(ktq), (thumb_q.shape[1], thumb_q.shape[0]),
                                        options={"context": ctx})
    for first_capacity in (0, 8):
        dlog, hlog, ds, hs = [], [], {}, {}
        dev = fastmatch.match(mc, img1, {"context": ctx, "feature_function": feat, "stats": ds, "log": dlog,
                                         "log_first_capacity": first_capacity})
        host = fastmatch.match(mc, img1, {"context": ctx, "feature_function": feat, "stats": hs, "log": hlog, "device_loop": False})
        for tau in (0.7, 0.9):
            got, ref = dev(tau), host(tau)
            assert len(got) == len(ref) > 20
            for (ia, da), (ib, db) in zip(got, ref):
                assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
        assert ds.get("device_loops") == 2 and "device_fallbacks" not in ds and ds["rounds"] == hs["rounds"] == len(dlog)
        assert len(dlog) == len(hlog)
        for a, b in zip(dlog, hlog):
            assert np.array_equal(a["query_pos"], b["query_pos"]) and np.array_equal(a["target_pos"], b["target_pos"])
            assert a["target_grid"] == b["target_grid"] and a["radius"] == b["radius"] and a["margin"] == b["margin"]
            assert np.array_equal(a["matches"], b["matches"]) and np.array_equal(a["ratios"], b["ratios"])
            assert np.asarray(a["matches"]).shape == np.asarray(b["matches"]).shape


@pytest.mark.gpu
def test_readme_flow_on_the_graf_pixels_with_real_sift(ctx, tmp_path, monkeypatch, capsys):
    """BASELINE.json configs[0] as the reference runs it: README.md:41-50 on images/graf/img1 <-> img4 with REAL SIFT
    (matchutil.get_features -> cv2, matchutil.py:22-33).  Skips where cv2 is absent (this image, the GPU image); the day
    it exists this runs by itself: device loop == host loop == oracle on cv2's features, and the precision against the
    dataset's homography H1to4p is PRINTED -- the quality number the reference never computed for graf."""
    pytest.importorskip("cv2")
    from PIL import Image
    from fastmatch_amd import cache, fastmatch, matchutil
    from oracle import fastmatch_oracle as fo
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graf")
    monkeypatch.chdir(tmp_path)
    os.makedirs("images/graf")
    for n in ("img1", "img4"):
        Image.open(os.path.join(here, n + ".png")).save("images/graf/%s.ppm" % n)
    H = evaluate.load_homography(os.path.join(here, "H1to4p"))
    target_path, query_path = "images/graf/img1.ppm", "images/graf/img4.ppm"
    feat = matchutil.get_features                                            # cv2 SIFT
    query_cache = cache.Metric_Cache(query_path, {"context": ctx})
    target_img = imaging.open_img(target_path)
    log, stats, hstats = [], {}, {}
    matches = list(fastmatch.match(query_cache, target_img, {"context": ctx, "log": log, "stats": stats})(0.7))
    host = list(fastmatch.match(query_cache, target_img, {"context": ctx, "stats": hstats, "device_loop": False})(0.7))
    assert len(matches) == len(host) and stats["rounds"] == hstats["rounds"] == len(log)
    for (ia, da), (ib, db) in zip(matches, host):
        assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    kq, dq = feat(imaging.open_img(query_path))
    thumb_q = imaging.get_thumbnail(query_path, (600, 600))
    ktq, dtq = feat(thumb_q)
    pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    oq = fo.OQuery(dq, pos(kq), (800, 640),
                   thumb={"descriptors": dtq, "positions": pos(ktq), "size": (thumb_q.shape[1], thumb_q.shape[0])})
    thumb_t = imaging.get_thumbnail(target_img, (400, 400))
    ktt, dtt = feat(thumb_t)
    ot = {"size": (800, 640), "image": target_img, "feature_function": feat,
          "thumb": {"descriptors": dtt, "positions": pos(ktt), "size": (thumb_t.shape[1], thumb_t.shape[0])}}
    exp = fo.o_match(oq, ot, {})(0.7)
    assert len(matches) == len(exp)
    for (ia, da), (ib, db) in zip(matches, exp):
        assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    score = evaluate.homography_scorer(H, 5.0, query_is_source=False)
    idx = np.array([m[0] for m in matches])
    p = np.array([m[1]["positions"] for m in matches]).reshape(-1, 2, 2)
    good = score(idx, p, None)
    with capsys.disabled():
        print("\n[graf img1-img4, cv2 SIFT, tau 0.7] %d keypoints/query image, %d matches, %d within 5 px of H1to4p = precision %.3f"
              % (len(dq), len(matches), int(good.sum()), float(good.mean()) if len(good) else 0.0))


@pytest.mark.gpu
def test_pixel_target_with_float32_descriptors_stays_on_the_device(ctx, monkeypatch):
    """RootSIFT-style (non-integer float32) descriptors on a PIXEL target (fastmatch.pyx:154 -> cache.pyx:102-106, 124-138):
    until r05 the one combination that always took the host-driven loop.  The target bank is a growing float32-route bank
    (fm_bank_create_f32_cap / fm_bank_append_f32, scaled like the query bank), the lazy kernel is the float32 chunked one:
    device loop == host loop == oracle (the device's accumulation order), same cells computed, log included."""
    from fastmatch_amd import cache, fastmatch, _ffi
    from oracle import fastmatch_oracle as fo
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    img1 = texture(800, 640, seed=1)
    mild = np.array([[1.0, 0.01, 18.0], [-0.008, 1.0, -11.0], [1e-5, -5e-6, 1.0]])
    img4 = warp(img1, mild)

    def feat(data):
        kp, ds = standin.standin_features(data)
        if ds is None or len(ds) == 0:
            return kp, ds
        d = np.asarray(ds, dtype=np.float32)
        return kp, np.sqrt(d / np.maximum(d.sum(axis=1, keepdims=True), 1.0)).astype(np.float32)
    kq, dq = feat(img4)
    thumb_q = imaging.get_thumbnail(img4, (600, 600))
    ktq, dtq = feat(thumb_q)
    pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    mc = cache.Metric_Cache.from_arrays(dq, pos(kq), (800, 640), dtq, pos(ktq), (thumb_q.shape[1], thumb_q.shape[0]),
                                        options={"context": ctx})
    assert mc.bank(ctx).kind == _ffi.FM_BANK_F32
    calls = {"dev": 0, "host": 0}

    def counting(which):
        def f(data):
            calls[which] += 1
            return feat(data)
        return f
    dlog, hlog, ds, hs = [], [], {}, {}
    dev = fastmatch.match(mc, img1, {"context": ctx, "feature_function": counting("dev"), "stats": ds, "log": dlog})
    host = fastmatch.match(mc, img1, {"context": ctx, "feature_function": counting("host"), "stats": hs, "log": hlog, "device_loop": False})
    oq = fo.OQuery(dq, pos(kq), (800, 640), thumb={"descriptors": dtq, "positions": pos(ktq), "size": (thumb_q.shape[1], thumb_q.shape[0])})
    thumb_t = imaging.get_thumbnail(img1, (400, 400))
    ktt, dtt = feat(thumb_t)
    ot = {"size": (800, 640), "image": img1, "feature_function": feat,
          "thumb": {"descriptors": dtt, "positions": pos(ktt), "size": (thumb_t.shape[1], thumb_t.shape[0])}}
    oget = fo.o_match(oq, ot, {})
    for tau in (0.8, 0.95):
        got, ref, exp = dev(tau), host(tau), oget(tau)
        assert len(got) == len(ref) == len(exp) > 20
        for (ia, da), (ib, db), (ic, dc) in zip(got, ref, exp):
            assert ia == ib == ic and da["ratio"] == db["ratio"] == dc["ratio"]
            assert np.array_equal(da["positions"], db["positions"]) and np.array_equal(da["positions"], dc["positions"])
    assert ds.get("device_loops") == 2 and "device_fallbacks" not in ds and ds["rounds"] == hs["rounds"]
    assert calls["dev"] == calls["host"] > 10
    assert len(dlog) == len(hlog) == ds["rounds"]
    for a, b in zip(dlog, hlog):
        assert a["target_grid"] == b["target_grid"] and np.array_equal(a["matches"], b["matches"]) and np.array_equal(a["ratios"], b["ratios"])
    # radius 300: subsets beyond the LDS tables (chunked float32 rounds), every cross-check delegated as well
    keep = ctx.get_option("expand_delegate")
    try:
        for deleg in (keep, 1):
            ctx.set_option("expand_delegate", deleg)
            big = {}
            wide = fastmatch.match(mc, img1, {"context": ctx, "feature_function": feat, "stats": big, "radius": 300})(0.8)
            wexp = fo.o_match(oq, ot, {"radius": 300})(0.8)
            assert big.get("device_loops") == 1 and "device_fallbacks" not in big and len(wide) == len(wexp) > 20
            for (ia, da), (ib, db) in zip(wide, wexp):
                assert ia == ib and da["ratio"] == db["ratio"] and np.array_equal(da["positions"], db["positions"])
    finally:
        ctx.set_option("expand_delegate", keep)


@pytest.mark.gpu
def test_pixel_target_whose_cells_turn_out_non_integer_stays_on_the_device(ctx, monkeypatch):
    """The last host-loop fallback of the lazy device loop (fastmatch.pyx:154 -> cache.pyx:102-106,
    124-138): an integer-valued query bank (plain SIFT) against a pixel target whose feature function returns descriptors
    that are NOT integer valued for some crops.  Until r06 the first such cell sent the run to the host-driven loop; now the
    pair moves to the float32 route on the device (the query bank's float32 twin, a fresh growing float32 target bank, the
    run from its start, cells already computed come from the Grid_Cache) -- device loop == host loop == oracle, log
    included, the feature function called as often as the host loop calls it."""
    from fastmatch_amd import cache, fastmatch, _ffi
    from oracle import fastmatch_oracle as fo
    monkeypatch.setattr(fo, "FLOAT_ORDER", 1)
    img1 = texture(800, 640, seed=1)
    mild = np.array([[1.0, 0.01, 18.0], [-0.008, 1.0, -11.0], [1e-5, -5e-6, 1.0]])
    img4 = warp(img1, mild)
    kq, dq = standin.standin_features(img4)
    thumb_q = imaging.get_thumbnail(img4, (600, 600))
    ktq, dtq = standin.standin_features(thumb_q)
    pos = lambda kp: np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
    mc = cache.Metric_Cache.from_arrays(dq, pos(kq), (800, 640), dtq, pos(ktq), (thumb_q.shape[1], thumb_q.shape[0]),
                                        options={"context": ctx})
    assert mc.bank(ctx).kind == _ffi.FM_BANK_I8
    calls = {"dev": 0, "host": 0, "oracle": 0}

    def feat(which):
        def f(data):
            calls[which] += 1
            kp, ds = standin.standin_features(data)
            if ds is None or len(ds) == 0 or data.shape[0] * data.shape[1] > 300 * 300:
                return kp, ds                                  # whole images and thumbnails: integer valued, as SIFT's are
            if int(data[::5, ::5, 0].sum()) % 3 == 0:               # (one channel: the texture has B = G = R)
                return kp, ds                                  # the cells' crops: integer valued ...
            return kp, np.asarray(ds, dtype=np.float32) + np.float32(0.25)     # ... except where this fires
        return f
    probe = [feat("oracle")(img1[y:y + 110, x:x + 110])[1] for y in range(0, 500, 100) for x in range(0, 600, 100)]
    kinds = set(bool(np.array_equal(np.asarray(p), np.round(np.asarray(p)))) for p in probe if p is not None and len(p))
    assert kinds == {True, False}, "the stand-in must return both kinds of cells for this test to mean anything"
    calls["oracle"] = 0
    dlog, hlog, ds, hs = [], [], {}, {}
    dev = fastmatch.match(mc, img1, {"context": ctx, "feature_function": feat("dev"), "stats": ds, "log": dlog})
    host = fastmatch.match(mc, img1, {"context": ctx, "feature_function": feat("host"), "stats": hs, "log": hlog, "device_loop": False})
    oq = fo.OQuery(dq, pos(kq), (800, 640), thumb={"descriptors": dtq, "positions": pos(ktq), "size": (thumb_q.shape[1], thumb_q.shape[0])})
    thumb_t = imaging.get_thumbnail(img1, (400, 400))
    ktt, dtt = standin.standin_features(thumb_t)
    ot = {"size": (800, 640), "image": img1, "feature_function": feat("oracle"),
          "thumb": {"descriptors": dtt, "positions": pos(ktt), "size": (thumb_t.shape[1], thumb_t.shape[0])}}
    oget = fo.o_match(oq, ot, {})
    for tau in (0.8, 0.95):
        got, ref, exp = dev(tau), host(tau), oget(tau)
        assert len(got) == len(ref) == len(exp) > 20
        for (ia, da), (ib, db), (ic, dc) in zip(got, ref, exp):
            assert ia == ib == ic and da["ratio"] == db["ratio"] == dc["ratio"]
            assert np.array_equal(da["positions"], db["positions"]) and np.array_equal(da["positions"], dc["positions"])
    assert ds.get("device_loops") == 2 and "device_fallbacks" not in ds and ds["rounds"] == hs["rounds"]
    assert calls["dev"] == calls["host"] > 10
    assert len(dlog) == len(hlog) == ds["rounds"]
    for a, b in zip(dlog, hlog):
        assert a["target_grid"] == b["target_grid"] and np.array_equal(a["matches"], b["matches"]) and np.array_equal(a["ratios"], b["ratios"])
        assert np.asarray(a["matches"]).shape == np.asarray(b["matches"]).shape
