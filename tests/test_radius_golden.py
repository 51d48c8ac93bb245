"""Radius query and Metric_Cache file format against vectors produced by the reference's own
dependency, sklearn.neighbors.BallTree, used exactly as cache.pyx:182-185,199-210,276 use it
(generator: tests/golden/make_radius_golden.py, run in the build container).

What is pinned: the SET of keypoints a radius query returns (boundary included), their
distances, and their ORDER wherever distances differ.  Among equal distances sklearn's order
is an implementation detail (not index order); there the product and the oracle use
ascending index -- a documented convention, checked here only for being a permutation of
sklearn's tie group."""
import json
import os
import pickle
import shutil

import numpy as np
import pytest

from fastmatch_amd import cache
from oracle import fastmatch_oracle as fo

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "radius_golden.json")))
METRIC_GOLDEN = json.load(open(os.path.join(HERE, "golden", "radius_metric_golden.json")))
NPZ_DIR = os.path.join(HERE, "golden", "metric_cache_npz")
NPZ_PATH = "images/graf/img4.ppm"


def _check(idx, dist, q):
    g_idx = np.array(q["indices"], dtype=np.int64)
    g_dist = np.array(q["distances"], dtype=np.float64)
    assert len(idx) == len(g_idx)
    assert sorted(idx.tolist()) == sorted(g_idx.tolist())            # same set, boundary inclusive
    assert np.array_equal(dist, g_dist)                               # same sorted distances, bit for bit
    assert np.all(np.diff(dist) >= 0)
    # order: identical wherever the distance is unique; tie groups are permutations
    uniq = np.ones(len(g_dist), dtype=bool)
    if len(g_dist) > 1:
        same = g_dist[1:] == g_dist[:-1]
        uniq[1:] &= ~same
        uniq[:-1] &= ~same
    assert np.array_equal(idx[uniq], g_idx[uniq])
    # our convention inside a tie group: ascending index
    for lo in range(len(dist)):
        if lo + 1 < len(dist) and dist[lo] == dist[lo + 1]:
            assert idx[lo] < idx[lo + 1]


@pytest.mark.parametrize("name", sorted(GOLDEN["sets"]))
def test_position_index_radius_equals_balltree(name):
    s = GOLDEN["sets"][name]
    pos = np.array(s["positions"], dtype=np.float64).reshape(-1, 2)
    index = cache.Position_Index(pos)
    n_ties = 0
    for q in s["queries"]:
        idx, d2 = index.radius(q["x"], q["y"], q["r"])
        _check(idx, np.sqrt(d2), q)
        n_ties += len(idx) - len(set(q["distances"]))
        # the BallTree-shaped method the reference calls (cache.pyx:182-185)
        inds, dists = index.query_radius(np.array((q["x"], q["y"])), r=q["r"], return_distance=True,
                                         sort_results=True)
        assert np.array_equal(inds[0], idx) and np.array_equal(dists[0], np.sqrt(d2))
    if name in ("lattice_60x40", "subpixel_800x640"):
        assert n_ties > 50                                             # the tie path is exercised


@pytest.mark.parametrize("name", sorted(GOLDEN["sets"]))
def test_metric_cache_get_and_oracle_get_equal_balltree(name):
    s = GOLDEN["sets"][name]
    pos = np.array(s["positions"], dtype=np.float64).reshape(-1, 2)
    n = len(pos)
    desc = (np.arange(n * 4, dtype=np.int64).reshape(n, 4) % 251).astype(np.uint8)
    sd = np.arange(n, dtype=np.float64) + 1.0
    mc = cache.Metric_Cache.from_arrays(desc, pos, (1000, 1000), distances=sd)
    oq = fo.OQuery(desc, pos, (1000, 1000), distances=sd)
    for q in s["queries"]:
        d, p, dis, idx = mc.get(q["x"], q["y"], q["r"])
        od, op, odis, oidx = oq.get(q["x"], q["y"], q["r"])
        assert np.array_equal(idx, oidx) and np.array_equal(d, od) and np.array_equal(p, op) and np.array_equal(dis, odis)
        dist = np.sqrt(((p - np.array([q["x"], q["y"]], dtype=np.float64)) ** 2).sum(1)) if len(idx) else np.zeros(0)
        assert sorted(idx.tolist()) == sorted(q["indices"])
        assert np.array_equal(d, desc[idx]) and np.array_equal(dis, sd[idx])
        assert np.allclose(dist, np.array(q["distances"]), rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", sorted(METRIC_GOLDEN["sets"]))
def test_other_metrics_equal_balltree(name):
    """options["metric"] (cache.pyx:160 -> BallTree(positions, metric = metric), cache.pyx:276): chebyshev,
    manhattan and euclidean against sklearn's BallTree built with that metric -- Position_Index,
    Metric_Cache.get and the oracle's OQuery.get: same set, same distances, same order outside ties."""
    s = METRIC_GOLDEN["sets"][name]
    metric = s["metric"]
    pos = np.array(s["positions"], dtype=np.float64).reshape(-1, 2)
    index = cache.Position_Index(pos, metric=metric)
    n = len(pos)
    desc = (np.arange(n * 4, dtype=np.int64).reshape(n, 4) % 251).astype(np.uint8)
    sd = np.arange(n, dtype=np.float64) + 1.0
    mc = cache.Metric_Cache.from_arrays(desc, pos, (1000, 1000), distances=sd, options={"metric": metric})
    oq = fo.OQuery(desc, pos, (1000, 1000), distances=sd, metric=metric)
    differs = 0
    l2 = cache.Position_Index(pos)
    for q in s["queries"]:
        idx, key = index.radius(q["x"], q["y"], q["r"])
        _check(idx, index.key_to_distance(key), q)
        inds, dists = index.query_radius(np.array((q["x"], q["y"])), r=q["r"], return_distance=True, sort_results=True)
        assert np.array_equal(inds[0], idx) and np.array_equal(dists[0], index.key_to_distance(key))
        _, _, _, midx = mc.get(q["x"], q["y"], q["r"])
        _, _, _, oidx = oq.get(q["x"], q["y"], q["r"])
        assert np.array_equal(midx, idx) and np.array_equal(oidx, idx)
        differs += int(len(l2.radius(q["x"], q["y"], q["r"])[0]) != len(idx))
    if metric != "euclidean" and "single" not in name:
        assert differs > 3                                  # the metric does change the selected sets


def test_unsupported_metric_is_rejected_not_ignored():
    pos = np.zeros((3, 2))
    for bad in ("haversine", "mahalanobis", "seuclidean"):
        with pytest.raises(ValueError):
            cache.Position_Index(pos, metric=bad)
        with pytest.raises(ValueError):
            cache.Metric_Cache.from_arrays(np.zeros((3, 4), np.uint8), pos, (10, 10), distances=np.ones(3), options={"metric": bad})
    with pytest.raises(ValueError):
        cache.Position_Index(pos, metric="minkowski", p=3)
    assert cache.Position_Index(pos, metric="minkowski", p=1).metric == cache.METRIC_L1
    assert cache.Position_Index(pos, metric="minkowski", p=float("inf")).metric == cache.METRIC_LINF
    assert cache.Position_Index(pos).metric == cache.METRIC_L2


def _install_fixture(tmp_path, monkeypatch):
    d = tmp_path / "data" / "image_data"
    d.mkdir(parents=True)
    for f in os.listdir(NPZ_DIR):
        shutil.copy(os.path.join(NPZ_DIR, f), d / f)
    monkeypatch.chdir(tmp_path)


def test_metric_cache_loads_a_reference_layout_npz(tmp_path, monkeypatch):
    """Metric_Cache(path) finds data/image_data/<ripemd160(path)>.npz written the way the
    reference writes it (cache.pyx:199-210: pickled BallTree bytes under `position_tree`,
    `size` as an array) and needs neither SIFT nor the device to load it."""
    _install_fixture(tmp_path, monkeypatch)
    mc = cache.Metric_Cache(NPZ_PATH)                       # load() hit -> no create_* (no cv2 needed)
    raw = np.load(os.path.join(NPZ_DIR, cache._ripemd160(NPZ_PATH.encode()) + ".npz"), allow_pickle=False)
    raw_t = np.load(os.path.join(NPZ_DIR, cache._ripemd160(NPZ_PATH.encode()) + "_thumb.npz"), allow_pickle=False)
    assert sorted(raw.files) == ["descriptors", "distances", "position_tree", "positions", "size"]
    assert sorted(raw_t.files) == ["descriptors", "distances", "positions", "size"]
    for k in ("descriptors", "positions", "distances"):
        assert np.array_equal(mc.original[k], raw[k]) and np.array_equal(mc.thumb[k], raw_t[k])
    assert mc.original["size"] == (800, 640) and mc.thumb["size"] == (600, 480)
    assert mc.original["descriptors"].dtype == np.float32 and mc.original["positions"].dtype == np.float64
    # the pickled tree in the file is sklearn's; ours is rebuilt from the positions and must
    # answer like it (the fixture is our own, trusted pickle; load() itself never unpickles)
    tree = pickle.loads(raw["position_tree"].tobytes())
    rng = np.random.default_rng(5)
    for _ in range(40):
        x, y, r = int(rng.integers(0, 800)), int(rng.integers(0, 640)), int(rng.choice([30, 100, 300]))
        ind, dist = tree.query_radius(np.array((x, y)).reshape(1, -1), r=r, return_distance=True, sort_results=True)
        d, p, dis, idx = mc.get(x, y, r)
        assert np.array_equal(idx, ind[0])                  # distinct distances: same order
        assert np.array_equal(d, raw["descriptors"][ind[0]])


def test_metric_cache_save_round_trips_in_the_reference_layout(tmp_path, monkeypatch):
    _install_fixture(tmp_path, monkeypatch)
    mc = cache.Metric_Cache(NPZ_PATH)
    out = tmp_path / "out"
    name = mc.save(str(out))
    assert name == cache._ripemd160(NPZ_PATH.encode())
    a = np.load(str(out / (name + ".npz")), allow_pickle=False)
    b = np.load(str(out / (name + "_thumb.npz")), allow_pickle=False)
    # the reference's keys (cache.pyx:199-210) + the name of the metric (the reference reads its keys by name)
    assert sorted(a.files) == ["descriptors", "distances", "fm_metric", "position_tree", "positions", "size"]
    assert sorted(b.files) == ["descriptors", "distances", "positions", "size"]
    # what the reference does with the file (cache.pyx:237: pickle.loads(position_tree), then query_radius at the first
    # get): a real sklearn BallTree is in there, answering like the tree the fixture was written with
    tree = pickle.loads(a["position_tree"].tobytes())
    ind, dist = tree.query_radius(np.array([[400.0, 320.0]]), r=100, return_distance=True, sort_results=True)
    assert np.array_equal(ind[0], mc.get(400, 320, 100)[3]) and len(ind[0]) > 3
    for k in ("descriptors", "positions", "distances"):
        assert np.array_equal(a[k], mc.original[k]) and np.array_equal(b[k], mc.thumb[k])
    assert a["size"].tolist() == [800, 640] and b["size"].tolist() == [600, 480]
