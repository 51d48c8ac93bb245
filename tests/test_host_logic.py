"""CPU: host-side logic of the product package against brute-force restatements."""
import os

import numpy as np
import pytest

from fastmatch_amd import cache, matchutil, synth
from oracle.fastmatch_oracle import OGrid, OQuery


def test_position_index_matches_bruteforce_order_and_boundary():
    rng = np.random.default_rng(5)
    pos = np.floor(rng.uniform(0, 300, (3000, 2)))          # integer coords: many exact ties
    pos[10] = pos[11] = pos[12]                              # coincident keypoints (SIFT orientations)
    idx = cache.Position_Index(pos, bucket=37.0)
    oq = OQuery(np.zeros((3000, 4), np.uint8), pos, (300, 300), distances=np.ones(3000))
    for (x, y, r) in [(150, 150, 100), (0, 0, 50), (299, 10, 75), (pos[10, 0], pos[10, 1], 0),
                      (150.7, 20.2, 5), (1000, 1000, 10)]:
        got, d2 = idx.radius(int(x), int(y), int(r))
        exp = oq.get(x, y, r)[3]
        assert np.array_equal(got, exp)
        assert np.all(d2 <= int(r) ** 2)
    # inclusive boundary: a point at distance exactly r
    p = np.array([[0.0, 0.0], [3.0, 4.0], [3.0, 4.1]])
    got, _ = cache.Position_Index(p).radius(0, 0, 5)
    assert got.tolist() == [0, 1]


def test_query_radius_sklearn_like_shape():
    pos = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 2.0], [5.0, 5.0]])
    tree = cache.Position_Index(pos)
    inds, dists = tree.query_radius(np.array((0, 0)), r=2, return_distance=True, sort_results=True)
    assert inds[0].tolist() == [0, 1, 2] and dists[0].tolist() == [0.0, 1.0, 2.0]


def test_metric_cache_get_truncates_and_orders():
    q, _ = synth.image_pair((400, 300), 500, seed=3)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"],
                                        distances=np.ones(500))
    oq = OQuery(q["descriptors"], q["positions"], q["size"], distances=np.ones(500))
    ds, ps, dis, idx = mc.get(120.9, 80.9, 60.9)              # C-int truncation: (120, 80, 60)
    eds, eps, edis, eidx = oq.get(120, 80, 60)
    assert np.array_equal(idx, eidx) and np.array_equal(ds, eds) and np.array_equal(ps, eps)
    assert len(idx) > 0


def test_feature_image_cells_match_oracle_grid():
    _, t = synth.image_pair((500, 333), 2000, seed=9)
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"])
    g = cache.Grid_Cache(fi, (50, 50), fi, margin=25)
    o = OGrid(t["size"], (50, 50), 25, t["positions"], t["descriptors"])
    for (x, y) in [(0, 0), (499, 332), (500, 333), (260, 170), (75, 25), (30, 310)]:
        kp, ds = g.get(x, y)
        okp, ods = o.get(x, y)
        assert np.array_equal(kp, okp)
        assert (ds is None and ods is None) or np.array_equal(ds, ods)
        assert g.last == o.last


def test_ripemd160_known_answers():
    import hashlib
    orig = hashlib.new
    try:
        def boom(*a, **k):
            raise ValueError("unsupported")
        hashlib.new = boom                                    # force the pure-Python path
        assert cache._ripemd160(b"") == "9c1185a5c5e9fc54612808977ee8f548b2258d31"
        assert cache._ripemd160(b"abc") == "8eb208f7e05d987a9b044a8e98c6b087f15a0bfc"
        assert cache._ripemd160(b"message digest") == "5d0689ef49d2fae572b881b123a85ffa21595f36"
        assert cache._ripemd160(b"a" * 1000000)[:8] == "52783243"
    finally:
        hashlib.new = orig


def test_metric_cache_save_load_roundtrip(tmp_path):
    q, _ = synth.image_pair((200, 200), 300, seed=4, n_thumb=50)
    sd = np.linspace(1, 2, 300)
    tsd = np.linspace(2, 3, 50)
    mc = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], distances=sd,
                                        thumb_distances=tsd, path=b"images/graf/img4.ppm")
    name = mc.save(str(tmp_path))
    assert name == cache._ripemd160(b"images/graf/img4.ppm")
    assert os.path.isfile(os.path.join(str(tmp_path), name + ".npz"))
    assert os.path.isfile(os.path.join(str(tmp_path), name + "_thumb.npz"))
    with np.load(os.path.join(str(tmp_path), name + ".npz")) as z:   # the reference's key set (+ the metric's name)
        assert set(z.files) == {"descriptors", "positions", "distances", "position_tree", "size", "fm_metric"}
        # what the reference does with the file at load (cache.pyx:237): unpickle the tree and query it
        import pickle
        tree = pickle.loads(z["position_tree"].tobytes())
        if tree is not None:                                          # (scikit-learn present: a real BallTree)
            ind, dist = tree.query_radius(np.array([[100.0, 100.0]]), r=50, return_distance=True, sort_results=True)
            got, d2 = mc.original["position_tree"].radius(100, 100, 50)
            assert sorted(ind[0].tolist()) == sorted(got.tolist()) and np.allclose(np.sort(dist[0]), np.sqrt(np.sort(d2)))
    mc2 = cache.Metric_Cache(None)
    mc2.path = b"images/graf/img4.ppm"
    assert mc2.load(str(tmp_path))
    for k in ("descriptors", "positions", "distances"):
        assert np.array_equal(mc2.original[k], mc.original[k])
        assert np.array_equal(mc2.thumb[k], mc.thumb[k])
    assert mc2.original["size"] == (200, 200) and mc2.thumb["size"] == mc.thumb["size"]
    assert np.array_equal(mc2.get(100, 100, 50)[3], mc.get(100, 100, 50)[3])
    mc3 = cache.Metric_Cache(None)
    mc3.path = b"some/other/path"
    assert mc3.load(str(tmp_path)) is False
    # the metric travels with the file (the reference's pickled tree keeps the metric it was built with, cache.pyx:276)
    mm = cache.Metric_Cache.from_arrays(q["descriptors"], q["positions"], q["size"], q["thumb_descriptors"],
                                        q["thumb_positions"], q["thumb_size"], distances=sd, thumb_distances=tsd,
                                        path=b"manhattan/img", options={"metric": "manhattan"})
    mm.save(str(tmp_path))
    back = cache.Metric_Cache(None)                                    # (default options: "minkowski")
    back.path = b"manhattan/img"
    assert back.load(str(tmp_path))
    assert np.array_equal(back.get(100, 100, 50)[3], mm.get(100, 100, 50)[3])
    assert not np.array_equal(back.get(100, 100, 50)[3], mc.get(100, 100, 50)[3])


def test_matches_from_arrays_drops_missing():
    idx = np.array([[2, -1], [0, 1]], dtype=np.int32)
    dist = np.array([[1.5, np.inf], [0.0, 2.0]], dtype=np.float32)
    m = matchutil.matches_from_arrays(idx, dist)
    assert [len(r) for r in m] == [1, 2]
    assert (m[0][0].queryIdx, m[0][0].trainIdx, m[0][0].distance, m[0][0].imgIdx) == (0, 2, 1.5, 0)
    m1 = matchutil.matches_from_arrays(np.array([-1, 3]), np.array([np.inf, 2.0], dtype=np.float32))
    assert m1[0] == [] and m1[1][0].trainIdx == 3


def test_synth_is_deterministic_and_sift_like():
    a = synth.planted_pair(200, 300, seed=1)
    b = synth.planted_pair(200, 300, seed=1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    Q, T, planted = a
    assert Q.dtype == np.uint8 and Q.shape == (200, 128) and (planted >= 0).sum() == 60
    norms = np.linalg.norm(T.astype(np.float64), axis=1)
    assert 400 < norms.mean() < 620                             # ~512 like OpenCV SIFT


@pytest.mark.parametrize("size,cell,margin", [((500, 333), (50, 50), 25), ((611, 389), (64, 48), 0),
                                              ((800, 640), (75, 75), 40), ((123, 77), (50, 50), 25),
                                              ((200, 150), (10, 12), 40), ((90, 90), (100, 100), 30)])
def test_vectorised_cell_packing_equals_cell_by_cell(size, cell, margin):
    import time
    _, t = synth.image_pair(size, 3000, seed=size[0])
    t["positions"][:5] = [[0, 0], [size[0] - 1e-9, size[1] - 1e-9], [cell[0], cell[1]], [cell[0] + margin, 0.5], [25.0, 25.0]]
    fi = cache.Feature_Image(t["size"], t["positions"], t["descriptors"])
    fast = cache.Grid_Cache(fi, cell, fi, margin=margin).pack_cells()
    slow_grid = cache.Grid_Cache(fi, cell, lambda _c, bounds: fi(None, bounds), margin=margin)
    slow_grid.fun.wants_bounds = True
    slow = slow_grid.pack_cells()                               # visits every cell (no pack_all on a lambda)
    for a, b in zip(fast, slow):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    # keypoints ON the crop bounds (multiples of the cell size +- the margin, the image's last pixel): the planner's
    # half-open intervals are the per-cell path's
    xs = sorted({v for i in range(0, size[0] // cell[0] + 2) for v in (i * cell[0] - margin, i * cell[0], i * cell[0] + margin)
                 if 0 <= v < size[0]} | {size[0] - 1})
    ys = sorted({v for i in range(0, size[1] // cell[1] + 2) for v in (i * cell[1] - margin, i * cell[1], i * cell[1] + margin)
                 if 0 <= v < size[1]} | {size[1] - 1})
    pos = np.array([[x, y] for x in xs for y in ys], dtype=np.float64)
    desc = (np.arange(len(pos))[:, None] % 251 + np.arange(128)[None, :] % 5).astype(np.uint8)
    fi2 = cache.Feature_Image(size, pos, desc)
    fast2 = cache.Grid_Cache(fi2, cell, fi2, margin=margin).pack_cells()
    g2 = cache.Grid_Cache(fi2, cell, lambda _c, bounds: fi2(None, bounds), margin=margin)
    g2.fun.wants_bounds = True
    for a, b in zip(fast2, g2.pack_cells()):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    src_row, t_pos, cell_off = fi2.pack_plan(g2)
    assert src_row.dtype == np.int32 and np.array_equal(desc[src_row], fast2[0]) and cell_off[-1] == len(src_row)


def test_cell_packing_rejects_bad_geometry_and_ignores_stray_keypoints():
    from fastmatch_amd import _ffi
    pos = np.array([[10.0, 10.0], [-3.0, 5.0], [5.0, 1e12], [np.nan, 4.0], [99.999, 49.0]])
    src_row, t_pos, cell_off = _ffi.grid_pack_cells(pos, 100, 50, 50, 50, 3, 2, 25)
    assert sorted(set(src_row.tolist())) == [0, 4]              # outside / NaN keypoints are in no cell
    assert cell_off.shape == (7,) and cell_off[-1] == len(src_row) == len(t_pos)
    with pytest.raises(_ffi.FastMatchHipError):
        _ffi.grid_pack_cells(pos, 100, 50, 0, 50, 3, 2, 25)
    with pytest.raises(_ffi.FastMatchHipError):
        _ffi.grid_pack_cells(pos, 100, 50, 2, 2, 51, 26, 400)


def test_homography_and_planted_scorers(tmp_path):
    from fastmatch_amd import evaluate
    H = np.array([[0.9, 0.1, 12.0], [-0.05, 1.1, -7.0], [1e-4, -2e-5, 1.0]])
    f = tmp_path / "H1to4p"
    f.write_text("\n".join("  ".join("%.7e" % v for v in r) for r in H) + "\n")
    assert np.allclose(evaluate.load_homography(str(f)), H)
    rng = np.random.default_rng(1)
    src = rng.uniform(0, 600, (50, 2))
    h = np.concatenate([src, np.ones((50, 1))], axis=1) @ H.T
    dst = h[:, :2] / h[:, 2:3]
    dst[::2] += 20.0                                            # every other match is wrong
    pos = np.stack([src, dst], axis=1)
    good = evaluate.homography_scorer(H, 3.0)(np.arange(50), pos, np.zeros(50))
    assert good.tolist() == [i % 2 == 1 for i in range(50)]
    # query on the destination side: positions swapped, inverse mapping
    good2 = evaluate.homography_scorer(H, 3.0, query_is_source=False)(np.arange(50), pos[:, ::-1, :], np.zeros(50))
    assert good2.tolist() == good.tolist()
    planted = np.array([3, -1, 0])
    tpos = np.array([[1.0, 2.0], [5.0, 5.0], [9.0, 9.0], [7.0, 8.0]])
    sc = evaluate.planted_scorer(planted, tpos)
    p = np.array([[[0, 0], [7.0, 8.0]], [[0, 0], [5.0, 5.0]], [[0, 0], [9.0, 9.0]]])
    assert sc(np.array([0, 1, 2]), p, None).tolist() == [True, False, False]
    assert sc(np.array([], dtype=int), np.zeros((0, 2, 2)), None).shape == (0,)


def test_launch_plan_keeps_new_run_states_inside_the_memory_budget():
    """fastmatch._launch_plan: pairs x thresholds in one launch create one run state per run; the plan cuts
    the list where the NEW states would exceed the budget, keeps the order, and counts states that already exist as free."""
    from fastmatch_amd import fastmatch

    class Ex(object):
        def __init__(self, state_bytes, slots):
            self.b, self.k = state_bytes, slots

        def info(self):
            return self.b, self.k

    class Ctx(object):
        def __init__(self, free):
            self.free = free

        def mem_info(self):
            return self.free, 10 * self.free

    a, b, c = Ex(100, 1), Ex(100, 1), Ex(100, 1)
    runs = [a] * 5 + [b] * 5 + [c] * 5
    # budget = 0.5 * free; 15 runs need 12 new states of 100
    assert fastmatch._launch_plan(Ctx(10 ** 9), runs) == [list(range(15))]
    plan = fastmatch._launch_plan(Ctx(800), runs)                  # 400 of budget: 4 new states per launch
    assert [i for l in plan for i in l] == list(range(15)) and len(plan) > 1
    for l in plan:
        assert len(l) >= 1
    # states that exist are free: an expander with 5 slots adds nothing
    assert fastmatch._launch_plan(Ctx(2), [Ex(100, 5)] * 5) == [list(range(5))]
    # a single run that does not fit still gets a launch of its own (the launch then reports FM_ENOMEM -> host loop)
    assert fastmatch._launch_plan(Ctx(2), [Ex(100, 0), Ex(100, 0)]) == [[0], [1]]


def test_cell_planner_against_the_reference_crop_bounds():
    """fm_grid_pack_cells against golden vectors from the reference's own Grid_Cache (bak/cache.py `cache()` bounds,
    tests/golden/grid_golden.json): a keypoint is in a cell's packed rows exactly when it lies inside the crop the reference
    hands its caching function -- [x_min, x_max) x [y_min, y_max) -- probed at the corners, just inside and just outside."""
    import json
    import os
    from fastmatch_amd import _ffi
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grid_golden.json")))
    checked = 0
    for case in g["cases"]:
        (w, h), (cw, ch), margin, rows, cols = case["size"], case["cell_size"], case["margin"], case["rows"], case["cols"]
        if cw != int(cw) or ch != int(ch):
            continue                                             # (Grid_Cache truncates the cell size: integer cells only here)
        pts, want = [], []
        e = 1e-7
        for col, row, x0, x1, y0, y1 in case["bounds"]:
            cell = col * rows + row
            for x, y, inside in ((x0, y0, True), (x1 - e, y1 - e, True), (x0, y1 - e, True), (x1 - e, y0, True),
                                 (x1, y0, False), (x0, y1, False), (x0 - e, y0, False), (x0, y0 - e, False)):
                if x1 > x0 and y1 > y0:
                    pts.append((x, y)); want.append((cell, inside))
        pos = np.array(pts, dtype=np.float64)
        src_row, t_pos, cell_off = _ffi.grid_pack_cells(pos, w, h, int(cw), int(ch), rows, cols, margin)
        cells_of = {}
        for cell in range(rows * cols):
            for p in src_row[cell_off[cell]:cell_off[cell + 1]]:
                cells_of.setdefault(int(p), set()).add(cell)
        for i, (cell, inside) in enumerate(want):
            assert (cell in cells_of.get(i, set())) == inside, (case["size"], case["cell_size"], margin, pts[i], cell, inside)
            checked += 1
    assert checked > 500


def test_cell_planner_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The library's host-only code (api_grid.hip) built for the HOST with -fsanitize=address,undefined and driven by
    tests/tools/planner_sanitize.hip: 4000 random geometries, keypoints incl. NaN / infinite / 1e300 coordinates, output
    buffers one row short -- no report, and every packed row inside its cell's crop by an independent formula."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "planner_sanitize")
    csrc = os.path.join(root, "fast-match_amd", "csrc")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-g", "--cuda-host-only", "--offload-arch=gfx950",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", csrc, "-I", os.path.join(root, "include"),
                           os.path.join(csrc, "api_grid.hip"), os.path.join(root, "tests", "tools", "planner_sanitize.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), p.stdout + p.stderr


def test_bench_cpu_baselines_follow_the_containers_cpu_quota(monkeypatch):
    """bench.py's CPU baselines run on the CPUs the container is GRANTED (r06: the GPU boxes show 256 logical CPUs and
    grant 16 through cpu.max; 128 threads under that quota were throttled and the line said "cores: 128")."""
    import builtins
    import io
    import bench
    real_open = builtins.open

    def fake(files):
        def _open(path, *a, **k):
            if str(path).startswith("/sys/fs/cgroup/"):
                if path in files:
                    return io.StringIO(files[path])
                raise FileNotFoundError(path)
            return real_open(path, *a, **k)
        return _open
    import oracle
    limit = min(len(__import__("os").sched_getaffinity(0)), oracle.max_threads())
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    n, quota = bench.host_cpu_allowance()
    assert quota == 16.0 and n == min(limit, 16)
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert bench.host_cpu_allowance() == (limit, None)
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "150000 100000\n"}))          # 1.5 CPUs: one thread
    assert bench.host_cpu_allowance() == (1, 1.5)
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "400000\n",    # cgroup v1
                                                "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    n, quota = bench.host_cpu_allowance()
    assert quota == 4.0 and n == min(limit, 4)
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n",
                                                "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.host_cpu_allowance() == (limit, None)
    monkeypatch.setattr(builtins, "open", fake({}))
    assert bench.host_cpu_allowance() == (limit, None)
