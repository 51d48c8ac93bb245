"""CPU: Grid_Cache geometry of the product (fastmatch_amd.cache.Grid_Cache) and of the
oracle (oracle.fastmatch_oracle.OGrid) against golden vectors generated from the
reference's own bak/cache.py (tests/golden/make_grid_golden.py)."""
import json
import os

import numpy as np
import pytest

from fastmatch_amd.cache import Grid_Cache
from oracle.fastmatch_oracle import OGrid

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grid_golden.json")))["cases"]


class _Shape(object):                      # image stand-in: only .shape is needed for geometry
    def __init__(self, w, h):
        self.shape = (h, w, 3)


@pytest.mark.parametrize("case", GOLD, ids=lambda c: "%dx%d_c%s_m%d" % (c["size"][0], c["size"][1], c["cell_size"][0], c["margin"]))
def test_grid_geometry_matches_reference(case):
    w, h = case["size"]
    g = Grid_Cache(_Shape(w, h), tuple(case["cell_size"]), None, case["margin"])
    o = OGrid((w, h), tuple(case["cell_size"]), case["margin"])
    assert (g.rows, g.cols) == (case["rows"], case["cols"]) == (o.rows, o.cols)
    for x, y, col, row in case["block"]:
        assert tuple(g.block(x, y)) == (col, row) == tuple(o.block(x, y))
    for x, y, ox, oy in case["offset"]:
        assert tuple(g.offset(x, y)) == (ox, oy) == tuple(o.offset(x, y))
    for col, row, cx, cy in case["center"]:
        assert g.center(col, row).tolist() == [cx, cy] == list(o.center(col, row))
    for col, row, x0, x1, y0, y1 in case["bounds"]:
        assert g.cell_bounds(col, row) == ((x0, x1), (y0, y1)) == o.bounds(col, row)
    for col, row, px, py, nx, ny in case["neighbor"]:
        assert g.get_neighbor(col, row, px, py).tolist() == [nx, ny] == list(o.neighbor(col, row, px, py))


def test_neighbor_truncates_position_like_cache_pyx():
    # cache.pyx:82-83 applies int() to the position (bak/cache.py does not): 124.9 -> 124
    g = Grid_Cache(_Shape(800, 640), (50, 50), None, 25)
    assert g.get_neighbor(1, 2, 124.9, 99.9).tolist() == g.get_neighbor(1, 2, 124, 99).tolist()


def test_get_bounds_check_and_lazy_cells():
    img = np.zeros((64, 100, 3), dtype=np.uint8)
    calls = []

    def fun(crop):
        calls.append(crop.shape)
        return ("kp", "ds")
    g = Grid_Cache(img, (50, 50), fun, 10)
    assert g.get(100, 64) == ("kp", "ds")          # x == w, y == h accepted ('>' not '>=')
    with pytest.raises(Exception):
        g.get(101, 0)
    assert g.last == ((90, 100), (40, 64))
    g.get(100, 64)
    assert len(calls) == 1                          # cached
    assert g.get(0, 0) == ("kp", "ds") and calls[-1] == (64, 70, 3)   # x: [0,70) no low-side margin; y: [0,70) clipped by the image
    assert g.offset(0, 0) == (-10, -10)             # offset() subtracts it anyway (quirk)
