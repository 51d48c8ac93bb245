"""Generates tests/golden/grid_golden.json from the reference's own importable
``bak/cache.py`` Grid_Cache (reference bak/cache.py:29-123).  Runs only in the build
container (needs /root/reference); the JSON it writes is the committed fixture.

cv2 / sklearn.neighbors.ball_tree / imaging / matchutil are stubbed in sys.modules: the
geometry methods exercised here touch none of them.  bak's get_neighbor takes an
un-truncated position and returns None off-grid, where the current cache.pyx truncates
with int() and returns [-1,-1] (cache.pyx:82-83,78): probes are integers and None is
recorded as [-1,-1].
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/bak/cache.py"


def load_ref():
    for name in ("cv2", "imaging", "matchutil"):
        sys.modules.setdefault(name, types.ModuleType(name))
    bt = types.ModuleType("sklearn.neighbors.ball_tree")
    bt.BallTree = object
    sys.modules["sklearn.neighbors.ball_tree"] = bt
    spec = importlib.util.spec_from_file_location("ref_bak_cache", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_ref()
    rng = np.random.default_rng(20250001)
    cases = []
    configs = [((800, 640), (50, 50), 25), ((800, 640), (75, 75), 30), ((800, 640), (75, 75), 40),
               ((1000, 1000), (50, 50), 25), ((6000, 4000), (50, 50), 25), ((640, 480), (64, 48), 0),
               ((123, 77), (50, 50), 25), ((400, 400), (50, 40), 10), ((51, 49), (50, 50), 5),
               ((1024, 768), (100, 60), 40)]
    for (w, h), cell, margin in configs:
        data = np.zeros((h, w, 3), dtype=np.uint8)
        g = ref.Grid_Cache(data, cell, None, margin)
        case = {"size": [w, h], "cell_size": list(cell), "margin": margin,
                "rows": g.rows, "cols": g.cols, "block": [], "offset": [], "center": [],
                "bounds": [], "neighbor": []}
        # positions: random floats for block/offset, including edges
        pts = [(0.0, 0.0), (w - 1.0, h - 1.0), (float(w), float(h)), (cell[0] * 1.0, cell[1] * 1.0),
               (cell[0] - 0.001, cell[1] - 0.001)]
        pts += [(float(x), float(y)) for x, y in zip(rng.uniform(0, w, 40), rng.uniform(0, h, 40))]
        for (x, y) in pts:
            col, row = g.block((x, y))
            case["block"].append([x, y, col, row])
            ox, oy = g.offset((x, y))
            case["offset"].append([x, y, ox, oy])
        cells = [(0, 0), (g.cols - 1, g.rows - 1), (0, g.rows - 1), (g.cols - 1, 0)]
        cells += [(int(c), int(r)) for c, r in zip(rng.integers(0, g.cols, 12), rng.integers(0, g.rows, 12))]
        for (col, row) in cells:
            cx, cy = g.center(col, row)
            case["center"].append([col, row, int(cx), int(cy)])
            (x0, x1), (y0, y1) = g.cache(col, row)
            case["bounds"].append([col, row, int(x0), int(x1), int(y0), int(y1)])
            # neighbour probes: integer positions around the cell, incl. exact diagonals
            cx, cy = int(cx), int(cy)
            probes = [(cx, cy), (cx + 7, cy + 7), (cx - 7, cy - 7), (cx + 7, cy - 7), (cx - 7, cy + 7),
                      (cx + 9, cy), (cx - 9, cy), (cx, cy + 9), (cx, cy - 9), (cx + 1, cy), (cx, cy + 1)]
            probes += [(cx + int(dx), cy + int(dy)) for dx, dy in
                       zip(rng.integers(-cell[0], cell[0] + 1, 8), rng.integers(-cell[1], cell[1] + 1, 8))]
            for (px, py) in probes:
                n = g.get_neighbor((col, row), (px, py))
                n = [-1, -1] if n is None else [int(n[0]), int(n[1])]
                case["neighbor"].append([col, row, px, py, n[0], n[1]])
        cases.append(case)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "grid_golden.json")
    with open(out, "w") as f:
        json.dump({"source": "reference bak/cache.py Grid_Cache (lines 29-123)", "cases": cases}, f)
    print("wrote", out, sum(len(c["neighbor"]) for c in cases), "neighbor vectors")


if __name__ == "__main__":
    main()
