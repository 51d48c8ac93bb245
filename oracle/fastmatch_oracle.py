"""CPU restatement of the Fast-Match algorithm around the matcher (TEST INFRASTRUCTURE ONLY;
see oracle/__init__.py -- PARITY UNPINNED for the matcher itself).

Restates, with brute-force NumPy and the C oracle matcher, what the reference does in
  fastmatch.pyx:32-53   match / get_matches          -> ``o_match``
  fastmatch.pyx:56-89   do_iter                      -> ``o_do_iter``
  fastmatch.pyx:92-103  get_neighbors                -> inside ``o_do_iter``
  fastmatch.pyx:107-141 match_thumbs                 -> ``o_match_thumbs``
  fastmatch.pyx:145-169 match_position               -> ``o_match_position``
  cache.pyx:33-138      Grid_Cache geometry + cells  -> ``OGrid``
  cache.pyx:173-188     Metric_Cache.get             -> ``OQuery.get``
It deliberately shares no code with the product package: radius queries and cell
look-ups are O(N) scans, the expansion loop is a plain list used as a stack front.

Conventions that the reference leaves to its un-pinned dependencies and that the product
defines (both sides follow them; DESIGN.md lists them):
  * radius query order = ascending (dx^2 + dy^2 in float64, index), boundary inclusive;
  * keypoints of a pre-extracted cell = points with x_min <= x < x_max, y_min <= y < y_max
    in ascending index, crop-local coordinates;
  * seeds sorted by ratio with a stable sort.
The Grid_Cache geometry is pinned by golden vectors generated from the reference's own
importable ``bak/cache.py`` (tests/golden/grid_golden.json).
"""
import numpy as np

import oracle


# float32 accumulation order for descriptors that are not integer valued (ignored otherwise):
# 0 = OpenCV's unrolled-by-4 order, 1 = the fixed fma chain of the device's float32 route
FLOAT_ORDER = 0


class OGrid(object):
    """Grid geometry and lazily filled cells over a pre-extracted target."""

    def __init__(self, size, cell_size, margin, positions=None, descriptors=None, image=None, fun=None):
        self.w, self.h = int(size[0]), int(size[1])
        self.cw, self.ch = int(cell_size[0]), int(cell_size[1])
        self.rows = int(self.w / cell_size[0]) + 1          # cells along x  (cache.pyx:41)
        self.cols = int(self.h / cell_size[1]) + 1          # cells along y  (cache.pyx:42)
        self.margin = int(margin)
        self.positions = positions
        self.descriptors = descriptors
        self.image, self.fun = image, fun                   # pixel target: cell = fun(crop), cache.pyx:132-137
        self.cells = {}
        self.last = None

    def block(self, x, y):                                   # cache.pyx:95-99
        return int(y / self.ch), int(x / self.cw)

    def offset(self, x, y):                                  # cache.pyx:64-69
        col, row = self.block(x, y)
        return row * self.cw - self.margin, col * self.ch - self.margin

    def center(self, col, row):                              # cache.pyx:116-121
        x = int((row + 0.5) * self.cw)
        y = int((col + 0.5) * self.ch)
        return min(x, self.w - 1), min(y, self.h - 1)

    def bounds(self, col, row):                              # cache.pyx:128-131
        x_min = row * self.cw - (self.margin if row > 0 else 0)
        x_max = (x_min + self.cw + 2 * self.margin) if row + 1 < self.rows else self.w
        y_min = col * self.ch - (self.margin if col > 0 else 0)
        y_max = (y_min + self.ch + 2 * self.margin) if col + 1 < self.cols else self.h
        return (x_min, x_max), (y_min, y_max)

    def neighbor(self, col, row, px, py):                    # cache.pyx:72-92
        cx, cy = self.center(col, row)
        xd = int(px) - cx
        yd = int(py) - cy
        if yd < xd and yd < -xd:
            return self.center(col - 1, row) if col - 1 >= 0 else (-1, -1)
        elif xd > yd:
            return self.center(col, row + 1) if row + 1 < self.rows else (-1, -1)
        elif yd > -xd:
            return self.center(col + 1, row) if col + 1 < self.cols else (-1, -1)
        else:
            return self.center(col, row - 1) if row - 1 >= 0 else (-1, -1)

    def get(self, x, y):                                     # cache.pyx:51-61, 102-106
        if x > self.w or y > self.h:
            raise Exception("(%i,%i) is outside data bounds of (%i,%i)" % (x, y, self.w, self.h))
        col, row = self.block(x, y)
        if (col, row) not in self.cells:
            b = self.bounds(col, row)
            (x0, x1), (y0, y1) = b
            if self.fun is not None:                         # lazy features of the crop (cache.pyx:132-137)
                kp, ds = self.fun(self.image[y0:y1, x0:x1, :])
                pts = np.array([k.pt for k in kp], dtype=np.float64).reshape(-1, 2)
                self.cells[(col, row)] = (pts, ds if ds is not None and len(ds) else None)
                self.last = b
                return self.cells[(col, row)]
            p = self.positions
            sel = np.nonzero((p[:, 0] >= x0) & (p[:, 0] < x1) & (p[:, 1] >= y0) & (p[:, 1] < y1))[0]
            self.cells[(col, row)] = (p[sel] - np.array([x0, y0], dtype=np.float64),
                                      self.descriptors[sel] if len(sel) else None)
            self.last = b
        return self.cells[(col, row)]


class OQuery(object):
    """Query bank: descriptors, positions, self distances (+ thumbnail bank)."""

    def __init__(self, descriptors, positions, size, distances=None, thumb=None, metric="minkowski"):
        # BallTree(positions, metric = metric), cache.pyx:276: "minkowski" (p = 2) / "euclidean",
        # "manhattan", "chebyshev"
        self.metric = {"minkowski": 2, "euclidean": 2, "l2": 2, "manhattan": 1, "cityblock": 1, "l1": 1,
                       "chebyshev": 0, "infinity": 0}[metric]
        self.descriptors = descriptors
        self.positions = np.asarray(positions, dtype=np.float64).reshape(-1, 2)
        self.size = size
        # exact self 2-NN, r[1].distance (cache.pyx:250-252 / 271-273 made exact)
        self.distances = oracle.self_dist(descriptors, order=FLOAT_ORDER) if distances is None else np.asarray(distances, np.float64)
        self.thumb = thumb
        if thumb is not None and "distances" not in thumb:
            thumb["distances"] = oracle.self_dist(thumb["descriptors"], order=FLOAT_ORDER)

    def get(self, x, y, radius):                             # cache.pyx:173-188
        x, y, radius = int(x), int(y), int(radius)
        dx = self.positions[:, 0] - float(x)
        dy = self.positions[:, 1] - float(y)
        if self.metric == 2:
            d2 = dx * dx + dy * dy
            idx = np.nonzero(d2 <= float(radius) * float(radius))[0]
        else:
            d2 = np.abs(dx) + np.abs(dy) if self.metric == 1 else np.maximum(np.abs(dx), np.abs(dy))
            idx = np.nonzero(d2 <= float(radius))[0]
        idx = idx[np.lexsort((idx, d2[idx]))]
        return self.descriptors[idx], self.positions[idx], self.distances[idx], idx


def o_match_position(pos, query, grid, radius=100):          # fastmatch.pyx:145-169
    qx, qy = int(pos[0][0]), int(pos[0][1])
    tx, ty = int(pos[1][0]), int(pos[1][1])
    q_ds, q_pos, q_dis, q_idx = query.get(qx, qy, radius)
    t_kp, t_ds = grid.get(tx, ty)
    if t_ds is None:
        return np.array([]), np.array([]), np.array([])
    ox, oy = grid.offset(tx, ty)
    t_pos = t_kp + np.array([ox, oy], dtype=np.float64)
    if len(q_idx) == 0:                                      # knnMatch of an empty query set: matches = [] (next case)
        return np.array([]), np.array([]), np.array([])
    tidx, dist = oracle.bf_xcheck1(q_ds, t_ds, order=FLOAT_ORDER)
    positions, ratios, indices = [], [], []
    for qi in range(len(tidx)):                              # knnMatch order = query order
        if tidx[qi] < 0:
            continue                                         # empty inner list (fastmatch.pyx:162)
        ratios.append(np.float64(dist[qi]) / q_dis[qi])      # fastmatch.pyx:165
        positions.append((q_pos[qi], t_pos[tidx[qi]]))
        indices.append(q_idx[qi])
    if not ratios:                                           # numpy.array([...]) of an empty list: shape (0,), float64
        return np.array([]), np.array([]), np.array([])      # (fastmatch.pyx:165-167)
    return np.array(positions), np.array(ratios), np.array(indices)


def o_do_iter(seeds, query, grid, tau, radius=100, log=None, max_rounds=None):   # fastmatch.pyx:56-89
    """``max_rounds`` (test aid, not in the reference): stop after that many rounds; the match
    list returned is then a prefix of the full run's list (matches are appended in round order)."""
    todo = [s for s in seeds]                                # front of the list = next
    matches, seen_keys, found = [], {}, {}
    rounds = 0
    while todo:
        if max_rounds is not None and rounds >= max_rounds:
            break
        query_pos, target_pos = todo.pop(0)
        col, row = grid.block(target_pos[0], target_pos[1])
        qcol, qrow = grid.block(query_pos[0], query_pos[1])
        if seen_keys.get((col, row, qcol, qrow), False):
            continue
        seen_keys[(col, row, qcol, qrow)] = True
        result_pos, ratios, query_idx = o_match_position((query_pos, target_pos), query, grid, radius)
        rounds += 1
        keep = ratios < tau
        new = []
        for p_query, p_target in result_pos[keep]:           # get_neighbors, fastmatch.pyx:92-103
            n = grid.neighbor(col, row, p_target[0], p_target[1])
            if n[0] != -1:
                new.append(np.array((p_query, n), dtype=np.float64))
        todo = new + todo                                    # prepend = depth first (:76-77)
        if log is not None:
            log.append({"query_pos": query_pos, "target_pos": target_pos, "target_grid": grid.last,
                        "matches": result_pos[keep], "radius": radius, "ratios": ratios[keep],
                        "margin": grid.margin})
        for p, r, index in zip(result_pos[keep], ratios[keep], query_idx[keep]):
            tup = [int(p[0, 0]), int(p[0, 1]), int(p[1, 0]), int(p[1, 1])]
            if tup not in found.get(float(r), []):           # dedup keyed on the ratio (:83-86)
                found[float(r)] = found.get(float(r), []) + [tup]
                matches.append((int(index), {"positions": p, "ratio": float(r)}))
    return matches, rounds


def o_match_thumbs(query, target_thumb, target_size):        # fastmatch.pyx:107-141
    q = query.thumb
    tidx, dist = oracle.bf_xcheck1(q["descriptors"], target_thumb["descriptors"], order=FLOAT_ORDER)
    m = np.nonzero(tidx >= 0)[0]
    ratios = dist[m].astype(np.float64) / q["distances"][m]
    t_pos = np.asarray(target_thumb["positions"], np.float64)[tidx[m]]
    q_pos = np.asarray(q["positions"], np.float64)[m]
    t_ratio = np.array([target_size[0] / float(target_thumb["size"][0]),
                        target_size[1] / float(target_thumb["size"][1])])
    q_ratio = np.array([query.size[0] / float(q["size"][0]), query.size[1] / float(q["size"][1])])
    pos = np.array([(qp * q_ratio, tp * t_ratio) for qp, tp in zip(q_pos, t_pos)]).reshape(-1, 2, 2)
    order = np.argsort(ratios, kind="stable")
    return pos[order], ratios[order]


def o_match(query, target, options={}):                      # fastmatch.pyx:32-53
    """``target`` = dict(size, positions, descriptors, thumb=dict(positions, descriptors, size)), or
    dict(size, image, feature_function, thumb=...) for a pixel target whose cells are computed lazily."""
    grid_size = options.get("grid_size", (50, 50))
    thumb_strategy = options.get("thumb_strategy", lambda n: n)
    log = options.get("log", None)
    if "image" in target:        # pixel target + feature function (the reference's own mode, fastmatch.pyx:45)
        grid = OGrid(target["size"], grid_size, options.get("grid_margin", 25), image=target["image"],
                     fun=target["feature_function"])
    else:
        grid = OGrid(target["size"], grid_size, options.get("grid_margin", 25),
                     np.asarray(target["positions"], np.float64).reshape(-1, 2), target["descriptors"])
    radius = options.get("radius", 100)
    thumb_pos, thumb_ratios = o_match_thumbs(query, target["thumb"], target["size"])

    def get_matches(tau):
        seeds = thumb_pos[thumb_ratios < thumb_strategy(tau)]
        matches, rounds = o_do_iter(seeds, query, grid, tau, radius, log, options.get("max_rounds"))
        get_matches.rounds = rounds
        return matches

    return get_matches
