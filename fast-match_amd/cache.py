"""Descriptor banks of Fast-Match with the reference's class surface, device resident.

Mirrors ``cache.pyx`` / ``cache.pxd`` of the reference:

* ``Grid_Cache``   (reference ``cache.pyx:31-138``) -- lazy grid of target-image cells;
  each cell holds ``caching_function(crop)`` = (keypoints, descriptors).  The integer
  geometry (block / offset / center / crop bounds / get_neighbor) reproduces the
  reference including its quirks.  Each computed cell's
  descriptors are uploaded once as a device bank (``cell_bank``).
* ``Metric_Cache`` (reference ``cache.pyx:151-284``) -- query-image bank: descriptors,
  positions, self 2-NN distances, a position index for radius queries, thumbnail bank;
  ``save``/``load`` in the reference's npz layout.  ``from_arrays`` builds one from
  pre-extracted features (no SIFT / OpenCV needed); self distances are computed by the
  HIP 2-NN kernel (exact, in place of the reference's approximate FLANN at
  ``cache.pyx:271``).

Additions that do not exist in the reference: ``Feature_Image`` (a pre-extracted target
image standing in for the uint8 array), ``Position_Index`` (deterministic uniform-grid
replacement for sklearn's BallTree: ascending (squared distance, index), ``<= r``
inclusive), ``radius_indices`` and the ``bank`` properties.
"""
import hashlib
import os
import pickle
import struct

import numpy as np

from . import matchutil


# ---------------------------------------------------------------------------------------
# Position index (replaces sklearn BallTree.query_radius, cache.pyx:180-186,276)
# ---------------------------------------------------------------------------------------
METRIC_L2, METRIC_L1, METRIC_LINF = 0, 1, 2
_METRICS = {"minkowski": METRIC_L2, "euclidean": METRIC_L2, "l2": METRIC_L2,
            "manhattan": METRIC_L1, "cityblock": METRIC_L1, "l1": METRIC_L1,
            "chebyshev": METRIC_LINF, "infinity": METRIC_LINF}
_METRIC_NAMES = {METRIC_L2: "euclidean", METRIC_L1: "manhattan", METRIC_LINF: "chebyshev"}   # as sklearn knows them


def metric_code(metric, p=None):
    """The reference hands ``options["metric"]`` to ``BallTree(positions, metric=metric)``
    (cache.pyx:160, 276; default "minkowski", i.e. p = 2).  Served here: the Minkowski family a 2-D
    keypoint index has a use for -- p = 2 ("minkowski", "euclidean", "l2"), p = 1 ("manhattan",
    "cityblock", "l1"), p = inf ("chebyshev", "infinity"); "minkowski" with ``p`` in {1, 2, inf}
    likewise.  Anything else raises instead of silently answering in another metric."""
    name = str(metric).lower()
    if name not in _METRICS:
        raise ValueError("metric %r is not supported by the position index (euclidean / manhattan / chebyshev)" % (metric,))
    code = _METRICS[name]
    if name == "minkowski" and p is not None:
        if p == 1:
            code = METRIC_L1
        elif p == 2:
            code = METRIC_L2
        elif p == float("inf"):
            code = METRIC_LINF
        else:
            raise ValueError("minkowski p = %r is not supported by the position index (1, 2 or inf)" % (p,))
    return code


class Position_Index(object):
    """Uniform-grid radius query over 2-D keypoint positions.

    Result order is ascending ``(distance, index)`` in float64 and a point at distance exactly
    r is included -- sklearn's BallTree also includes it but leaves the order among equal
    distances unspecified, so this order is OUR definition.
    ``metric``: see ``metric_code``; distances are computed the way sklearn's DistanceMetric
    classes do (p = 2: dx*dx + dy*dy then sqrt; p = 1: |dx| + |dy|; p = inf: max(|dx|, |dy|))."""

    def __init__(self, positions, bucket=64.0, metric="minkowski", p=None):
        self.positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 2)
        self.bucket = float(bucket)
        self.metric = metric_code(metric, p)
        n = self.positions.shape[0]
        if n:
            self.x0 = float(np.floor(self.positions[:, 0].min()))
            self.y0 = float(np.floor(self.positions[:, 1].min()))
            bx = np.floor((self.positions[:, 0] - self.x0) / self.bucket).astype(np.int64)
            by = np.floor((self.positions[:, 1] - self.y0) / self.bucket).astype(np.int64)
            self.nbx = int(bx.max()) + 1
            self.nby = int(by.max()) + 1
            key = by * self.nbx + bx
            self.order = np.argsort(key, kind="stable").astype(np.int64)
            self.start = np.searchsorted(key[self.order], np.arange(self.nbx * self.nby + 1))
        else:
            self.x0 = self.y0 = 0.0
            self.nbx = self.nby = 0
            self.order = np.zeros(0, dtype=np.int64)
            self.start = np.zeros(1, dtype=np.int64)

    def key_to_distance(self, key):
        """Distances from the sort keys ``radius`` returns (p = 2: the key is the SQUARED distance)."""
        return np.sqrt(key) if self.metric == METRIC_L2 else key

    def radius(self, x, y, r):
        """(idx int64[m], key float64[m]) of the points within r, sorted by (key, idx); key = squared
        distance for p = 2 (d2 <= r*r), the distance itself for p = 1 / inf (d <= r)."""
        if self.positions.shape[0] == 0 or r < 0:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float64)
        b = self.bucket
        bx0 = max(int(np.floor((x - r - self.x0) / b)), 0)
        bx1 = min(int(np.floor((x + r - self.x0) / b)), self.nbx - 1)
        by0 = max(int(np.floor((y - r - self.y0) / b)), 0)
        by1 = min(int(np.floor((y + r - self.y0) / b)), self.nby - 1)
        if bx1 < bx0 or by1 < by0:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float64)
        parts = []
        for by in range(by0, by1 + 1):
            s = self.start[by * self.nbx + bx0]
            e = self.start[by * self.nbx + bx1 + 1]
            if e > s:
                parts.append(self.order[s:e])
        if not parts:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float64)
        cand = np.concatenate(parts) if len(parts) > 1 else parts[0]
        dx = self.positions[cand, 0] - float(x)
        dy = self.positions[cand, 1] - float(y)
        if self.metric == METRIC_L2:
            d2 = dx * dx + dy * dy
            keep = d2 <= float(r) * float(r)
        else:
            d2 = np.abs(dx) + np.abs(dy) if self.metric == METRIC_L1 else np.maximum(np.abs(dx), np.abs(dy))
            keep = d2 <= float(r)
        cand = cand[keep]
        d2 = d2[keep]
        o = np.lexsort((cand, d2))
        return cand[o], d2[o]

    def query_radius(self, X, r, return_distance=False, sort_results=False):
        """Subset of sklearn's BallTree.query_radius used by the reference
        (cache.pyx:182-185): one query point, object arrays of per-point results."""
        X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
        inds = np.empty(X.shape[0], dtype=object)
        dists = np.empty(X.shape[0], dtype=object)
        for i in range(X.shape[0]):
            idx, d2 = self.radius(X[i, 0], X[i, 1], r)
            if not sort_results:
                o = np.argsort(idx, kind="stable")
                idx, d2 = idx[o], d2[o]
            inds[i] = idx
            dists[i] = self.key_to_distance(d2)
        if return_distance:
            return inds, dists
        return inds


# ---------------------------------------------------------------------------------------
# Pre-extracted target image
# ---------------------------------------------------------------------------------------
class Feature_Image(object):
    """A target image given as pre-extracted features instead of pixels.

    Stands in for the ``uint8[H, W, 3]`` array the reference passes to ``match`` and
    ``Grid_Cache`` when SIFT is not available (or already done elsewhere): ``shape``
    looks like the image's, ``features_in`` plays the role of SIFT on a crop (keypoints
    in crop-local coordinates, ascending original index), ``thumb`` holds the thumbnail
    features ``match_thumbs`` would extract (fastmatch.pyx:113-115)."""

    wants_bounds = True

    def __init__(self, size, positions, descriptors, thumb_positions=None, thumb_descriptors=None,
                 thumb_size=None):
        self.size = (int(size[0]), int(size[1]))                # (width, height)
        self.shape = (self.size[1], self.size[0], 3)
        self.positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 2)
        d = np.asarray(descriptors)
        self.descriptors = np.ascontiguousarray(d if d.dtype == np.uint8 else d.astype(np.float32))
        if self.descriptors.shape[0] != self.positions.shape[0]:
            raise ValueError("positions and descriptors disagree on the number of keypoints")
        self.thumb = None
        if thumb_descriptors is not None:
            td = np.asarray(thumb_descriptors)
            self.thumb = {
                "positions": np.ascontiguousarray(thumb_positions, dtype=np.float64).reshape(-1, 2),
                "descriptors": np.ascontiguousarray(td if td.dtype == np.uint8 else td.astype(np.float32)),
                "size": (int(thumb_size[0]), int(thumb_size[1])),
            }
        self._index = None                 # built at the first per-cell query (the device path plans all cells at once)

    def features_in(self, x_min, x_max, y_min, y_max):
        """Keypoints with x_min <= x < x_max and y_min <= y < y_max, crop-local coords."""
        cx, cy = 0.5 * (x_min + x_max), 0.5 * (y_min + y_max)
        rad = 0.5 * float(np.hypot(x_max - x_min, y_max - y_min)) + 1.0
        if self._index is None:
            self._index = Position_Index(self.positions)
        idx, _ = self._index.radius(cx, cy, rad)
        if idx.shape[0]:
            p = self.positions[idx]
            keep = (p[:, 0] >= x_min) & (p[:, 0] < x_max) & (p[:, 1] >= y_min) & (p[:, 1] < y_max)
            idx = np.sort(idx[keep])
        if idx.shape[0] == 0:
            return np.zeros((0, 2), dtype=np.float64), None
        local = self.positions[idx] - np.array([x_min, y_min], dtype=np.float64)
        return local, self.descriptors[idx]

    def __call__(self, data_cell, bounds):
        (x_min, x_max), (y_min, y_max) = bounds
        return self.features_in(x_min, x_max, y_min, y_max)

    def pack_plan(self, grid):
        """All cells of ``grid`` at once: (src_row int32[nt], positions f64[nt, 2], cell_off int64[cols * rows + 1]) --
        packed row i is keypoint src_row[i], cells one after the other (cell id = col * rows + row), a cell's keypoints in
        ascending index, positions in full-image coordinates with offset() applied as match_position does
        (fastmatch.pyx:157-158).  One pass in the library's host code (fm_grid_pack_cells; r04: the NumPy form of this
        took 6 ms per 12.5k keypoints, 200 ms at 300k -- more than the device loop it prepares)."""
        from . import _ffi
        return _ffi.grid_pack_cells(self.positions, grid.width, grid.height, grid.cell_width, grid.cell_height,
                                    grid.rows, grid.cols, grid.margin)

    def pack_all(self, grid):
        """The same (descriptors, positions, cell_off) Grid_Cache.pack_cells builds by visiting every cell (the device
        path uploads ``descriptors`` once and gathers on the device instead: fastmatch.make_expander)."""
        src_row, t_pos, cell_off = self.pack_plan(grid)
        return self.descriptors[src_row], t_pos, cell_off


def keypoint_positions(keypoints):
    """[n, 2] float64 positions from cv2.KeyPoint-like objects (``.pt``) or an array."""
    if isinstance(keypoints, np.ndarray):
        return keypoints.reshape(-1, 2).astype(np.float64, copy=False)
    if len(keypoints) == 0:
        return np.zeros((0, 2), dtype=np.float64)
    return np.array([k.pt for k in keypoints], dtype=np.float64).reshape(-1, 2)


# ---------------------------------------------------------------------------------------
# Grid cache (target descriptor bank + expansion geometry)
# ---------------------------------------------------------------------------------------
class Grid_Cache(object):

    def __init__(self, data, cell_size, caching_function=None, margin=25, options={}):
        shape = data.shape
        self.width = int(shape[1])
        self.height = int(shape[0])
        self.cell_width = int(cell_size[0])        # cdef int: truncated (cache.pxd:15-16)
        self.cell_height = int(cell_size[1])
        # NB: "rows" counts cells along x and "cols" along y (cache.pyx:41-42)
        self.rows = int(self.width / cell_size[0]) + 1
        self.cols = int(self.height / cell_size[1]) + 1
        self.data = data
        self.fun = caching_function
        self.last = None
        self.margin = int(margin)
        self.grid = {n: {} for n in range(self.cols)}
        self._banks = {}
        self._options = options

    # -- geometry (cache.pyx:64-121) -----------------------------------------------------
    def block(self, x, y):
        row = int(x / self.cell_width)
        col = int(y / self.cell_height)
        return col, row

    def offset(self, x, y):
        # always subtracts the margin, even for row/col 0 (cache.pyx:67-68 vs :128,130)
        col, row = self.block(x, y)
        return (row * self.cell_width - self.margin, col * self.cell_height - self.margin)

    def center(self, col, row):
        x = int((row + 0.5) * self.cell_width)
        y = int((col + 0.5) * self.cell_height)
        x_in = x if x < self.width - 1 else self.width - 1
        y_in = y if y < self.height - 1 else self.height - 1
        return np.array((x_in, y_in), dtype=np.int64)

    def get_neighbor(self, col, row, pos_x, pos_y):
        """Centre of the 4-neighbour cell whose border (col,row)'s point is closest to;
        [-1,-1] when that neighbour is off the grid (cache.pyx:72-92, branch order kept)."""
        none = np.array((-1, -1), dtype=np.int64)
        cx, cy = self.center(col, row)
        x_diff = int(pos_x) - int(cx)
        y_diff = int(pos_y) - int(cy)
        if y_diff < x_diff and y_diff < -x_diff:
            return self.center(col - 1, row) if col - 1 >= 0 else none
        if x_diff > y_diff:
            return self.center(col, row + 1) if row + 1 < self.rows else none
        if y_diff > -x_diff:
            return self.center(col + 1, row) if col + 1 < self.cols else none
        return self.center(col, row - 1) if row - 1 >= 0 else none

    def cell_bounds(self, col, row):
        """Crop of cell (col,row) including margins: ((x_min,x_max),(y_min,y_max))
        (cache.pyx:128-131)."""
        x_min = row * self.cell_width - (self.margin * (row > 0))
        x_max = x_min + self.cell_width + self.margin * 2 if row + 1 < self.rows else self.width
        y_min = col * self.cell_height - (self.margin * (col > 0))
        y_max = y_min + self.cell_height + self.margin * 2 if col + 1 < self.cols else self.height
        return ((x_min, x_max), (y_min, y_max))

    # -- cell store (cache.pyx:51-61,102-114,124-138) -------------------------------------
    def get(self, x, y):
        if x > self.width or y > self.height:      # '>' not '>=' as in the reference
            raise Exception("(%i,%i) is outside data bounds of (%i,%i)" % (x, y, self.width, self.height))
        col, row = self.block(x, y)
        return self.get_cell(col, row)

    def get_cell(self, col, row):
        if row not in self.grid[col]:
            self.last = self.cache(col, row)
        return self.grid[col][row]

    def is_cached(self, x, y):
        col, row = self.block(x, y)
        return row in self.grid[col]

    def cache(self, col, row):
        bounds = self.cell_bounds(col, row)
        (x_min, x_max), (y_min, y_max) = bounds
        if self.fun is None:
            self.grid[col][row] = self.data[y_min:y_max, x_min:x_max, :]
        elif getattr(self.fun, "wants_bounds", False):
            self.grid[col][row] = self.fun(None, bounds)
        else:
            self.grid[col][row] = self.fun(self.data[y_min:y_max, x_min:x_max, :])
        return bounds

    # -- device residency -----------------------------------------------------------------
    def pack_cells(self):
        """Every cell's features packed cell after cell (cell id = col * rows + row) for the
        device-resident loop: (descriptors [nt, dim], positions f64[nt, 2] in full-image
        coordinates with offset() applied as match_position does (fastmatch.pyx:157-158),
        cell_off int64[cols*rows + 1]).  Computes every cell (cache.pyx:124-138)."""
        if hasattr(self.fun, "pack_all") and self.fun is self.data:
            return self.fun.pack_all(self)
        descs, poss = [], []
        cell_off = np.zeros(self.cols * self.rows + 1, dtype=np.int64)
        dim, dtype = None, None
        for col in range(self.cols):
            for row in range(self.rows):
                kp, ds = self.get_cell(col, row)
                n = 0 if ds is None else len(ds)
                if n:
                    ds = np.asarray(ds)
                    dim, dtype = ds.shape[1], ds.dtype
                    off = np.array([row * self.cell_width - self.margin, col * self.cell_height - self.margin],
                                   dtype=np.float64)
                    descs.append(ds)
                    poss.append(keypoint_positions(kp) + off)
                cell_off[col * self.rows + row + 1] = cell_off[col * self.rows + row] + n
        if descs:
            return np.concatenate(descs), np.concatenate(poss), cell_off
        return np.zeros((0, 128), dtype=np.uint8), np.zeros((0, 2), dtype=np.float64), cell_off

    def cell_bank(self, col, row, context, float_route=False):
        """Device bank of the cell's descriptors (uploaded once), or None if it has none.
        ``float_route``: the query bank is not integer valued, so the cell must not be either."""
        key = (col, row, bool(float_route))
        if key not in self._banks:
            value = self.get_cell(col, row)
            ds = value[1] if isinstance(value, tuple) else None
            if ds is None or len(ds) == 0:
                self._banks[key] = None
            else:
                self._banks[key] = context.bank(np.asarray(ds), float_route=float_route)
        return self._banks[key]


# ---------------------------------------------------------------------------------------
# RIPEMD-160 (file naming of Metric_Cache.save/load, cache.pyx:194-196); OpenSSL 3 builds
# of hashlib often lack it, so a small pure-Python version backs it up.
# ---------------------------------------------------------------------------------------
def _ripemd160(data):
    try:
        h = hashlib.new("ripemd160")
        h.update(data)
        return h.hexdigest()
    except (ValueError, TypeError):
        pass
    rl = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5, 2, 14, 11, 8,
          3, 10, 14, 4, 9, 15, 8, 1, 2, 7, 0, 6, 13, 11, 5, 12, 1, 9, 11, 10, 0, 8, 12, 4, 13, 3, 7, 15, 14, 5, 6, 2,
          4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13]
    rr = [5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12, 6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12, 4, 9, 1, 2,
          15, 5, 1, 3, 7, 14, 6, 9, 11, 8, 12, 2, 10, 0, 4, 13, 8, 6, 4, 1, 3, 11, 15, 0, 5, 12, 2, 13, 9, 7, 10, 14,
          12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11]
    sl = [11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8, 7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9, 11, 7, 13, 12,
          11, 13, 6, 7, 14, 9, 13, 15, 14, 8, 13, 6, 5, 12, 7, 5, 11, 12, 14, 15, 14, 15, 9, 8, 9, 14, 5, 6, 8, 6, 5, 12,
          9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8, 5, 6]
    sr = [8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6, 9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7, 6, 15, 13, 11,
          9, 7, 15, 11, 8, 6, 6, 14, 12, 13, 5, 14, 13, 13, 7, 5, 15, 5, 8, 11, 14, 14, 6, 14, 6, 9, 12, 9, 12, 5, 15, 8,
          8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11, 11]
    kl = [0x00000000, 0x5A827999, 0x6ED9EBA1, 0x8F1BBCDC, 0xA953FD4E]
    kr = [0x50A28BE6, 0x5C4DD124, 0x6D703EF3, 0x7A6D76E9, 0x00000000]
    M = 0xFFFFFFFF

    def rol(x, n):
        return ((x << n) | (x >> (32 - n))) & M

    def f(j, x, y, z):
        if j < 16:
            return x ^ y ^ z
        if j < 32:
            return (x & y) | (~x & M & z)
        if j < 48:
            return (x | (~y & M)) ^ z
        if j < 64:
            return (x & z) | (y & (~z & M))
        return x ^ (y | (~z & M))

    msg = bytearray(data)
    bits = len(msg) * 8
    msg.append(0x80)
    while len(msg) % 64 != 56:
        msg.append(0)
    msg += struct.pack("<Q", bits)
    h0, h1, h2, h3, h4 = 0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0
    for off in range(0, len(msg), 64):
        X = struct.unpack("<16I", bytes(msg[off:off + 64]))
        al, bl, cl, dl, el = h0, h1, h2, h3, h4
        ar, br, cr, dr, er = h0, h1, h2, h3, h4
        for j in range(80):
            t = (rol((al + f(j, bl, cl, dl) + X[rl[j]] + kl[j // 16]) & M, sl[j]) + el) & M
            al, el, dl, cl, bl = el, dl, rol(cl, 10), bl, t
            t = (rol((ar + f(79 - j, br, cr, dr) + X[rr[j]] + kr[j // 16]) & M, sr[j]) + er) & M
            ar, er, dr, cr, br = er, dr, rol(cr, 10), br, t
        t = (h1 + cl + dr) & M
        h1 = (h2 + dl + er) & M
        h2 = (h3 + el + ar) & M
        h3 = (h4 + al + br) & M
        h4 = (h0 + bl + cr) & M
        h0 = t
    return struct.pack("<5I", h0, h1, h2, h3, h4).hex()


# ---------------------------------------------------------------------------------------
# Metric cache (query descriptor bank)
# ---------------------------------------------------------------------------------------
class Metric_Cache(object):

    def __init__(self, path, options={}):
        """ Caches an image so it's ready for matching (cache.pyx:153-170) """
        force_reload = options.get("force_reload", False)
        max_size = options.get("max_size", -1)
        metric = options.get("metric", "minkowski")
        thumb_x, thumb_y = options.get("thumb_size", (600, 600))
        self.path = path
        self.thumb = {}
        self.original = {}
        self._options = options
        self._metric = (metric, options.get("p"))
        metric_code(*self._metric)                      # an unsupported metric fails here, not at the first query
        self._bank = None
        self._thumb_bank = None
        if path is None:
            return                                   # filled by from_arrays
        if not force_reload and self.load():
            return
        self.create_thumbnail(path, thumb_x, thumb_y)
        self.create_image(path, max_size, metric)
        self.save()

    # -- construction from pre-extracted features (addition) ------------------------------
    @classmethod
    def from_arrays(cls, descriptors, positions, size, thumb_descriptors=None, thumb_positions=None,
                    thumb_size=None, distances=None, thumb_distances=None, path=None, options={}):
        self = cls(None, options)
        self.path = path
        desc = np.asarray(descriptors)
        desc = np.ascontiguousarray(desc if desc.dtype == np.uint8 else desc.astype(np.float32))
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 2)
        if desc.shape[0] != pos.shape[0]:
            raise ValueError("descriptors and positions disagree on the number of keypoints")
        # banks are created lazily; only missing self distances need the device right away
        if distances is None:
            ctx = matchutil._context(options)
            self._bank = ctx.bank(desc)
            distances = ctx.self_dist(self._bank)            # exact 2-NN (cache.pyx:271-273)
        distances = np.ascontiguousarray(distances, dtype=np.float64)
        if self._bank is not None:
            self._bank.set_selfdist(distances)
        self.original = {
            "descriptors": desc,
            "positions": pos,
            "distances": distances,
            "position_tree": Position_Index(pos, metric=self._metric[0], p=self._metric[1]),
            "size": (int(size[0]), int(size[1])),
        }
        if thumb_descriptors is not None:
            tdesc = np.asarray(thumb_descriptors)
            tdesc = np.ascontiguousarray(tdesc if tdesc.dtype == np.uint8 else tdesc.astype(np.float32))
            if thumb_distances is None:
                ctx = matchutil._context(options)
                self._thumb_bank = ctx.bank(tdesc)
                thumb_distances = ctx.self_dist(self._thumb_bank)   # cache.pyx:250-252
            thumb_distances = np.ascontiguousarray(thumb_distances, dtype=np.float64)
            if self._thumb_bank is not None:
                self._thumb_bank.set_selfdist(thumb_distances)
            self.thumb = {
                "descriptors": tdesc,
                "positions": np.ascontiguousarray(thumb_positions, dtype=np.float64).reshape(-1, 2),
                "distances": thumb_distances,
                "size": (int(thumb_size[0]), int(thumb_size[1])),
            }
        return self

    @classmethod
    def from_arrays_many(cls, images, options={}):
        """The Metric_Caches of a whole dataset (addition; the reference builds them one image at a time,
        cache.pyx:243-281, turntable.py:52-57).  ``images``: dicts with the keyword arguments of ``from_arrays``
        (descriptors, positions, size, thumb_descriptors, thumb_positions, thumb_size, path).  The self distances of
        ALL banks -- originals and thumbnails, of any sizes -- come from ``Context.self_dist_batch``: up to sixteen
        banks per launch of the triangular sweep instead of one launch (and one synchronisation) per bank, which for
        images of ~12.5k keypoints is the difference between 65 and 23 us per bank.  Same values
        as ``from_arrays`` image by image."""
        ctx = matchutil._context(options)
        conv = lambda d: np.ascontiguousarray(d if np.asarray(d).dtype == np.uint8 else np.asarray(d).astype(np.float32))
        banks, slots = [], []
        for k, im in enumerate(images):
            for key, dkey in (("descriptors", "distances"), ("thumb_descriptors", "thumb_distances")):
                if im.get(key) is not None and im.get(dkey) is None:
                    banks.append(ctx.bank(conv(im[key])))
                    slots.append((k, dkey))
        dists = ctx.self_dist_batch(banks) if banks else []
        extra = [dict() for _ in images]
        for (k, dkey), b, d in zip(slots, banks, dists):
            extra[k][dkey] = d
            extra[k]["_bank" if dkey == "distances" else "_thumb_bank"] = b
        out = []
        for im, ex in zip(images, extra):
            kw = {key: im[key] for key in ("descriptors", "positions", "size", "thumb_descriptors", "thumb_positions", "thumb_size",
                                          "distances", "thumb_distances", "path") if im.get(key) is not None}
            kw.update({key: v for key, v in ex.items() if not key.startswith("_")})
            mc = cls.from_arrays(options=options, **kw)
            mc._bank, mc._thumb_bank = ex.get("_bank"), ex.get("_thumb_bank")       # (the uploaded banks carry their distances)
            out.append(mc)
        return out

    # -- device banks -----------------------------------------------------------------------
    def bank(self, context=None):
        if self._bank is None:
            ctx = context or matchutil._context(self._options)
            self._bank = ctx.bank(self.original["descriptors"])
            self._bank.set_selfdist(self.original["distances"])
        return self._bank

    def thumb_bank(self, context=None):
        if self._thumb_bank is None:
            ctx = context or matchutil._context(self._options)
            self._thumb_bank = ctx.bank(self.thumb["descriptors"])
            self._thumb_bank.set_selfdist(self.thumb["distances"])
        return self._thumb_bank

    # -- radius select (cache.pyx:173-188) ----------------------------------------------------
    def radius_indices(self, x, y, radius, sort_results=True):
        idx, _ = self.original["position_tree"].radius(int(x), int(y), int(radius))
        if not sort_results:
            idx = np.sort(idx)
        return idx

    def get(self, x, y, radius, options={}):
        """ Retrieve all features within radius of position (x, y, radius are C ints in
        the reference: truncated here) """
        idx = self.radius_indices(x, y, radius, options.get("sort_results", True))
        return (self.original["descriptors"][idx], self.original["positions"][idx],
                self.original["distances"][idx], idx)

    # -- persistence (cache.pyx:191-239) --------------------------------------------------------
    def _data_path(self):
        p = self.path
        if isinstance(p, str):
            p = p.encode()
        return _ripemd160(p)

    def save(self, dir="data/image_data"):
        """ Exports cache to file: <dir>/<ripemd160(path)>.npz and ..._thumb.npz """
        data_path = self._data_path()
        if not os.path.exists(dir):
            os.makedirs(dir)
        np.savez("%s/%s" % (dir, data_path),
                 descriptors=self.original["descriptors"],
                 positions=self.original["positions"],
                 distances=self.original["distances"],
                 position_tree=np.frombuffer(self._tree_pickle(), dtype=np.uint8),
                 size=self.original["size"],
                 # (not read by the reference, which finds the metric inside its pickled tree)
                 fm_metric=np.array(_METRIC_NAMES[self.original["position_tree"].metric]))
        np.savez("%s/%s_thumb" % (dir, data_path),
                 positions=self.thumb["positions"],
                 descriptors=self.thumb["descriptors"],
                 distances=self.thumb["distances"],
                 size=self.thumb["size"])
        return data_path

    def _tree_pickle(self):
        """What the reference stores under ``position_tree`` (cache.pyx:204: pickle.dumps of its sklearn BallTree) and
        unpickles at load (cache.pyx:237), so that a file written here opens there: a real BallTree over the positions in
        this cache's metric when scikit-learn is importable, else a pickled None (this package rebuilds its own index
        from the positions and never reads the entry)."""
        try:
            from sklearn.neighbors import BallTree
            pos = self.original["positions"]
            if len(pos) == 0:
                return pickle.dumps(None)
            return pickle.dumps(BallTree(pos, metric=_METRIC_NAMES[self.original["position_tree"].metric]))
        except Exception:
            return pickle.dumps(None)

    def load(self, dir="data/image_data"):
        """ Loads file to Cache; False when no file exists for this path """
        data_path = self._data_path()
        full_path_npz = "%s/%s.npz" % (dir, data_path)
        full_path_thumb = "%s/%s_thumb.npz" % (dir, data_path)
        if not os.path.isfile(full_path_npz):
            return False
        data = np.load(full_path_npz, allow_pickle=False)
        data_thumb = np.load(full_path_thumb, allow_pickle=False)
        self.thumb = {k: data_thumb[k] for k in ("positions", "descriptors", "distances")}
        self.thumb["size"] = tuple(int(v) for v in data_thumb["size"])
        self.original = {k: data[k] for k in ("descriptors", "positions", "distances")}
        self.original["size"] = tuple(int(v) for v in data["size"])
        # The reference pickles its sklearn BallTree, which carries the metric it was BUILT with (cache.pyx:276), so a
        # cache saved under "manhattan" answers in Manhattan whatever the options at load time say.  Files written
        # here carry the metric's name (fm_metric) and are reloaded in it; files without it (written by the reference:
        # the pickle is not opened) take the metric of this cache's options.
        if "fm_metric" in data.files:
            self._metric = (str(data["fm_metric"]), None)
        self.original["position_tree"] = Position_Index(self.original["positions"], metric=self._metric[0], p=self._metric[1])
        self._bank = None
        self._thumb_bank = None
        return True

    # -- construction from an image file (needs OpenCV SIFT on the host) ----------------------
    def create_thumbnail(self, path, thumb_x, thumb_y):
        """ Thumbnail features + exact self 2-NN distances (cache.pyx:242-260) """
        from . import imaging
        thumbnail = imaging.get_thumbnail(path, (thumb_x, thumb_y))
        keypoints, descriptors = self._options.get("feature_function", matchutil.get_features)(thumbnail)
        ctx = matchutil._context(self._options)
        self._thumb_bank = ctx.bank(descriptors)
        nn_distances = ctx.self_dist(self._thumb_bank)
        self._thumb_bank.set_selfdist(nn_distances)
        self.thumb = {
            "descriptors": descriptors,
            "positions": keypoint_positions(keypoints),
            "distances": nn_distances,
            "size": (thumbnail.shape[1], thumbnail.shape[0]),
        }

    def create_image(self, path, max_size, metric):
        """ Full-image features, self distances and position index (cache.pyx:263-284) """
        from . import imaging
        img_data = imaging.open_img(path, max_size)
        keypoints, descriptors = self._options.get("feature_function", matchutil.get_features)(img_data)
        ctx = matchutil._context(self._options)
        self._bank = ctx.bank(descriptors)
        distances = ctx.self_dist(self._bank)
        self._bank.set_selfdist(distances)
        positions = keypoint_positions(keypoints)
        self.original = {
            "descriptors": descriptors,
            "positions": positions,
            "distances": distances,
            "position_tree": Position_Index(positions, metric=metric, p=self._metric[1]),
            "size": (img_data.shape[1], img_data.shape[0]),
        }
