"""STAND-IN feature extractor -- NOT SIFT.  Plumbing aid for BASELINE config 1 only.

The reference extracts features with OpenCV SIFT on the host (``matchutil.get_features``,
reference ``matchutil.py:31-33``); SIFT is out of scope for this package (north_star) and
``cv2`` is not installed in the build / GPU images.  So that ``fastmatch.match()`` can still
be driven end to end on a real ``uint8[H, W, 3]`` pixel array -- thumbnail -> features ->
seeding, and the lazy ``Grid_Cache.cache()`` crop -> features -> per-cell bank path of
reference ``cache.pyx:124-138`` -- this module offers a small deterministic detector +
descriptor with SIFT's *output format*: keypoints in crop-local pixel coordinates and
128-D float32 descriptors holding integers 0..255 (4 x 4 cells x 8 orientation bins of
gradient magnitude, L2-normalised, clamped at 0.2, renormalised, x 512, saturated).

It has none of SIFT's scale or rotation invariance and its descriptors are not SIFT's:
results obtained with it say that the plumbing works, nothing about matching quality.
Pass it as ``options["feature_function"]`` (fastmatch.match, Metric_Cache); with ``cv2``
installed, leave the option out and the real SIFT is used.
"""
import numpy as np


class KeyPoint(object):
    """Stand-in for cv2.KeyPoint: only ``pt`` (x, y) is consumed by the callers
    (reference cache.pyx:253,274; fastmatch.pyx:127,158)."""
    __slots__ = ("pt", "size", "angle", "response")

    def __init__(self, x, y, response=0.0):
        self.pt = (float(x), float(y))
        self.size = 16.0
        self.angle = -1.0
        self.response = float(response)


def _box(a, r):
    """(2r+1)^2 box sum with zero padding, via a summed-area table."""
    h, w = a.shape
    s = np.zeros((h + 1, w + 1), dtype=np.float64)
    np.cumsum(np.cumsum(a, axis=0, dtype=np.float64), axis=1, out=s[1:, 1:])
    y0 = np.clip(np.arange(h) - r, 0, h)
    y1 = np.clip(np.arange(h) + r + 1, 0, h)
    x0 = np.clip(np.arange(w) - r, 0, w)
    x1 = np.clip(np.arange(w) + r + 1, 0, w)
    return s[y1][:, x1] - s[y0][:, x1] - s[y1][:, x0] + s[y0][:, x0]


def standin_features(data, max_keypoints=None, threshold=1.0, nms=1):
    """(keypoints, descriptors) of a uint8 [H, W] or [H, W, 3] array, cv2.detectAndCompute
    style: a list of KeyPoint and float32 [n, 128], or ([], None) when nothing is found."""
    img = np.asarray(data)
    gray = img.astype(np.float64).mean(axis=2) if img.ndim == 3 else img.astype(np.float64)
    h, w = gray.shape
    if h < 20 or w < 20:
        return [], None
    # detector: local maxima ((2 nms + 1)^2 window) of |box3 - box7| difference-of-boxes, 8 px inside the border
    b3 = _box(gray, 1) / 9.0
    b7 = _box(gray, 3) / 49.0
    resp = np.abs(b3 - b7)
    inner = np.zeros_like(resp, dtype=bool)
    inner[8:h - 8, 8:w - 8] = True
    peak = inner & (resp > threshold)
    for dy in range(-nms, nms + 1):
        for dx in range(-nms, nms + 1):
            if dy == 0 and dx == 0:
                continue
            sh = np.full_like(resp, -1.0)
            ys0, ys1 = max(dy, 0), h + min(dy, 0)
            xs0, xs1 = max(dx, 0), w + min(dx, 0)
            sh[ys0 - dy:ys1 - dy, xs0 - dx:xs1 - dx] = resp[ys0:ys1, xs0:xs1]
            # strict on one side so that a plateau yields one keypoint (deterministic)
            peak &= (resp > sh) if (dy, dx) > (0, 0) else (resp >= sh)
    ys, xs = np.nonzero(peak)
    if len(ys) == 0:
        return [], None
    if max_keypoints is not None and len(ys) > max_keypoints:
        keep = np.sort(np.argsort(-resp[ys, xs], kind="stable")[:max_keypoints])
        ys, xs = ys[keep], xs[keep]
    # descriptor: gradient orientation histograms, 8 bins, 4 x 4 cells of 4 x 4 pixels
    gx = np.zeros_like(gray)
    gy = np.zeros_like(gray)
    gx[:, 1:-1] = gray[:, 2:] - gray[:, :-2]
    gy[1:-1, :] = gray[2:, :] - gray[:-2, :]
    mag = np.hypot(gx, gy)
    b = np.floor((np.arctan2(gy, gx) + np.pi) * (8.0 / (2.0 * np.pi))).astype(np.int64) % 8
    n = len(ys)
    desc = np.zeros((n, 4, 4, 8), dtype=np.float64)
    for k in range(8):
        plane = np.where(b == k, mag, 0.0)
        s = np.zeros((h + 1, w + 1), dtype=np.float64)
        np.cumsum(np.cumsum(plane, axis=0), axis=1, out=s[1:, 1:])
        for cy in range(4):
            for cx in range(4):
                y0 = ys - 8 + 4 * cy
                x0 = xs - 8 + 4 * cx
                desc[:, cy, cx, k] = s[y0 + 4, x0 + 4] - s[y0, x0 + 4] - s[y0 + 4, x0] + s[y0, x0]
    d = desc.reshape(n, 128)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    np.minimum(d, 0.2, out=d)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    d = np.clip(np.rint(512.0 * d), 0, 255).astype(np.float32)
    keypoints = [KeyPoint(x, y, resp[y, x]) for y, x in zip(ys, xs)]
    return keypoints, d


standin_features.is_standin = True
