"""Matcher wrappers with the reference's operator signatures, running on the HIP path.

Mirrors ``matchutil.py`` of the reference:

* ``bf_match(dt1, dt2, k=1, options={})``     -- reference ``matchutil.py:39-43``
  (``cv2.BFMatcher(cv2.NORM_L2, crossCheck).knnMatch(dt1, dt2, k=k)``; crossCheck is
  honoured only when ``k == 1``).
* ``flann_match(dt1, dt2, k=1, options={})``  -- reference ``matchutil.py:46-67``.  The
  reference's FLANN index is randomized and approximate; the exact brute-force k-NN it
  approximates is returned instead, so results are
  deterministic.
* ``sift / get_features / get_keypoints``     -- reference ``matchutil.py:22-36``; SIFT
  stays in OpenCV on the host and needs ``cv2``.

The return value is the same shape OpenCV gives: a list with one inner list per query
row holding up to ``k`` ``DMatch`` objects (attributes ``queryIdx, trainIdx, imgIdx,
distance``).  ``*_arrays`` variants return NumPy arrays and skip the Python objects.
Errors (dtype/width mismatch, unsupported k) raise ``FastMatchHipError``/``ValueError``
where cv2 would raise ``cv2.error``.  There is no CPU fallback.
"""
import numpy as np

from . import _ffi


class DMatch(object):
    """Stand-in for cv2.DMatch (same attribute names)."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return "DMatch(queryIdx=%d, trainIdx=%d, distance=%r)" % (self.queryIdx, self.trainIdx, self.distance)


def _context(options):
    ctx = options.get("context") if options else None
    if ctx is not None:
        return ctx
    return _ffi.default_context(options.get("device") if options else None)


def _as_bank(ctx, d):
    """Accept a resident Bank or an ndarray (uploaded for this call only)."""
    if isinstance(d, _ffi.Bank):
        return d, False
    a = np.asarray(d)
    if a.ndim != 2:
        raise ValueError("descriptors must be a 2-D [n, dim] array")
    return ctx.bank(a), True


def _as_bank_pair(ctx, dt1, dt2):
    """Both operands as banks of ONE kind.  A float32 array whose values all happen to be
    integers in 0..255 is uploaded on the exact int8 route, any other float32 array on the
    float32 route; cv2.BFMatcher takes any two float32 arrays, so when the two kinds differ
    (say a one-row all-zero bank against normalised descriptors) the integer-valued array is
    uploaded again on the float32 route, which yields the same numbers for it.  A resident
    Bank of the other kind cannot be re-uploaded: the library then reports the mismatch."""
    qb, q_tmp = _as_bank(ctx, dt1)
    try:
        tb, t_tmp = _as_bank(ctx, dt2)
    except Exception:
        if q_tmp:
            qb.close()
        raise
    if qb.kind != tb.kind:
        if q_tmp and qb.kind == _ffi.FM_BANK_I8 and np.asarray(dt1).dtype != np.uint8:
            qb.close()
            qb = ctx.bank(np.asarray(dt1), float_route=True)
        elif t_tmp and tb.kind == _ffi.FM_BANK_I8 and np.asarray(dt2).dtype != np.uint8:
            tb.close()
            tb = ctx.bank(np.asarray(dt2), float_route=True)
    return qb, q_tmp, tb, t_tmp


def bf_match_arrays(dt1, dt2, k=1, options={}):
    """Array form of :func:`bf_match`.

    crossCheck (k == 1 only): returns ``(tidx int32[nq], dist float32[nq])`` with
    ``tidx == -1`` where OpenCV returns an empty inner list.
    Otherwise returns ``(idx int32[nq, k], dist float32[nq, k])`` with ``-1`` / ``inf``
    where the train set has fewer than k rows (1 <= k <= 8; ``matchutil.py:39-43`` passes any k to cv2)."""
    k = int(k)
    if k < 1:
        raise ValueError("bf_match: k must be at least 1")
    if k > 8:
        # cv2.BFMatcher.knnMatch takes any k; the reference calls k = 1 and 2 (fastmatch.pyx:122-123, 161-162, cache.pyx:250)
        raise ValueError("bf_match: k = %d: the HIP path builds k-NN lists up to k = 8 (FM_EUNSUPPORTED beyond)" % k)
    crossCheck = k == 1 and options.get("crossCheck", False) == True   # noqa: E712  (reference semantics)
    ctx = _context(options)
    qb, q_tmp, tb, t_tmp = _as_bank_pair(ctx, dt1, dt2)
    try:
        if crossCheck:
            return ctx.xcheck1(qb, tb)
        if k > 2:
            return ctx.knn(qb, tb, k)                   # exact lists off the matrix cores (fm_knn)
        idx, dist = ctx.knn2(qb, tb)
        return idx[:, :k], dist[:, :k]
    finally:
        if q_tmp:
            qb.close()
        if t_tmp:
            tb.close()


def matches_from_arrays(idx, dist):
    """Build OpenCV's list-of-lists of DMatch from index/distance arrays."""
    idx = np.asarray(idx)
    dist = np.asarray(dist)
    if idx.ndim == 1:
        idx = idx[:, None]
        dist = dist[:, None]
    out = []
    for qi in range(idx.shape[0]):
        row = [DMatch(qi, idx[qi, j], dist[qi, j]) for j in range(idx.shape[1]) if idx[qi, j] >= 0]
        out.append(row)
    return out


def bf_match(dt1, dt2, k=1, options={}):
    """ Use the HIP brute-force matcher with OpenCV BFMatcher(NORM_L2) semantics """
    idx, dist = bf_match_arrays(dt1, dt2, k=k, options=options)
    return matches_from_arrays(idx, dist)


def flann_match(dt1, dt2, k=1, options={}):
    """ Exact k-NN in place of the reference's approximate FLANN kd-tree search.
    ``algorithm`` / ``trees`` / ``checks`` are accepted and ignored. """
    opts = dict(options)
    opts.pop("crossCheck", None)
    return bf_match(dt1, dt2, k=k, options=opts)


def flann_match_arrays(dt1, dt2, k=1, options={}):
    opts = dict(options)
    opts.pop("crossCheck", None)
    return bf_match_arrays(dt1, dt2, k=k, options=opts)


def ratio_match_arrays(dt1, dt2, tau, options={}):
    """Classic Ratio-Match (the reference's baseline, ``Classic Matching.ipynb`` cell 3):
    brute-force 2-NN then ``m[0].distance / m[1].distance < tau`` in float64, on the device.
    Returns (query idx, train idx, distance, ratio) of the accepted matches, ascending query."""
    ctx = _context(options)
    qb, q_tmp, tb, t_tmp = _as_bank_pair(ctx, dt1, dt2)
    try:
        return ctx.knn2_ratio(qb, tb, tau)
    finally:
        if q_tmp:
            qb.close()
        if t_tmp:
            tb.close()


# ---- SIFT stays in OpenCV on the host -------------------------------------------------

def sift():
    try:
        import cv2
    except ImportError:
        raise Exception("Can't find SIFT: OpenCV (cv2) is not installed; pass pre-extracted "
                        "features (cache.Feature_Image / Metric_Cache.from_arrays) instead")
    if hasattr(cv2, "SIFT_create"):
        return cv2.SIFT_create()
    if hasattr(cv2, "SIFT"):
        return cv2.SIFT()
    if hasattr(cv2, "xfeatures2d"):
        return cv2.xfeatures2d.SIFT_create()
    raise Exception("Can't find SIFT")


def get_features(data, feature_type="SIFT"):
    return sift().detectAndCompute(data, None)


def get_keypoints(data, feature_type="SIFT"):
    return sift().detect(data)
