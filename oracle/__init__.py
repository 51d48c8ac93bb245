"""CPU oracle for the Fast-Match descriptor-matching hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the
checker.  The product path (``fast-match_amd/``) never imports it.

PARITY UNPINNED (see ``bfmatch_oracle.c`` header and DESIGN.md): the reference holds
no tests or golden vectors for the matcher and delegates the arithmetic to an
un-pinned OpenCV; this is a restatement of OpenCV's published BFMatcher semantics
 anchored on the reference's call sites
(``fastmatch.pyx:122-124,161-165``; ``matchutil.py:39-43``; ``cache.pyx:250-252``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
            os.path.join(_HERE, "bfmatch_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    try:
        lib = ctypes.CDLL(so)
    except OSError:
        build(force=True)
        lib = ctypes.CDLL(so)
    c = ctypes
    p = c.c_void_p
    lib.orc_bf_knn_f32.argtypes = [p, c.c_int64, p, c.c_int64, c.c_int, c.c_int, c.c_int, p, p, c.c_int]
    lib.orc_bf_knn_u8.argtypes = [p, c.c_int64, p, c.c_int64, c.c_int, c.c_int, p, p, c.c_int]
    lib.orc_bf_xcheck1_f32.argtypes = [p, c.c_int64, p, c.c_int64, c.c_int, c.c_int, p, p, c.c_int]
    lib.orc_bf_xcheck1_u8.argtypes = [p, c.c_int64, p, c.c_int64, c.c_int, p, p, c.c_int]
    lib.orc_bf_xcheck1_u8_simd.argtypes = [p, c.c_int64, p, c.c_int64, c.c_int, p, p, c.c_int]
    lib.orc_bf_xcheck1_u8_blocked.argtypes = [p, c.c_int64, p, c.c_int64, c.c_int, p, p, c.c_int]
    lib.orc_ratio_filter.argtypes = [p, p, p, c.c_int64, c.c_double, p, p, p]
    lib.orc_lowe_ratio.argtypes = [p, c.c_int64, p]
    lib.orc_max_threads.restype = c.c_int
    _LIB = lib
    return lib


def max_threads():
    return int(_lib().orc_max_threads())


def _prep(a):
    a = np.asarray(a)
    if a.dtype == np.uint8:
        return np.ascontiguousarray(a), "u8"
    return np.ascontiguousarray(a, dtype=np.float32), "f32"


def _pair(Q, T):
    Q, kq = _prep(Q)
    T, kt = _prep(T)
    if kq != kt:
        # cv2 raises on dtype mismatch
        raise TypeError("query and train descriptors must have the same dtype")
    if Q.ndim != 2 or T.ndim != 2 or (Q.shape[1] != T.shape[1]):
        raise ValueError("descriptor banks must be [n, dim] with equal dim")
    return Q, T, kq


def bf_knn(Q, T, k=2, order=0, threads=0):
    """cv2.BFMatcher(NORM_L2, crossCheck=False).knnMatch(Q, T, k) as arrays.

    Returns (idx int32[nq,k], dist float32[nq,k]); missing neighbours (nt < k) are
    idx -1 / dist +inf.  ``order``: fp32 accumulation order (0 = OpenCV generic unrolled-4,
    1 = fma chain, the order the device fp32 kernel uses, 2 = OpenCV 2.4.x SSE2 2x4 lanes,
    3 = OpenCV 4.x 128-bit universal intrinsics 4x4 lanes); ignored for uint8."""
    Q, T, kind = _pair(Q, T)
    nq, nt, dim = Q.shape[0], T.shape[0], Q.shape[1]
    idx = np.empty((nq, k), dtype=np.int32)
    dist = np.empty((nq, k), dtype=np.float32)
    lib = _lib()
    if kind == "u8":
        rc = lib.orc_bf_knn_u8(Q.ctypes.data, nq, T.ctypes.data, nt, dim, k,
                               idx.ctypes.data, dist.ctypes.data, threads)
    else:
        rc = lib.orc_bf_knn_f32(Q.ctypes.data, nq, T.ctypes.data, nt, dim, k, order,
                                idx.ctypes.data, dist.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("oracle bf_knn failed: %d" % rc)
    return idx, dist


def bf_xcheck1(Q, T, order=0, threads=0):
    """cv2.BFMatcher(NORM_L2, crossCheck=True).knnMatch(Q, T, k=1) as arrays.

    Returns (tidx int32[nq] (-1 = empty inner list), dist float32[nq] (+inf if none))."""
    Q, T, kind = _pair(Q, T)
    nq, nt, dim = Q.shape[0], T.shape[0], Q.shape[1]
    tidx = np.empty(nq, dtype=np.int32)
    dist = np.empty(nq, dtype=np.float32)
    lib = _lib()
    if kind == "u8":
        rc = lib.orc_bf_xcheck1_u8(Q.ctypes.data, nq, T.ctypes.data, nt, dim,
                                   tidx.ctypes.data, dist.ctypes.data, threads)
    else:
        rc = lib.orc_bf_xcheck1_f32(Q.ctypes.data, nq, T.ctypes.data, nt, dim, order,
                                    tidx.ctypes.data, dist.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("oracle bf_xcheck1 failed: %d" % rc)
    return tidx, dist


def bf_xcheck1_simd(Q, T, threads=0):
    """``bf_xcheck1`` for uint8 banks through the vectorised scan (int16 differences, vpmaddwd, eight output rows per
    pass): the same results bit for bit, the CPU programmed the way a CPU would be -- bench.py's second baseline."""
    Q, T, kind = _pair(Q, T)
    if kind != "u8":
        raise ValueError("the vectorised baseline exists for uint8 banks")
    nq, nt, dim = Q.shape[0], T.shape[0], Q.shape[1]
    tidx = np.empty(nq, dtype=np.int32)
    dist = np.empty(nq, dtype=np.float32)
    rc = _lib().orc_bf_xcheck1_u8_simd(Q.ctypes.data, nq, T.ctypes.data, nt, dim, tidx.ctypes.data, dist.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("oracle bf_xcheck1_simd failed: %d" % rc)
    return tidx, dist


def have_vnni():
    """Does this host run the blocked baseline (AVX-512 VNNI)?"""
    return bool(_lib().orc_have_vnni())


def bf_xcheck1_blocked(Q, T, threads=0):
    """``bf_xcheck1`` for uint8 banks the way a CPU with AVX-512 VNNI would be programmed (vpdpbusd, 64 output rows x 6
    candidates per register block, vector compares; bfmatch_oracle.c ``knn1_rows_u8_vnni``): the same results bit for bit --
    bench.py's third CPU baseline, the honest denominator of a GPU / CPU ratio.  RuntimeError on a host without VNNI."""
    Q, T, kind = _pair(Q, T)
    if kind != "u8":
        raise ValueError("the blocked baseline exists for uint8 banks")
    nq, nt, dim = Q.shape[0], T.shape[0], Q.shape[1]
    tidx = np.empty(nq, dtype=np.int32)
    dist = np.empty(nq, dtype=np.float32)
    rc = _lib().orc_bf_xcheck1_u8_blocked(Q.ctypes.data, nq, T.ctypes.data, nt, dim, tidx.ctypes.data, dist.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("oracle bf_xcheck1_blocked failed: %d%s" % (rc, " (no AVX-512 VNNI on this host)" if rc == -2 else ""))
    return tidx, dist


def self_dist(D, order=0, threads=0):
    """Metric_Cache self-distances: bf_match(d, d, k=2) then r[1].distance
    (cache.pyx:250-252).  Rows with fewer than 2 neighbours get +inf."""
    _, dist = bf_knn(D, D, k=2, order=order, threads=threads)
    return dist[:, 1].astype(np.float64)


def ratio_filter(dist, selfdist, tau, qrows=None):
    """ratio = float64(dist) / selfdist[qrow]; passed = ratio < tau
    (fastmatch.pyx:124,165 and :50,75,82).  Returns (ratio f64[n], pass bool[n])."""
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    selfdist = np.ascontiguousarray(selfdist, dtype=np.float64)
    n = dist.shape[0]
    ratio = np.empty(n, dtype=np.float64)
    passed = np.empty(n, dtype=np.uint8)
    npass = ctypes.c_int64(0)
    qp = None
    if qrows is not None:
        qrows = np.ascontiguousarray(qrows, dtype=np.int32)
        qp = qrows.ctypes.data
    _lib().orc_ratio_filter(dist.ctypes.data, selfdist.ctypes.data, qp, n, float(tau),
                            ratio.ctypes.data, passed.ctypes.data, ctypes.byref(npass))
    return ratio, passed.astype(bool)


def lowe_ratio(dist2):
    """Classic Ratio-Match d1/d2 in float64 (Classic Matching.ipynb cell 3)."""
    dist2 = np.ascontiguousarray(dist2, dtype=np.float32)
    out = np.empty(dist2.shape[0], dtype=np.float64)
    _lib().orc_lowe_ratio(dist2.ctypes.data, dist2.shape[0], out.ctypes.data)
    return out
