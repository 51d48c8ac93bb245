"""Optional live cross-check against the reference's real dependency (SURVEY.md section 4
item 5, 8(c)): when ``import cv2`` works, the oracle -- and on a GPU box the HIP path -- must
reproduce ``cv2.BFMatcher`` itself on the same arrays.  cv2 is absent from the build image and
the GPU image, so these tests normally SKIP; they are the hook that would lift the oracle's
"parity unpinned" status on a machine that has OpenCV."""
import numpy as np
import pytest

import oracle

cv2 = pytest.importorskip("cv2")


def _cv_knn(Q, T, k, cross):
    m = cv2.BFMatcher(cv2.NORM_L2, crossCheck=cross).knnMatch(Q, T, k=k)
    idx = np.full((len(Q), k), -1, dtype=np.int32)
    dist = np.full((len(Q), k), np.inf, dtype=np.float32)
    for qi, row in enumerate(m):
        for j, d in enumerate(row):
            idx[qi, j], dist[qi, j] = d.trainIdx, d.distance
    return idx, dist


def _data(seed):
    rng = np.random.default_rng(seed)
    Q = rng.integers(0, 256, (400, 128)).astype(np.float32)      # integer valued like cv2 SIFT output
    T = rng.integers(0, 256, (300, 128)).astype(np.float32)
    T[7] = T[3]
    Q[5] = T[3]
    return Q, T


def test_oracle_equals_cv2_bfmatcher():
    Q, T = _data(1)
    idx, dist = oracle.bf_knn(Q, T, 2)
    cidx, cdist = _cv_knn(Q, T, 2, False)
    assert np.array_equal(idx, cidx) and np.array_equal(dist, cdist)
    tidx, xd = oracle.bf_xcheck1(Q, T)
    ctidx, cxd = _cv_knn(Q, T, 1, True)
    assert np.array_equal(tidx, ctidx[:, 0]) and np.array_equal(xd, cxd[:, 0])


@pytest.mark.gpu
def test_hip_equals_cv2_bfmatcher(ctx):
    Q, T = _data(2)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    idx, dist = ctx.knn2(qb, tb)
    cidx, cdist = _cv_knn(Q, T, 2, False)
    assert np.array_equal(idx, cidx) and np.array_equal(dist, cdist)
    tidx, xd = ctx.xcheck1(qb, tb)
    ctidx, cxd = _cv_knn(Q, T, 1, True)
    assert np.array_equal(tidx, ctidx[:, 0]) and np.array_equal(xd, cxd[:, 0])
