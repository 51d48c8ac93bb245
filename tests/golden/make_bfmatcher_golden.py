"""ONE run of this script on any machine that has OpenCV pins the matcher oracle for good.

The arithmetic of the hot path lives in cv2 (``cv2.BFMatcher(cv2.NORM_L2, crossCheck).knnMatch`` -- the reference's call
sites: fastmatch.pyx:122-123, 161-162; matchutil.py:39-43; cache.pyx:250), which is absent from the build image and from
the GPU image, so ``oracle/`` restates OpenCV's semantics from SURVEY.md Appendix A and the parity of everything green is
"unpinned" (DESIGN.md section 2).  This script runs the REAL BFMatcher on the suite's known-answer, tie, duplicate and
float32-root-tie banks (tests/kat.py) and on seeded SIFT-like banks -- uint8 rows handed over as integer-valued float32,
which is what cv2's SIFT emits, and non-integer float32 rows -- for k = 1 cross-checked and k = 2, and writes inputs +
cv2's answers + ``cv2.__version__`` to ``tests/golden/bfmatcher_golden.npz``.  The fixture is DATA (arrays): it travels to
machines where the live module cannot, and tests/test_bfmatcher_golden.py (CPU: the oracle; ``-m gpu``: the HIP path)
consumes it when present and skips otherwise.

    python tests/golden/make_bfmatcher_golden.py            # needs: import cv2

Nothing of the reference's source is read or copied; cv2 is called exactly as the reference calls it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))               # tests/ (kat.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def cv_knn(cv2, Q, T, k, cross):
    """knnMatch as (idx int32[nq, k], dist float32[nq, k]); -1 / +inf where an inner list is shorter than k
    (crossCheck=True: 0 or 1 entries; k = 2 against one train row: 1 entry)."""
    idx = np.full((len(Q), k), -1, dtype=np.int32)
    dist = np.full((len(Q), k), np.inf, dtype=np.float32)
    if len(Q) == 0 or len(T) == 0:
        return idx, dist
    for qi, row in enumerate(cv2.BFMatcher(cv2.NORM_L2, crossCheck=cross).knnMatch(Q, T, k=k)):
        for j, d in enumerate(row):
            idx[qi, j], dist[qi, j] = d.trainIdx, d.distance
    return idx, dist


def cases():
    """(name, Q, T) -- Q, T float32 C-contiguous, as the reference hands them to cv2 (cache.pyx:188: fancy-indexed copies)."""
    import kat
    from fastmatch_amd import synth
    out = []
    for name, Q, T, _, _ in kat.xcheck_cases() + kat.sqrt_tie_xcheck_cases():
        out.append(("x_" + name, Q, T))
    for name, Q, T, _, _ in kat.knn2_cases() + kat.sqrt_tie_knn2_cases():
        out.append(("k_" + name, Q, T))
    D, _ = kat.selfdist_case()
    out.append(("self_duplicates", D, D))
    rng = np.random.default_rng(20250005)
    Q, T, _ = synth.planted_pair(700, 500, seed=11)
    T[7] = T[3]
    Q[5] = T[3]
    Q[600] = Q[12]
    out.append(("sift_like_u8", Q, T))
    out.append(("sift_like_self", T, T))
    fq, ft = kat.far_banks(60, 40, rng)
    out.append(("sqrt_tie_banks", fq, ft))
    qf = Q[:300].astype(np.float32) + rng.uniform(-0.5, 0.5, (300, 128)).astype(np.float32)
    tf = T[:260].astype(np.float32) + rng.uniform(-0.5, 0.5, (260, 128)).astype(np.float32)
    out.append(("non_integer_f32", qf, tf))
    r = np.sqrt(Q[:200].astype(np.float32) / np.maximum(Q[:200].sum(1, keepdims=True), 1)).astype(np.float32)
    out.append(("rootsift_f32_self", r, r))
    return [(n, np.ascontiguousarray(q, dtype=np.float32), np.ascontiguousarray(t, dtype=np.float32)) for n, q, t in out]


def main():
    try:
        import cv2
    except ImportError:
        sys.stderr.write("make_bfmatcher_golden.py: `import cv2` failed -- run this where OpenCV is installed\n")
        return 2
    arrays = {"cv2_version": np.array(cv2.__version__), "names": np.array([n for n, _, _ in cases()])}
    for name, Q, T in cases():
        arrays[name + "__Q"], arrays[name + "__T"] = Q, T
        if len(T) == 0:                                    # cv2 raises on an empty train set; the reference never calls it so
            continue
        arrays[name + "__x_idx"], arrays[name + "__x_dist"] = cv_knn(cv2, Q, T, 1, True)
        arrays[name + "__k_idx"], arrays[name + "__k_dist"] = cv_knn(cv2, Q, T, 2, False)
    path = os.path.join(HERE, "bfmatcher_golden.npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%d cases, cv2 %s)" % (path, len(arrays["names"]), cv2.__version__))
    return 0


if __name__ == "__main__":
    sys.exit(main())
