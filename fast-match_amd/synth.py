"""Seeded synthetic SIFT-like descriptor banks; pure NumPy, host side.

Used by the tests and by bench.py to build the BASELINE.json workloads without SIFT:
``synth_sift`` mirrors OpenCV SIFT's post-processing (L2 normalise, clamp at 0.2,
re-normalise, x512, saturate to uint8); ``planted_pair`` derives a query bank from a
target bank so that a non-trivial set of matches survives the ratio test at tau 0.7.
"""
import numpy as np


def synth_sift(n, rng, dim=128):
    """[n, dim] uint8 SIFT-like descriptors."""
    out = np.empty((n, dim), dtype=np.uint8)
    step = 65536
    for s in range(0, n, step):
        m = min(step, n - s)
        x = rng.gamma(0.6, 1.0, (m, dim))
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
        np.minimum(x, 0.2, out=x)
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
        out[s:s + m] = np.clip(np.rint(512.0 * x), 0, 255).astype(np.uint8)
    return out


def planted_pair(nq, nt, seed, p=0.3, sigma=6.0, dim=128):
    """(Q uint8[nq,dim], T uint8[nt,dim], planted int64[nq] (-1 = independent row)).

    T = synth_sift(nt); a fraction p of the query rows are noisy copies of distinct
    target rows (Gaussian noise sigma, rounded, clipped); the rest are independent."""
    rng = np.random.default_rng(seed)
    T = synth_sift(nt, rng, dim)
    Q = synth_sift(nq, rng, dim)
    planted = np.full(nq, -1, dtype=np.int64)
    k = min(int(round(p * nq)), nt)
    if k > 0:
        qsel = rng.choice(nq, size=k, replace=False)
        tsel = rng.choice(nt, size=k, replace=False)
        noise = np.rint(rng.normal(0.0, sigma, (k, dim)))
        Q[qsel] = np.clip(T[tsel].astype(np.float64) + noise, 0, 255).astype(np.uint8)
        planted[qsel] = tsel
    return Q, T, planted


def _thumb_dims(size, box):
    s = min(float(box[0]) / size[0], float(box[1]) / size[1], 1.0)
    return (max(1, int(size[0] * s)), max(1, int(size[1] * s)))


def _positions(rng, n, w, h, centres, cluster_sigma, cluster_frac):
    """Keypoint positions: uniform, or (centres given) a Gaussian mixture over the uniform background -- real SIFT
    keypoints crowd on texture, 5-10x the mean density and more."""
    pos = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], axis=1)
    if centres is not None and len(centres):
        inc = rng.random(n) < cluster_frac
        k = rng.integers(0, len(centres), n)
        g = centres[k] + rng.normal(0.0, cluster_sigma, (n, 2))
        pos[inc] = np.clip(g[inc], 0.0, [w - 1.0, h - 1.0])
    return pos


def image_pair(size, n_keypoints, seed, p=0.3, sigma=6.0, shift=(37.0, -21.0), jitter=2.0,
               n_thumb=600, q_thumb_box=(600, 600), t_thumb_box=(400, 400),
               clusters=0, cluster_sigma=60.0, cluster_frac=0.5):
    """A synthetic query/target image pair as pre-extracted features.

    Keypoint positions are uniform in the image; a fraction p of the query keypoints are
    planted copies of distinct target keypoints: descriptor = noisy copy, position =
    target position - shift + N(0, jitter), so accepted matches propagate the expansion.
    Thumbnail banks are a subset of the keypoints (planted pairs first) with positions
    scaled to the thumbnail sizes and freshly perturbed descriptors.

    clusters = k > 0: a fraction ``cluster_frac`` of the keypoints of BOTH images sits in k Gaussian blobs
    (sigma ``cluster_sigma`` px, the query's blobs displaced by ``shift`` like the planted pairs), the rest is
    uniform: radius subsets and grid cells in a blob hold many times the uniform count.

    Returns (query, target) dicts:
      query : descriptors u8[n,128], positions f64[n,2], size, thumb_descriptors,
              thumb_positions, thumb_size
      target: same keys; plus 'planted' int64[n] on the query (target row or -1)."""
    w, h = size
    rng = np.random.default_rng(seed)
    T = synth_sift(n_keypoints, rng)
    Q = synth_sift(n_keypoints, rng)
    if clusters > 0:
        centres = np.stack([rng.uniform(0.15 * w, 0.85 * w, clusters), rng.uniform(0.15 * h, 0.85 * h, clusters)], axis=1)
        t_pos = _positions(rng, n_keypoints, w, h, centres, cluster_sigma, cluster_frac)
        q_pos = _positions(rng, n_keypoints, w, h, centres - np.array(shift), cluster_sigma, cluster_frac)
    else:           # (the draws of the uniform case are kept as they were: seeds of existing tests and bench legs)
        t_pos = np.stack([rng.uniform(0, w, n_keypoints), rng.uniform(0, h, n_keypoints)], axis=1)
        q_pos = np.stack([rng.uniform(0, w, n_keypoints), rng.uniform(0, h, n_keypoints)], axis=1)
    planted = np.full(n_keypoints, -1, dtype=np.int64)
    k = int(round(p * n_keypoints))
    qsel = rng.choice(n_keypoints, size=k, replace=False)
    tsel = rng.choice(n_keypoints, size=k, replace=False)
    Q[qsel] = np.clip(T[tsel].astype(np.float64) + np.rint(rng.normal(0.0, sigma, (k, 128))), 0, 255).astype(np.uint8)
    qp = t_pos[tsel] - np.array(shift) + rng.normal(0.0, jitter, (k, 2))
    q_pos[qsel] = np.clip(qp, 0.0, [w - 1.0, h - 1.0])
    planted[qsel] = tsel
    # thumbnails: planted pairs first, then unrelated keypoints on both sides
    n_thumb = min(n_thumb, n_keypoints)
    kp = min(k, n_thumb // 2)
    pick = rng.choice(k, size=kp, replace=False) if kp else np.zeros(0, dtype=np.int64)
    rest_q = rng.choice(n_keypoints, size=n_thumb - kp, replace=False)
    rest_t = rng.choice(n_keypoints, size=n_thumb - kp, replace=False)
    q_rows = np.concatenate([qsel[pick], rest_q])
    t_rows = np.concatenate([tsel[pick], rest_t])
    q_ts, t_ts = _thumb_dims(size, q_thumb_box), _thumb_dims(size, t_thumb_box)

    def thumb(desc, pos, rows, tsize):
        d = np.clip(desc[rows].astype(np.float64) + np.rint(rng.normal(0.0, 3.0, (len(rows), 128))), 0, 255)
        scale = np.array([tsize[0] / float(w), tsize[1] / float(h)])
        o = rng.permutation(len(rows))
        return d.astype(np.uint8)[o], (pos[rows] * scale)[o]

    qtd, qtp = thumb(Q, q_pos, q_rows, q_ts)
    ttd, ttp = thumb(T, t_pos, t_rows, t_ts)
    query = {"descriptors": Q, "positions": q_pos, "size": (w, h), "thumb_descriptors": qtd,
             "thumb_positions": qtp, "thumb_size": q_ts, "planted": planted}
    target = {"descriptors": T, "positions": t_pos, "size": (w, h), "thumb_descriptors": ttd,
              "thumb_positions": ttp, "thumb_size": t_ts}
    return query, target
