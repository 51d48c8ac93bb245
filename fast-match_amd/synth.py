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
