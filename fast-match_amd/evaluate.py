"""Precision harness in the style of the reference's ``turntable.evaluate`` (reference
``turntable.py:27-80``): build the matcher once per image pair, then query it for a list of
thresholds (``turntable.py:59-60``) -- the device-resident banks and the expansion state are
reused across thresholds -- and score every match list against a ground truth.

The reference's ground truth module (``turntable_ground_truth.pyx``) is not in its repository
(``turntable.py:16``, ``setup.py:10``); two scorers that need nothing external are provided:

* ``homography_scorer(H, distance_threshold)`` -- a match (p_query, p_target) is correct when
  H maps p_query to within ``distance_threshold`` pixels of p_target.  ``load_homography``
  reads the 3x3 text files shipped with the Oxford sets (reference ``images/graf/H1to4p``;
  H maps img1 coordinates to imgN coordinates).
* ``planted_scorer(planted, target_positions)`` -- for the synthetic pairs of ``synth.image_pair``.
"""
import numpy as np

from . import fastmatch


def load_homography(path):
    H = np.loadtxt(path, dtype=np.float64)
    if H.shape != (3, 3):
        raise ValueError("%s does not hold a 3x3 homography" % path)
    return H


def homography_scorer(H, distance_threshold=5.0, query_is_source=True):
    """Scorer for pairs related by a plane homography.  H maps source-image coordinates to
    destination-image coordinates; ``query_is_source`` says which side the query image is."""
    H = np.asarray(H, dtype=np.float64)
    Hi = H if query_is_source else np.linalg.inv(H)

    def score(index, positions, ratio):
        if len(positions) == 0:
            return np.zeros(0, dtype=bool)
        p = np.asarray(positions, dtype=np.float64).reshape(-1, 2, 2)
        src = np.concatenate([p[:, 0, :], np.ones((p.shape[0], 1))], axis=1)
        proj = src @ Hi.T
        proj = proj[:, :2] / proj[:, 2:3]
        return np.hypot(proj[:, 0] - p[:, 1, 0], proj[:, 1] - p[:, 1, 1]) <= distance_threshold
    return score


def planted_scorer(planted, target_positions, atol=1e-9):
    planted = np.asarray(planted)
    tp = np.asarray(target_positions, dtype=np.float64)

    def score(index, positions, ratio):
        if len(index) == 0:
            return np.zeros(0, dtype=bool)
        index = np.asarray(index)
        p = np.asarray(positions, dtype=np.float64).reshape(-1, 2, 2)
        want = planted[index]
        ok = want >= 0
        good = np.zeros(len(index), dtype=bool)
        good[ok] = np.all(np.abs(p[ok, 1, :] - tp[want[ok]]) <= atol, axis=1)
        return good
    return score


def evaluate(pairs, thresholds, scorers, options={}, match_fun=None):
    """``pairs`` = [(query_cache, target_img), ...], ``scorers`` one per pair.  Returns a list of
    rows {"tau", "correct", "total", "precision"} accumulated over the pairs, like the table of
    ``Evaluate Turntable.ipynb``.  ``match_fun`` defaults to ``fastmatch.match``."""
    match_fun = match_fun or fastmatch.match
    opts = dict(options, return_arrays=True)
    getters = [match_fun(q, t, opts) for q, t in pairs]            # seeding once per pair
    thresholds = [float(t) for t in thresholds]
    results = matches_for_thresholds(getters, thresholds)
    rows = []
    for ti, tau in enumerate(thresholds):
        correct = total = 0
        for pi, score in enumerate(scorers):
            res = results[pi][ti]
            if isinstance(res, tuple):
                index, positions, ratio = res
            else:                                                  # host loop: list of (index, dict)
                index = np.array([m[0] for m in res], dtype=np.int64)
                positions = np.array([m[1]["positions"] for m in res]).reshape(-1, 2, 2)
                ratio = np.array([m[1]["ratio"] for m in res])
            good = score(index, positions, ratio)
            correct += int(good.sum())
            total += int(len(index))
        rows.append({"tau": float(tau), "correct": correct, "total": total,
                     "precision": (correct / total) if total else float("nan")})
    return rows


def matches_for_thresholds(getters, thresholds, as_arrays=True):
    """results[pair][threshold] for closures returned by ``fastmatch.match``: every (pair, threshold)
    run whose pair has device-resident expansion state goes into ONE launch of the device loop (one
    workgroup per run) and one fetch; the rest -- and runs the device gave up on -- take the closure's
    own path.  The reference evaluates them one after the other: ``{ tau : f(tau) for tau in thresholds }``
    per pair (turntable.py:59-60)."""
    results = [[None] * len(thresholds) for _ in getters]
    runs = []                                                      # (pair, threshold index, expander)
    for pi, get in enumerate(getters):
        ex = get.expander() if hasattr(get, "expander") else None
        if ex is not None:
            runs.extend((pi, ti, ex) for ti in range(len(thresholds)))
    by_ctx = {}
    for r in runs:
        by_ctx.setdefault(id(getters[r[0]].context), []).append(r)
    for group in by_ctx.values():
        ctx = getters[group[0][0]].context
        got = fastmatch.run_device_loops(ctx, [ex for _, _, ex in group],
                                         [getters[pi].seeds_for(thresholds[ti]) for pi, ti, _ in group],
                                         [thresholds[ti] for _, ti, _ in group], as_arrays=as_arrays)
        for (pi, ti, _), res in zip(group, got):
            results[pi][ti] = res
    for pi, get in enumerate(getters):
        for ti, tau in enumerate(thresholds):
            if results[pi][ti] is None:
                results[pi][ti] = get(tau)
    return results
