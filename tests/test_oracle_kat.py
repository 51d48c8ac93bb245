"""CPU: pins the oracle to hand-derived OpenCV-semantics known answers (the reference
holds no golden vectors for the matcher: SURVEY.md 8(c), 'parity unpinned')."""
import numpy as np
import pytest

import oracle
from kat import (xcheck_cases, knn2_cases, selfdist_case, sqrt_tie_knn2_cases, sqrt_tie_xcheck_cases,
                 SQRT_TIE_MIN, row_with_sumsq, far_banks)


@pytest.mark.parametrize("case", xcheck_cases() + sqrt_tie_xcheck_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("as_f32", [False, True])
def test_xcheck_kat(case, as_f32):
    _, Q, T, etidx, edist = case
    if as_f32:
        Q, T = Q.astype(np.float32), T.astype(np.float32)
    tidx, dist = oracle.bf_xcheck1(Q, T)
    assert tidx.tolist() == etidx
    assert np.array_equal(dist, np.array(edist, dtype=np.float32))


@pytest.mark.parametrize("case", knn2_cases() + sqrt_tie_knn2_cases(), ids=lambda c: c[0])
def test_knn2_kat(case):
    _, Q, T, eidx, edist = case
    idx, dist = oracle.bf_knn(Q, T, 2)
    assert idx.tolist() == eidx
    assert np.array_equal(dist, np.array(edist, dtype=np.float32))


def test_selfdist_kat():
    D, exp = selfdist_case()
    assert oracle.self_dist(D).tolist() == exp


def test_empty_query():
    tidx, dist = oracle.bf_xcheck1(np.zeros((0, 128), np.uint8), np.zeros((5, 128), np.uint8))
    assert tidx.shape == (0,) and dist.shape == (0,)
    idx, d = oracle.bf_knn(np.zeros((0, 128), np.uint8), np.zeros((5, 128), np.uint8), 2)
    assert idx.shape == (0, 2)


def test_ratio_filter_semantics():
    # float64 division of float32 distance by float64 self distance; x/0 -> inf, 0/0 -> nan;
    # both rejected by '<' (fastmatch.pyx:165, :75)
    dist = np.array([3.0, 0.0, 2.0, 7.0], dtype=np.float32)
    selfd = np.array([4.0, 0.0, 0.0, 10.0], dtype=np.float64)
    ratio, passed = oracle.ratio_filter(dist, selfd, 0.75)
    assert ratio[0] == 0.75 and np.isnan(ratio[1]) and np.isinf(ratio[2]) and ratio[3] == 0.7
    assert passed.tolist() == [False, False, False, True]      # 0.75 < 0.75 is False
    # float64, not float32: a ratio that differs from tau only beyond float32 precision
    d = np.float32(0.7) * np.float32(3.0)
    r, p = oracle.ratio_filter(np.array([d], np.float32), np.array([3.0]), 0.7)
    assert r[0] == float(d) / 3.0 and bool(p[0]) == (float(d) / 3.0 < 0.7)
    # qrows indirection
    r, p = oracle.ratio_filter(np.array([5.0], np.float32), np.array([1.0, 10.0]), 0.7, qrows=[1])
    assert r[0] == 0.5 and p[0]


def test_accumulation_orders_agree_on_integer_valued_input():
    rng = np.random.default_rng(3)
    Q = rng.integers(0, 256, (64, 128)).astype(np.float32)
    T = rng.integers(0, 256, (80, 128)).astype(np.float32)
    a = oracle.bf_knn(Q, T, 2, order=0)
    b = oracle.bf_knn(Q, T, 2, order=1)
    c = oracle.bf_knn(Q.astype(np.uint8), T.astype(np.uint8), 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


def test_against_numpy_bruteforce():
    rng = np.random.default_rng(11)
    Q = rng.integers(0, 256, (150, 128), dtype=np.uint8)
    T = rng.integers(0, 256, (97, 128), dtype=np.uint8)
    T[5] = T[9]                                     # duplicate train rows: index tie-break
    Q[3] = T[9]
    d2 = ((Q[:, None, :].astype(np.int64) - T[None].astype(np.int64)) ** 2).sum(-1)
    idx, dist = oracle.bf_knn(Q, T, 2)
    order = np.argsort(d2, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order)
    assert np.array_equal(dist, np.sqrt(np.take_along_axis(d2, order, 1).astype(np.float32)))
    assert idx[3].tolist() == [5, 9]
    tidx, xd = oracle.bf_xcheck1(Q, T)
    rq = np.argmin(d2, axis=0)                      # first minimum = lowest q
    exp = np.full(len(Q), -1)
    best = np.full(len(Q), np.inf)
    for t in range(len(T)):
        if d2[rq[t], t] < best[rq[t]]:
            best[rq[t]], exp[rq[t]] = d2[rq[t], t], t
    assert np.array_equal(tidx, exp)
    assert np.array_equal(xd[exp >= 0], np.sqrt(best[exp >= 0].astype(np.float32)))


def test_dtype_mismatch_raises():
    with pytest.raises(TypeError):
        oracle.bf_xcheck1(np.zeros((2, 4), np.uint8), np.zeros((2, 4), np.float32))


def test_sqrt_tie_table():
    """The facts the device's tie repair rests on (tile_ops.h kSqrtTieMin): below 4 197 200 no two
    integers share a float32 square root; up to 128 * 255^2 no three do."""
    n = np.arange(0, 128 * 255 * 255 + 2, dtype=np.int64)
    bits = np.sqrt(n.astype(np.float32)).view(np.uint32)
    same = bits[1:] == bits[:-1]
    assert int(n[:-1][same][0]) == SQRT_TIE_MIN
    assert not np.any(same[1:] & same[:-1])
    assert np.all(np.diff(bits.astype(np.int64)) >= 0)          # the root is monotone: ties are runs
    for target in (SQRT_TIE_MIN, SQRT_TIE_MIN + 1, 8323200, 1, 0):
        r = row_with_sumsq(target).astype(np.int64)
        assert int((r * r).sum()) == target


def test_far_banks_separate_float_order_from_integer_order():
    """The tie-heavy banks of the GPU parity tests really tell the two orders apart: on them the
    oracle (OpenCV's float32 order) and an integer-d2 order disagree for many rows."""
    rng = np.random.default_rng(1)
    Q, T = far_banks(300, 200, rng)
    d2 = ((Q[:, None].astype(np.int64) - T[None].astype(np.int64)) ** 2).sum(-1)
    assert d2.min() >= SQRT_TIE_MIN
    idx, _ = oracle.bf_knn(Q, T, 2)
    by_int = np.argsort(d2, axis=1, kind="stable")[:, :2]
    assert (idx != by_int).any(axis=1).sum() > 20
    # and the oracle is the float32 order: stable sort on the float32 roots
    by_f32 = np.argsort(np.sqrt(d2.astype(np.float32)), axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, by_f32)


def test_vectorised_baseline_equals_the_faithful_scan():
    """oracle.bf_xcheck1_simd (bench.py's second CPU baseline: int16 differences + vpmaddwd, eight output rows per pass)
    returns what the one-pair-at-a-time restatement returns, bit for bit -- sizes around the block of eight, dim < 128,
    duplicates, empty banks and the float32-root tie range."""
    import oracle as orc
    if not orc._lib().orc_have_simd():
        pytest.skip("liboracle.so was built without AVX2")
    rng = np.random.default_rng(31)

    def same(Q, T):
        a, b = orc.bf_xcheck1(Q, T), orc.bf_xcheck1_simd(Q, T)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    for nq, nt, dim in [(1, 1, 128), (7, 9, 128), (8, 8, 128), (9, 17, 128), (300, 200, 128), (200, 300, 100), (50, 64, 3), (0, 5, 128), (5, 0, 128)]:
        Q = rng.integers(0, 256, (nq, dim), dtype=np.uint8)
        T = rng.integers(0, 256, (nt, dim), dtype=np.uint8)
        if nq > 20 and nt > 20:
            T[3] = Q[5]; T[11] = Q[5]; Q[17] = Q[5]
        same(Q, T)
    same(*far_banks(300, 200, rng))


def test_blocked_vnni_baseline_equals_the_faithful_scan():
    """oracle.bf_xcheck1_blocked (bench.py's third CPU baseline: vpdpbusd on sign-flipped bytes, 64 output rows x 6 candidates
    per register block, integer pre-test + float32-root compare per lane) returns what the one-pair-at-a-time restatement
    returns, bit for bit -- sizes around the block edges (64 rows, 6 candidates), dims that are not multiples of 4,
    duplicates, empty banks, the float32-root tie range, descriptor values 0 and 255 (the sign flip's corners)."""
    import oracle as orc
    if not orc.have_vnni():
        pytest.skip("this host has no AVX-512 VNNI")
    rng = np.random.default_rng(37)

    def same(Q, T):
        a, b = orc.bf_xcheck1(Q, T), orc.bf_xcheck1_blocked(Q, T)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    for nq, nt, dim in [(1, 1, 128), (5, 6, 128), (6, 64, 128), (7, 65, 128), (13, 63, 127), (300, 200, 128), (200, 300, 100),
                        (50, 64, 3), (129, 13, 1), (0, 5, 128), (5, 0, 128), (1000, 700, 128)]:
        Q = rng.integers(0, 256, (nq, dim), dtype=np.uint8)
        T = rng.integers(0, 256, (nt, dim), dtype=np.uint8)
        if nq > 20 and nt > 20:
            T[3] = Q[5]; T[11] = Q[5]; Q[17] = Q[5]
        same(Q, T)
    same(*far_banks(300, 200, rng))
    ext = rng.choice(np.array([0, 255], dtype=np.uint8), size=(90, 128))
    same(ext, ext[::-1].copy())
    same(np.full((70, 128), 255, np.uint8), np.zeros((9, 128), np.uint8))       # the largest d2 there is


def test_oracle_c_code_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """oracle/bfmatch_oracle.c compiled with -fsanitize=address,undefined into tests/tools/oracle_sanitize.c: 600 random
    shapes (0 / 1 rows, k = 2 against one row, dim 1 .. 128, exact-size heap blocks), the four float32 accumulation
    orders, the vectorised cross-check against the scalar one -- no report."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.check_call(["gcc", "-O1", "-g", "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", os.path.join(root, "oracle", "bfmatch_oracle.c"),
                           os.path.join(root, "tests", "tools", "oracle_sanitize.c"), "-lm", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), p.stdout + p.stderr
