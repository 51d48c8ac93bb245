"""Property-based tests (hypothesis): the oracle's invariants on the CPU, and HIP-vs-oracle
parity on randomly shaped, tie-heavy inputs on the GPU."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

import oracle


def _bank(rng, n, dim, levels):
    vals = rng.integers(0, 256, size=levels)
    return vals[rng.integers(0, levels, size=(n, dim))].astype(np.uint8)


shape = st.tuples(st.integers(0, 70), st.integers(0, 70), st.integers(1, 128), st.integers(1, 6), st.integers(0, 2**31 - 1))


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(shape)
def test_oracle_shift_invariance_and_gemm_form(s):
    """L2 is invariant under a common shift of both banks (the XOR 0x80 trick of the device
    path), and the int8 GEMM form |a'|^2 + |b'|^2 - 2 a'.b' equals the direct form exactly."""
    nq, nt, dim, levels, seed = s
    rng = np.random.default_rng(seed)
    Q, T = _bank(rng, nq, dim, levels) // 2, _bank(rng, nt, dim, levels) // 2       # leave room for the shift
    a = oracle.bf_knn(Q, T, 2)
    b = oracle.bf_knn((Q + 100).astype(np.uint8), (T + 100).astype(np.uint8), 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if nq and nt:
        qi = (Q.astype(np.int64) ^ 0x80).astype(np.uint8).view(np.int8).astype(np.int64)
        ti = (T.astype(np.int64) ^ 0x80).astype(np.uint8).view(np.int8).astype(np.int64)
        gemm = (qi * qi).sum(1)[:, None] + (ti * ti).sum(1)[None, :] - 2 * qi @ ti.T
        direct = ((Q[:, None, :].astype(np.int64) - T[None].astype(np.int64)) ** 2).sum(-1)
        assert np.array_equal(gemm, direct) and gemm.max() <= 8323200


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(shape)
def test_oracle_permutation_equivariance(s):
    """Permuting the train rows permutes the reported indices unless a tie is involved, and
    never changes the reported distances."""
    nq, nt, dim, levels, seed = s
    rng = np.random.default_rng(seed)
    Q, T = _bank(rng, nq, dim, 256), _bank(rng, nt, dim, 256)
    perm = rng.permutation(nt)
    a = oracle.bf_knn(Q, T, 2)
    b = oracle.bf_knn(Q, T[perm], 2)
    assert np.array_equal(a[1], b[1])
    ok = (b[0] >= 0) & (a[1] != np.roll(a[1], 1, axis=1))          # rows without a 1st/2nd tie
    ok &= (a[1][:, :1] != a[1][:, 1:])
    mapped = np.where(b[0] >= 0, perm[np.clip(b[0], 0, max(nt - 1, 0))] if nt else b[0], -1)
    d2 = ((Q[:, None, :].astype(np.int64) - T[None].astype(np.int64)) ** 2).sum(-1) if nq and nt else np.zeros((nq, nt))
    for qi in range(nq):
        for k in range(2):
            if b[0][qi, k] >= 0:                                    # same distance class as reported
                assert np.sqrt(np.float32(d2[qi, mapped[qi, k]])) == b[1][qi, k]


@pytest.mark.gpu
@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(st.tuples(st.integers(0, 300), st.integers(0, 300), st.sampled_from([1, 2, 7, 31, 64, 127, 128]),
                 st.integers(1, 5), st.booleans(), st.integers(0, 2**31 - 1)))
def test_hip_equals_oracle_on_random_tie_heavy_inputs(ctx, s):
    nq, nt, dim, levels, as_f32, seed = s
    rng = np.random.default_rng(seed)
    Q, T = _bank(rng, nq, dim, levels), _bank(rng, nt, dim, levels)
    if as_f32:
        Q, T = Q.astype(np.float32), T.astype(np.float32)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    idx, dist = ctx.knn2(qb, tb)
    oidx, odist = oracle.bf_knn(Q, T, 2)
    assert np.array_equal(idx, oidx) and np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
    tidx, xd = ctx.xcheck1(qb, tb)
    otidx, oxd = oracle.bf_xcheck1(Q, T)
    assert np.array_equal(tidx, otidx) and np.array_equal(xd.view(np.uint32), oxd.view(np.uint32))
    if nq and nt:
        rows = rng.integers(0, nq, size=min(nq, 40)).astype(np.int32)             # repeats allowed
        lo = int(rng.integers(0, nt))
        bt, bd, _ = ctx.xcheck1_batched(qb, rows, [0, len(rows)], tb, [lo, nt])
        et, ed = oracle.bf_xcheck1(Q[rows], T[lo:nt])
        assert np.array_equal(bt, et) and np.array_equal(bd.view(np.uint32), ed.view(np.uint32))


@pytest.mark.gpu
@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(st.tuples(st.integers(1, 300), st.integers(2, 300), st.integers(1, 12), st.integers(1, 3),
                 st.integers(4197200, 6400000), st.booleans(), st.integers(0, 2**31 - 1)))
def test_hip_equals_oracle_in_the_sqrt_tie_range(ctx, s):
    """d2 >= 4 197 200: OpenCV orders by the float32 root, then index.  Banks whose every distance lies
    in that range (a handful of d2 values, again and again), plus a planted pair of train rows at
    d2 = n and n + 1 from a zero query row, larger d2 first or second."""
    from kat import far_banks, row_with_sumsq
    nq, nt, small_dims, small_max, n, larger_first, seed = s
    rng = np.random.default_rng(seed)
    Q, T = far_banks(nq, nt, rng, small_dims, small_max)
    Q[0] = 0
    i, j = sorted(rng.choice(nt, 2, replace=False))
    T[i], T[j] = (row_with_sumsq(n + 1), row_with_sumsq(n)) if larger_first else (row_with_sumsq(n), row_with_sumsq(n + 1))
    qb, tb = ctx.bank(Q), ctx.bank(T)
    idx, dist = ctx.knn2(qb, tb)
    oidx, odist = oracle.bf_knn(Q, T, 2)
    assert np.array_equal(idx, oidx) and np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
    if np.sqrt(np.float32(n)) == np.sqrt(np.float32(n + 1)):
        assert idx[0].tolist() == [i, j]                          # index order inside the tie, whichever d2 is larger
    tidx, xd = ctx.xcheck1(qb, tb)
    otidx, oxd = oracle.bf_xcheck1(Q, T)
    assert np.array_equal(tidx, otidx) and np.array_equal(xd.view(np.uint32), oxd.view(np.uint32))
    tidx, xd = ctx.xcheck1(tb, qb)                                # roles swapped: the election sees the planted pair
    otidx, oxd = oracle.bf_xcheck1(T, Q)
    assert np.array_equal(tidx, otidx) and np.array_equal(xd.view(np.uint32), oxd.view(np.uint32))
    rows = rng.integers(0, nq, size=min(nq, 40)).astype(np.int32)
    lo = int(rng.integers(0, nt))
    bt, bd, _ = ctx.xcheck1_batched(qb, rows, [0, len(rows)], tb, [lo, nt])
    et, ed = oracle.bf_xcheck1(Q[rows], T[lo:nt])
    assert np.array_equal(bt, et) and np.array_equal(bd.view(np.uint32), ed.view(np.uint32))


@pytest.fixture(scope="module")
def filter_ctx_module():
    import os
    import fastmatch_amd
    old = os.environ.get("FM_F32_FILTER")
    os.environ["FM_F32_FILTER"] = "2"                      # the fp16-MFMA filter for every float32 call
    c = fastmatch_amd.Context(0)
    yield c
    c.close()
    if old is None:
        del os.environ["FM_F32_FILTER"]
    else:
        os.environ["FM_F32_FILTER"] = old


@pytest.mark.gpu
@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(st.tuples(st.integers(1, 400), st.integers(1, 400), st.sampled_from([1, 3, 32, 100, 128]),
                 st.sampled_from(["gauss", "levels", "twins"]), st.integers(-40, 40), st.integers(-3, 3),
                 st.integers(0, 2**31 - 1)))
def test_float32_filter_route_equals_oracle_on_random_inputs(filter_ctx_module, s):
    """Non-integer float32 banks of random shape, magnitude (2^-40 .. 2^40, the two banks up to
    2^3 apart), and tie structure: the filter route (rescoring, per-row rescan or whole-call
    redo, whichever a case needs) returns the oracle's bits."""
    c = filter_ctx_module
    nq, nt, dim, kind, e, de, seed = s
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        Q, T = rng.normal(0, 1, (nq, dim)), rng.normal(0, 1, (nt, dim))
    elif kind == "levels":                                  # few distinct non-integer values: exact ties everywhere
        lv = rng.normal(0, 1, 4)
        Q, T = lv[rng.integers(0, 4, (nq, dim))], lv[rng.integers(0, 4, (nt, dim))]
    else:                                                   # train rows that are tiny perturbations of query rows
        Q = rng.normal(0, 1, (nq, dim))
        T = Q[rng.integers(0, nq, nt)] + rng.normal(0, 1e-3, (nt, dim))
    Q = np.ldexp(Q, e).astype(np.float32)
    T = np.ldexp(T, e + de).astype(np.float32)
    Q[0, 0] += np.float32(np.ldexp(0.37, e))                # (never integer valued)
    T[0, 0] += np.float32(np.ldexp(0.37, e + de))
    qb, tb = c.bank(Q), c.bank(T)
    if qb.kind != tb.kind:
        return                                              # an all-integer bank by accident: not this route
    idx, dist = c.knn2(qb, tb)
    oidx, odist = oracle.bf_knn(Q, T, 2, order=1)
    assert np.array_equal(idx, oidx) and np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
    tidx, xd = c.xcheck1(qb, tb)
    otidx, oxd = oracle.bf_xcheck1(Q, T, order=1)
    assert np.array_equal(tidx, otidx) and np.array_equal(xd.view(np.uint32), oxd.view(np.uint32))
