"""GPU: no variable of the environment changes what the library computes.

The FM_* variables are read ONCE, when a context is created, and only seed its tuning options (launch shapes, refresh
intervals, diagnostics on stderr); the switches of the measurement builds (FM_TRI_NOCOL, FM_TRI_MERGE,
FM_ABLATE_KEEP_BOUNDS) exist in -DFM_ABLATE builds only.  This test sets every one of them to hostile values --
extremes of the accepted range, values outside it, garbage -- creates a fresh context and asserts that the known-answer
tests and the oracle comparisons stay bit-identical."""
import os
import re

import numpy as np
import pytest

import oracle
from fastmatch_amd import synth
from kat import xcheck_cases, knn2_cases, selfdist_case, sqrt_tie_knn2_cases, sqrt_tie_xcheck_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _env_names():
    """Every FM_* name the library's sources mention in a getenv() or in the option table."""
    names = set()
    csrc = os.path.join(ROOT, "fast-match_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            txt = open(os.path.join(csrc, f)).read()
            names.update(re.findall(r'getenv\("(FM_[A-Z0-9_]+)"\)', txt))
            names.update(re.findall(r'"(FM_[A-Z0-9_]+)"\}', txt))
    return sorted(names)


HOSTILE = [
    {"default": "1"},                                         # every switch "on", every number 1
    {"default": "0"},
    {"default": "-1"},
    {"default": "999999999999"},                              # outside every range
    {"default": "garbage", "FM_F32_DEBUG": "", "FM_EXPAND_DEBUG": ""},
    # extremes of the accepted ranges
    {"default": None, "FM_NB": "8", "FM_NSPLIT": "1048576", "FM_NW": "16", "FM_NBUF": "2", "FM_PRIO": "0", "FM_GLDS": "0", "FM_COOP": "0",
     "FM_F32_FILTER": "2", "FM_F32_NW": "8", "FM_F32_NSPLIT": "1048576", "FM_F32_FUSED": "0", "FM_F32_LPC": "64", "FM_F32_BOUND_EVERY": "64",
     "FM_BATCH_GROUP": "16", "FM_BATCH_TAIL": "16", "FM_K1_ORDER": "2", "FM_BOUND_EVERY": "1024", "FM_SELF_TRI": "2", "FM_TRI_STAGES": "4096",
     "FM_EXPAND_DELEGATE": "1", "FM_TRI_NOCOL": "1", "FM_TRI_MERGE": "1", "FM_ABLATE_KEEP_BOUNDS": "1", "FM_EXPAND_NO_BIG": "1"},
    {"default": None, "FM_NB": "4", "FM_NSPLIT": "3", "FM_NW": "4", "FM_NBUF": "3", "FM_F32_FILTER": "2", "FM_F32_NW": "4", "FM_F32_NSPLIT": "1",
     "FM_F32_FUSED": "1", "FM_F32_LPC": "1", "FM_F32_BOUND_EVERY": "1", "FM_K1_ORDER": "1", "FM_BOUND_EVERY": "1", "FM_SELF_TRI": "2",
     "FM_TRI_STAGES": "4", "FM_TRI_NOCOL": "1", "FM_TRI_MERGE": "1", "FM_REFILL_GRID": "1", "FM_ASYNC_TIME_EVERY": "1"},
]


@pytest.mark.parametrize("setting", range(len(HOSTILE)))
def test_results_under_a_hostile_environment(monkeypatch, capfd, setting):
    import fastmatch_amd
    env = HOSTILE[setting]
    names = _env_names()
    assert "FM_TRI_NOCOL" in names and "FM_F32_BOUND_EVERY" in names and len(names) >= 25
    for n in names:
        v = env.get(n, env["default"])
        if v is not None:
            monkeypatch.setenv(n, v)
    c = fastmatch_amd.Context(0)
    try:
        # known-answer tests, integer and float32 route
        for _, Q, T, etidx, edist in xcheck_cases() + sqrt_tie_xcheck_cases():
            for cast in (None, np.float32):
                q, t = (Q, T) if cast is None else (Q.astype(cast), T.astype(cast))
                tidx, dist = c.xcheck1(c.bank(q), c.bank(t))
                assert tidx.tolist() == etidx and _eq(dist, np.array(edist, dtype=np.float32))
        for _, Q, T, eidx, edist in knn2_cases() + sqrt_tie_knn2_cases():
            idx, dist = c.knn2(c.bank(Q), c.bank(T))
            assert idx.tolist() == eidx and _eq(dist, np.array(edist, dtype=np.float32))
        D, exp = selfdist_case()
        assert c.self_dist(c.bank(D)).tolist() == exp
        # against the oracle: planted integer pair (self distances by whichever sweep the environment asks for,
        # accepted matches), non-integer float32 pair
        Q, T, _ = synth.planted_pair(2500, 36000, seed=5 + setting)
        qb, tb = c.bank(Q), c.bank(T)
        oi, od = oracle.bf_knn(Q, T, 2)
        idx, dist = c.knn2(qb, tb)
        assert _eq(idx, oi) and _eq(dist, od)
        ot, ox = oracle.bf_xcheck1(Q, T)
        tidx, xd = c.xcheck1(qb, tb)
        assert _eq(tidx, ot) and _eq(xd, ox)
        assert _eq(c.self_dist(tb), oracle.self_dist(T))
        assert _eq(c.self_dist(qb), oracle.self_dist(Q))
        rng = np.random.default_rng(77 + setting)
        Qf = (Q[:900] + rng.uniform(-0.5, 0.5, (900, 128))).astype(np.float32)
        Tf = (T[:7000] + rng.uniform(-0.5, 0.5, (7000, 128))).astype(np.float32)
        qf, tf = c.bank(Qf), c.bank(Tf)
        oi, od = oracle.bf_knn(Qf, Tf, 2, order=1)
        idx, dist = c.knn2(qf, tf)
        assert _eq(idx, oi) and _eq(dist, od)
        ot, ox = oracle.bf_xcheck1(Qf, Tf, order=1)
        tidx, xd = c.xcheck1(qf, tf)
        assert _eq(tidx, ot) and _eq(xd, ox)
        assert _eq(c.self_dist(tf), oracle.self_dist(Tf, order=1))
    finally:
        c.close()
