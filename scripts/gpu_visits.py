"""Where the exact-path visits of K1 / K2 happen (build rowreduce.hip with -DFM_COUNT_VISITS):
visits per split index / (units x blocks) per split, 100k x 100k."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, _ffi

ctx = fm.Context(0)
lib = _ffi.load_library()
raw = getattr(lib, "_lib", lib)
Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
qb, tb = ctx.bank(Q), ctx.bank(T)
buf = (ctypes.c_ulonglong * 256)()
for name, fn in (("k1", lambda: ctx.xcheck1(qb, tb)), ("k2", lambda: ctx.knn2(qb, tb))):
    fn()
    raw.fm_debug_visits(None, 1)
    fn()
    raw.fm_debug_visits(buf, 1)
    v = np.array(buf[:64], dtype=np.float64); w = np.array(buf[64:128], dtype=np.float64); u = np.array(buf[128:], dtype=np.float64)
    n = int((u > 0).sum())
    print(name, "splits", n, "overall visit rate %.4f" % (v.sum() / u.sum()), "visits %.3g" % v.sum(),
          "lanes wanting %.3g (%.2f per visit)" % (w.sum(), w.sum() / max(v.sum(), 1)))
    print("  per split:", " ".join("%.3f" % (v[i] / u[i]) for i in range(n)))
    print("  lanes per visit, per split:", " ".join("%.1f" % (w[i] / max(v[i], 1)) for i in range(n)))
