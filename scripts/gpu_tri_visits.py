"""Triangular self sweep: how often the exact paths run (library built with -DFM_COUNT_VISITS, see the shell line in
the doc string of scripts/gpu_tri_visits.sh).  Counters: [0] row-direction visits (wave, block), [1] column-direction
second-level visits, [2] column fires (lanes), [128] units x blocks swept."""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, _ffi

ctx = fm.Context(0)
lib = _ffi.load_library()
raw = getattr(lib, "_lib", lib)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = synth.synth_sift(N, np.random.default_rng(7))
bank = ctx.bank(D)
buf = (ctypes.c_ulonglong * 256)()
for st in [int(a) for a in sys.argv[2:]] or [0, 32, 61]:
    ctx.set_option("self_tri", 1)
    ctx.set_option("tri_stages", st)
    ctx.self_dist(bank)
    raw.fm_debug_visits(None, 1)
    t0 = time.perf_counter()
    ctx.self_dist(bank)
    dt = time.perf_counter() - t0
    raw.fm_debug_visits(buf, 1)
    v = np.array(buf[:], dtype=np.float64)
    print("S", st, "ms %.2f" % (dt * 1e3), "units x blocks %.4g" % v[128], "row visits %.4g (%.4f)" % (v[0], v[0] / v[128]),
          "col visits %.4g (%.4f)" % (v[1], v[1] / v[128]), "col fires %.4g (%.2f per row)" % (v[2], v[2] / N))
