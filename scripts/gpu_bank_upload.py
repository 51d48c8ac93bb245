import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth
ctx = fm.Context(0)
rng = np.random.default_rng(1)
A = synth.synth_sift(100000, rng)
F = synth.synth_sift(1000000, rng).astype(np.float32)
for name, arr in (("u8 100k", A), ("f32 int-valued 1M", F), ("f32 non-int 1M", F + np.float32(0.25))):
    b = ctx.bank(arr); b.close()
    t0 = time.perf_counter()
    for _ in range(3):
        b = ctx.bank(arr); b.close()
    print(name, "%.2f ms per bank (incl. H2D copy)" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
