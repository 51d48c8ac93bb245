import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth
ctx = fm.Context(0)
rng = np.random.default_rng(20250005)
NT, NQ = int(os.environ.get("NT", 1000000)), 10000
T = synth.synth_sift(NT, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (NT, 128)).astype(np.float32)
tb = ctx.bank(T)
print("bank kind", tb.kind)
for b in range(3):
    Q = synth.synth_sift(NQ, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (NQ, 128)).astype(np.float32)
    qb = ctx.bank(Q)
    ctx.reset_stats()
    t0 = time.perf_counter(); idx, dist = ctx.knn2(qb, tb); t1 = time.perf_counter()
    s = ctx.stats()
    print("f32 knn2 %d x %d: kernel %.2f ms wall %.2f ms -> %.3e pairs/s (%.1f%% of 3.07e11 fp32 VALU bound)" % (
        NQ, NT, s["kernel_ms"], 1e3 * (t1 - t0), NQ * NT / s["kernel_ms"] * 1e3, 100 * NQ * NT / s["kernel_ms"] * 1e3 / 3.07e11))
    ctx.reset_stats()
    tidx, xd = ctx.xcheck1(qb, tb)
    s = ctx.stats()
    print("f32 xcheck %d x %d: kernel %.2f ms -> %.3e pairs/s" % (NQ, NT, s["kernel_ms"], NQ * NT / s["kernel_ms"] * 1e3))
# same shape on the int8 route for comparison
Ti = synth.synth_sift(NT, rng); Qi = synth.synth_sift(NQ, rng)
tbi, qbi = ctx.bank(Ti), ctx.bank(Qi)
for _ in range(2):
    ctx.reset_stats(); ctx.knn2(qbi, tbi); s = ctx.stats()
    print("int8 knn2 %d x %d: kernel %.3f ms -> %.3e pairs/s" % (NQ, NT, s["kernel_ms"], NQ * NT / s["kernel_ms"] * 1e3))
    ctx.reset_stats(); ctx.xcheck1(qbi, tbi); s = ctx.stats()
    print("int8 xcheck %d x %d: kernel %.3f ms -> %.3e pairs/s" % (NQ, NT, s["kernel_ms"], NQ * NT / s["kernel_ms"] * 1e3))
