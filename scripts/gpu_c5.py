"""BASELINE config 5 (1M-row float32 target bank vs 10k-row query batches): K8 timings, for rocprofv3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastmatch_amd
from fastmatch_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
NQ = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
rng = np.random.default_rng(20250005)
ctx = fastmatch_amd.Context(0)
T = synth.synth_sift(NT, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (NT, 128)).astype(np.float32)
tb = ctx.bank(T)
Q = synth.synth_sift(NQ, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (NQ, 128)).astype(np.float32)
qb = ctx.bank(Q)
for name, fn in (("knn2", lambda: ctx.knn2(qb, tb)), ("xcheck1", lambda: ctx.xcheck1(qb, tb))):
    fn()
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    w = (time.perf_counter() - t0) / reps
    st = ctx.stats()
    k = st["kernel_ms"] / st["kernel_launches"]
    print("%s %d x %d: kernel %.3f ms (%.3e pairs/s, %.1f%% of the fp16 MFMA peak at 256 flop/pair), call %.3f ms on the stream, wall %.3f ms, filter %s"
          % (name, NQ, NT, k, NQ * NT / (k * 1e-3), 100 * NQ * NT * 256 / (k * 1e-3) / 2.5e15, st["total_ms"] / st["calls"], w * 1e3,
             ctx.f32_filter_stats()), flush=True)
