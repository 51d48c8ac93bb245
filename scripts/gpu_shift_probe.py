"""Does the MFMA stream clock higher when the int8 operands are small in magnitude?
Times X1 on the same descriptors shifted by a constant (L2 is shift invariant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth
ctx = fm.Context(0)
Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
print("data max", Q.max(), T.max(), "mean %.1f" % Q.mean(), "p99", np.percentile(Q, 99))
ref = None
for shift in (0, 64, 96, 112, 128):
    Qs = np.clip(Q.astype(np.int32) + shift, 0, 255).astype(np.uint8)
    Ts = np.clip(T.astype(np.int32) + shift, 0, 255).astype(np.uint8)
    clipped = int((Q.astype(np.int32) + shift > 255).sum())
    qb, tb = ctx.bank(Qs), ctx.bank(Ts)
    ts = []
    for _ in range(8):
        ctx.reset_stats(); out = ctx.xcheck1(qb, tb); ts.append(ctx.stats()["kernel_ms"])
    print("shift %3d (int8 mean %.0f, clipped %d): kernel min %.3f med %.3f ms" % (shift, Qs.mean() - 128, clipped, min(ts), float(np.median(ts))), flush=True)
