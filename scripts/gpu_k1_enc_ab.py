"""K1 under two int8 encodings of the SAME data, interleaved in one process (library built by scripts/gpu_k1_enc_ab.sh:
-DFM_ENC_AB in api_ctx.hip, -DFM_CLOCK_STAMP in rowreduce.hip): shift 128 (the product's u ^ 0x80) against shift 0, which is
exact for rows whose bytes are <= 127 -- SIFT-like rows clipped there.  Per variant: distance-kernel ms per 100k x 100k
pair in the bench's 12-pair launch, the in-kernel clock (100 MHz x s_memtime / s_memrealtime around the stage loop, median
over workgroups) and the accepted counts, which must agree."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, _ffi

NP, ROUNDS = 12, int(os.environ.get("FM_AB_ROUNDS", "5"))
ctx = fm.Context(0)
ctx.set_option("batch_group", 16)
ctx.set_option("batch_tail", 0)
raw = getattr(_ffi.load_library(), "_lib", _ffi.load_library())
assert hasattr(raw, "fm_debug_enc_shift"), "build with scripts/gpu_k1_enc_ab.sh"
Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
print("bytes > 127 before clipping: %.4f %% of Q, max %d" % (100.0 * (Q > 127).mean(), int(Q.max())))
Q, T = np.minimum(Q, 127), np.minimum(T, 127)
rng = np.random.default_rng(1)
perm = [(rng.permutation(100000), rng.permutation(100000)) for _ in range(NP)]
variants = {}
for shift in (128, 0):
    raw.fm_debug_enc_shift(shift)
    banks = []
    for j, (pq, pt) in enumerate(perm):
        banks.append((ctx.bank(np.ascontiguousarray(np.roll(Q[pq], 8 * j, axis=1))), ctx.bank(np.ascontiguousarray(np.roll(T[pt], 8 * j, axis=1)))))
    ctx.self_dist_batch([q for q, _ in banks], want_host=False)
    ctx.sync()
    sets = []
    for _ in range(2):
        outs = [tuple(ctx.pinned_empty(100000, dt) for dt in (np.int32, np.int32, np.float32, np.float64)) for _ in range(NP)]
        cnts = [ctx.pinned_empty(1, np.int64) for _ in range(NP)]
        sets.append((ctx.prepare_batch(banks, outs, cnts), outs, cnts))
    variants[shift] = (banks, sets)
raw.fm_debug_enc_shift(128)


def clock_mhz():
    buf = (ctypes.c_ulonglong * (2 * 8192))()
    raw.fm_debug_clock(buf, 8192)
    a = np.array(buf[:], dtype=np.float64).reshape(-1, 2)
    a = a[a[:, 1] > 0]
    return float(np.median(100.0 * a[:, 0] / a[:, 1])) if len(a) else float("nan")


def run(sets, steps=8):
    ctx.sync()
    ctx.reset_stats()
    prev = None
    for i in range(steps):
        ctx.match_accepted_batch(sets[i % 2][0], 0.7)
        tk = ctx.mark()
        if prev is not None:
            ctx.wait(prev)
        prev = tk
    ctx.sync()
    st = ctx.stats()
    return st["kernel_ms"] / (st["pairs"] / 1e10), clock_mhz(), [int(c[0]) for c in sets[(steps - 1) % 2][2]]


res = {128: [], 0: []}
counts = {}
for rnd in range(ROUNDS + 1):
    for shift in (128, 0):
        k, clk, cn = run(variants[shift][1])
        counts[shift] = cn
        if rnd:
            res[shift].append((k, clk))
assert counts[128] == counts[0], "the two encodings disagree"
for shift in (128, 0):
    ks, cl = [x[0] for x in res[shift]], [x[1] for x in res[shift]]
    print("shift %3d: kernel ms per pair %s  mean %.4f | in-kernel clock MHz %s  mean %.0f"
          % (shift, " ".join("%.4f" % x for x in ks), float(np.mean(ks)), " ".join("%.0f" % x for x in cl), float(np.mean(cl))))
print("accepted per pair (both): %s" % counts[0][:4])
print("shift 0 / shift 128 kernel time: %.4f" % (np.mean([x[0] for x in res[0]]) / np.mean([x[0] for x in res[128]])))
