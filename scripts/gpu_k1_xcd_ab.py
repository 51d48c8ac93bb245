"""K1 under the three workgroup -> (output chunk, split) orders (option "k1_order", rowreduce.hip map_block),
A/B-ed in ONE process on one box, interleaved: 0 = split major (r01-r03), 1 = the workgroups of an XCD own
output chunks for all splits, 2 = an XCD owns a contiguous share of the split-major order.
  - the bench's launch: 12 image pairs (100k x 100k, distinct banks) in one rowreduce_batch_kernel launch, steps
    pipelined two deep -- distance-kernel ms per image pair (HIP events on the library's stream) and wall per pair;
  - one pair per launch (fm_xcheck1) and the masked self sweep (fm_self_dist);
  - with a library built with -DFM_CLOCK_STAMP (make FLAGS_rowreduce=-DFM_CLOCK_STAMP): the in-kernel clock,
    100 MHz x s_memtime / s_memrealtime around the stage loop, median over the workgroups of the last launch.
Results are compared between the orders (must be identical)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth, _ffi

NP = int(os.environ.get("FM_AB_PAIRS", "12"))
ROUNDS = int(os.environ.get("FM_AB_ROUNDS", "5"))
# variants: a bare number = k1_order, or option lists "bound_every=4,k1_order=2" (fm_ctx_set_option names)
ORDERS = [x if "=" in x else "k1_order=%s" % x for x in (sys.argv[1:] or ["0", "1", "2"])]
BASE = {"k1_order": 0, "bound_every": 1}
ctx = fm.Context(0)
ctx.set_option("batch_group", 16)
ctx.set_option("batch_tail", 0)
raw = getattr(_ffi.load_library(), "_lib", _ffi.load_library())
have_clock = hasattr(raw, "fm_debug_clock")
Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
rng = np.random.default_rng(1)
banks = []
for j in range(NP):
    pq, pt = rng.permutation(100000), rng.permutation(100000)
    qb, tb = ctx.bank(np.ascontiguousarray(np.roll(Q[pq], 8 * j, axis=1))), ctx.bank(np.ascontiguousarray(np.roll(T[pt], 8 * j, axis=1)))
    banks.append((qb, tb))
ctx.self_dist_batch([q for q, _ in banks], want_host=False)
ctx.sync()
sets = []
for _ in range(2):
    outs = [tuple(ctx.pinned_empty(100000, dt) for dt in (np.int32, np.int32, np.float32, np.float64)) for _ in range(NP)]
    cnts = [ctx.pinned_empty(1, np.int64) for _ in range(NP)]
    sets.append((ctx.prepare_batch(banks, outs, cnts), outs, cnts))


def clock_mhz():
    if not have_clock:
        return None
    buf = (ctypes.c_ulonglong * (2 * 8192))()
    raw.fm_debug_clock(buf, 8192)
    a = np.array(buf[:], dtype=np.float64).reshape(-1, 2)
    a = a[(a[:, 1] > 0)]
    return float(np.median(100.0 * a[:, 0] / a[:, 1])) if len(a) else None


def batch_run(steps=8):
    ctx.sync()
    ctx.reset_stats()
    t0 = time.perf_counter()
    prev = None
    for i in range(steps):
        ctx.match_accepted_batch(sets[i % 2][0], 0.7)
        tk = ctx.mark()
        if prev is not None:
            ctx.wait(prev)
        prev = tk
    ctx.sync()
    wall = (time.perf_counter() - t0) / steps / NP * 1e3
    st = ctx.stats()
    return st["kernel_ms"] / (st["pairs"] / 1e10), wall


res = {o: {"batch_k": [], "batch_wall": [], "k1": [], "self": [], "clk": []} for o in ORDERS}
ref = None
for rnd in range(ROUNDS + 1):
    for o in ORDERS:
        for k, v in BASE.items():
            ctx.set_option(k, v)
        for kv in o.split(","):
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        k, w = batch_run()
        clk = clock_mhz()
        qb, tb = banks[0]
        ctx.xcheck1(qb, tb)
        ctx.reset_stats()
        for _ in range(6):
            x = ctx.xcheck1(qb, tb)
        st = ctx.stats()
        k1 = st["kernel_ms"] / st["kernel_launches"]
        ctx.reset_stats()
        for _ in range(6):
            sd = ctx.self_dist(qb)
        st = ctx.stats()
        ks = st["kernel_ms"] / st["kernel_launches"]
        got = (x[0], x[1], sd, sets[0][1][0][0][:int(sets[0][2][0][0])].copy())
        if ref is None:
            ref = got
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), "results differ between orders"
        if rnd:                                   # round 0 warms up
            r = res[o]
            r["batch_k"].append(k); r["batch_wall"].append(w); r["k1"].append(k1); r["self"].append(ks); r["clk"].append(clk)
for o in ORDERS:
    r = res[o]
    med = lambda v: float(np.median(v))
    print("%s: batched launch %.4f ms kernel / %.4f ms wall per image pair (min %.4f / %.4f) | one pair per launch %.4f ms | "
          "self sweep %.4f ms | in-kernel clock %s MHz"
          % (o, med(r["batch_k"]), med(r["batch_wall"]), min(r["batch_k"]), min(r["batch_wall"]), med(r["k1"]), med(r["self"]),
             ("%.0f" % med([c for c in r["clk"] if c])) if any(r["clk"]) else "n/a (build with -DFM_CLOCK_STAMP)"), flush=True)
