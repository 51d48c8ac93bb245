"""The bench's 12-pair launch under one K1 workgroup order, for rocprofv3 counter passes
(scripts/profile_k1_xcd.sh):  python3 gpu_k1_order_run.py <order> [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastmatch_amd as fm
from fastmatch_amd import synth

order, launches = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4       # "2" or "bound_every=4,k1_order=0"
NP = 12
ctx = fm.Context(0)
ctx.set_option("batch_group", 16)
ctx.set_option("batch_tail", 0)
for kv in (order if "=" in order else "k1_order=%s" % order).split(","):
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
Q, T, _ = synth.planted_pair(100000, 100000, 20250002)
rng = np.random.default_rng(1)
banks = []
for j in range(NP):
    pq, pt = rng.permutation(100000), rng.permutation(100000)
    banks.append((ctx.bank(np.ascontiguousarray(np.roll(Q[pq], 8 * j, axis=1))), ctx.bank(np.ascontiguousarray(np.roll(T[pt], 8 * j, axis=1)))))
ctx.self_dist_batch([q for q, _ in banks], want_host=False)
outs = [tuple(ctx.pinned_empty(100000, dt) for dt in (np.int32, np.int32, np.float32, np.float64)) for _ in range(NP)]
cnts = [ctx.pinned_empty(1, np.int64) for _ in range(NP)]
args = ctx.prepare_batch(banks, outs, cnts)
for _ in range(launches):
    ctx.match_accepted_batch(args, 0.7)
    ctx.sync()
print("order", order, "accepted", int(sum(int(c[0]) for c in cnts)))
