"""Soak of the enqueue-only entry points: many iterations of the per-pair async call, the batch call and
the device-rows batch call over the same pairs, EVERY output of EVERY iteration compared with the
synchronous call's (workspace slots, bound / qbest re-arming and stream ordering under reuse)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastmatch_amd as fm
from fastmatch_amd import synth, sharding

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ctx = fm.Context(0)
dev = torch.device("cuda", 0)
shapes = [(6000, 33000)] * 7 + [(2500, 40000)] * 2 + [(3000, 2000)]
pairs, want = [], []
for k, (nq, nt) in enumerate(shapes):
    Q, T, _ = synth.planted_pair(nq, nt, seed=500 + k)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    qb.set_selfdist(ctx.self_dist(qb))
    pairs.append((qb, tb))
    want.append(ctx.match_accepted(qb, tb, 0.75))
cap = 6000
n = len(pairs)
outs = [(ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.int32), ctx.pinned_empty(cap, np.float32),
         ctx.pinned_empty(cap, np.float64)) for _ in pairs]
counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
block = ctx.prepare_batch(pairs, outs, counts)
pblock = ctx.prepare_pairs(pairs)
rows = torch.zeros((n, cap, 3), dtype=torch.int32, device=dev)
cnts = torch.zeros(n, dtype=torch.int64, device=dev)
hc = ctx.pinned_empty(n, np.int64)
packed = [sharding.pack_matches(w[0], w[1], w[2]) for w in want]


def check_host(tag, it):
    for j, (qa, ta, da, ra) in enumerate(want):
        m = int(counts[j][0])
        ok = (m == len(qa) and np.array_equal(outs[j][0][:m], qa) and np.array_equal(outs[j][1][:m], ta)
              and np.array_equal(outs[j][2][:m], da) and np.array_equal(outs[j][3][:m], ra))
        if not ok:
            raise SystemExit("MISMATCH %s iteration %d pair %d (count %d, expected %d)" % (tag, it, j, m, len(qa)))


t0 = time.perf_counter()
for it in range(iters):
    mode = it % 3
    for c in counts:
        c[0] = -1
    if mode == 0:
        for j, (qb, tb) in enumerate(pairs):
            ctx.match_accepted_async(qb, tb, 0.75, outs[j], counts[j])
        ctx.sync()
        check_host("async", it)
    elif mode == 1:
        ctx.match_accepted_batch(block, 0.75)
        ctx.sync()
        check_host("batch", it)
    else:
        rows.fill_(-1)
        torch.cuda.synchronize()
        ctx.match_accepted_dev_batch(pblock, 0.75, rows.data_ptr(), cnts.data_ptr(), cap, h_counts=hc,
                                     consumer_stream=torch.cuda.current_stream().cuda_stream)
        got_rows, got_cnt = rows.cpu().numpy(), cnts.cpu().numpy()       # (on the consumer stream: ordered behind the fills)
        ctx.sync()
        for j, w in enumerate(packed):
            if int(got_cnt[j]) != len(w) or int(hc[j]) != len(w) or not np.array_equal(got_rows[j, :len(w)], w):
                raise SystemExit("MISMATCH dev_batch iteration %d pair %d" % (it, j))
print("stress ok: %d iterations x %d pairs in %.1f s" % (iters, n, time.perf_counter() - t0))
