"""GPU: accepted matches that stay on the device (fm_match_accepted_dev) and feed the multi-GPU
result gather without a host hop.  The rows must equal what fm_match_accepted returns to the
host (which tests/test_parity_gpu.py checks against the oracle)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from fastmatch_amd import synth, sharding, _ffi
import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _banks(ctx, nq, nt, seed):
    Q, T, _ = synth.planted_pair(nq, nt, seed)
    qb, tb = ctx.bank(Q), ctx.bank(T)
    qb.set_selfdist(ctx.self_dist(qb))
    return Q, T, qb, tb


@pytest.mark.parametrize("nq,nt", [(3000, 2500), (257, 9000), (1, 1), (5000, 300)])
def test_device_rows_equal_host_outputs_and_oracle(ctx, nq, nt):
    import torch
    Q, T, qb, tb = _banks(ctx, nq, nt, seed=nq + nt)
    dev = torch.device("cuda", ctx.device)
    rows = torch.full((nq, 3), -7, dtype=torch.int32, device=dev)
    count = torch.full((1,), -1, dtype=torch.int64, device=dev)
    n = ctx.match_accepted_dev(qb, tb, 0.7, rows.data_ptr(), count.data_ptr(), nq)
    q_acc, t_acc, d_acc, r_acc = ctx.match_accepted(qb, tb, 0.7)
    assert n == len(q_acc) == int(count.item())
    got = rows.cpu().numpy()
    assert np.array_equal(got[:n], sharding.pack_matches(q_acc, t_acc, d_acc))
    assert (got[n:] == -7).all()                       # nothing written past the count
    # and against the oracle directly
    otidx, odist = oracle.bf_xcheck1(Q, T)
    m = np.nonzero(otidx >= 0)[0]
    sd = oracle.self_dist(Q)
    _, opass = oracle.ratio_filter(odist[m], sd, 0.7, qrows=m.astype(np.int32))
    keep = m[opass]
    assert np.array_equal(got[:n, 0], keep) and np.array_equal(got[:n, 1], otidx[keep])
    assert np.array_equal(got[:n, 2].view(np.float32), odist[keep])


def test_device_rows_capacity_and_errors(ctx):
    import torch
    from fastmatch_amd import _ffi
    Q, T, qb, tb = _banks(ctx, 2000, 2000, seed=11)
    dev = torch.device("cuda", ctx.device)
    q_acc, t_acc, d_acc, _ = ctx.match_accepted(qb, tb, 0.9)
    cap = len(q_acc) // 2
    assert cap > 10
    rows = torch.full((cap + 5, 3), -7, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    n = ctx.match_accepted_dev(qb, tb, 0.9, rows.data_ptr(), count.data_ptr(), cap)
    assert n == len(q_acc) and int(count.item()) == cap     # host: the total; device word: the rows that are there
    got = rows.cpu().numpy()
    assert np.array_equal(got[:cap], sharding.pack_matches(q_acc, t_acc, d_acc)[:cap]) and (got[cap:] == -7).all()
    host = np.zeros((cap, 3), np.int32)
    with pytest.raises(_ffi.FastMatchHipError):         # host memory is not accepted
        ctx.match_accepted_dev(qb, tb, 0.9, host.ctypes.data, count.data_ptr(), cap)
    with pytest.raises(_ffi.FastMatchHipError):
        ctx.match_accepted_dev(qb, tb, 0.9, 0, count.data_ptr(), cap)


def test_async_stream_of_pairs_equals_the_synchronous_calls(ctx):
    """fm_match_accepted_async: several image pairs enqueued back to back, one fm_sync; every
    pair's accepted matches equal the synchronous call's, and the stats account every launch."""
    from fastmatch_amd import _ffi
    pairs = [_banks(ctx, 2500 + 300 * k, 2000 + 250 * k, seed=40 + k) for k in range(5)]
    outs = [(ctx.pinned_empty(4000, np.int32), ctx.pinned_empty(4000, np.int32),
             ctx.pinned_empty(4000, np.float32), ctx.pinned_empty(4000, np.float64)) for _ in pairs]
    counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    ctx.sync()
    ctx.reset_stats()
    for rep in range(2):
        for (Q, T, qb, tb), out, cnt in zip(pairs, outs, counts):
            cnt[0] = -1
            ctx.match_accepted_async(qb, tb, 0.7, out, cnt)
        ctx.sync()
        for (Q, T, qb, tb), out, cnt in zip(pairs, outs, counts):
            m = int(cnt[0])
            qa, ta, da, ra = ctx.match_accepted(qb, tb, 0.7)
            assert m == len(qa) > 50
            assert np.array_equal(out[0][:m], qa) and np.array_equal(out[1][:m], ta)
            assert np.array_equal(out[2][:m], da) and np.array_equal(out[3][:m], ra)
    st = ctx.stats()
    # 10 synchronous calls, all timed, + 10 async ones of which every FM_ASYNC_TIME_EVERY-th (default:
    # 4th) carries the start-of-kernel event; pairs are accounted for the timed launches only
    assert 10 < st["kernel_launches"] <= 20 and st["calls"] == st["kernel_launches"]
    # pairs of equal shape reuse a workspace slot without re-initialising it: the slot's tail kernels
    # must leave bound[] / qbest[] clean (three passes over three same-shape pairs + an odd one out)
    same = [_banks(ctx, 3000, 2400, seed=60 + k) for k in range(3)] + [pairs[1]]
    want = [ctx.match_accepted(qb, tb, 0.7) for (_, _, qb, tb) in same]
    souts = [(ctx.pinned_empty(4000, np.int32), ctx.pinned_empty(4000, np.int32),
              ctx.pinned_empty(4000, np.float32), ctx.pinned_empty(4000, np.float64)) for _ in range(12)]
    scnt = [ctx.pinned_empty(1, np.int64) for _ in range(12)]
    for j in range(12):
        ctx.match_accepted_async(same[j % 4][2], same[j % 4][3], 0.7, souts[j], scnt[j])
    ctx.sync()
    for j in range(12):
        qa, ta, da, ra = want[j % 4]
        m = int(scnt[j][0])
        assert m == len(qa) > 50
        assert np.array_equal(souts[j][0][:m], qa) and np.array_equal(souts[j][1][:m], ta)
        assert np.array_equal(souts[j][2][:m], da) and np.array_equal(souts[j][3][:m], ra)
    with pytest.raises(_ffi.FastMatchHipError):                        # pageable outputs are refused
        ctx.match_accepted_async(pairs[0][2], pairs[0][3], 0.7,
                                 (np.empty(4000, np.int32), np.empty(4000, np.int32), np.empty(4000, np.float32), np.empty(4000, np.float64)),
                                 counts[0])
    ctx.sync()


def test_rccl_gather_through_the_c_abi_one_rank():
    """fm_comm_unique_id / fm_comm_init / fm_gather_matches: the library's own RCCL all-gather of
    the device rows (world size 1 here; RCCL refuses two ranks on one GPU)."""
    import torch
    import fastmatch_amd
    c = fastmatch_amd.Context(0)                      # own context: the communicator binds to it
    Q, T, qb, tb = _banks(c, 3000, 2600, seed=5)
    dev = torch.device("cuda", 0)
    cap = 3000
    rows = torch.full((cap, 3), -7, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    all_rows = torch.full((1, cap, 3), -9, dtype=torch.int32, device=dev)
    all_counts = torch.full((1,), -1, dtype=torch.int64, device=dev)
    uid = c.comm_unique_id()
    assert len(uid) == 128
    c.comm_init(1, 0, uid)
    try:
        for tau in (0.7, 0.9):
            n = c.match_accepted_dev(qb, tb, tau, rows.data_ptr(), count.data_ptr(), cap)
            c.gather_matches(rows.data_ptr(), count.data_ptr(), cap, all_rows.data_ptr(), all_counts.data_ptr())
            qa, ta, da, _ = c.match_accepted(qb, tb, tau)
            assert int(all_counts[0].item()) == n == len(qa) > 100
            assert np.array_equal(all_rows[0, :n].cpu().numpy(), sharding.pack_matches(qa, ta, da))
            # counts first, then only the rows that are there: [nranks, m, 3] with m = the fullest rank's count
            all_rows.fill_(-9)
            m = c.gather_matches_counted(rows.data_ptr(), count.data_ptr(), cap, all_rows.data_ptr(), all_counts.data_ptr())
            assert m == n and int(all_counts[0].item()) == n
            flat = all_rows.view(-1, 3)
            assert np.array_equal(flat[:m].cpu().numpy(), sharding.pack_matches(qa, ta, da)) and bool((flat[m:] == -9).all())
        with pytest.raises(_ffi.FastMatchHipError):   # a second communicator on the same context
            c.comm_init(1, 0, uid)
    finally:
        c.comm_destroy()
    with pytest.raises(fastmatch_amd.FastMatchHipError):
        c.gather_matches(rows.data_ptr(), count.data_ptr(), cap, all_rows.data_ptr(), all_counts.data_ptr())
    c.close()


_RANK_SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import fastmatch_amd
from fastmatch_amd import synth, sharding, _ffi
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ["FM_TEST_BACKEND"]
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    tdev = torch.device("cuda", 0)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tdev = "cpu"
ctx = fastmatch_amd.Context(0)
if os.environ.get("FM_TEST_ASYNC") == "2":
    # whole steps: three pairs per step through fm_match_accepted_dev_batch, ONE all-gather per step
    PPS = 3
    g = sharding.MatchGatherer(tdev, capacity=4000, fill_device=torch.device("cuda", 0), pairs_per_step=PPS)
    keep, last = [], None
    for step in range(3):
        per_rank = []
        for r in range(world):
            per_rank.append([synth.planted_pair(4000, 33000 if i < 2 else 3000, seed=1000 * step + 10 * r + i)[:2] for i in range(PPS)])
        banks = []
        for Q, T in per_rank[rank]:
            qb, tb = ctx.bank(Q), ctx.bank(T)
            qb.set_selfdist(ctx.self_dist(qb))
            banks.append((qb, tb))
        keep.append(banks)
        rows, cnts = g.send_buffers()
        hc = ctx.pinned_empty(PPS, np.int64)
        ctx.match_accepted_dev_batch(banks, 0.7, rows.data_ptr(), cnts.data_ptr(), 4000, h_counts=hc,
                                     consumer_stream=g.consumer_stream())
        g.submit_device()
        last = per_rank
    counts, allrows = g.finish()
    ctx.sync()
    assert counts.shape == (world, PPS) and allrows.shape == (world, PPS, 4000, 3)
    for r in range(world):
        for i, (Q, T) in enumerate(last[r]):
            qr, tr = ctx.bank(Q), ctx.bank(T)
            qr.set_selfdist(ctx.self_dist(qr))
            qa, ta, da, _ = ctx.match_accepted(qr, tr, 0.7)
            m = int(counts[r, i].item())
            assert m == len(qa) > 100, (m, len(qa))
            assert np.array_equal(allrows[r, i, :m].cpu().numpy(), sharding.pack_matches(qa, ta, da))
            if r == rank:
                assert int(hc[i]) == m
    dist.barrier()
    dist.destroy_process_group()
    print("RANK_OK", rank)
    sys.exit(0)
g = sharding.MatchGatherer(tdev, capacity=4000, fill_device=torch.device("cuda", 0))
expect = []
keep = []
for step in range(3):                                   # a stream of pairs, gathers overlapped
    allq = []
    for r in range(world):
        Q, T, _ = synth.planted_pair(4000, 3000, seed=100 * step + r)
        allq.append((Q, T))
    Q, T = allq[rank]
    qb, tb = ctx.bank(Q), ctx.bank(T)
    qb.set_selfdist(ctx.self_dist(qb))
    keep.append((qb, tb))                               # banks outlive the kernels enqueued on them
    rows, count = g.send_buffers()
    if os.environ.get("FM_TEST_ASYNC") == "1":          # no host synchronisation between the pairs
        ctx.match_accepted_dev_async(qb, tb, 0.7, rows.data_ptr(), count.data_ptr(), 4000,
                                     consumer_stream=g.consumer_stream())
    else:
        n = ctx.match_accepted_dev(qb, tb, 0.7, rows.data_ptr(), count.data_ptr(), 4000)
    g.submit_device()
    counts, allrows = g.finish() if step == 2 else (None, None)
    if step == 2:
        ctx.sync()
        for r in range(world):                          # every rank's rows == its host-path result
            Qr, Tr = allq[r]
            qr, tr = ctx.bank(Qr), ctx.bank(Tr)
            qr.set_selfdist(ctx.self_dist(qr))
            qa, ta, da, _ = ctx.match_accepted(qr, tr, 0.7)
            m = int(counts[r].item())
            assert m == len(qa) > 100, (m, len(qa))
            assert np.array_equal(allrows[r, :m].cpu().numpy(), sharding.pack_matches(qa, ta, da))
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
'''


def _run_ranks(world, backend, tmp_path, async_fill=False):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    env = dict(os.environ, FM_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", FM_TEST_ASYNC=str(int(async_fill)))
    import socket
    with socket.socket() as sk:                # a port of its own per launch (the tests of this file follow each other)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:                      # keep the evidence where a later look can find it
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "rank_failure_%d_%s_%s.log" % (world, backend, int(async_fill))), "w") as f:
                f.write(p.stdout[-20000:] + "\n---- stderr ----\n" + p.stderr[-40000:])
        except OSError:
            pass
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("RANK_OK") == world


def test_gatherer_device_path_one_rank_rccl(tmp_path):
    """The real transport (RCCL all_gather_into_tensor from the device buffers), world size 1."""
    _run_ranks(1, "nccl", tmp_path)


def test_gatherer_device_path_two_ranks_sharing_the_gpu(tmp_path):
    """Two ranks on the one GPU of the box: device-side fill, gloo transport (RCCL refuses two
    ranks on one device); every rank sees both ranks' rows."""
    _run_ranks(2, "gloo", tmp_path)


def test_gatherer_async_fill_one_rank_rccl(tmp_path):
    """fm_match_accepted_dev_async feeding the RCCL all-gather through stream ordering only."""
    _run_ranks(1, "nccl", tmp_path, async_fill=True)


def test_gatherer_async_fill_two_ranks_sharing_the_gpu(tmp_path):
    _run_ranks(2, "gloo", tmp_path, async_fill=True)


def test_device_rows_async_are_ordered_against_the_consumer_stream(ctx):
    """fm_match_accepted_dev_async: a stream of pairs into two alternating device buffers that a
    torch stream snapshots right behind every call -- no host synchronisation until the end.  Every
    snapshot equals the synchronous call's rows (fill -> consumer ordering), although each buffer
    is refilled two pairs later (consumer -> fill ordering)."""
    import torch
    dev = torch.device("cuda", 0)
    cap = 4000
    pairs = [_banks(ctx, 3600, 3000, seed=80 + k) for k in range(3)] + [_banks(ctx, 2100, 2500, seed=90)]
    want = []
    for (_, _, qb, tb) in pairs:
        qa, ta, da, _ = ctx.match_accepted(qb, tb, 0.8)
        want.append(sharding.pack_matches(qa, ta, da))
    bufs = [torch.full((cap, 3), -7, dtype=torch.int32, device=dev) for _ in range(2)]
    cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
    hcnt = [ctx.pinned_empty(1, np.int64) for _ in range(10)]
    side = torch.cuda.Stream(device=dev)
    snaps = []
    with torch.cuda.stream(side):
        for j in range(10):
            k = j % 2
            ctx.match_accepted_dev_async(pairs[j % 4][2], pairs[j % 4][3], 0.8, bufs[k].data_ptr(), cnts[k].data_ptr(), cap,
                                         h_count=hcnt[j], consumer_stream=side.cuda_stream)
            snaps.append((bufs[k].clone(), cnts[k].clone()))        # on `side`, behind the fill
            bufs[k].fill_(-5)                                        # the next fill of this buffer must come after this
    ctx.sync()
    side.synchronize()
    for j, (rows, cnt) in enumerate(snaps):
        w = want[j % 4]
        assert int(cnt.item()) == len(w) == int(hcnt[j][0]) > 100
        assert np.array_equal(rows[:len(w)].cpu().numpy(), w)
        assert (rows[len(w):] == -5).all() or j < 2                  # (first use of a buffer: still the -7 fill)
    with pytest.raises(fastmatch_amd_error()):                       # host memory is refused
        ctx.match_accepted_dev_async(pairs[0][2], pairs[0][3], 0.8, np.zeros((cap, 3), np.int32).ctypes.data, cnts[0].data_ptr(), cap)


def fastmatch_amd_error():
    from fastmatch_amd import _ffi
    return _ffi.FastMatchHipError


def test_rccl_gather_follows_the_async_fill_one_rank():
    """fm_gather_matches behind fm_match_accepted_dev_async runs on the stream that fills the rows."""
    import torch
    import fastmatch_amd
    c = fastmatch_amd.Context(0)
    dev = torch.device("cuda", 0)
    cap = 3000
    pairs = [_banks(c, 3000, 2600, seed=15 + k) for k in range(3)]
    rows = torch.full((cap, 3), -7, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    all_rows = [torch.full((1, cap, 3), -9, dtype=torch.int32, device=dev) for _ in pairs]
    all_counts = [torch.full((1,), -1, dtype=torch.int64, device=dev) for _ in pairs]
    c.comm_init(1, 0, c.comm_unique_id())
    try:
        for k, (_, _, qb, tb) in enumerate(pairs):      # one send buffer, three pairs, no host synchronisation
            c.match_accepted_dev_async(qb, tb, 0.8, rows.data_ptr(), count.data_ptr(), cap)
            c.gather_matches(rows.data_ptr(), count.data_ptr(), cap, all_rows[k].data_ptr(), all_counts[k].data_ptr(), wait=False)
        c.sync()
        for k, (_, _, qb, tb) in enumerate(pairs):
            qa, ta, da, _ = c.match_accepted(qb, tb, 0.8)
            n = int(all_counts[k][0].item())
            assert n == len(qa) > 100
            assert np.array_equal(all_rows[k][0, :n].cpu().numpy(), sharding.pack_matches(qa, ta, da))
    finally:
        c.comm_destroy()
    c.close()


def test_a_synchronous_call_completes_the_async_calls_before_it(ctx):
    """include/fastmatch_hip.h: async results are valid after fm_sync OR any synchronous call."""
    pairs = [_banks(ctx, 5000, 4200, seed=70 + k) for k in range(3)]
    outs = [(ctx.pinned_empty(5000, np.int32), ctx.pinned_empty(5000, np.int32),
             ctx.pinned_empty(5000, np.float32), ctx.pinned_empty(5000, np.float64)) for _ in pairs]
    counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    for (Q, T, qb, tb), out, cnt in zip(pairs, outs, counts):
        cnt[0] = -1
        ctx.match_accepted_async(qb, tb, 0.7, out, cnt)
    ref0 = ctx.match_accepted(pairs[0][2], pairs[0][3], 0.7)          # synchronous: no fm_sync in between
    got = [(int(c[0]), o[0][:int(c[0])].copy(), o[1][:int(c[0])].copy()) for o, c in zip(outs, counts)]
    for (Q, T, qb, tb), (m, qa, ta) in zip(pairs, got):
        rq, rt, _, _ = ctx.match_accepted(qb, tb, 0.7)
        assert m == len(rq) > 50 and np.array_equal(qa, rq) and np.array_equal(ta, rt)
    ctx.sync()


def test_batch_call_groups_pairs_of_one_shape_into_shared_launches(ctx):
    """fm_match_accepted_batch: seven pairs of one shape (launches of 5 + 2 pairs), then a pair of
    another shape, a pair with an empty train bank and two more of the first shape -- every pair's
    accepted matches equal the synchronous call's; repeated, so the workspace slots are reused.
    (Train banks of >= 32768 rows: below that the planner picks 4-wave workgroups and the pairs are
    enqueued one by one.)"""
    same = [_banks(ctx, 3000, 33000, seed=120 + k) for k in range(9)]
    other = _banks(ctx, 1700, 2100, seed=140)
    Qe, Te, qbe, _ = _banks(ctx, 900, 800, seed=141)
    empty_t = ctx.bank(np.zeros((0, 128), np.uint8))
    order = same[:7] + [other] + [(Qe, None, qbe, empty_t)] + same[7:]
    pairs = [(p[2], p[3]) for p in order]
    want = [ctx.match_accepted(qb, tb, 0.75) for qb, tb in pairs]
    outs = [(ctx.pinned_empty(3000, np.int32), ctx.pinned_empty(3000, np.int32),
             ctx.pinned_empty(3000, np.float32), ctx.pinned_empty(3000, np.float64)) for _ in pairs]
    counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    ctx.sync()
    ctx.reset_stats()
    for rep in range(3):
        for c in counts:
            c[0] = -1
        ctx.match_accepted_batch(pairs, 0.75, outs, counts)
        ctx.sync()
        for (qa, ta, da, ra), out, cnt in zip(want, outs, counts):
            m = int(cnt[0])
            assert m == len(qa)
            assert np.array_equal(out[0][:m], qa) and np.array_equal(out[1][:m], ta)
            assert np.array_equal(out[2][:m], da) and np.array_equal(out[3][:m], ra)
    assert sum(len(w[0]) for w in want) > 1000
    st = ctx.stats()
    assert st["pairs"] >= 3 * 9 * 3000 * 33000          # every grouped launch is timed and accounts its pairs
    assert st["kernel_launches"] <= 3 * 6               # 7 -> 5 + 2, 2 -> one launch, the odd ones alone
    with pytest.raises(fastmatch_amd_error()):          # pageable outputs are refused
        ctx.match_accepted_batch(pairs[:2], 0.75, [(np.empty(3000, np.int32), np.empty(3000, np.int32),
                                                    np.empty(3000, np.float32), np.empty(3000, np.float64))] * 2, counts[:2])
    ctx.sync()


def test_batch_call_takes_pairs_of_different_sizes_in_one_launch(ctx):
    """r05: the pairs of a batched launch need not share a shape (a dataset's images all differ in size; before, such pairs
    were enqueued one by one and cost 7 % more per descriptor pair): six pairs of six different (query, train) sizes go
    out in two launches (4 + 2), every pair's accepted matches equal the synchronous call's; repeated."""
    shapes = [(3000, 33000), (2500, 35000), (4100, 40001), (3000, 36500), (1800, 33000), (2900, 47000)]
    sets = [_banks(ctx, nq, nt, seed=150 + k) for k, (nq, nt) in enumerate(shapes)]
    pairs = [(p[2], p[3]) for p in sets]
    want = [ctx.match_accepted(qb, tb, 0.75) for qb, tb in pairs]
    outs = [(ctx.pinned_empty(4100, np.int32), ctx.pinned_empty(4100, np.int32),
             ctx.pinned_empty(4100, np.float32), ctx.pinned_empty(4100, np.float64)) for _ in pairs]
    counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    ctx.sync()
    ctx.reset_stats()
    for rep in range(3):
        for c in counts:
            c[0] = -1
        ctx.match_accepted_batch(pairs, 0.75, outs, counts)
        ctx.sync()
        for (qa, ta, da, ra), out, cnt in zip(want, outs, counts):
            m = int(cnt[0])
            assert m == len(qa)
            assert np.array_equal(out[0][:m], qa) and np.array_equal(out[1][:m], ta)
            assert np.array_equal(out[2][:m], da) and np.array_equal(out[3][:m], ra)
    assert sum(len(w[0]) for w in want) > 1000
    st = ctx.stats()
    assert st["pairs"] >= 3 * sum(nq * nt for nq, nt in shapes)
    assert st["kernel_launches"] <= 3 * 2


def test_gatherer_whole_steps_one_rank_rccl(tmp_path):
    """fm_match_accepted_dev_batch + one RCCL all-gather per step (three pairs), stream ordering only."""
    _run_ranks(1, "nccl", tmp_path, async_fill=2)


def test_gatherer_whole_steps_two_ranks_sharing_the_gpu(tmp_path):
    _run_ranks(2, "gloo", tmp_path, async_fill=2)


def test_device_rows_batch_equals_the_synchronous_calls(ctx):
    """fm_match_accepted_dev_batch in one process: grouped and odd pairs into one [n, cap, 3] block that a
    side stream snapshots behind the call; twice, so the second fill has to wait for the snapshot."""
    import torch
    dev = torch.device("cuda", 0)
    cap = 3000
    pairs_full = [_banks(ctx, 3000, 33000, seed=160 + k) for k in range(4)] + [_banks(ctx, 1500, 2000, seed=170)]
    pairs = [(p[2], p[3]) for p in pairs_full]
    n = len(pairs)
    want = []
    for qb, tb in pairs:
        qa, ta, da, _ = ctx.match_accepted(qb, tb, 0.8)
        want.append(sharding.pack_matches(qa, ta, da))
    rows = torch.full((n, cap, 3), -7, dtype=torch.int32, device=dev)
    cnts = torch.zeros(n, dtype=torch.int64, device=dev)
    hc = ctx.pinned_empty(n, np.int64)
    side = torch.cuda.Stream(device=dev)
    snaps = []
    block = ctx.prepare_pairs(pairs)
    with torch.cuda.stream(side):
        for rep in range(2):
            ctx.match_accepted_dev_batch(block, 0.8, rows.data_ptr(), cnts.data_ptr(), cap, h_counts=hc,
                                         consumer_stream=side.cuda_stream)
            snaps.append((rows.clone(), cnts.clone()))
            rows.fill_(-5)
    ctx.sync()
    side.synchronize()
    for r, c in snaps:
        for i, w in enumerate(want):
            assert int(c[i].item()) == len(w) == int(hc[i]) > 50
            assert np.array_equal(r[i, :len(w)].cpu().numpy(), w)
    with pytest.raises(fastmatch_amd_error()):
        ctx.match_accepted_dev_batch(pairs, 0.8, np.zeros((n, cap, 3), np.int32).ctypes.data, cnts.data_ptr(), cap)


def test_the_null_stream_is_a_consumer_stream_too(ctx):
    """PyTorch's default stream has the handle 0: passing it as consumer_stream must order the fills
    against it like any other stream (FM_NO_STREAM / None is the value for "none").  Copies issued on
    the default stream right behind each call see the finished rows, every one of 40 back-to-back calls
    (found by scripts/stress_async.py: the copy ran ahead of the compaction)."""
    import torch
    dev = torch.device("cuda", 0)
    cap = 6000
    pairs_full = [_banks(ctx, 6000, 33000, seed=210 + k) for k in range(5)] + [_banks(ctx, 2000, 1500, seed=220)]
    pairs = [(p[2], p[3]) for p in pairs_full]
    n = len(pairs)
    want = []
    for qb, tb in pairs:
        qa, ta, da, _ = ctx.match_accepted(qb, tb, 0.75)
        want.append(sharding.pack_matches(qa, ta, da))
    rows = torch.zeros((n, cap, 3), dtype=torch.int32, device=dev)
    cnts = torch.zeros(n, dtype=torch.int64, device=dev)
    block = ctx.prepare_pairs(pairs)
    assert torch.cuda.current_stream().cuda_stream == 0
    for it in range(40):
        rows.fill_(-1)
        ctx.match_accepted_dev_batch(block, 0.75, rows.data_ptr(), cnts.data_ptr(), cap, consumer_stream=0)
        got_rows, got_cnt = rows.cpu().numpy(), cnts.cpu().numpy()
        for j, w in enumerate(want):
            assert int(got_cnt[j]) == len(w) > 50, (it, j)
            assert np.array_equal(got_rows[j, :len(w)], w), (it, j)
        rows1 = torch.full((cap, 3), -1, dtype=torch.int32, device=dev)
        cnt1 = torch.zeros(1, dtype=torch.int64, device=dev)
        ctx.match_accepted_dev_async(pairs[it % n][0], pairs[it % n][1], 0.75, rows1.data_ptr(), cnt1.data_ptr(), cap,
                                     consumer_stream=0)
        w = want[it % n]
        assert int(cnt1.cpu()[0]) == len(w) and np.array_equal(rows1.cpu().numpy()[:len(w)], w), it
    ctx.sync()


def test_mark_and_wait_let_a_consumer_trail_the_enqueued_batches_by_one(ctx):
    """fm_mark / fm_wait: batch i + 1 is enqueued before the host waits for batch i (two output sets);
    what the host reads after wait(ticket_i) is batch i's complete result, for 12 batches in a row."""
    from fastmatch_amd import _ffi
    shapes = [(5000, 33000)] * 4 + [(1800, 2500)]
    pairs_full = [_banks(ctx, nq, nt, seed=230 + k) for k, (nq, nt) in enumerate(shapes)]
    pairs = [(p[2], p[3]) for p in pairs_full]
    want = [ctx.match_accepted(qb, tb, 0.75) for qb, tb in pairs]
    sets = []
    for _ in range(2):
        outs = [(ctx.pinned_empty(5000, np.int32), ctx.pinned_empty(5000, np.int32), ctx.pinned_empty(5000, np.float32),
                 ctx.pinned_empty(5000, np.float64)) for _ in pairs]
        cnts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
        sets.append((outs, cnts, ctx.prepare_batch(pairs, outs, cnts)))

    def check(s):
        outs, cnts, _ = sets[s]
        for (qa, ta, da, ra), out, cnt in zip(want, outs, cnts):
            m = int(cnt[0])
            assert m == len(qa) > 50
            assert np.array_equal(out[0][:m], qa) and np.array_equal(out[1][:m], ta) and np.array_equal(out[3][:m], ra)
            out[0][:m] = -1                      # so that a stale buffer cannot pass the next time round
            cnt[0] = -1

    prev = None
    for i in range(12):
        s = i % 2
        ctx.match_accepted_batch(sets[s][2], 0.75)
        ticket = ctx.mark()
        if prev is not None:
            ctx.wait(prev[1])
            check(prev[0])
        prev = (s, ticket)
    ctx.wait(prev[1])
    check(prev[0])
    ctx.wait(0)                                  # an old ticket: covered by the marks that followed it
    with pytest.raises(_ffi.FastMatchHipError):
        ctx.wait(10 ** 6)
    ctx.sync()


def test_batch_of_more_pairs_than_workspace_slots(ctx):
    """20 same-shape pairs in one fm_match_accepted_batch call (launches of 8 + 8 + 2 + ... pairs; the
    ring of 16 workspaces wraps inside the call), twice: every pair equals its synchronous call."""
    base = [_banks(ctx, 2500, 33000, seed=300 + k) for k in range(5)]
    pairs = [(base[k % 5][2], base[(k * 2 + 1) % 5][3]) for k in range(20)]          # 20 distinct (query, train) combinations
    want = [ctx.match_accepted(qb, tb, 0.9) for qb, tb in pairs]
    outs = [(ctx.pinned_empty(2500, np.int32), ctx.pinned_empty(2500, np.int32),
             ctx.pinned_empty(2500, np.float32), ctx.pinned_empty(2500, np.float64)) for _ in pairs]
    counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    block = ctx.prepare_batch(pairs, outs, counts)
    for rep in range(2):
        for c in counts:
            c[0] = -1
        ctx.match_accepted_batch(block, 0.9)
        ctx.sync()
        for j, ((qa, ta, da, ra), out, cnt) in enumerate(zip(want, outs, counts)):
            m = int(cnt[0])
            assert m == len(qa), j
            assert np.array_equal(out[0][:m], qa) and np.array_equal(out[1][:m], ta) and np.array_equal(out[3][:m], ra), j
    assert sum(len(w[0]) for w in want) > 100


def test_batch_outputs_shorter_than_the_accepted_lists_are_truncated_not_overrun(ctx):
    """Capacity smaller than the number of accepted matches (host and device outputs of the batch calls):
    the count is the full number, the buffers hold the first `cap` rows in query order and nothing is
    written beyond them."""
    import torch
    pairs_full = [_banks(ctx, 3000, 33000, seed=400 + k) for k in range(3)]
    pairs = [(p[2], p[3]) for p in pairs_full]
    want = [ctx.match_accepted(qb, tb, 0.9) for qb, tb in pairs]
    cap = 64
    assert all(len(w[0]) > cap for w in want)
    guard = 16
    outs, raw = [], []
    for _ in pairs:
        bufs = (ctx.pinned_empty(cap + guard, np.int32), ctx.pinned_empty(cap + guard, np.int32),
                ctx.pinned_empty(cap + guard, np.float32), ctx.pinned_empty(cap + guard, np.float64))
        for b in bufs:
            b[:] = 77
        raw.append(bufs)
        outs.append(tuple(b[:cap] for b in bufs))
    counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    ctx.match_accepted_batch(pairs, 0.9, outs, counts)
    ctx.sync()
    for (qa, ta, da, ra), out, bufs, cnt in zip(want, outs, raw, counts):
        assert int(cnt[0]) == len(qa)
        assert np.array_equal(out[0], qa[:cap]) and np.array_equal(out[1], ta[:cap]) and np.array_equal(out[3], ra[:cap])
        assert all((b[cap:] == 77).all() for b in bufs)
    dev = torch.device("cuda", 0)
    cnts = torch.zeros(len(pairs), dtype=torch.int64, device=dev)
    # (device rows: pair i at rows_ptr + i * cap * 3 -- a flat buffer with the exact stride and a guard at the very end)
    flat = torch.full((len(pairs) * cap + guard, 3), -7, dtype=torch.int32, device=dev)
    ctx.match_accepted_dev_batch(pairs, 0.9, flat.data_ptr(), cnts.data_ptr(), cap)
    ctx.sync()
    got = flat.cpu().numpy()
    for i, (qa, ta, da, ra) in enumerate(want):
        assert int(cnts[i].item()) == cap                       # the device word counts the rows that are there ...
        assert np.array_equal(got[i * cap:(i + 1) * cap], sharding.pack_matches(qa, ta, da)[:cap])
    assert (got[len(pairs) * cap:] == -7).all()
    hc = ctx.pinned_empty(len(pairs), np.int64)                 # ... the host word the full number of accepted matches
    ctx.match_accepted_dev_batch(pairs, 0.9, flat.data_ptr(), cnts.data_ptr(), cap, h_counts=hc)
    ctx.sync()
    assert hc.tolist() == [len(w[0]) for w in want] and cnts.cpu().tolist() == [cap] * len(pairs)


def test_float32_route_pairs_inside_a_batch(ctx):
    """A batch that mixes integer-valued pairs with float32-route pairs (RootSIFT-style descriptors): the
    float32 pairs run synchronously in place, every pair's outputs equal its single call; a bad pair anywhere
    in the batch is refused before anything is enqueued."""
    rng = np.random.default_rng(11)
    ints = [_banks(ctx, 3000, 33000, seed=900 + k) for k in range(3)]
    fl = []
    for k in range(2):
        Qf = (synth.synth_sift(700, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (700, 128)).astype(np.float32))
        Tf = (synth.synth_sift(900, rng).astype(np.float32) + rng.uniform(-0.5, 0.5, (900, 128)).astype(np.float32))
        Tf[:200] = Qf[:200] + rng.normal(0, 2, (200, 128)).astype(np.float32)
        qb, tb = ctx.bank(Qf), ctx.bank(Tf)
        assert qb.kind == _ffi.FM_BANK_F32
        qb.set_selfdist(ctx.self_dist(qb))
        fl.append((qb, tb))
    pairs = [(ints[0][2], ints[0][3]), fl[0], (ints[1][2], ints[1][3]), (ints[2][2], ints[2][3]), fl[1]]
    want = [ctx.match_accepted(q, t, 0.8) for q, t in pairs]
    assert all(len(w[0]) > 20 for w in want)
    outs = [tuple(ctx.pinned_empty(3000, dt) for dt in (np.int32, np.int32, np.float32, np.float64)) for _ in pairs]
    counts = [ctx.pinned_empty(1, np.int64) for _ in pairs]
    for rep in range(2):
        for c in counts:
            c[0] = -1
        ctx.match_accepted_batch(pairs, 0.8, outs, counts)
        ctx.sync()
        for w, o, c in zip(want, outs, counts):
            m = int(c[0])
            assert m == len(w[0]) and all(np.array_equal(a[:m].view(np.uint8), b.view(np.uint8)) for a, b in zip(o, w))
    # device outputs
    import torch
    dev = torch.device("cuda", 0)
    rows = torch.full((len(pairs) * 3000, 3), -7, dtype=torch.int32, device=dev)
    cnts = torch.zeros(len(pairs), dtype=torch.int64, device=dev)
    ctx.match_accepted_dev_batch(pairs, 0.8, rows.data_ptr(), cnts.data_ptr(), 3000)
    ctx.sync()
    got = rows.cpu().numpy().reshape(len(pairs), 3000, 3)
    for i, w in enumerate(want):
        m = int(cnts[i].item())
        assert m == len(w[0]) and np.array_equal(got[i, :m], sharding.pack_matches(w[0], w[1], w[2]))
    # a pair without self distances in the MIDDLE: refused up front, nothing written
    bad_q = ctx.bank(synth.planted_pair(3000, 33000, 1)[0])
    for c in counts:
        c[0] = -5
    with pytest.raises(_ffi.FastMatchHipError):
        ctx.match_accepted_batch(pairs[:2] + [(bad_q, ints[0][3])] + pairs[2:4], 0.8, outs, counts)
    ctx.sync()
    assert all(int(c[0]) == -5 for c in counts)


def test_benchmarked_launch_shape_against_the_oracle():
    """The launch bench.py times: twelve DISTINCT pairs in ONE rowreduce_batch_kernel launch (options
    batch_group = 16, batch_tail = 0), steps pipelined two deep with fm_mark / fm_wait on two output
    sets -- every pair's accepted rows (query index, train index, distance bits, ratio bits) against
    oracle.bf_xcheck1 + ratio_filter, not against another call of the library."""
    import fastmatch_amd
    import oracle
    c = fastmatch_amd.Context(0)
    c.set_option("batch_group", 16)
    c.set_option("batch_tail", 0)
    nq, nt, tau = 8000, 33000, 0.7
    pairs, want = [], []
    for k in range(12):
        Q, T, _ = synth.planted_pair(nq, nt, seed=700 + k)
        qb, tb = c.bank(Q), c.bank(T)
        sd = oracle.self_dist(Q)
        qb.set_selfdist(sd)
        ot, od = oracle.bf_xcheck1(Q, T)
        m = ot >= 0
        orat, opass = oracle.ratio_filter(od[m], sd, tau, qrows=np.nonzero(m)[0].astype(np.int32))
        acc = np.nonzero(m)[0][opass]
        want.append((acc.astype(np.int32), ot[acc], od[acc], orat[opass]))
        pairs.append((qb, tb))
    assert sum(len(w[0]) for w in want) > 5000
    sets = []
    for s in range(2):
        outs = [(c.pinned_empty(nq, np.int32), c.pinned_empty(nq, np.int32), c.pinned_empty(nq, np.float32),
                 c.pinned_empty(nq, np.float64)) for _ in pairs]
        counts = [c.pinned_empty(1, np.int64) for _ in pairs]
        sets.append((c.prepare_batch(pairs, outs, counts), outs, counts))

    def check(s):
        _, outs, counts = sets[s]
        for j, (w, out, cnt) in enumerate(zip(want, outs, counts)):
            m = int(cnt[0])
            assert m == len(w[0]), j
            for a, b in zip(out, w):
                assert np.array_equal(a[:m].view(np.uint8), b.view(np.uint8)), j
            cnt[0] = -1

    c.reset_stats()
    prev = None
    for step in range(5):
        s = step % 2
        c.match_accepted_batch(sets[s][0], tau)
        ticket = c.mark()
        if prev is not None:
            c.wait(prev[1])
            check(prev[0])
        prev = (s, ticket)
    c.wait(prev[1])
    check(prev[0])
    c.sync()
    st = c.stats()
    assert st["kernel_launches"] == 5 and st["pairs"] == 5 * 12 * nq * nt      # one launch per step, twelve pairs each
    c.close()
