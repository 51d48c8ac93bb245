"""GPU: accepted matches that stay on the device (fm_match_accepted_dev) and feed the multi-GPU
result gather without a host hop.  The rows must equal what fm_match_accepted returns to the
host (which tests/test_parity_gpu.py checks against the oracle)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from fastmatch_amd import synth, sharding
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
    assert n == len(q_acc) == int(count.item())        # the total, although only cap rows fit
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
    assert st["kernel_launches"] == 20 and st["calls"] == 20           # 10 async + 10 sync
    assert st["pairs"] == 4 * sum(len(p[0]) * len(p[1]) for p in pairs)
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
        from fastmatch_amd import _ffi
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
from fastmatch_amd import synth, sharding
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
g = sharding.MatchGatherer(tdev, capacity=4000, fill_device=torch.device("cuda", 0))
expect = []
for step in range(3):                                   # a stream of pairs, gathers overlapped
    allq = []
    for r in range(world):
        Q, T, _ = synth.planted_pair(4000, 3000, seed=100 * step + r)
        allq.append((Q, T))
    Q, T = allq[rank]
    qb, tb = ctx.bank(Q), ctx.bank(T)
    qb.set_selfdist(ctx.self_dist(qb))
    rows, count = g.send_buffers()
    n = ctx.match_accepted_dev(qb, tb, 0.7, rows.data_ptr(), count.data_ptr(), 4000)
    g.submit_device()
    counts, allrows = g.finish() if step == 2 else (None, None)
    if step == 2:
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


def _run_ranks(world, backend, tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    env = dict(os.environ, FM_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("RANK_OK") == world


def test_gatherer_device_path_one_rank_rccl(tmp_path):
    """The real transport (RCCL all_gather_into_tensor from the device buffers), world size 1."""
    _run_ranks(1, "nccl", tmp_path)


def test_gatherer_device_path_two_ranks_sharing_the_gpu(tmp_path):
    """Two ranks on the one GPU of the box: device-side fill, gloo transport (RCCL refuses two
    ranks on one device); every rank sees both ranks' rows."""
    _run_ranks(2, "gloo", tmp_path)
