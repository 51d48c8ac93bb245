"""CPU, world_size 2 over gloo: pair sharding and the variable-length match all-gather
that bench.py / multi-GPU callers use (RCCL on GPUs, same code path)."""
import os
import socket
import subprocess
import sys

import numpy as np

from fastmatch_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from fastmatch_amd import sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pairs = sharding.shard_items(7, rank, world)
rows = []
for p in pairs:                                  # fake per-pair match lists, sizes differ by rank
    m = 3 + 2 * p
    rows.append(sharding.pack_matches(np.arange(m) + 1000 * p, np.arange(m)[::-1] + 10 * p,
                                      (np.arange(m, dtype=np.float32) + 0.5) * (p + 1)))
mine = np.concatenate(rows) if rows else np.zeros((0, 3), np.int32)
for cap in (None, 64):
    got = sharding.all_gather_matches(mine, capacity=cap)
    assert len(got) == world
    for r in range(world):
        exp = []
        for p in sharding.shard_items(7, r, world):
            m = 3 + 2 * p
            exp.append(sharding.pack_matches(np.arange(m) + 1000 * p, np.arange(m)[::-1] + 10 * p,
                                             (np.arange(m, dtype=np.float32) + 0.5) * (p + 1)))
        exp = np.concatenate(exp)
        assert np.array_equal(got[r], exp), (rank, r)
        q, t, d = sharding.unpack_matches(got[r])
        assert d.dtype == np.float32 and np.array_equal(d.view(np.int32), exp[:, 2])
empty = sharding.all_gather_matches(np.zeros((0, 3), np.int32) if rank == 0 else mine[:2])
assert len(empty[0]) == 0 and len(empty[1]) == 2
# overlapped gatherer: a stream of steps, double-buffered, last result checked
g = sharding.MatchGatherer("cpu", capacity=64)
for step in range(5):
    g.submit(mine[: max(0, len(mine) - step)])
counts, rows = g.finish()
for r in range(world):
    n_r = sum(3 + 2 * p for p in sharding.shard_items(7, r, world)) - 4
    assert int(counts[r]) == n_r, (rank, r, int(counts[r]), n_r)
    if r == rank:
        assert np.array_equal(rows[r, :n_r].numpy(), mine[:n_r])
# whole steps: three pairs per step in one send buffer, ONE all-gather per step (what bench.py does at
# N > 1 with fm_match_accepted_dev_batch; here the buffers are filled on the host)
import torch
PPS, CAP = 3, 16
gs = sharding.MatchGatherer("cpu", capacity=CAP, pairs_per_step=PPS)
assert gs.consumer_stream() is None
for step in range(4):
    rows_t, cnts_t = gs.send_buffers()
    assert rows_t.shape == (PPS * CAP, 3) and cnts_t.shape == (PPS,)
    for i in range(PPS):
        m = (step + 2 * i + 3 * rank) %% (CAP + 1)
        cnts_t[i] = m
        rows_t[i * CAP:i * CAP + m] = torch.arange(m * 3, dtype=torch.int32).view(m, 3) + 1000 * rank + 100 * i + step
    gs.submit_device()
counts_s, rows_s = gs.finish()
assert counts_s.shape == (world, PPS) and rows_s.shape == (world, PPS, CAP, 3)
for r in range(world):
    for i in range(PPS):
        m = (3 + 2 * i + 3 * r) %% (CAP + 1)
        assert int(counts_s[r, i]) == m, (rank, r, i)
        exp = torch.arange(m * 3, dtype=torch.int32).view(m, 3) + 1000 * r + 100 * i + 3
        assert torch.equal(rows_s[r, i, :m], exp), (rank, r, i)
# the same steps through the counts-first form: only max(count) rows per pair travel, same contents
g2 = sharding.MatchGatherer("cpu", capacity=CAP, pairs_per_step=PPS, two_phase=True)
for step in range(4):
    rows_t, cnts_t = g2.send_buffers()
    for i in range(PPS):
        m = (step + 2 * i + 3 * rank) %% (CAP + 1)
        cnts_t[i] = m
        rows_t[i * CAP:i * CAP + m] = torch.arange(m * 3, dtype=torch.int32).view(m, 3) + 1000 * rank + 100 * i + step
    g2.submit_device()
counts_2, rows_2 = g2.finish()
mmax = int(counts_s.max())
assert torch.equal(counts_2, counts_s) and rows_2.shape == (world, PPS, mmax, 3) and mmax < CAP
assert g2.rows_shipped == PPS * mmax and gs.rows_shipped == PPS * CAP
for r in range(world):
    for i in range(PPS):
        m = int(counts_s[r, i])
        assert torch.equal(rows_2[r, i, :m], rows_s[r, i, :m]), (rank, r, i)
g1 = sharding.MatchGatherer("cpu", capacity=64, two_phase=True)          # one pair per step, host path
for step in range(3):
    g1.submit(mine[: max(0, len(mine) - step)])
c1, r1 = g1.finish()
for r in range(world):
    n_r = sum(3 + 2 * p for p in sharding.shard_items(7, r, world)) - 2
    assert int(c1[r]) == n_r and r1.shape[1] == int(c1.max())
    if r == rank:
        assert np.array_equal(r1[r, :n_r].numpy(), mine[:n_r])
# ONE large problem, train rows sharded: election keys per shard (here from the CPU oracle, on
# the GPU from fm_xcheck1_keys), one all-reduce(min), decode == the unsharded cross-check
sys.path.insert(0, %r)
import oracle
rng = np.random.default_rng(5)
Q = rng.integers(0, 4, (90, 128)).astype(np.uint8) * 60          # few levels: plenty of ties
T = rng.integers(0, 4, (150, 128)).astype(np.uint8) * 60
lo, hi = sharding.shard_rows(150, rank, world)
tl, dl = oracle.bf_xcheck1(Q, T[lo:hi])
db = dl.view(np.uint32).astype(np.uint64)                           # key = float32 distance bits << 32 | global train row
keys = np.where(tl >= 0, (db << np.uint64(32)) | (tl.astype(np.int64) + lo).astype(np.uint64), np.uint64(0xFFFFFFFFFFFFFFFF))
tidx, dd = sharding.decode_keys(sharding.reduce_keys(keys))
ft, fd = oracle.bf_xcheck1(Q, T)
assert np.array_equal(tidx, ft) and np.array_equal(dd.view(np.uint32), fd.view(np.uint32)), rank
# query rows sharded (2-NN): padded row shards, one all-gather
for n in (0, 1, 7, 150):
    full = (np.arange(n * 4, dtype=np.int32).reshape(n, 4) * 7) ^ 0x5a5a
    lo, hi = sharding.shard_rows(n, rank, world)
    got = sharding.gather_row_shards(full[lo:hi], n)
    assert got.shape == (n, 4) and np.array_equal(got, full), (rank, n)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
''' % (ROOT, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_rows_partition():
    for n in (0, 1, 7, 100000):
        for w in (1, 2, 3, 8):
            edges = [sharding.shard_rows(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))


def test_decode_keys():
    none = np.uint64(0xFFFFFFFFFFFFFFFF)
    five = np.uint64(np.array([5.0], np.float32).view(np.uint32)[0])
    keys = np.array([(five << np.uint64(32)) | np.uint64(7), none, np.uint64(0)], dtype=np.uint64)
    t, d = sharding.decode_keys(keys)
    assert t.tolist() == [7, -1, 0] and d[0] == 5.0 and np.isinf(d[1]) and d[2] == 0.0
    f = np.array([1.25], np.float32).view(np.uint32)[0]
    t, d = sharding.decode_keys(np.array([(np.uint64(f) << np.uint64(32)) | np.uint64(3)], dtype=np.uint64))
    assert t.tolist() == [3] and d[0] == np.float32(1.25)
    assert np.array_equal(sharding.reduce_keys(keys), keys)          # single process: identity


def test_shard_items_partition():
    for n in (0, 1, 7, 64):
        for w in (1, 2, 8):
            parts = [sharding.shard_items(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_pack_roundtrip_and_single_process_gather():
    d = np.array([1.5, np.inf, 0.0], dtype=np.float32)
    p = sharding.pack_matches([1, 2, 3], [9, 8, 7], d)
    q, t, dd = sharding.unpack_matches(p)
    assert q.tolist() == [1, 2, 3] and t.tolist() == [9, 8, 7] and np.array_equal(dd, d)
    assert np.array_equal(sharding.all_gather_matches(p)[0], p)


def test_all_gather_matches_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o
