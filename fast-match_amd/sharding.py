"""Multi-GPU layout of the hot path: independent image pairs are sharded over ranks
(one process per GPU), matching needs no communication, and the per-pair match lists are
returned with ONE variable-length all-gather (RCCL over xGMI on GPUs; gloo in the CPU
tests).  The reference has no distributed code; this is the data-parallel axis it implies
(turntable.py:59 maps the matcher over independent pairs) -- SURVEY.md 8(e).
For ONE large problem: shard the train rows for the cross-check (``xcheck1_sharded``: one
all-reduce(min) of nq packed keys) or the query rows for 2-NN (``knn2_sharded``: one
all-gather); both reproduce the single-GPU result bit for bit.
"""
import numpy as np


def shard_items(n_items, rank, world_size):
    """Indices of the items (image pairs) rank owns: i -> rank i mod world_size."""
    return list(range(rank, n_items, world_size))


def pack_matches(qidx, tidx, dist):
    """[m, 3] int32 rows (query index, train index, float32 distance bits)."""
    out = np.empty((len(qidx), 3), dtype=np.int32)
    out[:, 0] = qidx
    out[:, 1] = tidx
    out[:, 2] = np.asarray(dist, dtype=np.float32).view(np.int32)
    return out


def unpack_matches(packed):
    packed = np.asarray(packed, dtype=np.int32).reshape(-1, 3)
    return packed[:, 0].copy(), packed[:, 1].copy(), packed[:, 2].copy().view(np.float32)


def all_gather_matches(packed, device=None, group=None, capacity=None, to_host=True):
    """Gather every rank's [m_r, 3] int32 match rows; returns a list (by rank) of arrays.

    One collective for the counts and one for the padded payloads (a match is 12 bytes, a
    100k-row pair yields < 1 MB: latency bound, never link bound).  ``capacity`` fixes the
    padded row count (e.g. the query count) so no extra size exchange is needed.
    ``to_host=False`` leaves the gathered rows on the device and returns
    ``(counts int64[world], rows int32[world, capacity, 3])`` tensors instead."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [np.asarray(packed, dtype=np.int32).reshape(-1, 3)]
    world = dist.get_world_size(group)
    dev = device if device is not None else "cpu"
    if not isinstance(dev, str):
        dev = dev if dev.type != "cpu" else "cpu"
    packed = np.ascontiguousarray(packed, dtype=np.int32).reshape(-1, 3)
    m = packed.shape[0]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([m], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group) if dev != "cpu" else \
        dist.all_gather(list(counts.split(1)), mine, group=group)
    if capacity is None:
        capacity = int(counts.max().item())
    if m > capacity:
        raise ValueError("capacity %d smaller than local match count %d" % (capacity, m))
    buf = torch.zeros((capacity, 3), dtype=torch.int32, device=dev)
    if m:
        buf[:m] = torch.from_numpy(packed).to(dev)
    if dev != "cpu":
        allbuf = torch.empty((world * capacity, 3), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allbuf, buf, group=group)
        allbuf = allbuf.view(world, capacity, 3)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        allbuf = torch.stack(parts)
    if not to_host:
        return counts, allbuf
    counts = counts.cpu().numpy()
    host = allbuf.cpu().numpy()
    return [host[r, :int(counts[r])].copy() for r in range(world)]


# ---- ONE large problem split over ranks -----------------------------------
def shard_rows(n_rows, rank, world_size):
    """Contiguous row range [lo, hi) of rank's shard of an n_rows bank."""
    per = (n_rows + world_size - 1) // world_size
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def reduce_keys(keys, device=None, group=None):
    """Element-wise minimum of every rank's uint64 election keys (fm_xcheck1_keys): ONE
    all-reduce(min) of nq words -- the only exchange a train-sharded cross-check needs.
    Keys are < 2^63 (d^2 < 2^24 or the bits of a non-negative float32 in the high word; ~0
    = "none" maps to int64 max), so the signed minimum is the unsigned one."""
    import torch
    import torch.distributed as dist
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return keys
    none = keys == np.uint64(0xFFFFFFFFFFFFFFFF)
    signed = keys.view(np.int64).copy()
    signed[none] = np.iinfo(np.int64).max
    t = torch.from_numpy(signed)
    if device is not None and not (isinstance(device, str) and device == "cpu"):
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    out = t.cpu().numpy().view(np.uint64).copy()
    out[t.cpu().numpy() == np.iinfo(np.int64).max] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return out


def reduce_keys_device(keys_t, group=None):
    """``reduce_keys`` on a device tensor (int64 view of the uint64 keys, as ``fm_xcheck1_keys_dev``
    leaves them): the "none" key ~0 (= -1 signed) is mapped to int64 max on the device, ONE
    all-reduce(min) runs on the tensor in place (RCCL), and the keys come back as uint64 NumPy."""
    import torch
    import torch.distributed as dist
    big = torch.iinfo(torch.int64).max
    keys_t.masked_fill_(keys_t == -1, big)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(keys_t, op=dist.ReduceOp.MIN, group=group)
    out = keys_t.cpu().numpy()
    res = out.view(np.uint64).copy()
    res[out == big] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return res


def decode_keys(keys, float32_route=None):
    """(tidx int32[nq] (-1 = no match), dist float32[nq] (+inf = no match)) from election keys:
    what fm_xcheck1 returns.  The high word of a key is the float32 distance (its bits) on both routes:
    OpenCV's cross-check compares the float32 distances, and two integer d^2 >= 4 197 200 can share one
    (``float32_route`` is accepted for older callers and ignored)."""
    keys = np.asarray(keys, dtype=np.uint64)
    none = keys == np.uint64(0xFFFFFFFFFFFFFFFF)
    tidx = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64).astype(np.int32)
    hi = (keys >> np.uint64(32)).astype(np.uint32)
    dist = hi.view(np.float32).copy()
    tidx[none] = -1
    dist[none] = np.inf
    return tidx, dist


def xcheck1_sharded(ctx, qbank, tbank_shard, t_offset, device=None, group=None):
    """Cross-checked 1-NN of the (replicated) query bank against a train set whose rows are
    split over the ranks (this rank holds rows [t_offset, t_offset + tbank_shard.n)).
    Every rank returns the full (tidx, dist) of the unsharded fm_xcheck1, bit for bit."""
    on_gpu = device is not None and not (isinstance(device, str) and device == "cpu")
    if on_gpu:                                  # keys stay in HBM from the election to the collective
        import torch
        keys_t = torch.empty(qbank.n, dtype=torch.int64, device=device)
        ctx.xcheck1_keys_dev(qbank, tbank_shard, t_offset, keys_t.data_ptr())
        keys = reduce_keys_device(keys_t, group=group)
    else:
        keys = ctx.xcheck1_keys(qbank, tbank_shard, t_offset)
        keys = reduce_keys(keys, device=device, group=group)
    return decode_keys(keys)


def gather_row_shards(local, n_rows, device=None, group=None):
    """All-gather of a row-sharded int32 [m_r, c] array whose shards follow ``shard_rows``:
    returns the full [n_rows, c] array on every rank (one collective, padded shards)."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=np.int32)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    per = (n_rows + world - 1) // world
    dev = "cpu" if device is None or (isinstance(device, str) and device == "cpu") else device
    buf = torch.zeros((per, local.shape[1]), dtype=torch.int32)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local)
    buf = buf.to(dev)
    if dev == "cpu":
        parts = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        allbuf = torch.cat(parts)
    else:
        allbuf = torch.empty((world * per, local.shape[1]), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allbuf, buf, group=group)
    return np.ascontiguousarray(allbuf.cpu().numpy()[:n_rows])      # shards are contiguous and only the last is short


def knn2_sharded(ctx, qbank_shard, tbank, n_query, device=None, group=None):
    """2-NN of a query bank split over the ranks by ``shard_rows`` (rows are independent):
    this rank computes its rows; one all-gather returns all n_query rows on every rank."""
    idx, d = ctx.knn2(qbank_shard, tbank)
    rows = gather_row_shards(np.concatenate([idx, d.view(np.int32)], axis=1), n_query, device=device, group=group)
    return np.ascontiguousarray(rows[:, :2]), np.ascontiguousarray(rows[:, 2:]).view(np.float32)


class MatchGatherer(object):
    """Overlapped result gather for a stream of steps, double buffered: the all-gather of one
    image pair's accepted matches runs (RCCL, its own stream) while the next pair's matching
    kernels execute on the library's HIP stream.  ``finish()`` waits for the last gather and
    returns ``(counts int64[world], rows int32[world, capacity, 3])`` tensors.

    Device path (the default on GPUs): ``rows, count = send_buffers()`` hands out the free
    slot's device tensors, ``Context.match_accepted_dev(q, t, tau, rows.data_ptr(),
    count.data_ptr(), capacity)`` fills them on the device, ``submit_device()`` starts the
    collective straight from them -- the match rows never leave HBM.
    Host path: ``submit(packed)`` with an [m, 3] int32 NumPy array (CPU tests over gloo).
    ``fill_device``: with a CPU transport (gloo dry run of the device path on a one-GPU box)
    the send buffers still live on that CUDA device and are copied to the host for transport."""

    def __init__(self, device, capacity, group=None, fill_device=None, pairs_per_step=1, two_phase=False):
        """``two_phase``: counts first, then only as many rows per pair as the fullest (rank, pair) holds
        instead of the padded capacity (the bench's step: 4.3 MB instead of 14 MB per rank) -- for one
        wait on the counts between the two collectives, so the gather no longer starts without the host.
        ``finish()`` then returns rows of shape [world, (pairs_per_step,) m, 3] with m <= capacity.
        ``pairs_per_step`` > 1: a send buffer holds the rows of a whole step -- int32
        [pairs_per_step * capacity, 3] (pair i at rows [i * capacity, (i + 1) * capacity)) and int64
        [pairs_per_step] counts, what ``Context.match_accepted_dev_batch`` fills -- and ONE all-gather
        ships the step (fewer, larger collectives); ``finish()`` then returns ``(counts [world,
        pairs_per_step], rows [world, pairs_per_step, capacity, 3])``."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.dev = device
        self.capacity = int(capacity)
        self.pps = int(pairs_per_step)
        self.on_cpu = isinstance(device, str) and device == "cpu"
        mk = lambda dev, *shape, dtype: [torch.zeros(shape, dtype=dtype, device=dev) for _ in range(2)]
        self.buf = mk(device, self.pps * self.capacity, 3, dtype=torch.int32)
        self.mine = mk(device, self.pps, dtype=torch.int64)
        self.allbuf = mk(device, self.world * self.pps * self.capacity, 3, dtype=torch.int32)
        self.counts = mk(device, self.world * self.pps, dtype=torch.int64)
        self.fill_buf = self.fill_mine = None
        if self.on_cpu and fill_device is not None:
            self.fill_buf = mk(fill_device, self.pps * self.capacity, 3, dtype=torch.int32)
            self.fill_mine = mk(fill_device, self.pps, dtype=torch.int64)
        self.pending = []
        self.slot = 0
        self.two_phase = bool(two_phase)
        self.rows_shipped = 0                      # rows per rank of the last collective (capacity unless two_phase)
        self._packed = None
        self._phase2 = None                        # two_phase: slot whose counts are under way and whose rows are still to ship

    def _wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self._phase2 is not None:
            # Second half of a counts-first gather, deferred to the NEXT submit (or finish): the host reads the counts
            # only after the caller has enqueued the following step, so the step pipeline stays two deep (r03 read
            # them inside submit_device and stalled the host on the step it had just enqueued).  The slot's send
            # buffer is not handed out again before this point (double buffering).
            k, self._phase2 = self._phase2, None
            dist = self.dist
            m = min(int(self.counts[k].max().item()), self.capacity)          # (the host waits for the counts here)
            send = self.buf[k].view(self.pps, self.capacity, 3)[:, :m].contiguous().view(self.pps * m, 3)
            recv = self.allbuf[k].view(-1)[:self.world * self.pps * m * 3].view(self.world * self.pps * m, 3)
            if m:
                if self.on_cpu:
                    parts = [self.torch.empty_like(send) for _ in range(self.world)]
                    dist.all_gather(parts, send, group=self.group)
                    recv.copy_(self.torch.cat(parts))
                else:
                    dist.all_gather_into_tensor(recv, send, group=self.group)
                    self.torch.cuda.current_stream().synchronize()
            self._packed = (recv, m, send)
            self.rows_shipped = self.pps * m

    def _start(self, k):
        dist = self.dist
        self._packed = None
        if self.two_phase:
            if self.on_cpu:
                self.pending = [dist.all_gather(list(self.counts[k].split(self.pps)), self.mine[k], group=self.group, async_op=True)]
            else:
                self.pending = [dist.all_gather_into_tensor(self.counts[k], self.mine[k], group=self.group, async_op=True)]
            self._phase2 = k                       # the rows follow at the next submit / finish (_wait)
            self.last = k
            return
        self.rows_shipped = self.pps * self.capacity
        if self.on_cpu:
            self.pending = [dist.all_gather(list(self.counts[k].split(self.pps)), self.mine[k], group=self.group, async_op=True),
                            dist.all_gather(list(self.allbuf[k].split(self.pps * self.capacity)), self.buf[k], group=self.group, async_op=True)]
        else:
            self.pending = [dist.all_gather_into_tensor(self.counts[k], self.mine[k], group=self.group, async_op=True),
                            dist.all_gather_into_tensor(self.allbuf[k], self.buf[k], group=self.group, async_op=True)]
        self.last = k

    def submit(self, packed):
        torch = self.torch
        packed = np.ascontiguousarray(packed, dtype=np.int32).reshape(-1, 3)
        m = packed.shape[0]
        if m > self.capacity:
            raise ValueError("capacity %d smaller than local match count %d" % (self.capacity, m))
        self._wait()                       # at most one gather in flight; frees the other slot
        k = self.slot
        self.slot ^= 1
        if m:
            self.buf[k][:m].copy_(torch.from_numpy(packed))
        self.mine[k][0] = m
        self._start(k)

    def send_buffers(self):
        """(rows int32[pairs_per_step * capacity, 3], counts int64[pairs_per_step]) tensors of the free slot."""
        if self.on_cpu and self.fill_buf is not None:
            return self.fill_buf[self.slot], self.fill_mine[self.slot]
        return self.buf[self.slot], self.mine[self.slot]     # (pure CPU: host tensors the caller fills itself)

    def consumer_stream(self):
        """Raw handle of the stream the collectives of this gatherer are ordered on (torch's current
        stream): what ``Context.match_accepted_dev_async`` takes as ``consumer_stream``.  None on CPU."""
        if self.on_cpu and self.fill_buf is None:
            return None
        return int(self.torch.cuda.current_stream().cuda_stream)        # (0 = the null stream: a stream all the same)

    def submit_device(self):
        """Start the all-gather of the slot handed out by the last ``send_buffers()``.  The fill
        must be complete (fm_match_accepted_dev) or ordered in front of torch's current stream
        (fm_match_accepted_dev_async with ``consumer_stream()``)."""
        self._wait()
        k = self.slot
        self.slot ^= 1
        if self.on_cpu and self.fill_buf is not None:      # dry run: device buffers, CPU transport
            self.buf[k].copy_(self.fill_buf[k])
            self.mine[k].copy_(self.fill_mine[k])
        self._start(k)

    def finish(self):
        self._wait()
        k = self.last
        if self._packed is not None:
            recv, m, _ = self._packed
            if self.pps > 1:
                return self.counts[k].view(self.world, self.pps), recv.view(self.world, self.pps, m, 3)
            return self.counts[k], recv.view(self.world, m, 3)
        if self.pps > 1:
            return self.counts[k].view(self.world, self.pps), self.allbuf[k].view(self.world, self.pps, self.capacity, 3)
        return self.counts[k], self.allbuf[k].view(self.world, self.capacity, 3)
