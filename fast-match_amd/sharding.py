"""Multi-GPU layout of the hot path: independent image pairs are sharded over ranks
(one process per GPU), matching needs no communication, and the per-pair match lists are
returned with ONE variable-length all-gather (RCCL over xGMI on GPUs; gloo in the CPU
tests).  The reference has no distributed code; this is the data-parallel axis it implies
(turntable.py:59 maps the matcher over independent pairs) -- SURVEY.md 8(e).
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


class MatchGatherer(object):
    """Overlapped result gather for a stream of steps: ``submit(packed)`` starts the
    all-gather of this step's accepted matches asynchronously (the previous one is waited
    for first, buffers are double-buffered), so the collective runs while the next step's
    matching kernels execute on the library's own HIP stream.  ``finish()`` waits for the
    last one and returns ``(counts int64[world], rows int32[world, capacity, 3])`` tensors."""

    def __init__(self, device, capacity, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.dev = device
        self.capacity = int(capacity)
        self.on_cpu = isinstance(device, str) and device == "cpu"
        mk = lambda *shape, dtype: [torch.zeros(shape, dtype=dtype, device=device) for _ in range(2)]
        self.buf = mk(self.capacity, 3, dtype=torch.int32)
        self.mine = mk(1, dtype=torch.int64)
        self.allbuf = mk(self.world * self.capacity, 3, dtype=torch.int32)
        self.counts = mk(self.world, dtype=torch.int64)
        self.pending = []
        self.slot = 0

    def _wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def submit(self, packed):
        torch, dist = self.torch, self.dist
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
        if self.on_cpu:
            self.pending = [dist.all_gather(list(self.counts[k].split(1)), self.mine[k], group=self.group, async_op=True),
                            dist.all_gather(list(self.allbuf[k].split(self.capacity)), self.buf[k], group=self.group, async_op=True)]
        else:
            self.pending = [dist.all_gather_into_tensor(self.counts[k], self.mine[k], group=self.group, async_op=True),
                            dist.all_gather_into_tensor(self.allbuf[k], self.buf[k], group=self.group, async_op=True)]
        self.last = k

    def finish(self):
        self._wait()
        k = self.last
        return self.counts[k], self.allbuf[k].view(self.world, self.capacity, 3)
