"""Multi-GPU sharding of independent units and the final framed-stream gather.

The encode path has no cross-unit dependency, so units are partitioned contiguously over
ranks (GPU g of G gets units [g*N/G, (g+1)*N/G), keeping output order == concatenation
order) and encoded with no data-path collective.  The only exchange step is the gather of
the already framed, concatenable zstd frames (zstd/encoder.go:719-720) to the root:
an all_gather of the per-rank byte counts followed by point-to-point sends over RCCL/xGMI
(each peer has a direct link to the root).  Works with the gloo backend on CPU tensors
too, which is how it is tested without GPUs.
"""
import torch
import torch.distributed as dist


def shard_range(n_units, rank, world):
    """Contiguous unit range [lo, hi) of `rank`."""
    return (n_units * rank) // world, (n_units * (rank + 1)) // world


def gather_sizes_start(nbytes, device):
    """Post ONE all_gather_into_tensor of the per-rank byte counts (a device-side collective, nothing read back yet)."""
    world = dist.get_world_size()
    mine = torch.tensor([int(nbytes)], dtype=torch.int64, device=device)
    allsz = torch.empty(world, dtype=torch.int64, device=device)
    work = dist.all_gather_into_tensor(allsz, mine, async_op=True)
    return work, allsz, mine


def gather_sizes_finish(h):
    """Complete it and read the counts with one copy (a single host synchronisation, not one per rank)."""
    work, allsz, _mine = h
    work.wait()
    return [int(x) for x in allsz.tolist()]


def gather_sizes(nbytes, device):
    return gather_sizes_finish(gather_sizes_start(nbytes, device))


def broadcast_bytes(data, device, root=0):
    """The dictionary of a sharded job comes from ONE place: rank `root` passes its bytes, every rank gets them back
    ("Dict is broadcast once").  `data` is ignored on the other ranks."""
    rank = dist.get_rank()
    n = torch.tensor([len(data) if rank == root else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=root)
    nb = int(n.item())
    if nb == 0:  # an empty dictionary (or none): nothing to send — and torch.frombuffer refuses a zero-length buffer
        return b""
    buf = torch.empty(nb, dtype=torch.uint8, device=device)
    if rank == root:
        buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    dist.broadcast(buf, src=root)
    return bytes(buf.cpu().numpy().tobytes())


def gather_frames(buf, nbytes, rank, world, root=0):
    """Gather the first `nbytes` bytes of each rank's uint8 tensor `buf` to `root`.

    Returns (gathered uint8 tensor, list of per-rank byte offsets [world+1]) on root, None elsewhere.
    """
    sizes = gather_sizes(nbytes, buf.device)
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    if rank == root:
        out = torch.empty(offs[-1], dtype=torch.uint8, device=buf.device)
        out[offs[root]:offs[root + 1]].copy_(buf[:nbytes])
        reqs = []
        for r in range(world):
            if r == root or sizes[r] == 0:
                continue
            reqs.append(dist.irecv(out[offs[r]:offs[r + 1]], src=r))
        for q in reqs:
            q.wait()
        return out, offs
    if nbytes > 0:
        dist.isend(buf[:nbytes], dst=root).wait()
    return None


class FrameGather:
    """The same gather, split so that it overlaps the next batch's encode: start() exchanges the byte counts and posts the
    point-to-point transfers, wait() completes them (and, on CUDA, blocks the host until the bytes have landed, so that the
    source buffer may be overwritten and the result read).  One gather in flight per object; the root keeps one receive
    buffer and grows it on demand."""

    def __init__(self, rank, world, root=0, bound_bytes=None):
        """bound_bytes: an upper bound of one rank's frames (the sum of MaxEncodedSize of its units): the root then allocates its
        receive buffer once, world x bound, instead of growing it when a step's frames turn out larger."""
        self.rank, self.world, self.root = rank, world, root
        self._bound = bound_bytes
        self._out = None
        self._reqs = None
        self._offs = None

    def start(self, buf, nbytes):
        assert self._reqs is None, "previous gather not waited for"
        h = gather_sizes_start(nbytes, buf.device)
        if self.rank == self.root and self._out is None and self._bound is not None:
            self._out = torch.empty(max(self.world * int(self._bound), 1), dtype=torch.uint8, device=buf.device)  # under the collective
        sizes = gather_sizes_finish(h)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        self._offs = offs
        self._reqs = []
        self._dev = buf.device
        ops = []
        if self.rank == self.root:
            if self._out is None or self._out.numel() < offs[-1]:
                self._out = torch.empty(max(offs[-1], 1), dtype=torch.uint8, device=buf.device)
            self._out[offs[self.root]:offs[self.root + 1]].copy_(buf[:nbytes])
            for r in range(self.world):
                if r == self.root or sizes[r] == 0:
                    continue
                ops.append(dist.P2POp(dist.irecv, self._out[offs[r]:offs[r + 1]], r))
        elif nbytes > 0:
            ops.append(dist.P2POp(dist.isend, buf[:nbytes], self.root))
        # one group per rank: over RCCL the root's receives from all peers proceed concurrently (one xGMI link each)
        # instead of one after the other
        if ops:
            self._reqs = dist.batch_isend_irecv(ops)
        return self

    def wait(self):
        """Returns (gathered uint8 tensor, per-rank byte offsets) on the root, None elsewhere."""
        if self._reqs is None:
            return None
        for q in self._reqs:
            q.wait()
        if self._dev.type == "cuda":
            torch.cuda.current_stream(self._dev).synchronize()
        self._reqs = None
        if self.rank == self.root:
            return self._out[:self._offs[-1]], self._offs
        return None


def write_shard(prefix, rank, buf, nbytes, out_off=None):
    """No-gather mode for consumers that only need the shards: rank r writes its own frames to `<prefix>.<r:05d>.zst`
    (and, when given, the frame offsets of its units to `<prefix>.<r:05d>.idx` as little-endian uint64).  zstd frames are
    concatenable (zstd/encoder.go:719-720) and the shards are contiguous unit ranges, so `cat <prefix>.*.zst` in rank order is
    the stream rank 0 would have gathered; nothing crosses xGMI."""
    import numpy as np
    path = "%s.%05d.zst" % (prefix, rank)
    data = buf[:nbytes]
    if hasattr(data, "cpu"):
        data = data.cpu().numpy()
    with open(path, "wb") as f:
        f.write(memoryview(np.ascontiguousarray(data, dtype=np.uint8)))
    if out_off is not None:
        with open("%s.%05d.idx" % (prefix, rank), "wb") as f:
            f.write(np.ascontiguousarray(out_off, dtype="<u8").tobytes())
    return path


def selftest(device):
    """World-size-1 exercise of every collective this module uses, on the real backend (RCCL on a GPU box): the byte-count
    all_gather, a grouped send + receive (to self), a broadcast.  The N > 1 path cannot run on a one-GPU box; this at least runs
    its calls through the backend every round.  Needs an initialised process group.  Returns a short report string."""
    sizes = gather_sizes(1234, device)
    assert sizes == [1234], sizes
    src = torch.arange(4096, dtype=torch.int64, device=device).to(torch.uint8)
    dst = torch.zeros(4096, dtype=torch.uint8, device=device)
    me = dist.get_rank()
    p2p = "skipped (gloo has no send-to-self)"
    if dist.get_backend() == "nccl":  # RCCL: a grouped send + receive to the own rank
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, src, me), dist.P2POp(dist.irecv, dst, me)])
        for q in reqs:
            q.wait()
        torch.cuda.current_stream(device).synchronize()
        assert torch.equal(src, dst), "send-to-self did not deliver the bytes"
        p2p = "ok"
    assert broadcast_bytes(b"dictionary", device) == b"dictionary"
    g = FrameGather(me, dist.get_world_size(), bound_bytes=4096)
    g.start(src, 1000)
    out, offs = g.wait()
    assert offs == [0, 1000] and torch.equal(out, src[:1000])
    return "all_gather_into_tensor ok, batch_isend_irecv to self %s, broadcast ok, FrameGather ok (backend %s)" % (p2p, dist.get_backend())
