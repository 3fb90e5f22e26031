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


def gather_sizes(nbytes, device):
    world = dist.get_world_size()
    mine = torch.tensor([int(nbytes)], dtype=torch.int64, device=device)
    allsz = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allsz, mine)
    return [int(t.item()) for t in allsz]


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

    def __init__(self, rank, world, root=0):
        self.rank, self.world, self.root = rank, world, root
        self._out = None
        self._reqs = None
        self._offs = None

    def start(self, buf, nbytes):
        assert self._reqs is None, "previous gather not waited for"
        sizes = gather_sizes(nbytes, buf.device)
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
