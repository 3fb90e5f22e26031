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
