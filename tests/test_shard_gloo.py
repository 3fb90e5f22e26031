"""Multi-rank path on CPU (gloo, world_size 2): contiguous sharding + variable-size frame gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from compress_amd.shard import shard_range, gather_frames
    lo, hi = shard_range(n_units, rank, world)
    # stand-in for encoded frames: unit i "compresses" to (i % 7) + 1 bytes of value i
    parts = [np.full((i % 7) + 1, i & 0xFF, dtype=np.uint8) for i in range(lo, hi)]
    mine = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    buf = torch.from_numpy(np.concatenate([mine, np.zeros(16, dtype=np.uint8)]))  # capacity > used
    got = gather_frames(buf, len(mine), rank, world)
    if rank == 0:
        out, offs = got
        ret["out"] = out.numpy().copy()
        ret["offs"] = offs
    else:
        assert got is None
    # overlappable form: three gathers back to back through one object (sizes differ per round), results identical
    from compress_amd.shard import FrameGather
    fg = FrameGather(rank, world)
    for rnd in range(3):
        cut = max(0, len(mine) - rnd)
        h = fg.start(buf, cut)
        got2 = h.wait()
        if rank == 0:
            out2, offs2 = got2
            ret["out2_%d" % rnd] = out2.numpy().copy()
            ret["offs2_%d" % rnd] = list(offs2)
        else:
            assert got2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [0, 1, 5, 64, 1001])
def test_shard_and_gather_world2(n_units):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_units, ret), nprocs=world, join=True)
    want = np.concatenate([np.full((i % 7) + 1, i & 0xFF, dtype=np.uint8) for i in range(n_units)]) if n_units else np.zeros(0, dtype=np.uint8)
    assert np.array_equal(ret["out"], want)
    assert ret["offs"][0] == 0 and ret["offs"][-1] == len(want) and len(ret["offs"]) == world + 1
    assert np.array_equal(ret["out2_0"], want) and ret["offs2_0"] == list(ret["offs"])
    for rnd in (1, 2):  # every rank dropped its last `rnd` bytes
        o = ret["offs2_%d" % rnd]
        assert len(ret["out2_%d" % rnd]) == o[-1] and o[-1] <= len(want)


def test_shard_range_partitions():
    from compress_amd.shard import shard_range
    for n in (0, 1, 7, 8, 32768, 32769):
        for w in (1, 2, 3, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            for a, b in zip(rs, rs[1:]):
                assert a[1] == b[0]
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
