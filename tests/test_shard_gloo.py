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


def _worker_real(rank, world, port, n_units, usz, steps, tmpdir, ret):
    """The multi-GPU bench loop on CPU: every rank encodes ITS shard of the corpus (the oracle stands in for the device
    encoder — the exchange step does not care who produced the frames), posts the gather of step i after step i's encode and
    completes it after step i+1's encode (two output buffers, as bench.py does), takes the MAX over ranks of the elapsed time."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib
    from compress_amd import _lib
    from compress_amd.shard import shard_range, FrameGather, write_shard
    lo, hi = shard_range(n_units, rank, world)
    host = _lib.corpus_fill("T", 0x5EED0001, lo, hi - lo, usz, threads=2)
    off = np.arange(hi - lo + 1, dtype=np.uint64) * usz
    bufs = [torch.zeros((hi - lo) * (usz + 64) + 64, dtype=torch.uint8) for _ in range(2)]
    fg = FrameGather(rank, world)
    pending, results = None, []
    t0 = time.perf_counter()
    for i in range(steps):
        frames, foff = oracle_lib.zstd_encode_units(host, off, threads=2, level=1)
        db = i % 2
        bufs[db][:len(frames)] = torch.from_numpy(np.asarray(frames))
        if pending is not None:
            results.append(pending.wait())   # step i-1's gather completes only now: its source buffer was bufs[1 - db]
        pending = fg.start(bufs[db], len(frames))
        bufs[1 - db].fill_(0xEE)             # the other buffer is free again: scribble on it
    results.append(pending.wait())
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    write_shard(os.path.join(tmpdir, "shard"), rank, bufs[(steps - 1) % 2], len(frames), foff)
    if rank == 0:
        ret["max_s"] = float(tt.item())
        for i, r in enumerate(results):
            out, offs = r
            ret["out_%d" % i] = out.numpy().copy()
            ret["offs_%d" % i] = list(offs)
    else:
        assert all(r is None for r in results)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_encode_gather_equals_single_rank(tmp_path, oracle):
    """Real frames through the multi-rank path: shard -> encode -> FrameGather (overlapped, double-buffered) -> concatenation
    == the single-rank output, every step; it decodes back; the per-rank-writes mode concatenates to the same stream."""
    world, n_units, usz, steps = 2, 37, 32768, 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_real, args=(world, port, n_units, usz, steps, str(tmp_path), ret), nprocs=world, join=True)
    from compress_amd import _lib
    host = _lib.corpus_fill("T", 0x5EED0001, 0, n_units, usz)
    off = np.arange(n_units + 1, dtype=np.uint64) * usz
    want, want_off = oracle.zstd_encode_units(host, off, threads=4, level=1)
    want = np.asarray(want)
    for i in range(steps):
        assert np.array_equal(ret["out_%d" % i], want), "step %d" % i
        assert ret["offs_%d" % i][-1] == len(want) and len(ret["offs_%d" % i]) == world + 1
    assert ret["max_s"] > 0
    cat = b"".join(open(os.path.join(str(tmp_path), "shard.%05d.zst" % r), "rb").read() for r in range(world))
    assert cat == want.tobytes()
    idx0 = np.frombuffer(open(os.path.join(str(tmp_path), "shard.00000.idx"), "rb").read(), dtype="<u8")
    assert np.array_equal(idx0, want_off[:len(idx0)])
    for u in (0, n_units // 2, n_units - 1):
        frame = want[int(want_off[u]):int(want_off[u + 1])].tobytes()
        assert oracle.zstd_decode(frame, usz + 16) == host[u * usz:(u + 1) * usz].tobytes()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_bench_dry_run_rank_plumbing(world):
    """bench.py --dry-run, launched exactly like the driver launches the real bench (torch.distributed.run, one process per
    rank): ragged contiguous shards, FrameGather overlapped with the next step over two buffers, barrier + all_reduce(MAX)
    timing, one JSON line from rank 0 with every step's gathered stream verified."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable]
    if world > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-run", "--steps", "5", "--warmup", "2"]
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines  # exactly one JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == world and j["steps"] == 5 and j["warmup"] == 2
    assert j["gather_verified"] is True and j["value"] is None and j["scaling"] == "weak"
    assert j["units_total"] == 64 * world + 5 and j["shard_of_rank0"] == [0, (64 * world + 5) // world]


def test_bench_self_launches_its_ranks_from_a_bare_shell():
    """`python3 bench.py --gpus 2 --dry-run` with no launcher environment (the shape of the driver's N-GPU command): bench.py
    re-runs itself as 2 ranks under torch.distributed.run; one JSON line with n_gpus 2 comes back."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["gather_verified"] is True and j["steps"] == 3


def _bcast_payload(n):
    return bytes((i * 7 + 3) & 0xFF for i in range(n))


def _bcast_worker(rank, world, port, nbytes, ret):
    payload = _bcast_payload(nbytes)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from compress_amd.shard import broadcast_bytes
    got = broadcast_bytes(payload if rank == 0 else b"", torch.device("cpu"))
    ret[rank] = bytes(got)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nbytes", [0, 1, 76800])
def test_broadcast_bytes_world2(nbytes):
    """The dictionary broadcast, including the empty dictionary: every rank returns rank 0's bytes."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bcast_worker, args=(world, _free_port(), nbytes, ret), nprocs=world, join=True)
    payload = _bcast_payload(nbytes)
    assert ret[0] == payload and ret[1] == payload


def _rccl_worker(rank, world, port, n_units, usz, ret):
    """Two GPUs, RCCL: each rank encodes its shard on its own device, FrameGather brings the frames to rank 0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from compress_amd import _lib, zstd
    from compress_amd.shard import shard_range, FrameGather, broadcast_bytes
    lo, hi = shard_range(n_units, rank, world)
    host = _lib.corpus_fill("T", 0x5EED0001, lo, hi - lo, usz, threads=2)
    off = np.arange(hi - lo + 1, dtype=np.uint64) * usz
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(zstd.SpeedFastest), device=rank)
    d_src = torch.from_numpy(host).to(dev)
    cap = (hi - lo) * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    fg = FrameGather(rank, world, bound_bytes=cap)
    pending, results = None, []
    for i in range(3):
        out_off = enc.EncodeUnitsDevice(d_src.data_ptr(), off, bufs[i % 2].data_ptr(), cap)
        if pending is not None:
            results.append(pending.wait())
        pending = fg.start(bufs[i % 2], int(out_off[hi - lo]))
    results.append(pending.wait())
    bc = broadcast_bytes(b"dictionary-bytes" * 4096 if rank == 0 else b"", dev)
    assert bytes(bc) == b"dictionary-bytes" * 4096
    if rank == 0:
        for i, r in enumerate(results):
            ret["out_%d" % i] = r[0].cpu().numpy().copy()
            ret["offs_%d" % i] = list(r[1])
    enc.Close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_frame_gather_world2_real_frames():
    """FrameGather + broadcast_bytes over RCCL at world size 2 with real device frames == the single-rank output.  Needs two
    GPUs in one box; on the one-GPU boxes of this pool it is collected and skips."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (torch.cuda.device_count() = %d)" % (torch.cuda.device_count() if torch.cuda.is_available() else 0))
    world, n_units, usz = 2, 301, 128 << 10
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(world, _free_port(), n_units, usz, ret), nprocs=world, join=True)
    from compress_amd import _lib, zstd
    host = _lib.corpus_fill("T", 0x5EED0001, 0, n_units, usz)
    off = np.arange(n_units + 1, dtype=np.uint64) * usz
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(zstd.SpeedFastest), device=0)
    want, want_off = enc.EncodeUnits(host, off)
    want = np.asarray(want)[:int(want_off[n_units])]
    for i in range(3):
        assert np.array_equal(ret["out_%d" % i], want), "step %d" % i
        assert ret["offs_%d" % i][-1] == len(want)


def _selftest_worker(port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from compress_amd import shard
    try:
        q.put(shard.selftest(torch.device("cpu")))
    except Exception as e:  # noqa: BLE001
        q.put("ERR %r" % (e,))
    dist.destroy_process_group()


def test_selftest_world1_gloo():
    """shard.selftest (what __graft_entry__.smoke() runs over RCCL on the GPU box) on gloo, world size 1."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_selftest_worker, args=(29571, q))
    p.start()
    msg = q.get(timeout=120)
    p.join(60)
    assert msg.startswith("all_gather_into_tensor ok"), msg
