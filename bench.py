#!/usr/bin/env python3
"""bench.py — benchmarks of the MI355X zstd / S2 encode hot path on the BASELINE.json configurations.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C2H|C3|C4|C5] [--gib G] [--kind T|H|J|M]

Default (what the driver runs): C2 = BASELINE.json configs[1], the configuration the headline metric is quoted on:
zstd SpeedFastest, synthetic enwik-style text 'T' in independent 128 KiB units (each unit == one reference EncodeAll
call: 1 frame, 2 x 64 KiB blocks sharing history), 4 GiB per GPU, inputs resident in HBM before the timed region.
The other configurations print the same JSON shape:
    C2H  the same on the high-entropy corpus 'H' (north_star: "enwik-style and high-entropy buffers")
    C3   zstd SpeedDefault (enc_dfast) on the C2 corpus
    C4   s2.Encode, 64 KiB blocks of synthetic JSON 'J', 2 GiB per GPU (= 16 GiB over 8 GPUs)
    C5   zstd SpeedBetterCompression with a 64 KiB raw dictionary on mixed text+binary 'M', 1 GiB per GPU (= 8 GiB over 8)
A "step" is one pass of the whole hot path (checksum + match finder + entropy/emit + compaction) over the batch; for
N > 1 each rank encodes its own shard (weak scaling) and the compressed frames are gathered to rank 0 over RCCL (the only
exchange step of this path; "gather_verified": rank 0's gathered stream checked against every rank's own frames by digest, after
the timed region).  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
SEEDS = {"T": 0x5EED0001, "H": 0x5EED0002, "J": 0x5EED0003, "M": 0x5EED0004}
DICT_SEED = 0x5EED0005

CONFIGS = {
    "C2": dict(codec="zstd", level=1, kind="T", unit=128 << 10, gib=4.0, dict_kib=0, kernel="kc_zfast_match_grp_kernel<8, false, false>",
               src="kc_zstd_match.hip", what="zstd SpeedFastest EncodeAll"),
    "C2H": dict(codec="zstd", level=1, kind="H", unit=128 << 10, gib=4.0, dict_kib=0, kernel="kc_zfast_match_grp_kernel<8, true, true>",
                src="kc_zstd_match.hip", what="zstd SpeedFastest EncodeAll"),
    "C3": dict(codec="zstd", level=2, kind="T", unit=128 << 10, gib=4.0, dict_kib=0, kernel="kc_zdfast_match_grp_kernel<8>",
               src="kc_zstd_match_dfast.hip", what="zstd SpeedDefault EncodeAll"),
    "C4": dict(codec="s2", level=0, kind="J", unit=64 << 10, gib=2.0, dict_kib=0, kernel="kc_s2_encode_kernel",
               src="kc_s2.hip", what="s2.Encode (default level)"),
    "C5": dict(codec="zstd", level=3, kind="M", unit=128 << 10, gib=1.0, dict_kib=64, kernel="kc_zbetter_match_grp_kernel<true>",
               src="kc_zstd_match_better.hip", what="zstd SpeedBetterCompression EncodeAll, 64 KiB raw dictionary"),
    # C4 matched against the reference's amd64 ASSEMBLY encoders (KC_S2_VARIANT_AMD64) instead of its portable Go encoders: the
    # build an amd64 user of s2.Encode runs, and the one oracle/_ref executes — cpu_baseline.kind "reference", bytes compared
    "C4A": dict(codec="s2", level=0, kind="J", unit=64 << 10, gib=2.0, dict_kib=0, kernel="kc_s2_encode_kernel",
                src="kc_s2.hip", what="s2.Encode, amd64 assembly variant", variant="amd64"),
    # not a BASELINE configuration (and not part of the default run's "also"): the best level, one wave per unit on persistent table slots
    "B4": dict(codec="zstd", level=4, kind="T", unit=128 << 10, gib=0.75, dict_kib=0, kernel="kc_zbest_match_kernel",
               src="kc_zstd_match_best.hip", what="zstd SpeedBestCompression EncodeAll"),
}
METRIC = "encode MB/s (input) + ratio, zstd SpeedFastest 128KiB blocks, 1/2/4/8 GPU"


def kernel_source_hash(name):
    """sha256[:16] of the kernel's source file + the shared device header: stamps PMC summaries so that a stale one is not reported."""
    h = hashlib.sha256()
    for f in (name, "kc_dev.h"):
        h.update(open(os.path.join(ROOT, "compress_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def cpu_quota():
    cores = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2 CPU quota of this container: more runnable threads than this only get throttled
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except Exception:
        pass
    return cores, quota


def collect_pmc(argv, kernel_key):
    """--pmc: re-run this command (1 step, rank 0, N == 1) under rocprofv3 once per counter and return the HBM bytes of the
    dominant kernel's dispatch: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots), both in KB."""
    import glob
    import sqlite3
    import tempfile
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="kcpmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "--", sys.executable, os.path.abspath(__file__)] + argv + [
            "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-device-verify", "--no-end-to-end", "--no-also", "--no-floor"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400, check=False)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            c = sqlite3.connect(dbs[0])
            cols = [x[1] for x in c.execute("pragma table_info(counters_collection)")]
            ik, ic, iv, idp = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
            acc = {}
            for r in c.execute("select * from counters_collection"):
                if r[ic] == ctr and kernel_key in r[ik]:
                    acc[r[idp]] = acc.get(r[idp], 0.0) + float(r[iv])
            res[ctr] = max(acc.values()) * 1024.0 if acc else None
        except Exception as e:  # profiling is best effort: the bench line must still print
            res[ctr] = None
            res["error"] = repr(e)[:200]
        finally:
            subprocess.run(["rm", "-rf", d], check=False)
    return res


def dry_run(args, cfg, rank, world):
    """The multi-rank control flow of main() without a GPU and without the encoder: every rank fabricates the frames of its
    contiguous shard of units (sizes and bytes are a function of the global unit index and the step), gathers them to rank 0
    with the same FrameGather / two-buffer overlap as the timed loop, and rank 0 checks every step's gathered stream against
    what the single-rank order gives.  Timing: barrier, K steps, barrier, MAX over ranks."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from compress_amd.shard import FrameGather, shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 64 * world + 5  # not a multiple of the world size: ragged shards
    lo, hi = shard_range(n_total, rank, world)

    def frame(u, step):  # fabricated frame of global unit u at a step
        n = 1 + (u * 2654435761 + step * 40503) % 997
        return np.full(n, (u + 7 * step) & 0xFF, dtype=np.uint8)

    def shard_bytes(a, b, step):
        return np.concatenate([frame(u, step) for u in range(a, b)]) if b > a else np.zeros(0, dtype=np.uint8)

    bufs = [torch.empty(1000 * (hi - lo) + 16, dtype=torch.uint8) for _ in range(2)]
    gather = FrameGather(rank, world) if world > 1 else None
    verified = True

    def check(step, res):
        nonlocal verified
        if rank != 0:
            return
        want = shard_bytes(0, n_total, step)
        got = res[0].numpy() if res is not None else None
        if got is None or not np.array_equal(got[:len(want)], want) or res[1][-1] != len(want):
            verified = False

    def run(k, step0):
        pending, pstep = None, None
        for i in range(k):
            db = i % 2
            mine = shard_bytes(lo, hi, step0 + i)
            bufs[db][:len(mine)] = torch.from_numpy(mine)  # "encode": the buffer that the gather of step i-2 has released
            if gather is not None:
                if pending is not None:
                    check(pstep, pending.wait())
                pending, pstep = gather.start(bufs[db], len(mine)), step0 + i
            elif rank == 0:
                check(step0 + i, (bufs[db][:len(mine)], [0, len(mine)]))
        if pending is not None:
            check(pstep, pending.wait())

    run(args.warmup, 0)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ok = torch.tensor([1 if verified else 0], dtype=torch.int64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        verified = bool(ok.item())
    if rank == 0:
        print(json.dumps({"metric": cfg["metric"] if "metric" in cfg else "dry run", "value": None, "unit": "MB/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / max(args.steps, 1), 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u8", "data": "fabricated frames", "dry_run": True, "backend": "gloo" if world > 1 else "none",
                          "units_total": n_total, "shard_of_rank0": [lo, hi], "gather_verified": verified,
                          "config": {"workload": "rank plumbing only: sharding, overlapped frame gather over two buffers, barrier + MAX timing"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if verified else 1


def self_launch(n):
    """`python bench.py --gpus N ...` without a launcher's environment: re-run this command as N ranks under
    torch.distributed.run on this node (rendezvous on 127.0.0.1, a free port), stream the ranks' output through, return
    their exit code.  Rank 0 of the child job prints the one JSON line (n_gpus: N)."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs the dmabuf IPC mode on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def main():
    try:  # a faulting GPU process must not fill the box's disk with a core file of the resident batch
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--gib", type=float, default=None, help="input GiB per GPU (default: the configuration's)")
    ap.add_argument("--kind", default=None, help="corpus override (T|H|J|M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-floor", action="store_true", help="skip roofline.floor (a ~0.3 s probe of the table access pattern after the timed region)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive host-buffer measurement after the timed region")
    ap.add_argument("--pipeline", dest="pipeline", action="store_true", default=None,
                    help="zstd: two contexts on two streams, the match finder of step i+1 under the entropy stage of step i — every step still "
                         "encodes the whole batch and all K steps end inside the timed region.  Default for C2 since round 4 (without the "
                         "per-batch table clear it is a gain there: 152.2 vs 156.9 and 154.5 vs 156.6 ms/step on two boxes, "
                         "profiles/r04_pipeline_ab.json; round 2 measured a loss, round 1 no overlap at all); not for the other "
                         "configurations (C3 / C5: no gain; C2H: 3.9 vs 2.9 ms; B4: a second set of 70 GB table slots).")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="one context: steps back to back on one stream")
    ap.add_argument("--contexts", type=int, default=0, help="zstd with --pipeline: contexts in flight (default 2; 3: the entropy stage of step i may drain under the match finders of steps i+1 AND i+2)")
    ap.add_argument("--split", type=int, default=0, help="zstd with --pipeline: every step's batch runs as this many launches of n_units / split units each, going round the contexts; "
                    "each part's frames are put right behind the previous part's (kc_zstd_encode_units_dev_end_at), so the pass produces the same contiguous output (0: per configuration)")
    ap.add_argument("--stage2-priority", type=int, default=0, help="zstd with --pipeline (measurement): 1 = every context's entropy stage and what follows on a HIGH-priority stream of its own "
                    "(KC_OPT_STAGE2_STREAM), the match finders on LOW-priority streams — what the rolling host pipeline's lanes do; 0 = one default-priority stream per context")
    ap.add_argument("--mf-in-flight", type=int, default=0, help="zstd with --pipeline: match finders of consecutive steps allowed on the chip together (0: per configuration; 1: one at a time; "
                    "a kernel that does not fill the CUs with one batch, C5's, can share them with the next batch's)")
    ap.add_argument("--no-device-verify", action="store_true",
                    help="skip the on-device round trip (decode ALL frames on the device + compare with the input)")
    ap.add_argument("--gather", default="root", choices=["root", "none"],
                    help="N > 1: 'root' gathers every step's frames to rank 0 over RCCL (the path's only exchange step, default); "
                         "'none' leaves each rank's contiguous shard of frames on its own GPU (consumers that write per-rank files: "
                         "compress_amd.shard.write_shard; the frames are concatenable in rank order)")
    ap.add_argument("--s2-level", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="C4 only: 0 s2.Encode (the BASELINE configuration), 1 s2.EncodeBetter, 2 s2.EncodeSnappy, 3 s2.EncodeSnappyBetter, "
                         "4 s2.EncodeBest, 5 s2.EncodeSnappyBest")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the rank plumbing of the N > 1 path on the gloo backend with fabricated frames — contiguous sharding, "
                         "FrameGather overlapped with the next step over two buffers, barrier + all_reduce(MAX) timing, one JSON line from "
                         "rank 0 (value null).  Run under torch.distributed.run like the real bench.")
    ap.add_argument("--pmc", action="store_true", help="measure roofline.traffic in this run (two extra rocprofv3 passes of one step each)")
    ap.add_argument("--cpu-sample-units", type=int, default=0, help="units of the CPU-baseline / byte-compare sample (0: per configuration)")
    ap.add_argument("--cpu-ref-units", type=int, default=256, help="units the translated reference encodes for cpu_baseline.reference_translated (one thread, ~35 MB/s)")
    ap.add_argument("--cpu-ref-units-per-thread", type=int, default=128, help="units per thread for cpu_baseline.reference_translated_parallel (the translated reference on one thread per CPU, child process)")
    ap.add_argument("--no-cpu-ref-parallel", action="store_true", help="skip cpu_baseline.reference_translated_parallel")
    ap.add_argument("--cpu-threads", type=int, default=0, help="override the CPU baseline thread count")
    ap.add_argument("--no-also", action="store_true",
                    help="default invocation (C2, 1 GPU): do NOT run the other BASELINE configurations (C2H, C3, C4, C5: 3 steps each at their "
                         "per-GPU sizes, one child process each) after the timed region and attach them as \"also\"")
    ap.add_argument("--also-steps", type=int, default=3)
    ap.add_argument("--path", default="auto", choices=["auto", "hbm", "lds"], help="kernel family of SpeedFastest / s2.Encode (KC_OPT_MATCH_PATH)")
    ap.add_argument("--e2e-calls", type=int, default=4, help="end_to_end steady state: host-buffer calls kept in flight (contexts used with submit / wait)")
    ap.add_argument("--e2e-steps", type=int, default=12, help="end_to_end steady state: batches timed")
    args = ap.parse_args()
    if args.contexts == 1:
        ap.error("--contexts counts the contexts of the two-stage pipeline (2 or 3); for one context with steps back to back use --no-pipeline")
    if args.pipeline is None:
        # C2 since round 4; C3 and C5 since round 6 (same-box A/B, gpurun_out/r6w: C3 272.3 -> 266.8 ms per step with two contexts,
        # C5 64.4 -> 63.2 with three; C2H / B4 lose, S2 has no second stage)
        args.pipeline = args.config in ("C2", "C3", "C5") or (args.config in ("C4", "C4A") and args.s2_level < 4)
        # s2.Encode (no second stage): the 2 GiB batch of 64 KiB blocks is one residency too; as two launches on three contexts
        # (kc_s2_encode_blocks_lvl_dev_begin / _end_at) 43.5 -> 41.1 ms (tools/s2_split_probe.py, gpurun_out/r8k)
        if args.config in ("C4", "C4A") and args.s2_level < 4 and args.contexts == 0 and args.split == 0:
            args.contexts, args.split = 3, 2
        # One 4 GiB batch as TWO launches of half the units, three contexts, two match finders in flight (same-box A/B, gpurun_out/r8h,
        # r8i): C2 151.3 -> 144.4-145.5 ms per 4 GiB (+4.4 %), C3 262-266 -> 244-247 (+7.5 %).  A launch of all 32 768 units is exactly
        # one residency of the chip: every workgroup starts together and the chip drains while the slowest finish; with halves the
        # next half's workgroups take the slots as the previous half's leave them.  (1 GiB quarters: slower — launches too small.)
        if args.config in ("C2", "C3") and args.contexts == 0 and args.split == 0 and args.mf_in_flight == 0:
            args.contexts, args.split, args.mf_in_flight = 3, 2, 2
        if args.config == "C5" and args.contexts == 0:
            args.contexts = 3
        # Two match finders on the chip together (same-box A/B, gpurun_out/r8a-r8c, r8e): C5's kernel fills 8 of 12 wave slots per CU with
        # one batch (a chain of dependent trips per unit, 0.63 of its request floor alone): 57.9 -> 49.2-50.6 ms per step.  C2's and C3's
        # are one residency per batch: a second launch gets nothing until the first ends (kernel trace: 299 ms for a launch beside
        # another) and the steady state is one after the other again (C2 150.9 -> 148.7, C3 277.9 -> 276.9): left at one at a time,
        # whose launch durations are the kernel's own
        # (C5, r8p: the three contexts' match finders unchained — the third takes wave slots as the first one's workgroups leave — 48.6-48.8 ms
        # against 50.5-51.1 with the chain of lag two; C2 within the noise of its box, C3 slower: they keep the chain)
        if args.config == "C5" and args.mf_in_flight == 0 and args.contexts >= 3:
            args.mf_in_flight = args.contexts
    if args.split <= 0 or not args.pipeline:
        args.split = 1
    cfg = dict(CONFIGS[args.config])
    if args.gib is not None:
        cfg["gib"] = args.gib
    if args.kind is not None:
        cfg["kind"] = args.kind
    UNIT = cfg["unit"]
    kind = cfg["kind"]

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` from a bare shell: become the launcher — N ranks of this very command under
        # torch.distributed.run (one process per GPU; rank 0 prints the one JSON line), the form the driver would have typed
        return self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)"
                         % (world, args.gpus, args.gpus, args.gpus))
    if args.dry_run:
        return dry_run(args, cfg, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from compress_amd import _lib, zstd, s2
    from compress_amd.shard import FrameGather

    n_units = int(cfg["gib"] * (1 << 30)) // UNIT
    first_unit = rank * n_units  # contiguous shard per rank keeps output order == concatenation order
    t0 = time.time()
    host = _lib.corpus_fill(kind, SEEDS[kind], first_unit, n_units, UNIT)
    gen_s = time.time() - t0
    d_src = torch.from_numpy(host).cuda(local_rank)
    unit_off = np.arange(n_units + 1, dtype=np.uint64) * UNIT
    in_bytes = n_units * UNIT
    dict_content = None
    if cfg["dict_kib"]:
        # one dictionary for the whole job: rank 0 builds it, the others receive it (dist.broadcast over RCCL)
        dict_content = _lib.corpus_fill("T", DICT_SEED, 0, 1, cfg["dict_kib"] << 10).tobytes() if rank == 0 else b""
        if world > 1:
            from compress_amd.shard import broadcast_bytes
            dict_content = broadcast_bytes(dict_content, torch.device("cuda", local_rank))

    is_s2 = cfg["codec"] == "s2"
    npipe = (args.contexts if args.contexts >= 2 else 2) if (args.pipeline and (not is_s2 or (args.split > 1 and args.s2_level < 4))) else 1
    streams = [torch.cuda.Stream() for _ in range(npipe)]
    streams2 = []
    if args.stage2_priority and npipe >= 2 and cfg["codec"] != "s2":
        streams = [torch.cuda.Stream(priority=0) for _ in range(npipe)]    # (ROCm: 0 = low / normal, -1 = high)
        streams2 = [torch.cuda.Stream(priority=-1) for _ in range(npipe)]
    if is_s2:
        encs = [s2.BlockEncoder(device=local_rank, stream=st.cuda_stream, level=args.s2_level, path=args.path, variant=cfg.get("variant")) for st in streams]
        cfg["what"] = {0: cfg["what"], 1: "s2.EncodeBetter", 2: "s2.EncodeSnappy", 3: "s2.EncodeSnappyBetter", 4: "s2.EncodeBest", 5: "s2.EncodeSnappyBest"}[args.s2_level]
        cfg["kernel"] = "kc_s2_encode_kernel<%d>" % args.s2_level if args.s2_level < 4 else "kc_s2_best_kernel<%s>" % ("true" if args.s2_level == 5 else "false")
        if args.s2_level >= 4:
            cfg["src"] = "kc_s2_best.hip"
        slot = (s2.MaxEncodedLen(UNIT) + 15) & ~15
    else:
        zopts = [zstd.WithEncoderLevel(cfg["level"]), zstd.WithMatchPath(args.path)]
        if dict_content:
            zopts.append(zstd.WithEncoderDictRaw(1, dict_content))
        encs = [zstd.NewWriter(None, *zopts, device=local_rank, stream=st.cuda_stream) for st in streams]
        for e_, s2_ in zip(encs, streams2):
            e_.ctx().set_option(_lib.OPT_STAGE2_STREAM, s2_.cuda_stream)
        slot = (encs[0].MaxEncodedSize(UNIT) + 15) & ~15
    enc = encs[0]
    cap = n_units * slot + 64
    # N > 1: the gather of step i reads one buffer while step i+1 fills the other; with two contexts step i+2 is begun while the gather
    # of step i may still be reading: a third buffer
    ndst = npipe + 1 if (npipe >= 2 and world > 1) else (npipe if npipe >= 2 else (2 if world > 1 else 1))
    d_dsts = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(ndst)]
    gather = FrameGather(rank, world, bound_bytes=cap) if (world > 1 and args.gather == "root") else None
    mf_lag = max(1, args.mf_in_flight)
    S = args.split if npipe >= 2 else 1
    if is_s2 and npipe >= 2 and S < 2:
        ap.error("s2 with --pipeline runs through the begin / end_at pair: give --split >= 2")
    cuts = [n_units * h // S for h in range(S + 1)]
    inflight = 1 if npipe < 2 else (npipe if (is_s2 or mf_lag >= npipe) else mf_lag)  # dominant-kernel launches that may be on the chip together
    if npipe >= 2 and mf_lag < npipe and not is_s2:  # at most mf_lag match finders at a time (default one): context j's waits for context j-mf_lag's
        for j in range(npipe):
            encs[j].ChainAfter(encs[(j - mf_lag) % npipe])
    ctx0 = enc._ctx if is_s2 else enc.ctx()
    info = ctx0.device_info()
    torch.cuda.synchronize()

    def run_steps(k):
        """k passes over the batch; returns (offsets of the last pass, its buffer index, per-pass kernel timings, per-pass wall ms).
        N > 1: the RCCL gather of step i's frames to rank 0 is posted after step i and completed after step i+1's encode,
        so the transfer over xGMI overlaps the next step's kernels; the last gather completes inside the timed region."""
        tms, walls, off, last = [], [], None, 0
        if k <= 0:
            return off, last, tms, walls
        pending = None
        tw = time.perf_counter()
        begun = 0
        if S > 1:
            # the batch of every step as S launches (parts), the parts of consecutive steps going round the contexts; a part's frames go
            # right behind the previous part's, so a step leaves the same contiguous frames + offsets as one launch would
            def begin_part(g):
                a, b = cuts[g % S], cuts[g % S + 1]
                if is_s2:
                    encs[g % npipe].EncodeBlocksDeviceBegin(d_src.data_ptr(), unit_off[a:b + 1])
                else:
                    encs[g % npipe].EncodeUnitsDeviceBegin(d_src.data_ptr(), unit_off[a:b + 1], d_dsts[(g // S) % ndst].data_ptr(), cap)
            for i in range(k):
                db = i % ndst
                pos = 0
                off = np.zeros(n_units + 1, dtype=np.uint64)
                for h in range(S):
                    g = i * S + h
                    while begun < k * S and begun < g + npipe:
                        begin_part(begun)
                        begun += 1
                    if is_s2:
                        o = encs[g % npipe].EncodeBlocksDeviceEnd(d_dsts[db].data_ptr() + pos, cap - pos)
                    else:
                        o = encs[g % npipe].EncodeUnitsDeviceEnd(d_dsts[db].data_ptr() + pos, cap - pos)
                    a, b = cuts[h], cuts[h + 1]
                    off[a + 1:b + 1] = o[1:] + np.uint64(pos)
                    pos += int(o[b - a])
                    tms.append((encs[g % npipe]._ctx if is_s2 else encs[g % npipe].ctx()).timings())
                if gather is not None:
                    if pending is not None:
                        pending.wait()
                    pending = gather.start(d_dsts[db], int(off[n_units]))
                last = db
                tn = time.perf_counter()
                walls.append((tn - tw) * 1e3)
                tw = tn
            if pending is not None:
                pending.wait()
            return off, last, tms, walls
        if not is_s2:
            encs[0].EncodeUnitsDeviceBegin(d_src.data_ptr(), unit_off, d_dsts[0].data_ptr(), cap)
            begun = 1
        for i in range(k):
            cur = i % npipe
            db = i % ndst
            if is_s2:
                off = enc.EncodeBlocksDevice(d_src.data_ptr(), unit_off, d_dsts[db].data_ptr(), cap)
                tms.append(ctx0.timings())
            else:
                while npipe >= 2 and begun < k and begun < i + npipe:  # the match finders of the next npipe - 1 steps are enqueued before this step's second half
                    encs[begun % npipe].EncodeUnitsDeviceBegin(d_src.data_ptr(), unit_off, d_dsts[begun % ndst].data_ptr(), cap)
                    begun += 1
                off = encs[cur].EncodeUnitsDeviceEnd()
                tms.append(encs[cur].ctx().timings())
            if gather is not None:
                if pending is not None:
                    pending.wait()
                pending = gather.start(d_dsts[db], int(off[n_units]))
            if not is_s2 and npipe == 1 and i + 1 < k:
                encs[0].EncodeUnitsDeviceBegin(d_src.data_ptr(), unit_off, d_dsts[(i + 1) % ndst].data_ptr(), cap)
            last = db
            tn = time.perf_counter()  # the encode call returned the offsets: this step's frames are complete on the device
            walls.append((tn - tw) * 1e3)
            tw = tn
        if pending is not None:
            pending.wait()
        return off, last, tms, walls

    if npipe >= 2:
        # set-up, not warm-up: every context encodes one launch-sized part once, one after the other, so that none allocates its scratch
        # (seconds for SpeedBetter's 32 GiB of tables) inside the timed region when --warmup is smaller than the number of contexts
        for j in range(npipe):
            a, b = cuts[j % S], cuts[j % S + 1]
            if is_s2:
                encs[j].EncodeBlocksDeviceBegin(d_src.data_ptr(), unit_off[a:b + 1])
                encs[j].EncodeBlocksDeviceEnd(d_dsts[0].data_ptr(), cap)
            else:
                encs[j].EncodeUnitsDeviceBegin(d_src.data_ptr(), unit_off[a:b + 1], d_dsts[0].data_ptr(), cap)
                encs[j].EncodeUnitsDeviceEnd()
        torch.cuda.synchronize()
    run_steps(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_off, last_buf, ktimes, walls = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    d_dst = d_dsts[last_buf]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt * 1000.0 / args.steps
    out_bytes = int(out_off[n_units])
    value = world * in_bytes / (ms_per_step / 1000.0) / 1e6  # MB/s (1e6) of input, whole job

    # ---- N > 1, outside the timed region: the gathered stream IS the ranks' frames in rank order.  Every rank digests the frames of
    # its last step on its own GPU (two wrapping 64-bit sums, one position-weighted), the digests travel in one small all_gather,
    # the last step's frames are gathered once more and rank 0 digests each segment of what arrived ----
    gather_ok = None
    if gather is not None:
        def digest(t):
            """two wrapping 64-bit sums of a byte tensor (one position-weighted) + its length, in 64 MiB pieces (no full-size temporaries)"""
            n = t.numel()
            s0, s1, pos, step = 0, 0, 0, 64 << 20
            for a0 in range(0, n, step):
                piece = t[a0:min(n, a0 + step)]
                m = piece.numel()
                pad = torch.zeros((m + 7) // 8 * 8, dtype=torch.uint8, device=t.device)
                pad[:m].copy_(piece)
                v = pad.view(torch.int64)
                w = (torch.arange(v.numel(), dtype=torch.int64, device=t.device) + pos) * 2654435761 + 1
                s0 = (s0 + int(v.sum().item())) & 0xFFFFFFFFFFFFFFFF
                s1 = (s1 + int((v * w).sum().item())) & 0xFFFFFFFFFFFFFFFF
                pos += v.numel()
            to_i64 = lambda x: x - (1 << 64) if x >= (1 << 63) else x
            return [to_i64(s0), to_i64(s1), n]
        # only the LOCAL digest may fail quietly: a rank that skipped a collective would leave the others waiting in it.  A
        # failed digest travels as a sentinel; every rank enters both collectives.
        local_err = None
        try:
            mine_l = digest(d_dst[:out_bytes])
        except Exception as e:
            mine_l, local_err = [0, 0, -1], repr(e)[:200]
        mine = torch.tensor(mine_l, dtype=torch.int64, device="cuda")
        allv = torch.empty(3 * world, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(allv, mine)
        res = gather.start(d_dst, out_bytes).wait()
        if rank == 0:
            try:
                g_out, g_offs = res
                want = allv.cpu().tolist()
                if any(x < 0 for x in want[2::3]):
                    gather_ok = "error: a rank could not digest its frames" + ((": " + local_err) if local_err else "")
                else:
                    gather_ok = g_offs[-1] == sum(want[2::3])
                    for r in range(world):
                        gather_ok = gather_ok and digest(g_out[g_offs[r]:g_offs[r + 1]]) == want[3 * r:3 * r + 3]
                    gather_ok = bool(gather_ok)
            except Exception as e:  # the timed result must still be reported
                gather_ok = "error: " + repr(e)[:200]

    # ---- roofline of the dominant kernel, from HIP events on the launch stream (kc_last_timings) ----
    tm = ktimes[-1]
    # the AVERAGE launch duration over the timed steps (what rocprofv3 --stats reports under the kernel's name); the per-step values are in the line
    k_match = float(np.mean([t["match_ms"] - t.get("prep_ms", 0.0) for t in ktimes]))  # (match_ms brackets table preparation + kernel)
    k_entropy = float(np.median([t["entropy_ms"] for t in ktimes]))
    k_total = float(np.median([t["total_ms"] for t in ktimes]))
    k_prep = float(np.median([t.get("prep_ms", 0.0) for t in ktimes]))
    algo_bytes = in_bytes + out_bytes  # 1 B read + ratio B written per input byte
    achieved = (algo_bytes / S) / (k_match / 1000.0) / 1e9  # one LAUNCH processes n_units / S units (--split)
    if args.config == "C2H":
        # high-entropy input: the match finder skips most bytes and no kernel dominates; the "dominant kernel" of this
        # configuration is the whole pipeline (match finder + entropy stage + checksum-and-copy + compaction of the headers)
        cfg["kernel"] = "pipeline: %s + kc_zstd_entropy_kernel + kc_xxh64_fin_kernel (checksum + raw payload copy) + kc_compact_kernel" % cfg["kernel"]
        k_match = k_total
        achieved = algo_bytes / (k_total / 1000.0) / 1e9
    # HBM traffic of the dominant kernel: measured in this run with --pmc (two rocprofv3 passes), else taken from the committed
    # PMC summary when it was collected on this workload AND on this kernel source (sha256 stamp), else null.  FETCH_SIZE is
    # reported raw: the x2 correction of MI355X_MICROARCH.md was calibrated on wide coalesced reads; tools/mem_probe.hip shows
    # this kernel's scattered 4-byte accesses are 64-byte requests (TCC_EA0_RDREQ_32B == 0), which FETCH_SIZE tallies at 64 B.
    khash = kernel_source_hash(cfg["src"])
    traffic, traffic_src = None, None
    if args.pmc and rank == 0 and world == 1:
        argv = ["--config", args.config, "--gib", str(cfg["gib"]), "--kind", kind, "--s2-level", str(args.s2_level), "--path", args.path, "--no-pipeline"]
        pm = collect_pmc(argv, cfg["kernel"].split("<")[0])
        if pm.get("FETCH_SIZE") and pm.get("WRITE_SIZE"):
            traffic, traffic_src = int(pm["FETCH_SIZE"] + pm["WRITE_SIZE"]), "measured in this run (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, 1 step each)"
    if traffic is None:
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            ents_ = pj.get("entries", [pj])
            # a launch of this run holds n_units / S units: the counters of a dispatch of exactly that size if the summary has one (collected with
            # --gib halved, one launch per step), else those of the whole-batch dispatch scaled (per-unit requests do not depend on the launch size)
            exact_ = S > 1 and any(e.get("config") == args.config and e.get("units") == n_units // S and e.get("corpus") == kind and e.get("kernel_source_sha16") == khash
                                   and (not is_s2 or not e.get("kernel") or e["kernel"] == cfg["kernel"]) for e in ents_)
            want_units = n_units // S if exact_ else n_units
            for ent in ents_:
                if ent.get("config") != args.config or ent.get("units") != want_units or ent.get("corpus") != kind:
                    continue
                if is_s2 and ent.get("kernel") and ent["kernel"] != cfg["kernel"]:  # (the S2 levels share a configuration name and, at 2 GiB, a size)
                    continue
                if ent.get("kernel_source_sha16") == khash:
                    traffic, traffic_src = ent["kernel_hbm_bytes"], "profiles/pmc_traffic.json (same workload, same kernel source)"
                    if S > 1 and not exact_:  # the counters were collected on the dispatch of all units; a launch of this run holds 1 / S of them
                        traffic = int(traffic / S)
                        traffic_src += "; per launch of %d units = the %d-unit dispatch's bytes / %d" % (n_units // S, n_units, S)
                    elif S > 1:
                        traffic_src += "; a dispatch of %d units — the size of this run's launches — running alone" % (n_units // S)
                        whole_ = [e for e in ents_ if e.get("config") == args.config and e.get("units") == n_units and e.get("corpus") == kind and e.get("kernel_source_sha16") == khash
                                  and (not is_s2 or not e.get("kernel") or e["kernel"] == cfg["kernel"])]
                        if whole_:  # beside another launch the caches are as full as under the whole batch's dispatch: its bytes / S bound the figure from above
                            traffic_src += "; the %d-unit dispatch's bytes / %d = %d" % (n_units, S, int(whole_[0]["kernel_hbm_bytes"] / S))
                elif khash in ent.get("also_valid_for_sha16", []):
                    # collected on an earlier form of this kernel's source; the entry says why it still describes the running one
                    traffic = ent["kernel_hbm_bytes"]
                    traffic_src = "profiles/pmc_traffic.json (same workload; measured on source %s: %s)" % (ent.get("kernel_source_sha16"), ent.get("note", ""))
        except Exception:
            pass
    roofline = {"bound": "hbm", "kernel": cfg["kernel"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "kernel_source_sha16": khash,
                "kernel_ms": round(k_match, 3), "kernel_ms_steps": [round(t["match_ms"] - t.get("prep_ms", 0.0), 2) for t in ktimes][:64], "table_prep_ms": round(k_prep, 3), "entropy_kernel_ms": round(k_entropy, 3), "pipeline_kernel_ms": round(k_total, 3),
                "pipeline_frac": round((algo_bytes / S) / (k_total / 1000.0) / 1e9 / HBM_PEAK_GBS, 5),
                # the step's algorithmic bytes over the step's wall time (every stage of the step, not the dominant kernel alone)
                "step_frac": round(algo_bytes / (ms_per_step / 1000.0) / 1e9 / HBM_PEAK_GBS, 5),
                "read_only_frac": round((in_bytes / S) / (k_match / 1000.0) / 1e9 / HBM_PEAK_GBS, 5)}
    # ---- the floor of this design: the dominant kernel's DRAM requests per dispatch (TCC_EA0_RDREQ / WRREQ,
    # profiles/r06_transactions.json, stamped with the kernel's source hash) priced at the request rates THIS box gives the same access
    # pattern right now (kc_probe_table_pattern: scattered 4-byte read + write-back pairs / plain reads into per-unit tables of the
    # configuration's geometry).  Every table store is one DRAM write behind a table read of the same line (a pair); the reads beyond
    # that (candidate bytes, source lines) are priced as plain scattered reads.  frac_of_floor = floor_ms / kernel_ms.
    if rank == 0 and world == 1 and not args.no_floor:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r06_transactions.json")))
            ent = [e for e in tj["entries"] if e["config"] == args.config and e["units"] == n_units and e["corpus"] == kind]
            if ent and args.config in ("C2", "C3", "C4", "C5") and (not is_s2 or args.s2_level == 0):
                ent = ent[0]
                pr = ctx0.probe_table_pattern(n_units, ent["table_bytes_per_unit"], 4096, 4096)
                rd, wr = ent["rdreq_per_dispatch"], ent["wrreq_per_dispatch"]
                floor_ms = (wr / pr["pairs_per_s"] + max(0.0, rd - wr) / pr["reads_per_s"]) * 1e3
                k_alone = k_match  # with two contexts the event bracket includes the overlapped entropy stage: the also-line C2/one-context carries the kernel alone
                shared = inflight >= 2 and ms_per_step < S * k_match
                if shared:  # launches share the chip (C5; the halves of C2 / C3): the time one batch's requests have is the step, not the launches (which also wait for CUs)
                    k_alone = ms_per_step
                roofline["floor"] = {
                    "transactions_per_unit": round((rd + wr) / n_units, 1), "reads_per_unit": ent["rdreq_per_unit"], "writes_per_unit": ent["wrreq_per_unit"],
                    "measured_pairs_per_s": round(pr["pairs_per_s"]), "measured_reads_per_s": round(pr["reads_per_s"]),
                    "measured_tx_per_s": round(2 * pr["pairs_per_s"]),
                    "floor_ms": round(floor_ms, 2), "kernel_ms": round(k_alone, 3), "frac_of_floor": round(floor_ms / k_alone, 3),
                    "model": "writes x (1 / pair rate) + (reads - writes) x (1 / scattered read rate); rates from kc_probe_table_pattern on this box in this run (%d tables of %d KiB, 4096 waves); requests from profiles/r06_transactions.json" % (n_units, ent["table_bytes_per_unit"] >> 10),
                    "transactions_source_current": ent["kernel_source_sha16"] == khash,
                    "kernel_ms_is": "ms_per_step (two launches share the chip: a launch lasts longer than a step; the step also holds the other stages, so this fraction is a lower bound)" if shared else "the launch duration",
                    "note": "a bit-exact %s keeps one hash table per unit in HBM (%d GiB live): every probe is a DRAM read and a DRAM write-back of a 64-byte line that carries 4 useful bytes; the kernel runs at the request ceiling of that pattern, not at a byte roofline" % (cfg["what"], (n_units * ent["table_bytes_per_unit"]) >> 30)}
        except Exception as e:  # the timed result must still be reported
            roofline["floor"] = {"error": repr(e)[:200]}
    if npipe >= 2 and is_s2:
        roofline["overlap_note"] = "%d contexts, the launches unchained: kernel_ms is one launch's duration with the other launches beside it" % npipe
    elif npipe >= 2:  # several steps in flight: the event brackets of one step's kernels contain the other steps' work
        roofline["overlap_note"] = ("%d contexts: kernel_ms is the match finder's duration WITH the previous step's entropy stage running beside it (alone: "
                                    "--no-pipeline); entropy_kernel_ms / pipeline_kernel_ms span the match finder they run under and do not add up to ms_per_step" % npipe)
    if npipe >= 2:
        if S > 1:
            roofline["launches_per_step"] = S
            roofline["overlap_note"] += "; every step's batch runs as %d launches of %d units (kernel_ms, achieved and traffic are per launch)" % (S, n_units // S)
        if inflight >= 2:
            # Launches of the dominant kernel overlap in time.  The contract's figure (bytes of ONE launch / its duration) is kept as
            # achieved_per_launch / frac_per_launch; a launch's duration then contains the time it shares the CUs with the other launch(es), so
            # the kernel's rate is that figure times the launches in flight on average = S x kernel_ms / ms_per_step, measured in this run
            # (with more than one in flight throughout, that product is the step's bytes over the step's time).
            conc = S * k_match / ms_per_step
            roofline["match_finders_in_flight"] = inflight
            roofline["launches_in_flight_avg"] = round(conc, 3)
            roofline["achieved_per_launch"] = roofline["achieved"]
            roofline["frac_per_launch"] = roofline["frac"]
            if conc > 1.0:
                roofline["achieved"] = round(achieved * conc, 2)
                roofline["frac"] = round(achieved * conc / HBM_PEAK_GBS, 5)
                roofline["read_only_frac_per_launch"] = roofline["read_only_frac"]
                roofline["read_only_frac"] = round(roofline["read_only_frac"] * conc, 5)
            roofline["overlap_note"] += ("; up to %d launches of the dominant kernel are on the chip together (%.2f on average = launches per step x kernel_ms / ms_per_step), so "
                                         "kernel_ms contains the time a launch shares the CUs: achieved_per_launch / frac_per_launch are one launch's bytes over its duration "
                                         "(what rocprofv3's average duration gives), achieved / frac that figure times the launches in flight on average" % (inflight, conc))

    # ---- CPU baseline (rank 0, N == 1 only): the oracle restatement of the reference on the host threads + byte compare ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            import oracle_lib
            cores, quota = cpu_quota()
            host_threads = cores
            if quota is not None and quota < cores:
                cores = quota
            if args.cpu_threads > 0:
                cores = args.cpu_threads
            default_sample = {"C2": 16384, "C2H": 16384, "C3": 8192, "C4": 16384, "C4A": 16384, "C5": 2048, "B4": 256}[args.config]
            sample = min(n_units, args.cpu_sample_units or default_sample)
            kind_cpu = "port"
            if is_s2 and cfg.get("variant") == "amd64":
                import oracle_ref  # oracle/_ref: the reference's own assembly encoders
                kind_cpu = "reference"

                def cpu_once():
                    return oracle_ref.encode_blocks(host[:sample * UNIT], unit_off[:sample + 1], level=args.s2_level, threads=cores)
            elif is_s2:
                def cpu_once():
                    return oracle_lib.s2_encode_blocks(host[:sample * UNIT], unit_off[:sample + 1], threads=cores, level=args.s2_level)
            else:
                kw = dict(level=cfg["level"])
                if dict_content:
                    kw.update(dict_id=1, dict_content=dict_content)

                def cpu_once():
                    return oracle_lib.zstd_encode_units(host[:sample * UNIT], unit_off[:sample + 1], threads=cores, **kw)
            # a measurement, not a single cold call: one untimed pass (thread start-up, page faults of the output; its bytes are the
            # parity sample), then 3 timed measurements of >= 1 s each (the sample looped), the median reported with the spread
            t0 = time.perf_counter()
            ref, ref_off = cpu_once()
            first = time.perf_counter() - t0
            loops = max(1, int(1.0 / max(first, 1e-4)) + 1) if first < 1.0 else 1
            rates = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _k in range(loops):
                    cpu_once()
                rates.append(loops * sample * UNIT / (time.perf_counter() - t0) / 1e6)
            rates.sort()
            cpu = {"value": round(rates[1], 1), "unit": "MB/s", "cores": cores, "kind": kind_cpu,
                   "spread": {"min": round(rates[0], 1), "median": round(rates[1], 1), "max": round(rates[2], 1), "first_cold_call": round(sample * UNIT / first / 1e6, 1)},
                   "sample": "first %d units (%.2f GiB) of the same corpus, %d std::threads (host has %d hardware threads%s); median of 3 measurements of %d pass%s each (>= 1 s of work per measurement) after one untimed pass"
                             % (sample, sample * UNIT / 2**30, cores, host_threads, ", cgroup cpu.max allows %d CPUs" % quota if quota else "", loops, "" if loops == 1 else "es")}
            if host_threads > cores:
                cpu["note"] = "the host has %d hardware threads; this container may run %d: a projection to the whole host (x%.1f) is NOT a measurement" % (host_threads, cores, host_threads / cores)
            got = d_dst[:int(out_off[sample])].cpu().numpy()
            parity = bool(np.array_equal(got, np.asarray(ref)) and np.array_equal(out_off[:sample + 1], ref_off))
            # The reference's OWN encoder beside the port (zstd configurations): its Go source translated statement by statement to
            # C++ (oracle/ref_go -> oracle/_ref/libzstdref*.so; no Go toolchain in this image, so this is not the Go compiler's code
            # generation), one thread, one pooled encoder re-used across the units (encoder.go:722 with its pool) — and a second byte
            # compare of the device's frames, against the reference itself.  A crash-safe leg: single-threaded, as the tests use it.
            if not is_s2:
                try:
                    import oracle_goref
                    if oracle_goref.available():
                        ns = min(sample, args.cpu_ref_units)
                        units_ = [bytes(host[i * UNIT:(i + 1) * UNIT]) for i in range(ns)]
                        kwr = dict(level=cfg["level"])
                        if dict_content:
                            kwr.update(dict_id=1, dict_content=dict_content)
                        fl = "amd64" if oracle_goref.amd64_available() else "generic"
                        with oracle_goref.flavour(fl):
                            t0 = time.perf_counter()
                            frames_ = oracle_goref.zstd_encode_all_reuse(units_, **kwr)
                            dt_ = time.perf_counter() - t0
                        same_ = all(bytes(got[int(out_off[i]):int(out_off[i + 1])]) == frames_[i] for i in range(ns))
                        cpu["reference_translated"] = {"value": round(ns * UNIT / dt_ / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "reference-translated",
                                                       "device_bytes_equal": bool(same_),
                                                       "sample": "first %d units through the reference's own zstd.Encoder.EncodeAll (Go source translated to C++ at build time, %s flavour: oracle/ref_go), one thread, one pooled encoder; one pass" % (ns, fl)}
                except Exception as e:
                    cpu["reference_translated"] = {"error": repr(e)[:200]}
                # ... and its goroutine-parallel form (north_star: "the reference's own goroutine-parallel CPU path"): one thread per
                # CPU this container may run, each with ONE encoder of the reference re-used from unit to unit (what N goroutines on a
                # zstd.Encoder get from its pool, encoder.go:90-99, 722-729).  A child process (tools/ref_parallel.py): a fault of the
                # translated runtime under threads must not take this line along.  The child's frames are compared by digest.
                if "value" in (cpu.get("reference_translated") or {}) and not args.no_cpu_ref_parallel:
                    try:
                        np_ = int(min(sample, cores * args.cpu_ref_units_per_thread))
                        cmd = [sys.executable, os.path.join(ROOT, "tools", "ref_parallel.py"), "--kind", kind, "--seed", hex(SEEDS[kind]),
                               "--first-unit", str(first_unit), "--units", str(np_), "--unit", str(UNIT), "--level", str(cfg["level"]),
                               "--threads", str(cores), "--passes", "2"]
                        if dict_content:
                            cmd += ["--dict-kib", str(cfg["dict_kib"]), "--dict-seed", hex(DICT_SEED)]
                        pr_ = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
                        if pr_.returncode != 0:
                            cpu["reference_translated_parallel"] = {"error": "child rc %d: %s" % (pr_.returncode, pr_.stderr.decode(errors="replace")[-200:])}
                        else:
                            rp = json.loads(pr_.stdout.decode().strip().splitlines()[-1])
                            if "value" in rp:
                                dig = hashlib.sha256(got[:int(out_off[np_])].tobytes()).hexdigest()
                                rp = {"value": rp["value"], "unit": "MB/s", "cores": rp["cores"], "kind": "reference-translated, one thread per CPU",
                                      "device_bytes_equal": bool(dig == rp["frames_sha256"] and int(out_off[np_]) == rp["frames_bytes"]),
                                      "sample": "first %d units, %d threads with one pooled encoder of the reference each (%s flavour), best of %d passes after an untimed one (passes: %s s); child process tools/ref_parallel.py"
                                                % (np_, rp["cores"], rp["flavour"], len(rp["passes_s"]), rp["passes_s"])}
                            cpu["reference_translated_parallel"] = rp
                    except Exception as e:
                        cpu["reference_translated_parallel"] = {"error": repr(e)[:200]}
        except Exception as e:  # the timed result must still be reported
            cpu = {"error": repr(e)[:300]}

    # ---- on-device round trip of EVERY frame of this rank (outside the timed region): decode + checksum + compare ----
    verified = None
    verify_ms = None
    if not args.no_device_verify:
        try:
            d_back = torch.empty(in_bytes + 64, dtype=torch.uint8, device="cuda")
            t0 = time.perf_counter()
            if is_s2:
                st = enc.DecodeBlocksDevice(d_dst.data_ptr(), out_off, d_back.data_ptr(), unit_off)
            else:
                st = enc.DecodeUnitsDevice(d_dst.data_ptr(), out_off, d_back.data_ptr(), unit_off, dict_content=dict_content)
            torch.cuda.synchronize()
            verify_ms = (time.perf_counter() - t0) * 1e3
            verified = bool((not st.any()) and torch.equal(d_back[:in_bytes], d_src))
            del d_back
        except Exception as e:
            verified = "error: " + repr(e)[:300]

    # ---- PCIe-inclusive rate of the host-buffer entry points (what the cgo shim calls; include/kcgpu.h), outside the timed region.
    # Source and destination are pageable host memory, the destination pre-faulted.  Large calls go through the device's rolling
    # pipeline (compress_amd/csrc/kc_roll.cpp): sub-batches of every call in flight are staged, encoded on lanes and drained in
    # arrival order.  Reported: (1) ONE call over the whole batch (latency form: transfer of the first sub-batch and drain of the last
    # are exposed); (2) the steady state of a caller that keeps --e2e-calls calls in flight through kc_*_submit / kc_wait on as many
    # contexts (SURVEY 8b "async submit/wait"), the same number of batches per second the device-resident `value` counts; (3) this
    # box's ceilings: pinned H2D / D2H rates and the pageable <-> pinned host copy rate (kc_probe_pcie). ----
    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        try:
            import ctypes as C
            ref_out = d_dst[:out_bytes].cpu().numpy()
            del d_dsts, d_dst, d_src  # the timed region's buffers: the host path brings its own
            d_dsts = d_dst = d_src = None
            torch.cuda.empty_cache()
            for e_ in encs:  # the timed region's contexts keep their handles; their device scratch (tens of GiB each) goes back: the
                (e_._ctx if is_s2 else e_.ctx()).trim()  # host path encodes on the rolling pipeline's own lanes
            pc = ctx0.probe_pcie(1 << 30)
            ncall = max(1, args.e2e_calls)
            if is_s2:
                eencs = [enc] + [s2.BlockEncoder(device=local_rank, level=args.s2_level, path=args.path, variant=cfg.get("variant")) for _ in range(ncall - 1)]
                ectx = [e._ctx for e in eencs]
            else:
                eencs = [enc] + [zstd.NewWriter(None, *zopts, device=local_rank) for _ in range(ncall - 1)]
                ectx = [e.ctx() for e in eencs]
            h_dsts = [np.zeros(cap, dtype=np.uint8) for _ in range(ncall)]
            eos = [np.zeros(n_units + 1, dtype=np.uint64) for _ in range(ncall)]

            def ecall(i, submit):
                c = ectx[i]
                if is_s2:
                    f = c.L.kc_s2_encode_blocks_lvl_submit if submit else c.L.kc_s2_encode_blocks_lvl
                    c.check(f(c.h, args.s2_level, host.ctypes.data, unit_off.ctypes.data, n_units, h_dsts[i].ctypes.data, cap, eos[i].ctypes.data))
                else:
                    f = c.L.kc_zstd_encode_units_submit if submit else c.L.kc_zstd_encode_units
                    c.check(f(c.h, C.byref(eencs[i].o), host.ctypes.data, unit_off.ctypes.data, n_units, h_dsts[i].ctypes.data, cap, eos[i].ctypes.data))

            def ewait(i):
                ectx[i].check(ectx[i].L.kc_wait(ectx[i].h))

            for _w in range(2):  # warm: every lane of the engine has held a sub-batch of this shape (its scratch is allocated) before a clock starts
                for i in range(ncall):
                    ecall(i, True)
                for i in range(ncall):
                    ewait(i)
            one = []
            for _ in range(3):
                t0 = time.perf_counter()
                ecall(0, False)
                one.append(time.perf_counter() - t0)
            ksteps = max(ncall, args.e2e_steps)
            t0 = time.perf_counter()
            sub = 0
            for k in range(ksteps):  # ncall calls in flight: call k + ncall - 1 is submitted before call k is waited for
                while sub < ksteps and sub < k + ncall:
                    ecall(sub % ncall, True)
                    sub += 1
                ewait(k % ncall)
            steady = (time.perf_counter() - t0) / ksteps
            same = all(bool(np.array_equal(eos[i], out_off)) and bool(np.array_equal(h_dsts[i][:out_bytes], ref_out)) for i in range(ncall))
            dev_rate = value / world  # MB/s, device-resident
            pcie_rate = min(pc["h2d_bidir_GBps"], pc["host_copy_in_GBps"]) * 1e3  # MB/s of input the link can take while frames flow back
            rate = in_bytes / steady / 1e6
            e2e = {"value": round(rate, 1), "unit": "MB/s", "frac_of_device_resident": round(rate / dev_rate, 3),
                   "ms_per_batch": round(steady * 1e3, 2), "calls_in_flight": ncall, "batches_timed": ksteps,
                   "sample": "all %d units (%.2f GiB) per call from pageable host memory into pageable host memory through kc_%s_submit / kc_wait on %d contexts, %d calls in flight, %d batches timed after warm-up; the calls' sub-batches share the device's rolling pipeline (staging, 8 encoder lanes on 4 queues, drain)"
                             % (n_units, in_bytes / 2**30, "s2_encode_blocks_lvl" if is_s2 else "zstd_encode_units", ncall, ncall, ksteps),
                   "single_call": {"value": round(in_bytes / min(one) / 1e6, 1), "unit": "MB/s", "ms": round(min(one) * 1e3, 2), "frac_of_device_resident": round(in_bytes / min(one) / 1e6 / dev_rate, 3),
                                   "note": "one synchronous kc_%s call over the whole batch, best of 3: the first sub-batch's transfer and the last one's drain are exposed" % ("s2_encode_blocks_lvl" if is_s2 else "zstd_encode_units")},
                   "pcie_ceiling": {"h2d_GBps": round(pc["h2d_GBps"], 2), "d2h_GBps": round(pc["d2h_GBps"], 2), "h2d_with_d2h_GBps": round(pc["h2d_bidir_GBps"], 2),
                                    "d2h_with_h2d_GBps": round(pc["d2h_bidir_GBps"], 2), "host_copy_in_GBps": round(pc["host_copy_in_GBps"], 2),
                                    "host_copy_out_GBps": round(pc["host_copy_out_GBps"], 2), "copy_threads": pc["copy_threads"],
                                    "note": "pinned 1 GiB copies each way, alone and both at once; pageable <-> pinned memcpy with the library's copy threads (kc_probe_pcie, this box, this run)"},
                   "frac_of_min_device_pcie": round(rate / min(dev_rate, pcie_rate), 3),
                   "same_bytes_as_device_path": same}
            for e in eencs[1:]:
                e.Close()
            del h_dsts
        except Exception as e:
            e2e = {"error": repr(e)[:300]}

    # ---- the other BASELINE configurations, driver-visible: one child process each, after the timed region ----
    also = None
    if rank == 0 and world == 1 and args.config == "C2" and not args.no_also and args.gib is None and args.kind is None:
        d_dsts = d_dst = d_src = None
        torch.cuda.empty_cache()
        # the children get the device to themselves: this process keeps its handles but gives the memory back (each context's scratch
        # for the 4 GiB batch, and what the rolling host pipeline's slots and lanes hold after end_to_end: ~230 GiB together)
        try:
            for e_ in encs:
                (e_._ctx if is_s2 else e_.ctx()).trim()
            _lib.device_trim(local_rank)
        except Exception:
            pass
        also = {}
        # the five BASELINE configurations at their per-GPU sizes, then the N3 levels at small sizes (not BASELINE configurations:
        # zstd SpeedBestCompression, s2.EncodeBetter, s2.EncodeBest) so that the driver's one run times those too
        for name, extra in (("C2/one-context", ["--no-pipeline"]), ("C2H", []), ("C3", []), ("C4", []), ("C4A", []), ("C5", []), ("B4", ["--gib", "0.75"]),
                            ("C4/s2.EncodeBetter", ["--s2-level", "1"]), ("C4/s2.EncodeBest", ["--s2-level", "4", "--gib", "1.5"])):  # (s2.EncodeBest: 24 576 blocks = one residency at 4 blocks per wave, 6 waves per SIMD; B4: 6 144 units = the default table slots)
            # (C5: three contexts with two match finders in flight — a launch lasts longer than a step, so 3 steps would be mostly fill and drain)
            cmd = [sys.executable, os.path.abspath(__file__), "--config", name.split("/")[0],
                   # (launches of consecutive steps share the chip: 3 steps would be mostly fill and drain)
                   "--steps", str(max(args.also_steps, 9) if name == "C5" else max(args.also_steps, 6) if name == "C3" else max(args.also_steps, 12) if (name.startswith("C4") and "Best" not in name) else args.also_steps),
                   "--warmup", "3" if (name == "C5" or (name.startswith("C4") and "Best" not in name)) else "2" if name == "C3" else "1",
                   "--no-also", "--cpu-sample-units", "1024" if not extra else "256", "--path", args.path] + extra
            if name not in ("C3", "C4", "C5"):  # the host-buffer rate of every BASELINE configuration; not of the side lines
                cmd.append("--no-end-to-end")
            if name == "C2/one-context":  # like for like with rounds 1-3 (one context, steps back to back): no second CPU baseline
                cmd += ["--no-cpu-baseline", "--no-device-verify"]
            t0 = time.perf_counter()
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, check=False)
                js = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
                j = json.loads(js[-1])
                also[name] = {"workload": j["config"]["workload"], "contexts": j.get("contexts"), "value": j["value"], "unit": j["unit"], "steps": j["steps"], "ms_per_step": j["ms_per_step"],
                              "ms_per_step_spread": j.get("ms_per_step_spread"),
                              "ratio": j["ratio"], "roofline": j["roofline"], "bit_exact_vs_oracle_on_sample": j["bit_exact_vs_oracle_on_sample"],
                              "device_roundtrip_all_frames": j["device_roundtrip_all_frames"], "cpu_baseline": j["cpu_baseline"], "end_to_end": j.get("end_to_end"),
                              "wall_s": round(time.perf_counter() - t0, 1)}
            except Exception as e:
                also[name] = {"error": repr(e)[:300]}

    if rank == 0:
        wl = "%s, %.2f GiB/GPU synthetic '%s' corpus in %d KiB %s%s, device-resident" % (
            cfg["what"], cfg["gib"], kind, UNIT >> 10,
            "blocks" if is_s2 else "units (2 x 64 KiB blocks with history)" if cfg["level"] == 1 else "units",
            "" if not dict_content else ", dictionary = 64 KiB of corpus 'T'")
        # two contexts: a step's call returns when its entropy stage ends, and that stage runs under the NEXT step's match finder — so the
        # first call of the timed region spans two match finders and the last one only an entropy stage (the pipeline filling and
        # draining; both are inside the timed region and in ms_per_step).  The spread is taken over the steps in between.
        walls_s = walls[npipe - 1:-(npipe - 1)] if (npipe >= 2 and len(walls) > 2 * npipe) else walls
        line = {
            "metric": METRIC if args.config in ("C2", "C2H") else "encode MB/s (input) + ratio, %s" % cfg["what"],
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": round(float(np.median(walls_s)), 3),
            # the spread over the timed steps of THIS run on THIS box (boxes of the pool differ by more than steps do: DESIGN.md 5)
            "ms_per_step_spread": {"min": round(float(np.min(walls_s)), 3), "median": round(float(np.median(walls_s)), 3), "max": round(float(np.max(walls_s)), 3)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl, "name": args.config, "units_per_gpu": n_units, "unit_bytes": UNIT, "corpus": kind,
                       "parallelism": ("units sharded contiguously over %d GPU(s); %s" % (world, "RCCL gather of frames to rank 0" if gather is not None else "no gather: every rank keeps its shard of frames")) if world > 1 else "1 GPU",
                       "pipeline": ("%d contexts / %d streams: every step's batch as %d launches of %d blocks going round the contexts (begin / end_at), unchained; blocks contiguous as from one launch" % (npipe, npipe, S, n_units // S)) if (is_s2 and npipe >= 2) else (("%d contexts / %d streams: " % (npipe, npipe)) + (
                                        "every step's batch as %d launches of %d units going round the contexts, up to %d match finders on the chip together, the entropy stage of a launch under the following launches' match finders; frames contiguous as from one launch" % (S, n_units // S, mf_lag) if S > 1 else
                                        "match finder of step i+1 overlaps the entropy stage of step i%s" % ("" if not (2 <= mf_lag < npipe) else " and the match finder of step i+%d" % (mf_lag - 1))) if npipe >= 2
                                    else "none: steps back to back on one stream"),
                       "device": info},
            "split": S,  # launches per step (--split): the batch's units in S parts, frames contiguous as from one launch
            "contexts": npipe,  # machine-readable: 2 = the two-context pipeline (kernel timings of consecutive steps overlap), 1 = steps back to back
            "ratio": round(out_bytes / in_bytes, 5),
            "value_GiBps": round(value * 1e6 / 2**30, 3),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "end_to_end": e2e,
            "bit_exact_vs_oracle_on_sample": parity,
            "device_roundtrip_all_frames": verified,
            "gather_verified": gather_ok,  # N > 1: rank 0's gathered stream == every rank's frames, by digest (null at N = 1 / --gather none)
            "device_roundtrip_ms": None if verify_ms is None else round(verify_ms, 1),
            "redo_units": tm["redo_units"],
            "host": {"gen_s": round(gen_s, 2), "nproc": os.cpu_count()},
            "match_path": ctx0.last_path(),
        }
        if also is not None:
            line["also"] = also
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
