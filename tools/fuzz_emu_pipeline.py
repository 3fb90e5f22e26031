"""Fuzz of the device's SpeedFastest EncodeAll pipeline ON THE WAVE EMULATOR against the oracle (no GPU): random structured inputs —
small alphabets, runs, periodic patterns, mutated corpus slices, concatenations — through checksum kernel, match finder (LDS-table
kernel and both forms of the group kernel), entropy stage, and batch_end's scan / checksum-and-copy / compaction sequence, with
random frame options.  A difference is a bug in the device code (or in the emulator); the failing input is saved.

    python tools/fuzz_emu_pipeline.py --seconds 3600 --seed 1 [--out profiles/r03b_fuzz_emu_pipeline.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import corpora  # noqa: E402
import emu_lib  # noqa: E402
import oracle_lib  # noqa: E402


def gen_piece(rng, pool):
    kind = int(rng.integers(0, 8))
    n = int(rng.choice([rng.integers(0, 40), rng.integers(0, 600), rng.integers(0, 9000), rng.integers(0, 70000)]))
    if kind == 0:
        a = int(rng.choice([1, 2, 3, 4, 16, 64, 256]))
        return bytes(rng.integers(0, a, n, dtype=np.uint8))
    if kind == 1:
        return bytes([int(rng.integers(0, 256))]) * n
    if kind == 2:
        p = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        return (p * (n // len(p) + 1))[:n]
    if kind == 3:
        src = pool[int(rng.integers(0, len(pool)))]
        o = int(rng.integers(0, max(1, len(src) - n)))
        return src[o:o + n]
    if kind == 4:  # a corpus slice with sparse mutations
        src = pool[int(rng.integers(0, len(pool)))]
        o = int(rng.integers(0, max(1, len(src) - n)))
        b = bytearray(src[o:o + n])
        for _ in range(int(rng.integers(0, 1 + len(b) // 50))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        return bytes(b)
    if kind == 5:  # period with drift: long matches that break
        p = bytearray(rng.integers(0, 8, int(rng.integers(4, 300)), dtype=np.uint8))
        out = bytearray()
        while len(out) < n:
            out += p
            p[int(rng.integers(0, len(p)))] = int(rng.integers(0, 8))
        return bytes(out[:n])
    if kind == 6:
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    return b""


def gen_unit(rng, pool):
    parts = [gen_piece(rng, pool) for _ in range(int(rng.integers(1, 6)))]
    if rng.integers(0, 4) == 0 and parts:  # an earlier piece again: far matches, repeat offsets
        parts.append(parts[int(rng.integers(0, len(parts)))])
    return b"".join(parts)[:300000]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--level", type=int, default=1, choices=[1, 2, 3, 4], help="zstd level (1: also both kernel families and the fused checksum path)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    pool = [corpora.corpus(k, 2, 131072, first_unit=f).tobytes() for k, f in (("T", 11), ("J", 12), ("M", 13), ("M", 14))]
    t0 = time.time()
    stats = dict(batches=0, units=0, bytes=0, redo_units=0, raw_fused=0, failures=0)
    fails = []
    while time.time() - t0 < args.seconds:
        units = [gen_unit(rng, pool) for _ in range(int(rng.integers(4, 24)))]
        okw, ekw = {}, {}
        if rng.integers(0, 3) == 0:
            okw["crc"] = ekw["crc"] = False
        r = int(rng.integers(0, 4))
        if r == 1:
            okw["single"], ekw["single"] = True, 1
        elif r == 2:
            okw["single"], ekw["single"] = False, 0
        if rng.integers(0, 4) == 0:
            w = 1 << int(rng.integers(10, 17))
            okw["window_size"] = w
            ekw["window"], ekw["block_size"] = w, min(w, 65536 if args.level == 1 else 131072)
        if rng.integers(0, 5) == 0:
            okw["full_zero"] = ekw["full_zero"] = False
        if rng.integers(0, 6) == 0:
            okw["no_entropy"] = ekw["no_entropy"] = True
        if rng.integers(0, 6) == 0 and args.level <= 2:  # (on by default above SpeedDefault)
            okw["all_lit_entropy"] = ekw["all_lit_entropy"] = True
        stream = rng.integers(0, 4) == 0
        finder = ["lds", "grp", "grp-tuned"][int(rng.integers(0, 3))]
        fused = None if (stream or rng.integers(0, 2) == 0) else int(rng.integers(0, 3))
        if args.level != 1:
            finder = "lds"  # (ignored: the level's own match finder runs)
            units = [u[:120000] for u in units[:10]] if args.level == 4 else units
        ref = oracle_lib.ZstdOracle(level=args.level, **okw)
        mes = (lambda n: ref.max_encoded_size(n) + 8)
        res = emu_lib.zstd_frames(units, use_grp=finder != "lds", tuned=int(finder == "grp-tuned"), stream_mode=int(stream), fused=fused,
                                  max_encoded_size=mes, level=args.level, **ekw)
        frames, err, redo = res[0], res[1], res[2]
        stats["batches"] += 1
        stats["units"] += len(units)
        stats["bytes"] += sum(map(len, units))
        if fused is not None:
            stats["raw_fused"] += res[3]
        if redo:  # a unit asked for the speculation re-run: the host re-runs such units (tests/test_redo_path.py); not comparable here
            stats["redo_units"] += 1
            continue
        want = [ref.encode_stream(u) if stream else ref.encode_all(u) for u in units]
        bad = [i for i in range(len(units)) if frames[i] != want[i]]
        if err or bad:
            stats["failures"] += 1
            tag = "fuzz_fail_seed%d_batch%d" % (args.seed, stats["batches"])
            np.save("/tmp/%s.npy" % tag, np.array([np.frombuffer(u, dtype=np.uint8) for u in units], dtype=object), allow_pickle=True)
            fails.append((tag, err, bad[:5], okw, finder, stream, fused, [len(units[i]) for i in bad[:5]]))
            print("FAIL", fails[-1], flush=True)
    line = "level %d, seed %d: %.0f s, %d batches, %d units, %.1f MB, %d batches skipped for a re-run request, %d raw-only frames through the fused copy, %d failures %r" % (
        args.level, args.seed, time.time() - t0, stats["batches"], stats["units"], stats["bytes"] / 1e6, stats["redo_units"], stats["raw_fused"], stats["failures"], fails)
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")
    return 1 if stats["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
