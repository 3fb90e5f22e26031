"""Fuzz of the S2 LDS-table kernel (kc_s2_lds.hip: s2.Encode / s2.EncodeSnappy, both byte-exact targets) ON THE WAVE EMULATOR against the
oracle (no GPU), with the input generators of tools/fuzz_emu_pipeline.py; blocks up to 256 KiB (the 64 KiB class held in LDS and the
larger one), every speculation width class, bare blocks and framed chunks.

    python tools/fuzz_emu_s2.py --seconds 3600 --seed 1 [--out profiles/r03b_fuzz_emu_s2.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import corpora  # noqa: E402
import emu_lib  # noqa: E402
import oracle_lib  # noqa: E402
from fuzz_emu_pipeline import gen_unit  # noqa: E402

try:  # the reference's own assembly (oracle/_ref), where it has been built: the third voice for the amd64 variant
    import oracle_ref
    HAVE_REF = oracle_ref.available()
except Exception:
    HAVE_REF = False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--kernel", default="lds", choices=["lds", "hbm", "best"],
                    help="kc_s2_lds.hip (levels 0, 2), kc_s2.hip (the throughput kernel: levels 0-3) or kc_s2_best.hip (s2.EncodeBest / EncodeSnappyBest)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    pool = [corpora.corpus(k, 2, 131072, first_unit=f).tobytes() for k, f in (("T", 21), ("J", 22), ("M", 23), ("J", 24))]
    t0 = time.time()
    nb = nbytes = nfail = batches = nref = 0
    fails = []
    while time.time() - t0 < args.seconds:
        blocks = [b for b in (gen_unit(rng, pool)[:int(rng.choice([300, 5000, 65536, 65537, 262144]))] for _ in range(int(rng.integers(4, 20)))) if b]
        if not blocks:
            continue
        variant = int(rng.integers(0, 2))
        if args.kernel == "best":  # one wave per block, every candidate of a phase on its own lane: short blocks keep the emulation fast
            blocks = [b[:int(rng.choice([300, 3000, 20000]))] for b in blocks[:8]]
            level, variant, w0 = int(rng.choice([4, 5])), 0, 0
            got = emu_lib.s2_best_blocks(blocks, snappy=level == 5)
            want = [(oracle_lib.s2_encode_snappy_best if level == 5 else oracle_lib.s2_encode_best)(b) for b in blocks]
        elif args.kernel == "hbm":  # kc_s2_encode_kernel<level>: all four levels, three speculation policies
            level = int(rng.integers(0, 4))
            w0, w0b, grow = [(2, 4, 1), (8, 8, 0), (1, 1, 2)][int(rng.integers(0, 3))]
            got = emu_lib.s2_encode_blocks_hbm(blocks, level=level, variant=variant, w0=w0, w0b=w0b, grow=grow)
        else:
            level = int(rng.choice([0, 2]))
            w0 = int(rng.choice([0, 0, 1, 8, 64]))  # 0: the fused step (the default since round 4)
            got = emu_lib.s2_encode_blocks(blocks, level=level, spec_w0=w0, variant=variant)
        if args.kernel == "best":
            pass
        elif variant == 1:
            want = [oracle_lib.s2_encode_asm(b, snappy=level in (2, 3), better=level in (1, 3)) for b in blocks]
        else:
            want = [getattr(oracle_lib, {0: "s2_encode", 1: "s2_encode_better", 2: "s2_encode_snappy", 3: "s2_encode_snappy_better"}[level])(b) for b in blocks]
        bad = [i for i in range(len(blocks)) if got[i] != want[i]]
        if variant == 1 and HAVE_REF:  # oracle restatement vs the assembly itself (a slip of the restatement shows here even when the device agrees with it)
            buf = np.frombuffer(b"".join(blocks), dtype=np.uint8)
            off = np.zeros(len(blocks) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(b) for b in blocks])
            enc, eo = oracle_ref.encode_blocks(buf, off, level=level, threads=1)
            bad += [i for i in range(len(blocks)) if bytes(enc[int(eo[i]):int(eo[i + 1])]) != want[i] and i not in bad]
            nref += len(blocks)
        batches += 1
        nb += len(blocks)
        nbytes += sum(map(len, blocks))
        if bad:
            nfail += 1
            tag = "fuzz_s2_fail_seed%d_batch%d" % (args.seed, batches)
            np.save("/tmp/%s.npy" % tag, np.array([np.frombuffer(b, dtype=np.uint8) for b in blocks], dtype=object), allow_pickle=True)
            fails.append((tag, level, variant, w0, bad[:5], [len(blocks[i]) for i in bad[:5]]))
            print("FAIL", fails[-1], flush=True)
    line = "kernel " + args.kernel + ", seed %d: %.0f s, %d batches, %d blocks, %.1f MB, %d failures %r; %d blocks also against the reference's assembly" % (args.seed, time.time() - t0, batches, nb, nbytes / 1e6, nfail, fails, nref)
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
