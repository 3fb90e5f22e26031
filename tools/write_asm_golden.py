"""Writes tests/golden/reference_amd64_sha256.txt: sha256 of what the REFERENCE's own amd64 assembly encoders (oracle/_ref) write
for seeded corpora — s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter over blocks of every size class — and XXH64 values.
Run here (where /root/reference exists): python tools/write_asm_golden.py.  The committed file lets the oracle (CPU test) and the
device (GPU test) be gated on the reference's bytes on machines that have neither the reference nor the built library."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpora
import oracle_ref

NAMES = {0: "s2", 1: "s2better", 2: "s2snappy", 3: "s2snappybetter"}
# (blocks, block length): every size class of s2/encode_amd64.go, and the multi-MiB classes
SHAPES = [(64, 65536), (48, 300), (48, 2000), (48, 9000), (32, 40000), (8, 1 << 20), (2, 4 << 20), (1, (4 << 20) - 1)]


# "A<k>": random bytes over k symbols — short matches at short offsets, i.e. every form of emitRepeat / emitCopy in every size class
# (added in round 3 after tools/fuzz_emu_s2.py found the oracle's restatement of encodeBetterBlockAsm8B using a repeat form the assembly
# does not have below 512 bytes: none of the corpora above produced a 9..11-byte repeat there)
ASHAPES = [(64, 100), (64, 300), (64, 511), (32, 2000), (16, 9000), (8, 40000)]


def blocks_of(kind, n, ln):
    if kind[0] == "A":
        return corpora.small_alphabet_blocks(int(kind[1:]), n, ln)
    buf = corpora.corpus(kind, (n * ln + 131071) // 131072, 131072, first_unit=11)
    return buf[:n * ln], np.arange(n + 1, dtype=np.uint64) * ln


def main():
    out = []
    for kind in "JTMH":
        for n, ln in SHAPES:
            buf, off = blocks_of(kind, n, ln)
            for level in range(4):
                enc, _ = oracle_ref.encode_blocks(buf, off, level=level, threads=8)
                out.append("%s.amd64.%s.%dx%d %s" % (NAMES[level], kind, n, ln, hashlib.sha256(enc.tobytes()).hexdigest()))
    for kind in ("A2", "A4", "A8"):
        for n, ln in ASHAPES:
            buf, off = blocks_of(kind, n, ln)
            for level in range(4):
                enc, _ = oracle_ref.encode_blocks(buf, off, level=level, threads=8)
                out.append("%s.amd64.%s.%dx%d %s" % (NAMES[level], kind, n, ln, hashlib.sha256(enc.tobytes()).hexdigest()))
    rng = np.random.default_rng(99)
    for n in (0, 1, 3, 4, 7, 8, 31, 32, 33, 63, 64, 1000, 131072):
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        out.append("xxh64.rand99.%d %016x" % (n, oracle_ref.xxh64(b)))
    path = os.path.join(ROOT, "tests", "golden", "reference_amd64_sha256.txt")
    with open(path, "w") as f:
        f.write("# written by tools/write_asm_golden.py from oracle/_ref (the reference's amd64 assembly, assembled here): <name> <sha256 | xxh64>\n")
        f.write("\n".join(out) + "\n")
    print("wrote", len(out), "lines to", path)


if __name__ == "__main__":
    main()
