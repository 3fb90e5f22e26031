"""Seeded test inputs shared by the CPU and GPU tests (no dependency on /root/reference)."""
import numpy as np

from compress_amd import _lib

SEED_T, SEED_H, SEED_J, SEED_M = 0x5EED0001, 0x5EED0002, 0x5EED0003, 0x5EED0004


def corpus(kind, n_units, unit_size, first_unit=0, seed=None):
    seed = {"T": SEED_T, "H": SEED_H, "J": SEED_J, "M": SEED_M}[kind] if seed is None else seed
    return _lib.corpus_fill(kind, seed, first_unit, n_units, unit_size)


def small_alphabet_blocks(k, n, ln):
    """n blocks of ln random bytes over k symbols (one seeded stream per (k, ln)): short matches at short offsets.  Returns (uint8 buffer,
    block offsets)."""
    rng = np.random.default_rng(0xA1FA0000 + k * 65536 + ln)
    return rng.integers(0, k, n * ln, dtype=np.uint8), np.arange(n + 1, dtype=np.uint64) * ln


def edge_units():
    """Small and pathological units: empty, tiny, RLE, periodic, noise+repeat, boundary sizes."""
    rng = np.random.default_rng(1234)
    units = [b"", b"a", b"ab" * 4, b"abcdefghi", b"0123456789", b"\x00" * 100, b"\x00" * 70000, b"ab" * 40000,
             bytes(rng.integers(0, 256, 1000, dtype=np.uint8)),
             bytes(rng.integers(0, 4, 5000, dtype=np.uint8)),
             b"the quick brown fox jumps over the lazy dog. " * 300,
             bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) * 9,
             bytes(rng.integers(97, 123, 65536, dtype=np.uint8)),
             bytes(rng.integers(97, 101, 65537, dtype=np.uint8)),
             bytes(rng.integers(0, 256, 65535, dtype=np.uint8)),
             (b"x" * 1023), (b"xy" * 512), (b"xyz" * 342)[:1025], b"q" * 131072,
             bytes(rng.integers(0, 2, 131072, dtype=np.uint8)),
             ]
    text = corpus("T", 1, 200000).tobytes()
    for n in (9, 10, 11, 15, 16, 17, 31, 32, 33, 255, 256, 257, 1023, 1024, 1025, 4095, 65535, 65536, 65537, 131071, 131072, 131073, 200000):
        units.append(text[:n])
    return units


def pack_units(units):
    off = np.zeros(len(units) + 1, dtype=np.uint64)
    for i, u in enumerate(units):
        off[i + 1] = off[i] + len(u)
    buf = np.frombuffer(b"".join(units), dtype=np.uint8).copy() if off[-1] else np.zeros(0, dtype=np.uint8)
    return buf, off


def stress_units(seed=7, n=96):
    """Adversarial mixes for the entropy stage: text pieces alternating with noise runs of every length class
    (below / around / far above the cooperative-copy threshold and the 8 KiB literal window), low-entropy noise
    (Huffman-compressible literals without matches), long zero runs, and sizes straddling block boundaries."""
    rng = np.random.default_rng(seed)
    text = corpus("T", 8, 131072, first_unit=900).tobytes()
    js = corpus("J", 2, 131072, first_unit=5).tobytes()
    units = []
    for u in range(n):
        target = int(rng.choice([300, 5000, 40000, 65536, 70000, 131072, 150000, 262144, 300001]))
        parts, tot = [], 0
        while tot < target:
            kind = rng.integers(0, 6)
            if kind == 0:
                ln = int(rng.choice([1, 3, 7, 20, 31, 32, 33, 34, 48, 49, 100, 600, 5000, 9000, 20000]))
                p = bytes(rng.integers(0, 256, ln, dtype=np.uint8))
            elif kind == 1:
                ln = int(rng.integers(4, 3000)); o = int(rng.integers(0, len(text) - ln)); p = text[o:o + ln]
            elif kind == 2:
                ln = int(rng.choice([10, 40, 500, 9000, 30000])); p = bytes(rng.integers(0, int(rng.choice([2, 5, 17, 64])), ln, dtype=np.uint8))
            elif kind == 3:
                ln = int(rng.choice([5, 64, 1000, 70000])); p = bytes([int(rng.integers(0, 256))]) * ln
            elif kind == 4:
                ln = int(rng.integers(4, 2000)); o = int(rng.integers(0, len(js) - ln)); p = js[o:o + ln]
            else:
                p = parts[int(rng.integers(0, len(parts)))] if parts else b"seed"
            parts.append(p); tot += len(p)
        units.append(b"".join(parts)[:target])
    return units


def rle_literal_units(n=12, seed=77):
    """(dictionary content, units) whose blocks carry an RLE LITERALS SECTION (zstd/blockenc.go:554-561, huff0.ErrUseRLE): the
    dictionary is noise; a unit is a chain of 12..24-byte snippets of it, each preceded by `gap` copies of ONE separator byte that
    differs from the dictionary bytes around the snippet — every snippet is found in the dictionary (no backward extension, no
    forward overrun), so the block's literals are the separators alone: more than 16 of them, all equal."""
    rng = np.random.default_rng(seed)
    # a SMALL dictionary: the encoders' tables keep the last position of a bucket only, and 4 KiB of noise rarely collide
    dct = bytes(rng.integers(0, 256, 4096, dtype=np.uint8))
    units = []
    for k in range(n):
        sep = int(rng.integers(0, 256))
        sniplen = int(rng.choice([12, 16, 24]))
        nsnip = int(rng.choice([20, 40, 300, 2500]))
        gap = 1 + (k & 1)
        out = bytearray()
        for p in rng.integers(16, len(dct) - 64, nsnip):
            p = int(p)
            while dct[p - 1] == sep or dct[p + sniplen] == sep or dct[p] == sep:
                p += 1
            out += bytes([sep]) * gap + dct[p:p + sniplen]
        units.append(bytes(out))
    return dct, units
