#!/usr/bin/env python3
"""Generate tests/golden/kats.json.

Two kinds of vectors:
  * KATs transcribed from the reference's own tests (file:line cited per table);
  * data-derived values computed HERE from the reference's fixtures under
    /root/reference/testdata with an independent pure-Python XXH64 (itself first checked
    against the reference's XXH64 KATs), so the GPU box never needs /root/reference.
Run in the build container:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import struct

REF = "/root/reference"
M = (1 << 64) - 1
P1, P2, P3, P4, P5 = 11400714785074694791, 14029467366897019727, 1609587929392839161, 9650029242287828579, 2870177450012600261


def rol(x, r):
    return ((x << r) | (x >> (64 - r))) & M


def rnd(acc, inp):
    return (rol((acc + inp * P2) & M, 31) * P1) & M


def merge(acc, val):
    return (((acc ^ rnd(0, val)) * P1) + P4) & M


def xxh64(b):
    n = len(b)
    i = 0
    if n >= 32:
        v1, v2, v3, v4 = (P1 + P2) & M, P2, 0, (-P1) & M
        while i + 32 <= n:
            a, bb, c, d = struct.unpack_from("<QQQQ", b, i)
            v1, v2, v3, v4 = rnd(v1, a), rnd(v2, bb), rnd(v3, c), rnd(v4, d)
            i += 32
        h = (rol(v1, 1) + rol(v2, 7) + rol(v3, 12) + rol(v4, 18)) & M
        for v in (v1, v2, v3, v4):
            h = merge(h, v)
    else:
        h = P5
    h = (h + n) & M
    while i + 8 <= n:
        h ^= rnd(0, struct.unpack_from("<Q", b, i)[0])
        h = (rol(h, 27) * P1 + P4) & M
        i += 8
    if i + 4 <= n:
        h ^= (struct.unpack_from("<I", b, i)[0] * P1) & M
        h = (rol(h, 23) * P2 + P3) & M
        i += 4
    while i < n:
        h ^= (b[i] * P5) & M
        h = (rol(h, 11) * P1) & M
        i += 1
    h ^= h >> 33
    h = (h * P2) & M
    h ^= h >> 29
    h = (h * P3) & M
    h ^= h >> 32
    return h


# zstd/internal/xxhash/xxhash_test.go:16-27
XXH_KATS = [["", 0xef46db3751d8e999], ["a", 0xd24ec4f1a98c6e5b], ["as", 0x1c330fb2d66be179], ["asd", 0x631c37ce72a97393],
            ["asdf", 0x415872f599cea71e],
            ["Call me Ishmael. Some years ago--never mind how long precisely-", 0x02a2e85470d6fd96]]
# s2/s2_test.go:827-862 TestEmitLiteral: (length, header hex)
EMIT_LITERAL = [[1, "00"], [2, "04"], [59, "e8"], [60, "ec"], [61, "f03c"], [62, "f03d"], [254, "f0fd"], [255, "f0fe"], [256, "f0ff"],
                [257, "f40001"], [65534, "f4fdff"], [65535, "f4feff"], [65536, "f4ffff"]]
# s2/s2_test.go:864-942 TestEmitCopy: (offset, length, bytes hex)
EMIT_COPY = [
    [8, 4, "0108"], [8, 11, "1d08"], [8, 12, "2e0800"], [8, 13, "320800"], [8, 59, "ea0800"], [8, 60, "ee0800"], [8, 61, "f20800"],
    [8, 62, "f60800"], [8, 63, "fa0800"], [8, 64, "fe0800"], [8, 65, "1108150031"], [8, 66, "1108150032"], [8, 67, "1108150033"],
    [8, 68, "1108150034"], [8, 69, "1108150035"], [8, 80, "1108150040"], [8, 800, "110819001402"], [8, 800000, "11081d00f4340b"],
    [256, 4, "2100"], [256, 11, "3d00"], [256, 12, "2e0001"], [256, 13, "320001"], [256, 59, "ea0001"], [256, 60, "ee0001"],
    [256, 61, "f20001"], [256, 62, "f60001"], [256, 63, "fa0001"], [256, 64, "fe0001"], [256, 65, "3100150031"], [256, 66, "3100150032"],
    [256, 67, "3100150033"], [256, 68, "3100150034"], [256, 69, "3100150035"], [256, 80, "3100150040"], [256, 800, "310019001402"],
    [256, 80000, "31001d00743800"],
    [2048, 4, "0e0008"], [2048, 11, "2a0008"], [2048, 12, "2e0008"], [2048, 13, "320008"], [2048, 59, "ea0008"], [2048, 60, "ee0008"],
    [2048, 61, "f20008"], [2048, 62, "f60008"], [2048, 63, "fa0008"], [2048, 64, "fe0008"], [2048, 65, "ee00080500"], [2048, 66, "ee00080900"],
    [2048, 67, "ee00080d00"], [2048, 68, "ee00081100"], [2048, 69, "ee0008150001"], [2048, 80, "ee000815000c"], [2048, 800, "ee00081900e001"],
    [2048, 80000, "ee00081d00403800"],
    [204800, 4, "0f00200300"], [204800, 65, "ff002003000300200300"], [204800, 69, "ff002003000500"], [204800, 800, "ff002003001900dc01"],
    [204800, 80000, "ff002003001d003c3800"]]
# s2/s2_test.go:37-76 TestMaxEncodedLen (64-bit ints): (in, out)
MAXU32 = 0xFFFFFFFF
MAX_ENCODED_LEN = [[0, 1], [1 << 24, (1 << 24) + 4 + 5], [MAXU32 - 5 - 5, MAXU32], [MAXU32 - 5 - 5, MAXU32]] + \
    [[MAXU32 - k, -1] for k in range(9, -1, -1)] + [[-1, -1], [-2, -1]]


def main():
    out = {"xxh64": [[s, "%016x" % h] for s, h in XXH_KATS], "s2_emit_literal": EMIT_LITERAL, "s2_emit_copy": EMIT_COPY,
           "s2_max_encoded_len": MAX_ENCODED_LEN, "files": {}}
    for s, h in XXH_KATS:
        assert xxh64(s.encode()) == h, s
    # data-derived: frame boundary bytes of EncodeAll(file) at SpeedFastest with defaults, which are fully
    # determined by frameenc.go:25-92 + enc_base.go:34-38, for the C1 fixture and friends.
    for name in ("e.txt", "pi.txt", "Mark.Twain-Tom.Sawyer.txt", "html.txt", "gettysburg.txt", "sharnd.out"):
        p = os.path.join(REF, "testdata", name)
        b = open(p, "rb").read()
        h = xxh64(b)
        n = len(b)
        single = n <= (4 << 20) and n > 1024
        fcs = (1 if n >= 256 else 0) + (1 if n >= 65536 + 256 else 0)
        fhd = (1 << 2) | ((1 << 5) if single else 0) | (fcs << 6)
        hdr = bytes([0x28, 0xb5, 0x2f, 0xfd, fhd])
        if not single:
            wlog = max(1 << n.bit_length(), 1024)
            hdr += bytes([((wlog - 1).bit_length() - 10) << 3])
        if fcs == 0:
            hdr += bytes([n]) if single else b""
        elif fcs == 1:
            hdr += struct.pack("<H", n - 256)
        else:
            hdr += struct.pack("<I", n)
        out["files"][name] = {"len": n, "sha256": hashlib.sha256(b).hexdigest(), "xxh64": "%016x" % h,
                              "frame_prefix": hdr.hex(), "frame_suffix": struct.pack("<I", h & 0xFFFFFFFF).hex()}
    z = open(os.path.join(REF, "zstd/testdata/z000028.zst"), "rb").read()
    zin = open(os.path.join(REF, "zstd/testdata/z000028"), "rb").read()
    out["files"]["z000028"] = {"len": len(zin), "xxh64": "%016x" % xxh64(zin), "zst_trailer": z[-4:].hex()}
    # Full-format dictionary fixtures of the reference's dictionary tests (zstd/dict_test.go:150 TestEncoder_SmallDict reads
    # testdata/dict-tests-small.zip): d0.dict + the d0/*.zst frames (compressed WITH that dictionary by the C zstd; the
    # tests decode them with libzstd to get inputs typical for the dictionary).  Frames <= 32 KiB only, to stay small.
    import zipfile
    zf = zipfile.ZipFile(os.path.join(REF, "zstd/testdata/dict-tests-small.zip"))
    ddir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dict")
    os.makedirs(ddir, exist_ok=True)
    blob = zf.read("d0.dict")
    open(os.path.join(ddir, "d0.dict"), "wb").write(blob)
    names = sorted(n for n in zf.namelist() if n.startswith("d0/") and n.endswith(".zst") and zf.getinfo(n).file_size <= 32768)
    for n in names:
        open(os.path.join(ddir, os.path.basename(n)), "wb").write(zf.read(n))
    out["dict"] = {"d0.dict": {"len": len(blob), "sha256": hashlib.sha256(blob).hexdigest(), "id": struct.unpack_from("<I", blob, 4)[0]},
                   "frames": [os.path.basename(n) for n in names]}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote kats.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
