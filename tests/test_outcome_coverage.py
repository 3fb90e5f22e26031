"""Which encoder decisions the device parity tests actually reach.

The GPU tests prove HIP == oracle on their inputs; this (CPU) test states which format-visible outcomes those inputs produce, by
inspecting the oracle's frames (oracle/kco_zstd_dec.h inspection hook): block types, literal section types, the compression mode
of each sequence stream, sequence-count classes, repeat-offset use.  An outcome that no input reaches is a decision of the
reference the parity tests say nothing about; the ones known to be out of reach are listed with the reason.
"""
import collections
import ctypes as C
import re

import pytest

import corpora


def _inputs():
    units = [corpora.corpus(k, 1, 131072, first_unit=i).tobytes() for k in "THJM" for i in range(12)]
    units += [u for u in corpora.edge_units() if len(u)]
    for seed in (3, 11, 21):
        units += corpora.stress_units(seed=seed, n=40)
    t = corpora.corpus("T", 6, 131072, first_unit=40).tobytes()
    m = corpora.corpus("M", 6, 131072, first_unit=3).tobytes()
    units += [t[:300000], t[5:5 + 5 * 131072], m[:4 * 131072 + 77], m[100000:500000], (t[:200000] + m[:200000])]  # multi-block units: history, table reuse
    return units


@pytest.mark.parametrize("level", [1, 2, 3])
def test_parity_inputs_reach_these_outcomes(oracle, level):
    L = oracle.lib()
    L.kco_zstd_inspect.restype = C.c_int64
    L.kco_zstd_inspect.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    buf = C.create_string_buffer(8 << 20)
    e = oracle.ZstdOracle(level=level)
    cov = collections.Counter()
    for u in _inputs():
        fr = e.encode_all(u)
        assert L.kco_zstd_inspect(fr, len(fr), buf, len(buf)) >= 0
        txt = buf.value.decode()
        for m in re.finditer(r"block type=(\d) size=\d+ last=\d(?: litType=(\d) litBytes=\d+ nSeq=(\d+)(?: modes\(ll,of,ml\)=(\d),(\d),(\d))?)?", txt):
            cov["block:" + ("raw", "rle", "compressed")[int(m.group(1))]] += 1
            if m.group(2) is not None:
                cov["literals:" + ("raw", "rle", "compressed", "treeless")[int(m.group(2))]] += 1
                ns = int(m.group(3))
                cov["nseq:" + ("0" if ns == 0 else "<128" if ns < 128 else "<32512" if ns < 0x7F00 else ">=32512")] += 1
                if m.group(4) is not None:
                    for k, g in zip(("ll", "of", "ml"), (4, 5, 6)):
                        cov["%s:%s" % (k, ("predefined", "rle", "compressed", "repeat")[int(m.group(g))])] += 1
        mm = re.search(r"REP1=(\d+) REP2=(\d+) REP3=(\d+)", txt)
        if mm and int(mm.group(1)):
            cov["repcode:1"] += 1
    want = {"block:raw", "block:rle", "block:compressed", "literals:raw", "literals:compressed", "literals:treeless",
            "nseq:<128", "nseq:<32512", "repcode:1"}
    want |= {"%s:%s" % (k, m) for k in ("ll", "of", "ml") for m in ("predefined", "rle", "compressed", "repeat")}
    if level != 1:
        # Huffman table reuse needs the old table to be no worse than a new one (huff0/compress.go:153-171 with hSize == 0 at
        # that point): 147 consecutive 4 KiB text blocks at SpeedDefault never take it.  The SpeedFastest inputs do, and the
        # entropy stage is the same code at every level.
        want.discard("literals:treeless")
    missing = sorted(w for w in want if cov[w] == 0)
    assert not missing, (level, missing, dict(cov))
    # Known to be out of reach of these inputs (and why):
    #   literals:rle   - not by these inputs; reached by corpora.rle_literal_units (dictionary matches separated by one repeated
    #                    byte): test_rle_literal_sections_are_reachable below and its GPU twin
    #   nseq:>=32512   - needs < 4.04 bytes per sequence in a 128 KiB block; the shortest findable match is 5-6 bytes
    #   repcode:2, 3   - the encoders only ever test offset1 (and offset2 right after a match, which is coded as code 1 with ll == 0)
    assert cov["nseq:>=32512"] == 0


def _s2_tags(enc: bytes, cov, prefix):
    """Walk an S2 block (s2/decode_other.go:22-260) and count the tag shapes."""
    p, shift = 0, 0
    while True:  # uvarint length
        b = enc[p]; p += 1
        if not b & 0x80:
            break
    n = len(enc)
    while p < n:
        tag = enc[p]
        kind = tag & 3
        if kind == 0:
            x = tag >> 2
            if x < 60:
                cov[prefix + "literal:1-byte header"] += 1; ln = x + 1; p += 1
            elif x == 60:
                cov[prefix + "literal:2-byte header"] += 1; ln = enc[p + 1] + 1; p += 2
            elif x == 61:
                cov[prefix + "literal:3-byte header"] += 1; ln = (enc[p + 1] | enc[p + 2] << 8) + 1; p += 3
            elif x == 62:
                cov[prefix + "literal:4-byte header"] += 1; ln = (enc[p + 1] | enc[p + 2] << 8 | enc[p + 3] << 16) + 1; p += 4
            else:
                cov[prefix + "literal:5-byte header"] += 1; ln = int.from_bytes(enc[p + 1:p + 5], "little") + 1; p += 5
            p += ln
        elif kind == 1:
            off = ((tag & 0xe0) << 3) | enc[p + 1]
            length = (tag >> 2) & 7
            if off == 0:  # repeat
                if length < 5:
                    cov[prefix + "repeat:2 bytes"] += 1; p += 2
                elif length == 5:
                    cov[prefix + "repeat:3 bytes"] += 1; p += 3
                elif length == 6:
                    cov[prefix + "repeat:4 bytes"] += 1; p += 4
                else:
                    cov[prefix + "repeat:5 bytes"] += 1; p += 5
            else:
                cov[prefix + "copy1"] += 1; p += 2
        elif kind == 2:
            cov[prefix + "copy2"] += 1; p += 3
        else:
            cov[prefix + "copy4"] += 1; p += 5
    assert p == n


def test_s2_parity_inputs_reach_these_tags(oracle):
    """The same for the S2 levels: tag shapes in the oracle's blocks of the inputs the S2 GPU parity tests use (64 KiB corpus blocks,
    > 64 KiB blocks, edge units, stress mixes with periodic blocks)."""
    import numpy as np
    blocks = []
    for kind in "JTMH":
        b = corpora.corpus(kind, 16, 65536)
        blocks += [b[i * 65536:(i + 1) * 65536].tobytes() for i in range(16)]
        big = corpora.corpus(kind, 2, 1 << 20).tobytes()
        blocks += [big[:65537], big[:700000], big[1 << 20:]]
    blocks += [u for u in corpora.edge_units() if 0 < len(u) < 70000]
    blocks += corpora.stress_units(seed=11, n=60)
    rng = np.random.default_rng(100)
    for _ in range(40):
        per = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        blocks.append((per * 20000)[:int(rng.integers(1, 140000))])
    cov = collections.Counter()
    for blk in blocks:
        _s2_tags(oracle.s2_encode(blk), cov, "s2:")
        _s2_tags(oracle.s2_encode_better(blk), cov, "better:")
        _s2_tags(oracle.s2_encode_snappy(blk), cov, "snappy:")
        _s2_tags(oracle.s2_encode_snappy_better(blk), cov, "snappybetter:")
    for lvl in ("s2:", "better:"):
        for k in ("literal:1-byte header", "literal:2-byte header", "literal:3-byte header", "copy1", "copy2", "copy4",
                  "repeat:2 bytes", "repeat:3 bytes", "repeat:4 bytes"):
            assert cov[lvl + k] > 0, (lvl + k, dict(cov))
    for lvl in ("snappy:", "snappybetter:"):
        for k in ("literal:1-byte header", "literal:2-byte header", "literal:3-byte header", "copy1", "copy2", "copy4"):
            assert cov[lvl + k] > 0, (lvl + k, dict(cov))
        assert not any(cov[lvl + r] for r in ("repeat:2 bytes", "repeat:3 bytes", "repeat:4 bytes", "repeat:5 bytes"))  # Snappy has no repeats
    # Not reached: 4/5-byte literal headers need a literal run of 64 KiB+ / 16 MiB+ inside a compressible block (a stored block is one
    # 3- or 4-byte-header literal by itself); 5-byte repeats need a repeat longer than 65 KiB + 260.


def _first_block_literal_type(fr):
    """Literal section type of a frame's first block, read off the bytes (zstd/frameenc.go:25-92, blockenc.go:109-238)."""
    fhd = fr[4]
    p = 5
    single = (fhd >> 5) & 1
    if not single:
        p += 1
    p += [0, 1, 2, 4][fhd & 3]
    p += [1 if single else 0, 2, 4, 8][fhd >> 6]
    bh = fr[p] | fr[p + 1] << 8 | fr[p + 2] << 16
    if (bh >> 1) & 3 != 2:
        return None
    return ("raw", "rle", "compressed", "treeless")[fr[p + 3] & 3]


@pytest.mark.parametrize("level", [1, 2, 3])
def test_rle_literal_sections_are_reachable(oracle, level):
    """literals:rle — the one literal-section type the corpus inputs never produce — IS reachable, with a dictionary:
    corpora.rle_literal_units builds units whose literals are > 16 copies of one byte between dictionary matches.  The oracle emits
    RLE literal sections for them at every level (the GPU twin of this test byte-compares the device frames:
    tests/test_gpu_zstd.py::test_rle_literal_sections_bit_exact) and its own decoder restores the input."""
    dct, units = corpora.rle_literal_units()
    e = oracle.ZstdOracle(level=level, dict_id=7, dict_content=dct)
    kinds = collections.Counter()
    for u in units:
        fr = e.encode_all(u)
        kinds[_first_block_literal_type(fr)] += 1
        assert oracle.zstd_decode(fr, len(u) + 16, dict_content=dct) == u
    assert kinds["rle"] >= len(units) // 2, dict(kinds)
