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
    #   literals:rle   - every literal of a block equal AND >16 of them: backward extension and the next probe swallow such runs
    #                    (tried: separators between dictionary tokens, zero runs between tokens - the literals end up raw)
    #   nseq:>=32512   - needs < 4.04 bytes per sequence in a 128 KiB block; the shortest findable match is 5-6 bytes
    #   repcode:2, 3   - the encoders only ever test offset1 (and offset2 right after a match, which is coded as code 1 with ll == 0)
    assert cov["nseq:>=32512"] == 0
