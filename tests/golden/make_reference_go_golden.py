#!/usr/bin/env python3
"""Write tests/golden/reference_go_sha256.txt: hashes of the REFERENCE's own output, for the GPU box (where /root/reference is absent).

The reference encoders are run through oracle/_ref/libzstdref.so — the reference's pure-Go source translated statement by
statement into C++ at build time (oracle/ref_go) — on the seeded corpora of tests/corpora.py.  One line `<name> <sha256 of the
concatenated frames / blocks>` per corpus and level, in the naming of shim/go's TestWriteGolden (which writes
reference_sha256.txt with a real Go toolchain; the two files must agree where both have a line):
    zstd.L<level>.<kind>.<n>x<unit>[.rawdict64k]       zstd.Encoder.EncodeAll per unit (encoder.go:722), level 1..4
    zstdstream.L<level>.<kind>.10x300001.flush         Write / Flush / Close streams (WithEncoderConcurrency(1)), level 1..4
    s2|s2better|s2snappy|s2snappybetter|s2best|s2snappybest.<kind>.<n>x<unit>   s2.Encode* per block, portable Go (`noasm`) form
    s2stream.<s2|s2better|s2best|s2snappy>.<kind>.16x65536.flush.index.pad4096   s2.NewWriter(w, WriterBlockSize(64K), level, WriterAddIndex(),
                                                       WriterPadding(4096) with a zero padding source): Write + Flush after 100000 and 100001 bytes, Close
Run in the build container:  python tests/golden/make_reference_go_golden.py      (about two minutes)"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import corpora  # noqa: E402
import oracle_goref  # noqa: E402
from compress_amd import _lib  # noqa: E402

DICT_SEED = 0x5EED0005
S2 = ["s2", "s2better", "s2snappy", "s2snappybetter", "s2best", "s2snappybest"]


def main():
    lines = {}
    dct = _lib.corpus_fill("T", DICT_SEED, 0, 1, 64 << 10).tobytes()
    for level in (1, 2, 3, 4):
        n = 96 if level < 4 else 24
        for kind in "THJM":
            buf = corpora.corpus(kind, n, 131072).tobytes()
            for with_dict in (False, True):
                kw = dict(level=level)
                if with_dict:
                    kw.update(dict_id=1, dict_content=dct)
                h = hashlib.sha256()
                for i in range(n):
                    h.update(oracle_goref.zstd_encode_all(buf[i * 131072:(i + 1) * 131072], **kw))
                lines["zstd.L%d.%s.%dx131072%s" % (level, kind, n, ".rawdict64k" if with_dict else "")] = h.hexdigest()
    # streams with Flush points, as TestWriteGolden writes them: 10 streams of 300001 bytes, Flush after 70000, 70010 and 200000+i bytes
    for level in (1, 2, 3, 4):
        for kind in "TM":
            data = corpora.corpus(kind, 24, 131072).tobytes()
            h = hashlib.sha256()
            for i in range(10):
                u = data[i * 300001:(i + 1) * 300001]
                h.update(oracle_goref.zstd_encode_stream(u, (70000, 70010, 200000 + i), level=level, concurrent=1))
            lines["zstdstream.L%d.%s.10x300001.flush" % (level, kind)] = h.hexdigest()
    for lv, name in enumerate(S2):
        n = 128 if lv < 4 else 32
        for kind in "JTMH":
            buf = corpora.corpus(kind, n, 65536).tobytes()
            h = hashlib.sha256()
            for i in range(n):
                h.update(oracle_goref.s2_encode(buf[i * 65536:(i + 1) * 65536], lv))
            lines["%s.%s.%dx65536" % (name, kind, n)] = h.hexdigest()
    for name, (lv, snappy) in (("s2", (0, False)), ("s2better", (1, False)), ("s2best", (2, False)), ("s2snappy", (0, True))):
        for kind in "JT":
            data = corpora.corpus(kind, 16, 65536).tobytes()
            out = oracle_goref.s2_stream(data, (100000, 100001), level=lv, snappy=snappy, block_size=65536, add_index=True, padding=4096)
            lines["s2stream.%s.%s.16x65536.flush.index.pad4096" % (name, kind)] = hashlib.sha256(out).hexdigest()
    with open(os.path.join(HERE, "reference_go_sha256.txt"), "w") as f:
        for k in sorted(lines):
            f.write("%s %s\n" % (k, lines[k]))
    print("wrote %d lines" % len(lines))


if __name__ == "__main__":
    main()
