"""Gate on the REAL reference's bytes.

tests/golden/reference_go_sha256.txt (committed) is written in the build container by tests/golden/make_reference_go_golden.py
from oracle/_ref/libzstdref.so — the reference's own Go source for EncodeAll / s2.Encode*, translated statement by statement
into C++ at build time (oracle/ref_go) — so the GPU box, which has no /root/reference, still holds the oracle and the HIP path to
the reference's bytes.  The second file below is the same thing from a real Go toolchain, when somebody has one:

tests/golden/reference_sha256.txt is written by the Go tests of shim/go (zstdgpu.TestWriteGolden, s2gpu.TestWriteGolden:
`KC_WRITE_GOLDEN=1 go test -tags noasm ./...` on a host with Go): one line `<name> <sha256>` per seeded corpus and level,
the hash of the concatenated output of the reference encoder itself.  With the file present these tests turn "parity
unpinned" into a measurement: the C++ oracle (CPU test) and the HIP path (gpu test) must hash to the same value.  The build
image has no Go toolchain, so the file is absent there and the tests skip, saying so."""
import hashlib
import os

import numpy as np
import pytest

import corpora

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "reference_sha256.txt")
GOLD_GO = os.path.join(HERE, "golden", "reference_go_sha256.txt")
S2_LEVELS = {"s2": 0, "s2better": 1, "s2snappy": 2, "s2snappybetter": 3, "s2best": 4, "s2snappybest": 5}
DICT_SEED = 0x5EED0005


def _lines():
    out = {}
    for p in (GOLD_GO, GOLD):
        if not os.path.exists(p):
            continue
        for l in open(p):
            f = l.split()
            if len(f) == 2:
                assert out.get(f[0], f[1]) == f[1], "the two golden files disagree on " + f[0]
                out[f[0]] = f[1]
    assert out, "tests/golden/reference_go_sha256.txt is a committed fixture"
    return out


def _parse(name):
    p = name.split(".")
    if p[0] == "s2stream":  # s2stream.<level name>.<kind>.<n>x<unit>.flush.index.pad4096: the framed stream of s2.Writer
        n, usz = p[3].split("x")
        return dict(codec="s2stream", level={"s2": 0, "s2better": 1, "s2best": 2, "s2snappy": 0}[p[1]], snappy=p[1] == "s2snappy", kind=p[2], n=int(n), unit=int(usz))
    if p[0] == "zstdstream":  # zstdstream.L<level>.<kind>.<n>x<len>.flush: streams with Flush after 70000, 70010 and 200000+i bytes
        n, usz = p[3].split("x")
        return dict(codec="zstdstream", level=int(p[1][1:]), kind=p[2], n=int(n), unit=int(usz))
    if p[0] == "zstd":
        n, usz = p[3].split("x")
        return dict(codec="zstd", level=int(p[1][1:]), kind=p[2], n=int(n), unit=int(usz), rawdict=(len(p) > 4 and p[4] == "rawdict64k"))
    n, usz = p[2].split("x")
    return dict(codec="s2", kind=p[1], n=int(n), unit=int(usz), level=S2_LEVELS[p[0]])


def _flush_points(i):
    return [70000, 70010, 200000 + i]


def _dict():
    from compress_amd import _lib
    return _lib.corpus_fill("T", DICT_SEED, 0, 1, 64 << 10).tobytes()


def test_oracle_matches_reference_hashes(oracle):
    lines = _lines()
    for name, want in sorted(lines.items()):
        c = _parse(name)
        if c["codec"] == "s2stream":
            import test_ref_s2_stream as ts
            data = corpora.corpus(c["kind"], c["n"], c["unit"]).tobytes()
            out = ts.expected_stream(oracle, data, (100000, 100001), bs=c["unit"], level=c["level"], snappy=c["snappy"], add_index=True, padding=4096)
            assert hashlib.sha256(out).hexdigest() == want, "oracle differs from the reference on " + name
            continue
        if c["codec"] == "zstdstream":
            data = corpora.corpus(c["kind"], 24, 131072).tobytes()
            e = oracle.ZstdOracle(level=c["level"])
            h = hashlib.sha256()
            for i in range(c["n"]):
                h.update(e.encode_stream(data[i * c["unit"]:(i + 1) * c["unit"]], _flush_points(i)))
            assert h.hexdigest() == want, "oracle differs from the reference on " + name
            continue
        buf = corpora.corpus(c["kind"], c["n"], c["unit"])
        off = np.arange(c["n"] + 1, dtype=np.uint64) * c["unit"]
        if c["codec"] == "zstd":
            kw = dict(level=c["level"])
            if c["rawdict"]:
                kw.update(dict_id=1, dict_content=_dict())
            out, _ = oracle.zstd_encode_units(buf, off, threads=8, **kw)
        elif c["level"] >= 4:
            b = buf.tobytes()
            fn = oracle.s2_encode_best if c["level"] == 4 else oracle.s2_encode_snappy_best
            out = np.frombuffer(b"".join(fn(b[i * c["unit"]:(i + 1) * c["unit"]]) for i in range(c["n"])), dtype=np.uint8)
        else:
            out, _ = oracle.s2_encode_blocks(buf, off, threads=8, better=c["level"] in (1, 3), snappy=c["level"] in (2, 3))
        assert hashlib.sha256(np.asarray(out).tobytes()).hexdigest() == want, "oracle differs from the reference on " + name


@pytest.mark.gpu
def test_gpu_matches_reference_hashes(kclib):
    from compress_amd import zstd, s2
    lines = _lines()
    for name, want in sorted(lines.items()):
        c = _parse(name)
        if c["codec"] == "s2stream":
            import io
            import test_ref_s2_stream as ts
            data = corpora.corpus(c["kind"], c["n"], c["unit"]).tobytes()
            opts = [s2.WriterBlockSize(c["unit"]), s2.WriterAddIndex(), s2.WriterPadding(4096), s2.WriterPaddingSrc(ts._Zeros())]
            opts += {0: [], 1: [s2.WriterBetterCompression()], 2: [s2.WriterBestCompression()]}[c["level"]] + ([s2.WriterSnappyCompat()] if c["snappy"] else [])
            sink = io.BytesIO()
            w = s2.NewWriter(sink, *opts)
            w.Write(data[:100000]); w.Flush(); w.Write(data[100000:100001]); w.Flush(); w.Write(data[100001:]); w.Close()
            assert hashlib.sha256(sink.getvalue()).hexdigest() == want, "HIP path differs from the reference on " + name
            continue
        if c["codec"] == "zstdstream":
            data = corpora.corpus(c["kind"], 24, 131072)[:c["n"] * c["unit"]]
            off = np.arange(c["n"] + 1, dtype=np.uint64) * c["unit"]
            enc = zstd.NewWriter(None, zstd.WithEncoderLevel(c["level"]))
            out, _ = enc.EncodeStreams(data, off, flush_at=[_flush_points(i) for i in range(c["n"])])
            enc.Close()
            assert hashlib.sha256(out.tobytes()).hexdigest() == want, "HIP path differs from the reference on " + name
            continue
        buf = corpora.corpus(c["kind"], c["n"], c["unit"])
        off = np.arange(c["n"] + 1, dtype=np.uint64) * c["unit"]
        if c["codec"] == "zstd":
            opts = [zstd.WithEncoderLevel(c["level"])]
            if c["rawdict"]:
                opts.append(zstd.WithEncoderDictRaw(1, _dict()))
            enc = zstd.NewWriter(None, *opts)
            out, _ = enc.EncodeUnits(buf, off)
        else:
            enc = s2.BlockEncoder(level=c["level"])
            out, _ = enc.EncodeBlocks(buf, off)
        enc.Close()
        assert hashlib.sha256(out.tobytes()).hexdigest() == want, "HIP path differs from the reference on " + name
