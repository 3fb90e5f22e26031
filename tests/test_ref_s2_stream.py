"""The reference's own s2.Writer — translated from its Go source (oracle/ref_go; the synchronous form WriterConcurrency(1): Write /
writeSync / Flush / Close / closeIndex of s2/writer.go, Index.add / appendTo of s2/index.go, skippableFrame) — against the oracle's
restatement of the stream format: chunk framing with the masked CRC32C, stored chunks, the stream identifiers, chunk boundaries under
Write / Flush, the index chunk and the padding chunk.  GPU: the device-backed façade compress_amd.s2.Writer writes the same stream."""
import io

import numpy as np
import pytest

import corpora
import oracle_goref

pytestmark = pytest.mark.skipif(not oracle_goref.available(), reason="oracle/_ref/libzstdref.so neither present nor buildable (no /root/reference)")

MAGIC_SNAPPY = b"\xff\x06\x00\x00sNaPpY"
OLEVEL = {(0, False): 0, (1, False): 1, (2, False): 4, (0, True): 2, (1, True): 3, (2, True): 5}  # (writer level, snappy) -> oracle level


def _blocks(n, cuts, bs):
    """Chunk boundaries of Write(src[a:b]) + Flush() per cut, then Write(rest) + Close() (writer.go:182-218: a Write larger than the
    free buffer goes out at once, short tail included; a smaller one waits in ibuf for the Flush): a grid restarting at every cut."""
    edges, pos = [0], 0
    for c in sorted(x for x in cuts if x <= n) + [n]:
        if c > pos:
            edges += list(range(pos + bs, c, bs)) + [c]
            pos = c
    return np.array(edges, dtype=np.uint64)


def _skippable(total):
    """skippableFrame(dst, total, zero source), writer.go:878-899."""
    if total == 0:
        return b""
    f = total - 4
    return bytes([0xfe, f & 0xFF, (f >> 8) & 0xFF, (f >> 16) & 0xFF]) + b"\0" * f


def _calc_skippable(written, mult):
    """calcSkippableFrame, writer.go:858-874."""
    left = written % mult
    if left == 0:
        return 0
    add = mult - left
    while add < 4:
        add += mult
    return add


def expected_stream(oracle, src, cuts=(), bs=1 << 20, level=0, snappy=False, add_index=False, padding=0):
    blk = _blocks(len(src), cuts, bs)
    nb = len(blk) - 1
    if nb == 0:
        stream, adds = b"", []
    else:
        s, oo = oracle.s2_encode_stream(np.frombuffer(src, dtype=np.uint8), blk, level=OLEVEL[(level, snappy)])
        stream = s.tobytes()
        if snappy:
            stream = MAGIC_SNAPPY + stream[10:]
        adds = [(int(oo[k]), int(blk[k])) for k in range(nb)]
    written = len(stream)
    index = b""
    if add_index:
        index = oracle.s2_index(bs, adds, len(src), written if padding <= 1 else -1)
        written += len(index)
    pad = _skippable(_calc_skippable(written, padding)) if padding > 1 else b""
    return stream + pad + index


def _inputs():
    j = corpora.corpus("J", 5, 65536, first_unit=3).tobytes()
    t = corpora.corpus("T", 3, 131072, first_unit=9).tobytes()
    h = corpora.corpus("H", 1, 131072, first_unit=2).tobytes()
    return j, t, h


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("snappy", [False, True])
def test_oracle_stream_framing_equals_the_translated_writer(oracle, level, snappy):
    """Stream identifier, chunk headers, masked CRC32C, compressed and stored chunks (the high-entropy input), chunk boundaries for
    block sizes 4 KiB / 64 KiB and — S2 only — the 1 MiB default, inputs that end on and off the block grid, the empty stream."""
    j, t, h = _inputs()
    bad = []
    for bs in (4096, 65536) + (() if snappy else (0,)):
        for data in (j, t[:200000], h[:70000] + j[:30000], j[:65536], j[:65537], j[:4095], j[:100], j[:1], b""):
            if level == 2 and len(data) > 140000:
                data = data[:140000]
            got = oracle_goref.s2_stream(data, level=level, snappy=snappy, block_size=bs)
            want = expected_stream(oracle, data, bs=bs or (1 << 20), level=level, snappy=snappy)
            if got != want:
                bad.append((bs, len(data), len(got), len(want)))
    assert not bad, bad


@pytest.mark.parametrize("level", [0, 1])
def test_oracle_chunk_boundaries_under_write_and_flush(oracle, level):
    """Write / Flush sequences: a Flush ends the chunk being filled, the grid restarts behind it; writes larger and smaller than the
    block buffer; WriterFlushOnWrite."""
    j, t, _ = _inputs()
    bad = []
    for bs, data, cuts in ((4096, j[:50000], (1000, 9000, 9001, 30000)), (65536, t[:300000], (65536, 70000, 200001)), (65536, j[:100000], (100000,)),
                           (4096, j[:20000], (4096, 8192)), (65536, j[:1000], (10, 20, 30))):
        got = oracle_goref.s2_stream(data, cuts, level=level, block_size=bs)
        want = expected_stream(oracle, data, cuts, bs=bs, level=level)
        if got != want:
            bad.append((bs, len(data), cuts))
        if oracle_goref.s2_stream(data, cuts, level=level, block_size=bs, flush_on_write=True) != want:
            bad.append(("flush_on_write", bs, len(data), cuts))
    assert not bad, bad


@pytest.mark.parametrize("padding", [0, 1024, 4096, 100000])
def test_oracle_index_and_padding_equal_the_translated_writer(oracle, padding):
    """WriterAddIndex (Index.add's 1 MiB spacing rule, appendTo's prediction-coded offsets, the total sizes — the compressed total is
    unknown (-1) when padding follows) and WriterPadding (the skippable chunk goes out BEFORE the index and is sized as if the index
    had been written, writer.go:808-847); zero bytes as padding source."""
    j, t, _ = _inputs()
    big = (j + t) * 6  # 4.3 MB: several index entries at the 1 MiB spacing
    bad = []
    for bs, data in ((65536, big), (1 << 20, big), (4096, j[:300000]), (65536, j[:65536]), (65536, j[:10])):
        for add_index in (True, False):
            if not add_index and padding == 0:
                continue
            got = oracle_goref.s2_stream(data, block_size=bs, add_index=add_index, padding=padding)
            want = expected_stream(oracle, data, bs=bs, add_index=add_index, padding=padding)
            if padding > 1:
                assert len(got) % padding == 0
            if got != want:
                bad.append((bs, len(data), add_index, len(got), len(want)))
    assert not bad, bad


def test_translated_writer_streams_decode(oracle):
    """... and the in-repo stream decoder returns the input of every translated stream (index and padding chunks are skippable; it
    checks every chunk's CRC).  WriterUncompressed (level 3): stored chunks only."""
    j, t, h = _inputs()
    for kw in (dict(), dict(level=1), dict(level=2), dict(snappy=True, block_size=65536), dict(add_index=True, padding=4096), dict(level=3, block_size=65536)):
        data = (j + h + t)[:400000]
        s = oracle_goref.s2_stream(data, (1234, 300000), **kw)
        if kw.get("snappy"):  # (the in-repo decoder wants the S2 stream identifier; the reference's reads both)
            assert s[:10] == MAGIC_SNAPPY
            s = b"\xff\x06\x00\x00S2sTwO" + s[10:]
        assert oracle.s2_decode_stream(s, len(data) + 64) == data, kw
        if kw.get("level") == 3:
            pos, sizes = 10, []
            while pos < len(s):
                assert s[pos] == 0x01, "WriterUncompressed wrote a chunk of type %#x" % s[pos]
                n = int.from_bytes(s[pos + 1:pos + 4], "little")
                sizes.append(n - 4)
                pos += 4 + n
            assert sizes == [1234, 65536, 65536, 65536, 65536, 300000 - 1234 - 4 * 65536, 65536, 100000 - 65536]


def test_reference_reader_reads_oracle_and_translated_streams_and_refuses_damage(oracle):
    """The reference's own s2.Reader (sequential Read, translated; its block decoder in the portable Go form) returns the input of every
    stream the translated Writer and the oracle write — S2 and Snappy identifiers, index, padding, stored chunks — and gives the in-repo
    stream decoder's verdict on 300 damaged streams (CRC, chunk types, lengths, block bodies)."""
    j, t, h = _inputs()
    data = (j + h + t)[:400000]
    for kw in (dict(), dict(level=1), dict(level=2), dict(snappy=True, block_size=65536), dict(add_index=True, padding=4096), dict(level=3, block_size=65536),
               dict(block_size=4096, add_index=True)):
        s = oracle_goref.s2_stream(data, (1234, 300000), **kw)
        assert oracle_goref.s2_read_stream(s, len(data)) == data, kw
    for olevel in (0, 1, 4):
        blk = _blocks(len(data), (), 65536)
        s, _ = oracle.s2_encode_stream(np.frombuffer(data, dtype=np.uint8), blk, level=olevel)
        assert oracle_goref.s2_read_stream(s.tobytes(), len(data)) == data, olevel
    assert oracle_goref.s2_read_stream(b"", 16) == b""
    good = oracle_goref.s2_stream(j[:150000], block_size=16384, add_index=True)
    rng = np.random.default_rng(5)
    agree = rejected = 0
    for _ in range(300):
        g = bytearray(good)
        for _k in range(int(rng.integers(1, 3))):
            g[int(rng.integers(0, len(g)))] ^= 1 << int(rng.integers(0, 8))
        try:
            a = oracle_goref.s2_read_stream(bytes(g), 200000)
        except ValueError:
            a = None
        try:
            b = oracle.s2_decode_stream(bytes(g), 200000)
        except RuntimeError:
            b = None
        # (a flip inside the index / a skippable chunk's payload leaves a valid stream: both read the input)
        assert (a is None) == (b is None) and (a is None or a == b), (a is None, b is None)
        agree += 1
        rejected += a is None
    assert rejected > 200 and agree == 300


class _Zeros:
    def read(self, n):
        return b"\0" * n


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1, 2])
def test_device_writer_equals_the_translated_writer(kclib, level):
    """GPU: compress_amd.s2.Writer (chunk bodies from the device) writes, byte for byte, what the reference's own Writer writes for
    the same options and the same Write / Flush / Close sequence: block sizes, S2 and Snappy-compatible streams, WriterAddIndex,
    WriterPadding, WriterFlushOnWrite, WriterUncompressed."""
    import torch
    from compress_amd import s2
    assert torch.cuda.is_available()
    j, t, h = _inputs()
    big = (j + t) * 4
    lvl = {0: [], 1: [s2.WriterBetterCompression()], 2: [s2.WriterBestCompression()]}[level]
    cases = [(dict(block_size=65536), j, ()), (dict(block_size=4096), j[:50000], (1000, 9000, 9001, 30000)), (dict(), big if level < 2 else big[:1500000], (2000000,)),
             (dict(block_size=65536, add_index=True), big if level < 2 else big[:700000], ()), (dict(block_size=65536, add_index=True, padding=4096), j + h, (70000,)),
             (dict(block_size=65536, padding=1024), t[:200000], ()), (dict(block_size=65536, snappy=True), j + h[:70000], (65536,)),
             (dict(block_size=65536, flush_on_write=True), j[:100000], (10, 70000)), (dict(block_size=65536), b"", ())]
    for kw, data, cuts in cases:
        opts = list(lvl)
        if "block_size" in kw:
            opts.append(s2.WriterBlockSize(kw["block_size"]))
        if kw.get("add_index"):
            opts.append(s2.WriterAddIndex())
        if kw.get("padding"):
            opts += [s2.WriterPadding(kw["padding"]), s2.WriterPaddingSrc(_Zeros())]
        if kw.get("snappy"):
            opts.append(s2.WriterSnappyCompat())
        if kw.get("flush_on_write"):
            opts.append(s2.WriterFlushOnWrite())
        sink = io.BytesIO()
        w = s2.NewWriter(sink, *opts)
        pos = 0
        for c in cuts:
            if c <= len(data):
                if c > pos:
                    w.Write(data[pos:c])
                    pos = c
                w.Flush()
        if len(data) > pos:
            w.Write(data[pos:])
        w.Close()
        want = oracle_goref.s2_stream(data, cuts, level=level, **kw)
        got = sink.getvalue()
        assert got == want, "level %d %r len %d cuts %r: %d bytes vs the reference's %d" % (level, kw, len(data), cuts, len(got), len(want))
        assert oracle_goref.s2_read_stream(got, len(data)) == data  # ... and the reference's own Reader returns the input
    sink = io.BytesIO()
    w = s2.NewWriter(sink, s2.WriterUncompressed(), s2.WriterBlockSize(65536))
    w.Write(j[:200000])
    w.Close()
    assert sink.getvalue() == oracle_goref.s2_stream(j[:200000], level=3, block_size=65536)


def test_facade_host_logic_on_an_empty_stream_equals_the_translated_writer(monkeypatch):
    """Host logic of compress_amd.s2.Writer that needs no device: an empty stream — nothing at all without options; with
    WriterAddIndex the index chunk (and with WriterPadding its padding in front) WITHOUT a stream identifier, exactly as the
    reference's Writer leaves it (its own Reader refuses that stream: a quirk kept as it is)."""
    from compress_amd import s2

    class _NoDevice:
        def __init__(self, *a, **k):
            pass

        def Close(self):
            pass
    monkeypatch.setattr(s2, "BlockEncoder", _NoDevice)
    for kw, opts in ((dict(), []), (dict(add_index=True), [s2.WriterAddIndex()]), (dict(padding=4096), [s2.WriterPadding(4096), s2.WriterPaddingSrc(_Zeros())]),
                     (dict(add_index=True, padding=4096), [s2.WriterAddIndex(), s2.WriterPadding(4096), s2.WriterPaddingSrc(_Zeros())])):
        sink = io.BytesIO()
        w = s2.NewWriter(sink, *opts)
        w.Close()
        assert sink.getvalue() == oracle_goref.s2_stream(b"", **kw), kw
    with pytest.raises(ValueError, match="corrupt input"):
        oracle_goref.s2_read_stream(oracle_goref.s2_stream(b"", add_index=True), 16)


@pytest.mark.parametrize("level", [0, 1, 2])
def test_s2_facade_host_logic_equals_the_translated_writer(oracle, monkeypatch, level):
    """Host logic of compress_amd.s2.Writer — how Write / Flush / Close sequences become chunks, where the stream identifier, the index
    entries, the padding chunk and the index go — against the reference's own Writer for the same options and call sequences, on
    the CPU.  Nothing of the product changes for it: the test stands in for the device underneath the class (a BlockEncoder whose
    EncodeStreamDevice frames the chunks with the oracle, tensors that stay in host memory); device == oracle on that call is the GPU
    suite's business, and the same cases run end to end on the device in test_device_writer_equals_the_translated_writer."""
    import ctypes as C
    import torch
    from compress_amd import s2

    class _HostEncoder:
        def __init__(self, device=0, stream=None, level=0, variant=None):
            self.level = int(level)

        def EncodeStreamDevice(self, src_ptr, off, dst_ptr, cap, with_stream_id=True):
            n = int(off[-1])
            src = np.ctypeslib.as_array((C.c_uint8 * n).from_address(src_ptr)).copy() if n else np.zeros(0, dtype=np.uint8)
            st, oo = oracle.s2_encode_stream(src, off, with_stream_id=with_stream_id, level=self.level)
            assert len(st) <= cap
            C.memmove(dst_ptr, st.ctypes.data, len(st))
            return oo

        def Close(self):
            pass
    monkeypatch.setattr(s2, "BlockEncoder", _HostEncoder)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    j, t, h = _inputs()
    big = (j + t) * 4
    lvl = {0: [], 1: [s2.WriterBetterCompression()], 2: [s2.WriterBestCompression()]}[level]
    cases = [(dict(block_size=65536), j, ()), (dict(block_size=4096), j[:50000], (1000, 9000, 9001, 30000)), (dict(), big if level < 2 else big[:1500000], (2000000,)),
             (dict(block_size=65536, add_index=True), big if level < 2 else big[:700000], ()), (dict(block_size=65536, add_index=True, padding=4096), j + h, (70000,)),
             (dict(block_size=65536, padding=1024), t[:200000], ()), (dict(block_size=65536, snappy=True), j + h[:70000], (65536,)),
             (dict(block_size=65536, flush_on_write=True), j[:100000], (10, 70000)), (dict(block_size=65536), b"", ()),
             (dict(block_size=16384, add_index=True, padding=512), j[:16384 * 3], (16384, 32768))]
    for kw, data, cuts in cases:
        opts = list(lvl)
        if "block_size" in kw:
            opts.append(s2.WriterBlockSize(kw["block_size"]))
        if kw.get("add_index"):
            opts.append(s2.WriterAddIndex())
        if kw.get("padding"):
            opts += [s2.WriterPadding(kw["padding"]), s2.WriterPaddingSrc(_Zeros())]
        if kw.get("snappy"):
            opts.append(s2.WriterSnappyCompat())
        if kw.get("flush_on_write"):
            opts.append(s2.WriterFlushOnWrite())
        sink = io.BytesIO()
        w = s2.NewWriter(sink, *opts, batch_bytes=1 << 20)
        pos = 0
        for c in cuts:
            if c <= len(data):
                if c > pos:
                    w.Write(data[pos:c])
                    pos = c
                w.Flush()
        if len(data) > pos:
            w.Write(data[pos:])
        w.Close()
        want = oracle_goref.s2_stream(data, cuts, level=level, **kw)
        got = sink.getvalue()
        assert got == want, "level %d %r len %d cuts %r: %d bytes vs the reference's %d" % (level, kw, len(data), cuts, len(got), len(want))
    # EncodeBuffer (writer.go:357-453): what is buffered goes out first, then ALL of buf as chunks now, short tail included — for the
    # bytes, a Flush in front of buf and one behind it; ReadFrom (writer.go:220-268): block-size reads, a short last one
    for a, b in ((1000, 150000), (65536, 65536 * 2), (0, 70000)):
        sink = io.BytesIO()
        w = s2.NewWriter(sink, *lvl, s2.WriterBlockSize(65536), batch_bytes=1 << 20)
        w.Write(j[:a])
        w.EncodeBuffer(j[a:b])
        w.Write(j[b:])
        w.Close()
        assert sink.getvalue() == oracle_goref.s2_stream(j, level=level, block_size=65536, encode_buffer=(a, b)), ("EncodeBuffer", a, b)
        assert sink.getvalue() == oracle_goref.s2_stream(j, (a, b), level=level, block_size=65536), ("EncodeBuffer as two Flush points", a, b)
        sink = io.BytesIO()
        w = s2.NewWriter(sink, *lvl, s2.WriterBlockSize(65536), batch_bytes=1 << 20)
        w.Write(j[:a])
        w.ReadFrom(io.BytesIO(j[a:]))
        w.Close()
        assert sink.getvalue() == oracle_goref.s2_stream(j, level=level, block_size=65536, readfrom_at=a), ("ReadFrom", a)
        assert sink.getvalue() == oracle_goref.s2_stream(j, (a,), level=level, block_size=65536), ("ReadFrom as a Flush point", a)
