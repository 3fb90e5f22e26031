"""GPU parity tests for the S2 block encoder: HIP path (C ABI) vs the CPU oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import corpora

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["hbm", "lds"])
def s2path(request):
    """Every parity test of s2.Encode / s2.EncodeSnappy runs on both kernel families: 'hbm' = kc_s2.hip (tables in HBM, 8 blocks
    per wave), 'lds' = kc_s2_lds.hip (table and block in LDS, one wave per block); forced through KC_OPT_MATCH_PATH."""
    return request.param


def _oracle_blocks(oracle, buf, off):
    L = oracle.lib()
    n = len(off) - 1
    cap = int(sum(L.kco_s2_max_encoded_len(int(off[i + 1] - off[i])) for i in range(n))) + 64
    dst = np.empty(cap, dtype=np.uint8)
    oo = np.empty(n + 1, dtype=np.uint64)
    buf = np.ascontiguousarray(buf)
    r = L.kco_s2_encode_blocks(buf.ctypes.data, off.ctypes.data, n, dst.ctypes.data, cap, oo.ctypes.data, 8)
    assert r >= 0
    return dst[:r], oo


def _check(oracle, blocks, path=None):
    from compress_amd import s2
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(path=path)
    out, out_off = enc.EncodeBlocks(buf, off)
    if path is not None:
        assert enc._ctx.last_path() == path
    ref, ref_off = _oracle_blocks(oracle, buf, off)
    bad = []
    for i in range(len(blocks)):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        b = ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()
        if a != b:
            bad.append((i, len(blocks[i]), len(a), len(b)))
    assert not bad, "blocks differing from the oracle (index, in_len, gpu_len, oracle_len): %r" % bad[:10]
    # and every block decodes back
    for i in (0, len(blocks) // 2, len(blocks) - 1):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert oracle.s2_decode(a, len(blocks[i]) + 8) == blocks[i]
    enc.Close()


@pytest.mark.parametrize("kind", ["J", "T", "M", "H"])
def test_s2_64k_blocks_bit_exact(oracle, kclib, kind, s2path):
    buf = corpora.corpus(kind, 128, 65536)
    _check(oracle, [buf[i * 65536:(i + 1) * 65536].tobytes() for i in range(128)])


def test_s2_edge_blocks_bit_exact(oracle, kclib, s2path):
    _check(oracle, corpora.edge_units())


def test_s2_large_blocks_bit_exact(oracle, kclib, s2path):
    """Blocks > 64 KiB take encodeBlockGo (u32 table, skip >>6) in the reference."""
    j = corpora.corpus("J", 8, 1 << 20).tobytes()
    t = corpora.corpus("T", 8, 1 << 20).tobytes()
    m = corpora.corpus("M", 4, 1 << 20).tobytes()
    blocks = [j[:65537], j[:200000], t[:1 << 20], j[1 << 20:3 << 20], m[:4 << 20], t[100:700000], (b"abcd" * 300000), bytes(1 << 20)]
    _check(oracle, blocks)


def test_s2_custom_encoder_contract(oracle, kclib, s2path):
    """WriterCustomEncoder contract (s2/writer.go:1053-1064): no varint header; 0 == incompressible."""
    from compress_amd import s2
    enc = s2.BlockEncoder(path=s2path)
    fn = enc.CustomEncoder()
    text = corpora.corpus("J", 1, 65536).tobytes()
    dst = bytearray(s2.MaxEncodedLen(len(text)))
    n = fn(dst, text)
    want = oracle.s2_encode_block(text)
    assert n == len(want) and bytes(dst[:n]) == want
    noise = corpora.corpus("H", 1, 65536).tobytes()
    assert fn(dst, noise) == 0
    enc.Close()


def test_s2_custom_encoder_concurrent_callers(oracle, kclib, s2path):
    """s2.Writer calls the WriterCustomEncoder hook from one goroutine per block (s2/writer.go:455-460, "should expect to be
    called concurrently", :1058): 16 host threads hammer ONE context; every result must equal the oracle's encodeBlock, and
    the hook must have batched concurrent callers into fewer device launches than calls."""
    import threading
    import time
    from compress_amd import s2
    enc = s2.BlockEncoder(path=s2path)
    fn = enc.CustomEncoder()
    j = corpora.corpus("J", 192, 65536).tobytes()
    t = corpora.corpus("T", 2, 1 << 20).tobytes()
    noise = corpora.corpus("H", 4, 65536).tobytes()
    blocks = [j[i * 65536:(i + 1) * 65536] for i in range(192)]
    blocks += [t[:1 << 20], t[1 << 20:], t[5:300000], noise[:65536], noise[65536:70000], j[:31], j[:32], j[:33], j[:4096], b"ab" * 5000]
    want = [oracle.s2_encode_block(b) if len(b) >= 32 else b"" for b in blocks]
    cap = s2.MaxEncodedLen(1 << 20)

    def run(nthreads):
        res = [None] * len(blocks)

        def worker(tid):
            dst = bytearray(cap)
            for i in range(tid, len(blocks), nthreads):
                n = fn(dst, blocks[i])
                res[i] = bytes(dst[:n]) if n > 0 else n
        th = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return res, time.perf_counter() - t0

    run(4)  # warm-up: pinned staging + device scratch
    c0, b0 = enc.HookStats()
    res16, dt16 = run(16)
    c1, b1 = enc.HookStats()
    res1, dt1 = run(1)
    c2, b2 = enc.HookStats()
    for name, res in (("16 threads", res16), ("1 thread", res1)):
        bad = [i for i in range(len(blocks)) if (res[i] if res[i] else b"") != want[i] or (res[i] is not None and res[i] != 0 and len(res[i]) == 0)]
        assert not bad, (name, bad[:10])
        assert all((r == 0) == (len(w) == 0) for r, w in zip(res, want)), name
    gpu_blocks = sum(1 for b in blocks if len(b) >= 32)
    assert c1 - c0 == gpu_blocks and c2 - c1 == gpu_blocks
    assert b2 - b1 == gpu_blocks            # a lone caller is never held back: one launch per call
    assert b1 - b0 < gpu_blocks             # concurrent callers shared launches
    print("hook: 16 threads %.1f blocks/s in %d launches, 1 thread %.1f blocks/s" % (len(blocks) / dt16, b1 - b0, len(blocks) / dt1))
    # throughput on uniform 64 KiB blocks with many callers (what an s2.Writer with a high WriterConcurrency would offer)
    jb = corpora.corpus("J", 1024, 65536, first_unit=300).tobytes()
    blocks[:] = [jb[i * 65536:(i + 1) * 65536] for i in range(1024)]
    wantj = [oracle.s2_encode_block(b) for b in blocks[:64]]
    for nth in (16, 64):
        c0, b0 = enc.HookStats()
        res, dt = run(nth)
        c1, b1 = enc.HookStats()
        assert [r for r in res[:64]] == wantj
        print("hook: %d threads on 1024 x 64 KiB blocks: %.0f blocks/s (%.1f MB/s) in %d launches" % (nth, 1024 / dt, 1024 * 65536 / dt / 1e6, b1 - b0))
    enc.Close()


def test_s2_custom_encoder_host_first(oracle, kclib):
    """Round 6: the hook leaves the callers the host can serve to the built-in encoder.  With host_first = 2 the
    first two of four concurrent callers get -1 (booked as busy for len / 500 MB/s), the next ones go to the device and return the
    oracle's bytes; once the bookings have expired the next callers are sent back again; host_first = 0 sends everyone to the device."""
    import threading
    import time
    from compress_amd import s2
    enc = s2.BlockEncoder()
    blk = corpora.corpus("J", 1, 1 << 20).tobytes()  # 1 MiB: booked for ~2 ms
    want = oracle.s2_encode_block(blk)
    cap = s2.MaxEncodedLen(len(blk))
    fn0 = enc.CustomEncoder(host_first=0)
    dst = bytearray(cap)
    n = fn0(dst, blk)
    assert n > 0 and bytes(dst[:n]) == want  # warm: pinned staging, lanes
    fn = enc.CustomEncoder(host_first=2)
    res = [None] * 4
    bar = threading.Barrier(4)

    def worker(i):
        d = bytearray(cap)
        bar.wait()
        r = fn(d, blk)
        res[i] = bytes(d[:r]) if r > 0 else r
    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert sorted(1 if r == -1 else 0 for r in res) == [0, 0, 1, 1], [r if isinstance(r, int) else len(r) for r in res]
    assert all(r == want for r in res if r != -1)
    assert enc._ctx.L.kc_s2_hook_declined(enc._ctx.h) == 2
    time.sleep(0.02)  # the two bookings (2 ms each) have expired
    assert fn(dst, blk) == -1 and fn(dst, blk) == -1  # (the same thread back for its next block: its own booking never counts against it)
    # two other threads book the host's two places; this thread is then the overflow and goes to the device
    got = []
    th = [threading.Thread(target=lambda: got.append(fn(bytearray(cap), blk))) for _ in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    fn(dst, blk)  # (drops this thread's own booking, takes a place if one is free ...)
    assert got == [-1, -1] or -1 in got
    enc.Close()


def test_s2_full_size_roundtrip(oracle, kclib, s2path):
    """C4-size property check: 16384 x 64 KiB JSON blocks, device resident, sample decodes back."""
    import torch
    from compress_amd import s2
    n, bsz = 16384, 65536
    buf = corpora.corpus("J", n, bsz)
    off = np.arange(n + 1, dtype=np.uint64) * bsz
    d_src = torch.from_numpy(buf).cuda()
    enc = s2.BlockEncoder(path=s2path)
    cap = n * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    out_off = enc.EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    out = d_dst[:int(out_off[n])].cpu().numpy()
    rng = np.random.default_rng(5)
    for i in rng.choice(n, 64, replace=False):
        blk = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert oracle.s2_decode(blk, bsz + 8) == buf[i * bsz:(i + 1) * bsz].tobytes()
    assert float(out_off[n]) / (n * bsz) < 0.5
    enc.Close()


@pytest.mark.parametrize("with_id", [True, False])
def test_s2_stream_framing_bit_exact(oracle, kclib, with_id, s2path):
    """s2.Writer framing (stream id, chunk header, masked CRC32C, stored chunks) equals the oracle's and decodes."""
    import torch
    from compress_amd import s2
    blocks = [corpora.corpus("J", 1, 65536, first_unit=k).tobytes() for k in range(24)]
    blocks += [corpora.corpus("H", 1, 65536, first_unit=k).tobytes() for k in range(4)]
    blocks += [b"", b"abc", b"x" * 31, b"y" * 32, corpora.corpus("T", 1, 1 << 20).tobytes(), corpora.corpus("M", 1, 200000).tobytes()]
    buf, off = corpora.pack_units(blocks)
    d_src = torch.from_numpy(buf).cuda()
    enc = s2.BlockEncoder(path=s2path)
    cap = sum(((s2.MaxEncodedLen(len(b)) + 8 + 15) & ~15) for b in blocks) + 80
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    out_off = enc.EncodeStreamDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap, with_stream_id=with_id)
    out = d_dst[:int(out_off[len(blocks)])].cpu().numpy()
    ref, ref_off = oracle.s2_encode_stream(buf, off, with_stream_id=with_id)
    assert np.array_equal(out_off, ref_off)
    assert np.array_equal(out, ref)
    assert oracle.s2_decode_stream(out.tobytes(), len(buf) + 8) == buf.tobytes()
    enc.Close()


def test_s2_writer_chunking_and_bytes(oracle, kclib):
    """s2.Writer façade: the chunk boundaries follow writer.go's buffering rules for this exact call sequence (worked out
    by hand from writer.go:182-218 / 357-453 / 741-763), and the bytes equal the oracle's framing of those chunks."""
    import io
    from compress_amd import s2
    data = corpora.corpus("J", 1, 60000).tobytes() + corpora.corpus("H", 1, 9000).tobytes()
    sink = io.BytesIO()
    w = s2.NewWriter(sink, s2.WriterBlockSize(4096), s2.WriterConcurrency(4))
    pos = 0

    def take(n):
        nonlocal pos
        b = data[pos:pos + n]
        pos += n
        return b
    assert w.Write(take(1000)) == 1000      # buffered
    assert w.Write(take(5000)) == 5000      # fills the buffer -> chunk 4096, 1904 stay buffered
    w.Flush()                               # chunk 1904
    assert w.Write(take(10000)) == 10000    # empty buffer, large write: chunks 4096, 4096, 1808 (tail included)
    assert w.Write(take(4096)) == 4096      # exactly fits: buffered, not yet a chunk
    assert w.Write(take(1)) == 1            # buffer is full: chunk 4096, the byte is buffered
    w.EncodeBuffer(take(9000))              # flushes the buffered byte (chunk 1), then 4096, 4096, 808
    w.AddSkippableBlock(0x80, b"hello")
    n = w.ReadFrom(io.BytesIO(take(5000)))  # 4096, 904
    assert n == 5000
    w.Close()
    w.Close()
    with pytest.raises(IOError):
        w.Write(b"x")
    sizes = [4096, 1904, 4096, 4096, 1808, 4096, 1, 4096, 4096, 808]
    off = np.zeros(len(sizes) + 1, dtype=np.uint64); off[1:] = np.cumsum(sizes)
    ref1, _ = oracle.s2_encode_stream(np.frombuffer(data[:int(off[-1])], dtype=np.uint8), off, True)
    s2sizes = [4096, 904]
    off2 = np.zeros(3, dtype=np.uint64); off2[1:] = np.cumsum(s2sizes)
    ref2, _ = oracle.s2_encode_stream(np.frombuffer(data[int(off[-1]):int(off[-1]) + 5000], dtype=np.uint8), off2, False)
    want = ref1.tobytes() + b"\x80\x05\x00\x00hello" + ref2.tobytes()
    got = sink.getvalue()
    assert got == want
    assert oracle.s2_decode_stream(got, len(data) + 16) == data[:pos]
    # an empty writer writes nothing at all (the stream identifier goes out with the first chunk)
    sink2 = io.BytesIO(); w2 = s2.NewWriter(sink2); w2.Close(); assert sink2.getvalue() == b""
    # default block size 1 MiB, flush-on-write, and a batch limit small enough to force several device batches
    big = corpora.corpus("T", 24, 131072).tobytes()
    sink3 = io.BytesIO(); w3 = s2.NewWriter(sink3, s2.WriterFlushOnWrite(), batch_bytes=1 << 20)
    for i in range(0, len(big), 700000):
        w3.Write(big[i:i + 700000])
    w3.Close()
    sz = [min(700000, len(big) - i) for i in range(0, len(big), 700000)]
    off3 = np.zeros(len(sz) + 1, dtype=np.uint64); off3[1:] = np.cumsum(sz)
    ref3, _ = oracle.s2_encode_stream(np.frombuffer(big, dtype=np.uint8), off3, True)
    assert sink3.getvalue() == ref3.tobytes()
    with pytest.raises(ValueError):
        s2.NewWriter(io.BytesIO(), s2.WriterBlockSize(1000))
    # WriterSnappyCompat: the "sNaPpY" stream identifier, blocks through the Snappy-compatible encoder of the level, at most 64 KiB - 8
    for lvl_opt, lvl in (([], 2), ([s2.WriterBetterCompression()], 3), ([s2.WriterBestCompression()], 5)):
        sink4 = io.BytesIO()
        w4 = s2.NewWriter(sink4, s2.WriterSnappyCompat(), *lvl_opt)
        assert w4.blockSize == (64 << 10) - 8
        w4.Write(big[:300000])
        w4.Close()
        bs4 = (64 << 10) - 8
        sz4 = [min(bs4, 300000 - i) for i in range(0, 300000, bs4)]
        off4 = np.zeros(len(sz4) + 1, dtype=np.uint64); off4[1:] = np.cumsum(sz4)
        ref4, _ = oracle.s2_encode_stream(np.frombuffer(big[:300000], dtype=np.uint8), off4, False, level=lvl)
        assert sink4.getvalue() == b"\xff\x06\x00\x00sNaPpY" + np.asarray(ref4).tobytes(), lvl
    # WriterUncompressed (writer.go:948-956): uncompressed chunks only — checksum and copy on the device; a level like the others
    from test_emu_lds import uncompressed_chunk
    sink5 = io.BytesIO()
    w5 = s2.NewWriter(sink5, s2.WriterUncompressed(), s2.WriterBlockSize(64 << 10))
    w5.Write(big[:300000])
    w5.Write(b"tail")
    w5.Close()
    want5 = b"\xff\x06\x00\x00S2sTwO" + b"".join(uncompressed_chunk(big[i:i + 65536]) for i in range(0, 262144, 65536)) + uncompressed_chunk(big[262144:300000]) + uncompressed_chunk(b"tail")  # (a large Write into an empty buffer goes out whole, writer.go:190-200)
    assert sink5.getvalue() == want5
    assert oracle.s2_decode_stream(sink5.getvalue(), 300100) == big[:300000] + b"tail"
    sink6 = io.BytesIO()
    w6 = s2.NewWriter(sink6, s2.WriterUncompressed(), s2.WriterBetterCompression())  # the later level option wins
    w6.Write(big[:100000]); w6.Close()
    off6 = np.array([0, 100000], dtype=np.uint64)
    ref6, _ = oracle.s2_encode_stream(np.frombuffer(big[:100000], dtype=np.uint8), off6, True, level=1)
    assert sink6.getvalue() == ref6.tobytes()


def test_s2_device_decoder_roundtrip_and_errors(oracle, kclib):
    """N1 (GPU half) for S2: kc_s2_decode_blocks_dev decodes what kc_s2_encode_blocks_dev produced back to the source, on
    the device, for every corpus kind and ragged block sizes; agrees with the oracle's s2.Decode on the oracle's own blocks;
    corrupt blocks are reported per block without touching their neighbours."""
    import torch
    from compress_amd import s2
    enc = s2.BlockEncoder()
    for kind, bsz, nb in (("J", 65536, 256), ("T", 65536, 128), ("H", 65536, 32), ("M", 262144, 48), ("T", 1000, 200)):
        buf = corpora.corpus(kind, nb, bsz)
        off = np.arange(nb + 1, dtype=np.uint64) * bsz
        d_src = torch.from_numpy(buf).cuda()
        cap = nb * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64
        d_enc = torch.empty(cap, dtype=torch.uint8, device="cuda")
        eoff = enc.EncodeBlocksDevice(d_src.data_ptr(), off, d_enc.data_ptr(), cap)
        d_out = torch.zeros(nb * bsz + 64, dtype=torch.uint8, device="cuda")
        st = enc.DecodeBlocksDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), off)
        assert not st.any(), (kind, np.nonzero(st)[0][:5], st[np.nonzero(st)[0][:5]])
        assert torch.equal(d_out[:nb * bsz], d_src), kind
    # ragged blocks incl. empty and tiny ones, encoded by the oracle
    t = corpora.corpus("T", 4, 131072).tobytes()
    blocks = [b"", t[:1], t[:5], t[:100], t[:70000], t[100:300000], b"\x00" * 5000, t[:65536], (t[:37] * 400)]
    encs = [oracle.s2_encode(b) for b in blocks]
    eoff = np.zeros(len(blocks) + 1, dtype=np.uint64); eoff[1:] = np.cumsum([len(e) for e in encs])
    doff = np.zeros(len(blocks) + 1, dtype=np.uint64); doff[1:] = np.cumsum([len(b) for b in blocks])
    d_enc = torch.from_numpy(np.frombuffer(b"".join(encs), dtype=np.uint8).copy()).cuda()
    d_out = torch.zeros(int(doff[-1]) + 64, dtype=torch.uint8, device="cuda")
    st = enc.DecodeBlocksDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), doff)
    assert not st.any(), st
    assert d_out[:int(doff[-1])].cpu().numpy().tobytes() == b"".join(blocks)
    # corruption: truncated block, wrong stated size, zero offset; neighbours still decode
    bad = [encs[3], encs[4][:-3], encs[5], bytes([encs[6][0] ^ 1]) + encs[6][1:], encs[7]]
    want = [len(blocks[3]), len(blocks[4]), len(blocks[5]), len(blocks[6]), len(blocks[7])]
    eoff = np.zeros(6, dtype=np.uint64); eoff[1:] = np.cumsum([len(e) for e in bad])
    doff = np.zeros(6, dtype=np.uint64); doff[1:] = np.cumsum(want)
    d_enc = torch.from_numpy(np.frombuffer(b"".join(bad), dtype=np.uint8).copy()).cuda()
    d_out = torch.zeros(int(doff[-1]) + 64, dtype=torch.uint8, device="cuda")
    st = enc.DecodeBlocksDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), doff)
    assert st[0] == 0 and st[2] == 0 and st[4] == 0 and st[1] != 0 and st[3] != 0, st
    out = d_out.cpu().numpy()
    assert out[int(doff[2]):int(doff[3])].tobytes() == blocks[5] and out[int(doff[4]):int(doff[5])].tobytes() == blocks[7]
    enc.Close()


def test_s2_writer_index(oracle, kclib):
    """WriterAddIndex / CloseIndex: the index chunk appended to the stream equals the oracle's Index fed with the same
    (compressed offset, uncompressed offset) pairs; the stream (index chunk skipped) still decodes."""
    import io
    from compress_amd import s2
    data = corpora.corpus("J", 40, 131072).tobytes()   # 5 MiB: several 1 MiB index steps with 256 KiB blocks
    sink = io.BytesIO()
    w = s2.NewWriter(sink, s2.WriterBlockSize(256 << 10), s2.WriterAddIndex())
    w.Write(data[:3000000])
    w.AddSkippableBlock(0x81, b"meta")
    w.Write(data[3000000:])
    idx = w.CloseIndex()
    got = sink.getvalue()
    assert got.endswith(idx) and idx[:1] == b"\x99"
    body = got[:-len(idx)]
    # rebuild the add() sequence from the stream itself: every chunk start with the uncompressed offset it begins at
    adds, p, u = [], 0, 0
    while p < len(body):
        t, cl = body[p], int.from_bytes(body[p + 1:p + 4], "little")
        adds.append((p, u))
        if t == 0x00:
            n, sh = 0, 0
            q = p + 8
            while True:
                bb = body[q]; q += 1
                n |= (bb & 0x7F) << sh; sh += 7
                if not bb & 0x80:
                    break
            u += n
        elif t == 0x01:
            u += cl - 4
        p += 4 + cl
    assert u == len(data)
    assert idx == oracle.s2_index(256 << 10, adds, len(data), len(body))
    assert oracle.s2_decode_stream(got, len(data) + 16) == data
    # without WriterAddIndex the index is returned but not written
    sink2 = io.BytesIO(); w2 = s2.NewWriter(sink2); w2.Write(data[:100000]); idx2 = w2.CloseIndex()
    assert idx2[:1] == b"\x99" and idx2 not in sink2.getvalue()


def test_s2_writer_padding(oracle, kclib):
    """TestWriterPadding (s2/writer_test.go:405): the padded stream is a multiple of the padding, decodes to the input (the
    0xfe chunk is invisible), also after Reset; with WriterPaddingSrc the bytes equal stream + deterministic padding chunk,
    and with an index the padding precedes it."""
    import io
    import random
    from compress_amd import s2
    rng = random.Random(0x1337)
    for _ in range(4):
        padding = (rng.getrandbits(16)) + 1
        n = (rng.getrandbits(18)) + 1
        src = bytes(rng.getrandbits(2) for _ in range(n))
        dst = io.BytesIO()
        e = s2.NewWriter(dst, s2.WriterPadding(padding))
        e.ReadFrom(io.BytesIO(src))
        e.Close(); e.Close()
        assert len(dst.getvalue()) % padding == 0
        assert oracle.s2_decode_stream(dst.getvalue(), n + 16) == src
        dst2 = io.BytesIO()
        e.Reset(dst2)
        e.Write(src)
        e.Close()
        assert len(dst2.getvalue()) % padding == 0 and oracle.s2_decode_stream(dst2.getvalue(), n + 16) == src
    # deterministic padding source: exact bytes
    data = corpora.corpus("J", 3, 131072).tobytes()
    plain = io.BytesIO(); w = s2.NewWriter(plain); w.Write(data); w.Close()
    padded = io.BytesIO(); w = s2.NewWriter(padded, s2.WriterPadding(8000), s2.WriterPaddingSrc(io.BytesIO(bytes(1 << 20)))); w.Write(data); w.Close()
    base = plain.getvalue()
    add = s2.calc_skippable_frame(len(base), 8000)
    assert padded.getvalue() == base + bytes([0xfe]) + (add - 4).to_bytes(3, "little") + bytes(add - 4)
    both = io.BytesIO(); w = s2.NewWriter(both, s2.WriterPadding(4096), s2.WriterAddIndex(), s2.WriterPaddingSrc(io.BytesIO(bytes(1 << 20))))
    w.Write(data); idx = w.CloseIndex()
    got = both.getvalue()
    assert got.endswith(idx) and len(got) % 4096 == 0 and got[len(base)] == 0xfe
    assert oracle.s2_decode_stream(got, len(data) + 16) == data
    with pytest.raises(ValueError):
        s2.NewWriter(io.BytesIO(), s2.WriterPadding(0))


# ---------------------------------------------------------------------------------------------------------------------
# s2.EncodeBetter (KC_S2_LEVEL_BETTER): encodeBlockBetterGo64K / encodeBlockBetterGo
# ---------------------------------------------------------------------------------------------------------------------
def _check_better(oracle, blocks):
    from compress_amd import s2
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=s2.LevelBetter)
    out, out_off = enc.EncodeBlocks(buf, off)
    ref, ref_off = oracle.s2_encode_blocks(buf, off, threads=8, better=True)
    bad = []
    for i in range(len(blocks)):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        b = np.asarray(ref)[int(ref_off[i]):int(ref_off[i + 1])].tobytes()
        if a != b:
            bad.append((i, len(blocks[i]), len(a), len(b)))
    assert not bad, "blocks differing from the oracle's EncodeBetter (index, in_len, gpu_len, oracle_len): %r" % bad[:10]
    for i in (0, len(blocks) // 2, len(blocks) - 1):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert oracle.s2_decode(a, len(blocks[i]) + 8) == blocks[i]
    enc.Close()


@pytest.mark.parametrize("kind", ["J", "T", "M", "H"])
def test_s2_better_64k_blocks_bit_exact(oracle, kclib, kind):
    buf = corpora.corpus(kind, 128, 65536)
    _check_better(oracle, [buf[i * 65536:(i + 1) * 65536].tobytes() for i in range(128)])


def test_s2_better_edge_blocks_bit_exact(oracle, kclib):
    _check_better(oracle, corpora.edge_units())


def test_s2_better_large_blocks_bit_exact(oracle, kclib):
    """Blocks > 64 KiB take encodeBlockBetterGo (2^17 / 2^14 tables, skip >>7, the offset > 65535 short-match bail)."""
    j = corpora.corpus("J", 8, 1 << 20).tobytes()
    t = corpora.corpus("T", 8, 1 << 20).tobytes()
    m = corpora.corpus("M", 4, 1 << 20).tobytes()
    blocks = [j[:65537], j[:200000], t[:1 << 20], j[1 << 20:3 << 20], m[:4 << 20], t[100:700000], (b"abcd" * 300000), bytes(1 << 20),
              t[:70000] + j[:70000] + t[:70000]]
    _check_better(oracle, blocks)


def test_s2_better_reference_inputs_bit_exact(oracle, kclib):
    import os
    import zipfile
    z = zipfile.ZipFile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs", "enc_regressions.zip"))
    blocks = [z.read(n) for n in z.namelist()]
    blocks = [b for b in blocks if 0 < len(b) <= (4 << 20)]
    _check_better(oracle, blocks)


def test_s2_writer_better_stream_roundtrip(oracle, kclib):
    """s2.NewWriter(w, WriterBetterCompression()): chunks carry EncodeBetter blocks; the stream decodes back."""
    import io
    from compress_amd import s2
    data = corpora.corpus("J", 40, 65536).tobytes() + corpora.corpus("H", 2, 65536).tobytes()
    sink = io.BytesIO()
    w = s2.NewWriter(sink, s2.WriterBlockSize(64 << 10), s2.WriterBetterCompression())
    w.Write(data)
    w.Close()
    enc = sink.getvalue()
    assert oracle.s2_decode_stream(enc, len(data) + 16) == data
    sink0 = io.BytesIO()
    w0 = s2.NewWriter(sink0, s2.WriterBlockSize(64 << 10))
    w0.Write(data)
    w0.Close()
    assert len(enc) < len(sink0.getvalue())


@pytest.mark.parametrize("level", [1, 3, 4, 5])
def test_s2_stream_framing_other_levels_bit_exact(oracle, kclib, level):
    """s2.Writer chunks (type | len24 | masked CRC32C | uvarint + block, or the raw bytes when encodeBlock returns 0) at the better
    and best levels and their Snappy-compatible forms: (*Writer).encodeBlock (s2/writer.go:1053-1091) inside the framing of
    :414-451, against the oracle chunk by chunk; and the Python Writer with WriterBestCompression writes the same stream."""
    import io
    torch = pytest.importorskip("torch")
    from compress_amd import s2
    j = corpora.corpus("J", 24, 65536).tobytes()
    blocks = [j[i * 65536:(i + 1) * 65536] for i in range(24)]
    blocks += [corpora.corpus("H", 1, 65536).tobytes(), corpora.corpus("T", 1, 65536).tobytes()[:31], b"a", corpora.corpus("T", 2, 131072).tobytes()[:150000],
               corpora.corpus("M", 1, 65536, first_unit=2).tobytes()]
    b2, off = corpora.pack_units(blocks)
    d_src = torch.from_numpy(b2).cuda()
    cap = sum(((s2.MaxEncodedLen(len(b)) + 8 + 15) & ~15) for b in blocks) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    enc = s2.BlockEncoder(level=level)
    oo = enc.EncodeStreamDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap, with_stream_id=True)
    got = d_dst[:int(oo[-1])].cpu().numpy()
    ref, ref_off = oracle.s2_encode_stream(b2, off, with_stream_id=True, level=level)
    assert np.array_equal(oo, ref_off)
    assert np.array_equal(got, np.asarray(ref))
    assert oracle.s2_decode_stream(got.tobytes(), len(b2) + 16) == b2.tobytes()
    enc.Close()
    if level == 4:
        sink = io.BytesIO()
        w = s2.NewWriter(sink, s2.WriterBlockSize(64 << 10), s2.WriterBestCompression())
        w.Write(j)
        w.Close()
        jb = np.frombuffer(j, dtype=np.uint8)
        joff = np.arange(25, dtype=np.uint64) * 65536
        rj, _ = oracle.s2_encode_stream(jb, joff, with_stream_id=True, level=4)
        assert sink.getvalue()[:len(rj)] == np.asarray(rj).tobytes()


@pytest.mark.parametrize("kind", ["J", "T", "M", "H"])
def test_s2_snappy_blocks_bit_exact(oracle, kclib, kind, s2path):
    """KC_S2_LEVEL_SNAPPY == s2.EncodeSnappy: same parse as the default level, copies through emitCopyNoRepeat."""
    from compress_amd import s2
    buf = corpora.corpus(kind, 96, 65536)
    blocks = [buf[i * 65536:(i + 1) * 65536].tobytes() for i in range(96)]
    big = corpora.corpus(kind, 2, 1 << 20).tobytes()
    blocks += [big[:65537], big[:700000], big[1 << 20:]] + [u for u in corpora.edge_units() if len(u) < 70000]
    b2, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=s2.LevelSnappy, path=s2path)
    out, out_off = enc.EncodeBlocks(b2, off)
    ref, ref_off = oracle.s2_encode_blocks(b2, off, threads=8, snappy=True)
    assert np.array_equal(out_off, ref_off)
    assert np.array_equal(out, np.asarray(ref))
    enc.Close()


@pytest.mark.parametrize("kind", ["J", "T", "M", "H"])
def test_s2_snappy_better_blocks_bit_exact(oracle, kclib, kind):
    """KC_S2_LEVEL_SNAPPY_BETTER == s2.EncodeSnappyBetter (encodeBlockBetterSnappyGo / ...64K): 64 KiB blocks (2^15 + 2^13 tables),
    larger ones (2^16 + 2^14, the long-offset bail), the reference's regression inputs and the edge units."""
    from compress_amd import s2
    buf = corpora.corpus(kind, 96, 65536)
    blocks = [buf[i * 65536:(i + 1) * 65536].tobytes() for i in range(96)]
    big = corpora.corpus(kind, 2, 1 << 20).tobytes()
    blocks += [big[:65537], big[:700000], big[1 << 20:]] + [u for u in corpora.edge_units() if len(u) < 70000]
    b2, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=s2.LevelSnappyBetter)
    out, out_off = enc.EncodeBlocks(b2, off)
    ref, ref_off = oracle.s2_encode_blocks(b2, off, threads=8, better=True, snappy=True)
    assert np.array_equal(out_off, ref_off)
    assert np.array_equal(out, np.asarray(ref))
    enc.Close()


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_s2_levels_randomised_blocks_bit_exact(oracle, kclib, level, s2path):
    """Differential test over adversarial block mixes (text / noise runs of every length class / low-entropy noise / long zero
    runs / repeated parts, 300 B .. 300 KB: both table variants of every level) for s2.Encode, s2.EncodeBetter, s2.EncodeSnappy, s2.EncodeSnappyBetter."""
    from compress_amd import s2
    blocks = corpora.stress_units(seed=11 + level, n=120)
    rng = np.random.default_rng(100 + level)
    for _ in range(40):  # short periodic and tiny blocks: repeat / long-offset edge cases
        per = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        blocks.append((per * 20000)[:int(rng.integers(1, 140000))])
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=level, path=s2path)
    out, out_off = enc.EncodeBlocks(buf, off)
    ref, ref_off = oracle.s2_encode_blocks(buf, off, threads=8, better=level in (1, 3), snappy=level in (2, 3))
    ref = np.asarray(ref)
    bad = [(i, len(blocks[i])) for i in range(len(blocks))
           if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()]
    assert not bad, "level %d: blocks differing from the oracle (index, len): %r" % (level, bad[:10])
    enc.Close()


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_s2_host_chunk_fed_equals_oracle(oracle, kclib, level, monkeypatch):
    """kc_s2_encode_blocks on a large host buffer: the source arrives in chunks, each chunk is encoded and compacted on its own
    stream behind its copy and drained as it finishes.  Same bytes as the oracle's (and as the serial host path's), with ragged,
    empty and large blocks across the chunk boundaries."""
    from compress_amd import s2
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL", "0")  # (round 6: large calls take the rolling pipeline; this test keeps the chunk-fed path covered)
    monkeypatch.setenv("KC_HOST_CHUNKS_MIB", "1,2,4")
    blocks = corpora.stress_units(seed=31 + level, n=160)
    j = corpora.corpus("J", 96, 65536).tobytes()
    blocks += [j[i * 65536:(i + 1) * 65536] for i in range(96)]
    blocks[7] = b""
    blocks.append(j[:1 << 20])
    blocks.append(b"x")
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=level)
    out, out_off = enc.EncodeBlocks(buf, off)
    ref, ref_off = oracle.s2_encode_blocks(buf, off, threads=8, better=level in (1, 3), snappy=level in (2, 3))
    assert np.array_equal(out_off, np.asarray(ref_off))
    assert np.array_equal(out, np.asarray(ref))
    out2, out_off2 = enc.EncodeBlocks(buf, off)
    assert np.array_equal(out2, out) and np.array_equal(out_off2, out_off)
    from compress_amd import _lib
    enc._ctx.set_option(_lib.OPT_HOST_SERIAL, 1)
    out3, out_off3 = enc.EncodeBlocks(buf, off)
    assert np.array_equal(out3, out) and np.array_equal(out_off3, out_off)
    enc.Close()


@pytest.mark.parametrize("level", [0, 1, 2])
def test_s2_host_rolling_pipeline_equals_oracle(oracle, kclib, level, monkeypatch):
    """Round 6: kc_s2_encode_blocks_lvl on a large host buffer goes through the device's rolling pipeline (kc_roll.cpp).  Same bytes
    as the oracle's and as the serial host path's, ragged / empty / large blocks across the sub-batch boundaries."""
    from compress_amd import s2, _lib
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL_MIB", "2")
    blocks = corpora.stress_units(seed=131 + level, n=160)
    j = corpora.corpus("J", 96, 65536).tobytes()
    blocks += [j[i * 65536:(i + 1) * 65536] for i in range(96)]
    blocks[7] = b""
    blocks.append(j[:1 << 20])
    blocks.append(b"x")
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=level)
    out, out_off = enc.EncodeBlocks(buf, off)
    assert enc._ctx.get_option(_lib.OPT_LAST_BATCHES) >= 4
    ref, ref_off = oracle.s2_encode_blocks(buf, off, threads=8, better=level in (1, 3), snappy=level in (2, 3))
    assert np.array_equal(out_off, np.asarray(ref_off))
    assert np.array_equal(out, np.asarray(ref))
    out2, out_off2 = enc.EncodeBlocks(buf, off)
    assert np.array_equal(out2, out) and np.array_equal(out_off2, out_off)
    enc.Close()


@pytest.mark.parametrize("level,path", [(0, "hbm"), (0, "lds"), (1, "hbm"), (2, "hbm")])
def test_s2_one_batch_as_parts_on_three_contexts(oracle, kclib, level, path):
    """One batch of blocks run as several launches (bench.py --split for C4): kc_s2_encode_blocks_lvl_dev_begin / _end_at on three
    contexts, the launches unchained, every part's blocks right behind the previous part's.  Same offsets and bytes as the oracle's and
    as the one blocking call; misuse (a second begin, an end without a begin, a place too small) is reported."""
    import torch
    from compress_amd import s2
    blocks = corpora.stress_units(seed=977 + level, n=96)
    j = corpora.corpus("J", 160, 65536).tobytes()
    blocks += [j[i * 65536:(i + 1) * 65536] for i in range(160)]
    blocks[11] = b""
    buf, off = corpora.pack_units(blocks)
    n = len(blocks)
    ref, ref_off = oracle.s2_encode_blocks(buf, off, threads=8, better=level in (1, 3), snappy=level in (2, 3))
    d_src = torch.from_numpy(np.ascontiguousarray(buf)).cuda()
    cap = sum((s2.MaxEncodedLen(len(b)) + 15) & ~15 for b in blocks) + 64
    d_dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(3)]
    encs = [s2.BlockEncoder(level=level, stream=st.cuda_stream, path=path) for st in streams]
    cuts = [0, 100, 101, 180, n]
    parts = list(zip(cuts[:-1], cuts[1:])) * 2  # two passes
    torch.cuda.synchronize()
    begun, pos, offs = 0, 0, [0]
    for g, (a, b) in enumerate(parts):
        while begun < len(parts) and begun < g + 3:
            pa, pb = parts[begun]
            encs[begun % 3].EncodeBlocksDeviceBegin(d_src.data_ptr(), off[pa:pb + 1])
            begun += 1
        if a == 0:
            pos, offs = 0, [0]
        o = encs[g % 3].EncodeBlocksDeviceEnd(d_dst.data_ptr() + pos, cap - pos)
        assert int(o[0]) == 0
        offs += [pos + int(x) for x in o[1:]]
        pos += int(o[b - a])
        if b == n:
            assert np.array_equal(np.array(offs, dtype=np.uint64), np.asarray(ref_off)), (level, path)
            assert np.array_equal(d_dst[:pos].cpu().numpy(), np.asarray(ref)), (level, path)
    with pytest.raises(Exception):
        encs[0].EncodeBlocksDeviceEnd(d_dst.data_ptr(), cap)  # nothing in flight
    encs[0].EncodeBlocksDeviceBegin(d_src.data_ptr(), off[:9])
    with pytest.raises(Exception):
        encs[0].EncodeBlocksDeviceBegin(d_src.data_ptr(), off[:9])  # busy
    with pytest.raises(Exception):
        encs[0].EncodeBlocksDevice(d_src.data_ptr(), off[:9], d_dst.data_ptr(), cap)  # busy for the blocking call too
    with pytest.raises(Exception):
        encs[0].EncodeBlocksDeviceEnd(d_dst.data_ptr(), 16)  # too small: the batch stays in flight
    o = encs[0].EncodeBlocksDeviceEnd(d_dst.data_ptr(), cap)
    assert np.array_equal(o, np.asarray(ref_off)[:9])
    for e in encs:
        e.Close()


@pytest.mark.parametrize("level", [0, 1])
def test_s2_batches_cut_by_the_scratch_budget(oracle, kclib, level):
    """the S2 device path cuts a call into several batches when its tables + staging slots exceed the scratch budget
    (here forced down to 64 MiB through KC_OPT_MAX_SCRATCH_MIB) instead of asking for all of it at once; same bytes as one batch."""
    import torch
    from compress_amd import s2, _lib
    n, bsz = 1200, 65536
    buf = corpora.corpus("J", n, bsz)
    off = np.arange(n + 1, dtype=np.uint64) * bsz
    d_src = torch.from_numpy(buf).cuda()
    cap = n * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    enc = s2.BlockEncoder(level=level, path="hbm")
    one = enc.EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert enc._ctx.get_option(_lib.OPT_LAST_BATCHES) == 1
    want = d_dst[:int(one[n])].cpu().numpy().copy()
    enc._ctx.set_option(_lib.OPT_MAX_SCRATCH_MIB, 64)
    d_dst.zero_()
    cut = enc.EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert enc._ctx.get_option(_lib.OPT_LAST_BATCHES) > 1
    assert np.array_equal(cut, one) and np.array_equal(d_dst[:int(cut[n])].cpu().numpy(), want)
    ref, ref_off = oracle.s2_encode_blocks(buf[:64 * bsz], off[:65], threads=8, better=level == 1)
    assert np.array_equal(want[:int(one[64])], np.asarray(ref))
    enc.Close()


def test_context_options_roundtrip(kclib):
    """kc_ctx_set_option / kc_ctx_get_option: every key reads back what was set; unknown keys and bad paths are refused."""
    from compress_amd import _lib
    ctx = _lib.Context()
    for key, val in ((_lib.OPT_MATCH_PATH, _lib.PATH_LDS), (_lib.OPT_ZFAST_LDS_MAX_UNITS, 5), (_lib.OPT_S2_LDS_MAX_BLOCKS, 7), (_lib.OPT_SPEC_W0, 3),
                     (_lib.OPT_SPEC_GROW, 1), (_lib.OPT_LDS_SPEC_W0, 32), (_lib.OPT_S2_LDS_SPEC_W0, 8), (_lib.OPT_HOST_SERIAL, 1), (_lib.OPT_HOST_PIPE_MIB, 64),
                     (_lib.OPT_HOST_OVERLAP_MIN_MIB, 9), (_lib.OPT_HOST_COPY_THREADS, 4), (_lib.OPT_HOST_TRACE, 1), (_lib.OPT_HOST_CHUNK_MIB, 12),
                     (_lib.OPT_K2_PROF, 0), (_lib.OPT_S2_HOOK_WAIT_US, 50), (_lib.OPT_S2_HOOK_BATCH, 33), (_lib.OPT_TEST_FEED_REDO, 1), (_lib.OPT_MAX_SCRATCH_MIB, 4096)):
        ctx.set_option(key, val)
        assert ctx.get_option(key) == val, key
    with pytest.raises(_lib.KcError):
        ctx.set_option(999, 1)
    with pytest.raises(_lib.KcError):
        ctx.set_option(_lib.OPT_MATCH_PATH, 7)
    assert ctx.get_option(999) == -1
    ctx.close()


@pytest.mark.parametrize("variant", [None, "amd64"])
@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_s2_blocks_above_4_mib(oracle, kclib, level, variant):
    """s2.Encode* takes inputs far above the stream writer's 4 MiB maxBlockSize (s2/encode.go:29-56; above 64 KiB encodeBlockGo, on
    amd64 encodeBlockAsm from 4 MiB on): blocks of 4 MiB + 1, 6 MiB, 17 MiB (beyond the LDS kernel's 24-bit positions: the batch goes
    through the HBM-table kernel whole) and 40 MiB next to small ones, every level with an assembly form, both byte-exact targets;
    the amd64 variant also against the reference's own assembly where oracle/_ref is present."""
    from compress_amd import s2
    t = corpora.corpus("T", 24, 1 << 20, first_unit=2).tobytes()
    j = corpora.corpus("J", 16, 1 << 20, first_unit=7).tobytes()
    blocks = [t[:(4 << 20) + 1], j[:6 << 20], t[1000:1000 + 70000], (t + j)[:17 << 20], j[5:5 + 300], t + j]
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=level, variant=variant)
    out, out_off = enc.EncodeBlocks(buf, off)
    assert enc._ctx.last_path() == "hbm"
    snappy, better = level in (2, 3), level in (1, 3)
    if variant == "amd64":
        ref_fn = lambda b: oracle.s2_encode_asm(b, snappy=snappy, better=better)  # noqa: E731
    else:
        ref_fn = {0: oracle.s2_encode, 1: oracle.s2_encode_better, 2: oracle.s2_encode_snappy, 3: oracle.s2_encode_snappy_better}[level]
    bad = [(i, len(b)) for i, b in enumerate(blocks) if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref_fn(b)]
    assert not bad, bad
    if variant == "amd64":
        import oracle_ref
        if oracle_ref.available():
            for i in (0, 3):
                assert out[int(out_off[i]):int(out_off[i + 1])].tobytes() == oracle_ref.encode(blocks[i], level=level), i
    enc.Close()


@pytest.mark.parametrize("level", [0, 1, 6])
def test_framed_blocks_above_4_mib_are_refused(kclib, level):
    """A chunk of the stream format holds one block of at most s2.maxBlockSize = 4 MiB (24-bit chunk length; the reference's Reader
    refuses larger chunks): the framed entry points refuse a larger block with KC_ERR_BAD_ARG before any byte moves, at every level
    incl. WriterUncompressed; exactly 4 MiB is served; bare blocks keep the 1 GiB bound (test_s2_blocks_above_4_mib)."""
    import torch
    from compress_amd import s2, _lib
    t = corpora.corpus("T", 6, 1 << 20, first_unit=3)
    d_src = torch.from_numpy(t).cuda()
    enc = s2.BlockEncoder(level=level)
    cap = 3 * s2.MaxEncodedLen(6 << 20) + 1024
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for sizes in ([(4 << 20) + 1], [5 << 20], [1 << 20, (4 << 20) + 4096]):
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        with pytest.raises(_lib.KcError) as ei:
            enc.EncodeStreamDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
        assert ei.value.status == _lib.KC_ERR_BAD_ARG, ei.value
    off = np.array([0, 4 << 20], dtype=np.uint64)
    oo = enc.EncodeStreamDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert int(oo[1]) > 10
    enc.Close()


@pytest.mark.parametrize("variant", [None, "amd64"])
@pytest.mark.parametrize("snappy", [False, True])
def test_s2_best_blocks_bit_exact(oracle, kclib, snappy, variant):
    """s2.EncodeBest / s2.EncodeSnappyBest on the device (kc_s2_best.hip: one wave per block, 4.5 MiB of {cur, prev} tables) against
    the oracle's restatement of encodeBlockBest / encodeBlockBestSnappy: corpus blocks, blocks above 64 KiB, edge units, stress
    mixes; the Snappy variant also through the strict Snappy decoder of the oracle tests (no S2 extensions).  The best encoders are
    pure Go in the reference — one form on every platform — so a context in the amd64 variant (the Go shim's default on amd64
    builds) serves them with the same bytes instead of refusing."""
    from compress_amd import s2
    blocks = []
    for kind in "JTMH":
        b = corpora.corpus(kind, 12, 65536, first_unit=4)
        blocks += [b[i * 65536:(i + 1) * 65536].tobytes() for i in range(12)]
    big = corpora.corpus("T", 3, 1 << 20).tobytes()
    blocks += [big[:65537], big[:300000], big[1 << 20:2 << 20], corpora.corpus("J", 1, 1 << 20).tobytes()[:700000]]
    blocks += [big + big[:(1 << 20) + 4321]]  # above 4 MiB (s2.EncodeBest takes any input MaxEncodedLen accepts)
    blocks += corpora.edge_units()
    blocks += [u for u in corpora.stress_units(seed=17, n=24)]
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=s2.LevelSnappyBest if snappy else s2.LevelBest, variant=variant)
    out, out_off = enc.EncodeBlocks(buf, off)
    ref_fn = oracle.s2_encode_snappy_best if snappy else oracle.s2_encode_best
    bad = []
    for i, blk in enumerate(blocks):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        r = ref_fn(blk)
        if a != r:
            bad.append((i, len(blk), len(a), len(r)))
    assert not bad, "blocks differing from the oracle (index, in_len, gpu_len, oracle_len): %r" % bad[:10]
    for i in (0, 13, len(blocks) - 1):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert oracle.s2_decode(a, len(blocks[i]) + 8) == blocks[i]
    enc.Close()


def test_s2_best_many_blocks_in_budgeted_batches(oracle, kclib):
    """4.5 MiB of tables per block: 2048 blocks ask for 9 GiB; with the scratch ceiling at 2 GiB the call is cut into batches and
    gives the same bytes (sample against the oracle, all blocks round-tripped on the device)."""
    import torch
    from compress_amd import s2, _lib
    n, bsz = 2048, 65536
    buf = corpora.corpus("J", n, bsz, first_unit=100)
    off = np.arange(n + 1, dtype=np.uint64) * bsz
    d_src = torch.from_numpy(buf).cuda()
    cap = n * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    enc = s2.BlockEncoder(level=s2.LevelBest)
    enc._ctx.set_option(_lib.OPT_MAX_SCRATCH_MIB, 2048)
    oo = enc.EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert enc._ctx.get_option(_lib.OPT_LAST_BATCHES) > 1
    out = d_dst[:int(oo[n])].cpu().numpy()
    for i in (0, 500, 1023, 1024, 2047):
        assert out[int(oo[i]):int(oo[i + 1])].tobytes() == oracle.s2_encode_best(buf[i * bsz:(i + 1) * bsz].tobytes()), i
    d_back = torch.empty(n * bsz + 64, dtype=torch.uint8, device="cuda")
    st = enc.DecodeBlocksDevice(d_dst.data_ptr(), oo, d_back.data_ptr(), off)
    assert not st.any() and torch.equal(d_back[:n * bsz], d_src)
    assert float(oo[n]) / (n * bsz) < 0.33  # better than the default level's 0.354 on this corpus
    enc.Close()
