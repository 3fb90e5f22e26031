"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import corpora

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


# "1L" = SpeedFastest forced onto the LDS-table match finder (kc_zstd_match_lds.hip), 1 = forced onto the HBM-table one
# (kc_zstd_match.hip): every parity test of this file runs on both (KC_OPT_MATCH_PATH, not an environment variable).
LEVELS = [1, "1L", 2, 3]
LEVELS_B = LEVELS + [4]  # + SpeedBestCompression (one wave per unit: the smaller tests)


def _li(level):
    return 1 if level == "1L" else int(level)


def _lo(level):
    from compress_amd import zstd
    if level == "1L":
        return [zstd.WithEncoderLevel(1), zstd.WithMatchPath("lds")]
    if level == 1:
        return [zstd.WithEncoderLevel(1), zstd.WithMatchPath("hbm")]
    return [zstd.WithEncoderLevel(level)]


def _enc(level=1, **kw):
    from compress_amd import zstd
    return zstd.NewWriter(None, *_lo(level), **kw)


def _path_ran(enc, level):
    """The forced path really ran (units of 256 KiB and more only fit the HBM path's position field)."""
    if level in (1, "1L"):
        assert enc.ctx().last_path() == ("lds" if level == "1L" else "hbm")


def _check_units(oracle, units, level=1):
    buf, off = corpora.pack_units(units)
    enc = _enc(level)
    out, out_off = enc.EncodeUnits(buf, off)
    ref, ref_off = oracle.zstd_encode_units(buf, off, threads=8, level=_li(level))
    bad = []
    for i in range(len(units)):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        b = ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()
        if a != b:
            bad.append((i, len(units[i]), len(a), len(b)))
    assert not bad, "units differing from the oracle (index, in_len, gpu_len, oracle_len): %r" % bad[:10]
    assert np.array_equal(out_off, ref_off)
    enc.Close()


@pytest.mark.parametrize("level", LEVELS)
def test_parse_matches_oracle(oracle, kclib, level):
    """Intermediate artefact parity: the sequence list of every block equals the oracle's."""
    torch = _torch()
    units = [corpora.corpus("T", 1, 131072, first_unit=k).tobytes() for k in range(4)]
    units += [corpora.corpus("M", 1, 131072, first_unit=k).tobytes() for k in range(4)]
    units += [corpora.corpus("J", 1, 65536, first_unit=3).tobytes(), corpora.corpus("H", 1, 131072).tobytes()]
    units += [u for u in corpora.edge_units() if len(u) > 0]
    buf, off = corpora.pack_units(units)
    d = torch.from_numpy(buf).cuda()
    enc = _enc(level)
    blocks = enc.DebugParseDevice(d.data_ptr(), off)
    bi = 0
    for ui, u in enumerate(units):
        ref = oracle.zstd_parse_unit(u, level=_li(level))
        for rb, (rseqs, rlits) in enumerate(ref):
            gseqs, gextra = blocks[bi]
            bi += 1
            assert len(gseqs) == len(rseqs), "unit %d block %d: nseq %d vs oracle %d" % (ui, rb, len(gseqs), len(rseqs))
            if len(rseqs):
                neq = np.nonzero((gseqs != rseqs).any(axis=1))[0]
                assert len(neq) == 0, "unit %d block %d first differing seq %d: gpu %r oracle %r" % (
                    ui, rb, neq[0], gseqs[neq[0]], rseqs[neq[0]])
    assert bi == len(blocks)
    enc.Close()


@pytest.mark.parametrize("level", LEVELS_B)
def test_edge_units_bit_exact(oracle, kclib, level):
    _torch()
    _check_units(oracle, corpora.edge_units(), level)


@pytest.mark.parametrize("level", LEVELS_B)
@pytest.mark.parametrize("kind", ["T", "H", "J", "M"])
def test_corpus_units_bit_exact(oracle, kclib, kind, level):
    _torch()
    buf = corpora.corpus(kind, 96, 131072)
    units = [buf[i * 131072:(i + 1) * 131072].tobytes() for i in range(96)]
    _check_units(oracle, units, level)


@pytest.mark.parametrize("level", LEVELS_B)
def test_ragged_units_bit_exact(oracle, kclib, level):
    _torch()
    rng = np.random.default_rng(7)
    text = corpora.corpus("T", 8, 131072).tobytes()
    mixed = corpora.corpus("M", 8, 131072).tobytes()
    units = []
    for k in range(200):
        src = text if k % 2 == 0 else mixed
        n = int(rng.integers(0, 300000)) if k % 5 else int(rng.integers(0, 64))
        s = int(rng.integers(0, len(src) - n))
        units.append(src[s:s + n])
    _check_units(oracle, units, level)


def test_xxh64_units(oracle, kclib):
    torch = _torch()
    import ctypes as C
    units = corpora.edge_units()
    buf, off = corpora.pack_units(units)
    d = torch.from_numpy(buf).cuda()
    enc = _enc(1)
    ctx = enc.ctx()
    out = np.zeros(len(units), dtype=np.uint64)
    ctx.check(ctx.L.kc_xxh64_units_dev(ctx.h, d.data_ptr(), off.ctypes.data, len(units), out.ctypes.data))
    for i, u in enumerate(units):
        assert int(out[i]) == oracle.lib().kco_xxh64(u, len(u)), i
    enc.Close()


@pytest.mark.parametrize("level", LEVELS)
def test_device_resident_roundtrip_full_size(oracle, kclib, level):
    """BASELINE-size property check (no oracle at this size): every frame decodes back with libzstd."""
    torch = _torch()
    n, usz = 2048, 131072
    buf = corpora.corpus("T", n, usz)
    off = (np.arange(n + 1, dtype=np.uint64) * usz)
    d_src = torch.from_numpy(buf).cuda()
    enc = _enc(level)
    cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    out_off = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    out = d_dst[:int(out_off[n])].cpu().numpy()
    rng = np.random.default_rng(3)
    for i in rng.choice(n, 64, replace=False):
        frame = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert oracle.zstd_decompress(frame, usz + 16) == buf[i * usz:(i + 1) * usz].tobytes()
    ratio = float(out_off[n]) / float(n * usz)
    assert 0.2 < ratio < 0.7, ratio
    enc.Close()


@pytest.mark.parametrize("level", LEVELS_B)
@pytest.mark.parametrize("dict_id", [0, 1, 70000])
def test_with_raw_dictionary_bit_exact(oracle, kclib, dict_id, level):
    """C5: 64 KiB raw-content dictionary (WithEncoderDictRaw) at every level, mixed corpus.  Units <= 32 KiB take the
    fastEncoderDict small-input variant (kSearchStrength 7) at SpeedFastest."""
    _torch()
    from compress_amd import zstd
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    buf = corpora.corpus("M", 48, 131072)
    units = [buf[i * 131072:(i + 1) * 131072].tobytes() for i in range(48)]
    t = corpora.corpus("T", 8, 131072, first_unit=77).tobytes()
    units += [t[:200000], t[5:40000], t[:9], t[:100], b"", t[100000:400000], dct[:50000], dct]
    units += [t[7:7 + n] for n in (32768, 32769, 20000, 4096, 1000, 17, 65536, 65537, 98304)] + [dct[1000:30000], buf[3000:30000].tobytes()]
    ubuf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, *_lo(level), zstd.WithEncoderDictRaw(dict_id, dct))
    out, out_off = enc.EncodeUnits(ubuf, off)
    ref, ref_off = oracle.zstd_encode_units(ubuf, off, threads=8, level=_li(level), dict_id=dict_id, dict_content=dct)
    bad = [i for i in range(len(units))
           if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()]
    assert not bad, bad[:10]
    if dict_id == 0:  # libzstd treats raw-content dictionaries as ID 0
        for i in (0, 5, 48, 53):
            frame = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
            assert oracle.zstd_decompress(frame, len(units[i]) + 16, dict_content=dct) == units[i]
    enc.Close()


@pytest.mark.parametrize("level", LEVELS_B)
def test_with_a_dictionary_above_one_mib_bit_exact(oracle, kclib, level):
    """Raw dictionaries of 1.5 and 3 MiB (the reference takes any dictionary below 2 GiB as history, zstd/dict.go:27; the window still
    bounds the offsets): units that repeat dictionary content from its start, its middle and its end."""
    _torch()
    from compress_amd import zstd
    big = corpora.corpus("T", 24, 131072, seed=0x5EED0009).tobytes()
    t = corpora.corpus("T", 4, 131072, first_unit=31).tobytes()
    for dl in ((3 << 19), (3 << 20)):
        dct = big[:dl]
        units = [t[:131072], dct[1000:60000] + t[:30000], dct[dl // 2:dl // 2 + 70000], dct[-50000:] + t[5:5000], t[:100], dct[:131072], t[131072:131072 + 200000]]
        ubuf, off = corpora.pack_units(units)
        enc = zstd.NewWriter(None, *_lo(level), zstd.WithEncoderDictRaw(9, dct))
        out, out_off = enc.EncodeUnits(ubuf, off)
        ref, ref_off = oracle.zstd_encode_units(ubuf, off, threads=8, level=_li(level), dict_id=9, dict_content=dct)
        bad = [i for i in range(len(units))
               if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()]
        assert not bad, (dl, bad)
        enc.Close()


@pytest.mark.parametrize("level", LEVELS_B)
@pytest.mark.parametrize("which", ["d0", "skewed"])
def test_with_full_format_dictionary_bit_exact(oracle, kclib, level, which):
    """a16: WithEncoderDict (zstd --train format): dictionary offsets, content as history and the literal Huffman table
    as prevTable of each unit's first block.  'd0' is the reference's own fixture with inputs of its kind; 'skewed' makes
    huff0 actually keep the dictionary table (treeless literals) and exercises encodeLits' 8..31-byte rule."""
    _torch()
    from compress_amd import zstd
    import test_oracle_kats as tk
    blob, ins = tk._dict_fixture(oracle)
    units = list(ins) + [ins[1][:40], ins[1][:20], ins[1][:9], b"", ins[2] + ins[4] + ins[2][:50000]]
    if which == "skewed":
        blob, probs = tk.skewed_dict(blob)
        units += tk.skewed_units(probs, seeds=4)
    units += [corpora.corpus("T", 3, 131072, first_unit=5).tobytes()[:n] for n in (300000, 131072, 65537, 31, 17)]
    ubuf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, *_lo(level), zstd.WithEncoderDict(blob))
    out, out_off = enc.EncodeUnits(ubuf, off)
    ref, ref_off = oracle.zstd_encode_units(ubuf, off, threads=8, level=_li(level), dict_blob=blob)
    bad = [i for i in range(len(units))
           if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()]
    assert not bad, bad[:10]
    for i in (1, 3, len(units) - 5):
        frame = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert oracle.zstd_decompress(frame, len(units[i]) + 16, dict_content=blob) == units[i]
    enc.Close()


@pytest.mark.parametrize("level", LEVELS_B)
def test_stress_mixes_bit_exact(oracle, kclib, level):
    """Adversarial literal/sequence mixes (corpora.stress_units): long literal runs across the LDS gather window,
    more cooperative runs than the per-batch list holds, Huffman-only blocks, RLE blocks, multi-block units."""
    _torch()
    _check_units(oracle, corpora.stress_units(), level=_li(level))


def test_stress_mixes_entropy_options(oracle, kclib):
    """The same mixes under the options that change the entropy stage's decisions."""
    _torch()
    from compress_amd import zstd
    units = corpora.stress_units(seed=11, n=40)
    ubuf, off = corpora.pack_units(units)
    for kw, ops in (({"no_entropy": True}, [zstd.WithNoEntropyCompression(True)]),
                    ({"all_lit_entropy": True}, [zstd.WithAllLitEntropyCompression(True)]),
                    ({"crc": False, "window_size": 1 << 15}, [zstd.WithEncoderCRC(False), zstd.WithWindowSize(1 << 15)])):
        enc = zstd.NewWriter(None, *ops, zstd.WithEncoderLevel(zstd.SpeedFastest))
        out, out_off = enc.EncodeUnits(ubuf, off)
        if "window_size" in kw:
            kw = dict(kw, block_size=1 << 15)
        ref, ref_off = oracle.zstd_encode_units(ubuf, off, threads=8, level=1, **kw)
        bad = [i for i in range(len(units))
               if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()]
        assert not bad, (kw, bad[:10])
        enc.Close()


def test_begin_end_pipeline_two_contexts(oracle, kclib):
    """kc_zstd_encode_units_dev_begin/_end on two chained contexts (the bench's software pipeline) produce the same
    frames as the blocking call, batch after batch."""
    torch = _torch()
    from compress_amd import zstd
    n, usz = 512, 131072
    bufs = [corpora.corpus(k, n, usz, first_unit=17 * j) for j, k in enumerate("TJMT")]
    off = np.arange(n + 1, dtype=np.uint64) * usz
    d_srcs = [torch.from_numpy(b).cuda() for b in bufs]
    ref_enc = _enc(1)
    cap = n * ((ref_enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    refs = []
    d_ref = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for d in d_srcs:
        o = ref_enc.EncodeUnitsDevice(d.data_ptr(), off, d_ref.data_ptr(), cap)
        refs.append((o.copy(), d_ref[:int(o[n])].cpu().numpy().copy()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    encs = [zstd.NewWriter(None, zstd.WithEncoderLevel(1), stream=s.cuda_stream) for s in streams]
    encs[0].ChainAfter(encs[1]); encs[1].ChainAfter(encs[0])
    d_dst = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    encs[0].EncodeUnitsDeviceBegin(d_srcs[0].data_ptr(), off, d_dst[0].data_ptr(), cap)
    for i in range(len(d_srcs)):
        if i + 1 < len(d_srcs):
            encs[(i + 1) % 2].EncodeUnitsDeviceBegin(d_srcs[i + 1].data_ptr(), off, d_dst[(i + 1) % 2].data_ptr(), cap)
        o = encs[i % 2].EncodeUnitsDeviceEnd()
        assert np.array_equal(o, refs[i][0]), i
        assert np.array_equal(d_dst[i % 2][:int(o[n])].cpu().numpy(), refs[i][1]), i
    # misuse is reported, not undefined: a second begin on a busy context, an end without a begin
    encs[0].EncodeUnitsDeviceBegin(d_srcs[0].data_ptr(), off, d_dst[0].data_ptr(), cap)
    with pytest.raises(Exception):
        encs[0].EncodeUnitsDeviceBegin(d_srcs[0].data_ptr(), off, d_dst[0].data_ptr(), cap)
    encs[0].EncodeUnitsDeviceEnd()
    with pytest.raises(Exception):
        encs[0].EncodeUnitsDeviceEnd()
    for e in encs + [ref_enc]:
        e.Close()


@pytest.mark.parametrize("n,usz", [(768, 131072), (40, 16384)])  # (the small shape also runs on the wave emulator under ASan: tools/emu_host_check.sh)
@pytest.mark.parametrize("level", [1, 2, 3])
def test_one_batch_as_parts_on_three_contexts(oracle, kclib, level, n, usz):
    """One EncodeAll batch run as several launches (bench.py --split): the parts go round three contexts chained TWO apart (two
    match finders on the device together) and every part's frames are put right behind the previous part's with
    kc_zstd_encode_units_dev_end_at.  Same offsets and bytes as the one blocking call, pass after pass, with ragged parts."""
    torch = _torch()
    from compress_amd import zstd
    bufs = [corpora.corpus(k, n, usz, first_unit=29 * j) for j, k in enumerate("TM")]
    off = np.arange(n + 1, dtype=np.uint64) * usz
    d_srcs = [torch.from_numpy(b).cuda() for b in bufs]
    ref_enc = _enc(level)
    cap = n * ((ref_enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    d_ref = torch.empty(cap, dtype=torch.uint8, device="cuda")
    refs = []
    for d in d_srcs:
        o = ref_enc.EncodeUnitsDevice(d.data_ptr(), off, d_ref.data_ptr(), cap)
        refs.append((o.copy(), d_ref[:int(o[n])].cpu().numpy().copy()))
    streams = [torch.cuda.Stream() for _ in range(3)]
    encs = [zstd.NewWriter(None, zstd.WithEncoderLevel(level), stream=s.cuda_stream) for s in streams]
    for j in range(3):
        encs[j].ChainAfter(encs[(j - 2) % 3])
    cuts = [0, n * 300 // 768, n * 300 // 768 + 1, n]  # ragged parts, one of a single unit
    parts = [(p, a, b) for p in range(len(d_srcs)) for a, b in zip(cuts[:-1], cuts[1:])]
    d_dst = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(len(d_srcs))]
    torch.cuda.synchronize()

    def begin(g):
        p, a, b = parts[g]
        encs[g % 3].EncodeUnitsDeviceBegin(d_srcs[p].data_ptr(), off[a:b + 1], d_dst[p].data_ptr(), cap)

    begun = 0
    pos = 0
    offs = [0]
    for g in range(len(parts)):
        while begun < len(parts) and begun < g + 3:
            begin(begun)
            begun += 1
        p, a, b = parts[g]
        if a == 0:
            pos, offs = 0, [0]
        o = encs[g % 3].EncodeUnitsDeviceEnd(d_dst[p].data_ptr() + pos, cap - pos)
        assert int(o[0]) == 0
        offs += [pos + int(x) for x in o[1:]]
        pos += int(o[b - a])
        if b == n:
            assert np.array_equal(np.array(offs, dtype=np.uint64), refs[p][0]), (level, p)
            assert np.array_equal(d_dst[p][:pos].cpu().numpy(), refs[p][1]), (level, p)
    # the capacity is checked against the place named at the end
    encs[0].EncodeUnitsDeviceBegin(d_srcs[0].data_ptr(), off[:9], d_dst[0].data_ptr(), cap)
    with pytest.raises(Exception):
        encs[0].EncodeUnitsDeviceEnd(d_dst[0].data_ptr(), 100)
    o = encs[0].EncodeUnitsDeviceEnd()  # the batch is still in flight after the refusal: finished where Begin said
    assert np.array_equal(o, refs[0][0][:9]) and np.array_equal(d_dst[0][:int(o[8])].cpu().numpy(), refs[0][1][:int(o[8])])
    for e in encs + [ref_enc]:
        e.Close()


@pytest.mark.parametrize("level", LEVELS_B)
def test_streams_bit_exact(oracle, kclib, level):
    """N2: NewWriter(w).Write(...) / Close() streams (kc_zstd_encode_streams_dev) against the oracle's restatement of
    Write -> nextBlock -> Close: EncodeAll frame below one block, streaming frame (no content size, history from the first
    block, trailing empty raw block when the input ends on a block boundary) from one block on; both decoders round-trip."""
    _torch()
    import io
    from compress_amd import zstd
    enc = zstd.NewWriter(None, *_lo(level))
    bs = enc.o.block_size
    t = corpora.corpus("T", 6, 131072, first_unit=40).tobytes()
    m = corpora.corpus("M", 2, 131072, first_unit=3).tobytes()
    units = [b"", t[:1], t[:100], t[:bs - 1], t[:bs], t[:bs + 1], t[:2 * bs], t[:2 * bs + 5], t[:3 * bs - 1], m[:bs], m, t[7:7 + 5 * bs], corpora.corpus("H", 1, 2 * bs).tobytes()]
    ubuf, off = corpora.pack_units(units)
    out, out_off = enc.EncodeStreams(ubuf, off)
    ref = oracle.ZstdOracle(level=_li(level))
    for i, u in enumerate(units):
        got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert got == ref.encode_stream(u), (i, len(u))
        if u:
            assert oracle.zstd_decompress(got, len(u) + 16) == u
    # options that change the header / checksum
    e2 = zstd.NewWriter(None, *_lo(level), zstd.WithEncoderCRC(False), zstd.WithZeroFrames(False))
    o2, oo2 = e2.EncodeStreams(ubuf, off)
    r2 = oracle.ZstdOracle(level=_li(level), crc=False, full_zero=False)
    for i, u in enumerate(units):
        assert o2[int(oo2[i]):int(oo2[i + 1])].tobytes() == r2.encode_stream(u), (i, len(u))
    # the io.Writer surface
    sink = io.BytesIO()
    w = zstd.NewWriter(sink, *_lo(level))
    for i in range(0, len(t[:2 * bs + 5]), 50000):
        w.Write(t[i:min(i + 50000, 2 * bs + 5)])
    w.Close()
    assert sink.getvalue() == ref.encode_stream(t[:2 * bs + 5])
    sink2 = io.BytesIO()
    w.Reset(sink2)
    w.ReadFrom(io.BytesIO(t[:1000]))
    w.Close()
    assert sink2.getvalue() == ref.encode_all(t[:1000])
    with pytest.raises(IOError):
        w.Write(b"x")
    enc.Close(); e2.Close()


@pytest.mark.parametrize("level", LEVELS_B)
def test_streams_with_flush_points_bit_exact(oracle, kclib, level):
    """Mid-stream Flush (zstd/encoder.go:547-570) ends the block being filled: kc_zstd_encode_streams_cuts against the oracle's
    Write / Flush / Close restatement — cuts inside the first block (header written early: no EncodeAll frame), on block boundaries
    (no-ops), at the very end (empty last block), repeated, many tiny blocks, and random mixes; the io.Writer surface
    (Write + Flush + ReadFrom + Close) writes the same bytes."""
    _torch()
    import io
    import random
    from compress_amd import zstd
    enc = zstd.NewWriter(None, *_lo(level))
    bs = enc.o.block_size
    t = corpora.corpus("T", 6, 131072, first_unit=90).tobytes()
    m = corpora.corpus("M", 3, 131072, first_unit=5).tobytes()
    cases = [(t[:1000], [10, 500]), (t[:1000], [1000]), (t[:1000], [0]), (t[:1000], []), (t[:bs + 100], [50, bs + 100]),
             (t[:bs], [bs]), (t[:2 * bs], [bs]), (t[:2 * bs + 9], [bs - 1, bs, bs + 1]), (t[:3 * bs], [7, 7, 7, 2 * bs + 7]),
             (m[:bs + 5000], [100 * k for k in range(1, 25)]), (b"", [0]), (b"", []), (t[:5], [1, 2, 3, 4, 5]),
             (corpora.corpus("H", 1, bs).tobytes(), [bs // 2]), (t[:4 * bs + 1], [3 * bs + 50000, 4 * bs + 1, 4 * bs + 9])]
    rnd = random.Random(1234 + _li(level))
    for _ in range(25):
        n = rnd.choice([rnd.randrange(1, 3000), rnd.randrange(bs - 2000, bs + 2000), rnd.randrange(2 * bs, 5 * bs)])
        d = (t if rnd.random() < 0.6 else m)[:n]
        cuts = sorted(rnd.randrange(0, n + 2) for _ in range(rnd.choice([1, 1, 2, 4, 9])))
        cases.append((d, cuts))
    units = [c[0] for c in cases]
    ubuf, off = corpora.pack_units(units)
    out, out_off = enc.EncodeStreams(ubuf, off, flush_at=[c[1] for c in cases])
    ref = oracle.ZstdOracle(level=_li(level))
    for i, (u, cuts) in enumerate(cases):
        got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert got == ref.encode_stream(u, cuts), (i, len(u), cuts)
        if u:
            assert oracle.zstd_decompress(got, len(u) + 16) == u
    # no cuts given at all == the plain stream API
    plain, plain_off = enc.EncodeStreams(ubuf, off)
    none, none_off = enc.EncodeStreams(ubuf, off, flush_at=[[] for _ in cases])
    assert np.array_equal(plain, none) and np.array_equal(plain_off, none_off)
    # the io.Writer surface
    sink = io.BytesIO()
    w = zstd.NewWriter(sink, *_lo(level))
    w.Flush()                      # nothing buffered: no effect
    w.Write(t[:70000]); w.Flush()
    w.Write(t[70000:70010]); w.Flush(); w.Flush()
    w.Write(t[70010:2 * bs + 30])
    w.ReadFrom(io.BytesIO(t[2 * bs + 30:3 * bs]))   # ends the block being filled first
    w.Close()
    assert sink.getvalue() == ref.encode_stream(t[:3 * bs], [70000, 70010, 2 * bs + 30])
    enc.Close()


def _dict_stream_cases(bs, kind):
    """(dictionary option kwargs for the oracle, dictionary bytes for the decoders, streams with their Flush points)."""
    import random
    import test_oracle_kats as tk
    t = corpora.corpus("T", 6, 131072, first_unit=21).tobytes()
    if kind == "raw":
        dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
        okw, extra = dict(dict_id=7, dict_content=dct), [dct[100:40000], dct]
    else:
        blob, ins = tk._dict_fixture(oracle_mod())
        extra = [ins[1], ins[2] + ins[4], ins[3][:50000]]
        if kind == "skewed":  # a literal table huff0 actually keeps: the two nextBlock forms then write different first blocks
            blob, probs = tk.skewed_dict(blob)
            extra += tk.skewed_units(probs, sizes=(40, 300, 1000, 4000, 70000), seeds=2)
        dct, okw = blob, dict(dict_blob=blob)
    cases = [(u, []) for u in extra] + [(u, [len(u) // 3]) for u in extra if len(u) > 30]
    cases += [(t[:bs], []), (t[:bs - 1], []), (t[:bs + 1], []), (t[:3 * bs + 77], []), (t[:1000], [10, 500]), (t[:2 * bs + 9], [bs - 1, bs + 1]),
              (t[:2 * bs], [2 * bs]), (b"", []), (b"", [0]), (t[:20000], [20000])]
    rnd = random.Random(99)
    for _ in range(10):
        n = rnd.choice([rnd.randrange(1, 3000), rnd.randrange(bs - 2000, bs + 2000), rnd.randrange(2 * bs, 4 * bs)])
        cases.append((t[5:5 + n], sorted(rnd.randrange(0, n + 2) for _ in range(rnd.choice([1, 2, 4])))))
    return okw, dct, cases


def oracle_mod():
    import oracle_lib
    return oracle_lib


@pytest.mark.parametrize("level", LEVELS_B)
@pytest.mark.parametrize("kind", ["raw", "d0", "skewed"])
def test_dictionary_streams_bit_exact(oracle, kclib, level, kind):
    """N2: Write / Flush / Close streams of an encoder with a dictionary (zstd/encoder.go:257-428 after Reset(dict)): the frame
    carries the dictionary id, the first block's history is the dictionary content, and — which of nextBlock's two forms wrote
    it decides — the first block's literals start from the dictionary table (default, asynchronous form) or never see it
    (WithEncoderConcurrency(1), synchronous form).  Both against the oracle's restatement, with Flush points."""
    _torch()
    import io
    from compress_amd import zstd
    bs = zstd.NewWriter(None, *_lo(level)).o.block_size
    okw, dct, cases = _dict_stream_cases(bs, kind)
    dopt = zstd.WithEncoderDictRaw(7, dct) if kind == "raw" else zstd.WithEncoderDict(dct)
    ubuf, off = corpora.pack_units([c[0] for c in cases])
    frames = {}
    for conc in (None, 1, 4):
        enc = zstd.NewWriter(None, *_lo(level), dopt, *([zstd.WithEncoderConcurrency(conc)] if conc else []))
        out, out_off = enc.EncodeStreams(ubuf, off, flush_at=[c[1] for c in cases])
        ref = oracle.ZstdOracle(level=_li(level), concurrent=conc or 0, **okw)
        for i, (u, cuts) in enumerate(cases):
            got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
            assert got == ref.encode_stream(u, cuts), (conc, i, len(u), cuts)
        frames[conc] = out.tobytes()
        if conc is None:  # the plain entry point (no cut list) is the same thing
            p_out, p_off = enc.EncodeStreams(ubuf, off)
            for i, (u, _c) in enumerate(cases):
                assert p_out[int(p_off[i]):int(p_off[i + 1])].tobytes() == ref.encode_stream(u), (i, len(u))
            for i in (0, 3, len(cases) - 1):
                u = cases[i][0]
                got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
                if u and kind == "raw":
                    assert oracle.zstd_decode(got, len(u) + 16, dict_content=dct) == u
                elif u:
                    assert oracle.zstd_decompress(got, len(u) + 16, dict_content=dct) == u
        enc.Close()
    assert frames[None] == frames[4]
    if kind == "skewed":
        assert frames[None] != frames[1]  # the dictionary literal table reached some first block
    else:
        assert kind != "raw" or frames[None] == frames[1]  # no literal table in a raw-content dictionary
    # the io.Writer face
    t = cases[-1][0] + cases[0][0]
    sink = io.BytesIO()
    w = zstd.NewWriter(sink, *_lo(level), dopt)
    w.Write(t[:700]); w.Flush(); w.Write(t[700:]); w.Close()
    assert sink.getvalue() == oracle.ZstdOracle(level=_li(level), **okw).encode_stream(t, [700])


@pytest.mark.parametrize("level", LEVELS)
def test_device_decoder_roundtrip(oracle, kclib, level):
    """N1 (GPU half) for zstd: kc_zstd_decode_units_dev decodes the frames the device encoder produced back to the source,
    on the device, checksum included — every corpus kind, edge units, adversarial mixes (raw / RLE / compressed blocks,
    raw / RLE / 1-stream / 4-stream / treeless literals, predefined / RLE / FSE / repeat sequence tables)."""
    torch = _torch()
    enc = _enc(level)
    units = corpora.edge_units() + corpora.stress_units(seed=3, n=24)
    for k in "THJM":
        b = corpora.corpus(k, 6, 131072, first_unit=9)
        units += [b[i * 131072:(i + 1) * 131072].tobytes() for i in range(6)]
    ubuf, off = corpora.pack_units(units)
    n = len(units)
    d_src = torch.from_numpy(ubuf).cuda()
    cap = sum(((enc.MaxEncodedSize(len(u)) + 15) & ~15) for u in units) + 64
    d_enc = torch.empty(cap, dtype=torch.uint8, device="cuda")
    eoff = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_enc.data_ptr(), cap)
    d_out = torch.zeros(len(ubuf) + 64, dtype=torch.uint8, device="cuda")
    st = enc.DecodeUnitsDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), off)
    bad = np.nonzero(st)[0]
    assert len(bad) == 0, [(int(i), int(st[i]), len(units[i])) for i in bad[:8]]
    assert torch.equal(d_out[:len(ubuf)], d_src)
    # corruption is detected per unit: a flipped payload byte (checksum or format error), a truncated frame, a wrong size
    frames = d_enc[:int(eoff[n])].cpu().numpy()
    pick = [i for i in range(n) if len(units[i]) >= 65536][:3]
    parts, sizes = [], []
    for j, i in enumerate(pick):
        f = frames[int(eoff[i]):int(eoff[i + 1])].copy()
        if j == 0:
            f[len(f) // 2] ^= 0x10
        elif j == 1:
            f = f[:-5]
        parts.append(f)
        sizes.append(len(units[i]) + (1 if j == 2 else 0))
    e2 = np.zeros(len(pick) + 1, dtype=np.uint64); e2[1:] = np.cumsum([len(p) for p in parts])
    d2 = np.zeros(len(pick) + 1, dtype=np.uint64); d2[1:] = np.cumsum(sizes)
    d_bad = torch.from_numpy(np.concatenate(parts)).cuda()
    d_o2 = torch.zeros(int(d2[-1]) + 64, dtype=torch.uint8, device="cuda")
    st2 = enc.DecodeUnitsDevice(d_bad.data_ptr(), e2, d_o2.data_ptr(), d2)
    assert all(int(x) != 0 for x in st2), st2
    enc.Close()


def test_device_decoder_on_foreign_frames(oracle, kclib):
    """Frames the device encoder did not write: streaming frames (no content size, window descriptor, trailing empty block)
    from the oracle, and no-checksum / multi-block frames."""
    torch = _torch()
    enc = _enc(1)
    t = corpora.corpus("T", 5, 131072, first_unit=70).tobytes()
    srcs, frames = [], []
    for lvl, kw in ((1, {}), (2, {}), (3, {}), (1, {"crc": False}), (2, {"no_entropy": True})):
        o = oracle.ZstdOracle(level=lvl, **kw)
        for n in (0, 1, 1000, 65536, 131072, 300000):
            srcs.append(t[:n]); frames.append(o.encode_stream(t[:n]))
            if n:
                srcs.append(t[7:7 + n]); frames.append(o.encode_all(t[7:7 + n]))
    eoff = np.zeros(len(frames) + 1, dtype=np.uint64); eoff[1:] = np.cumsum([len(f) for f in frames])
    doff = np.zeros(len(frames) + 1, dtype=np.uint64); doff[1:] = np.cumsum([len(s) for s in srcs])
    d_enc = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).cuda()
    d_out = torch.zeros(int(doff[-1]) + 64, dtype=torch.uint8, device="cuda")
    st = enc.DecodeUnitsDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), doff)
    bad = np.nonzero(st)[0]
    assert len(bad) == 0, [(int(i), int(st[i]), len(srcs[i])) for i in bad[:8]]
    assert d_out[:int(doff[-1])].cpu().numpy().tobytes() == b"".join(srcs)
    enc.Close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_randomized_options_bit_exact(oracle, kclib, seed):
    """Differential test over the option space: random level x CRC x entropy switches x single segment x window (hence block
    size) x raw dictionary x EncodeAll / streams, on spliced adversarial units; every frame must equal the oracle's."""
    _torch()
    import random
    from compress_amd import zstd
    rnd = random.Random(seed)
    pool = corpora.stress_units(seed=20 + seed, n=30) + corpora.edge_units()
    for it in range(10):
        lvl = rnd.choice([1, 2, 3])
        ops = [zstd.WithEncoderLevel(lvl)]
        kw = dict(level=lvl)
        if rnd.random() < 0.4:
            kw["crc"] = False; ops.append(zstd.WithEncoderCRC(False))
        if rnd.random() < 0.2:
            kw["no_entropy"] = True; ops.append(zstd.WithNoEntropyCompression(True))
        if rnd.random() < 0.3:
            v = rnd.random() < 0.5
            kw["all_lit_entropy"] = v; ops.append(zstd.WithAllLitEntropyCompression(v))
        if rnd.random() < 0.3:
            v = rnd.random() < 0.5
            kw["single"] = v; ops.append(zstd.WithSingleSegment(v))
        if rnd.random() < 0.2:
            kw["full_zero"] = False; ops.append(zstd.WithZeroFrames(False))
        bs = (1 << 16) if lvl == 1 else (128 << 10)
        if rnd.random() < 0.3:
            ws = 1 << rnd.choice([15, 16, 17, 20])
            # WithWindowSize before the level: the level then keeps the custom window AND the block size derived from the
            # default 128 KiB (encoder_options.go:110-133, 236-266), also at SpeedFastest
            bs = min(ws, 128 << 10)
            kw["window_size"] = ws; kw["block_size"] = bs
            ops.insert(0, zstd.WithWindowSize(ws))
        units = [u for u in rnd.sample(pool, 14) if len(u) <= 32 * bs]
        stream = rnd.random() < 0.35
        dct = None
        if not stream and rnd.random() < 0.35:
            dct = rnd.choice(pool)[:rnd.choice([300, 20000, 65536])] or b"dictionary"
            did = rnd.choice([0, 9, 70000])
            kw["dict_id"] = did; kw["dict_content"] = dct
            ops.append(zstd.WithEncoderDictRaw(did, dct))
        ubuf, off = corpora.pack_units(units)
        enc = zstd.NewWriter(None, *ops)
        assert (enc.o.window_size, enc.o.block_size) == (kw.get("window_size", enc.o.window_size), kw.get("block_size", enc.o.block_size))
        out, out_off = (enc.EncodeStreams if stream else enc.EncodeUnits)(ubuf, off)
        ref = oracle.ZstdOracle(**kw)
        for i, u in enumerate(units):
            want = ref.encode_stream(u) if stream else ref.encode_all(u)
            got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
            assert got == want, (seed, it, sorted(kw), stream, i, len(u), len(got), len(want))
        enc.Close()


@pytest.mark.parametrize("level", LEVELS)
def test_device_decoder_with_dictionary(oracle, kclib, level):
    """The device verifier with a raw-content dictionary as history (C5's configuration): frames written by the device encoder
    with WithEncoderDictRaw decode back to the source; without the dictionary they are refused (status 20), not mis-decoded."""
    torch = _torch()
    from compress_amd import zstd
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    buf = corpora.corpus("M", 24, 131072)
    t = corpora.corpus("T", 4, 131072, first_unit=77).tobytes()
    units = [buf[i * 131072:(i + 1) * 131072].tobytes() for i in range(24)] + [t[:200000], t[5:40000], t[:9], b"", dct[:50000], dct, dct[100:300] * 50]
    ubuf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, *_lo(level), zstd.WithEncoderDictRaw(7, dct))
    d_src = torch.from_numpy(ubuf).cuda()
    cap = sum(((enc.MaxEncodedSize(len(u)) + 15) & ~15) for u in units) + 64
    d_enc = torch.empty(cap, dtype=torch.uint8, device="cuda")
    eoff = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_enc.data_ptr(), cap)
    d_out = torch.zeros(len(ubuf) + 64, dtype=torch.uint8, device="cuda")
    st = enc.DecodeUnitsDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), off, dict_content=dct)
    bad = np.nonzero(st)[0]
    assert len(bad) == 0, [(int(i), int(st[i]), len(units[i])) for i in bad[:8]]
    assert torch.equal(d_out[:len(ubuf)], d_src)
    st2 = enc.DecodeUnitsDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), off)
    assert all(int(x) == 20 for i, x in enumerate(st2) if len(units[i]) > 0), st2
    enc.Close()


@pytest.mark.parametrize("level", [1, "1L", 2])
def test_host_pipeline_equals_device_path(oracle, kclib, level, monkeypatch):
    """kc_zstd_encode_units above two sub-batches runs the pinned three-stage pipeline (stager / encode / drainer threads):
    its bytes must equal the device-resident path's, which the oracle pins on a sample."""
    torch = _torch()
    monkeypatch.setenv("KC_HOST_PIPE_MIB", "16")  # 16 MiB sub-batches: 96 MiB of input -> 6 stages in flight
    n, usz = 768, 131072
    buf = corpora.corpus("T", n, usz)
    buf[5 * usz:9 * usz] = corpora.corpus("H", 4, usz)  # raw blocks in the middle of a sub-batch
    sizes = np.full(n, usz, dtype=np.uint64)
    sizes[::7] = 100000  # ragged units: offsets that are not multiples of anything
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    buf = buf[:int(off[n])]
    enc = _enc(level)
    out, out_off = enc.EncodeUnits(buf, off)
    d_src = torch.from_numpy(buf).cuda()
    cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    dev_off = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert np.array_equal(out_off, dev_off)
    assert np.array_equal(out, d_dst[:int(dev_off[n])].cpu().numpy())
    ref, ref_off = oracle.zstd_encode_units(buf[:int(off[64])], off[:65], threads=8, level=_li(level))
    assert np.array_equal(out[:int(out_off[64])], ref) and np.array_equal(out_off[:65], ref_off)
    from compress_amd import _lib
    enc.ctx().set_option(_lib.OPT_HOST_SERIAL, 1)
    out2, out_off2 = enc.EncodeUnits(buf, off)
    assert np.array_equal(out2, out) and np.array_equal(out_off2, out_off)
    enc.Close()


@pytest.mark.parametrize("level", [1, "1L", 2])
def test_host_chunk_fed_batch_equals_device_path(oracle, kclib, level, monkeypatch):
    """kc_zstd_encode_units on one device batch: the source arrives in chunks, each chunk's checksum / match finder / entropy
    stage / compaction run on the chunk's own stream behind its copy and the frames drain chunk by chunk.  Same bytes and offsets
    as the device-resident path (chunk boundaries fall inside the ragged part; empty and tiny units included)."""
    torch = _torch()
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL", "0")  # (round 6: large calls take the rolling pipeline; this test keeps the one-batch chunk-fed path covered)
    monkeypatch.setenv("KC_HOST_CHUNKS_MIB", "1,3,8,16")
    n, usz = 512, 131072
    buf = corpora.corpus("T", n, usz)
    buf[5 * usz:9 * usz] = corpora.corpus("H", 4, usz)
    buf[40 * usz:56 * usz] = corpora.corpus("M", 16, usz)
    sizes = np.full(n, usz, dtype=np.uint64)
    sizes[::7] = 100000
    sizes[3] = 0
    sizes[11] = 5
    sizes[n - 1] = 17
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    buf = buf[:int(off[n])]
    enc = _enc(level)
    out, out_off = enc.EncodeUnits(buf, off)
    d_src = torch.from_numpy(buf).cuda()
    cap = sum(((enc.MaxEncodedSize(int(x)) + 15) & ~15) for x in sizes) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    dev_off = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert np.array_equal(out_off, dev_off)
    assert np.array_equal(out, d_dst[:int(dev_off[n])].cpu().numpy())
    ref, ref_off = oracle.zstd_encode_units(buf[:int(off[64])], off[:65], threads=8, level=_li(level))
    assert np.array_equal(out[:int(out_off[64])], ref) and np.array_equal(out_off[:65], ref_off)
    out3, out_off3 = enc.EncodeUnits(buf, off)  # buffers and events reused
    assert np.array_equal(out3, out) and np.array_equal(out_off3, out_off)
    from compress_amd import _lib
    enc.ctx().set_option(_lib.OPT_TEST_FEED_REDO, 1)  # a unit needing the speculation re-run: the batch is encoded again the plain way
    out4, out_off4 = enc.EncodeUnits(buf, off)
    assert np.array_equal(out4, out) and np.array_equal(out_off4, out_off)
    enc.Close()


@pytest.mark.parametrize("level", LEVELS_B)
def test_long_units_and_streams_bit_exact(oracle, kclib, level):
    """Units and streams of more than 32 blocks (the re-run bookkeeping is per block, not a 32-bit mask per unit), longer than the
    window (matches beyond it are refused exactly like the reference's, whose history buffer has slid by then), with a small
    window and small blocks as well; EncodeAll, Write/Close and Write/Flush/Close against the oracle, and the device decoder reads
    the frames back."""
    torch = _torch()
    from compress_amd import zstd
    t = corpora.corpus("T", 80, 131072, first_unit=300).tobytes()
    m = corpora.corpus("M", 48, 131072, first_unit=30).tobytes()
    for opts, okw in (((), {}), ((zstd.WithWindowSize(1 << 16),), {"window_size": 1 << 16})):
        enc = zstd.NewWriter(None, *_lo(level), *opts)
        ref = oracle.ZstdOracle(level=_li(level), **okw)
        bs = enc.o.block_size
        units = [t[:33 * bs], t[5:5 + 70 * bs + 123], m[:40 * bs + 1], (t[:3 * bs] + m[:2 * bs]) * 8]
        ubuf, off = corpora.pack_units(units)
        out, out_off = enc.EncodeUnits(ubuf, off)
        for i, u in enumerate(units):
            assert out[int(out_off[i]):int(out_off[i + 1])].tobytes() == ref.encode_all(u), ("EncodeAll", i, len(u))
        so, so_off = enc.EncodeStreams(ubuf, off)
        for i, u in enumerate(units):
            assert so[int(so_off[i]):int(so_off[i + 1])].tobytes() == ref.encode_stream(u), ("stream", i, len(u))
        cuts = [[bs // 2, 20 * bs + 7], [], [40 * bs + 1], [k * bs * 3 + k for k in range(1, 12)]]
        fo, fo_off = enc.EncodeStreams(ubuf, off, flush_at=cuts)
        for i, u in enumerate(units):
            assert fo[int(fo_off[i]):int(fo_off[i + 1])].tobytes() == ref.encode_stream(u, cuts[i]), ("flush", i, len(u))
        # device verifier on the EncodeAll frames
        d_enc = torch.from_numpy(np.ascontiguousarray(out)).cuda()
        d_out = torch.zeros(len(ubuf) + 64, dtype=torch.uint8, device="cuda")
        st = enc.DecodeUnitsDevice(d_enc.data_ptr(), out_off, d_out.data_ptr(), off)
        assert not np.any(st), st
        assert np.array_equal(d_out[:len(ubuf)].cpu().numpy(), ubuf)
        enc.Close()


@pytest.mark.parametrize("with_dict", [False, True])
def test_better_level_epoch_stamped_tables_over_many_batches(oracle, kclib, with_dict):
    """SpeedBetterCompression keeps its 4 MiB-per-unit tables between batches and tells old entries from new by an epoch stamp
    instead of clearing them: twenty batches on one context — more than the stamp's 15 values (wrap: the arena is cleared), growing
    and shrinking unit counts, unit lengths that change the position width — each bit-exact.  With a dictionary the stamped form
    reads the shared dictionary table for buckets the unit has not written (KC_OPT_BETTER_DICT_EPOCH; off by default: measured
    slower than copying the dictionary tables)."""
    _torch()
    from compress_amd import zstd
    t = corpora.corpus("T", 40, 131072, first_unit=4).tobytes()
    m = corpora.corpus("M", 40, 131072, first_unit=9).tobytes()
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    opts = [zstd.WithEncoderLevel(3)] + ([zstd.WithEncoderDictRaw(3, dct)] if with_dict else [])
    enc = zstd.NewWriter(None, *opts)
    if with_dict:
        enc.ctx().set_option(21, 1)
    okw = dict(dict_id=3, dict_content=dct) if with_dict else {}
    shapes = [(8, 131072), (40, 131072), (3, 131072), (12, 40000), (5, 400000), (40, 131072), (1, 700)] * 3
    for k, (n, ln) in enumerate(shapes[:20]):
        src = t if k % 2 == 0 else m
        units = [src[(i * 7919 + k * 131) % (len(src) - ln):][:ln] for i in range(n)]
        ubuf, off = corpora.pack_units(units)
        out, out_off = enc.EncodeUnits(ubuf, off)
        ref, ref_off = oracle.zstd_encode_units(ubuf, off, threads=8, level=3, **okw)
        assert np.array_equal(out_off, ref_off) and np.array_equal(out, np.asarray(ref)), (k, n, ln)
    enc.Close()


def test_fastest_epoch_stamped_tables_over_many_batches(oracle, kclib):
    """SpeedFastest, HBM-table kernel: the 128 KiB-per-unit tables are not cleared per batch; entries carry the launch's stamp
    (KC_OPT_ZFAST_EPOCH, default on).  Twenty-odd batches on one context — past the stamp's 15 values (wrap: the arena is cleared),
    growing and shrinking unit counts, unit lengths that change the position width, one batch in between with the stamps switched
    off (the arena changes hands and must be cleared before the next stamped batch) — each bit-exact.  The kernel's form is left to the
    context (KC_OPT_ZFAST_VARIANT -1): a batch behind one that did not compress runs the form for input without matches."""
    _torch()
    t = corpora.corpus("T", 40, 131072, first_unit=4).tobytes()
    m = corpora.corpus("M", 40, 131072, first_unit=9).tobytes()
    h = corpora.corpus("H", 8, 131072, first_unit=2).tobytes()
    enc = _enc(1)
    assert enc.ctx().get_option(22) == 1
    shapes = [(8, 131072), (40, 131072), (3, 131072), (12, 40000), (5, 400000), (40, 131072), (1, 700), (9, 131072)] * 3
    for k, (n, ln) in enumerate(shapes[:22]):
        src = (t, m, h)[k % 3]
        units = [src[(i * 7919 + k * 131) % (len(src) - ln):][:ln] for i in range(n)]
        ubuf, off = corpora.pack_units(units)
        if k == 6:
            enc.ctx().set_option(22, 0)
        out, out_off = enc.EncodeUnits(ubuf, off)
        if k == 6:
            enc.ctx().set_option(22, 1)
        _path_ran(enc, 1)
        ref, ref_off = oracle.zstd_encode_units(ubuf, off, threads=8, level=1)
        assert np.array_equal(out_off, ref_off) and np.array_equal(out, np.asarray(ref)), (k, n, ln)
    enc.Close()


@pytest.mark.parametrize("level", [1, 2, 3])
def test_second_stage_on_a_stream_of_its_own(oracle, kclib, level):
    """KC_OPT_STAGE2_STREAM: the entropy stage and everything behind it (size scan, checksum, compaction, the speculation re-run) on a
    second stream behind an event of the match finder's — what a caller does who gives the two stages different CU masks
    (tools/cu_mask_probe.py).  Same bytes as on one stream, over several batches, and the option can be taken back."""
    torch = _torch()
    from compress_amd import _lib
    t = corpora.corpus("T", 24, 131072, first_unit=6).tobytes()
    m = corpora.corpus("M", 24, 131072, first_unit=3).tobytes()
    enc = _enc(level)
    s2 = torch.cuda.Stream()
    enc.ctx().set_option(_lib.OPT_STAGE2_STREAM, s2.cuda_stream)
    assert enc.ctx().get_option(_lib.OPT_STAGE2_STREAM) == s2.cuda_stream
    for k, (src, n, ln) in enumerate([(t, 24, 131072), (m, 17, 100000), (t, 5, 300000), (m, 24, 131072)]):
        units = [src[(i * 7919 + k * 131) % (len(src) - ln):][:ln] for i in range(n)]
        ubuf, off = corpora.pack_units(units)
        if k == 3:
            enc.ctx().set_option(_lib.OPT_STAGE2_STREAM, 0)
        out, out_off = enc.EncodeUnits(ubuf, off)
        ref, ref_off = oracle.zstd_encode_units(ubuf, off, threads=8, level=_li(level))
        assert np.array_equal(out_off, ref_off) and np.array_equal(out, np.asarray(ref)), (k, n, ln)
    enc.Close()


@pytest.mark.parametrize("variant,xseg,filt", [(1, 0, 1), (1, 2, 1), (1, 1 << 20, 1), (1, 0, 0), (0, 0, 1)])
def test_probe_rounds_across_skip_segments(oracle, kclib, variant, xseg, filt):
    """SpeedFastest, HBM-table kernel: a probe round follows the reference's position recurrence across skip-segment boundaries
    (KC_OPT_ZFAST_XSEG_K: always, once the step has grown, never = round 2's rounds) — the same bytes every way, on text, mixed,
    high-entropy and edge inputs, with and without history; with and without the empty-group filter (KC_OPT_ZFAST_FILTER)."""
    _torch()
    units = [corpora.corpus(k, 1, 131072, first_unit=f).tobytes() for k, f in (("T", 1), ("H", 0), ("H", 5), ("M", 2), ("M", 3), ("J", 4))]
    units += [corpora.corpus("H", 4, 131072, first_unit=9).tobytes()[:300001], corpora.corpus("M", 8, 131072, first_unit=1).tobytes()]
    units += [u for u in corpora.edge_units() if len(u) < 200000] + corpora.stress_units(seed=11, n=10)
    buf, off = corpora.pack_units(units)
    enc = _enc(1)
    enc.ctx().set_option(27, variant)  # the kernel's compiled form: 1 = cross-segment rounds + filter, 0 = plain (default: chosen per batch)
    enc.ctx().set_option(23, xseg)
    enc.ctx().set_option(25, filt)  # the "nothing written there yet" filter of units without a sequence so far
    out, out_off = enc.EncodeUnits(buf, off)
    _path_ran(enc, 1)
    ref, ref_off = oracle.zstd_encode_units(buf, off, threads=8, level=1)
    assert np.array_equal(out_off, ref_off) and np.array_equal(out, np.asarray(ref))
    enc.Close()


@pytest.mark.parametrize("level", [1, "1L"])
@pytest.mark.parametrize("variant", [0, 1])
def test_no_match_prescan_settles_incompressible_units(oracle, kclib, level, variant):
    """KC_OPT_ZFAST_PRESCAN (kc_zstd_prescan.hip): units whose probe inserts never repeat a (bucket, 4 bytes) pair are settled
    before the match finder — their frames are raw blocks, written by the pre-scan, hashed and copied by kc_xxh64_fin_kernel — and
    every other unit goes the regular way; the output is the oracle's either way.  High-entropy units of one to three blocks, ragged,
    tiny and empty ones, text and mixed units, high-entropy units with a planted repetition at probed positions; with the default
    and a small window, and (checksum off: the deferred-payload path is off) with the pre-scan not eligible."""
    _torch()
    from compress_amd import zstd
    h = corpora.corpus("H", 40, 131072, first_unit=3).tobytes()
    t = corpora.corpus("T", 4, 131072, first_unit=8).tobytes()
    m = corpora.corpus("M", 4, 131072, first_unit=5).tobytes()
    planted = bytearray(h[:131072]); planted[2:10] = planted[0:8]
    planted_b = bytearray(h[131072:262144]); planted_b[65536 + 2:65536 + 10] = planted_b[0:8]   # block 1 repeats a probed 8 bytes of block 0
    units = [h[:131072], h[7:7 + 131071], h[100:100 + 65536 + 13], h[:196608], h[5:305], h[9:40], h[3:4], b"", bytes(planted), bytes(planted_b),
             t[:131072], h[:65536] + t[:65536], t[:65536] + h[:65536], m[:131072], h[13:13 + 9], h[17:17 + 10], h[19:19 + 65545]]
    units += [h[i * 131072:(i + 1) * 131072] for i in range(2, 34)]
    buf, off = corpora.pack_units(units)
    for opts, okw, eligible in (((), {}, True), ((zstd.WithWindowSize(1 << 16),), {"window_size": 1 << 16}, True), ((zstd.WithEncoderCRC(False),), {"crc": False}, False)):
        enc = zstd.NewWriter(None, *_lo(level), *opts)
        enc.ctx().set_option(27, variant)
        enc.ctx().set_option(28, 1)
        out, out_off = enc.EncodeUnits(buf, off)
        ref, ref_off = oracle.zstd_encode_units(buf, off, threads=8, level=1, **okw)
        bad = [i for i in range(len(units)) if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()]
        assert not bad and np.array_equal(out_off, ref_off), (okw, bad)
        settled = enc.ctx().get_option(102)
        assert (settled >= 38) if eligible else (settled == 0), (okw, settled)
        assert settled <= len(units) - 7  # text, mixed, planted and empty units are never settled
        enc.Close()
    # the default: the pre-scan follows the context's "previous batch did not compress" signal
    enc = zstd.NewWriter(None, *_lo(level))
    hb, hoff = corpora.pack_units([h[i * 131072:(i + 1) * 131072] for i in range(16)])
    for k in range(3):
        out, out_off = enc.EncodeUnits(hb, hoff)
        ref, ref_off = oracle.zstd_encode_units(hb, hoff, threads=8, level=1)
        assert np.array_equal(out_off, ref_off) and np.array_equal(out, np.asarray(ref))
        assert enc.ctx().get_option(102) == (0 if k == 0 else 16), k
    enc.Close()


@pytest.mark.parametrize("fuse,mode", [(1, 0), (1, 1), (1, 2), (1, 3), (0, 0)])
@pytest.mark.parametrize("level", [1, "1L", 2, 3])
def test_raw_only_frames_checksum_and_copy_in_one_pass(oracle, kclib, level, fuse, mode):
    """Frames that end up as raw blocks only get their payload copied by the kernel that hashes it, behind the entropy stage
    (kc_xxh64_fin_kernel, KC_OPT_FUSE_RAW_XXH); every other frame gets its checksum field from the same kernel.  Incompressible
    units of one and several blocks, ragged lengths, exact block multiples, tiny and empty units, units whose blocks are raw and
    compressed in turn, next to compressible ones; with the default and with a small window (64 KiB blocks); without the
    checksum."""
    _torch()
    from compress_amd import zstd
    h = corpora.corpus("H", 12, 131072, first_unit=3).tobytes()
    t = corpora.corpus("T", 4, 131072, first_unit=8).tobytes()
    units = [h[:131072], h[7:7 + 131071], h[100:100 + 65536 + 13], h[:400000], h[131072:131072 + 262144], h[5:305], h[9:40], h[3:4], b"",
             h[:1 << 20], h[:131072] + t[:131072], t[:131072] + h[:131072] + t[:50000], t[:131072], h[1:1 + 131072 + 255], h[2:2 + 256]]
    buf, off = corpora.pack_units(units)
    for opts, okw in (((), {}), ((zstd.WithWindowSize(1 << 16),), {"window_size": 1 << 16}), ((zstd.WithEncoderCRC(False),), {"crc": False})):
        enc = zstd.NewWriter(None, *_lo(level), *opts)
        enc.ctx().set_option(24, fuse)
        enc.ctx().set_option(26, mode)  # the kernel's three store schedules (KC_OPT_XXH_FIN_MODE)
        out, out_off = enc.EncodeUnits(buf, off)
        ref, ref_off = oracle.zstd_encode_units(buf, off, threads=8, level=_li(level), **okw)
        bad = [i for i in range(len(units)) if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()]
        assert not bad and np.array_equal(out_off, ref_off), (okw, bad)
        so, so_off = enc.EncodeStreams(buf, off)  # Write ... Close: stream frames (an empty last block behind exact block multiples)
        sref = oracle.ZstdOracle(level=_li(level), **okw)
        for i, u in enumerate(units):
            assert so[int(so_off[i]):int(so_off[i + 1])].tobytes() == sref.encode_stream(u), ("stream", okw, i, len(u))
        enc.Close()


def test_submit_wait_two_contexts_overlap(oracle, kclib, monkeypatch):
    """kc_zstd_encode_units_submit / kc_wait: two contexts alternate over six batches, each call running its chunk-fed host path on
    its own thread while the other context's call is in flight; the frames equal the synchronous call's, a second submit on a
    busy context and a wait without a job are refused."""
    _torch()
    from compress_amd import zstd, _lib
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL", "0")
    monkeypatch.setenv("KC_HOST_CHUNKS_MIB", "2")
    n, usz = 128, 131072
    batches = [corpora.corpus("TMJ"[k % 3], n, usz, first_unit=1000 * k) for k in range(6)]
    off = np.arange(n + 1, dtype=np.uint64) * usz
    sync = _enc(1)
    want = [sync.EncodeUnits(b, off) for b in batches]
    encs = [_enc(1), _enc(1)]
    got = [None] * 6
    encs[0].EncodeUnitsSubmit(batches[0], off)
    with pytest.raises(_lib.KcError):
        encs[0].EncodeUnitsSubmit(batches[1], off)
    for k in range(1, 7):
        if k < 6:
            encs[k & 1].EncodeUnitsSubmit(batches[k], off)
        got[k - 1] = encs[(k - 1) & 1].Wait()
    for k in range(6):
        assert np.array_equal(got[k][1], want[k][1]) and np.array_equal(got[k][0], want[k][0]), k
    with pytest.raises(_lib.KcError):
        ctx = encs[0].ctx()
        ctx.check(ctx.L.kc_wait(ctx.h))
    for e in encs + [sync]:
        e.Close()


@pytest.mark.parametrize("level", [1, 2, 3])
def test_host_rolling_pipeline_equals_device_path(oracle, kclib, level, monkeypatch):
    """Round 6: kc_zstd_encode_units on a large input goes through the device's rolling pipeline (kc_roll.cpp): sub-batches staged,
    encoded on the engine's four lanes and drained in arrival order.  Same bytes and offsets as the device-resident path (sub-batch
    boundaries inside the ragged part, empty and tiny units, raw blocks), again on a second call, with a raw dictionary, and with a
    destination that is too small (refused, no crash, the engine serves the next call)."""
    torch = _torch()
    from compress_amd import _lib, zstd
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL_MIB", "5")  # 5 MiB sub-batches: ~11 of them in flight through 6 slots / 4 lanes
    n, usz = 448, 131072
    buf = corpora.corpus("T", n, usz)
    buf[5 * usz:9 * usz] = corpora.corpus("H", 4, usz)
    buf[40 * usz:56 * usz] = corpora.corpus("M", 16, usz)
    sizes = np.full(n, usz, dtype=np.uint64)
    sizes[::7] = 100000
    sizes[3] = 0
    sizes[11] = 5
    sizes[n - 1] = 17
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    buf = buf[:int(off[n])]
    enc = _enc(level)
    out, out_off = enc.EncodeUnits(buf, off)
    assert enc.ctx().get_option(_lib.OPT_LAST_BATCHES) >= 8  # the call really was cut into sub-batches
    d_src = torch.from_numpy(buf).cuda()
    cap = sum(((enc.MaxEncodedSize(int(x)) + 15) & ~15) for x in sizes) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    dev_off = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert np.array_equal(out_off, dev_off)
    assert np.array_equal(out, d_dst[:int(dev_off[n])].cpu().numpy())
    ref, ref_off = oracle.zstd_encode_units(buf[:int(off[64])], off[:65], threads=8, level=_li(level))
    assert np.array_equal(out[:int(out_off[64])], ref) and np.array_equal(out_off[:65], ref_off)
    out3, out_off3 = enc.EncodeUnits(buf, off)  # slots, lanes and rings reused
    assert np.array_equal(out3, out) and np.array_equal(out_off3, out_off)
    # a destination one byte short: KC_ERR_DST_TOO_SMALL from the drainer, and the engine is fine afterwards
    ctx = enc.ctx()
    import ctypes as C
    small = np.zeros(int(out_off[n]) - 1, dtype=np.uint8)
    eo = np.zeros(n + 1, dtype=np.uint64)
    st = ctx.L.kc_zstd_encode_units(ctx.h, C.byref(enc.o), buf.ctypes.data, off.ctypes.data, n, small.ctypes.data, small.size, eo.ctypes.data)
    assert st == _lib.KC_ERR_DST_TOO_SMALL
    out4, out_off4 = enc.EncodeUnits(buf, off)
    assert np.array_equal(out4, out) and np.array_equal(out_off4, out_off)
    enc.Close()
    # with a raw dictionary (every lane stages the dictionary itself)
    d = corpora.corpus("T", 1, 65536, first_unit=77).tobytes()
    encd = zstd.NewWriter(None, *_lo(level), zstd.WithEncoderDictRaw(1, d))
    outd, outd_off = encd.EncodeUnits(buf, off)
    refd, refd_off = oracle.zstd_encode_units(buf[:int(off[48])], off[:49], threads=8, level=_li(level), dict_id=1, dict_content=d)
    assert np.array_equal(outd[:int(outd_off[48])], refd) and np.array_equal(outd_off[:49], refd_off)
    dd_off = encd.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert np.array_equal(outd_off, dd_off) and np.array_equal(outd, d_dst[:int(dd_off[n])].cpu().numpy())
    encd.Close()


def test_host_rolling_pipeline_two_callers(oracle, kclib, monkeypatch):
    """Two contexts keep calls in flight through submit / wait: their sub-batches interleave in the ONE engine of the device (arrival
    order), every call gets its own frames back."""
    _torch()
    from compress_amd import _lib
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL_MIB", "3")
    n, usz = 128, 131072
    batches = [corpora.corpus("TMJ"[k % 3], n, usz, first_unit=1000 * k) for k in range(6)]
    off = np.arange(n + 1, dtype=np.uint64) * usz
    sync = _enc(1)
    sync.ctx().set_option(_lib.OPT_HOST_SERIAL, 1)
    want = [sync.EncodeUnits(b, off) for b in batches]
    encs = [_enc(1), _enc(2)]
    want2 = [encs[1].EncodeUnits(b, off) for b in batches]
    got = [None] * 6
    encs[0].EncodeUnitsSubmit(batches[0], off)
    for k in range(1, 7):
        if k < 6:
            encs[k & 1].EncodeUnitsSubmit(batches[k], off)
        got[k - 1] = encs[(k - 1) & 1].Wait()
    for k in range(6):
        w = want[k] if (k & 1) == 0 else want2[k]
        assert np.array_equal(got[k][1], w[1]) and np.array_equal(got[k][0], w[0]), k
    for e in encs + [sync]:
        e.Close()


def test_host_rolling_pipeline_with_pinned_caller_buffers(oracle, kclib, monkeypatch):
    """kc_host_alloc: page-locked source and destination buffers are recognised by the host-buffer entry points and used for the
    DMA directly (no staging copies); pinned source + pageable destination and the reverse work too.  Same bytes as pageable."""
    _torch()
    import ctypes as C
    from compress_amd import _lib
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL_MIB", "6")
    n, usz = 320, 131072
    buf = corpora.corpus("M", n, usz)
    off = np.arange(n + 1, dtype=np.uint64) * usz
    enc = _enc(1)
    want, want_off = enc.EncodeUnits(buf, off)
    cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    psrc, pdst = _lib.PinnedBuffer(buf.size), _lib.PinnedBuffer(cap)
    psrc.a[:] = buf
    ctx = enc.ctx()
    for src_a, dst_a in ((psrc.a, pdst.a), (psrc.a, np.zeros(cap, dtype=np.uint8)), (buf, pdst.a)):
        eo = np.zeros(n + 1, dtype=np.uint64)
        dst_a[:int(want_off[n])] = 0
        ctx.check(ctx.L.kc_zstd_encode_units(ctx.h, C.byref(enc.o), src_a.ctypes.data, off.ctypes.data, n, dst_a.ctypes.data, cap, eo.ctypes.data))
        assert np.array_equal(eo, want_off) and np.array_equal(dst_a[:int(eo[n])], want)
    enc.Close()
    psrc.free()
    pdst.free()


def test_trim_gives_memory_back_and_the_next_call_is_the_same(oracle, kclib, monkeypatch):
    """kc_ctx_trim / kc_device_trim: a context's scratch and the rolling pipeline's slots and lane scratch are freed (device memory
    in use drops), the handles stay usable and the next calls produce the same bytes; a trim while a submitted call is in flight is
    refused."""
    torch = _torch()
    from compress_amd import _lib
    monkeypatch.setenv("KC_HOST_OVERLAP_MIN_MIB", "1")
    monkeypatch.setenv("KC_HOST_ROLL_MIB", "4")
    n, usz = 256, 131072
    buf = corpora.corpus("T", n, usz)
    off = np.arange(n + 1, dtype=np.uint64) * usz
    enc = _enc(1)
    out, out_off = enc.EncodeUnits(buf, off)  # through the rolling pipeline: slots + lanes allocated
    d_src = torch.from_numpy(buf).cuda()
    cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    dev_off = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)  # the context's own scratch allocated
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    enc.ctx().trim()
    _lib.device_trim(0)
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 > free0 + (32 << 20), (free0, free1)
    out2, out_off2 = enc.EncodeUnits(buf, off)
    assert np.array_equal(out2, out) and np.array_equal(out_off2, out_off)
    dev_off2 = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    assert np.array_equal(dev_off2, dev_off) and np.array_equal(d_dst[:int(dev_off[n])].cpu().numpy(), out)
    enc.EncodeUnitsSubmit(buf, off)
    with pytest.raises(_lib.KcError):
        enc.ctx().trim()
    enc.Wait()
    enc.Close()


def test_many_small_units_fit_the_scratch_budget(oracle, kclib):
    """A batch of many small units needs scratch per unit and per block, not per input byte (tables 640 KiB per unit at
    SpeedDefault): batches are cut by a scratch budget instead of asking hipMalloc for hundreds of GiB."""
    _torch()
    n, usz = 40000, 256
    rng = np.random.default_rng(5)
    text = corpora.corpus("T", 80, 131072)
    buf = text[:n * usz].copy()
    off = np.arange(n + 1, dtype=np.uint64) * usz
    enc = _enc(2)
    enc.ctx()
    out, out_off = enc.EncodeUnits(buf, off)
    idx = rng.choice(n, 200, replace=False)
    for i in idx:
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        assert a == oracle.ZstdOracle(level=2).encode_all(buf[i * usz:(i + 1) * usz].tobytes()), i
    enc.Close()


def test_device_decoder_empty_units_and_tiny_blocks(oracle, kclib):
    """(1) an empty unit written with WithZeroFrames(false) is no bytes at all — the verifier must report it ok
    whatever a previous decode left in its per-unit checksum flags; (2) a compressed block too small for its own literals
    header must be rejected (blockdec.go ErrBlockTooSmall), never read past."""
    torch = _torch()
    from compress_amd import zstd
    t = corpora.corpus("T", 2, 131072, first_unit=5).tobytes()
    # poison the per-unit flags with a checksummed decode first
    enc = _enc(1)
    units = [t[:70000], t[:50], t[100:131072]]
    ubuf, off = corpora.pack_units(units)
    d_src = torch.from_numpy(ubuf).cuda()
    cap = sum(((enc.MaxEncodedSize(len(u)) + 15) & ~15) for u in units) + 64
    d_enc = torch.empty(cap, dtype=torch.uint8, device="cuda")
    eoff = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_enc.data_ptr(), cap)
    d_out = torch.zeros(len(ubuf) + 64, dtype=torch.uint8, device="cuda")
    assert not enc.DecodeUnitsDevice(d_enc.data_ptr(), eoff, d_out.data_ptr(), off).any()
    # (1) empty units, no zero frames
    enc0 = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithZeroFrames(False))
    units0 = [b"", t[:1000], b"", b""]
    ubuf0, off0 = corpora.pack_units(units0)
    out0, eoff0 = enc0.EncodeUnits(ubuf0, off0)
    assert int(eoff0[1]) == 0 and int(eoff0[3]) == int(eoff0[2]) == int(eoff0[4])
    d_e0 = torch.from_numpy(np.concatenate([out0, np.zeros(16, dtype=np.uint8)])).cuda()
    d_o0 = torch.zeros(len(ubuf0) + 64, dtype=torch.uint8, device="cuda")
    st0 = enc.DecodeUnitsDevice(d_e0.data_ptr(), eoff0, d_o0.data_ptr(), off0)
    assert not st0.any(), st0
    # (2) frames whose single compressed block is cut to 0..4 bytes: magic + FHD(single segment, 1-byte FCS) + block header
    bad_frames = []
    for bn in range(0, 5):
        body = bytes([0x02, 0xFF, 0xFF, 0xFF][:bn]) if bn else b""  # literals type 2 (compressed), size format 0: needs 3 header bytes
        hdr = (bn << 3) | (2 << 1) | 1                              # last block, type 2 = compressed
        bad_frames.append(bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 10]) + bytes([hdr & 0xFF, (hdr >> 8) & 0xFF, (hdr >> 16) & 0xFF]) + body)
    eo = np.zeros(len(bad_frames) + 1, dtype=np.uint64); eo[1:] = np.cumsum([len(f) for f in bad_frames])
    do = np.arange(len(bad_frames) + 1, dtype=np.uint64) * 10
    d_b = torch.from_numpy(np.frombuffer(b"".join(bad_frames), dtype=np.uint8).copy()).cuda()
    d_bo = torch.zeros(int(do[-1]) + 64, dtype=torch.uint8, device="cuda")
    stb = enc.DecodeUnitsDevice(d_b.data_ptr(), eo, d_bo.data_ptr(), do)
    assert stb.all(), stb  # every one of them is refused
    enc.Close()
    enc0.Close()


@pytest.mark.parametrize("level", LEVELS_B)
def test_rle_literal_sections_bit_exact(oracle, kclib, level):
    """RLE literal sections (zstd/blockenc.go:554-561: huff0.ErrUseRLE): units whose literals are one repeated byte between
    dictionary matches (corpora.rle_literal_units; tests/test_outcome_coverage.py shows the oracle takes the RLE branch for them).
    The device's RLE-literals code path against the oracle, at every level and on both SpeedFastest kernel families."""
    _torch()
    from compress_amd import zstd
    dct, units = corpora.rle_literal_units()
    buf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, *_lo(level), zstd.WithEncoderDictRaw(7, dct))
    out, out_off = enc.EncodeUnits(buf, off)
    _path_ran(enc, level)
    ref, ref_off = oracle.zstd_encode_units(buf, off, threads=4, level=_li(level), dict_id=7, dict_content=dct)
    assert np.array_equal(out_off, ref_off)
    assert np.array_equal(out, ref)
    enc.Close()
