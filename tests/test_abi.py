"""Host-side checks of the product library that need no GPU: it loads, exports every symbol
include/kcgpu.h declares, resolves options like zstd/encoder_options.go, and refuses to
compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(kclib):
    from compress_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "kcgpu.h")).read()
    declared = set(re.findall(r"\b(kc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"kc_ctx", "kc_status", "kc_level", "kc_zstd_opts", "kc_timings"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(kclib, s), s


def _opts(*ops):
    from compress_amd import zstd
    e = zstd.NewWriter(None, *ops)
    return e.o


def test_option_resolution_matches_reference():
    """encoder_options.go:36-48 (defaults), :236-266 (WithEncoderLevel), :110-133 (WithWindowSize) incl.
    the order dependence of the two."""
    from compress_amd import zstd
    o = _opts()
    assert (o.level, o.window_size, o.block_size, o.crc, o.single, o.full_zero, o.all_lit_entropy) == (2, 8 << 20, 128 << 10, 1, -1, 1, 0)
    o = _opts(zstd.WithEncoderLevel(zstd.SpeedFastest))
    assert (o.window_size, o.block_size, o.all_lit_entropy) == (4 << 20, 1 << 16, 0)  # App. A-1
    o = _opts(zstd.WithEncoderLevel(zstd.SpeedBetterCompression))
    assert (o.window_size, o.block_size, o.all_lit_entropy) == (8 << 20, 128 << 10, 1)
    # custom window first: level keeps it and keeps the block size derived from it
    o = _opts(zstd.WithWindowSize(1 << 15), zstd.WithEncoderLevel(zstd.SpeedFastest))
    assert (o.window_size, o.block_size) == (1 << 15, 1 << 15)
    # level first, then a large custom window: block size stays at the level's 64 KiB
    o = _opts(zstd.WithEncoderLevel(zstd.SpeedFastest), zstd.WithWindowSize(1 << 20))
    assert (o.window_size, o.block_size) == (1 << 20, 1 << 16)
    o = _opts(zstd.WithAllLitEntropyCompression(True), zstd.WithEncoderLevel(zstd.SpeedFastest))
    assert o.all_lit_entropy == 1  # customALEntropy sticks
    with pytest.raises(ValueError):
        _opts(zstd.WithWindowSize(1000))
    with pytest.raises(ValueError):
        _opts(zstd.WithWindowSize(1 << 30))
    with pytest.raises(ValueError):
        _opts(zstd.WithEncoderLevel(0))
    # zstd/encoder_options_test.go:9-124 level mappings
    assert zstd.EncoderLevelFromString("Fastest") == (True, zstd.SpeedFastest)
    assert zstd.EncoderLevelFromString("nope")[0] is False
    assert [zstd.EncoderLevelFromZstd(i) for i in (1, 2, 3, 5, 6, 9, 10, 22)] == [1, 1, 2, 2, 3, 3, 4, 4]


def test_max_encoded_size_matches_oracle(oracle):
    from compress_amd import zstd
    for lvl in (1, 2, 3):
        e = zstd.NewWriter(None, zstd.WithEncoderLevel(lvl))
        oe = oracle.ZstdOracle(level=lvl)
        for n in (0, 1, 255, 256, 65535 + 256, 65536 + 256, 65536, 131072, 1 << 20, (1 << 31) - 1, 1 << 31):
            assert e.MaxEncodedSize(n) == oe.max_encoded_size(n), (lvl, n)
    e = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithEncoderCRC(False))
    assert e.MaxEncodedSize(131072) == 131072 + 6 + 4 + 9


def test_no_cpu_fallback_without_device():
    """On a machine without a GPU the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from compress_amd import zstd, KcError
    e = zstd.NewWriter(None, zstd.WithEncoderLevel(zstd.SpeedFastest))
    with pytest.raises(KcError):
        e.EncodeAll(b"hello world")


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under compress_amd/ may import, link or mention it."""
    for dp, _, fns in os.walk(os.path.join(ROOT, "compress_amd")):
        if "_build" in dp or "__pycache__" in dp:
            continue
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "kcoracle" not in txt and "oracle_lib" not in txt and "kco_" not in txt, os.path.join(dp, fn)


def test_corpus_generator_deterministic_and_shardable(kclib):
    import numpy as np
    import corpora
    a = corpora.corpus("T", 8, 4096)
    b = np.concatenate([corpora.corpus("T", 3, 4096), corpora.corpus("T", 5, 4096, first_unit=3)])
    assert np.array_equal(a, b)
    for k in "THJM":
        x = corpora.corpus(k, 2, 70000)
        assert np.array_equal(x, corpora.corpus(k, 2, 70000))
        assert not np.array_equal(x[:70000], x[70000:])
    h = corpora.corpus("H", 1, 1 << 16)
    assert len(np.unique(h)) == 256


def test_full_format_dictionary_loader_matches_oracle(oracle):
    """kc_zstd_opts_dict (WithEncoderDict) against the oracle's loadDict restatement (zstd/dict.go:71-150) on the
    reference's own dictionary fixture (tests/golden/dict/d0.dict): ID, offsets, literal cTable, content start;
    and loadDict's error cases."""
    import ctypes as C
    from compress_amd import zstd, _lib
    blob = open(os.path.join(ROOT, "tests", "golden", "dict", "d0.dict"), "rb").read()
    ref = oracle.zstd_load_dict(blob)
    assert ref is not None and ref["huf_len"] > 0
    e = zstd.NewWriter(None, zstd.WithEncoderDict(blob))
    o = e.o
    assert o.dict_id == ref["id"] == int.from_bytes(blob[4:8], "little")
    assert list(o.dict_offsets) == ref["offsets"]
    assert (o.dict_huf_len, o.dict_huf_log) == (ref["huf_len"], ref["huf_log"])
    assert list(o.dict_huf_val) == ref["val"]
    assert list(o.dict_huf_nbits) == ref["nbits"]
    assert o.dict_len == len(blob) - ref["content_off"]
    base = C.addressof(o._dict_keep)
    assert o.dict - base == ref["content_off"]
    # error behaviour: bad magic, ID 0, truncated tables, offsets beyond the content
    for bad in (b"\x00" + blob[1:], blob[:4] + b"\0\0\0\0" + blob[8:], blob[:20], blob[:60], blob[:ref["content_off"] + 2]):
        assert oracle.zstd_load_dict(bad) is None
        with pytest.raises(ValueError):
            zstd.NewWriter(None, zstd.WithEncoderDict(bad))
    # a raw dictionary afterwards resets offsets and drops the literal table (last option wins, like o.dict = ...)
    e = zstd.NewWriter(None, zstd.WithEncoderDict(blob), zstd.WithEncoderDictRaw(7, b"x" * 100))
    assert (list(e.o.dict_offsets), e.o.dict_huf_len, e.o.dict_id) == ([1, 4, 8], 0, 7)


def test_s2_index_builder_matches_oracle(oracle):
    """s2.Index add / reduce / appendTo (s2/index.go:57-236): the product's writer-side index (compress_amd/s2.py) against the
    oracle's restatement on random write sequences, incl. skippable blocks, sub-MiB blocks and > 65536 entries (reduce)."""
    import random
    from compress_amd import s2
    rnd = random.Random(5)
    for trial in range(30):
        bs = rnd.choice([4096, 65536, 1 << 20, 4 << 20])
        n = rnd.choice([0, 1, 5, 300, 5000, 70000, 140000])
        adds, c, u = [(0, 0)], 10, 0
        for _ in range(n):
            if rnd.random() < 0.02:
                adds.append((c, u)); c += rnd.randint(5, 200)
            adds.append((c, u)); c += rnd.randint(20, bs); u += rnd.randint(1, bs) if rnd.random() < 0.3 else bs
        ix = s2.Index(bs)
        for a in adds:
            ix.add(*a)
        assert ix.append_to(u, c) == oracle.s2_index(bs, adds, u, c), (trial, bs, n)
    ix = s2.Index(1 << 20)
    b = ix.append_to(0, 0)
    assert b[:10] == b"\x99" + (len(b) - 4).to_bytes(3, "little") + b"s2idx\x00" and b[-6:] == b"\x00xdi2s"


def test_reference_option_tables():
    """The tables of zstd/encoder_options_test.go: TestEncoderLevelFromString (:9-85), TestEncoderLevelFromZstd (:87-124),
    TestWindowSize (:126-155)."""
    from compress_amd import zstd
    for s, ok, lvl in (("fastest", True, 1), ("FASTEST", True, 1), ("default", True, 2), ("Default", True, 2), ("invalid", False, 2),
                       ("unknown", False, 2), ("", False, 2)):
        assert zstd.EncoderLevelFromString(s) == (ok, lvl), s
    for z, lvl in ((1, 1), (-1, 1), (3, 2), (4, 2)):
        assert zstd.EncoderLevelFromZstd(z) == lvl
    for ws, err in ((1 << 9, True), (1 << 10, False), ((1 << 10) + 1, True), ((1 << 10) * 3, True), (zstd.MaxWindowSize, False)):
        if err:
            with pytest.raises(ValueError):
                zstd.NewWriter(None, zstd.WithWindowSize(ws))
        else:
            assert zstd.NewWriter(None, zstd.WithWindowSize(ws)).o.window_size == ws


def test_s2_calc_skippable_frame():
    """calcSkippableFrame (s2/writer.go:858-875): the result makes the total a multiple and is 0 or at least a chunk header."""
    from compress_amd import s2
    for written in (0, 1, 3, 4, 5, 99, 100, 4095, 65536, 1234567):
        for mult in (1, 2, 3, 4, 5, 7, 8, 100, 4096, 8000, 4 << 20):
            add = s2.calc_skippable_frame(written, mult)
            assert (written + add) % mult == 0 and (add == 0 or add >= 4), (written, mult, add)
            assert add == 0 or add < mult + 4 or mult < 4
    assert s2.calc_skippable_frame(10, 4) == 6 and s2.calc_skippable_frame(9, 4) == 7 and s2.calc_skippable_frame(8, 4) == 0


def test_dictionary_loader_differential_fuzz(oracle):
    """Mutated copies of the reference's d0.dict (bit flips / byte changes in the entropy tables, offsets and header,
    truncations): the product's loader (kc_dict.cpp, written from the format description) and the oracle's restatement of
    loadDict must agree on accept / reject, and on every table and offset when they accept.  (This test found two bugs: the
    product ended FSE weight streams on a partial read, the oracle skipped fseDecoder.transform's alphabet check.)"""
    import random
    from compress_amd import _lib
    L = _lib.load()
    blob = open(os.path.join(ROOT, "tests", "golden", "dict", "d0.dict"), "rb").read()
    co = oracle.zstd_load_dict(blob)["content_off"]
    rnd = random.Random(11)
    accepted = 0
    for it in range(2500):
        b = bytearray(blob[:co + 40])
        for _ in range(rnd.choice([1, 1, 2, 3, 8])):
            p = rnd.randrange(4 if rnd.random() < 0.9 else 0, co + 12)
            if rnd.random() < 0.5:
                b[p] = rnd.getrandbits(8)
            else:
                b[p] ^= 1 << rnd.randrange(8)
        if rnd.random() < 0.1:
            b = b[:rnd.randrange(8, len(b))]
        bb = bytes(b)
        ref = oracle.zstd_load_dict(bb)
        o = _lib.ZstdOpts()
        L.kc_zstd_opts_default(C.byref(o))
        buf = C.create_string_buffer(bb, len(bb))
        ok = L.kc_zstd_opts_dict(C.byref(o), C.cast(buf, C.c_void_p), len(bb)) == 0
        assert ok == (ref is not None), (it, ok)
        if ok:
            accepted += 1
            assert list(o.dict_huf_val) == ref["val"] and list(o.dict_huf_nbits) == ref["nbits"], it
            assert list(o.dict_offsets) == ref["offsets"] and o.dict_len == len(bb) - ref["content_off"], it
    assert 100 < accepted < 2400


def test_stream_block_plan_matches_the_oracle_frames():
    """Host logic of kc_zstd_encode_streams_cuts without a GPU: kc_zstd_plan_stream_blocks (the function batch_begin lays the
    blocks out with) against the block structure of the oracle's Write / Flush / Close frames — block count and decoded sizes,
    stream frame vs EncodeAll frame, the empty last block."""
    import ctypes as C
    import random
    import numpy as np
    import corpora
    import oracle_lib as oracle
    from compress_amd import _lib
    L = _lib.load()
    e = oracle.ZstdOracle(level=2)
    bs = e.opts.block_size
    data = corpora.corpus("H", 5, 131072).tobytes()   # incompressible: every block is stored raw, so the frame shows the block sizes
    rnd = random.Random(5)
    cases = [(1000, [10, 500]), (1000, [1000]), (1000, [0]), (1000, []), (bs, []), (bs, [bs]), (2 * bs, [bs]), (bs + 100, [50, bs + 100]),
             (3 * bs + 1, [7, 7, 2 * bs + 7, 3 * bs + 9]), (0, []), (0, [0]), (5, [1, 2, 3, 4, 5])]
    for _ in range(60):
        n = rnd.choice([rnd.randrange(1, 3000), rnd.randrange(bs - 100, bs + 100), rnd.randrange(2 * bs, 5 * bs)])
        cases.append((n, sorted(rnd.randrange(0, n + 2) for _ in range(rnd.choice([0, 1, 2, 5])))))
    for n, cuts in cases:
        c = np.array(cuts + [0], dtype=np.uint64)
        starts = np.zeros(64, dtype=np.uint32)
        flags = C.c_uint32()
        nb = L.kc_zstd_plan_stream_blocks(bs, n, c.ctypes.data, len(cuts), starts.ctypes.data, len(starts), C.byref(flags))
        assert nb >= 0
        fr = e.encode_stream(data[:n], cuts)
        stream = bool(flags.value & 1)
        if n == 0:
            assert nb == 0 and not stream
            continue
        if not stream:
            assert nb == 1 and fr == e.encode_all(data[:n]), (n, cuts)
            continue
        assert fr[4] == 0x04 and fr != e.encode_all(data[:n]), (n, cuts)
        p, sizes = 6, []
        while True:
            bh = fr[p] | fr[p + 1] << 8 | fr[p + 2] << 16
            typ, sz = (bh >> 1) & 3, bh >> 3
            assert typ == 0, (n, cuts)
            p += 3 + sz
            sizes.append(sz)
            if bh & 1:
                break
        want = [int(starts[i + 1]) - int(starts[i]) for i in range(nb - 1)] + [n - int(starts[nb - 1])]
        if flags.value & 2:
            want.append(0)
        assert sizes == want, (n, cuts, sizes, want)
