"""Pin the CPU oracle against the reference's own known-answer tests and fixtures (no GPU)."""
import ctypes as C
import hashlib
import json
import os

import pytest

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))
REF = "/root/reference"


def test_xxh64_kats(oracle):
    L = oracle.lib()
    for s, h in KATS["xxh64"]:
        b = s.encode()
        assert L.kco_xxh64(b, len(b)) == int(h, 16)
        for chunk in range(1, max(2, len(b) + 1)):  # xxhash_test.go testDigest: every chunking of Write
            assert L.kco_xxh64_chunked(b, len(b), chunk) == int(h, 16)


def test_matchlen_kat(oracle):
    """zstd/zstd_test.go:45-64 TestMatchLen."""
    L = oracle.lib()
    b = bytes(range(130))
    for l in range(130):
        a = bytearray(b)
        a[l] ^= 0xFF
        assert L.kco_zstd_matchlen(bytes(a), len(a), b) == l
        assert L.kco_zstd_matchlen(bytes(a[:l]), l, b) == l


def test_hashlen_matches_formula(oracle):
    """zstd/hash.go:20-35 restated independently in Python."""
    L = oracle.lib()
    M = (1 << 64) - 1
    primes = {3: 506832829, 4: 2654435761, 5: 889523592379, 6: 227718039650203, 7: 58295818150454627, 8: 0xcf1bbcdcb7a56463}
    for u in (0, 1, 0x0123456789abcdef, M, 0x8000000000000000, 0x00ff00ff00ff00ff):
        for bits in (13, 15, 17, 19):
            for mls in (3, 4, 5, 6, 7, 8):
                if mls == 3:
                    want = ((((u << 8) & 0xFFFFFFFF) * primes[3]) & 0xFFFFFFFF) >> (32 - bits)
                elif mls == 4:
                    want = (((u & 0xFFFFFFFF) * primes[4]) & 0xFFFFFFFF) >> (32 - bits)
                elif mls == 8:
                    want = ((u * primes[8]) & M) >> (64 - bits)
                else:
                    want = ((((u << (64 - 8 * mls)) & M) * primes[mls]) & M) >> (64 - bits)
                assert L.kco_zstd_hashlen(u, bits, mls) == (want & 0xFFFFFFFF)


def test_s2_emit_literal_kat(oracle):
    L = oracle.lib()
    dst = C.create_string_buffer(70000)
    nines = b"\x99" * 65536
    for length, want in KATS["s2_emit_literal"]:
        n = L.kco_s2_emit_literal(dst, nines[:length], length)
        assert dst.raw[n - length:n] == nines[:length]
        assert dst.raw[:n - length].hex() == want


def test_s2_emit_copy_kat(oracle):
    L = oracle.lib()
    dst = C.create_string_buffer(1024)
    for offset, length, want in KATS["s2_emit_copy"]:
        n = L.kco_s2_emit_copy(dst, offset, length)
        assert dst.raw[:n].hex() == want, (offset, length)


def test_s2_max_encoded_len_kat(oracle, kclib):
    L = oracle.lib()
    for n, want in KATS["s2_max_encoded_len"]:
        assert L.kco_s2_max_encoded_len(n) == want, n
        assert kclib.kc_s2_max_encoded_len(n) == want, n
    for i in list(range(0, 70000, 7)) + [(1 << 22) - 1, 1 << 22]:
        varint = 1
        z = i << 1
        while z >= 0x80:
            z >>= 7
            varint += 1
        extra = 0 if i == 0 else 1 if i < 60 else 2 if i < 256 else 3 if i < 65536 else 4 if i < (1 << 24) else 5
        assert L.kco_s2_max_encoded_len(i) == i + varint + extra == kclib.kc_s2_max_encoded_len(i)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present on this machine")
def test_frame_boundaries_on_reference_fixtures(oracle):
    """C1: EncodeAll(e.txt) at SpeedFastest starts 28 B5 2F FD A4 A3 86 01 00 and ends 5F 0C 04 7D."""
    enc = oracle.ZstdOracle(level=1)
    for name, g in KATS["files"].items():
        if "frame_prefix" not in g:
            continue
        data = open(os.path.join(REF, "testdata", name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == g["sha256"]
        out = enc.encode_all(data)
        assert out.hex().startswith(g["frame_prefix"]), name
        assert out[-4:].hex() == g["frame_suffix"], name
        assert oracle.lib().kco_xxh64(data, len(data)) == int(g["xxh64"], 16)
        assert oracle.zstd_decompress(out, len(data) + 16) == data
    e = open(os.path.join(REF, "testdata", "e.txt"), "rb").read()
    out = enc.encode_all(e)
    assert out[:9].hex() == "28b52ffda4a3860100" and out[-4:].hex() == "5f0c047d"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present on this machine")
@pytest.mark.parametrize("level", [1, 2, 3])
def test_reference_corpora_roundtrip(oracle, level):
    """The reference's encoder tests are round-trip tests (zstd/encoder_test.go:68-160, fuzz_test.go:154):
    every oracle output must decode to the input with an independent decoder (libzstd 1.4.8) and stay
    within MaxEncodedSize; a persistent encoder state must give the same bytes as a fresh one."""
    import zipfile
    enc = oracle.ZstdOracle(level=level)
    n = 0
    for zf in ("zstd/testdata/fuzz/encode-corpus-raw.zip", "zstd/testdata/comp-crashers.zip"):
        z = zipfile.ZipFile(os.path.join(REF, zf))
        for k, name in enumerate(z.namelist()):
            d = z.read(name)
            out = enc.encode_all(d)
            if len(d) == 0:
                assert out.hex() == "28b52ffd2000010000"
                continue
            assert len(out) <= enc.max_encoded_size(len(d))
            assert oracle.zstd_decompress(out, len(d) + 16) == d, (zf, name)
            if k % 16 == 0:
                assert oracle.ZstdOracle(level=level).encode_all(d) == out, "encoder state leaked between frames"
            n += 1
    assert n > 3000


def test_empty_and_tiny_frames(oracle):
    enc = oracle.ZstdOracle(level=1)
    assert enc.encode_all(b"").hex() == "28b52ffd2000010000"  # App. A-18
    assert oracle.ZstdOracle(level=1, full_zero=False).encode_all(b"") == b""
    for n in range(1, 40):
        d = bytes((i * 7) & 0xFF for i in range(n))
        out = enc.encode_all(d)
        assert oracle.zstd_decompress(out, n + 16) == d
    # The one whole frame the reference's tests spell out byte for byte (zstd/decoder_test.go:2093, TestIgnoreChecksum: "zstd file
    # containing text 'compress\n' and has an xxhash checksum"): single-segment header with a 1-byte content size, one raw last
    # block, the low four bytes of XXH64 — what EncodeAll writes for the same nine bytes with WithSingleSegment(true) (by default inputs
    # of at most MinWindowSize get a window descriptor instead: encoder.go:755-759) and the default WithEncoderCRC(true).
    blob = bytes([0x28, 0xb5, 0x2f, 0xfd, 0x24, 0x09, 0x49, 0x00, 0x00]) + b"compress\n" + bytes([0x79, 0x6e, 0xe0, 0xd2])
    for level in (1, 2, 3, 4):
        assert oracle.ZstdOracle(level=level, single=True).encode_all(b"compress\n") == blob
        assert oracle.ZstdOracle(level=level).encode_all(b"compress\n") == blob[:4] + b"\x04\x00" + blob[6:]


def test_s2_roundtrip_and_bounds(oracle):
    import corpora
    for kind in "TJHM":
        buf = corpora.corpus(kind, 4, 65536).tobytes()
        for n in (0, 1, 31, 32, 33, 1000, 65535, 65536, 65537, 200000):
            d = buf[:n]
            e = oracle.s2_encode(d)
            assert len(e) <= oracle.lib().kco_s2_max_encoded_len(n)
            assert oracle.s2_decode(e, n + 8) == d
    # TestEncodeNoiseThenRepeats (s2/s2_test.go:736-753): noise then repeats must compress the repeats
    import numpy as np
    rng = np.random.default_rng(1)
    for orig in (256 * 1024, 2048 * 1024):
        src = bytearray(rng.integers(0, 256, orig, dtype=np.uint8).tobytes())
        half = orig // 2
        src[half:] = bytes([(i >> 8) & 0xFF for i in range(half)])
        e = oracle.s2_encode(bytes(src))
        assert len(e) <= orig * 3 // 4
        assert oracle.s2_decode(e, orig + 8) == bytes(src)


@pytest.mark.parametrize("level", [1, 2, 3])
def test_raw_dictionary_roundtrip(oracle, level):
    """WithEncoderDictRaw (encoder_options.go:398-406): dictionary content is history only; libzstd decodes
    raw-content dictionaries as ID 0; a non-zero ID only adds the dictID field to the frame header."""
    import corpora
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    e0 = oracle.ZstdOracle(level=level, dict_id=0, dict_content=dct)
    e1 = oracle.ZstdOracle(level=level, dict_id=1, dict_content=dct)
    plain = oracle.ZstdOracle(level=level)
    m = corpora.corpus("M", 6, 131072)
    t = corpora.corpus("T", 3, 131072, first_unit=100).tobytes()
    units = [m[i * 131072:(i + 1) * 131072].tobytes() for i in range(6)] + [t[:n] for n in (5, 100, 20000, 40000, 131072, 300000)] + [dct]
    won = 0
    for u in units:
        out = e0.encode_all(u)
        assert oracle.zstd_decompress(out, len(u) + 16, dict_content=dct) == u
        o1 = e1.encode_all(u)
        k = 5 if len(u) > 1024 else 6  # the dictID follows the window descriptor of non-single-segment frames (frameenc.go:66-73)
        assert o1[:4] == out[:4] and o1[4] == out[4] | 1 and o1[5:k] == out[5:k] and o1[k] == 1 and o1[k + 1:] == out[k:]
        won += len(out) < len(plain.encode_all(u))
    assert won >= 6  # the dictionary helps on text-like units
    assert e0.encode_all(b"").hex() == "28b52ffd2000010000"


def _dict_fixture(oracle):
    """The reference's dictionary fixture (zstd/dict_test.go:150 TestEncoder_SmallDict; tests/golden/make_golden.py):
    d0.dict + inputs recovered by decoding the C-zstd frames with libzstd and the dictionary."""
    import glob
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict")
    blob = open(os.path.join(gdir, "d0.dict"), "rb").read()
    ins = [oracle.zstd_decompress(open(f, "rb").read(), 1 << 22, dict_content=blob) for f in sorted(glob.glob(os.path.join(gdir, "*.zst")))]
    return blob, ins


def skewed_dict(blob):
    """d0.dict with its (nearly flat, hence never reused) literal Huffman description replaced by a skewed 7-symbol code
    (direct weights 6,5,4,3,2,1,(1) -> code lengths 1..6,6 for bytes 0..6) and another ID.  Returns (blob, probabilities)."""
    hb = blob[8]
    end = 9 + hb if hb < 128 else 9 + ((hb - 127) + 1) // 2
    wts = [6, 5, 4, 3, 2, 1]
    desc = bytes([127 + len(wts)]) + bytes([(wts[i] << 4) | wts[i + 1] for i in range(0, len(wts), 2)])
    return blob[:4] + (0x1234567).to_bytes(4, "little") + desc + blob[end:], [2.0 ** -k for k in (1, 2, 3, 4, 5, 6, 6)]


def skewed_units(probs, sizes=(20, 31, 40, 100, 300, 1000, 1024, 4000, 70000), seeds=3):
    import numpy as np
    out = []
    for n in sizes:
        for seed in range(seeds):
            out.append(np.random.default_rng(seed * 1000 + n).choice(7, size=n, p=probs).astype(np.uint8).tobytes())
    return out


@pytest.mark.parametrize("level", [1, 2, 3])
def test_full_dictionary_roundtrip_like_TestEncoder_SmallDict(oracle, level):
    """WithEncoderDict (encoder_options.go:382-391): the same loop as the reference's TestEncoder_SmallDict
    (decode fixture -> EncodeAll with the dictionary -> decode with the dictionary), with libzstd as the decoder."""
    blob, ins = _dict_fixture(oracle)
    enc = oracle.ZstdOracle(level=level, dict_blob=blob)
    plain = oracle.ZstdOracle(level=level)
    tot = tot_plain = 0
    for d in ins + [ins[1][:40], ins[1][:20], ins[1][:9], b""]:
        fr = enc.encode_all(d)
        assert oracle.zstd_decompress(fr, len(d) + 16, dict_content=blob) == d
        tot += len(fr)
        tot_plain += len(plain.encode_all(d))
    assert tot < tot_plain * 0.95  # the dictionary pays on its own kind of data


@pytest.mark.parametrize("level", [1, 2, 3])
def test_full_dictionary_literal_table_is_used_and_decodes(oracle, level):
    """Pins loadDict's Huffman table (huff0.ReadTable -> prevTable) with an independent decoder: on literals drawn from
    the dictionary code's own distribution huff0 keeps the dictionary table (ReusePolicyAllow, compress.go:121-135),
    the frame carries "treeless" literals, and libzstd must decode them with the same dictionary."""
    blob, _ = _dict_fixture(oracle)
    syn, probs = skewed_dict(blob)
    ld = oracle.zstd_load_dict(syn)
    assert (ld["id"], ld["huf_len"], ld["huf_log"], ld["nbits"][:7]) == (0x1234567, 7, 6, [1, 2, 3, 4, 5, 6, 6])
    enc = oracle.ZstdOracle(level=level, dict_blob=syn)
    raw = oracle.ZstdOracle(level=level, dict_id=ld["id"], dict_content=syn[ld["content_off"]:])
    differs = 0
    for d in skewed_units(probs):
        fr = enc.encode_all(d)
        assert oracle.zstd_decompress(fr, len(d) + 16, dict_content=syn) == d
        differs += fr != raw.encode_all(d)
    assert differs >= 4


def test_in_repo_decoder_on_c_zstd_frames(oracle):
    """oracle/kco_zstd_dec.h on frames it did not produce: the reference's dictionary fixtures were compressed
    by the C zstd with d0.dict (Huffman + FSE tables, repeat modes, treeless literals from the dictionary), and must decode
    to what libzstd gives."""
    import glob
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dict")
    blob = open(os.path.join(gdir, "d0.dict"), "rb").read()
    try:
        Z = oracle.libzstd()
    except OSError:
        pytest.skip("no system libzstd to compare with")
    n = 0
    for f in sorted(glob.glob(os.path.join(gdir, "*.zst"))):
        z = open(f, "rb").read()
        buf = C.create_string_buffer(1 << 22)
        ctx = Z.ZSTD_createDCtx()
        r = Z.ZSTD_decompress_usingDict(ctx, buf, 1 << 22, z, len(z), blob, len(blob))
        Z.ZSTD_freeDCtx(ctx)
        assert not Z.ZSTD_isError(r)
        assert oracle.zstd_decode(z, 1 << 22, dict_blob=blob) == buf.raw[:r]
        n += 1
    assert n >= 8
    # malformed input is rejected, not mis-decoded
    z = open(sorted(glob.glob(os.path.join(gdir, "z0076*.zst")))[0], "rb").read()
    for bad in (z[:-1], z[:20], b"\x00" + z[1:], z[:9] + bytes([z[9] ^ 0x40]) + z[10:]):
        with pytest.raises(RuntimeError):
            oracle.zstd_decode(bad, 1 << 22, dict_blob=blob)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present on this machine")
def test_in_repo_decoder_on_reference_regression_frames(oracle):
    """The reference's decoder test inputs (zstd/testdata/good.zip, benchdecoder.zip: C-zstd frames at many levels, skippable
    frames, multi-frame files): wherever libzstd decodes a file, the in-repo decoder must produce the same bytes."""
    import zipfile
    try:
        Z = oracle.libzstd()
    except OSError:
        pytest.skip("no system libzstd to compare with")
    checked = 0
    for name in ("good.zip", "benchdecoder.zip"):
        zf = zipfile.ZipFile(os.path.join(REF, "zstd", "testdata", name))
        for n in zf.namelist()[:400]:
            z = zf.read(n)
            if len(z) < 6 or len(z) > 300000:
                continue
            cap = 16 << 20
            buf = C.create_string_buffer(cap)
            r = Z.ZSTD_decompress(buf, cap, z, len(z))
            if Z.ZSTD_isError(r):
                continue
            assert oracle.zstd_decode(z, cap) == buf.raw[:r], (name, n)
            checked += 1
    assert checked >= 20


@pytest.mark.parametrize("level", [1, 2, 3])
def test_stream_restatement_structure_and_roundtrip(oracle, level):
    """Write ... Flush ... Close (encoder.go:154-428, 567-649) as restated in OracleEncoder::encodeStream: below one block the
    stream IS the EncodeAll frame; from one block on the frame has no content size and no single-segment flag, every cut
    (block size or Flush) starts a block, an input ending on a cut gets a trailing empty raw last block; both decoders agree."""
    import corpora
    e = oracle.ZstdOracle(level=level)
    bs = e.opts.block_size
    t = corpora.corpus("T", 3, 131072, first_unit=11).tobytes()
    assert e.encode_stream(t[:bs - 1]) == e.encode_all(t[:bs - 1])
    assert e.encode_stream(b"").hex() == "28b52ffd04%02x01000099e9d851" % (((e.opts.window_size - 1).bit_length() - 10) << 3)
    for n, cuts in ((bs, ()), (bs + 1, ()), (2 * bs, ()), (2 * bs + 7, ()), (1000, (10, 500)), (1000, (1000,)), (bs + 100, (50, bs + 100))):
        fr = e.encode_stream(t[:n], cuts)
        assert fr[:4] == b"\x28\xb5\x2f\xfd" and fr[4] == 0x04  # checksum flag only: no FCS, not single segment
        assert oracle.zstd_decompress(fr, n + 16) == t[:n]
        # walk the blocks: sizes follow the cuts, `last` is set exactly once, at the end
        p, sizes, lasts = 6, [], []
        while True:
            bh = fr[p] | fr[p + 1] << 8 | fr[p + 2] << 16
            typ, sz = (bh >> 1) & 3, bh >> 3
            p += 3 + (sz if typ in (0, 2) else 1)
            sizes.append((typ, sz)); lasts.append(bh & 1)
            if bh & 1:
                break
        assert p + 4 == len(fr) and lasts.count(1) == 1
        ends_on_cut = (n % bs == 0 and not cuts) or (cuts and cuts[-1] >= n)
        assert (sizes[-1] == (0, 0)) == bool(ends_on_cut), (n, cuts, sizes)


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_randomized_roundtrips(oracle, seed):
    """Randomised oracle configurations (level, CRC, entropy switches, single segment, small windows, raw dictionaries,
    EncodeAll vs Write/Flush/Close streams) on spliced inputs: every frame must decode to its input with libzstd AND with
    the in-repo decoder (oracle_lib.zstd_decompress cross-checks the two)."""
    import random
    import corpora
    rnd = random.Random(seed)
    text = corpora.corpus("T", 4, 131072, first_unit=300).tobytes()
    js = corpora.corpus("J", 2, 131072, first_unit=30).tobytes()

    def make(n):
        parts, tot = [], 0
        while tot < n:
            k = rnd.randint(0, 5)
            if k == 0:
                p = bytes(rnd.getrandbits(8) for _ in range(rnd.choice([1, 5, 40, 300, 3000])))
            elif k == 1:
                ln = rnd.randint(4, 4000); o = rnd.randint(0, len(text) - ln); p = text[o:o + ln]
            elif k == 2:
                p = bytes([rnd.getrandbits(8)]) * rnd.choice([3, 70, 1000, 70000])
            elif k == 3:
                ln = rnd.randint(4, 3000); o = rnd.randint(0, len(js) - ln); p = js[o:o + ln]
            elif k == 4 and parts:
                p = parts[rnd.randrange(len(parts))]
            else:
                p = bytes(rnd.choice(b"abcdefgh ") for _ in range(rnd.choice([10, 200, 5000])))
            parts.append(p); tot += len(p)
        return b"".join(parts)[:n]

    for it in range(40):
        lvl = rnd.choice([1, 2, 3])
        n = rnd.choice([0, 1, 9, 100, 5000, 65535, 65536, 65537, 131072, 200000])
        kw = dict(level=lvl, crc=rnd.random() < 0.7, no_entropy=rnd.random() < 0.15, full_zero=rnd.random() < 0.8)
        if rnd.random() < 0.3:
            kw["all_lit_entropy"] = rnd.random() < 0.5
        if rnd.random() < 0.3:
            kw["single"] = rnd.random() < 0.5
        if rnd.random() < 0.25:
            ws = 1 << rnd.choice([10, 12, 15, 17, 20])
            kw["window_size"] = ws
            kw["block_size"] = min(ws, (1 << 16) if lvl == 1 else (128 << 10))
        d = make(n)
        dct = None
        if rnd.random() < 0.3 and n > 0:
            dct = make(rnd.choice([100, 5000, 65536]))
            kw["dict_id"] = rnd.choice([0, 0, 7])
            kw["dict_content"] = dct
        e = oracle.ZstdOracle(**kw)
        stream = dct is None and rnd.random() < 0.4
        cuts = tuple(sorted(rnd.sample(range(n + 1), min(n + 1, rnd.choice([0, 0, 1, 3])))))
        fr = e.encode_stream(d, cuts) if stream else e.encode_all(d)
        if not fr:
            assert n == 0 and not kw["full_zero"]
            continue
        if dct is not None and kw["dict_id"] != 0:
            got = oracle.zstd_decode(fr, n + 16, dict_content=dct)  # libzstd only takes raw dictionaries as ID 0
        else:
            got = oracle.zstd_decompress(fr, n + 16, dict_content=dct)
        assert got == d, (it, sorted(kw), n, stream)


_HUF_TABLE = [  # huff0/compress_test.go:20-52: (name, input, err1X, err4X); 0 nil, 1 ErrIncompressible, 2 ErrUseRLE
    ("digits", "e.txt", 0, 0), ("gettysburg", "gettysburg.txt", 0, 0), ("twain", "Mark.Twain-Tom.Sawyer.txt", 0, 0),
    ("random", "sharnd.out", 1, 1), ("low-ent.10k", b"1221" * 10000, 0, 0), ("superlow-ent-10k", b"1" * 10000 + b"2" * 500, 0, 0),
    ("zeroes", bytes(10000), 2, 2), ("crash1", "crash1.bin", 1, 1), ("crash2", "crash2.bin", 0, 1), ("crash3", "crash3.bin", 1, 1),
    ("endzerobits", "endzerobits.bin", 0, 1), ("endnonzero", "endnonzero.bin", 0, 1), ("case1", "case1.bin", 0, 0),
    ("case2", "case2.bin", 0, 0), ("case3", "case3.bin", 0, 0), ("pngdata.001", "pngdata.bin", 0, 0), ("normcount2", "normcount2.bin", 0, 0)]
_FSE_TABLE = [  # fse/fse_test.go:19-50: (name, input, err)
    ("gettysburg", "gettysburg.txt", 0), ("digits", "e.txt", 0), ("twain", "Mark.Twain-Tom.Sawyer.txt", 0), ("random", "sharnd.out", 1),
    ("low-ent", b"1221" * 10000, 0), ("superlow-ent", b"1" * 10000 + b"2" * 500, 0), ("zeroes", bytes(10000), 2), ("crash1", "crash1.bin", 1),
    ("crash2", "crash2.bin", 1), ("crash3", "crash3.bin", 1), ("endzerobits", "endzerobits.bin", 0), ("endnonzero", "endnonzero.bin", 1),
    ("case1", "case1.bin", 1), ("case2", "case2.bin", 1), ("case3", "case3.bin", 1), ("pngdata.001", "pngdata.bin", 0), ("normcount2", "normcount2.bin", 0)]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present on this machine")
def test_huff0_and_fse_outcome_tables(oracle):
    """The reference's own expectation tables for huff0.Compress1X / Compress4X (huff0/compress_test.go:20-52,
    TestCompress1X :225, TestCompress4X :361) and fse.Compress (fse/fse_test.go:19-50, TestCompress :68): which inputs
    compress, which are ErrIncompressible, which are ErrUseRLE — on a fresh Scratch, inputs cut to BlockSizeMax for huff0."""
    L = oracle.lib()

    def load(x):
        return x if isinstance(x, bytes) else open(os.path.join(REF, "testdata", x), "rb").read()

    for name, src, e1, e4 in _HUF_TABLE:
        d = load(src)[:(1 << 18) - 1]
        buf = C.create_string_buffer(len(d) + 1024)
        for four, want in ((0, e1), (1, e4)):
            r = L.kco_huff0_compress(d, len(d), four, 0, buf, len(buf))
            got = 0 if r >= 0 else -r
            assert got == want, ("huff0", name, "4X" if four else "1X", got, want)
    for name, src, e in _FSE_TABLE:
        d = load(src)
        buf = C.create_string_buffer(len(d) + 1024)
        r = L.kco_fse_compress(d, len(d), buf, len(buf))
        got = 0 if r >= 0 else -r
        assert got == e, ("fse", name, got, e)


def test_s2_encode_better_roundtrip_and_gain(oracle):
    """s2.EncodeBetter restatement (oracle/kco_s2.h: encodeBlockBetterGoT): every block decodes back through the restated
    decoder, is never larger than MaxEncodedLen, and is smaller than s2.Encode's block on compressible corpora — the reference's
    own criterion for its levels is the round trip (s2/encode_test.go TestEncoderRegression runs Encode, EncodeBetter, EncodeBest
    through Decode).  Both table variants: <= 64 KiB (u16 tables, 2^16 / 2^13) and larger (2^17 / 2^14 + the long-offset bail)."""
    import corpora
    for kind in "TJM":
        for n in (65536, 65537, 1 << 20):
            d = corpora.corpus(kind, 1, n).tobytes()
            b = oracle.s2_encode_better(d)
            assert oracle.s2_decode(b, n + 8) == d
            assert len(b) <= oracle.lib().kco_s2_max_encoded_len(n)
            assert len(b) < len(oracle.s2_encode(d)), (kind, n)
    for u in corpora.edge_units():
        b = oracle.s2_encode_better(u)
        assert oracle.s2_decode(b, len(u) + 8) == u
    h = corpora.corpus("H", 1, 65536).tobytes()
    assert len(oracle.s2_encode_better(h)) == len(h) + 3 + 3  # incompressible: uvarint(3) + one literal (tag 61<<2 + 2 length bytes)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference fixtures not present on this machine")
def test_s2_encode_better_reference_regressions_roundtrip(oracle):
    import zipfile
    z = zipfile.ZipFile(os.path.join(REF, "s2/testdata/enc_regressions.zip"))
    n = 0
    for name in z.namelist():
        d = z.read(name)
        if not d:
            continue
        assert oracle.s2_decode(oracle.s2_encode_better(d), len(d) + 8) == d, name
        assert oracle.s2_decode(oracle.s2_encode_best(d), len(d) + 8) == d, name
        assert _snappy_decode_strict(oracle.s2_encode_snappy_better(d)) == d, name
        n += 1
    assert n > 40


def _snappy_decode_strict(enc: bytes) -> bytes:
    """A decoder of the ORIGINAL Snappy block format only (google/snappy format_description.txt): literals, copy1 (11-bit offset),
    copy2, copy4 — offset 0 is invalid, which is exactly where S2's repeat tags live.  Independent of the oracle's S2 decoder."""
    n, shift, p = 0, 0, 0
    while True:
        b = enc[p]; p += 1
        n |= (b & 0x7F) << shift
        if b < 0x80:
            break
        shift += 7
    out = bytearray()
    while p < len(enc):
        tag = enc[p]; t = tag & 3
        if t == 0:
            ln = tag >> 2
            p += 1
            if ln >= 60:
                k = ln - 59
                ln = int.from_bytes(enc[p:p + k], "little"); p += k
            ln += 1
            out += enc[p:p + ln]; p += ln
            continue
        if t == 1:
            ln = 4 + ((tag >> 2) & 7); off = ((tag >> 5) << 8) | enc[p + 1]; p += 2
        elif t == 2:
            ln = 1 + (tag >> 2); off = enc[p + 1] | (enc[p + 2] << 8); p += 3
        else:
            ln = 1 + (tag >> 2); off = int.from_bytes(enc[p + 1:p + 5], "little"); p += 5
        assert 0 < off <= len(out), "not a Snappy block: offset %d (repeat tag or corrupt)" % off
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n
    return bytes(out)


def test_s2_encode_snappy_is_snappy_and_roundtrips(oracle):
    """s2.EncodeSnappy restatement: a strict Snappy decoder (no S2 extensions) reads the blocks; the reference's golden
    Snappy file (s2/testdata/Mark.Twain-Tom.Sawyer.txt.rawsnappy, s2_test.go:595-615) checks that decoder."""
    import corpora
    gold_dir = os.path.join(REF, "s2", "testdata")
    if os.path.isdir(gold_dir):
        want = open(os.path.join(gold_dir, "Mark.Twain-Tom.Sawyer.txt"), "rb").read()
        assert _snappy_decode_strict(open(os.path.join(gold_dir, "Mark.Twain-Tom.Sawyer.txt.rawsnappy"), "rb").read()) == want
    for kind in "TJMH":
        for n in (65536, 65537, 300000):
            d = corpora.corpus(kind, 1, n).tobytes()
            b = oracle.s2_encode_snappy(d)
            assert _snappy_decode_strict(b) == d
            assert oracle.s2_decode(b, n + 8) == d
            assert len(b) <= oracle.lib().kco_s2_max_encoded_len(n)
    for u in corpora.edge_units():
        b = oracle.s2_encode_snappy(u)
        assert _snappy_decode_strict(b) == u
    # the default level does use repeat tags on such data: the strict decoder must refuse at least one of these blocks
    refused = 0
    for kind in "TJ":
        try:
            _snappy_decode_strict(oracle.s2_encode(corpora.corpus(kind, 1, 65536).tobytes()))
        except AssertionError:
            refused += 1
    assert refused > 0


def test_s2_encode_snappy_better_is_snappy_and_not_larger(oracle):
    """s2.EncodeSnappyBetter restatement (encodeBlockBetterSnappyGo / ...64K): strict Snappy blocks that round-trip; on compressible
    corpora the better parse is not larger than s2.EncodeSnappy's by more than a percent (it searches every position)."""
    import corpora
    for kind in "TJMH":
        for n in (65536, 65537, 300000):
            d = corpora.corpus(kind, 1, n).tobytes()
            b = oracle.s2_encode_snappy_better(d)
            assert _snappy_decode_strict(b) == d
            assert oracle.s2_decode(b, n + 8) == d
            assert len(b) <= oracle.lib().kco_s2_max_encoded_len(n)
            if kind in "TJ":
                assert len(b) <= len(oracle.s2_encode_snappy(d)) * 1.01, (kind, n, len(b), len(oracle.s2_encode_snappy(d)))
    for u in corpora.edge_units():
        b = oracle.s2_encode_snappy_better(u)
        assert _snappy_decode_strict(b) == u


def test_s2_encode_best_restatement_roundtrips_and_is_smallest(oracle):
    """s2.EncodeBest restatement (encodeBlockBest, no dictionary; oracle only — groundwork for the next device level): every block
    decodes back, stays within MaxEncodedLen, and on compressible corpora is not larger than s2.EncodeBetter's (it scores up to
    thirteen candidates per position and indexes every byte of a match)."""
    import corpora
    for kind in "TJMH":
        for n in (65536, 65537, 300000):
            d = corpora.corpus(kind, 1, n).tobytes()
            b = oracle.s2_encode_best(d)
            assert oracle.s2_decode(b, n + 8) == d, (kind, n)
            assert len(b) <= oracle.lib().kco_s2_max_encoded_len(n)
            if kind in "TJ":
                assert len(b) <= len(oracle.s2_encode_better(d)), (kind, n, len(b), len(oracle.s2_encode_better(d)))
    for u in corpora.edge_units():
        assert oracle.s2_decode(oracle.s2_encode_best(u), len(u) + 8) == u
    for u in corpora.stress_units(seed=77, n=40):
        assert oracle.s2_decode(oracle.s2_encode_best(u), len(u) + 8) == u


def test_s2_encode_snappy_best_restatement_is_snappy_and_roundtrips(oracle):
    """s2.EncodeSnappyBest restatement (encodeBlockBestSnappy; oracle only): strict Snappy blocks that round-trip, not larger than
    s2.EncodeSnappyBetter's on compressible corpora."""
    import corpora
    for kind in "TJMH":
        for n in (65536, 65537, 300000):
            d = corpora.corpus(kind, 1, n).tobytes()
            b = oracle.s2_encode_snappy_best(d)
            assert _snappy_decode_strict(b) == d, (kind, n)
            assert oracle.s2_decode(b, n + 8) == d
            assert len(b) <= oracle.lib().kco_s2_max_encoded_len(n)
            if kind in "TJ":
                assert len(b) <= len(oracle.s2_encode_snappy_better(d)), (kind, n, len(b), len(oracle.s2_encode_snappy_better(d)))
    for u in corpora.edge_units() + corpora.stress_units(seed=78, n=40):
        assert _snappy_decode_strict(oracle.s2_encode_snappy_best(u)) == u
