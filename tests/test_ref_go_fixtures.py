"""The translator itself, against vectors the REFERENCE holds (not ones this repository derived): oracle/_ref/libzstdref.so is the
reference's Go source compiled through a translator written here (oracle/ref_go), so the weak link of "the oracle is pinned by the
reference" is that translator and its runtime (gort.h).  The reference ships decode fixtures with known plaintext and known-answer
strings for its S2 emitters; its own translated DECODER and EMITTERS — the same front end, the same runtime: typed wrap-around
integers, slices, shifts — must reproduce them, in all three builds of the decoder (portable Go; amd64 with its assembly, BMI2 on and
off).  zstd/decoder_test.go (testDecoderFile / testDecoderDecodeAll / testDecoderFileBad fixture loops), zstd/dict_test.go:16-100,
s2/s2_test.go:37-76, 827-942.  CPU only; skipped where /root/reference (and with it the built library) is absent."""
import glob
import io
import json
import os
import zipfile

import pytest

import oracle_goref as G

REF = "/root/reference"
TD = os.path.join(REF, "zstd", "testdata")
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))
pytestmark = pytest.mark.skipif(not (os.path.isdir(TD) and G.available()), reason="reference fixtures / oracle/_ref not present on this machine")


def _flavours():
    return [f for f in G.FLAVOURS if f == "generic" or G.amd64_available()]


def _zip(name):
    return zipfile.ZipFile(os.path.join(TD, name))


@pytest.mark.parametrize("fl", _flavours())
def test_translated_decoder_returns_the_reference_held_plaintexts(fl):
    """z000028.zst -> z000028 (decoder_test.go:539-905) and every member of good.zip that travels with its plaintext
    (testDecoderDecodeAll, decoder_test.go:1820-1880: `want[name + ".zst"]` is the member without the suffix)."""
    with G.flavour(fl):
        want = open(os.path.join(TD, "z000028"), "rb").read()
        assert G.zstd_decode_all(open(os.path.join(TD, "z000028.zst"), "rb").read(), len(want)) == want
        zf = _zip("good.zip")
        names = set(zf.namelist())
        n = 0
        for m in sorted(names):
            if m.endswith(".zst") and m[:-4] in names:
                plain = zf.read(m[:-4])
                assert G.zstd_decode_all(zf.read(m), len(plain)) == plain, m
                n += 1
        assert n >= 11


@pytest.mark.parametrize("fl", _flavours())
def test_translated_decoder_on_the_reference_archives(fl, oracle):
    """benchdecoder.zip (12 frames of the classic corpora, each with a content checksum the decoder verifies), the .zst members of
    good.zip without a plaintext beside them, decode-regression.zip: the translated decoder agrees with the system libzstd (same
    bytes, or both refuse) — and, where there is no libzstd, with the hand-written decoder of oracle/."""
    try:
        Z = oracle.libzstd()
    except OSError:
        Z = None
    import ctypes as C

    def libz(z, cap):
        if Z is None:
            return oracle.zstd_decode(z, cap)
        buf = C.create_string_buffer(cap)
        r = Z.ZSTD_decompress(buf, cap, z, len(z))
        if Z.ZSTD_isError(r):
            raise ValueError("libzstd refuses")
        return buf.raw[:r]

    checked = 0
    with G.flavour(fl):
        for arc in ("benchdecoder.zip", "good.zip", "decode-regression.zip"):
            zf = _zip(arc)
            names = set(zf.namelist())
            for m in sorted(names):
                if zf.getinfo(m).is_dir() or (arc == "good.zip" and (not m.endswith(".zst") or m[:-4] in names)):
                    continue
                z = zf.read(m)
                try:
                    got = G.zstd_decode_all(z, 8 << 20)
                except ValueError:
                    got = None
                if arc == "good.zip":
                    # a .zst of good.zip without a plaintext beside it: the reference's own test expects it to decode to nothing
                    # (`wantB := want[tt.Name]` is nil, decoder_test.go:1844-1856) — libzstd refuses that 11-byte frame, the reference does not
                    assert got == b"", (m, got)
                    checked += 1
                    continue
                try:
                    want = libz(z, 8 << 20)
                except (ValueError, RuntimeError):
                    want = None
                if arc == "benchdecoder.zip":
                    assert got is not None and len(got) > 1000, m
                assert got == want, (arc, m, None if got is None else len(got), None if want is None else len(want))
                checked += 1
    assert checked >= 13


@pytest.mark.parametrize("fl", _flavours())
def test_translated_decoder_refuses_the_reference_bad_frames(fl):
    """bad.zip (testDecoderFileBad, decoder_test.go:1135-1200): truncated and corrupted frames — the reference expects an error
    from every member that is a frame; so does its translation (a panic of the translated code would surface as error -1 too, but
    none of these may decode)."""
    zf = _zip("bad.zip")
    n = 0
    with G.flavour(fl):
        for m in sorted(zf.namelist()):
            if not m.endswith(".zst"):
                continue
            with pytest.raises(ValueError):
                G.zstd_decode_all(zf.read(m), 4 << 20)
            n += 1
    assert n >= 30


@pytest.mark.parametrize("fl", _flavours())
def test_translated_decoder_with_the_reference_dictionaries(fl, oracle):
    """dict-tests-small.zip (dict_test.go:16-100): frames that need one of the archive's dictionaries; the translated
    WithDecoderDicts + DecodeAll returns what libzstd's ZSTD_decompress_usingDict returns (the hand decoder without a libzstd)."""
    import ctypes as C
    zf = _zip("dict-tests-small.zip")
    dicts = {}
    for m in zf.namelist():
        if m.endswith(".dict"):
            b = zf.read(m)
            dicts[int.from_bytes(b[4:8], "little")] = b
    assert dicts
    try:
        Z = oracle.libzstd()
    except OSError:
        Z = None
    n = 0
    with G.flavour(fl):
        for m in sorted(zf.namelist()):
            if not m.endswith(".zst"):
                continue
            z = zf.read(m)
            # frame header: magic, descriptor, [window], dictionary id
            fhd = z[4]
            did_len = (0, 1, 2, 4)[fhd & 3]
            pos = 5 + (0 if fhd & 0x20 else 1)
            did = int.from_bytes(z[pos:pos + did_len], "little") if did_len else 0
            if did not in dicts:
                continue
            blob = dicts[did]
            if Z is not None:
                buf = C.create_string_buffer(4 << 20)
                ctx = Z.ZSTD_createDCtx()
                r = Z.ZSTD_decompress_usingDict(ctx, buf, 4 << 20, z, len(z), blob, len(blob))
                Z.ZSTD_freeDCtx(ctx)
                assert not Z.ZSTD_isError(r), m
                want = buf.raw[:r]
            else:
                want = oracle.zstd_decode(z, 4 << 20, dict_blob=blob)
            assert G.zstd_decode_all(z, 4 << 20, dict_blob=blob) == want, m
            n += 1
    assert n >= 8


def test_translated_s2_emitters_give_the_reference_kat_strings():
    """TestEmitLiteral / TestEmitCopy (s2/s2_test.go:827-942): the byte strings the reference's tests hold, from the reference's own
    emitLiteral / emitCopy run through the translator (tests/golden/kats.json holds the strings as hex, copied from that file)."""
    nines = b"\x99" * 65536
    for length, want in KATS["s2_emit_literal"]:
        out = G.s2_emit("literal", 0, 0, lit=nines[:length])
        assert out[len(out) - length:] == nines[:length]
        assert out[:len(out) - length].hex() == want, length
    for offset, length, want in KATS["s2_emit_copy"]:
        assert G.s2_emit("copy", offset, length).hex() == want, (offset, length)
    assert len(KATS["s2_emit_literal"]) >= 13 and len(KATS["s2_emit_copy"]) >= 59


def test_translated_max_encoded_len_gives_the_reference_vectors():
    """TestMaxEncodedLen (s2/s2_test.go:37-76): the table's fixed entries and its formula for every size below maxBlockSize (sampled)."""
    for n, want in KATS["s2_max_encoded_len"]:
        assert G.s2_max_encoded_len(n) == want, n
    for i in list(range(0, 70000, 11)) + [(4 << 20) - 1, 4 << 20, (1 << 24) - 1, 1 << 24]:
        varint = 1
        z = i << 1   # binary.PutVarint zig-zags the value
        while z >= 0x80:
            z >>= 7
            varint += 1
        extra = 0 if i == 0 else 1 if i < 60 else 2 if i < 256 else 3 if i < 65536 else 4 if i < (1 << 24) else 5
        assert G.s2_max_encoded_len(i) == i + varint + extra, i


def test_translated_s2_decode_on_the_reference_golden_snappy_block(oracle):
    """TestDecodeGoldenInput (s2/s2_test.go:598-614): Mark.Twain-Tom.Sawyer.txt.rawsnappy -> Mark.Twain-Tom.Sawyer.txt through the
    reference's own s2.Decode, translated; and the block encoders' round trip through it on that text at all six levels."""
    td = os.path.join(REF, "s2", "testdata")
    want = open(os.path.join(td, "Mark.Twain-Tom.Sawyer.txt"), "rb").read()
    assert G.s2_decode(open(os.path.join(td, "Mark.Twain-Tom.Sawyer.txt.rawsnappy"), "rb").read(), len(want)) == want
    for level in range(6):
        assert G.s2_decode(G.s2_encode(want, level=level), len(want)) == want, level
    with pytest.raises(ValueError):
        G.s2_decode(b"\x05\xff\xff\xff", 64)
