"""oracle/_ref: the reference's own amd64 S2 encoders, run here (tests/oracle_ref.py) — what they pin.

1. The translation is faithful enough to be trusted: every stream they write decodes back to the input through the in-repo S2
   decoder (itself pinned by the reference's golden byte strings, tests/golden/kats.json), at every level and in every size class of
   s2/encode_amd64.go.
2. The oracle's emit helpers (restated from s2/encode_go.go) write exactly the bytes the assembly's emitLiteral / emitRepeat /
   emitCopy / emitCopyNoRepeat write, and matchLen agrees."""
import numpy as np
import pytest

import corpora
import oracle_ref

pytestmark = pytest.mark.skipif(not oracle_ref.available(), reason="oracle/_ref needs an x86-64 host and the reference sources or the built library")


@pytest.mark.parametrize("kind", ["J", "T", "M", "H"])
def test_reference_asm_streams_decode(oracle, kind):
    d = corpora.corpus(kind, 48, 131072).tobytes()
    for n in (1, 31, 32, 33, 100, 511, 512, 513, 4095, 4096, 16383, 16384, 65535, 65536, 65537, 300000, 1 << 20, (4 << 20) - 1, 4 << 20, (4 << 20) + 1):
        u = d[:n]
        for level in range(4):
            e = oracle_ref.encode(u, level)
            assert oracle.s2_decode(e, len(u) + 16) == u, (kind, n, level)
    assert oracle_ref.encode(b"", 0) == b"\x00"


def test_reference_asm_regression_inputs_decode(oracle):
    import os
    import zipfile
    z = zipfile.ZipFile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs", "enc_regressions.zip"))
    for name in z.namelist():
        u = z.read(name)
        if 0 < len(u) <= (4 << 20):
            for level in range(4):
                assert oracle.s2_decode(oracle_ref.encode(u, level), len(u) + 16) == u, (name, level)


def test_oracle_emit_helpers_equal_the_assembly(oracle):
    """kco_s2_emit_literal / emit_copy / emit_repeat (the restatements of s2/encode_go.go:80-310 every oracle encoder and every
    device kernel uses) against the assembly's stand-alone emitLiteral / emitCopy / emitRepeat, over the length and offset classes
    of the format."""
    rng = np.random.default_rng(1)
    for n in list(range(1, 70)) + [255, 256, 257, 65535, 65536, 65537, 70000, (1 << 24) + 5]:
        lit = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert oracle.s2_emit_literal(lit) == oracle_ref.emit_literal(lit), n
    lengths = list(range(4, 80)) + [255, 256, 259, 260, 263, 264, 265, 1000, 65535, 65536, 65539, 65540, 65800, 100000, (1 << 24) + 100]
    offsets = [1, 2, 7, 255, 256, 1023, 1024, 2047, 2048, 2049, 65535, 65536, 65537, 100000, (4 << 20) - 1]
    for off in offsets:
        for ln in lengths:
            assert oracle.s2_emit_copy(off, ln) == oracle_ref.emit("copy", off, ln), (off, ln)
            assert oracle.s2_emit_repeat(off, ln) == oracle_ref.emit("repeat", off, ln), (off, ln)
    for _ in range(300):
        n = int(rng.integers(0, 300))
        a = bytes(rng.integers(0, 3, n, dtype=np.uint8))
        b = bytearray(a + bytes(rng.integers(0, 3, 40, dtype=np.uint8)))
        k = int(rng.integers(0, n + 1))
        if k < n:
            b[k] ^= 0x40
        want = next((i for i in range(n) if a[i] != b[i]), n)
        assert oracle_ref.match_len(a, bytes(b)) == want


def test_zstd_matchlen_equals_the_reference_assembly(oracle):
    """a8: the oracle's matchLen (what every zstd match finder extends matches with; the device's grp_matchlen / wave_matchlen are held
    to it through the parse tests) == matchLen of zstd/matchlen_amd64.s, assembled into oracle/_ref: lengths through the 8-, 4-, 2- and
    1-byte tails, first difference at every position."""
    rng = np.random.default_rng(5)
    L = oracle.lib()
    for _ in range(600):
        n = int(rng.integers(0, 200))
        a = bytes(rng.integers(0, 3, n, dtype=np.uint8))
        b = bytearray(a + bytes(rng.integers(0, 3, 40, dtype=np.uint8)))
        k = int(rng.integers(0, n + 1))
        if k < n:
            b[k] ^= 0x40
        want = next((i for i in range(n) if a[i] != b[i]), n)
        assert oracle_ref.zstd_match_len(a, bytes(b)) == want
        assert L.kco_zstd_matchlen(a, n, bytes(b)) == want


def test_xxh64_equals_the_reference_assembly(oracle):
    """a15: the oracle's XXH64 == xxhash.Sum64 of the reference's amd64 assembly (zstd/internal/xxhash/xxhash_amd64.s, assembled
    into oracle/_ref like the S2 encoders), at every length through the 32-byte stripe, 8-, 4- and 1-byte tails."""
    rng = np.random.default_rng(3)
    assert oracle_ref.xxh64(b"") == 0xEF46DB3751D8E999
    for n in list(range(0, 200)) + [255, 256, 1000, 4095, 65536, 131072, 1000003]:
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert oracle_ref.xxh64(b) == oracle.lib().kco_xxh64(b, n), n


@pytest.mark.gpu
def test_device_xxh64_equals_the_reference_assembly(oracle, kclib):
    """The device's XXH64 kernel (frame content checksums) against the reference's assembly on ragged units."""
    torch = pytest.importorskip("torch")
    from compress_amd import zstd
    rng = np.random.default_rng(5)
    units = [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in list(range(0, 70)) + [int(x) for x in rng.integers(70, 300000, 60)]]
    buf, off = corpora.pack_units(units)
    d = torch.from_numpy(buf).cuda()
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1))
    ctx = enc.ctx()
    out = np.zeros(len(units), dtype=np.uint64)
    ctx.check(ctx.L.kc_xxh64_units_dev(ctx.h, d.data_ptr(), off.ctypes.data, len(units), out.ctypes.data))
    for i, u in enumerate(units):
        assert int(out[i]) == oracle_ref.xxh64(u), (i, len(u))
    enc.Close()


def test_reference_decoder_accepts_every_oracle_level(oracle):
    """s2.Decode of the reference itself (its assembly block decoder, oracle/_ref) decodes what the oracle's encoders write — all six
    levels in their portable-Go form and the four assembly forms — back to the input; and on damaged blocks the in-repo decoder
    (the verifier of the other tests) and the reference's agree: both refuse, or both return the same bytes."""
    rng = np.random.default_rng(2)
    for kind in "JTMH":
        d = corpora.corpus(kind, 12, 131072).tobytes()
        for n in (0, 1, 31, 32, 100, 511, 5000, 65536, 300000, 1 << 20):
            u = d[:n]
            encs = [oracle.s2_encode(u), oracle.s2_encode_better(u), oracle.s2_encode_snappy(u), oracle.s2_encode_snappy_better(u),
                    oracle.s2_encode_best(u), oracle.s2_encode_snappy_best(u)]
            encs += [oracle.s2_encode_asm(u, snappy=l >= 2, better=bool(l & 1)) for l in range(4)]
            for i, e in enumerate(encs):
                assert oracle_ref.decode(e, len(u)) == u, (kind, n, i)
                if len(e) > 8:
                    for _ in range(3):
                        bad = bytearray(e)
                        bad[int(rng.integers(1, len(e)))] ^= int(rng.integers(1, 256))
                        a = oracle_ref.decode(bytes(bad), len(u))
                        try:
                            b = oracle.s2_decode(bytes(bad), len(u) + 16)
                        except RuntimeError:
                            b = None
                        assert (a is None) == (b is None) and (a is None or a == b), (kind, n, i)


@pytest.mark.gpu
def test_reference_decoder_accepts_every_device_level(oracle, kclib):
    """The same from the other side: blocks the DEVICE writes, at every level and in both variants, decoded by the reference's own
    decoder; and the device decoder's verdict on damaged blocks equals the reference decoder's."""
    torch = pytest.importorskip("torch")
    from compress_amd import s2
    rng = np.random.default_rng(4)
    blocks = []
    for kind in "JTMH":
        d = corpora.corpus(kind, 6, 131072).tobytes()
        blocks += [d[:n] for n in (32, 100, 511, 5000, 65536, 150000)]
    b2, off = corpora.pack_units(blocks)
    for level in range(6):
        for variant in ((None, "amd64") if level < 4 else (None,)):
            enc = s2.BlockEncoder(level=level, variant=variant)
            out, oo = enc.EncodeBlocks(b2, off)
            for i, u in enumerate(blocks):
                assert oracle_ref.decode(out[int(oo[i]):int(oo[i + 1])].tobytes(), len(u)) == u, (level, variant, i)
            enc.Close()
    # damaged blocks: device decoder status vs the reference decoder
    enc = s2.BlockEncoder(level=0)
    out, oo = enc.EncodeBlocks(b2, off)
    dam = out.copy()
    for i in range(len(blocks)):
        a, b = int(oo[i]), int(oo[i + 1])
        if b - a > 8 and i % 2 == 0:
            dam[a + int(rng.integers(1, b - a))] ^= np.uint8(rng.integers(1, 256))
    d_enc = torch.from_numpy(dam).cuda()
    d_dst = torch.zeros(len(b2) + 64, dtype=torch.uint8, device="cuda")
    st = enc.DecodeBlocksDevice(d_enc.data_ptr(), oo, d_dst.data_ptr(), off)
    back = d_dst.cpu().numpy()
    for i, u in enumerate(blocks):
        ref = oracle_ref.decode(dam[int(oo[i]):int(oo[i + 1])].tobytes(), len(u))
        ok_ref = ref is not None and len(ref) == len(u)
        assert (st[i] == 0) == ok_ref, (i, int(st[i]), ok_ref)
        if ok_ref:
            assert back[int(off[i]):int(off[i + 1])].tobytes() == ref, i
    enc.Close()


def _pin_inputs():
    import os
    import zipfile
    rng = np.random.default_rng(7)
    out = []
    for kind in "JTMH":
        d = corpora.corpus(kind, 40, 131072).tobytes()
        for n in [32, 33, 100, 511, 512, 513, 2000, 4095, 4096, 5000, 16383, 16384, 20000, 65535, 65536, 65537, 300000, 1 << 20, (4 << 20) - 1, 4 << 20] + [
                int(x) for x in rng.integers(32, 70000, 40)]:
            start = int(rng.integers(0, len(d) - n)) if n < len(d) else 0
            out.append(d[start:start + n])
    z = zipfile.ZipFile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs", "enc_regressions.zip"))
    out += [z.read(n) for n in z.namelist() if 0 < len(z.read(n)) <= (4 << 20)]
    out += [u for u in corpora.edge_units() + corpora.stress_units(seed=5, n=8) if u]
    for _ in range(200):  # low-entropy noise: repeats of every length, the 8B encoder's three-byte repeat form among them
        out.append(bytes(rng.integers(0, int(rng.integers(2, 6)), int(rng.integers(32, 3000)), dtype=np.uint8)))
    return out


def test_oracle_restatement_of_the_assembly_is_pinned(oracle):
    """oracle/kco_s2_asm.h (the assembly encoders restated from their generator) == the assembly itself, byte for byte, for
    s2.Encode, s2.EncodeBetter, s2.EncodeSnappy and s2.EncodeSnappyBetter: every size class of encode_amd64.go, the reference's regression inputs, edge and stress units,
    low-entropy noise.  This is the one whole-encoder path of the oracle that is PINNED by running the reference."""
    bad = []
    ins = _pin_inputs()
    for i, u in enumerate(ins):
        for lvl in range(4):  # s2.Encode, EncodeBetter, EncodeSnappy, EncodeSnappyBetter
            if oracle.s2_encode_asm(u, snappy=lvl >= 2, better=bool(lvl & 1)) != oracle_ref.encode(u, lvl):
                bad.append((i, len(u), lvl))
    assert not bad, bad[:10]
    assert len(ins) > 500


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 3])
def test_device_amd64_variant_better_levels_equal_the_assembly(oracle, kclib, level):
    """The same for s2.EncodeBetter (1) and s2.EncodeSnappyBetter (3)."""
    pytest.importorskip("torch")
    from compress_amd import s2
    ins = _pin_inputs()
    b2, off = corpora.pack_units(ins)
    enc = s2.BlockEncoder(level=level, variant="amd64")
    out, out_off = enc.EncodeBlocks(b2, off)
    bad = []
    for i, u in enumerate(ins):
        got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        if got != oracle_ref.encode(u, level):
            bad.append((i, len(u), len(got), len(oracle_ref.encode(u, level))))
    assert not bad, bad[:10]
    enc.Close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["hbm", "lds"])
@pytest.mark.parametrize("snappy", [False, True])
def test_device_amd64_variant_equals_the_assembly(oracle, kclib, snappy, path):
    """KC_S2_VARIANT_AMD64 on the device == the reference's assembly encoders (and the oracle's restatement of them), on the same
    inputs: bytes produced by hand-written HIP against bytes produced by the reference's own code."""
    pytest.importorskip("torch")
    from compress_amd import s2
    ins = _pin_inputs()
    b2, off = corpora.pack_units(ins)
    enc = s2.BlockEncoder(level=s2.LevelSnappy if snappy else s2.LevelDefault, variant="amd64", path=path)
    out, out_off = enc.EncodeBlocks(b2, off)
    assert enc._ctx.last_path() == path
    bad = []
    for i, u in enumerate(ins):
        got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        want = oracle_ref.encode(u, 2 if snappy else 0)
        assert oracle.s2_encode_asm(u, snappy) == want
        if got != want:
            bad.append((i, len(u), len(got), len(want)))
    assert not bad, bad[:10]
    enc.Close()
