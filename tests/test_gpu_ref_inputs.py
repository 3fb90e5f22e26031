"""GPU parity on the reference's OWN test inputs (committed copies under tests/golden/ref_inputs/, see
tests/golden/fetch_ref_inputs.py): the HIP path must equal the oracle byte for byte on every input the
reference's encoder tests feed their encoders (zstd/encoder_test.go:68-160 TestEncoderRegression,
zstd/fuzz_test.go:154 FuzzEncoding seeds, s2/encode_test.go TestEncoderRegression, testdata/*), at every
level the device path serves, including BASELINE config C1 (testdata/e.txt, SpeedFastest)."""
import os
import zipfile

import numpy as np
import pytest

import corpora

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REFIN = os.path.join(HERE, "golden", "ref_inputs")
PLAIN = ["e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "sharnd.out", "pi.txt", "html.txt", "pngdata.bin", "z000028"]


def _zip_inputs(name, limit=None):
    z = zipfile.ZipFile(os.path.join(REFIN, name))
    out = []
    for n in z.namelist():
        if n.endswith("/"):
            continue
        out.append((name + ":" + n, z.read(n)))
        if limit and len(out) >= limit:
            break
    return out


def _plain_inputs():
    return [(n, open(os.path.join(REFIN, n), "rb").read()) for n in PLAIN]


def _check_zstd(oracle, named, level, max_unit=32 * 65536):
    from compress_amd import zstd
    named = [(n, d) for n, d in named if len(d) <= max_unit]  # keeps the test short: long units are covered by test_long_units_and_streams_bit_exact
    units = [d for _, d in named]
    buf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(level))
    out, out_off = enc.EncodeUnits(buf, off)
    ref, ref_off = oracle.zstd_encode_units(buf, off, threads=8, level=level)
    bad = []
    for i, (n, d) in enumerate(named):
        a = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        b = ref[int(ref_off[i]):int(ref_off[i + 1])].tobytes()
        if a != b:
            bad.append((n, len(d), len(a), len(b)))
    enc.Close()
    assert not bad, "inputs whose GPU frame differs from the oracle (name, in_len, gpu_len, oracle_len): %r" % bad[:8]
    return len(named)


def test_c1_e_txt_speed_fastest(oracle, kclib):
    """BASELINE.json configs[0]: zstd.Encoder SpeedFastest EncodeAll on testdata/e.txt."""
    from compress_amd import zstd
    e = open(os.path.join(REFIN, "e.txt"), "rb").read()
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(zstd.SpeedFastest))
    got = enc.EncodeAll(e)
    ref = oracle.ZstdOracle(level=1).encode_all(e)
    assert got == ref
    assert got[:9].hex() == "28b52ffda4a3860100" and got[-4:].hex() == "5f0c047d"  # frame header + XXH64 low word of e.txt
    assert oracle.zstd_decompress(got, len(e) + 16) == e
    enc.Close()


@pytest.mark.parametrize("level", [1, 2, 3])
def test_reference_plain_files_bit_exact(oracle, kclib, level):
    assert _check_zstd(oracle, _plain_inputs(), level) == len(PLAIN)


@pytest.mark.parametrize("level", [1, 2, 3])
@pytest.mark.parametrize("zf", ["encode-corpus-raw.zip", "comp-crashers.zip"])
def test_reference_zip_corpora_bit_exact(oracle, kclib, zf, level):
    n = _check_zstd(oracle, _zip_inputs(zf), level)
    assert n > 100


def test_s2_enc_regressions_bit_exact(oracle, kclib):
    """s2/testdata/enc_regressions.zip + testdata/* through s2.Encode (block format) on the device vs the oracle."""
    from compress_amd import s2
    named = [(n, d) for n, d in _zip_inputs("enc_regressions.zip") + _plain_inputs() if 0 < len(d) <= (4 << 20)]
    buf, off = corpora.pack_units([d for _, d in named])
    enc = s2.BlockEncoder()
    out, out_off = enc.EncodeBlocks(buf, off)
    bad = []
    for i, (n, d) in enumerate(named):
        got = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
        ref = oracle.s2_encode(d)
        if got != ref:
            bad.append((n, len(d), len(got), len(ref)))
    enc.Close()
    assert not bad, bad[:8]
    assert len(named) > 40
