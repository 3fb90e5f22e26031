"""The reference's OWN encoder test inputs through the device's SpeedFastest pipeline on the wave emulator (no GPU): C1 of BASELINE.json
(testdata/e.txt) with its frame boundaries, the reference's shared testdata files, and a slice of its fuzz / regression corpora
(zstd/encoder_test.go TestEncoderRegression, zstd/fuzz_test.go FuzzEncoding seeds) — the frames the device code writes equal the
oracle's.  tests/test_gpu_ref_inputs.py runs all of them (and the other levels) on the device."""
import os
import zipfile

import emu_lib
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
REFIN = os.path.join(HERE, "golden", "ref_inputs")
PLAIN = ["e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "sharnd.out", "pi.txt", "html.txt", "pngdata.bin", "z000028"]


def _zip_inputs(name, every, max_len):
    z = zipfile.ZipFile(os.path.join(REFIN, name))
    names = [n for n in z.namelist() if not n.endswith("/")]
    return [(name + ":" + n, z.read(n)) for n in names[::every] if z.getinfo(n).file_size <= max_len]


def _check(named, finder="lds", level=1):
    ref = oracle_lib.ZstdOracle(level=level)
    units = [d for _, d in named]
    frames, err, redo = emu_lib.zstd_frames(units, use_grp=finder != "lds", tuned=int(finder == "grp-tuned"), max_encoded_size=ref.max_encoded_size,
                                            level=level)
    assert err == 0
    # a unit that asks for the speculation re-run is the host's business (tests/test_redo_path.py): none of these may
    assert redo == 0
    bad = [(n, len(d), len(f)) for (n, d), f in zip(named, frames) if f != ref.encode_all(d)]
    assert not bad, "inputs whose emulated frame differs from the oracle's (name, length, frame length): %r" % bad[:8]
    return len(named)


def test_c1_e_txt_on_the_emulator():
    e = open(os.path.join(REFIN, "e.txt"), "rb").read()
    ref = oracle_lib.ZstdOracle(level=1)
    frames, err, redo = emu_lib.zstd_frames([e], max_encoded_size=ref.max_encoded_size)
    assert err == 0 and redo == 0 and frames[0] == ref.encode_all(e)
    assert frames[0][:9].hex() == "28b52ffda4a3860100" and frames[0][-4:].hex() == "5f0c047d"  # frame header + XXH64 low word of e.txt


def test_reference_plain_files_on_the_emulator():
    named = [(n, open(os.path.join(REFIN, n), "rb").read()) for n in PLAIN]
    named = [(n, d) for n, d in named if len(d) <= 512 << 10]
    assert _check(named) >= 6
    assert _check(named[:3], finder="grp") == 3


def test_reference_fuzz_and_regression_corpora_on_the_emulator():
    named = _zip_inputs("encode-corpus-raw.zip", 9, 200000) + _zip_inputs("comp-crashers.zip", 9, 200000)
    assert _check(named) > 300
    assert _check(named[::4], finder="grp-tuned") > 80


def test_reference_inputs_at_the_other_levels_on_the_emulator():
    """SpeedDefault, SpeedBetterCompression and SpeedBestCompression on the reference's files and a slice of its corpora."""
    plain = [(n, open(os.path.join(REFIN, n), "rb").read()) for n in PLAIN]
    plain = [(n, d) for n, d in plain if len(d) <= 160 << 10]
    corp = _zip_inputs("encode-corpus-raw.zip", 23, 100000) + _zip_inputs("comp-crashers.zip", 23, 100000)
    for level in (2, 3):
        assert _check(plain + corp, level=level) > 100
    assert _check(plain[:2] + corp[::6], level=4) > 20
