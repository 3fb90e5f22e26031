"""WithEncoderPadding / WithEncoderDictDelete of the zstd façade: the host-side pieces, without a GPU.

The padding is a skippable frame of random bytes behind the frame (zstd/encoder.go:829-837, frameenc.go:100-137): what can be checked
is its arithmetic (calcSkippableFrame), its header, that the frame in front of it is untouched, and that a decoder skips it."""
import numpy as np
import pytest

import oracle_lib
from compress_amd import zstd


def _ref_calc(written, mult):
    """calcSkippableFrame restated from zstd/frameenc.go:100-116 a second time, the slow way: the smallest total >= 8 (or 0) that makes
    written + total a multiple."""
    if written % mult == 0:
        return 0
    t = mult - written % mult
    while t < 8:
        t += mult
    return t


@pytest.mark.parametrize("mult", [2, 3, 7, 8, 9, 16, 100, 4096, 1 << 20])
def test_calc_skippable_frame(mult):
    for written in list(range(0, 40)) + [mult - 1, mult, mult + 1, 3 * mult - 7 if 3 * mult > 7 else 1, 123456789]:
        add = zstd.calc_skippable_frame(written, mult)
        assert add == _ref_calc(written, mult)
        assert (written + add) % mult == 0 and (add == 0 or add >= 8)
    with pytest.raises(ValueError):
        zstd.calc_skippable_frame(5, 0)
    with pytest.raises(ValueError):
        zstd.calc_skippable_frame(-1, 8)


def test_skippable_frame_layout():
    assert zstd.skippable_frame(0) == b""
    with pytest.raises(ValueError):
        zstd.skippable_frame(7)
    for total in (8, 9, 23, 4096):
        f = zstd.skippable_frame(total, rand=lambda n: b"\xEE" * n)
        assert len(f) == total and f[:4] == bytes([0x50, 0x2A, 0x4D, 0x18])
        assert int.from_bytes(f[4:8], "little") == total - 8 and f[8:] == b"\xEE" * (total - 8)


def test_option_validation_like_the_reference():
    for bad in (0, -3, (1 << 30) + 1):
        with pytest.raises(ValueError):
            zstd.WithEncoderPadding(bad)
    assert zstd.WithEncoderPadding(1)._kc_pad == 0  # "No need to waste our time." (encoder_options.go:148-151)
    assert zstd.WithEncoderPadding(1 << 30)._kc_pad == 1 << 30


def test_pad_frames_keeps_the_frames_and_decoders_skip_the_padding():
    """Real frames (the oracle's) padded the way the façade pads a batch: every frame is found unchanged in front of its skippable
    frame, every padded frame is a multiple of the padding, empty frames stay empty, and the oracle's decoder — which follows the
    reference's in skipping skippable frames — reads the concatenation back."""
    rng = np.random.default_rng(7)
    units = [b"", b"a", bytes(rng.integers(0, 4, 3000, dtype=np.uint8)), b"hello world " * 500, bytes(rng.integers(0, 256, 70000, dtype=np.uint8))]
    ref = oracle_lib.ZstdOracle(level=1, full_zero=False)
    frames = [ref.encode_all(u) for u in units]
    out = np.frombuffer(b"".join(frames), dtype=np.uint8)
    off = np.cumsum([0] + [len(f) for f in frames]).astype(np.uint64)
    for pad in (2, 13, 64, 1000):
        pout, poff = zstd.pad_frames(out, off, pad, rand=lambda n: b"\x5A" * n)
        assert len(poff) == len(off)
        for i, f in enumerate(frames):
            got = pout[int(poff[i]):int(poff[i + 1])].tobytes()
            if not f:
                assert got == b""
                continue
            assert got.startswith(f) and len(got) % pad == 0
            tail = got[len(f):]
            assert tail == zstd.skippable_frame(len(tail), rand=lambda n: b"\x5A" * n)
        assert oracle_lib.zstd_decode(pout.tobytes(), sum(map(len, units)) + 64) == b"".join(units)


def test_dict_delete_restores_the_default_fields():
    from compress_amd import _lib
    import ctypes as C
    o = _lib.ZstdOpts()
    d = _lib.ZstdOpts()
    L = _lib.load()
    L.kc_zstd_opts_default(C.byref(o))
    L.kc_zstd_opts_default(C.byref(d))
    content = bytes(range(256)) * 40
    zstd.WithEncoderDictRaw(77, content)(o)
    assert o.dict_len == len(content) and o.dict_id == 77
    zstd.WithEncoderDictDelete()(o)
    for f in ("dict_id", "dict_len", "dict_huf_len", "dict_huf_log"):
        assert getattr(o, f) == getattr(d, f), f
    assert not o.dict and list(o.dict_offsets) == list(d.dict_offsets)
