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
