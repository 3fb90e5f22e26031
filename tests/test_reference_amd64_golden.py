"""Gate on the reference's own bytes everywhere: tests/golden/reference_amd64_sha256.txt holds sha256 hashes of what the reference's
amd64 assembly encoders wrote (tools/write_asm_golden.py, run where oracle/_ref can be built) for seeded corpora at the four
assembly-backed S2 levels, and XXH64 values of its assembly.  Unlike tests/test_ref_s2asm.py these need neither the reference
sources nor the built library: the oracle's restatement (CPU) and the device's KC_S2_VARIANT_AMD64 (GPU) must hash to the same."""
import hashlib
import os

import numpy as np
import pytest

import corpora

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "reference_amd64_sha256.txt")
LEVEL = {"s2": 0, "s2better": 1, "s2snappy": 2, "s2snappybetter": 3}


def _lines():
    out = {}
    for l in open(GOLD):
        f = l.split()
        if len(f) == 2 and not l.startswith("#"):
            out[f[0]] = f[1]
    assert len(out) > 100
    return out


def _blocks(kind, n, ln):
    if kind[0] == "A":  # random bytes over 2 / 4 / 8 symbols (tools/write_asm_golden.py)
        return corpora.small_alphabet_blocks(int(kind[1:]), n, ln)
    buf = corpora.corpus(kind, (n * ln + 131071) // 131072, 131072, first_unit=11)
    return buf[:n * ln], np.arange(n + 1, dtype=np.uint64) * ln


def test_oracle_restatement_matches_the_reference_hashes(oracle):
    for name, want in sorted(_lines().items()):
        p = name.split(".")
        if p[0] == "xxh64":
            continue
        n, ln = (int(x) for x in p[3].split("x"))
        buf, off = _blocks(p[2], n, ln)
        lvl = LEVEL[p[0]]
        h = hashlib.sha256()
        for i in range(n):
            h.update(oracle.s2_encode_asm(buf[int(off[i]):int(off[i + 1])].tobytes(), snappy=lvl >= 2, better=bool(lvl & 1)))
        assert h.hexdigest() == want, name
    # XXH64 (the generator draws the inputs in this order from one stream)
    rng = np.random.default_rng(99)
    lines = _lines()
    for n in (0, 1, 3, 4, 7, 8, 31, 32, 33, 63, 64, 1000, 131072):
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert "%016x" % oracle.lib().kco_xxh64(b, n) == lines["xxh64.rand99.%d" % n], n


@pytest.mark.gpu
def test_device_matches_the_reference_hashes(oracle, kclib):
    pytest.importorskip("torch")
    from compress_amd import s2
    encs = {}
    for name, want in sorted(_lines().items()):
        p = name.split(".")
        if p[0] == "xxh64":
            continue
        n, ln = (int(x) for x in p[3].split("x"))
        buf, off = _blocks(p[2], n, ln)
        lvl = LEVEL[p[0]]
        if lvl not in encs:
            encs[lvl] = s2.BlockEncoder(level=lvl, variant="amd64")
        out, oo = encs[lvl].EncodeBlocks(buf, off)
        assert hashlib.sha256(out[:int(oo[n])].tobytes()).hexdigest() == want, name
    for e in encs.values():
        e.Close()


def test_device_kernels_on_the_emulator_match_the_reference_hashes():
    """No GPU, no reference: the DEVICE kernels (compiled for the wave emulator) must hash to what the reference's assembly wrote —
    the HBM-table kernel at the four levels and the LDS-table kernel at s2.Encode / s2.EncodeSnappy, on the small-alphabet blocks of
    every size class and on the short shapes of the corpora."""
    import emu_lib
    checked = 0
    for name, want in sorted(_lines().items()):
        p = name.split(".")
        if p[0] == "xxh64":
            continue
        n, ln = (int(x) for x in p[3].split("x"))
        if n * ln > 100000 or (p[2][0] != "A" and ln > 2000):
            continue  # (emulation time: the long shapes stay with the GPU test)
        buf, off = _blocks(p[2], n, ln)
        blocks = [buf[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)]
        lvl = LEVEL[p[0]]
        got = emu_lib.s2_encode_blocks_hbm(blocks, level=lvl, variant=1)
        assert hashlib.sha256(b"".join(got)).hexdigest() == want, ("HBM-table kernel", name)
        if lvl in (0, 2):
            got = emu_lib.s2_encode_blocks(blocks, level=lvl, variant=1, spec_w0=1 if ln <= 65536 else 8)
            assert hashlib.sha256(b"".join(got)).hexdigest() == want, ("LDS-table kernel", name)
        checked += 1
    assert checked >= 60
