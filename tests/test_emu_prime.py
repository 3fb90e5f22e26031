"""kc_zstd_prime_kernel (compress_amd/csrc/kc_zstd_prime.hip) on the wave emulator: the tables of a WithConcurrentBlocks job as the
reference's ResetPrefix leaves them, against a position-by-position restatement of the reference's loops (zstd/enc_fast.go:800-811,
zstd/enc_dfast.go:1040-1050, zstd/enc_better.go:1099-1112) in the device entry format ((position + 1) | tag << pos_bits)."""
import numpy as np
import pytest

import corpora
import emu_lib

M64 = (1 << 64) - 1
PRIME8 = 0xcf1bbcdcb7a56463
PRIME6 = 227718039650203
PRIME5 = 889523592379


def _fmt(pos_bits):
    tb = min(16, 32 - pos_bits)

    def mk(pos, val):
        tag = ((val * 2654435761) & 0xFFFFFFFF) >> (32 - tb) if tb > 0 else 0
        return ((pos + 1) | (tag << pos_bits)) & 0xFFFFFFFF
    return mk


def reset_prefix_tables(level, prefix, pos_bits):
    """The reference's loops, one insert at a time."""
    mk = _fmt(pos_bits)
    n = len(prefix)
    words = {1: 1 << 15, 2: (1 << 17) + (1 << 15), 3: (2 << 19) + (1 << 13)}[level]
    out = np.zeros(words, dtype=np.uint32)
    if n < 8:
        return out
    end = n - 8
    ld = lambda i: int.from_bytes(prefix[i:i + 8], "little")
    if level == 3:  # enc_better.go:1099-1112
        for i in range(0, end, 2):
            cv = ld(i)
            h = ((cv * PRIME8) & M64) >> (64 - 19)
            out[2 * h + 1] = out[2 * h]
            out[2 * h] = mk(i, cv & 0xFFFFFFFF)
            v = cv >> 8
            hs = ((((v << 24) & M64) * PRIME5) & M64) >> (64 - 13)
            out[(2 << 19) + hs] = mk(i + 1, v & 0xFFFFFFFF)
        return out
    f0 = (1 << 17) if level == 2 else 0
    for i in range(1, end, 4):  # enc_fast.go:800-811
        cv = ld(i)
        out[f0 + (((((cv << 16) & M64) * PRIME6) & M64) >> (64 - 15))] = mk(i, cv & 0xFFFFFFFF)
    if level == 2:  # enc_dfast.go:1042-1050
        for i in range(1, end, 2):
            cv = ld(i)
            out[((cv * PRIME8) & M64) >> (64 - 17)] = mk(i, cv & 0xFFFFFFFF)
    return out


def _prefixes():
    t = corpora.corpus("T", 1, 131072, first_unit=3).tobytes()
    m = corpora.corpus("M", 1, 131072, first_unit=5).tobytes()
    rep = (b"abcdefgh" * 40 + b"0123456789abcdef" * 20) * 12     # few distinct 8-byte windows: buckets hit many times per round
    runs = b"\0" * 700 + b"ab" * 900 + t[:300] + b"\0" * 333       # one bucket for whole rounds (the chain's prev = the position before)
    return [t[:20000], m[:9000], rep, runs, t[:8], t[:9], t[:7], b"", t[:264], t[:265], t[:1033], m[:521]]


@pytest.mark.parametrize("level", [1, 2, 3])
@pytest.mark.parametrize("reverse", [False, True])
def test_prime_kernel_equals_reset_prefix(level, reverse):
    """Every prefix of the set, at a position field that leaves 16 tag bits and at one that leaves 3; `reverse`: the emulator keeps
    the lowest lane's value where lanes of one store share an address (the kernel's retry / ordered-insert paths do the work)."""
    pre = _prefixes()
    for pos_bits in (16, 29):
        got = emu_lib.zstd_prime(level, pre, pos_bits, reverse=reverse)
        for i, p in enumerate(pre):
            want = reset_prefix_tables(level, p, pos_bits)
            bad = np.flatnonzero(got[i] != want)
            assert bad.size == 0, "level %d pos_bits %d prefix %d (%d bytes): %d words differ, first at %d" % (level, pos_bits, i, len(p), bad.size, bad[0])


def test_prime_kernel_unit_list_indirection():
    """Re-runs prime slot i from unit list[i]."""
    pre = _prefixes()[:5]
    got = emu_lib.zstd_prime(1, pre, 20, unit_list=[3, 0, 3])
    for slot, u in enumerate((3, 0, 3)):
        assert np.array_equal(got[slot], reset_prefix_tables(1, pre[u], 20))
