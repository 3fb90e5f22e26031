"""Inputs that reach the speculation re-run of the device path (kc_batch.cpp batch_end).

A block whose sequences look worth coding (saved >= 16) but whose coded form ends no smaller than the block is re-emitted raw
AFTER entropy coding and its repeat offsets are popped (zstd/blockenc.go:811-817).  The device match finder has by then parsed
the following blocks with the un-popped offsets, so the unit is parsed again with that verdict forced.  None of the synthetic
corpora, stress mixes or reference inputs reaches this (the oracle counts it: 0 everywhere); these streams do: 64 KiB of noise,
6 MiB of zeros (which insert almost nothing into the hash tables, so the noise stays findable at offsets of ~2^23), then small
blocks (Flush every few hundred bytes) of noise with 5-7 byte snippets of the first region: every sequence costs ~4.7 bytes for
a 5-7 byte match.
"""
import ctypes as C

import numpy as np
import pytest

import corpora

CASES = [(2, 250, 30, 5), (3, 400, 20, 6), (3, 250, 30, 7)]  # level, block bytes, snippets per block, snippet length


def _make(seed, rlen, zlen, nblk, blk, nsnip, mlen):
    rng = np.random.default_rng(seed)
    R = rng.integers(0, 256, rlen, dtype=np.uint8)
    parts = [R, np.zeros(zlen, dtype=np.uint8)]
    cuts = [rlen + zlen]
    pos = rlen + zlen
    for _ in range(nblk):
        blkb = rng.integers(0, 256, blk, dtype=np.uint8)
        gap = max(1, (blk - 40) // max(nsnip, 1))
        p = 10
        for _k in range(nsnip):
            src = int(rng.integers(0, rlen - mlen - 1))
            if p + mlen + 2 >= blk:
                break
            blkb[p:p + mlen] = R[src:src + mlen]
            p += mlen + int(rng.integers(max(1, gap - mlen - 3), gap + 3))
        parts.append(blkb)
        pos += blk
        cuts.append(pos)
    return np.concatenate(parts), cuts


def _late_raw_pops(oracle, reset=True):
    L = oracle.lib()
    L.kco_debug_late_raw_pops.restype = C.c_uint64
    L.kco_debug_late_raw_pops.argtypes = [C.c_int]
    return L.kco_debug_late_raw_pops(1 if reset else 0)


@pytest.mark.parametrize("level,blk,nsnip,mlen", CASES)
def test_inputs_reach_the_late_raw_fallback(oracle, level, blk, nsnip, mlen):
    """CPU: the oracle (which pops the offsets in place, like the reference) reports a late raw fallback on a non-last block with
    changed offsets for each case, and none for an ordinary stream; the frames decode."""
    d, cuts = _make(7, 1 << 16, 6 << 20, 30, blk, nsnip, mlen)
    e = oracle.ZstdOracle(level=level)
    _late_raw_pops(oracle)
    fr = e.encode_stream(d.tobytes(), cuts)
    assert _late_raw_pops(oracle) >= 1
    assert oracle.zstd_decompress(fr, len(d) + 16) == d.tobytes()
    e.encode_stream(corpora.corpus("T", 4, 131072).tobytes(), [1000, 200000])
    assert _late_raw_pops(oracle) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("level,blk,nsnip,mlen", CASES)
def test_device_re_run_path_bit_exact(oracle, kclib, level, blk, nsnip, mlen):
    """GPU: the streams go through kc_zstd_encode_streams_cuts, the batch takes the re-run (redo_units >= 1) and the frames equal
    the oracle's — per-block re-run flags, irregular blocks and units of more than 32 blocks in one test."""
    from compress_amd import zstd
    d, cuts = _make(7, 1 << 16, 6 << 20, 30, blk, nsnip, mlen)
    t = corpora.corpus("T", 3, 131072).tobytes()
    units = [d.tobytes(), t, d.tobytes()[:len(d) - 3 * blk]]
    flush = [cuts, [1000], [c for c in cuts if c <= len(d) - 3 * blk]]
    ubuf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(level))
    out, out_off = enc.EncodeStreams(ubuf, off, flush_at=flush)
    redo = enc.ctx().timings()["redo_units"]
    ref = oracle.ZstdOracle(level=level)
    for i, u in enumerate(units):
        assert out[int(out_off[i]):int(out_off[i + 1])].tobytes() == ref.encode_stream(u, flush[i]), (i, len(u))
    assert redo >= 1, "the batch did not take the re-run path"
    enc.Close()
