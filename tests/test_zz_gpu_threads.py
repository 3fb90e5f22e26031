"""One Python Encoder hammered from threads on the device (sorted last in the suite on purpose).  (*Encoder).EncodeAll "can be
called concurrently" in the reference (zstd/encoder.go:717); the façade serves concurrent callers with one kc_ctx each
(Encoder._held) — the same arrangement as the Go shim's context pool — so this is also the test of several contexts working
on one GPU at the same time from different host threads.  Every frame is compared with the oracle's."""
import threading

import numpy as np
import pytest

import corpora

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level", [1, 3])
def test_one_encoder_many_threads_bit_exact(oracle, kclib, level):
    from compress_amd import zstd
    buf = corpora.corpus("T", 8, 65536).tobytes() + corpora.corpus("M", 8, 65536).tobytes()
    sizes = [1, 700, 4096, 65536, 65537, 131072, 200000, 300001]
    inputs = [buf[(37 * k) % 1000:][:n] for k, n in enumerate(sizes)]
    want = [oracle.ZstdOracle(level=level).encode_all(d) for d in inputs]
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(level), zstd.WithEncoderConcurrency(4))
    got, errs = {}, []

    def worker(t):
        try:
            for r in range(3):
                for k in range(len(inputs)):
                    j = (k + t) % len(inputs)
                    got[(t, r, j)] = enc.EncodeAll(inputs[j])
        except BaseException as e:
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert len(got) == 8 * 3 * len(inputs)
    for (t, r, j), out in got.items():
        assert out == want[j], (level, t, r, j)
    assert enc.EncodeAll(inputs[3]) == want[3]  # the encoder's own context still serves a lone caller
    enc.Close()


def test_one_s2_block_encoder_many_threads_bit_exact(oracle, kclib):
    """s2.Encode is a pure function in the reference; one BlockEncoder serves concurrent callers one batch at a time (the
    batched calls share the context's stream and scratch)."""
    from compress_amd import s2
    buf = corpora.corpus("J", 8, 65536).tobytes()
    inputs = [buf[(91 * k) % 500:][:n] for k, n in enumerate([1, 31, 32, 700, 4096, 65535, 65536, 65537, 200000])]
    want = [oracle.s2_encode(d) for d in inputs]
    enc = s2.BlockEncoder()
    got, errs = {}, []

    def worker(t):
        try:
            for r in range(3):
                for k in range(len(inputs)):
                    j = (k + t) % len(inputs)
                    got[(t, r, j)] = enc.Encode(None, inputs[j])
        except BaseException as e:
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert len(got) == 8 * 3 * len(inputs)
    for (t, r, j), out in got.items():
        assert out == want[j], (t, r, j)
    enc.Close()
