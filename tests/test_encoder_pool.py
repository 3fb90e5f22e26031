"""The Python Encoder's context pool (compress_amd/zstd.py: Encoder._held) without a GPU: EncodeAll is concurrency-safe on ONE
encoder in the reference (zstd/encoder.go:717, the e.encoders channel of encoder.go:90-99) — here every concurrent caller must be
inside the library with a kc_ctx of its own, a lone caller must always get the encoder's own context, and Close releases all of
them.  The library call is replaced by a stand-in that records which context it was entered with."""
import threading
import time

import numpy as np

from compress_amd import _lib, zstd


class _FakeLib:
    def __init__(self, log):
        self.log = log

    def kc_zstd_encode_units(self, h, opts, src, unit_off, n, dst, cap, out_off):
        with self.log["m"]:
            assert h not in self.log["inside"], "two callers inside the library with one context"
            self.log["inside"].add(h)
            self.log["peak"] = max(self.log["peak"], len(self.log["inside"]))
            self.log["used"].append(h)
        time.sleep(0.02)
        with self.log["m"]:
            self.log["inside"].discard(h)
        return _lib.KC_OK


def _patch(monkeypatch):
    log = {"m": threading.Lock(), "inside": set(), "peak": 0, "used": [], "made": 0, "closed": 0, "paths": []}

    class FakeContext:
        def __init__(self, device=0, stream=None):
            with log["m"]:
                log["made"] += 1
                self.h = log["made"]
            self.L = _FakeLib(log)

        def set_path(self, p):
            log["paths"].append((self.h, p))

        def check(self, st):
            assert st == _lib.KC_OK

        def close(self):
            if self.h is not None:
                with log["m"]:
                    log["closed"] += 1
                self.h = None

    monkeypatch.setattr(_lib, "Context", FakeContext)
    return log


def test_lone_caller_always_gets_the_encoders_own_context(kclib, monkeypatch):
    log = _patch(monkeypatch)
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(zstd.SpeedFastest), path="lds")
    own = enc.ctx()
    for _ in range(5):
        enc.EncodeAll(b"abc")
        enc.EncodeUnits(np.zeros(8, np.uint8), np.array([0, 4, 8], np.uint64))
    assert log["made"] == 1 and set(log["used"]) == {own.h} and log["paths"] == [(own.h, "lds")]
    enc.Close()
    assert log["closed"] == 1


def test_concurrent_encodeall_takes_one_context_per_caller(kclib, monkeypatch):
    log = _patch(monkeypatch)
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(zstd.SpeedFastest), zstd.WithEncoderConcurrency(4), path="hbm")
    errs = []

    def worker():
        try:
            for _ in range(6):
                enc.EncodeAll(b"x" * 100)
        except BaseException as e:  # the stand-in asserts on a shared context
            errs.append(e)

    ts = [threading.Thread(target=worker) for _ in range(12)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert len(log["used"]) == 72
    assert log["peak"] > 1, "the callers never overlapped: the test did not test anything"
    assert log["made"] <= 12
    assert all(p == "hbm" for _, p in log["paths"]) and len(log["paths"]) == log["made"]
    # at most WithEncoderConcurrency spare contexts are kept; the rest were closed when their call ended
    assert len(enc._spare) <= 4 and log["made"] - log["closed"] == 1 + len(enc._spare)
    enc.Close()
    assert log["closed"] == log["made"] and enc._spare == []
    # usable again after Close, like the reference's encoder after Reset
    enc.EncodeAll(b"y")
    assert log["made"] - log["closed"] == 1
