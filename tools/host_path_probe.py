#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point kc_zstd_encode_units (never the headline value)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from compress_amd import _lib, zstd
usz = 131072; n = 8192
buf = _lib.corpus_fill("T", 0x5EED0001, 0, n, usz); off = np.arange(n + 1, dtype=np.uint64) * usz
enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1))
for it in range(3):
    t = time.perf_counter(); out, oo = enc.EncodeUnits(buf, off); dt = time.perf_counter() - t
    print("host buffers, 1 GiB: %.1f ms  %.2f GB/s (pageable host memory, H2D + encode + D2H)" % (dt * 1e3, n * usz / dt / 1e9))
