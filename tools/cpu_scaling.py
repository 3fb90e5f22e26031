#!/usr/bin/env python3
"""CPU-baseline scaling probe: oracle (port) throughput vs thread count on this host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from compress_amd import _lib
os.environ["KC_NO_TORCH_PRELOAD"] = "1"
UNIT = 131072
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
host = _lib.corpus_fill("T", 0x5EED0001, 0, n, UNIT)
off = np.arange(n + 1, dtype=np.uint64) * UNIT
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max: n/a", e)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 8, 32, 64, 128, 256):
    t = time.perf_counter(); r, ro = O.zstd_encode_units(host, off, threads=th, level=1); dt = time.perf_counter() - t
    print("%3d threads: %8.1f MB/s" % (th, n * UNIT / dt / 1e6), flush=True)
