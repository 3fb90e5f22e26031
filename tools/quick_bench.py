#!/usr/bin/env python3
"""Quick device-resident throughput probe (not the contract bench): python tools/quick_bench.py [kind] [GiB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from compress_amd import _lib, zstd
kind = sys.argv[1] if len(sys.argv) > 1 else "T"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
usz = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
n = int(gib * 2**30) // usz
seed = {"T": 0x5EED0001, "H": 0x5EED0002, "J": 0x5EED0003, "M": 0x5EED0004}[kind]
buf = _lib.corpus_fill(kind, seed, 0, n, usz)
d = torch.from_numpy(buf).cuda()
enc = zstd.NewWriter(None, zstd.WithEncoderLevel(int(os.environ.get('KC_LEVEL', '1'))))
off = np.arange(n + 1, dtype=np.uint64) * usz
cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
dd = torch.empty(cap, dtype=torch.uint8, device="cuda")
for it in range(3):
    t = time.time(); oo = enc.EncodeUnitsDevice(d.data_ptr(), off, dd.data_ptr(), cap); torch.cuda.synchronize(); dt = time.time() - t
    tm = enc.ctx().timings()
    print("%s %.2f GiB unit %d: %.3fs %.2f GB/s ratio %.4f match %.1f ms entropy %.1f ms other %.1f ms redo %d" % (
        kind, gib, usz, dt, len(buf) / dt / 1e9, int(oo[n]) / len(buf), tm["match_ms"], tm["entropy_ms"], tm["other_ms"], tm["redo_units"]))
if len(sys.argv) > 4 and sys.argv[4] == "s2":
    from compress_amd import s2
    e2 = s2.BlockEncoder()
    cap2 = n * ((s2.MaxEncodedLen(usz) + 15) & ~15) + 64
    d2 = torch.empty(cap2, dtype=torch.uint8, device="cuda")
    for it in range(3):
        t = time.time(); oo = e2.EncodeBlocksDevice(d.data_ptr(), off, d2.data_ptr(), cap2); torch.cuda.synchronize(); dt = time.time() - t
        tm = e2._ctx.timings()
        print("S2 %s %.2f GiB block %d: %.3fs %.2f GB/s ratio %.4f kernel %.1f ms other %.1f ms" % (kind, gib, usz, dt, len(buf) / dt / 1e9, int(oo[n]) / len(buf), tm["match_ms"], tm["other_ms"]))
