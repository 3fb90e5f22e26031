#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs other than the headline (C3, C4, C5) on one GPU, device resident."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from compress_amd import _lib, zstd, s2

def run(name, fn, nbytes, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return {"config": name, "GB_per_s": round(nbytes / best / 1e9, 2), "ms": round(best * 1e3, 1), "ratio": round(out / nbytes, 4)}

res = []
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
# C3: zstd SpeedDefault, T, 128 KiB units
usz = 131072; n = int(gib * 2**30) // usz
buf = _lib.corpus_fill("T", 0x5EED0001, 0, n, usz); d = torch.from_numpy(buf).cuda(); off = np.arange(n + 1, dtype=np.uint64) * usz
for lvl, nm in ((1, "C2 zstd fastest T"), (2, "C3 zstd default T")):
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(lvl)); cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    dd = torch.empty(cap, dtype=torch.uint8, device="cuda")
    res.append(run(nm, lambda: int(enc.EncodeUnitsDevice(d.data_ptr(), off, dd.data_ptr(), cap)[n]), n * usz)); res[-1]["kernels"] = enc.ctx().timings(); enc.Close(); del dd
del d
# C4: S2, J, 64 KiB blocks
bsz = 65536; nb = int(gib * 2**30) // bsz
buf = _lib.corpus_fill("J", 0x5EED0003, 0, nb, bsz); d = torch.from_numpy(buf).cuda(); boff = np.arange(nb + 1, dtype=np.uint64) * bsz
e2 = s2.BlockEncoder(); cap = nb * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64; dd = torch.empty(cap, dtype=torch.uint8, device="cuda")
res.append(run("C4 s2 J 64KiB", lambda: int(e2.EncodeBlocksDevice(d.data_ptr(), boff, dd.data_ptr(), cap)[nb]), nb * bsz)); e2.Close(); del d, dd
# C5: zstd better + 64 KiB raw dict, M, 128 KiB units (1 GiB)
n5 = min(n, 8192)
buf = _lib.corpus_fill("M", 0x5EED0004, 0, n5, usz); d = torch.from_numpy(buf).cuda(); off5 = np.arange(n5 + 1, dtype=np.uint64) * usz
dct = _lib.corpus_fill("T", 0x5EED0005, 0, 1, 65536).tobytes()
for nm, opts in (("C5 zstd better+dict M", [zstd.WithEncoderLevel(3), zstd.WithEncoderDictRaw(1, dct)]), ("zstd better (no dict) M", [zstd.WithEncoderLevel(3)])):
    enc = zstd.NewWriter(None, *opts); cap = n5 * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64; dd = torch.empty(cap, dtype=torch.uint8, device="cuda")
    res.append(run(nm, lambda: int(enc.EncodeUnitsDevice(d.data_ptr(), off5, dd.data_ptr(), cap)[n5]), n5 * usz)); res[-1]["kernels"] = enc.ctx().timings(); enc.Close(); del dd
for r in res: print(json.dumps(r))
