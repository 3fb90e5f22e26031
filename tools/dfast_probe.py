import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from compress_amd import _lib, zstd
usz = 131072; n = 16384
buf = _lib.corpus_fill("T", 0x5EED0001, 0, n, usz); d = torch.from_numpy(buf).cuda(); off = np.arange(n + 1, dtype=np.uint64) * usz
enc = zstd.NewWriter(None, zstd.WithEncoderLevel(2)); cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
dd = torch.empty(cap, dtype=torch.uint8, device="cuda")
for it in range(2):
    oo = enc.EncodeUnitsDevice(d.data_ptr(), off, dd.data_ptr(), cap)
print(os.environ.get("KC_SPEC_W0"), os.environ.get("KC_SPEC_GROW"), enc.ctx().timings())
