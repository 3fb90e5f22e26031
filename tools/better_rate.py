"""SpeedBetterCompression without a dictionary: N units of 128 KiB of corpus M, device resident (table clearing vs epoch stamps)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from compress_amd import zstd
import corpora
n, usz = 8192, 131072
buf = corpora.corpus(sys.argv[1] if len(sys.argv) > 1 else "M", n, usz)
d = torch.from_numpy(buf).cuda()
off = np.arange(n + 1, dtype=np.uint64) * usz
enc = zstd.NewWriter(None, zstd.WithEncoderLevel(3))
cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    oo = enc.EncodeUnitsDevice(d.data_ptr(), off, dst.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tm = enc.ctx().timings()
    print("step %d: %.1f ms, %.1f MB/s, ratio %.4f, match %.1f prep %.2f entropy %.1f" % (it, dt * 1e3, n * usz / dt / 1e6, int(oo[n]) / (n * usz), tm["match_ms"], tm["prep_ms"], tm["entropy_ms"]), flush=True)
