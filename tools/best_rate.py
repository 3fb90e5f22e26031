"""SpeedBestCompression device rate: N units of 128 KiB of a corpus, resident in HBM.  python tools/best_rate.py [kind] [units] [slots]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from compress_amd import zstd
import corpora
kind = sys.argv[1] if len(sys.argv) > 1 else "T"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
usz = 131072
buf = corpora.corpus(kind, n, usz)
d = torch.from_numpy(buf).cuda()
off = np.arange(n + 1, dtype=np.uint64) * usz
enc = zstd.NewWriter(None, zstd.WithEncoderLevel(4))
enc.ctx().set_option(19, slots)
cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    oo = enc.EncodeUnitsDevice(d.data_ptr(), off, dst.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tm = enc.ctx().timings()
    print("kind %s units %d slots %d: %.1f ms, %.1f MB/s, ratio %.4f, match %.1f ms entropy %.1f ms" % (kind, n, slots, dt * 1e3, n * usz / dt / 1e6, int(oo[n]) / (n * usz), tm["match_ms"], tm["entropy_ms"]), flush=True)
