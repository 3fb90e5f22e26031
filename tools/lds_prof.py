"""Per-phase shader-clock profile of the LDS-table match finder (build kc_zstd_match_lds.hip with -DKC_LDS_PROF first:
   touch compress_amd/csrc/kc_zstd_match_lds.hip; KC_EXTRA_FLAGS=-DKC_LDS_PROF python -m compress_amd.build)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["KC_K2_PROF"] = "1"
import torch
from compress_amd import zstd
import corpora

kind = sys.argv[1] if len(sys.argv) > 1 else "T"
n, usz = int(sys.argv[2]) if len(sys.argv) > 2 else 256, 131072
buf = corpora.corpus(kind, n, usz)
d_src = torch.from_numpy(buf).cuda()
enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("lds"))
cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
off = np.arange(n + 1, dtype=np.uint64) * usz
for _ in range(2):
    t0 = time.perf_counter()
    enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
    print("%s x %d: %.2f ms, timings %r" % (kind, n, (time.perf_counter() - t0) * 1e3, enc.ctx().timings()), flush=True)
