"""Does the SpeedFastest match finder need all 256 CUs?  C2 (4 GiB text, one context) on streams restricted to a share of the CUs
(hipExtStreamCreateWithCUMask); prints the match finder's and the entropy stage's kernel time per mask.
python tools/cu_mask_probe.py"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from compress_amd import zstd, _lib
hip = C.CDLL("libamdhip64.so")
n, usz = 32768, 131072
host = _lib.corpus_fill("T", 0x5EED0001, 0, n, usz)
d_src = torch.from_numpy(host).cuda()
off = np.arange(n + 1, dtype=np.uint64) * usz
for name, word in (("all 256", None), ("7/8", 0xFEFEFEFE), ("3/4", 0xEEEEEEEE), ("1/2", 0xAAAAAAAA), ("1/4", 0x88888888)):
    st = C.c_void_p(0)
    if word is None:
        s_t = torch.cuda.Stream(); handle = s_t.cuda_stream
    else:
        mask = (C.c_uint32 * 8)(*([word] * 8))
        r = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
        assert r == 0, r
        handle = st.value
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("hbm"), stream=handle)
    cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        oo = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
        dt = (time.perf_counter() - t0) * 1e3
        t = enc.ctx().timings()
        if best is None or dt < best[0]:
            best = (dt, t["match_ms"], t["entropy_ms"])
    print("%-8s step %.1f ms, match finder %.1f ms, entropy stage %.1f ms, out %d" % (name, best[0], best[1], best[2], int(oo[n])), flush=True)
    enc.Close(); del d_dst
