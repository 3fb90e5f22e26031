"""Co-resident LDS-table + HBM-table match finders on one batch (VERDICT r2, item 1 "then the hybrid").

    python tools/hybrid_probe.py [C2|C4] [n_lds ...]

Two contexts on two streams share one batch: context A (KC_PATH_HBM) takes units [0, N - n_lds), context B (KC_PATH_LDS) the
last n_lds, both launched before either is waited for, so the kernels co-reside.  Compared with one context taking the whole
batch on the HBM path.  One JSON line per split."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from compress_amd import zstd
import corpora


def main():
    n, usz = 32768, 131072
    splits = [int(x) for x in sys.argv[2:]] or [0, 256, 512, 1024, 2048]
    buf = corpora.corpus("T", n, usz)
    d_src = torch.from_numpy(buf).cuda()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ea = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("hbm"), stream=sa.cuda_stream)
    eb = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("lds"), stream=sb.cuda_stream)
    slot = (ea.MaxEncodedSize(usz) + 15) & ~15
    cap = n * slot + 64
    da = torch.empty(cap, dtype=torch.uint8, device="cuda")
    db = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for nl in splits:
        na = n - nl
        offa = np.arange(na + 1, dtype=np.uint64) * usz
        offb = np.arange(nl + 1, dtype=np.uint64) * usz + np.uint64(na * usz)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ea.EncodeUnitsDeviceBegin(d_src.data_ptr(), offa, da.data_ptr(), cap)
            if nl:
                eb.EncodeUnitsDeviceBegin(d_src.data_ptr(), offb, db.data_ptr(), cap)
            ea.EncodeUnitsDeviceEnd()
            if nl:
                eb.EncodeUnitsDeviceEnd()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(json.dumps({"hbm_units": na, "lds_units": nl, "ms": round(best, 2), "GBps": round(n * usz / best / 1e6, 2),
                          "hbm_match_ms": round(ea.ctx().timings()["match_ms"], 2),
                          "lds_match_ms": round(eb.ctx().timings()["match_ms"], 2) if nl else None}), flush=True)


if __name__ == "__main__":
    main()
