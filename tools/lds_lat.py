"""Latency-path probe (GPU box): the LDS-table kernels at few units in flight, per kernel mode.

    python tools/lds_lat.py [--out gpurun_out/lds_lat.json] [--zstd]

S2: 64 KiB 'J' / 'T' blocks, device resident, N = 1 / 16 / 256 / 512 / 1024 in flight, KC_OPT_S2_LDS_SPEC_W0 = 0 (fused step) / 1
(first one-step form) / 8 (speculative rounds); every mode's bytes are compared with mode 1's.  --zstd adds SpeedFastest 128 KiB
'T' units on the LDS path.  Wall clock of the whole call (best of 5) and the kernel time the library reports."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
OPT_S2_LDS_SPEC_W0 = 17
OPT_LDS_SPEC_W0 = 6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/lds_lat.json")
    ap.add_argument("--zstd", action="store_true")
    ap.add_argument("--modes", default="0,1,8")
    ap.add_argument("--zmodes", default="16")
    a = ap.parse_args()
    import torch
    from compress_amd import zstd, s2
    import corpora
    res = {}
    ns = [1, 16, 256, 512, 1024]
    for kind in "JT":
        bsz = 65536
        buf = corpora.corpus(kind, ns[-1], bsz)
        d_src = torch.from_numpy(buf).cuda()
        enc = s2.BlockEncoder(level=0, path="lds")
        cap = ns[-1] * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64
        d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
        ref = {}
        for mode in [int(m) for m in a.modes.split(",")]:
            enc.ctx().set_option(OPT_S2_LDS_SPEC_W0, mode)
            for n in ns:
                off = np.arange(n + 1, dtype=np.uint64) * bsz
                best, bk = 1e9, 0.0
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    oo = enc.EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
                    dt = (time.perf_counter() - t0) * 1e3
                    if dt < best:
                        best, bk = dt, enc.ctx().timings()["match_ms"]
                got = d_dst[:int(oo[n])].cpu().numpy().tobytes()
                if n not in ref:
                    ref[n] = got
                res["s2.%s.mode%d.n%d" % (kind, mode, n)] = {"ms": round(best, 3), "kernel_ms": round(bk, 3), "GBps": round(n * bsz / best / 1e6, 3),
                                                              "same_bytes_as_first_mode": got == ref[n], "path": enc.ctx().last_path()}
                print("s2", kind, "mode", mode, "n", n, res["s2.%s.mode%d.n%d" % (kind, mode, n)], flush=True)
        enc.Close()
    if a.zstd:
        usz = 131072
        nz = [1, 16, 256, 512]
        buf = corpora.corpus("T", nz[-1], usz)
        d_src = torch.from_numpy(buf).cuda()
        enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("lds"))
        cap = nz[-1] * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
        d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
        ref = {}
        for mode in [int(m) for m in a.zmodes.split(",")]:
            enc.ctx().set_option(OPT_LDS_SPEC_W0, mode)
            for n in nz:
                off = np.arange(n + 1, dtype=np.uint64) * usz
                best, bk = 1e9, 0.0
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    oo = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
                    dt = (time.perf_counter() - t0) * 1e3
                    if dt < best:
                        best, bk = dt, enc.ctx().timings()["match_ms"]
                got = d_dst[:int(oo[n])].cpu().numpy().tobytes()
                if n not in ref:
                    ref[n] = got
                k = "zstd.T.mode%d.n%d" % (mode, n)
                res[k] = {"ms": round(best, 3), "match_ms": round(bk, 3), "GBps": round(n * usz / best / 1e6, 3), "same_bytes_as_first_mode": got == ref[n],
                          "path": enc.ctx().last_path()}
                print(k, res[k], flush=True)
        enc.Close()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
