"""Units in flight vs throughput for the two kernel families (KC_PATH_HBM / KC_PATH_LDS), on the GPU box.

    python tools/crossover.py [--out gpurun_out] [--max-units 32768]

Writes r03_crossover_zfast.csv (zstd SpeedFastest, 128 KiB 'T' units, device resident), r03_crossover_s2.csv (s2.Encode,
64 KiB 'J' blocks, device resident) and r03_latency.json (one 128 KiB EncodeAll from host memory; the WriterCustomEncoder hook
with 1 / 16 / 64 callers).  The crossovers in these files are what KC_OPT_ZFAST_LDS_MAX_UNITS / KC_OPT_S2_LDS_MAX_BLOCKS default to."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out")
    ap.add_argument("--max-units", type=int, default=32768)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from compress_amd import zstd, s2, _lib
    import corpora
    os.makedirs(a.out, exist_ok=True)
    ns = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
    ns = [n for n in ns if n <= a.max_units]

    # ---------------- zstd SpeedFastest ----------------
    usz = 131072
    nmax = ns[-1]
    buf = corpora.corpus("T", nmax, usz)
    d_src = torch.from_numpy(buf).cuda()
    rows = []
    encs = {p: zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath(p)) for p in ("hbm", "lds")}
    cap = nmax * ((encs["hbm"].MaxEncodedSize(usz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for n in ns:
        off = np.arange(n + 1, dtype=np.uint64) * usz
        r = {"units": n}
        outs = {}
        for p, enc in encs.items():
            best, bm = 1e9, 0.0
            for _ in range(a.reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                oo = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
                dt = (time.perf_counter() - t0) * 1e3
                if dt < best:
                    best, bm = dt, enc.ctx().timings()["match_ms"]
            outs[p] = d_dst[:int(oo[n])].cpu().numpy().copy() if n <= 1024 else int(oo[n])
            r[p + "_ms"] = round(best, 3)
            r[p + "_match_ms"] = round(bm, 3)
            r[p + "_gbps"] = round(n * usz / best / 1e6, 3)
        r["same_bytes"] = bool(np.array_equal(outs["hbm"], outs["lds"]))
        rows.append(r)
        print(r, flush=True)
    with open(os.path.join(a.out, "r03_crossover_zfast.csv"), "w") as f:
        keys = list(rows[0].keys())
        f.write(",".join(keys) + "\n")
        for r in rows:
            f.write(",".join(str(r[k]) for k in keys) + "\n")
    for e in encs.values():
        e.Close()

    # ---------------- s2.Encode ----------------
    bsz = 65536
    jbuf = corpora.corpus("J", nmax, bsz)
    d_src = torch.from_numpy(jbuf).cuda()
    cap = nmax * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    rows = []
    bencs = {p: s2.BlockEncoder(path=p) for p in ("hbm", "lds")}
    for n in ns:
        off = np.arange(n + 1, dtype=np.uint64) * bsz
        r = {"blocks": n}
        outs = {}
        for p, enc in bencs.items():
            best = 1e9
            for _ in range(a.reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                oo = enc.EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
                best = min(best, (time.perf_counter() - t0) * 1e3)
            outs[p] = d_dst[:int(oo[n])].cpu().numpy().copy() if n <= 1024 else int(oo[n])
            r[p + "_ms"] = round(best, 3)
            r[p + "_gbps"] = round(n * bsz / best / 1e6, 3)
        r["same_bytes"] = bool(np.array_equal(outs["hbm"], outs["lds"]))
        rows.append(r)
        print(r, flush=True)
    with open(os.path.join(a.out, "r03_crossover_s2.csv"), "w") as f:
        keys = list(rows[0].keys())
        f.write(",".join(keys) + "\n")
        for r in rows:
            f.write(",".join(str(r[k]) for k in keys) + "\n")

    # ---------------- latency: one EncodeAll, the hook ----------------
    lat = {}
    unit = buf[:usz].tobytes()
    for p in ("hbm", "lds", "auto"):
        enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath(p))
        enc.EncodeAll(unit)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            enc.EncodeAll(unit)
            ts.append((time.perf_counter() - t0) * 1e3)
        lat["encode_all_128k_%s_ms" % p] = round(min(ts), 3)
        tm = enc.ctx().timings()
        lat["encode_all_128k_%s_kernels_ms" % p] = {k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")}
        enc.Close()
    blocks = [jbuf[i * bsz:(i + 1) * bsz].tobytes() for i in range(1024)]
    for p in ("hbm", "lds"):
        for nthr in (1, 16, 64):
            enc = s2.BlockEncoder(path=p)
            fn = enc.CustomEncoder()
            fn(bytearray(bsz + 64), blocks[0])
            todo = blocks if (p == "lds" or nthr > 1) else blocks[:64]
            idx = [0]
            lock = threading.Lock()

            def work():
                dst = bytearray(bsz + 64)
                while True:
                    with lock:
                        i = idx[0]
                        idx[0] += 1
                    if i >= len(todo):
                        return
                    fn(dst, todo[i])
            t0 = time.perf_counter()
            th = [threading.Thread(target=work) for _ in range(nthr)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            lat["hook_%s_%dcallers_MBps" % (p, nthr)] = round(len(todo) * bsz / dt / 1e6, 1)
            enc.Close()
    # ---------------- ONE stream as jobs (WithConcurrentBlocks): 1 GiB, SpeedFastest, its own 4 MiB window -> 64 jobs of 16 MiB ----------------
    if a.max_units >= 8192:
        stream = buf[:8192 * usz]
        enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithConcurrentBlocks(True), zstd.WithEncoderConcurrency(4))
        enc.EncodeJobs(stream[:64 << 20])
        t0 = time.perf_counter()
        out = enc.EncodeJobs(stream)
        dt = time.perf_counter() - t0
        lat["jobs_1GiB_speedfastest_MBps"] = round(len(stream) / dt / 1e6, 1)
        lat["jobs_1GiB_ratio"] = round(len(out) / len(stream), 4)
        lat["jobs_1GiB_match_ms"] = round(enc.ctx().timings()["match_ms"], 1)
        enc.Close()
    print(json.dumps(lat), flush=True)
    with open(os.path.join(a.out, "r03_latency.json"), "w") as f:
        json.dump(lat, f, indent=1)


if __name__ == "__main__":
    main()
