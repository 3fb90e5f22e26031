#!/usr/bin/env python3
"""The reference's goroutine-parallel CPU form of the hot path, timed: T threads, each with ONE encoder of the reference re-used
from unit to unit, calling EncodeAll on its contiguous share of the units — what N goroutines on one zstd.Encoder get from its
encoder pool (zstd/encoder.go:90-99, 722-729).  The encoder is the reference's own Go source translated to C++ at build time
(oracle/_ref/libzstdref*.so, oracle/ref_go): TEST / MEASUREMENT INFRASTRUCTURE, never on the product path.

bench.py runs this file as a CHILD process (cpu_baseline.reference_translated_parallel): a fault inside the translated runtime
under threads must not take the bench line along.  It regenerates the same synthetic units (kc_corpus_fill is host code),
prints one JSON line with the rate and the SHA-256 of the concatenated frames; the parent compares that digest with the
device's frames for the same units.

  python tools/ref_parallel.py --kind T --seed 0x5EED0001 --units 2048 --unit 131072 --level 1 --threads 16 [--dict-kib 64 --dict-seed N]
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="T")
    ap.add_argument("--seed", type=lambda s: int(s, 0), required=True)
    ap.add_argument("--first-unit", type=int, default=0)
    ap.add_argument("--units", type=int, required=True)
    ap.add_argument("--unit", type=int, default=128 * 1024)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--dict-kib", type=int, default=0)
    ap.add_argument("--dict-seed", type=lambda s: int(s, 0), default=0)
    ap.add_argument("--passes", type=int, default=1, help="timed passes after the untimed one (the best is reported with all of them)")
    a = ap.parse_args()

    from compress_amd import _lib
    import oracle_goref
    if not oracle_goref.available():
        print(json.dumps({"error": "oracle/_ref/libzstdref.so is not present"}))
        return 0
    n, usz, T = a.units, a.unit, max(1, min(a.threads, a.units))
    host = _lib.corpus_fill(a.kind, a.seed, a.first_unit, n, usz).tobytes()
    units = [host[i * usz:(i + 1) * usz] for i in range(n)]
    kw = dict(level=a.level)
    if a.dict_kib:
        kw.update(dict_id=1, dict_content=_lib.corpus_fill("T", a.dict_seed, 0, 1, a.dict_kib << 10).tobytes())
    fl = "amd64" if oracle_goref.amd64_available() else "generic"
    bounds = [k * n // T for k in range(T + 1)]
    with oracle_goref.flavour(fl):
        # package initialisation (the reference's sync.Once tables) on one thread, before any concurrency
        oracle_goref.zstd_encode_all_reuse(units[:1], **kw)
        outs = [None] * T
        errs = []

        def work(k):
            try:
                outs[k] = oracle_goref.zstd_encode_all_reuse(units[bounds[k]:bounds[k + 1]], **kw)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        def one_pass():
            th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            return time.perf_counter() - t0

        one_pass()  # untimed: thread stacks, the encoders' tables, the output buffers
        times = [one_pass() for _ in range(max(1, a.passes))]
    if errs:
        print(json.dumps({"error": errs[0][:300]}))
        return 0
    h = hashlib.sha256()
    total = 0
    for o in outs:
        for f in o:
            h.update(f)
            total += len(f)
    best = min(times)
    print(json.dumps({"value": round(n * usz / best / 1e6, 1), "unit": "MB/s", "cores": T, "units": n, "flavour": fl,
                      "passes_s": [round(t, 3) for t in times], "frames_bytes": total, "frames_sha256": h.hexdigest()}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
