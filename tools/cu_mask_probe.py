"""Does the SpeedFastest match finder need all 256 CUs, and does the entropy stage hide under it on the CUs it leaves?

C2 (4 GiB text, 32 768 units of 128 KiB) on streams restricted to a share of the CUs (hipExtStreamCreateWithCUMask).
Part 1: one context, both stages on the masked stream: the match finder's and the entropy stage's kernel time per mask.
Part 2: two contexts pipelined as bench.py does (match finder of step i+1 beside the entropy stage of step i), the match finders on
mask M and the entropy stages (KC_OPT_STAGE2_STREAM) on the complement: wall ms per step.

python tools/cu_mask_probe.py [out.json]"""
import ctypes as C
import json
import os
import sys
import time

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
NW = 8  # 256 CUs = 8 mask words


def words_from_bits(keep):
    w = [0] * NW
    for i in range(NW * 32):
        if keep(i):
            w[i // 32] |= 1 << (i % 32)
    return w


def popc(w):
    return sum(bin(x).count("1") for x in w)


def mk_stream(words):
    if words is None:
        s = torch.cuda.Stream()
        return s, s.cuda_stream
    st = C.c_void_p(0)
    mask = (C.c_uint32 * NW)(*words)
    r = hip.hipExtStreamCreateWithCUMask(C.byref(st), NW, mask)
    assert r == 0, "hipExtStreamCreateWithCUMask -> %d" % r
    return None, st.value


# mask families: "stripe k/8" drops mask bits i with i % 8 < k' (if the driver deals mask bits round-robin over the 8 XCDs this removes whole
# XCDs); "block" drops the lowest bits of every group of 32 (if the driver deals round-robin, this takes the same CUs out of every XCD)
MASKS = [("all 256", None)]
for drop in (1, 2):  # 224, 192 CUs
    MASKS.append(("stripe: bits i%%8 >= %d (%d)" % (drop, 256 - 32 * drop), words_from_bits(lambda i, d=drop: i % 8 >= d)))
for drop in (4, 6, 8, 16):  # of every 32 bits: 224, 208, 192, 128 CUs
    MASKS.append(("block: bits i%%32 >= %d (%d)" % (drop, 256 - 8 * drop), words_from_bits(lambda i, d=drop: i % 32 >= d)))
for drop in (32, 64):  # the lowest bits altogether: 224, 192
    MASKS.append(("low: bits i >= %d (%d)" % (drop, 256 - drop), words_from_bits(lambda i, d=drop: i >= d)))

out = {"what": __doc__.split("\n\n")[0], "library": os.environ.get("KC_LIB_TAG", "product"), "part1_one_context": [], "part2_two_contexts": []}
cap = None
if os.environ.get("PROBE_PARTS", "12") == "2":  # part 2 only (measurement builds)
    MASKS = MASKS[:1]
for name, words in MASKS:
    keep, handle = mk_stream(words)
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("hbm"), stream=handle)
    if cap is None:
        cap = n * ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        oo = enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
        dt = (time.perf_counter() - t0) * 1e3
        t = enc.ctx().timings()
        if best is None or dt < best[0]:
            best = (dt, t["match_ms"] - t.get("prep_ms", 0.0), t["entropy_ms"])
    rec = {"mask": name, "cus": 256 if words is None else popc(words), "step_ms": round(best[0], 2), "match_ms": round(best[1], 2),
           "entropy_ms": round(best[2], 2), "out_bytes": int(oo[n])}
    out["part1_one_context"].append(rec)
    print(json.dumps(rec), flush=True)
    enc.Close()
    del d_dst

# ---- part 2 ----
PAIRS = [("same stream, unmasked (bench.py --pipeline)", None, "same"),
         ("own stage-2 streams, unmasked", None, None)]
for drop in (4, 6, 8):
    m = words_from_bits(lambda i, d=drop: i % 32 >= d)
    cm = words_from_bits(lambda i, d=drop: i % 32 < d)
    PAIRS.append(("block %d | %d" % (popc(m), popc(cm)), m, cm))
for drop in (1, 2):
    m = words_from_bits(lambda i, d=drop: i % 8 >= d)
    cm = words_from_bits(lambda i, d=drop: i % 8 < d)
    PAIRS.append(("stripe %d | %d" % (popc(m), popc(cm)), m, cm))
PAIRS.append(("match finder unmasked | entropy on block 32", None, words_from_bits(lambda i: i % 32 < 4)))
PAIRS.append(("match finder unmasked | entropy on block 64", None, words_from_bits(lambda i: i % 32 < 8)))
d_dsts = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
for name, m, cm in PAIRS:
    keeps, encs = [], []
    for k in range(2):
        k1, h1 = mk_stream(m)
        keeps.append(k1)
        e = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("hbm"), stream=h1)
        if cm != "same":
            k2, h2 = mk_stream(cm)
            keeps.append(k2)
            e.ctx().set_option(_lib.OPT_STAGE2_STREAM, h2)
        encs.append(e)
    encs[0].ChainAfter(encs[1])
    encs[1].ChainAfter(encs[0])

    def run(k):
        tms = []
        encs[0].EncodeUnitsDeviceBegin(d_src.data_ptr(), off, d_dsts[0].data_ptr(), cap)
        for i in range(k):
            cur, nxt = i % 2, (i + 1) % 2
            if i + 1 < k:
                encs[nxt].EncodeUnitsDeviceBegin(d_src.data_ptr(), off, d_dsts[nxt].data_ptr(), cap)
            oo = encs[cur].EncodeUnitsDeviceEnd()
            tms.append(encs[cur].ctx().timings())
        return oo, tms
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 6
    oo, tms = run(K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3 / K
    rec = {"arrangement": name, "ms_per_step": round(dt, 2), "match_ms_median": round(float(np.median([t["match_ms"] - t.get("prep_ms", 0.0) for t in tms])), 2),
           "entropy_ms_median": round(float(np.median([t["entropy_ms"] for t in tms])), 2), "out_bytes": int(oo[n])}
    out["part2_two_contexts"].append(rec)
    print(json.dumps(rec), flush=True)
    for e in encs:
        e.Close()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
