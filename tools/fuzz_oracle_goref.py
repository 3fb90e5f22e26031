"""CPU campaign: the oracle (hand restatement) against the reference's own Go code (oracle/_ref/libzstdref.so, translated) on random
units with random options — EncodeAll at four levels (window, checksum, single segment, zero frames, entropy options, low memory, raw
dictionary), Write / Flush / Close streams in both forms of nextBlock, WithConcurrentBlocks job streams, the six S2 block encoders,
framed S2 streams (block size, index, padding, Flush points; each read back by the reference's s2.Reader), Write + ReadFrom
sequences — and every zstd frame decoded by the reference's decoder in its amd64
build (the package's assembly).  python tools/fuzz_oracle_goref.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpora, oracle_goref, oracle_lib as oracle
import test_ref_s2_stream as ts

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
pools = {k: corpora.corpus(k, 48, 131072, first_unit=int(rng.integers(0, 2000))).tobytes() for k in "JTMH"}
amd = oracle_goref.amd64_available()


def unit():
    r = rng.random()
    if r < 0.1:
        return bytes(rng.integers(0, int(rng.integers(2, 9)), int(rng.integers(0, 9000)), dtype=np.uint8))
    k = "JTMH"[int(rng.integers(0, 4))]
    ln = int(rng.choice([rng.integers(0, 600), rng.integers(400, 9000), rng.integers(3000, 70000), rng.integers(60000, 140000), rng.integers(131072, 600000)]))
    st = int(rng.integers(0, len(pools[k]) - ln))
    b = bytearray(pools[k][st:st + ln])
    if r > 0.8 and ln > 1:
        k2 = "JTMH"[int(rng.integers(0, 4))]
        m = int(rng.integers(0, ln))
        b[m:] = pools[k2][st:st + ln - m]
    return bytes(b)


def pick(*xs):
    return xs[int(rng.integers(0, len(xs)))]


cnt = {"encode_all": 0, "stream": 0, "jobs": 0, "s2": 0, "s2stream": 0, "decoded": 0}
bad = []
t0 = time.time()
while time.time() - t0 < budget and len(bad) < 20:
    u = unit()
    level = int(rng.integers(1, 5))
    if level == 4 and len(u) > 200000:
        u = u[:200000]
    mode = rng.random()
    if mode < 0.45:
        kw = {}
        if rng.random() < 0.5:
            kw["window_size"] = 1 << int(rng.integers(10, 24))
        for name in ("crc", "single", "full_zero", "no_entropy", "all_lit_entropy"):
            if rng.random() < 0.25:
                kw[name] = bool(rng.integers(0, 2))
        if rng.random() < 0.15:
            kw["low_mem"] = True
        dkw = {}
        if rng.random() < 0.3:
            d = pools["T"][:int(rng.integers(8, 70000))]
            kw.update(dict_id=9, dict_content=d)
            dkw = dict(dict_id=9, dict_content=d)
        f = oracle_goref.zstd_encode_all(u, level=level, **kw)
        if oracle.ZstdOracle(level=level, **kw).encode_all(u) != f:
            bad.append(("encode_all", level, len(u), sorted(kw)))
        cnt["encode_all"] += 1
    elif mode < 0.65:
        cuts = tuple(sorted(int(x) for x in rng.integers(0, len(u) + 1, int(rng.integers(0, 4)))))
        conc = int(rng.integers(0, 2))
        kw = {}
        if rng.random() < 0.3:
            kw["window_size"] = 1 << int(rng.integers(14, 22))
        if rng.random() < 0.3:
            kw["crc"] = False
        a = int(rng.integers(0, len(u) + 1)) if rng.random() < 0.3 else None  # from here on through ReadFrom: a Flush point for the bytes
        if a is not None and cuts:
            a = max(a, max(cuts))  # (the driver flushes at every cut first: ReadFrom takes over behind the last one)
        eff = cuts if a is None or a == 0 else tuple(sorted(set(cuts) | {a}))
        f = oracle_goref.zstd_encode_stream(u, cuts, level=level, concurrent=conc, readfrom_at=a, **kw)
        if oracle.ZstdOracle(level=level, concurrent=conc, **kw).encode_stream(u, eff) != f:
            bad.append(("stream", level, len(u), cuts, a, conc, sorted(kw)))
        dkw = {}
        cnt["stream"] += 1
    elif mode < 0.75:
        win = 1 << int(rng.integers(17, 19))
        big = (u * (1 + (1 << 20) // max(1, len(u))))[:int(rng.integers(1, 1400000))] if level < 4 else u
        cuts = tuple(sorted(int(x) for x in rng.integers(0, len(big) + 1, int(rng.integers(0, 3)))))
        f = oracle_goref.zstd_encode_stream(big, cuts, level=level, window_size=win, concurrent=4, jobs=True)
        if oracle.ZstdOracle(level=level, window_size=win).encode_jobs(big, cuts) != f:
            bad.append(("jobs", level, len(big), cuts, win))
        u, dkw = big, {}
        cnt["jobs"] += 1
    elif mode < 0.9:
        lv = int(rng.integers(0, 6))
        b = u[:int(rng.integers(0, min(len(u), 262144) + 1))]
        fn = [oracle.s2_encode, oracle.s2_encode_better, oracle.s2_encode_snappy, oracle.s2_encode_snappy_better, oracle.s2_encode_best, oracle.s2_encode_snappy_best][lv]
        if (lv in (2, 3, 5) and len(b) > 65536) or len(b) == 0:
            continue
        if fn(b) != oracle_goref.s2_encode(b, lv):
            bad.append(("s2", lv, len(b)))
        cnt["s2"] += 1
        continue
    else:
        lv, snappy = int(rng.integers(0, 3)), bool(rng.integers(0, 2))
        bs = pick(4096, 16384, 65536) if snappy else pick(4096, 65536, 1 << 18, 0)
        cuts = tuple(sorted(int(x) for x in rng.integers(0, len(u) + 1, int(rng.integers(0, 4)))))
        kw = dict(add_index=bool(rng.integers(0, 2)), padding=pick(0, 0, 512, 4096, 70000))
        if lv == 2 and len(u) > 150000:
            u = u[:150000]
            cuts = tuple(c for c in cuts if c <= len(u))
        got = oracle_goref.s2_stream(u, cuts, level=lv, snappy=snappy, block_size=bs, **kw)
        if got != ts.expected_stream(oracle, u, cuts, bs=bs or (1 << 20), level=lv, snappy=snappy, **kw):
            bad.append(("s2stream", lv, snappy, bs, len(u), cuts, sorted(kw.items())))
        try:  # ... and the reference's own s2.Reader returns the input
            if oracle_goref.s2_read_stream(got, len(u)) != u:
                bad.append(("s2.Reader: wrong bytes", lv, snappy, bs, len(u)))
        except ValueError as e:
            # (the reference's quirk: an EMPTY stream closed with WriterAddIndex is an index chunk — and its padding — without a stream
            # identifier in front, which its own Reader refuses: "s2: corrupt input")
            if not (len(u) == 0 and kw["add_index"] and "corrupt input" in str(e)):
                bad.append(("s2.Reader", str(e), lv, snappy, bs, len(u)))
        cnt["s2stream"] += 1
        continue
    if amd:  # the frame just written, read by the reference's amd64 decoders (assembly), BMI2 on and off in turn
        with oracle_goref.flavour(pick("amd64", "amd64-nobmi")):
            try:
                if oracle_goref.zstd_decode_all(f, len(u), **dkw) != u:
                    bad.append(("decode: wrong bytes", level, len(u)))
            except ValueError as e:
                bad.append(("decode", str(e), level, len(u)))
        cnt["decoded"] += 1
print("seed %d, %.0f s: %r; differences: %d" % (seed, time.time() - t0, cnt, len(bad)))
for b in bad:
    print("  ", b)
sys.exit(1 if bad else 0)
