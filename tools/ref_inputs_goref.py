"""Every input of the reference's own test archives kept under tests/golden/ref_inputs (fuzz corpora, crashers, regression inputs)
through the oracle and through the translated reference (oracle/_ref/libzstdref.so) at the four zstd levels and the six S2 levels,
each zstd frame read back by the reference's amd64 decoders.  CPU only.  python tools/ref_inputs_goref.py [max bytes per input]"""
import os, sys, time, zipfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_goref, oracle_lib as oracle

REFIN = os.path.join(ROOT, "tests", "golden", "ref_inputs")
cap = int(sys.argv[1]) if len(sys.argv) > 1 else (4 << 20)
amd = oracle_goref.amd64_available()
s2fn = [oracle.s2_encode, oracle.s2_encode_better, oracle.s2_encode_snappy, oracle.s2_encode_snappy_better, oracle.s2_encode_best, oracle.s2_encode_snappy_best]
t0 = time.time()
tot = {"inputs": 0, "bytes": 0, "zstd": 0, "s2": 0, "decoded": 0}
bad = []
for name in sorted(os.listdir(REFIN)):
    if not name.endswith(".zip"):
        continue
    with zipfile.ZipFile(os.path.join(REFIN, name)) as z:
        for n in sorted(z.namelist()):
            data = z.read(n)
            if len(data) > cap:
                continue
            tot["inputs"] += 1
            tot["bytes"] += len(data)
            for level in (1, 2, 3, 4):
                f = oracle_goref.zstd_encode_all(data, level=level)
                if oracle.ZstdOracle(level=level).encode_all(data) != f:
                    bad.append((name, n, "zstd", level, len(data)))
                tot["zstd"] += 1
                if amd and level in (1, 3):
                    with oracle_goref.flavour("amd64" if level == 1 else "amd64-nobmi"):
                        if oracle_goref.zstd_decode_all(f, len(data)) != data:
                            bad.append((name, n, "decode", level, len(data)))
                    tot["decoded"] += 1
            for lv in range(6):
                if len(data) == 0 or (lv in (2, 3, 5) and len(data) > 65536):
                    continue
                if s2fn[lv](data) != oracle_goref.s2_encode(data, lv):
                    bad.append((name, n, "s2", lv, len(data)))
                tot["s2"] += 1
    print("%s done: %r, %.0f s, differences so far %d" % (name, tot, time.time() - t0, len(bad)), flush=True)
for b in bad[:40]:
    print("  ", b)
sys.exit(1 if bad else 0)
