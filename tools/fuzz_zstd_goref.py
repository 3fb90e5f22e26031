"""Device against the reference's own Go encoder (oracle/_ref/libzstdref.so, translated) on random units: EncodeAll at the four levels
(SpeedFastest on both kernel families), streams with Flush points, raw dictionaries.
python tools/fuzz_zstd_goref.py [n_units] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpora, oracle_goref
from compress_amd import zstd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
pools = {k: corpora.corpus(k, 32, 131072, first_unit=int(rng.integers(0, 1000))).tobytes() for k in "JTMH"}
units = []
for i in range(n):
    r = rng.random()
    if r < 0.12:
        units.append(bytes(rng.integers(0, int(rng.integers(2, 9)), int(rng.integers(1, 9000)), dtype=np.uint8)))
        continue
    k = "JTMH"[int(rng.integers(0, 4))]
    ln = int(rng.choice([rng.integers(1, 600), rng.integers(400, 9000), rng.integers(3000, 70000), rng.integers(60000, 140000), rng.integers(131072, 400000)]))
    st = int(rng.integers(0, len(pools[k]) - ln))
    b = bytearray(pools[k][st:st + ln])
    if r > 0.8:
        k2 = "JTMH"[int(rng.integers(0, 4))]
        m = int(rng.integers(0, ln))
        b[m:] = pools[k2][st:st + ln - m]
    units.append(bytes(b))
buf, off = corpora.pack_units(units)
dct = pools["T"][:65536]
tot_bad = 0
t0 = time.time()
for level in (1, "1L", 2, 3, 4):
    lv = 1 if level == "1L" else level
    sub = list(range(len(units))) if lv < 4 else list(range(0, len(units), 4))  # (SpeedBestCompression: a quarter of the set)
    for with_dict in (False, True):
        opts = [zstd.WithEncoderLevel(lv)] + ([zstd.WithMatchPath("lds")] if level == "1L" else ([zstd.WithMatchPath("hbm")] if level == 1 else []))
        kw = {}
        if with_dict:
            opts.append(zstd.WithEncoderDictRaw(5, dct))
            kw = dict(dict_id=5, dict_content=dct)
        enc = zstd.NewWriter(None, *opts)
        out, oo = enc.EncodeUnits(buf, off)
        bad = [i for i in sub if out[int(oo[i]):int(oo[i + 1])].tobytes() != oracle_goref.zstd_encode_all(units[i], level=lv, **kw)]
        cuts = [sorted(int(x) for x in rng.integers(0, len(u) + 1, int(rng.integers(0, 3)))) for u in units]
        sout, so = enc.EncodeStreams(buf, off, flush_at=cuts)
        sbad = [i for i in sub[::2] if sout[int(so[i]):int(so[i + 1])].tobytes() != oracle_goref.zstd_encode_stream(units[i], cuts[i], level=lv, **kw)]
        print("level", level, "dict" if with_dict else "no dict", ": EncodeAll", len(sub), "units differing", len(bad), bad[:5], "; streams", len(sub[::2]), "differing", len(sbad), sbad[:5],
              "(%.0f s)" % (time.time() - t0), flush=True)
        tot_bad += len(bad) + len(sbad)
        enc.Close()
sys.exit(1 if tot_bad else 0)
