"""One encoder object of the reference (translated) fed a long run of EncodeAll units: every Reset advances its position counter
`cur` by the window plus the history length (enc_base.go:150-154), so `cur` passes bufferReset = MaxInt32 - 2 x window every few
hundred units and the encoder clears its tables and rebases ("Protect against e.cur wraparound", enc_fast.go / enc_dfast.go /
enc_better.go / enc_best.go) — EVERY frame must still be what a fresh encoder writes (the oracle's, itself equal to a fresh translated
encoder: tests/test_ref_go.py).  python tools/reuse_wrap_goref.py [level] [units]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpora, oracle_goref, oracle_lib as oracle

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_units = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
pool = corpora.corpus("T", 64, 131072, first_unit=5).tobytes() + corpora.corpus("M", 32, 131072, first_unit=9).tobytes()
sizes = [131072, 131072, 70001, 131072, 300, 131072, 262144, 131072]
units = []
for i in range(n_units):
    k = (i * 131072 + 7 * i) % (len(pool) - 262144)
    units.append(pool[k:k + sizes[i % len(sizes)]])
t0 = time.time()
frames = oracle_goref.zstd_encode_all_reuse(units, level=level)
t1 = time.time()
ref = oracle.ZstdOracle(level=level)
bad = [i for i in range(n_units) if frames[i] != ref.encode_all(units[i])]
window = {1: 4 << 20, 2: 8 << 20, 3: 8 << 20, 4: 8 << 20}[level]  # WithEncoderLevel's windows (encoder_options.go:248-258)
per = ((1 << 31) - 1 - 2 * window) // (window + 131072)  # bufferReset = MaxInt32 - 2 x window; cur += window + len(hist) per Reset
print("level %d: %d units (%.2f GiB) through ONE reference encoder in %.0f s (cur passes bufferReset about every %d units: ~%d times); "
      "all %d frames compared with a fresh encoder's: %d differ" % (level, n_units, sum(map(len, units)) / 2**30, t1 - t0, per, n_units // per, n_units, len(bad)))
if bad:
    print("  first differing units:", bad[:10])
sys.exit(1 if bad else 0)
