"""kc_zstd_prime_kernel on the wave emulator against the position-by-position restatement of the reference's ResetPrefix loops
(tests/test_emu_prime.py) on random prefixes: corpus text, low-entropy bytes (whole rounds in one bucket), every length class, both
lane orders of the emulator, random position-field widths.  python tools/fuzz_emu_prime.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpora, emu_lib
from test_emu_prime import reset_prefix_tables

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
pools = {k: corpora.corpus(k, 8, 131072, first_unit=int(rng.integers(0, 500))).tobytes() for k in "JTMH"}
t0, n, nbytes, bad = time.time(), 0, 0, []
while time.time() - t0 < budget and not bad:
    pre = []
    for _ in range(6):
        r = rng.random()
        ln = int(rng.choice([rng.integers(0, 20), rng.integers(8, 300), rng.integers(200, 3000), rng.integers(2000, 12000)]))
        if r < 0.3:
            p = bytes(rng.integers(0, int(rng.integers(1, 5)), ln, dtype=np.uint8))
        elif r < 0.45:
            w = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
            p = (w * (ln // len(w) + 1))[:ln]
        else:
            k = "JTMH"[int(rng.integers(0, 4))]
            st = int(rng.integers(0, len(pools[k]) - ln))
            p = pools[k][st:st + ln]
        pre.append(p)
    level = int(rng.integers(1, 4))
    pb = int(rng.integers(15, 31))
    rev = bool(rng.integers(0, 2))
    got = emu_lib.zstd_prime(level, pre, pb, reverse=rev)
    for i, p in enumerate(pre):
        if not np.array_equal(got[i], reset_prefix_tables(level, p, pb)):
            bad.append((level, pb, rev, len(p), p[:32].hex()))
        n += 1
        nbytes += len(p)
print("seed %d, %.0f s: %d prefixes (%d bytes) at random levels / position widths / lane orders; differences: %d" % (seed, time.time() - t0, n, nbytes, len(bad)))
for b in bad:
    print("  ", b)
sys.exit(1 if bad else 0)
