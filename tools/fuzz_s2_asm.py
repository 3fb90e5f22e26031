"""Device against the reference itself on many random blocks: KC_S2_VARIANT_AMD64 against the reference's assembly encoders
(oracle/_ref/libs2ref.so) and — round 4 — KC_S2_VARIANT_GO against its portable Go encoders (oracle/_ref/libzstdref.so, translated),
both kernel families (the LDS one runs the fused step on blocks up to 64 KiB).
python tools/fuzz_s2_asm.py [n_blocks] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpora, oracle_ref
from compress_amd import s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
pools = {k: corpora.corpus(k, 64, 131072, first_unit=int(rng.integers(0, 1000))).tobytes() for k in "JTMH"}
blocks = []
for i in range(n):
    r = rng.random()
    if r < 0.15:
        blocks.append(bytes(rng.integers(0, int(rng.integers(2, 9)), int(rng.integers(32, 6000)), dtype=np.uint8)))
        continue
    k = "JTMH"[int(rng.integers(0, 4))]
    ln = int(rng.choice([rng.integers(32, 600), rng.integers(400, 5000), rng.integers(3000, 20000), rng.integers(15000, 70000), rng.integers(60000, 300000)]))
    st = int(rng.integers(0, len(pools[k]) - ln))
    b = bytearray(pools[k][st:st + ln])
    if r > 0.8:  # splice another kind in the middle
        k2 = "JTMH"[int(rng.integers(0, 4))]
        m = int(rng.integers(0, ln))
        b[m:] = pools[k2][st:st + ln - m]
    blocks.append(bytes(b))
b2, off = corpora.pack_units(blocks)
tot_bad = 0
for level in range(4):
    for path in (("hbm", "lds") if level in (0, 2) else ("hbm",)):
        enc = s2.BlockEncoder(level=level, variant="amd64", path=path)
        out, oo = enc.EncodeBlocks(b2, off)
        ref, ro = oracle_ref.encode_blocks(b2, off, level=level, threads=16)
        bad = [i for i in range(len(blocks)) if out[int(oo[i]):int(oo[i + 1])].tobytes() != ref[int(ro[i]):int(ro[i + 1])].tobytes()]
        print("level", level, "path", path, "blocks", len(blocks), "bytes", len(b2), "differing", len(bad), bad[:5], flush=True)
        tot_bad += len(bad)
        enc.Close()
try:
    import oracle_goref
    have_go = oracle_goref.available()
except Exception:
    have_go = False
if have_go:
    sub = list(range(0, len(blocks), 3))  # (one call of the translated reference per block: a third of the set)
    for level in (0, 2):
        want = {i: oracle_goref.s2_encode(blocks[i], level) for i in sub}
        for path in ("hbm", "lds"):
            enc = s2.BlockEncoder(level=level, path=path)
            out, oo = enc.EncodeBlocks(b2, off)
            bad = [i for i in sub if out[int(oo[i]):int(oo[i + 1])].tobytes() != want[i]]
            print("go variant: level", level, "path", path, "blocks", len(sub), "differing", len(bad), bad[:5], flush=True)
            tot_bad += len(bad)
            enc.Close()
sys.exit(1 if tot_bad else 0)
