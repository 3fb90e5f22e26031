"""SpeedBestCompression in the oracle (oracle/kco_zstd_best.h; zstd/enc_best.go): the entropy estimate it decides with, and frames
that decode.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

import corpora


def _go_log(x):
    """math.Log of the Go runtime (math/log.go, FDLIBM e_log), in Python floats: IEEE doubles, one rounding per operation."""
    Ln2Hi, Ln2Lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    L1, L2, L3, L4 = 6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01
    L5, L6, L7 = 1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01
    f1, ki = math.frexp(x)
    if f1 < math.sqrt(2) / 2:
        f1 *= 2
        ki -= 1
    f = f1 - 1
    k = float(ki)
    s = f / (2 + f)
    s2 = s * s
    s4 = s2 * s2
    t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)))
    t2 = s4 * (L2 + s4 * (L4 + s4 * L6))
    R = t1 + t2
    hfsq = 0.5 * f * f
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f)


def _go_log2(x):
    frac, e = math.frexp(x)
    if frac == 0.5:
        return float(e - 1)
    return _go_log(frac) * (1 / math.log(2)) + float(e)


def _shannon(b):
    if not b:
        return 0
    hist = np.bincount(np.frombuffer(b, dtype=np.uint8), minlength=256)
    inv = 1.0 / float(len(b))
    sh = 0.0
    for v in hist:
        if v > 0:
            n = float(v)
            sh += math.ceil(-_go_log2(n * inv) * n)
    return int(math.ceil(sh))


def test_go_log2_and_shannon_entropy_bits(oracle):
    """The oracle's math.Log2 restatement (C++, contraction off) against an independent one in Python floats, bit for bit, on the
    arguments the estimate feeds it — count / total for every total a block can have is too many, so: every count of a few totals,
    random pairs, and the exact powers of two; then ShannonEntropyBits on blocks of every corpus kind."""
    L = oracle.lib()
    L.kco_go_log2.restype = C.c_double
    L.kco_go_log2.argtypes = [C.c_double]
    L.kco_shannon_entropy_bits.restype = C.c_int64
    L.kco_shannon_entropy_bits.argtypes = [C.c_char_p, C.c_uint64]
    assert (1 / math.log(2)).hex() == (1.4426950408889634).hex() == "0x1.71547652b82fep+0"
    rng = np.random.default_rng(1)
    for tot in (16, 1000, 65536, 131072, 100003):
        inv = 1.0 / tot
        for v in list(range(1, min(tot, 3000))) + [int(x) for x in rng.integers(1, tot, 2000)]:
            x = v * inv
            assert L.kco_go_log2(x).hex() == _go_log2(x).hex(), (tot, v)
    for e in range(-20, 1):
        assert L.kco_go_log2(2.0 ** e) == float(e)
    # close to the libm value (a sanity check of the restated constants, not a bit-exact one)
    for x in (0.3, 0.0001234, 0.999, 0.5000001):
        assert abs(_go_log2(x) - math.log2(x)) < 1e-14
    for kind in "THJM":
        buf = corpora.corpus(kind, 3, 131072, first_unit=5).tobytes()
        for blk in (buf[:131072], buf[131072:131072 + 65536], buf[200000:200017], buf[:1000], buf[5:100003]):
            assert L.kco_shannon_entropy_bits(blk, len(blk)) == _shannon(blk), (kind, len(blk))
    assert L.kco_shannon_entropy_bits(b"a" * 64, 64) == 0
    assert L.kco_shannon_entropy_bits(b"ab" * 64, 128) == 128


@pytest.mark.parametrize("kind", ["T", "J", "M", "H"])
def test_best_level_frames_decode_and_beat_better(oracle, kind):
    """EncodeAll, a multi-block unit, a stream with Flush points and a job stream at SpeedBestCompression decode back through the
    independent decoder, and are not larger than SpeedBetterCompression's on compressible input."""
    d = corpora.corpus(kind, 6, 131072, first_unit=3).tobytes()
    best, better = oracle.ZstdOracle(level=4), oracle.ZstdOracle(level=3)
    tot = {3: 0, 4: 0}
    for lvl, e in ((3, better), (4, best)):
        for i in range(6):
            u = d[i * 131072:(i + 1) * 131072]
            fr = e.encode_all(u)
            assert oracle.zstd_decompress(fr, len(u) + 16) == u
            tot[lvl] += len(fr)
        for fr in (e.encode_all(d), e.encode_stream(d, (1000, 300000, len(d)))):
            assert oracle.zstd_decompress(fr, len(d) + 16) == d
            tot[lvl] += len(fr)
    if kind != "H":
        assert tot[4] < tot[3], tot
    ej = oracle.ZstdOracle(level=4, window_size=1 << 17)
    assert oracle.zstd_decompress(ej.encode_jobs(d, (70000,)), len(d) + 16) == d


def test_best_level_edges_dictionaries(oracle):
    e = oracle.ZstdOracle(level=4)
    for u in corpora.edge_units() + [x[:120000] for x in corpora.stress_units(seed=3, n=6)]:
        fr = e.encode_all(u)
        if u:
            assert oracle.zstd_decompress(fr, len(u) + 16) == u, len(u)
    import test_oracle_kats as tk
    blob, ins = tk._dict_fixture(oracle)
    raw = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    ed, er = oracle.ZstdOracle(level=4, dict_blob=blob), oracle.ZstdOracle(level=4, dict_id=5, dict_content=raw)
    t = corpora.corpus("T", 2, 131072).tobytes()
    for u in list(ins[:4]) + [t[:1000], t[:200000], raw[100:30000]]:
        assert oracle.zstd_decompress(ed.encode_all(u), len(u) + 16, blob) == u
        assert oracle.zstd_decode(er.encode_all(u), len(u) + 16, dict_content=raw) == u
        assert oracle.zstd_decode(er.encode_stream(u, (500,)), len(u) + 16, dict_content=raw) == u
    # the dictionary helps: frames of dictionary-like input are smaller than without it
    assert len(er.encode_all(raw[100:30000])) < len(e.encode_all(raw[100:30000])) // 4
