"""The oracle — and the device — against the reference's OWN Go encoders, translated and compiled (oracle/_ref/libzstdref.so).

Whole-encoder bytes of every zstd level used to be "parity unpinned": the reference is pure Go there and the image has no Go
toolchain.  oracle/ref_go translates the reference's Go source statement by statement into C++ at build time (like
oracle/ref_s2asm re-spells its assembly); these tests hold the hand-written oracle (CPU) and the HIP path (GPU) to that library's
bytes: EncodeAll at SpeedFastest / Default / BetterCompression / BestCompression over the synthetic corpora, the edge and stress
sets, the frame options, raw dictionaries and the reference's own test inputs, and the six s2.Encode* levels in their portable
Go form."""
import io
import os
import zipfile

import numpy as np
import pytest

import corpora
import oracle_goref

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = os.environ.get("KC_TEST_FULL", "1") == "1"  # the long forms (SpeedBestCompression on every input of the other levels' sets): the default
# since the translation releases a call's memory (gort.h rt::Scope) — the whole file is ~40 s; KC_TEST_FULL=0 trims it
REFIN = os.path.join(HERE, "golden", "ref_inputs")

pytestmark = pytest.mark.skipif(not oracle_goref.available(), reason="oracle/_ref/libzstdref.so neither present nor buildable (no /root/reference)")


def _flavours():
    """The builds of the reference the decoder / encoder comparisons run against: the portable one, and — on x86-64 — its amd64 build
    (the package's assembly routines assembled into oracle/_ref/libzstdref_amd64.so) with and without the BMI2 forms."""
    return list(oracle_goref.FLAVOURS) if oracle_goref.amd64_available() else ["generic"]


def _units():
    u = []
    for kind, first in (("T", 1), ("M", 2), ("J", 3), ("H", 4)):
        c = corpora.corpus(kind, 3, 131072, first_unit=first).tobytes()
        u += [c[:131072], c[131072:131072 + 70001], c[5:5 + 300000], c[1000:1000 + 4096]]
    u += corpora.edge_units()
    u += corpora.stress_units(seed=23, n=40)
    u += corpora.rle_literal_units(n=6, seed=5)[1]
    return u


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_encode_all(oracle, level):
    ref = oracle.ZstdOracle(level=level)
    bad = []
    for i, u in enumerate(_units()):
        if oracle_goref.zstd_encode_all(u, level=level) != ref.encode_all(u):
            bad.append((i, len(u)))
    assert not bad, "oracle differs from the reference's own Go encoder (unit, length): %r" % bad[:10]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_with_options(oracle, level):
    """The encoderOptions that change bytes: window size (block size follows it below 128 KiB), checksum off, single segment forced
    on / off, zero frames, WithNoEntropyCompression, WithAllLitEntropyCompression on / off, WithLowerEncoderMem."""
    t = corpora.corpus("T", 3, 131072, first_unit=6).tobytes()
    m = corpora.corpus("M", 2, 131072, first_unit=1).tobytes()
    units = [t[:131072], t[:200001], m[:90000], t[:3000], b"", t[:1], m[:1025]]
    cases = [dict(window_size=1 << 16), dict(window_size=1 << 20), dict(crc=False), dict(single=True), dict(single=False), dict(full_zero=False),
             dict(no_entropy=True), dict(all_lit_entropy=True), dict(all_lit_entropy=False), dict(low_mem=True), dict(window_size=1 << 10, crc=False)]
    bad = []
    for kw in cases:
        ref = oracle.ZstdOracle(level=level, **kw)
        for i, u in enumerate(units):
            if oracle_goref.zstd_encode_all(u, level=level, **kw) != ref.encode_all(u):
                bad.append((sorted(kw.items()), i, len(u)))
    assert not bad, bad[:10]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_with_a_raw_dictionary(oracle, level):
    """WithEncoderDictRaw: the dictionary content is history in front of every frame (fastEncoderDict, doubleFastEncoderDict,
    betterFastEncoderDict, bestFastEncoder.Reset with a dictionary)."""
    from compress_amd import _lib
    dct = _lib.corpus_fill("T", 0x5EED0005, 0, 1, 64 << 10).tobytes()
    t = corpora.corpus("T", 3, 131072, first_unit=2).tobytes()
    units = [t[:131072], t[:20000], t[7:7 + 250000], t[:100], dct[1000:9000] + t[:5000]]
    rd, runits = corpora.rle_literal_units(n=6, seed=9)   # blocks with an RLE literals section (huff0.ErrUseRLE) need their dictionary
    for d, us in ((dct, units), (dct[:4096], units), (dct[:9], units), (rd, runits)):
        ref = oracle.ZstdOracle(level=level, dict_id=7, dict_content=d)
        bad = [(len(d), i, len(u)) for i, u in enumerate(us) if oracle_goref.zstd_encode_all(u, level=level, dict_id=7, dict_content=d) != ref.encode_all(u)]
        assert not bad, bad


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_with_a_dictionary_above_one_mib(oracle, level):
    big = corpora.corpus("T", 12, 131072, seed=0x5EED0009).tobytes()
    t = corpora.corpus("T", 2, 131072, first_unit=31).tobytes()
    dct = big[:3 << 19]
    units = [t[:131072], dct[1000:60000] + t[:30000], dct[-50000:] + t[5:5000], t[:100]]
    ref = oracle.ZstdOracle(level=level, dict_id=9, dict_content=dct)
    bad = [(i, len(u)) for i, u in enumerate(units) if oracle_goref.zstd_encode_all(u, level=level, dict_id=9, dict_content=dct) != ref.encode_all(u)]
    assert not bad, bad


@pytest.mark.parametrize("level", [1, 2, 3, 4])
@pytest.mark.parametrize("concurrent", [0, 1])
def test_oracle_equals_the_translated_reference_streams(oracle, level, concurrent):
    """Write ... Flush ... Close streams (encoder.go:203-283 writeBlocks, :286-438 nextBlock, :547-570 Flush, :590-660 Close) in both
    forms of nextBlock — the asynchronous one and WithEncoderConcurrency(1)'s synchronous one —, with Flush points inside and on
    block boundaries, short streams that take the single-block EncodeAll shortcut, empty streams, and raw dictionaries (the two forms
    differ there: the synchronous one resets the block before its first Encode, encoder.go:371)."""
    t = corpora.corpus("T", 4, 131072, first_unit=11).tobytes()
    m = corpora.corpus("M", 3, 131072, first_unit=2).tobytes()
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    bs = 65536 if level == 1 else 131072
    cases = [(t[:300000], ()), (t[:300000], (70000, 70010, 200001)), (t[:bs], ()), (t[:bs], (bs,)), (t[:2 * bs], (bs,)), (t[:5000], ()),
             (t[:5000], (100, 4000)), (b"", ()), (b"", (0,)), (m[:250000], (1, 131072, 249999)), (t[:bs + 1], ()), (t[:3 * bs], (bs - 1, bs, bs + 1))]
    bad = []
    for kw in (dict(), dict(crc=False), dict(dict_id=3, dict_content=dct), dict(window_size=1 << 16)):
        ref = oracle.ZstdOracle(level=level, concurrent=concurrent, **kw)
        for i, (data, cuts) in enumerate(cases):
            if oracle_goref.zstd_encode_stream(data, cuts, level=level, concurrent=concurrent, **kw) != ref.encode_stream(data, cuts):
                bad.append((sorted(kw), i, len(data), cuts))
    assert not bad, bad[:10]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_a_reused_reference_encoder_writes_what_a_fresh_one_writes(oracle, level):
    """The premise of the device design — every EncodeAll unit is stateless — checked on the reference's OWN encoder objects: one
    zstd.Encoder / one pooled encoder re-used for a run of EncodeAll calls (its tables and `cur` carry over from call to call,
    encoder.go:722-729 + enc_*.go Reset) gives, frame for frame, the bytes of a fresh encoder per unit; also with a raw dictionary."""
    t = corpora.corpus("T", 8, 131072, first_unit=11).tobytes()
    m = corpora.corpus("M", 4, 131072, first_unit=2).tobytes()
    j = corpora.corpus("J", 4, 65536, first_unit=7).tobytes()
    base = [t[:131072], m[:200000], t[:100], b"", t[131072:262144], t[:131072], m[5:90000], t[:300001], j[:65536], j[:9], t[:131072]]
    units = base * (6 if level < 4 else 1)
    fresh = {}
    for u in base:
        fresh[u] = oracle.ZstdOracle(level=level).encode_all(u)  # (== the translated reference with a fresh encoder: the tests above)
    assert oracle_goref.zstd_encode_all_reuse(units, level=level) == [fresh[u] for u in units]
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    dref = oracle.ZstdOracle(level=level, dict_id=4, dict_content=dct)
    assert oracle_goref.zstd_encode_all_reuse(base, level=level, dict_id=4, dict_content=dct) == [dref.encode_all(u) for u in base]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_read_from(oracle, level):
    """Encoder.ReadFrom (encoder.go:444-496) and readFromJobs (:498-542): what is buffered from earlier Writes goes out as a block /
    a job first, then the source is read block-size (job-size) pieces at a time — for the bytes, a Flush point where ReadFrom takes
    over, which is how the oracle, the device entry points (`cuts`) and the façades model it."""
    t = corpora.corpus("T", 6, 131072, first_unit=11).tobytes()
    bad = []
    for data, cuts, a in ((t[:300000], (), 0), (t[:300000], (), 70000), (t[:300000], (1000,), 131072), (t[:5000], (), 100), (t[:700000], (5000,), 9000),
                          (b"", (), 0), (t[:262144], (), 262144), (t[:300000], (70000,), 70000), (t[:5000], (), 5000)):
        eff = tuple(sorted(set(cuts) | ({a} if 0 < a <= len(data) else set())))  # (a == len: ReadFrom of an empty source still flushes what is buffered)
        if oracle_goref.zstd_encode_stream(data, cuts, level=level, readfrom_at=a) != oracle.ZstdOracle(level=level).encode_stream(data, eff):
            bad.append(("blocks", len(data), cuts, a))
        if oracle_goref.zstd_encode_stream(data, cuts, level=level, concurrent=1, readfrom_at=a) != oracle.ZstdOracle(level=level, concurrent=1).encode_stream(data, eff):
            bad.append(("blocks, synchronous", len(data), cuts, a))
    win = 1 << 17
    e = oracle.ZstdOracle(level=level, window_size=win)
    big = t[:700000] if level == 4 else t
    for cuts, a in (((), 0), ((), 100000), ((600000,), 600001), ((1000,), 524288 + 1000), ((), len(big))):
        eff = tuple(sorted(set(cuts) | ({a} if 0 < a <= len(big) else set())))
        if oracle_goref.zstd_encode_stream(big, cuts, level=level, window_size=win, concurrent=4, jobs=True, readfrom_at=a) != e.encode_jobs(big, eff):
            bad.append(("jobs", len(big), cuts, a))
    assert not bad, bad


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_with_full_format_dictionaries(oracle, level):
    """WithEncoderDict: the reference's loadDict (zstd/dict.go:70-150 — huff0.ReadTable with its FSE-compressed weights, three
    readNCount tables, the offsets, the content) and what the encoders do with a loaded dictionary (offsets, content as history, the
    literal table as the first block's prevTable): the reference's own d0.dict fixture with inputs of its kind, a skewed dictionary
    whose table huff0 actually keeps (treeless literals), EncodeAll and — both forms of nextBlock — streams."""
    import test_oracle_kats as tk
    blob, ins = tk._dict_fixture(oracle)
    t = corpora.corpus("T", 2, 131072, first_unit=5).tobytes()
    units = list(ins)[:6 if level < 4 or FULL else 3] + [ins[1][:40], ins[1][:9], b"", t[:200000 if level < 4 or FULL else 70000], t[:31]]
    sblob, probs = tk.skewed_dict(blob)
    bad = []
    for name, b, us in (("d0", blob, units), ("skewed", sblob, units[:4] + tk.skewed_units(probs, seeds=3))):
        ref = oracle.ZstdOracle(level=level, dict_blob=b)
        for i, u in enumerate(us):
            if oracle_goref.zstd_encode_all(u, level=level, dict_blob=b) != ref.encode_all(u):
                bad.append((name, "all", i, len(u)))
        for conc in (0, 1):
            refs = oracle.ZstdOracle(level=level, dict_blob=b, concurrent=conc)
            for i, u in enumerate(us[:5]):
                cuts = (len(u) // 3, len(u) // 2) if len(u) > 10 else ()
                if oracle_goref.zstd_encode_stream(u, cuts, level=level, concurrent=conc, dict_blob=b) != refs.encode_stream(u, cuts):
                    bad.append((name, "stream", conc, i, len(u)))
    assert not bad, bad[:10]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_job_mode(oracle, level):
    """WithConcurrentBlocks (zstd/enc_jobs.go; encoder.go writeJobs / flushJobs / closeJobs): jobs of 4 x window with the previous
    job's tail as prefix (ResetPrefix of every encoder), the one-block shortcut, Flush points that dispatch short jobs, streams of
    exactly k jobs, the empty stream.  The reference's job cutting, per-job encode and frame assembly are translated as they stand;
    its worker goroutines are not (every job is compressed and written where it is dispatched: oracle/ref_go/manifest.py)."""
    t = corpora.corpus("T", 20, 131072, first_unit=3).tobytes()
    m = corpora.corpus("M", 9, 131072, first_unit=1).tobytes()
    bad = []
    for win in ((1 << 17, 1 << 18) if level < 4 or FULL else (1 << 17,)):  # jobs of 512 KiB / 1 MiB
        ref = oracle.ZstdOracle(level=level, window_size=win)
        js = max(4 * win, 512 << 10)
        datas = (t, m, t[:js], t[:2 * js], t[:js + 1], t[:js - 1], t[:70000], t[:100], b"") if level < 4 or FULL else (t[:js + 70000], t[:js], t[:70000], t[:100], b"")
        for di, data in enumerate(datas):
            for cuts in ((), (1000, 300000), (len(data),), (js // 2, js // 2 + 10, js + 77)):
                if level == 4 and len(data) > (1 << 20) and cuts:
                    continue  # (SpeedBestCompression on the long inputs once)
                got = oracle_goref.zstd_encode_stream(data, cuts, level=level, window_size=win, concurrent=4, jobs=True)
                if got != ref.encode_jobs(data, cuts):
                    bad.append((win, di, len(data), cuts))
    assert not bad, bad[:10]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_reference_decoder_accepts_the_oracles_frames_and_rejects_damage(oracle, level):
    """The reference's OWN zstd decoder (DecodeAll, pure-Go form: framedec / blockdec / seqdec / fse_decoder / huff0 decompress,
    translated) decodes what the oracle writes — EncodeAll frames of every corpus kind and size class, streams with Flush points,
    job-mode streams — back to the input, and refuses damaged frames where the in-repo decoder refuses them too."""
    ref = oracle.ZstdOracle(level=level, window_size=1 << 17)
    big = oracle.ZstdOracle(level=level)  # the level's own window: offset codes beyond the 56-bit fast path of the assembly (seqdec_asm.go:218)
    t = corpora.corpus("T", 5, 131072, first_unit=2).tobytes()
    units = [t[:131072], t[:300000], t[:100], b"", t[:70000]] + [corpora.corpus(k, 1, 131072, first_unit=3).tobytes() for k in "HJM"]
    units += corpora.stress_units(seed=7, n=6) + corpora.rle_literal_units(n=3, seed=2)[1][:0]
    frames = [(ref.encode_all(u), u) for u in units] + [(big.encode_all(t[:400000]), t[:400000])]
    frames.append((ref.encode_stream(t[:300000], (70000, 70010, 200001)), t[:300000]))
    frames.append((ref.encode_jobs(t[:640000], (1000, 600000)), t[:640000]))
    frame = bytearray(ref.encode_all(t[:131072]))
    rng = np.random.default_rng(level)
    damaged = []
    for pos in [5, 7, 12, len(frame) - 1, len(frame) - 5] + [int(x) for x in rng.integers(13, len(frame) - 6, 12)]:
        g = bytearray(frame)
        g[pos] ^= 1 << int(rng.integers(0, 8))
        damaged.append((pos, bytes(g)))
    for fl in _flavours():
        with oracle_goref.flavour(fl):
            for f, u in frames:
                assert oracle_goref.zstd_decode_all(f, len(u)) == u, (fl, len(u))
            for pos, g in damaged:
                try:
                    ok_ref = oracle_goref.zstd_decode_all(g, 131072 + 64) == t[:131072]
                except ValueError:
                    ok_ref = False
                try:
                    ok_own = oracle.zstd_decompress(g, 131072 + 64) == t[:131072]
                except Exception:
                    ok_own = False
                assert ok_ref == ok_own and not ok_ref, (fl, pos, ok_ref, ok_own)  # (a flipped bit always breaks the checksum if nothing else)


def _dict_cases(oracle):
    """(keyword arguments for the encoders / decoders, inputs): a raw dictionary, the reference's d0.dict fixture with inputs of its kind."""
    import test_oracle_kats as tk
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    t = corpora.corpus("T", 3, 131072, first_unit=8).tobytes()
    blob, ins = tk._dict_fixture(oracle)
    return [(dict(dict_id=7, dict_content=dct), [t[:131072], t[:300000], dct[100:9000] + t[:4000], t[:50]]),
            (dict(dict_blob=blob), list(ins)[:5] + [ins[1][:40], t[:70000]])]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_reference_decoder_reads_dictionary_frames(oracle, level):
    """WithDecoderDictRaw / WithDecoderDicts (decoder_options.go:112-141; history.setDict): the reference's decoder, in every build
    flavour, returns the input of the oracle's dictionary frames, and refuses them without the dictionary (`unknown dictionary`)."""
    for kw, units in _dict_cases(oracle):
        okw = dict(dict_id=kw["dict_id"], dict_content=kw["dict_content"]) if "dict_id" in kw else dict(dict_blob=kw["dict_blob"])
        ref = oracle.ZstdOracle(level=level, **okw)
        frames = [ref.encode_all(u) for u in units]
        for fl in _flavours():
            with oracle_goref.flavour(fl):
                for f, u in zip(frames, units):
                    assert oracle_goref.zstd_decode_all(f, len(u), **kw) == u, (fl, len(u))
                with pytest.raises(ValueError, match="unknown dictionary"):
                    oracle_goref.zstd_decode_all(frames[0], len(units[0]))


@pytest.mark.skipif(not oracle_goref.amd64_available(), reason="the amd64 flavour of oracle/_ref needs an x86-64 host")
@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_the_references_amd64_build_writes_and_reads_the_same_frames(oracle, level):
    """The amd64 build of the reference — matchLen of zstd/matchlen_amd64.s inside all four encoders; sequenceDecs_decode[_56] /
    decodeSync[_safe] / executeSimple[_safe] of zstd/seqdec_amd64.s in their plain and BMI2 forms, buildDtable_asm of
    zstd/fse_decoder_amd64.s, the 1X / 4X loops of huff0/decompress_amd64.s inside its decoder — against its portable build: the
    same frames out of EncodeAll / streams / dictionaries, and on 400 damaged frames per level the same verdict and, where a frame is
    still accepted (no checksum), the same bytes."""
    t = corpora.corpus("T", 4, 131072, first_unit=21).tobytes()
    m = corpora.corpus("M", 2, 131072, first_unit=4).tobytes()
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    units = [t[:131072], t[:300001], m[:200000], t[:4000], t[:9], b""] + corpora.stress_units(seed=31, n=12)
    want = [oracle_goref.zstd_encode_all(u, level=level) for u in units]
    wantd = [oracle_goref.zstd_encode_all(u, level=level, dict_id=5, dict_content=dct) for u in units[:4]]
    wants = oracle_goref.zstd_encode_stream(t[:400000], (100000, 100001, 333333), level=level)
    nocrc = oracle_goref.zstd_encode_all(t[:200000], level=level, crc=False, window_size=1 << 17)
    rng = np.random.default_rng(100 + level)
    damaged = []
    for _ in range(400):
        g = bytearray(nocrc)
        for _k in range(int(rng.integers(1, 3))):
            g[int(rng.integers(4, len(g)))] ^= 1 << int(rng.integers(0, 8))
        damaged.append(bytes(g))

    def verdict(g):
        try:
            return oracle_goref.zstd_decode_all(g, 300000)
        except ValueError:
            return None
    base = [verdict(g) for g in damaged]
    assert any(b is not None for b in base) and any(b is None for b in base)
    for fl in ("amd64", "amd64-nobmi"):
        with oracle_goref.flavour(fl):
            assert [oracle_goref.zstd_encode_all(u, level=level) for u in units] == want, fl
            assert [oracle_goref.zstd_encode_all(u, level=level, dict_id=5, dict_content=dct) for u in units[:4]] == wantd, fl
            assert oracle_goref.zstd_encode_stream(t[:400000], (100000, 100001, 333333), level=level) == wants, fl
            for f, u in zip(want, units):
                assert oracle_goref.zstd_decode_all(f, len(u)) == u, (fl, len(u))
            got = [verdict(g) for g in damaged]
            diff = [i for i in range(len(damaged)) if got[i] != base[i]]
            assert not diff, (fl, diff[:5])


def _ref_inputs(limit):
    out = []
    for name in ("encode-corpus-raw.zip", "comp-crashers.zip", "enc_regressions.zip"):
        p = os.path.join(REFIN, name)
        if not os.path.exists(p):
            continue
        with zipfile.ZipFile(p) as z:
            names = sorted(z.namelist())
            step = max(1, len(names) // limit)
            for n in names[::step]:
                if not n.endswith("/"):
                    out.append((name + ":" + n, z.read(n)))
    for f in ("e.txt", "gettysburg.txt", "Mark.Twain-Tom.Sawyer.txt", "sharnd.out", "pi.txt", "html.txt", "pngdata.bin"):
        p = os.path.join(REFIN, f)
        if os.path.exists(p):
            out.append((f, open(p, "rb").read()))
    return out


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_equals_the_translated_reference_on_the_references_own_inputs(oracle, level):
    """The reference's fuzz corpora, regression inputs and shared testdata files (tests/golden/ref_inputs): a sample of ~150 per zip at
    the three fast levels, ~20 at SpeedBestCompression (KC_TEST_FULL=1: 300 / 60)."""
    ref = oracle.ZstdOracle(level=level)
    bad = []
    n = 0
    for name, data in _ref_inputs((300 if FULL else 150) if level < 4 else (60 if FULL else 20)):
        if len(data) > (4 << 20):
            continue
        n += 1
        if oracle_goref.zstd_encode_all(data, level=level) != ref.encode_all(data):
            bad.append((name, len(data)))
    assert n > (100 if level < 4 else 40) and not bad, bad[:10]


_S2 = {0: "s2_encode", 1: "s2_encode_better", 2: "s2_encode_snappy", 3: "s2_encode_snappy_better", 4: "s2_encode_best", 5: "s2_encode_snappy_best"}


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5])
def test_oracle_equals_the_translated_reference_s2(oracle, level):
    """s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter / EncodeBest / EncodeSnappyBest, portable Go form: blocks of every
    size class (below 32 bytes: literals only; 64 KiB; above 64 KiB; 1 MiB), four corpora, edge and stress blocks, the reference's
    enc_regressions inputs."""
    blocks = []
    for kind in "JTMH":
        c = corpora.corpus(kind, 2, 1 << 20, first_unit=3).tobytes()
        blocks += [c[:65536], c[65536:65536 + 65537], c[:300000], c[100:131], c[:32], c[:1 << 20] if level < 4 else c[:200000], c[7000:7000 + 5000]]
    blocks += corpora.edge_units() + corpora.stress_units(seed=31, n=24)
    p = os.path.join(REFIN, "enc_regressions.zip")
    if os.path.exists(p):
        with zipfile.ZipFile(p) as z:
            blocks += [z.read(n) for n in sorted(z.namelist())[::4] if not n.endswith("/")]
    fn = getattr(oracle, _S2[level])
    bad = [(i, len(b)) for i, b in enumerate(blocks) if len(b) < (4 << 20) and oracle_goref.s2_encode(b, level) != fn(b)]
    assert not bad, bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, "1L", 2, 3, 4])
def test_device_equals_the_translated_reference_encode_all(kclib, level):
    """The HIP path against the reference's own Go encoder directly (no hand-written oracle in between)."""
    from compress_amd import zstd
    lv = 1 if level == "1L" else level
    opts = [zstd.WithEncoderLevel(lv)] + ([zstd.WithMatchPath("lds")] if level == "1L" else ([zstd.WithMatchPath("hbm")] if level == 1 else []))
    units = [u for u in _units() if len(u) <= 300000][:60 if lv < 4 else 24]
    buf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, *opts)
    out, out_off = enc.EncodeUnits(buf, off)
    enc.Close()
    bad = [(i, len(u)) for i, u in enumerate(units) if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != oracle_goref.zstd_encode_all(u, level=lv)]
    assert not bad, bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 2, 3, 4])
@pytest.mark.parametrize("concurrent", [0, 1])
def test_device_equals_the_translated_reference_streams(kclib, level, concurrent):
    """The device's Write / Flush / Close streams (kc_zstd_encode_streams_cuts) against the reference's own streaming writer, both
    forms of nextBlock, with and without a raw dictionary."""
    from compress_amd import zstd
    t = corpora.corpus("T", 4, 131072, first_unit=11).tobytes()
    dct = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    bs = 65536 if level == 1 else 131072
    cases = [(t[:300000], ()), (t[:300000], (70000, 70010, 200001)), (t[:bs], (bs,)), (t[:5000], (100, 4000)), (b"", ()), (t[:3 * bs], (bs - 1, bs, bs + 1))]
    buf, off = corpora.pack_units([c[0] for c in cases])
    for with_dict in (False, True):
        opts = [zstd.WithEncoderLevel(level)] + ([zstd.WithEncoderConcurrency(1)] if concurrent else []) + ([zstd.WithEncoderDictRaw(3, dct)] if with_dict else [])
        enc = zstd.NewWriter(None, *opts)
        out, out_off = enc.EncodeStreams(buf, off, flush_at=[c[1] for c in cases])
        enc.Close()
        kw = dict(dict_id=3, dict_content=dct) if with_dict else {}
        bad = [(with_dict, i, len(u)) for i, (u, cuts) in enumerate(cases)
               if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != oracle_goref.zstd_encode_stream(u, cuts, level=level, concurrent=concurrent, **kw)]
        assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_device_equals_the_translated_reference_job_mode_and_full_dictionaries(oracle, kclib, level):
    """kc_zstd_encode_jobs against the reference's WithConcurrentBlocks stream, and EncodeAll / streams with a full-format dictionary
    (the reference's d0.dict fixture) against the reference's loadDict + encoders."""
    from compress_amd import zstd
    import test_oracle_kats as tk
    t = corpora.corpus("T", 10, 131072, first_unit=3).tobytes()
    win = 1 << 17
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(level), zstd.WithConcurrentBlocks(True), zstd.WithEncoderConcurrency(4), zstd.WithWindowSize(win))
    js = enc.JobSize()
    for data, cuts in ((t[:js + 70000], ()), (t[:2 * js], (1000, 300000)), (t[:js], (js,)), (t[:100], ()), (t[:70000], (50,))):
        assert enc.EncodeJobs(data, cuts) == oracle_goref.zstd_encode_stream(data, cuts, level=level, window_size=win, concurrent=4, jobs=True), (len(data), cuts)
    enc.Close()
    blob, ins = tk._dict_fixture(oracle)
    units = list(ins)[:4] + [ins[1][:40], b"", t[:70000]]
    buf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(level), zstd.WithEncoderDict(blob))
    out, out_off = enc.EncodeUnits(buf, off)
    bad = [i for i, u in enumerate(units) if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != oracle_goref.zstd_encode_all(u, level=level, dict_blob=blob)]
    assert not bad, bad
    cuts = [((len(u) // 3, len(u) // 2) if len(u) > 10 else ()) for u in units]
    sout, soff = enc.EncodeStreams(buf, off, flush_at=cuts)
    bad = [i for i, u in enumerate(units) if sout[int(soff[i]):int(soff[i + 1])].tobytes() != oracle_goref.zstd_encode_stream(u, cuts[i], level=level, dict_blob=blob)]
    assert not bad, bad
    enc.Close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, "1L", 2, 3, 4])
def test_reference_decoder_accepts_every_device_frame(kclib, level):
    """What the DEVICE writes — EncodeAll frames at every level and kernel family, streams with Flush points, a job-mode stream — is
    decoded back to the input by the reference's own zstd decoder (translated; the S2 counterpart is
    tests/test_ref_s2asm.py::test_reference_decoder_accepts_every_device_level)."""
    from compress_amd import zstd
    lv = 1 if level == "1L" else level
    opts = [zstd.WithEncoderLevel(lv)] + ([zstd.WithMatchPath("lds")] if level == "1L" else ([zstd.WithMatchPath("hbm")] if level == 1 else []))
    t = corpora.corpus("T", 5, 131072, first_unit=2).tobytes()
    units = [t[:131072], t[:300000], t[:100], t[:70000]] + [corpora.corpus(k, 1, 131072, first_unit=3).tobytes() for k in "HJM"] + corpora.stress_units(seed=7, n=6)
    buf, off = corpora.pack_units(units)
    enc = zstd.NewWriter(None, *opts)
    out, out_off = enc.EncodeUnits(buf, off)
    cuts = [((len(u) // 3, len(u) // 2) if len(u) > 10 else ()) for u in units]
    sout, soff = enc.EncodeStreams(buf, off, flush_at=cuts)
    enc.Close()
    jenc = zstd.NewWriter(None, zstd.WithEncoderLevel(lv), zstd.WithConcurrentBlocks(True), zstd.WithEncoderConcurrency(4), zstd.WithWindowSize(1 << 17))
    jframe = jenc.EncodeJobs(t[:640000], (1000, 600000))
    jenc.Close()
    # dictionary frames: a raw dictionary and the reference's d0.dict (EncodeAll; C5 is a dictionary configuration)
    import oracle_lib
    dframes = []
    for kw, dunits in _dict_cases(oracle_lib):
        dopts = [zstd.WithEncoderDictRaw(kw["dict_id"], kw["dict_content"])] if "dict_id" in kw else [zstd.WithEncoderDict(kw["dict_blob"])]
        denc = zstd.NewWriter(None, *(opts + dopts))
        dbuf, doff = corpora.pack_units(dunits)
        dout, dout_off = denc.EncodeUnits(dbuf, doff)
        denc.Close()
        dframes.append((kw, dunits, dout, dout_off))
    for fl in _flavours():  # the portable decoder, and the amd64 build's assembly decoders with and without BMI2
        with oracle_goref.flavour(fl):
            for i, u in enumerate(units):
                assert oracle_goref.zstd_decode_all(out[int(out_off[i]):int(out_off[i + 1])].tobytes(), len(u)) == u, (fl, i, len(u))
                assert oracle_goref.zstd_decode_all(sout[int(soff[i]):int(soff[i + 1])].tobytes(), len(u)) == u, (fl, i, len(u))
            assert oracle_goref.zstd_decode_all(jframe, 640000) == t[:640000], fl
            for kw, dunits, dout, dout_off in dframes:
                for i, u in enumerate(dunits):
                    assert oracle_goref.zstd_decode_all(dout[int(dout_off[i]):int(dout_off[i + 1])].tobytes(), len(u), **kw) == u, (fl, sorted(kw), i, len(u))


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5])
def test_device_equals_the_translated_reference_s2(kclib, level):
    from compress_amd import s2
    blocks = []
    for kind in "JTM":
        c = corpora.corpus(kind, 8, 65536, first_unit=5).tobytes()
        blocks += [c[i * 65536:(i + 1) * 65536] for i in range(8 if level < 4 else 3)] + [c[:200000], c[9:40], c[:5000]]
    buf, off = corpora.pack_units(blocks)
    enc = s2.BlockEncoder(level=level)
    out, out_off = enc.EncodeBlocks(buf, off)
    enc.Close()
    bad = [(i, len(b)) for i, b in enumerate(blocks) if out[int(out_off[i]):int(out_off[i + 1])].tobytes() != oracle_goref.s2_encode(b, level)]
    assert not bad, bad[:10]


@pytest.mark.parametrize("level", [1, 2, 3])
def test_zstd_facade_host_logic_equals_the_references_writer(oracle, level):
    """Host logic of compress_amd.zstd.Encoder that needs no device — how Write / Flush / ReadFrom / Close sequences become the stream and
    its block (job) boundaries — against the reference's own Writer for the same call sequences.  The device entry points it hands
    the stream to (EncodeStreams, EncodeJobs) are replaced by the oracle here; device == oracle on them is the GPU suite's business."""
    import io
    from compress_amd import zstd
    t = corpora.corpus("T", 8, 131072, first_unit=31).tobytes()

    def facade(data, ops, jobs=False):
        opts = [zstd.WithEncoderLevel(level)] + ([zstd.WithWindowSize(1 << 17), zstd.WithConcurrentBlocks(True), zstd.WithEncoderConcurrency(4)] if jobs else [])
        sink = io.BytesIO()
        w = zstd.NewWriter(sink, *opts)
        ref = oracle.ZstdOracle(level=level, window_size=1 << 17) if jobs else oracle.ZstdOracle(level=level)
        w.EncodeStreams = lambda src, off, flush_at=None: (np.frombuffer(ref.encode_stream(bytes(src), tuple(flush_at[0]) if flush_at else ()), dtype=np.uint8), None)
        w.EncodeJobs = lambda d, cuts: ref.encode_jobs(bytes(d), tuple(cuts))
        pos = 0
        for op, arg in ops:
            if op == "write":
                w.Write(data[pos:arg])
                pos = arg
            elif op == "flush":
                w.Flush()
            else:
                w.ReadFrom(io.BytesIO(data[pos:]))
                pos = len(data)
        w.Close()
        return sink.getvalue()
    data = t[:700000]
    cases = [([("write", len(data))], (), None), ([("write", 70000), ("flush", 0), ("write", len(data))], (70000,), None),
             ([("write", 70000), ("readfrom", 0)], (), 70000), ([("write", 100), ("flush", 0), ("write", 5000), ("readfrom", 0)], (100,), 5000),
             ([("readfrom", 0)], (), 0), ([("write", 131072), ("flush", 0), ("flush", 0), ("write", 131073), ("flush", 0), ("write", len(data))], (131072, 131073), None)]
    for ops, cuts, rf in cases:
        assert facade(data, ops) == oracle_goref.zstd_encode_stream(data, cuts, level=level, readfrom_at=rf), (ops, "blocks")
        assert facade(data, ops, jobs=True) == oracle_goref.zstd_encode_stream(data, cuts, level=level, window_size=1 << 17, concurrent=4, jobs=True, readfrom_at=rf), (ops, "jobs")


def test_size_bounds_of_the_boundary_equal_the_references(oracle, kclib):
    """kc_zstd_max_encoded_size == (*Encoder).MaxEncodedSize and kc_s2_max_encoded_len == s2.MaxEncodedLen of the reference itself
    (translated): the caller-side buffer arithmetic of the drop-in (include/kcgpu.h), over sizes around every block / window boundary."""
    import ctypes as C
    from compress_amd import zstd
    sizes = sorted(set([0, 1, 9, 10, 100, 65535, 65536, 65537, 131071, 131072, 131073, 1 << 20, (1 << 20) + 1, (4 << 20) - 1, 4 << 20, (8 << 20) + 7, 1 << 27, (1 << 30) - 1]
                       + [int(x) for x in np.random.default_rng(3).integers(0, 1 << 26, 200)]))
    for level in (1, 2, 3, 4):
        for win in (None, 1 << 10, 1 << 16, 1 << 17, 1 << 20, 1 << 23):
            opts = [zstd.WithEncoderLevel(level)] + ([zstd.WithWindowSize(win)] if win else [])
            e = zstd.NewWriter(None, *opts)
            bad = [n for n in sizes if e.MaxEncodedSize(n) != oracle_goref.zstd_max_encoded_size(n, level=level, window_size=win)]
            assert not bad, (level, win, bad[:5])
    for n in sizes + [(1 << 32) - 1, 1 << 32, 0xFFFFFFFF - 100, 6 * (1 << 29)]:
        assert int(kclib.kc_s2_max_encoded_len(C.c_int64(n))) == oracle_goref.s2_max_encoded_len(n), n


def test_job_geometry_and_padding_arithmetic_equal_the_references(kclib):
    """kc_zstd_job_size / kc_zstd_overlap_size == encoderOptions.jobSize() / overlapSize(); the façades' padding arithmetic ==
    calcSkippableFrame of the reference's zstd and s2 packages (the zstd one needs 8 bytes of frame header, the s2 one 4)."""
    from compress_amd import zstd, s2
    for level in (1, 2, 3, 4):
        for win in (None, 1 << 10, 1 << 16, 1 << 17, 1 << 19, 1 << 20, 1 << 23, 1 << 25):
            e = zstd.NewWriter(None, zstd.WithEncoderLevel(level), *([zstd.WithWindowSize(win)] if win else []))
            assert (e.JobSize(), e.OverlapSize()) == oracle_goref.zstd_job_geometry(level, win), (level, win)
    rng = np.random.default_rng(8)
    for _ in range(3000):
        mult = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 512, 4096, 65536, int(rng.integers(1, 1 << 22))]))
        written = int(rng.choice([0, 1, mult - 1, mult, mult + 1, int(rng.integers(0, 1 << 33))]))
        assert zstd.calc_skippable_frame(written, mult) == oracle_goref.calc_skippable_frame(written, mult), (written, mult)
        assert s2.calc_skippable_frame(written, mult) == oracle_goref.calc_skippable_frame(written, mult, s2=True), (written, mult)
