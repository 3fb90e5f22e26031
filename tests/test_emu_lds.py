"""The LDS-table kernels (kc_s2_lds.hip, kc_zstd_match_lds.hip) run on the CPU wave emulator and compared with the oracle.

No GPU needed: the .hip sources are compiled by g++ against tools/hipemu (see tests/emu_lib.py).  These tests pin the
device ALGORITHM (speculative rounds, marker-byte conflict detection, ordered commit, emit paths) bit for bit; the
`-m gpu` tests run the same sources on the device through the C ABI."""
import os
import zipfile

import numpy as np
import pytest

import corpora
import emu_lib
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
# KC_TEST_FULL=1: the long forms of the emulated cases (about four more minutes; the default run keeps every kernel, level, path and
# edge shape but takes the slowest emulations — speculation width 1, SpeedBestCompression, SpeedBetterCompression — on shorter inputs)
FULL = os.environ.get("KC_TEST_FULL", "0") == "1"


def _s2_blocks(small=False):
    if small and not FULL:
        blocks = [corpora.corpus("J", 1, 65536, first_unit=3).tobytes(), corpora.corpus("T", 1, 65536).tobytes()[:30000],
                  corpora.corpus("H", 1, 65536).tobytes()[:20000], corpora.corpus("M", 1, 65536, first_unit=3).tobytes()[:30000]]
        blocks += corpora.edge_units()
        blocks += [u[:70000] for u in corpora.stress_units(seed=11, n=2)]
        return blocks
    blocks = [corpora.corpus("J", 1, 65536, first_unit=3).tobytes(), corpora.corpus("T", 1, 65536).tobytes(),
              corpora.corpus("H", 1, 65536).tobytes(), corpora.corpus("M", 1, 65536, first_unit=2).tobytes(),
              corpora.corpus("M", 1, 65536, first_unit=3).tobytes(), corpora.corpus("J", 1, 40000, first_unit=9).tobytes()]
    blocks += corpora.edge_units()
    blocks += [u[:70000] for u in corpora.stress_units(seed=11, n=6)]
    return blocks


def _cmp(blocks, got, ref_fn):
    bad = []
    for i, b in enumerate(blocks):
        ref = ref_fn(b)
        if ref != got[i]:
            k = next((j for j in range(min(len(ref), len(got[i]))) if ref[j] != got[i][j]), -1)
            bad.append((i, len(b), len(ref), len(got[i]), k))
    assert not bad, "blocks differing from the oracle (index, len, oracle bytes, emulated bytes, first differing byte): %r" % bad[:8]


@pytest.mark.parametrize("w0", [0, 1, 8, 64])
def test_s2_lds_blocks_bit_exact(w0):
    """s2.Encode through kc_s2_encode_lds_kernel<0, *>: blocks below and above 64 KiB (LDS / global source), every
    speculation width (0 = blocks in LDS take the fused wave-uniform step, 1 = its first form, longer ones one-step rounds; 64 = a whole wave
    of steps per round)."""
    blocks = _s2_blocks()
    _cmp(blocks, emu_lib.s2_encode_blocks(blocks, level=0, spec_w0=w0), oracle_lib.s2_encode)


@pytest.mark.parametrize("w0", [0, 1, 8])
def test_s2_lds_snappy_bit_exact(w0):
    blocks = _s2_blocks(small=w0 == 1)
    _cmp(blocks, emu_lib.s2_encode_blocks(blocks, level=2, spec_w0=w0), oracle_lib.s2_encode_snappy)


@pytest.mark.parametrize("w0", [0, 1, 8])
def test_s2_lds_framed_chunks_bit_exact(w0):
    """Framed mode: chunk header + masked CRC32C (wave-parallel CRC with the advance-by-zeros combine) + body."""
    blocks = [b for b in _s2_blocks(small=w0 == 1) if len(b) > 0]
    buf, off = corpora.pack_units(blocks)
    ref, ro = oracle_lib.s2_encode_stream(buf, off, with_stream_id=False)
    got = emu_lib.s2_encode_blocks(blocks, level=0, framed=True, spec_w0=w0)
    for i in range(len(blocks)):
        r = ref[int(ro[i]):int(ro[i + 1])].tobytes()
        assert r == got[i], "chunk %d (len %d): header %r vs %r" % (i, len(blocks[i]), r[:8], got[i][:8])


def _crc32c(b):
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    c = 0xFFFFFFFF
    for x in b:
        c = t[(c ^ x) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def uncompressed_chunk(b):
    """s2.Writer with WriterUncompressed (writer.go:414-451 with encodeBlock returning 0): 0x01 | len24(4 + n) | masked CRC32C | bytes."""
    c = _crc32c(b)
    m = (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF
    n = 4 + len(b)
    return bytes([1, n & 0xFF, (n >> 8) & 0xFF, (n >> 16) & 0xFF]) + m.to_bytes(4, "little") + b


def test_s2_lds_uncompressed_chunks():
    """s2.WriterUncompressed: every block an uncompressed chunk (the kernels' stored path forced), blocks of every size class."""
    blocks = [b for b in _s2_blocks(small=True) if len(b) > 0]
    got = emu_lib.s2_encode_blocks(blocks, level=0, framed=True, spec_w0=0, stored_only=True)
    for i, b in enumerate(blocks):
        assert got[i] == uncompressed_chunk(b), (i, len(b))


@pytest.mark.parametrize("w0", [0, 1, 8])
def test_s2_lds_reference_regressions(w0):
    """The reference's own encoder regression inputs (s2/testdata/enc_regressions.zip, committed copy)."""
    zp = os.path.join(HERE, "golden", "ref_inputs", "enc_regressions.zip")
    if not os.path.exists(zp):
        pytest.skip("no committed copy of enc_regressions.zip")
    with zipfile.ZipFile(zp) as z:
        blocks = [z.read(n) for n in z.namelist() if not n.endswith("/")]
    blocks = [b for b in blocks if len(b) < (1 << 20)]
    _cmp(blocks, emu_lib.s2_encode_blocks(blocks, level=0, spec_w0=w0), oracle_lib.s2_encode)


def _zfast_units(big=True):
    big = big or FULL
    units = [corpora.corpus("T", 1, 131072, first_unit=k).tobytes() for k in range(2)]
    units += [corpora.corpus("M", 1, 131072, first_unit=k).tobytes() for k in range(4 if big else 2)]
    units += [corpora.corpus("J", 1, 65536, first_unit=3).tobytes(), corpora.corpus("H", 1, 131072).tobytes()]
    units += [u for u in corpora.edge_units() if 0 < len(u) < 262000]
    units += corpora.stress_units(seed=5, n=8 if big else 4)          # up to 300001 bytes: five blocks with history
    if big:
        units += [corpora.corpus("T", 8, 131072, first_unit=77).tobytes()]  # one 1 MiB unit: 16 blocks, positions beyond 2^18
    return units


@pytest.mark.parametrize("w0", [0, 1, 8, 64])
def test_zfast_lds_parse_matches_oracle(w0):
    """kc_zfast_match_lds_kernel: the sequence list of every block equals the oracle's fastEncoder (EncodeNoHist for
    one-block units, Encode with history for longer ones), at every speculation width; 0: units up to 128 KiB through the instantiation
    with the source in a 64 KiB LDS ring and the untagged 17-bit table, the longer ones of the same launch through the tagged form."""
    units = _zfast_units(big=w0 in (0, 64))
    got = emu_lib.zfast_parse(units, spec_w0=w0)
    bi = 0
    for ui, u in enumerate(units):
        ref = oracle_lib.zstd_parse_unit(u, level=1)
        for rb, (rseqs, rlits) in enumerate(ref):
            gseqs, gnlit, gextra, gflags = got[bi]
            bi += 1
            assert len(gseqs) == len(rseqs), "unit %d (len %d) block %d: nseq %d vs oracle %d" % (ui, len(u), rb, len(gseqs), len(rseqs))
            if len(rseqs):
                neq = np.nonzero((gseqs != rseqs).any(axis=1))[0]
                assert len(neq) == 0, "unit %d block %d first differing seq %d: emulated %r oracle %r" % (ui, rb, neq[0], gseqs[neq[0]], rseqs[neq[0]])
            assert gnlit == len(rlits), "unit %d block %d: nlit %d vs oracle %d" % (ui, rb, gnlit, len(rlits))
    assert bi == len(got)


@pytest.mark.parametrize("snappy", [False, True])
def test_s2_best_kernel_bit_exact(snappy):
    """kc_s2_best_kernel (s2.EncodeBest / s2.EncodeSnappyBest: candidates of a phase evaluated one per lane, folded in the
    reference's order with its same-offset rule; chain-ordered table updates, 64 positions per pass) against the oracle."""
    blocks = [corpora.corpus("J", 1, 65536, first_unit=3).tobytes(), corpora.corpus("T", 1, 65536).tobytes(), corpora.corpus("H", 1, 20000).tobytes(),
              corpora.corpus("M", 1, 65536, first_unit=2).tobytes(), corpora.corpus("T", 2, 131072, first_unit=9).tobytes()[:150000 if FULL else 80000]]
    blocks += [u for u in corpora.edge_units() if len(u) <= 70000]
    blocks += [u[:40000] for u in corpora.stress_units(seed=13, n=6 if FULL else 3)]
    _cmp(blocks, emu_lib.s2_best_blocks(blocks, snappy=snappy), oracle_lib.s2_encode_snappy_best if snappy else oracle_lib.s2_encode_best)


@pytest.mark.parametrize("w0,grow,xseg", [(1, 1, 0), (1, 1, 2), (1, 1, 1 << 20), (8, 0, 0), (2, 2, 1 << 20), (1, 1, 1 << 24)])
def test_zfast_grp_parse_matches_oracle(w0, grow, xseg):
    """kc_zfast_match_grp_kernel<8> (the HBM-table throughput kernel, 8 lanes per unit): every block's sequence list equals the
    oracle's fastEncoder — at every speculation policy, with rounds confined to one skip segment (round 2), crossing segments
    always, or only once the step has grown (xseg_k); with the "nothing written there yet" filter of units that have no sequence yet
    (default) and without it (bit 24 of the last argument)."""
    units = _zfast_units()
    _cmp_parse(units, emu_lib.zfast_parse_grp(units, spec_w0=w0, spec_grow=grow, xseg_k=xseg), level=1)


def test_zfast_grp_epoch_stamped_tables():
    """The table arena is not cleared between launches: entries carry the launch's stamp, and what earlier launches left in a slot
    (other units' positions and tags) reads as empty.  Same slots, other units, consecutive stamps — and the stamp's bits taken
    from the tag at the position width of 128 KiB units and of a 1 MiB unit."""
    sets = [_zfast_units()[:8], _zfast_units()[4:12], list(reversed(_zfast_units()[:8])),
            [corpora.corpus("T", 8, 131072, first_unit=77).tobytes()] + _zfast_units()[:3]]
    for ep in (1, 2, 3, 14, 15):  # (the wrap of the stamp is the host's business: tests/test_gpu_zstd.py)
        units = sets[ep % len(sets)]
        _cmp_parse(units, emu_lib.zfast_parse_grp(units, epoch=ep, slots=21, fresh=ep == 1), level=1)


def _cmp_parse(units, got, **okw):
    bi = 0
    for ui, u in enumerate(units):
        if len(u) == 0:
            continue
        ref = oracle_lib.zstd_parse_unit(u, **okw)
        for rb, (rseqs, rlits) in enumerate(ref):
            gseqs, gnlit, gextra, gflags = got[bi]
            bi += 1
            k = min(len(gseqs), len(rseqs))
            neq = np.nonzero((gseqs[:k] != rseqs[:k]).any(axis=1))[0] if k else []
            assert len(neq) == 0, "unit %d (len %d) block %d first differing seq %d: emulated %r oracle %r" % (ui, len(u), rb, neq[0], gseqs[neq[0]], rseqs[neq[0]])
            assert len(gseqs) == len(rseqs), "unit %d (len %d) block %d: nseq %d vs oracle %d" % (ui, len(u), rb, len(gseqs), len(rseqs))
            assert gnlit == len(rlits), "unit %d block %d: nlit %d vs oracle %d" % (ui, rb, gnlit, len(rlits))
    assert bi == len(got)


def _zbest_units():
    c = (lambda n: n) if FULL else (lambda n: n // 2)
    units = [corpora.corpus("T", 1, 131072, first_unit=1).tobytes()[:c(100000)], corpora.corpus("M", 1, 131072, first_unit=2).tobytes()[:c(90000)],
             corpora.corpus("J", 1, 65536, first_unit=3).tobytes()[:c(65536)], corpora.corpus("H", 1, 40000).tobytes()[:c(40000)]]
    # (tiny alphabets make the longest candidate chains: 27 s of emulation for 128 KiB of two symbols — a fraction of that is plenty here)
    units += [u[:c(48000)] if len(set(u[:4096])) <= 4 else u[:c(140000)] for u in corpora.edge_units() if 0 < len(u) < 140000]
    units += [u[:140000 if (FULL or k == 0) else 40000] for k, u in enumerate(corpora.stress_units(seed=5, n=3))]   # two blocks with history
    units += [corpora.corpus("T", 2, 131072, first_unit=77).tobytes()[:150000 if FULL else 135000]]
    # long repeats: matches beyond goodEnough, repeat-offset forms straight after a match, period-1..7 runs
    rng = np.random.default_rng(3)
    pat = bytes(rng.integers(0, 256, 700, dtype=np.uint8))
    units += [(pat * 40)[:24000], b"abcabcabcabd" * 1500, b"".join(pat[:k] * 9 for k in range(3, 60)), b"x" * 5000 + pat + b"x" * 300 + pat[:100] + b"y" * 40]
    return units


def test_zbest_parse_matches_oracle():
    """kc_zbest_match_kernel (SpeedBestCompression): the sequence list of every block equals the oracle's bestFastEncoder — the
    candidates of a phase priced one per lane, improve()'s order-dependent part replayed in the reference's order, the entropy
    estimate through the restated math.Log2 — on two persistent table slots that many units pass through."""
    units = _zbest_units()
    got = emu_lib.zbest_parse_fresh(units, n_slots=2)
    _cmp_parse(units, got, level=4)


def test_zbest_parse_history_forms():
    """The same with history in front of the units: a raw-content dictionary (Reset with a dictionary: every position of it in
    the long table, the short one filled four positions per step), a full-format dictionary's repeat offsets, a job's overlap
    prefix (ResetPrefix), stream mode, a small window (matches beyond it refused), and a window so large that the slot is cleared
    before every unit (the cur wrap-around path)."""
    t = corpora.corpus("T", 2, 131072, first_unit=11).tobytes()
    dct = corpora.corpus("T", 1, 20000, seed=0x5EED0005).tobytes()
    units = [t[:30000], t[100:9000], dct[500:9000] + t[:100], t[:7], t[131072:131072 + 140000]]
    got = emu_lib.zbest_parse(units, n_slots=1, hist=dct, fresh=True)
    _cmp_parse(units, got, level=4, dict_id=9, dict_content=dct)
    for hl in ((8, 9, 12, 13, 20) if FULL else (8, 13, 20)):  # the edges of the two index ranges
        got = emu_lib.zbest_parse(units[:2], n_slots=1, hist=dct[:hl])
        _cmp_parse(units[:2], got, level=4, dict_id=9, dict_content=dct[:hl])
    # stream mode (Encode from the first block on) and a small window
    u2 = [t[:200000 if FULL else 120000], t[:65536]]
    got = emu_lib.zbest_parse(u2, n_slots=2, window=1 << 15, block_size=1 << 15, stream_mode=1)
    _cmp_parse(u2, got, level=4, window_size=1 << 15, block_size=1 << 15)
    # the clear path: window 2^29 -> bufferReset 2^30, reached by the second unit of a slot
    u3 = [t[:20000], t[20000:50000], t[:20000]]
    got = emu_lib.zbest_parse(u3, n_slots=1, window=1 << 29, fresh=True)
    _cmp_parse(u3, got, level=4, window_size=1 << 29)


@pytest.mark.parametrize("level,w0", [(0, 0), (2, 0), (0, 1), (0, 8), (2, 1), (2, 64)])
def test_s2_lds_amd64_variant_equals_the_assembly_restatement(level, w0):
    """KC_S2_VARIANT_AMD64 in the LDS-table kernel (one wave-uniform step at a time, and speculative rounds): every size class of
    s2/encode_amd64.go against the oracle's restatement of the assembly encoders — which tests/test_ref_s2asm.py pins to the
    assembly itself."""
    rng = np.random.default_rng(11)
    blocks = []
    for kind in "JTM":
        d = corpora.corpus(kind, 4, 131072).tobytes()
        for n in (32, 100, 511, 512, 2000, 4095, 4096, 16383, 16384, 65535, 65536, 65537, 150000):
            st = int(rng.integers(0, len(d) - n))
            if w0 == 1 and not FULL and kind != "J" and n >= 65535:
                continue  # (the one-step path's emulation is the slowest: the large size classes on one corpus)
            blocks.append(d[st:st + n])
    blocks += [bytes(rng.integers(0, 4, int(rng.integers(32, 1500)), dtype=np.uint8)) for _ in range(60)]
    blocks += [u for u in corpora.edge_units() if 0 < len(u) < 70000]
    got = emu_lib.s2_encode_blocks(blocks, level=level, spec_w0=w0, variant=1)
    bad = [(i, len(b)) for i, b in enumerate(blocks) if got[i] != oracle_lib.s2_encode_asm(b, snappy=level == 2)]
    assert not bad, bad[:10]


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_xxh_fin_kernel_checksum_and_raw_payload_copy(mode):
    """kc_xxh64_fin_kernel (checksum behind the entropy stage; the payload of raw-only frames copied by the pass that hashes it) in its
    three store schedules: every unit's XXH64 equals the oracle's and lands in the frame's last four bytes; the payloads of the flagged
    frames arrive at their place — every alignment of the frame in the output, one and several blocks, ragged ends, lengths around
    every loop boundary — and not one byte outside them is touched; frames that are not flagged get no payload."""
    rng = np.random.default_rng(1234 + mode)
    lens = [0, 1, 31, 32, 63, 64, 65, 255, 256, 257, 300, 511, 512, 1023, 1024, 1025, 4096, 4097 + 13, 65536, 65536 + 255, 65536 + 256, 131072, 131071,
            3 * 65536, 3 * 65536 + 17, 200000]
    units = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in lens]
    flags = [(i % 5) != 3 for i in range(len(units))]
    pos, outp = 0, []
    for i, u in enumerate(units):
        pos += int(rng.integers(0, 40))  # (gaps: every residue of the frame start modulo 64 shows up)
        outp.append(pos)
        pos += len(u) + 9 + 3 * ((len(u) + 65535) // 65536) + 4
    bs = 65536
    dst, stage, soff, sizes, xxh = emu_lib.xxh_fin(units, flags, bs, outp, mode)
    for i, u in enumerate(units):
        want = int(oracle_lib.lib().kco_xxh64(u, len(u)))
        assert int(xxh[i]) == want, (i, len(u))
        if len(u) == 0:
            continue
        fs = int(sizes[i])
        ck = stage[int(soff[i]) + fs - 4:int(soff[i]) + fs].tobytes()
        assert ck == (want & 0xFFFFFFFF).to_bytes(4, "little"), (i, len(u))
        assert np.all(stage[int(soff[i]):int(soff[i]) + fs - 4] == 0x55)
        frame = dst[outp[i]:outp[i] + fs]
        expect = np.full(fs, 0xAA, dtype=np.uint8)
        if flags[i]:
            p = 9
            for b in range((len(u) + bs - 1) // bs):
                blk = u[b * bs:(b + 1) * bs]
                expect[p + 3:p + 3 + len(blk)] = np.frombuffer(blk, dtype=np.uint8)
                p += 3 + len(blk)
        bad = np.nonzero(frame != expect)[0]
        assert len(bad) == 0, (i, len(u), flags[i], bad[:8], outp[i] % 64)
    # nothing between or behind the frames was written
    mask = np.ones(len(dst), dtype=bool)
    for i in range(len(units)):
        mask[outp[i]:outp[i] + int(sizes[i])] = False
    assert np.all(dst[mask] == 0xAA)


def _pipeline_units(small=False):
    if small and not FULL:
        units = [corpora.corpus("T", 1, 131072, first_unit=5).tobytes(), corpora.corpus("M", 1, 131072, first_unit=2).tobytes()[:70001],
                 corpora.corpus("J", 1, 65536, first_unit=3).tobytes()[:30000], corpora.corpus("H", 1, 131072).tobytes()[:20000],
                 corpora.corpus("T", 2, 131072, first_unit=40).tobytes()[:140000]]
        units += [u[:70000] for u in corpora.edge_units() if len(u) < 140000]
        units += [u[:150000 if k == 0 else 50000] for k, u in enumerate(corpora.stress_units(seed=9, n=4))]
        return units
    units = [corpora.corpus("T", 1, 131072, first_unit=5).tobytes(), corpora.corpus("M", 1, 131072, first_unit=1).tobytes(),
             corpora.corpus("M", 1, 131072, first_unit=2).tobytes()[:70001], corpora.corpus("J", 1, 65536, first_unit=3).tobytes(),
             corpora.corpus("H", 1, 131072).tobytes()[:66000], corpora.corpus("T", 2, 131072, first_unit=40).tobytes()[:200000]]
    units += [u for u in corpora.edge_units() if len(u) < 140000]
    units += [u[:150000] for u in corpora.stress_units(seed=9, n=8)]
    return units


@pytest.mark.parametrize("level", [2, 3])
def test_whole_pipeline_speed_default_and_better(level):
    """SpeedDefault (kc_zdfast_match_grp_kernel) and SpeedBetterCompression (kc_zbetter_match_grp_kernel, 16 lanes per unit) end to
    end on the emulator, frames against the oracle's."""
    units = _pipeline_units(small=level == 3)
    ref = oracle_lib.ZstdOracle(level=level)
    frames, err, redo = emu_lib.zstd_frames(units, level=level, max_encoded_size=ref.max_encoded_size)
    assert err == 0
    want = [ref.encode_all(u) for u in units]
    if redo:  # a unit asked for the speculation re-run (the host's business): compare the others
        pytest.skip("a unit of the set asks for the re-run at this level")
    bad = [(i, len(u), len(f)) for i, (u, f) in enumerate(zip(units, frames)) if f != want[i]]
    assert not bad, bad[:8]


@pytest.mark.parametrize("w0,grow", [(1, 1), (1, 2), (3, 1), (4, 0), (8, 0)])
def test_speed_default_rounds_of_other_shapes(w0, grow, monkeypatch):
    """The SpeedDefault match finder (round 5: LDS source ring, one 16-byte load per candidate with fused lengths, the s+1 lookup's
    bytes from the winner's registers) under other speculation policies than the shipped one (2 then doubling): narrower and wider
    rounds put the winner on other lanes, move the ring's refills and change which candidates are verified speculatively — the frames
    must not change."""
    monkeypatch.setenv("KC_EMU_SPEC_W0", str(w0))
    monkeypatch.setenv("KC_EMU_SPEC_GROW", str(grow))
    units = [corpora.corpus("T", 1, 131072, first_unit=5).tobytes(), corpora.corpus("M", 1, 131072, first_unit=2).tobytes()[:70001],
             corpora.corpus("J", 1, 65536, first_unit=3).tobytes()[:40000], corpora.corpus("T", 2, 131072, first_unit=40).tobytes()[:150000]]
    units += [u[:60000] for u in corpora.edge_units() if len(u) < 140000][:24]
    units += [u[:100000] for u in corpora.stress_units(seed=11, n=4)]
    ref = oracle_lib.ZstdOracle(level=2)
    frames, err, redo = emu_lib.zstd_frames(units, level=2, max_encoded_size=ref.max_encoded_size)
    assert err == 0
    if redo:
        pytest.skip("a unit of the set asks for the re-run at this level")
    bad = [(i, len(u), len(f)) for i, (u, f) in enumerate(zip(units, frames)) if f != ref.encode_all(u)]
    assert not bad, bad[:8]


@pytest.mark.parametrize("finder", ["lds", "grp", "grp-tuned"])
def test_whole_pipeline_frames_equal_the_oracle(finder):
    """The device's whole SpeedFastest EncodeAll pipeline on the wave emulator — XXH64 kernel, match finder (LDS-table kernel, or the
    HBM-table group kernel in either of its forms), entropy stage with its Huffman / FSE table construction, literal and sequence
    bitstreams, block and frame assembly — frame for frame against the oracle: what the GPU suite checks on the device, here without
    one.  (Units whose late raw fallback asks for the speculation re-run are the host's business; none of these does.)"""
    units = _pipeline_units()
    ref = oracle_lib.ZstdOracle(level=1)
    frames, err, redo = emu_lib.zstd_frames(units, use_grp=finder != "lds", tuned=int(finder == "grp-tuned"), max_encoded_size=ref.max_encoded_size)
    assert err == 0 and redo == 0
    bad = [(i, len(u), len(f)) for i, (u, f) in enumerate(zip(units, frames)) if f != ref.encode_all(u)]
    assert not bad, "units whose emulated frame differs from the oracle's (index, length, frame length): %r" % bad[:8]


def test_whole_pipeline_options_and_streams():
    """The same with the options that change the frame: no checksum, single segment forced on and off, a small window (= small
    blocks), zero-length input with and without WithZeroFrames, WithNoEntropyCompression, WithAllLitEntropyCompression, and the
    Write ... Close stream form."""
    units = [corpora.corpus("T", 1, 131072, first_unit=6).tobytes()[:90000], corpora.corpus("M", 1, 131072, first_unit=4).tobytes()[:50000], b"",
             b"abc", corpora.corpus("J", 1, 65536, first_unit=1).tobytes()[:20000], corpora.corpus("H", 1, 8192).tobytes()]
    cases = [(dict(crc=False), dict(crc=False)), (dict(single=1), dict(single=True)), (dict(single=0), dict(single=False)),
             (dict(window=1 << 14, block_size=1 << 14), dict(window_size=1 << 14)), (dict(full_zero=False), dict(full_zero=False)),
             (dict(no_entropy=True), dict(no_entropy=True)), (dict(all_lit_entropy=True), dict(all_lit_entropy=True))]
    for ekw, okw in cases:
        ref = oracle_lib.ZstdOracle(level=1, **okw)
        frames, err, redo = emu_lib.zstd_frames(units, max_encoded_size=ref.max_encoded_size, **ekw)
        assert err == 0 and redo == 0
        bad = [(i, len(u)) for i, (u, f) in enumerate(zip(units, frames)) if f != ref.encode_all(u)]
        assert not bad, (okw, bad)
    ref = oracle_lib.ZstdOracle(level=1)
    frames, err, redo = emu_lib.zstd_frames(units, stream_mode=1, max_encoded_size=lambda n: ref.max_encoded_size(n) + 8)
    assert err == 0 and redo == 0
    bad = [(i, len(u)) for i, (u, f) in enumerate(zip(units, frames)) if f != ref.encode_stream(u)]
    assert not bad, ("stream", bad)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_whole_pipeline_with_the_checksum_behind_the_entropy_stage(mode):
    """batch_end's sequence on the emulator — entropy stage with the raw payloads deferred and the checksum field left open, size scan,
    kc_xxh64_fin_kernel (checksum of every frame, payload of the raw-only ones, in each store schedule), kc_compact_kernel — gives
    the oracle's frames back to back: compressible units, incompressible ones of one and several blocks, units whose blocks are raw
    and compressed in turn, tiny and empty ones."""
    h = corpora.corpus("H", 4, 131072, first_unit=3).tobytes()
    t = corpora.corpus("T", 2, 131072, first_unit=8).tobytes()
    units = [h[:131072], t[:100000], h[7:7 + 65536 + 13], h[:200000], b"", h[5:305], h[9:40], h[:65536] + t[:65536], t[:65536] + h[:65536] + t[:30000],
             h[3:4], t[:3000], h[1:1 + 131072 + 255]]
    ref = oracle_lib.ZstdOracle(level=1)
    frames, err, redo, nraw = emu_lib.zstd_frames(units, max_encoded_size=ref.max_encoded_size, fused=mode)
    assert err == 0 and redo == 0
    assert nraw >= 5  # the incompressible units took the fused copy
    bad = [(i, len(u), len(f)) for i, (u, f) in enumerate(zip(units, frames)) if f != ref.encode_all(u)]
    assert not bad, bad


@pytest.mark.parametrize("use_grp,tuned", [(1, 1), (1, 0), (0, 0)])
def test_whole_pipeline_with_the_no_match_prescan(use_grp, tuned):
    """kc_zfast_prescan_kernel in front of the match finder (KC_OPT_ZFAST_PRESCAN): units whose probe inserts never repeat a (bucket,
    4 bytes) pair are settled by the pre-scan — block records, frame header, raw block headers, payload descriptors — and skipped by
    the match finder and the entropy stage; everything else goes the regular way.  High-entropy units of one, two and three blocks,
    ragged and tiny ones, next to text, and high-entropy units with ONE planted repetition (at a probed position: must not be
    settled; at an unprobed one: may be): every frame equals the oracle's."""
    h = corpora.corpus("H", 6, 131072, first_unit=3).tobytes()
    t = corpora.corpus("T", 2, 131072, first_unit=8).tobytes()
    planted = bytearray(h[:131072])
    planted[70000:70008] = planted[0:8]          # position 0 is probed in block 0, 70000 - 65536 = 4464 ... (probed or not: the oracle decides)
    planted2 = bytearray(h[131072:262144])
    planted2[2:10] = planted2[0:8]                # probes at 0 and 2: the same 6 bytes -> a real candidate
    units = [h[:131072], t[:100000], h[7:7 + 65536 + 13], h[:196608], b"", h[5:305], h[9:40], h[:65536] + t[:65536], h[3:4], h[1:1 + 131072 + 255],
             bytes(planted), bytes(planted2), h[11:11 + 65536], h[13:13 + 9], h[17:17 + 10], h[19:19 + 65545]]
    ref = oracle_lib.ZstdOracle(level=1)
    frames, err, redo, nraw, ndone = emu_lib.zstd_frames(units, max_encoded_size=ref.max_encoded_size, fused=1, use_grp=use_grp, tuned=tuned | 0x100)
    assert err == 0 and redo == 0
    bad = [(i, len(u), len(f)) for i, (u, f) in enumerate(zip(units, frames)) if f != ref.encode_all(u)]
    assert not bad, bad
    assert ndone >= 9, ndone  # the plain high-entropy units (the three-block one has too many probes for the on-chip set: 1 332 keys)


def test_whole_pipeline_speed_best_compression():
    """SpeedBestCompression end to end on the emulator: kc_zbest_cost_kernel (the bit costs from the predefined FSE tables),
    kc_zbest_match_kernel on two persistent table slots, the entropy stage with allLitEntropy — the oracle's frames."""
    t = corpora.corpus("T", 1, 131072, first_unit=3).tobytes()
    m = corpora.corpus("M", 1, 131072, first_unit=2).tobytes()
    units = [t[:50000], m[:30000], b"", b"a", b"abcabcabcabc" * 50, corpora.corpus("J", 1, 65536, first_unit=5).tobytes()[:20000],
             corpora.corpus("H", 1, 8192).tobytes()[:5000], t[100000:131072] + t[:20000]]
    ref = oracle_lib.ZstdOracle(level=4)
    frames, err, redo = emu_lib.zstd_frames(units, level=4, max_encoded_size=ref.max_encoded_size)
    assert err == 0 and redo == 0
    bad = [(i, len(u), len(f)) for i, (u, f) in enumerate(zip(units, frames)) if f != ref.encode_all(u)]
    assert not bad, bad


_S2_REF = {0: "s2_encode", 1: "s2_encode_better", 2: "s2_encode_snappy", 3: "s2_encode_snappy_better"}


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_s2_hbm_kernel_blocks_bit_exact(level):
    """kc_s2_encode_kernel<level> — the HBM-table throughput kernel of BASELINE configuration C4, 8 blocks per wave — on the emulator:
    s2.Encode, s2.EncodeBetter, s2.EncodeSnappy, s2.EncodeSnappyBetter blocks against the oracle, at two speculation policies."""
    blocks = _s2_blocks()
    ref = getattr(oracle_lib, _S2_REF[level])
    _cmp(blocks, emu_lib.s2_encode_blocks_hbm(blocks, level=level), ref)
    _cmp(blocks[:10], emu_lib.s2_encode_blocks_hbm(blocks[:10], level=level, w0=8, w0b=8, grow=0), ref)


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_s2_hbm_kernel_amd64_variant(level):
    """The same kernel writing the bytes of the reference's amd64 assembly encoders (KC_S2_VARIANT_AMD64), against the oracle's
    restatement of them (pinned to the assembly itself by tests/test_ref_s2asm.py); every size class of encode_amd64.go."""
    blocks = [b for b in _s2_blocks() if len(b) > 0]
    t = corpora.corpus("T", 1, 65536, first_unit=4).tobytes()
    blocks += [t[:n] for n in (100, 511, 512, 4095, 4096, 16383, 16384, 65535)]
    got = emu_lib.s2_encode_blocks_hbm(blocks, level=level, variant=1)
    bad = [(i, len(b)) for i, b in enumerate(blocks) if got[i] != oracle_lib.s2_encode_asm(b, snappy=level in (2, 3), better=level in (1, 3))]
    assert not bad, bad[:8]


def test_s2_hbm_kernel_framed_chunks():
    """s2.Writer chunks (type, length, masked CRC32C, body) from the HBM-table kernel."""
    blocks = [b for b in _s2_blocks() if len(b) > 0]
    buf, off = corpora.pack_units(blocks)
    ref, ro = oracle_lib.s2_encode_stream(buf, off, with_stream_id=False)
    got = emu_lib.s2_encode_blocks_hbm(blocks, level=0, framed=True)
    for i in range(len(blocks)):
        assert ref[int(ro[i]):int(ro[i + 1])].tobytes() == got[i], "chunk %d (len %d)" % (i, len(blocks[i]))
