"""WithConcurrentBlocks job mode (zstd/enc_jobs.go): the oracle's restatement, and the device path against it.

One stream is cut into jobs of max(4 * window, 512 KiB) input bytes; each job is encoded on a freshly reset encoder whose
history is the tail of the previous job (ResetPrefix); the outputs are concatenated behind one frame header.  The jobs are
independent units, which is how ONE stream becomes device work (kc_zstd_encode_jobs)."""
import numpy as np
import pytest

import corpora


def _cases():
    t = corpora.corpus("T", 24, 131072, first_unit=300).tobytes()  # 3 MiB
    m = corpora.corpus("M", 24, 131072, first_unit=40).tobytes()
    return t, m


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_oracle_jobs_frames_decode_and_have_the_documented_shape(oracle, level):
    """CPU: the oracle's job-mode frames round-trip through the independent libzstd decoder for every job count, Flush pattern
    and window; the special cases of dispatchJob hold (a one-block stream is the EncodeAll frame, an empty stream the 9-byte...
    streaming frame, a stream of exactly k jobs ends in an empty raw last block)."""
    t, m = _cases()
    for win in (1 << 17, 1 << 18):  # jobs of 512 KiB / 1 MiB
        e = oracle.ZstdOracle(level=level, window_size=win)
        js = max(4 * win, 512 << 10)
        for data in (t, m, t[:js], t[:2 * js], t[:js + 1], t[:js - 1], t[:70000], m[:1 << 20]):
            for cuts in ((), (1000, 300000), (len(data),), (js // 2, js // 2 + 10, js + 77)):
                fr = e.encode_jobs(data, cuts)
                assert oracle.zstd_decompress(fr, len(data) + 16) == data, (level, win, len(data), cuts)
        # a stream of at most one block with no Flush before Close is the EncodeAll frame (enc_jobs.go:263-279)
        bs = 65536 if level == 1 else 131072
        assert e.encode_jobs(t[:bs]) == e.encode_all(t[:bs])
        assert e.encode_jobs(t[:100]) == e.encode_all(t[:100])
        # ... but not when a Flush dispatched a job first: then a header was written and the blocks follow
        assert e.encode_jobs(t[:100], (50,)) != e.encode_all(t[:100])
        # exactly two jobs' worth of input: both are dispatched by Write, Close adds the empty raw last block + checksum
        fr = e.encode_jobs(t[:2 * js])
        assert fr[-7:-4] == b"\x01\x00\x00", fr[-8:].hex()
        # no window descriptor games: same header as the plain stream's
        assert fr[:6] == e.encode_stream(t[:2 * js])[:6]
        assert e.encode_jobs(b"") == e.encode_stream(b"")


def test_jobs_differ_from_the_plain_stream_only_in_history(oracle):
    """Same blocks, different history: below one job the two modes give the same bytes, above they differ (the second job starts
    from the overlap prefix alone)."""
    t, _ = _cases()
    e = oracle.ZstdOracle(level=1, window_size=1 << 17)
    js = 512 << 10
    assert e.encode_jobs(t[:js - 5]) == e.encode_stream(t[:js - 5])
    assert e.encode_jobs(t[:3 * js]) != e.encode_stream(t[:3 * js])


# level 1 runs on both SpeedFastest kernel families: "1L" the LDS-table kernel (what the dispatcher picks for a few hundred jobs:
# a job is one wave's unit, its table primed from the overlap prefix), "1H" the HBM-table kernel
GPU_LEVELS = ["1L", "1H", 2, 3]


def _li(level):
    return 1 if level in ("1L", "1H") else int(level)


def _enc(level, **kw):
    from compress_amd import zstd
    opts = [zstd.WithEncoderLevel(_li(level)), zstd.WithConcurrentBlocks(True), zstd.WithEncoderConcurrency(4)]
    if level in ("1L", "1H"):
        opts.append(zstd.WithMatchPath("lds" if level == "1L" else "hbm"))
    if "window" in kw:
        opts.append(zstd.WithWindowSize(kw["window"]))
    return zstd.NewWriter(None, *opts)


@pytest.mark.gpu
@pytest.mark.parametrize("level", GPU_LEVELS + [4])
def test_device_jobs_bit_exact_small(oracle, kclib, level):
    """GPU: kc_zstd_encode_jobs == the oracle's job mode for every job count / Flush pattern of the CPU test (small windows: many
    jobs, short prefixes, prefixes shorter than the overlap, empty final jobs)."""
    import torch
    assert torch.cuda.is_available()
    t, m = _cases()
    for win in (1 << 17, 1 << 18):
        e = oracle.ZstdOracle(level=_li(level), window_size=win)
        enc = _enc(level, window=win)
        js = enc.JobSize()
        assert js == max(4 * win, 512 << 10)
        for data in (t, m, t[:js], t[:2 * js], t[:js + 1], t[:js - 1], t[:70000], t[:100], b""):
            for cuts in ((), (1000, 300000), (len(data),), (js // 2, js // 2 + 10, js + 77)):
                got = enc.EncodeJobs(data, cuts)
                ref = e.encode_jobs(data, cuts)
                assert got == ref, "level %s window %d len %d cuts %r: %d bytes vs oracle %d" % (level, win, len(data), cuts, len(got), len(ref))
        enc.Close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", GPU_LEVELS)
def test_device_jobs_bit_exact_256mib_stream(oracle, kclib, level):
    """GPU: a 256 MiB stream at every level, 1 MiB window -> 64 jobs of 4 MiB with 128 / 256 KiB of overlap; and through the
    Write / Flush / Close face of the Encoder."""
    import io
    import torch
    assert torch.cuda.is_available()
    data = corpora.corpus("T" if level != 3 else "M", 2048, 131072, first_unit=7000).tobytes()
    e = oracle.ZstdOracle(level=_li(level), window_size=1 << 20)
    ref = e.encode_jobs(data)
    enc = _enc(level, window=1 << 20)
    got = enc.EncodeJobs(data)
    assert len(got) == len(ref) and got == ref
    if level in ("1L", "1H"):
        assert enc.ctx().last_path() == ("lds" if level == "1L" else "hbm")
    assert oracle.zstd_decompress(got[:], len(data) + 16) == data
    enc.Close()
    # the Writer face: Write, Flush, ReadFrom, Close
    sink = io.BytesIO()
    from compress_amd import zstd
    w = zstd.NewWriter(sink, zstd.WithEncoderLevel(_li(level)), zstd.WithWindowSize(1 << 20), zstd.WithConcurrentBlocks(True), zstd.WithEncoderConcurrency(2))
    w.Write(data[:5000000])
    w.Flush()
    w.Write(data[5000000:9000000])
    w.ReadFrom(io.BytesIO(data[9000000:20000000]))
    w.Close()
    assert sink.getvalue() == e.encode_jobs(data[:20000000], (5000000, 9000000))


@pytest.mark.gpu
def test_device_jobs_best_level_stream(oracle, kclib):
    """SpeedBestCompression jobs: overlap = window / 2 (encoder_options.go:364), every position of a job's prefix indexed by the
    job's wave itself (ResetPrefix, enc_best.go:554-568): a 20 MiB stream at a 1 MiB window = five jobs of 4 MiB with 512 KiB of
    overlap, plus Flush cuts."""
    import torch
    assert torch.cuda.is_available()
    data = corpora.corpus("T", 160, 131072, first_unit=9000).tobytes()
    e = oracle.ZstdOracle(level=4, window_size=1 << 20)
    enc = _enc(4, window=1 << 20)
    assert enc.JobSize() == 4 << 20 and enc.OverlapSize() == 512 << 10
    for cuts in ((), (3000000, 9000000)):
        got = enc.EncodeJobs(data, cuts)
        assert got == e.encode_jobs(data, cuts), cuts
    assert oracle.zstd_decompress(got, len(data) + 16) == data
    enc.Close()


@pytest.mark.gpu
def test_device_jobs_default_window_speedfastest(oracle, kclib):
    """GPU: SpeedFastest with its own 4 MiB window: jobs of 16 MiB, 512 KiB of overlap, 64 KiB blocks — 256 MiB = 16 jobs + the
    empty final job."""
    import torch
    assert torch.cuda.is_available()
    data = corpora.corpus("T", 2048, 131072, first_unit=11000).tobytes()
    enc = _enc(1)
    assert enc.JobSize() == 16 << 20 and enc.OverlapSize() == 512 << 10
    got = enc.EncodeJobs(data)
    ref = oracle.ZstdOracle(level=1).encode_jobs(data)
    assert got == ref
    assert enc.ctx().last_path() == "lds"  # 17 jobs: the dispatcher's choice
    enc.Close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_device_jobs_in_scratch_budgeted_batches(oracle, kclib, level):
    """GPU: the jobs of one stream go to the device in batches bounded by the scratch budget (KC_OPT_MAX_SCRATCH_MIB), like the
    units of kc_zstd_encode_units_dev: same frame whether the 41 jobs of a 20 MiB stream at a 128 KiB window run as one batch or
    as many, with Flush cuts landing inside and on batch boundaries."""
    import torch
    from compress_amd import _lib
    assert torch.cuda.is_available()
    data = corpora.corpus("T" if level != 3 else "M", 160, 131072, first_unit=13000).tobytes()
    e = oracle.ZstdOracle(level=level, window_size=1 << 17)
    enc = _enc(level, window=1 << 17)
    for cuts in ((), (1 << 20, 3000001, 3000002, 19 << 20)):
        ref = e.encode_jobs(data, cuts)
        enc.ctx().set_option(_lib.OPT_MAX_SCRATCH_MIB, 160 << 10)
        assert enc.EncodeJobs(data, cuts) == ref
        assert enc.ctx().get_option(_lib.OPT_LAST_BATCHES) == 1
        enc.ctx().set_option(_lib.OPT_MAX_SCRATCH_MIB, 24)
        got = enc.EncodeJobs(data, cuts)
        assert enc.ctx().get_option(_lib.OPT_LAST_BATCHES) > 3
        assert got == ref, "level %d cuts %r: %d bytes vs oracle %d" % (level, cuts, len(got), len(ref))
    enc.Close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", GPU_LEVELS)
def test_device_jobs_tables_primed_on_the_device_or_on_the_host(oracle, kclib, level):
    """GPU: a job's tables come from kc_zstd_prime_kernel (default) or from the host's restatement of ResetPrefix uploaded per batch
    (KC_OPT_JOB_PRIME 0): the same frame, the oracle's — prefixes of 16 / 32 / 64 KiB (window 128 / 256 / 512 KiB at the level's
    overlap), a job shorter than the overlap (the prefix is the whole previous job), repetitive text whose buckets collide within a
    round of 64 inserts, and the speculation re-run's re-priming (KC_OPT_TEST_FEED_REDO does not apply to jobs: the re-run is the
    regular one of units whose first block had to be redone)."""
    import torch
    from compress_amd import _lib
    assert torch.cuda.is_available()
    t = corpora.corpus("T", 24, 131072, first_unit=17000).tobytes()
    rep = (b"abcdefgh" * 40 + b"0123456789abcdef" * 20) * 2000
    for win in (1 << 17, 1 << 19):
        e = oracle.ZstdOracle(level=_li(level), window_size=win)
        enc = _enc(level, window=win)
        for data, cuts in ((t, ()), (rep + t[:300000] + rep, ()), (t[:1500000], (600000, 600100, 600101)), (rep[:700000], (1,))):
            ref = e.encode_jobs(data, cuts)
            for prime in (1, 0):
                enc.ctx().set_option(_lib.OPT_JOB_PRIME, prime)
                assert enc.ctx().get_option(_lib.OPT_JOB_PRIME) == prime
                got = enc.EncodeJobs(data, cuts)
                assert got == ref, "level %s window %d len %d cuts %r prime %d: %d bytes vs oracle %d" % (level, win, len(data), cuts, prime, len(got), len(ref))
        enc.Close()
