"""Host mirror of the reference's zstd encoder API for the EncodeAll hot path.

Names, argument meaning and error behaviour follow zstd/encoder.go and
zstd/encoder_options.go; the bytes come from the HIP engine behind include/kcgpu.h.

    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(zstd.SpeedFastest))
    frame = enc.EncodeAll(src, b"")                     # == reference EncodeAll(src, nil)
    frames, off = enc.EncodeUnits(buf, unit_off)        # N independent EncodeAll calls, one launch
"""
import contextlib
import ctypes as C
import os
import threading

from . import _lib
from ._lib import KcError

# zstd.EncoderLevel (encoder_options.go:163-179)
SpeedFastest, SpeedDefault, SpeedBetterCompression, SpeedBestCompression = 1, 2, 3, 4
MinWindowSize, MaxWindowSize = 1 << 10, 1 << 29


def EncoderLevelFromString(s):
    """encoder_options.go:181-198: case-insensitive name -> (ok, level)."""
    m = {"fastest": SpeedFastest, "default": SpeedDefault, "better": SpeedBetterCompression, "best": SpeedBestCompression}
    lvl = m.get(s.lower())
    return (lvl is not None), (lvl if lvl is not None else SpeedDefault)


def EncoderLevelFromZstd(level):
    """encoder_options.go:200-218."""
    if level < 3:
        return SpeedFastest
    if level < 6:
        return SpeedDefault
    if level < 10:
        return SpeedBetterCompression
    return SpeedBestCompression


# ---- EOption constructors: each returns a callable applied in order to the options struct ----
def _opt(name, *args):
    def apply(o):
        L = _lib.load()
        st = getattr(L, "kc_zstd_opts_" + name)(C.byref(o), *args)
        if st != 0:
            raise ValueError("zstd option %s%r rejected" % (name, args))
    return apply


def WithEncoderLevel(l):
    return _opt("level", int(l))


def WithWindowSize(n):
    return _opt("window", int(n))


def WithEncoderCRC(b):
    return _opt("crc", int(bool(b)))


def WithZeroFrames(b):
    return _opt("zero_frames", int(bool(b)))


def WithNoEntropyCompression(b):
    return _opt("no_entropy", int(bool(b)))


def WithAllLitEntropyCompression(b):
    return _opt("all_lit_entropy", int(bool(b)))


def WithSingleSegment(b):
    return _opt("single_segment", int(bool(b)))


def WithEncoderDictRaw(id, content):
    content = bytes(content)

    def apply(o):
        L = _lib.load()
        buf = C.create_string_buffer(content, len(content))
        o._dict_keep = buf
        if L.kc_zstd_opts_dict_raw(C.byref(o), id, C.cast(buf, C.c_void_p), len(content)) != 0:
            raise ValueError("dictionary rejected")
    return apply


def WithEncoderDict(dict):
    """zstd.WithEncoderDict (zstd/encoder_options.go:382-391): a dictionary in the `zstd --train` format."""
    blob = bytes(dict)

    def apply(o):
        L = _lib.load()
        buf = C.create_string_buffer(blob, len(blob))
        o._dict_keep = buf  # o.dict points into the blob
        if L.kc_zstd_opts_dict(C.byref(o), C.cast(buf, C.c_void_p), len(blob)) != 0:
            raise ValueError("dictionary rejected (loadDict error)")
    return apply


SKIPPABLE_FRAME_HEADER = 8  # zstd/frameenc.go: skippableFrameHeader = 4 + 4


def calc_skippable_frame(written, want_multiple):
    """calcSkippableFrame (zstd/frameenc.go:100-116): bytes of skippable frame that bring `written` to a multiple of
    `want_multiple`; a frame needs its 8-byte header, so a remainder below that is padded by one more multiple."""
    if want_multiple <= 0:
        raise ValueError("wantMultiple <= 0")
    if written < 0:
        raise ValueError("written < 0")
    left_over = written % want_multiple
    if left_over == 0:
        return 0
    to_add = want_multiple - left_over
    while to_add < SKIPPABLE_FRAME_HEADER:
        to_add += want_multiple
    return to_add


def skippable_frame(total, rand=os.urandom):
    """skippableFrame (zstd/frameenc.go:118-137): a skippable frame of `total` bytes in all — magic 0x184D2A50, the 32-bit size of what
    follows, and that many bytes from `rand` (the reference reads crypto/rand.Reader: the padding is not reproducible by design)."""
    if total == 0:
        return b""
    if total < SKIPPABLE_FRAME_HEADER:
        raise ValueError("requested skippable frame (%d) < 8" % total)
    if total > 0xFFFFFFFF:
        raise ValueError("requested skippable frame (%d) > max uint32" % total)
    f = total - SKIPPABLE_FRAME_HEADER
    return bytes([0x50, 0x2A, 0x4D, 0x18]) + int(f).to_bytes(4, "little") + bytes(rand(f))


def pad_frames(out, out_off, pad, rand=os.urandom):
    """Every frame of a batch followed by the skippable frame that WithEncoderPadding(pad) appends to it — per frame what EncodeAll
    does with an empty dst (zstd/encoder.go:829-837).  A zero-length frame (empty input without WithZeroFrames) stays empty:
    EncodeAll returns before the padding there (zstd/encoder.go:732-752).  Returns (bytes, offsets)."""
    import numpy as np
    n = len(out_off) - 1
    parts, offs, pos = [], [0], 0
    for i in range(n):
        fr = bytes(out[int(out_off[i]):int(out_off[i + 1])])
        if fr:
            fr += skippable_frame(calc_skippable_frame(len(fr), pad), rand)
        parts.append(fr)
        pos += len(fr)
        offs.append(pos)
    return np.frombuffer(b"".join(parts), dtype=np.uint8), np.asarray(offs, dtype=np.uint64)


def WithEncoderPadding(n):
    """zstd.WithEncoderPadding (encoder_options.go:142-158): the output of EncodeAll / of a stream is padded to a multiple of n with a
    skippable frame of random bytes.  Host-side: the device frames are the unpadded ones."""
    n = int(n)
    if n <= 0:
        raise ValueError("padding must be at least 1")
    if n > 1 << 30:
        raise ValueError("padding must less than 1GB (1<<30 bytes) ")

    def apply(o):
        pass
    apply._kc_pad = 0 if n == 1 else n  # "No need to waste our time."
    return apply


def WithEncoderDictDelete():
    """zstd.WithEncoderDictDelete (encoder_options.go:408-415): no dictionary from here on."""
    def apply(o):  # every field a dictionary sets, back to kc_zstd_opts_default's
        o.dict = None
        o.dict_len = 0
        o.dict_id = 0
        o.dict_offsets[0], o.dict_offsets[1], o.dict_offsets[2] = 1, 4, 8  # blockenc.go:78
        o.dict_huf_len = 0
        o.dict_huf_log = 0
    apply._kc_dict_delete = True
    return apply


def WithMatchPath(path):
    """Not a reference option: which kernel family serves SpeedFastest ('auto' by units in flight, 'hbm', 'lds';
    KC_OPT_MATCH_PATH in include/kcgpu.h).  The bytes are the reference's either way."""
    def apply(o):
        pass
    apply._kc_path = path
    return apply


def WithEncoderConcurrency(n):
    """zstd.WithEncoderConcurrency (zstd/encoder_options.go:76-87).  The device path is batch-parallel; the bytes that depend on
    it are those of WithConcurrentBlocks, which the reference switches off when the concurrency is 1 (zstd/encoder.go:81), and
    those of dictionary streams: with 1 the reference's synchronous nextBlock form drops the dictionary's literal table before
    the first block (zstd/encoder.go:371)."""
    if n <= 0:
        raise ValueError("concurrency must be at least 1")

    def apply(o):
        o.concurrent = int(n)
    apply._kc_concurrency = int(n)
    return apply


def WithConcurrentBlocks(b):
    """zstd.WithConcurrentBlocks (zstd/encoder_options.go:340-353): the stream written through Write / ReadFrom / Flush / Close is
    cut into jobs of max(4 * window, 512 KiB) bytes, each encoded with the tail of the previous job as its history
    (zstd/enc_jobs.go) — independent units for the device (kc_zstd_encode_jobs).  As in the reference it has no effect with a
    dictionary or with WithEncoderConcurrency(1)."""
    def apply(o):
        pass
    apply._kc_conc_blocks = bool(b)
    return apply


def WithLowerEncoderMem(b):
    return lambda o: setattr(o, "low_mem", int(bool(b)))


class Encoder:
    """zstd.Encoder: EncodeAll and its batched forms, plus the streaming surface Write / ReadFrom / Flush / Close / Reset
    (zstd/encoder.go:140-649).  A stream is buffered on the host and encoded on Close as ONE device unit whose bytes equal the
    reference's for the same Write / Flush sequence, dictionaries included; streams of more than 1 GiB are not served
    (the caller falls back to the reference).  EncodeStreams / EncodeStreamsDevice batch many streams per launch."""

    def __init__(self, *opts, device=0, stream=None, w=None, path=None):
        """path: None / 'auto' (by units in flight), 'hbm' or 'lds' — the kernel family of the SpeedFastest match finder
        (KC_OPT_MATCH_PATH, include/kcgpu.h); both give the reference's bytes."""
        self._path = path
        self._w = w
        self._buf = bytearray()
        self._cuts = []
        self._closed = False
        L = _lib.load()
        self.o = _lib.ZstdOpts()
        L.kc_zstd_opts_default(C.byref(self.o))
        self._conc_blocks, self._concurrency = False, os.cpu_count() or 1  # o.concurrent defaults to GOMAXPROCS (encoder_options.go:38)
        self._pad = 0
        for op in opts:
            if hasattr(op, "_kc_pad"):
                self._pad = op._kc_pad
                continue
            if hasattr(op, "_kc_path"):
                self._path = op._kc_path
                continue
            if hasattr(op, "_kc_conc_blocks"):
                self._conc_blocks = op._kc_conc_blocks
                continue
            if hasattr(op, "_kc_concurrency"):
                self._concurrency = op._kc_concurrency
                op(self.o)
                continue
            op(self.o)
        self._device, self._stream = device, stream
        self._ctx = None
        self._lock = threading.Lock()
        self._primary_busy = False
        self._spare = []  # contexts of concurrent EncodeAll / EncodeUnits callers (see _held)

    # -- reference API --
    def MaxEncodedSize(self, size):
        m = int(_lib.load().kc_zstd_max_encoded_size(C.byref(self.o), int(size)))
        if self._pad > 1:  # zstd/encoder.go:867-871
            m += calc_skippable_frame(m, self._pad)
        return m

    def EncodeAll(self, src, dst=b""):
        """Encode all of src as one frame and append to dst (zstd/encoder.go:722)."""
        import numpy as np
        src = bytes(src)
        off = np.array([0, len(src)], dtype=np.uint64)
        out, _ = self._encode_units_unpadded(np.frombuffer(src, dtype=np.uint8), off)
        res = bytes(dst) + out.tobytes()
        if self._pad > 0 and len(out):  # the TOTAL size becomes a multiple (dst included: encoder_options.go:140, encoder.go:829-837)
            res += skippable_frame(calc_skippable_frame(len(res), self._pad))
        return res

    # -- batched form: what the cgo shim calls --
    def _new_ctx(self):
        c = _lib.Context(self._device, self._stream)
        if self._path is not None:
            c.set_path(self._path)
        return c

    def ctx(self):
        """The encoder's own context (options set on it, timings, the streaming / device-resident / asynchronous calls)."""
        if self._ctx is None:
            self._ctx = self._new_ctx()
        return self._ctx

    @contextlib.contextmanager
    def _held(self):
        """EncodeAll "can be called concurrently" on one Encoder (zstd/encoder.go:717): every call takes an encoder state from
        e.encoders and puts it back (encoder.go:90-99, 722-729).  Here the state is a kc_ctx: a lone caller always gets the
        encoder's own context; a caller that finds it taken gets a spare one (created on demand, at most WithEncoderConcurrency
        kept), with the kernel-family choice of the constructor — options set on ctx() afterwards stay with ctx()."""
        with self._lock:
            if not self._primary_busy:
                self._primary_busy = True
                c = self.ctx()
            else:
                c = self._spare.pop() if self._spare else None
        if c is None:
            c = self._new_ctx()
        try:
            yield c
        finally:
            with self._lock:
                if c is self._ctx:
                    self._primary_busy = False
                elif self._ctx is not None and len(self._spare) < max(1, self._concurrency):
                    self._spare.append(c)
                else:
                    c.close()

    def EncodeUnits(self, src, unit_off):
        """src: numpy uint8 (host); unit_off: uint64[n+1].  Returns (numpy uint8 frames, uint64[n+1] offsets).  With
        WithEncoderPadding every frame is followed by its skippable frame (N x EncodeAll(unit, nil))."""
        out, out_off = self._encode_units_unpadded(src, unit_off)
        if self._pad > 0:
            return pad_frames(out, out_off, self._pad)
        return out, out_off

    def _encode_units_unpadded(self, src, unit_off):
        import numpy as np
        src = np.ascontiguousarray(src, dtype=np.uint8)
        unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
        n = len(unit_off) - 1
        cap = sum(((self.MaxEncodedSize(int(unit_off[i + 1] - unit_off[i])) + 15) & ~15) for i in range(n)) + 64
        dst = np.empty(cap, dtype=np.uint8)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        with self._held() as ctx:
            ctx.check(ctx.L.kc_zstd_encode_units(ctx.h, C.byref(self.o), src.ctypes.data, unit_off.ctypes.data, n,
                                                 dst.ctypes.data, cap, out_off.ctypes.data))
        return dst[:int(out_off[n])], out_off

    def EncodeUnitsSubmit(self, src, unit_off, dst=None):
        """Asynchronous EncodeUnits (kc_zstd_encode_units_submit): returns at once; Wait() returns what EncodeUnits returns.
        dst: optional pre-allocated (pre-faulted) uint8 array of at least the summed MaxEncodedSize."""
        import numpy as np
        ctx = self.ctx()
        src = np.ascontiguousarray(src, dtype=np.uint8)
        unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
        n = len(unit_off) - 1
        cap = sum(((self.MaxEncodedSize(int(unit_off[i + 1] - unit_off[i])) + 15) & ~15) for i in range(n)) + 64
        if dst is None:
            dst = np.empty(cap, dtype=np.uint8)
        assert dst.dtype == np.uint8 and dst.size >= cap
        out_off = np.zeros(n + 1, dtype=np.uint64)
        ctx.check(ctx.L.kc_zstd_encode_units_submit(ctx.h, C.byref(self.o), src.ctypes.data, unit_off.ctypes.data, n,
                                                    dst.ctypes.data, dst.size, out_off.ctypes.data))
        self._job = (src, unit_off, dst, out_off, n)  # kept alive until Wait

    def Wait(self):
        src, unit_off, dst, out_off, n = self._job
        self._job = None
        ctx = self.ctx()
        ctx.check(ctx.L.kc_wait(ctx.h))
        return dst[:int(out_off[n])], out_off

    def EncodeUnitsDevice(self, d_src_ptr, unit_off, d_dst_ptr, dst_cap):
        """Device-resident form (pointers are ints).  Returns uint64[n+1] offsets (host numpy)."""
        import numpy as np
        ctx = self.ctx()
        unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
        n = len(unit_off) - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        ctx.check(ctx.L.kc_zstd_encode_units_dev(ctx.h, C.byref(self.o), d_src_ptr, unit_off.ctypes.data, n, d_dst_ptr,
                                                 int(dst_cap), out_off.ctypes.data))
        return out_off

    # -- streaming surface (encoder.go:140-253, 531-649) --
    def Reset(self, w):
        self._w = w
        self._buf = bytearray()
        self._cuts = []
        self._closed = False

    def Write(self, p):
        if self._closed:
            raise IOError("zstd: encoder closed")  # ErrEncoderClosed
        self._buf += bytes(p)
        return len(p)

    def ReadFrom(self, r):
        self._cuts.append(len(self._buf))  # ReadFrom first ends the block being filled (encoder.go:482-486)
        n = 0
        while True:
            b = r.read(1 << 20)
            if not b:
                return n
            n += self.Write(b)

    def Flush(self):
        """Encoder.Flush (encoder.go:547): ends the block being filled.  The bytes reach w on Close (the device path encodes the
        whole stream in one call); they are the bytes the reference's writer receives, block boundaries included."""
        if self._closed:
            return
        self._cuts.append(len(self._buf))

    def _finish_stream(self):
        """The frame the reference writes for Write(everything) + Close() goes to w."""
        if self._closed or self._w is None:
            self._closed = True
            return
        import numpy as np
        data = bytes(self._buf)
        cuts, self._cuts = self._cuts, []
        self._buf = bytearray()
        self._closed = True
        if self._conc_blocks and self._concurrency > 1 and not self.o.dict_len:  # zstd/encoder.go:81: else the option is off
            frame = bytes(self.EncodeJobs(data, cuts))
        else:
            out, _ = self.EncodeStreams(np.frombuffer(data, dtype=np.uint8), np.array([0, len(data)], dtype=np.uint64),
                                        flush_at=[cuts] if cuts else None)
            frame = out.tobytes()
        if self._pad > 0:  # Close: padding from the bytes written in this stream (zstd/encoder.go:637-645, 700-708)
            frame += skippable_frame(calc_skippable_frame(len(frame), self._pad))
        self._w.write(frame)

    def JobSize(self):
        return int(_lib.load().kc_zstd_job_size(C.byref(self.o)))

    def OverlapSize(self):
        return int(_lib.load().kc_zstd_overlap_size(C.byref(self.o)))

    def EncodeJobs(self, src, flush_at=()):
        """ONE stream with WithConcurrentBlocks(true): NewWriter(w, opts, WithConcurrentBlocks(true)); Write(src) with Flush at
        flush_at; Close() (kc_zstd_encode_jobs: the jobs of zstd/enc_jobs.go are the device's units).  Returns bytes."""
        import numpy as np
        ctx = self.ctx()
        src = np.frombuffer(bytes(src), dtype=np.uint8) if not isinstance(src, np.ndarray) else np.ascontiguousarray(src, dtype=np.uint8)
        n = len(src)
        cuts = np.ascontiguousarray(sorted(int(x) for x in flush_at), dtype=np.uint64)
        js, ov = self.JobSize(), self.OverlapSize()
        njobs = n // js + len(cuts) + 2
        cap = njobs * (((self.MaxEncodedSize(js + ov) - js - ov) + 31) & ~15) + n + njobs * ov + 4096
        dst = np.empty(cap, dtype=np.uint8)
        out_len = C.c_uint64(0)
        sp = src.ctypes.data if n else None
        ctx.check(ctx.L.kc_zstd_encode_jobs(ctx.h, C.byref(self.o), sp, n, cuts.ctypes.data if len(cuts) else None, len(cuts),
                                            dst.ctypes.data, cap, C.byref(out_len)))
        return dst[:out_len.value].tobytes()

    def EncodeStreams(self, src, unit_off, flush_at=None):
        """Like EncodeUnits, but every unit is a stream: NewWriter(w).Write(unit) ... Close().  flush_at: per stream, the byte
        counts at which Flush was called (kc_zstd_encode_streams_cuts)."""
        import numpy as np
        ctx = self.ctx()
        src = np.ascontiguousarray(src, dtype=np.uint8)
        unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
        n = len(unit_off) - 1
        ncut = [len(flush_at[i]) if flush_at is not None else 0 for i in range(n)]
        cap = sum(((self.MaxEncodedSize(int(unit_off[i + 1] - unit_off[i])) + (3 * ncut[i] + 3 if flush_at is not None else 0) + 15) & ~15)
                  for i in range(n)) + 64
        dst = np.empty(cap, dtype=np.uint8)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        if len(src) == 0:
            src = np.zeros(1, dtype=np.uint8)
        if flush_at is None:
            ctx.check(ctx.L.kc_zstd_encode_streams(ctx.h, C.byref(self.o), src.ctypes.data, unit_off.ctypes.data, n,
                                                   dst.ctypes.data, cap, out_off.ctypes.data))
        else:
            cut_off = np.zeros(n + 1, dtype=np.uint64)
            cut_off[1:] = np.cumsum(ncut)
            cuts = np.array([int(x) for f in flush_at for x in sorted(f)] + [0], dtype=np.uint64)
            ctx.check(ctx.L.kc_zstd_encode_streams_cuts(ctx.h, C.byref(self.o), src.ctypes.data, unit_off.ctypes.data, n,
                                                        cut_off.ctypes.data, cuts.ctypes.data, dst.ctypes.data, cap, out_off.ctypes.data))
        return dst[:int(out_off[n])], out_off

    def EncodeStreamsDevice(self, d_src_ptr, unit_off, d_dst_ptr, dst_cap):
        import numpy as np
        ctx = self.ctx()
        unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
        n = len(unit_off) - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        ctx.check(ctx.L.kc_zstd_encode_streams_dev(ctx.h, C.byref(self.o), d_src_ptr, unit_off.ctypes.data, n, d_dst_ptr,
                                                   int(dst_cap), out_off.ctypes.data))
        return out_off

    def DecodeUnitsDevice(self, d_enc_ptr, enc_off, d_dst_ptr, dst_off, dict_content=None):
        """Verifier: decode one frame per unit on the device; returns uint32[n] status (0 = content and checksum match).
        dict_content: the (raw) dictionary content the frames were written with, if any."""
        import numpy as np
        ctx = self.ctx()
        enc_off = np.ascontiguousarray(enc_off, dtype=np.uint64)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
        n = len(enc_off) - 1
        status = np.zeros(max(n, 1), dtype=np.uint32)
        dc = bytes(dict_content) if dict_content else None
        ctx.check(ctx.L.kc_zstd_decode_units_dict_dev(ctx.h, d_enc_ptr, enc_off.ctypes.data, n, d_dst_ptr, dst_off.ctypes.data, status.ctypes.data,
                                                      dc, len(dc) if dc else 0))
        return status[:n]

    def EncodeUnitsDeviceBegin(self, d_src_ptr, unit_off, d_dst_ptr, dst_cap):
        """First half of EncodeUnitsDevice (one device batch): enqueue up to and including the match finder, no wait."""
        import numpy as np
        ctx = self.ctx()
        unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
        self._pending_n = len(unit_off) - 1
        ctx.check(ctx.L.kc_zstd_encode_units_dev_begin(ctx.h, C.byref(self.o), d_src_ptr, unit_off.ctypes.data, self._pending_n,
                                                       d_dst_ptr, int(dst_cap)))

    def EncodeUnitsDeviceEnd(self, d_dst_ptr=None, dst_cap=0):
        """Second half: entropy stage, wait, offsets (uint64[n+1], host numpy).  d_dst_ptr: the frames go THERE instead of to the
        place given to Begin (kc_zstd_encode_units_dev_end_at: the parts of one batch run as several launches, each part's frames
        right behind the previous part's); the offsets are relative to it."""
        import numpy as np
        ctx = self.ctx()
        out_off = np.zeros(self._pending_n + 1, dtype=np.uint64)
        if d_dst_ptr is None:
            ctx.check(ctx.L.kc_zstd_encode_units_dev_end(ctx.h, out_off.ctypes.data))
        else:
            ctx.check(ctx.L.kc_zstd_encode_units_dev_end_at(ctx.h, d_dst_ptr, int(dst_cap), out_off.ctypes.data))
        return out_off

    def ChainAfter(self, other):
        """Pipelining two encoders: this one's match finder waits for `other`'s last one (see include/kcgpu.h)."""
        self.ctx().L.kc_ctx_chain_after(self.ctx().h, other.ctx().h if other is not None else None)

    def DebugParseDevice(self, d_src_ptr, unit_off):
        """Match-finder intermediates for parity tests: list over blocks of (seqs[n,3] u32, extra_lits)."""
        import numpy as np
        ctx = self.ctx()
        unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
        n = len(unit_off) - 1
        total = int(unit_off[n] - unit_off[0])
        bs = self.o.block_size
        blk_cap = total // bs + n + 2
        seqs = np.empty((total // 4 + 16, 3), dtype=np.uint32)
        first = np.zeros(blk_cap + 1, dtype=np.uint64)
        extra = np.zeros(blk_cap, dtype=np.uint32)
        flags = np.zeros(blk_cap, dtype=np.uint32)
        nb = C.c_uint32()
        ctx.check(ctx.L.kc_zstd_debug_parse_dev(ctx.h, C.byref(self.o), d_src_ptr, unit_off.ctypes.data, n, seqs.ctypes.data,
                                                len(seqs), first.ctypes.data, extra.ctypes.data, flags.ctypes.data, blk_cap,
                                                C.byref(nb)))
        self.last_block_flags = flags[:nb.value].copy()
        return [(seqs[int(first[b]):int(first[b + 1])].copy(), int(extra[b])) for b in range(nb.value)]

    def Close(self):
        """Encoder.Close (encoder.go:589): finish the stream, if one was written to a writer; the device context is released
        and re-created on the next use (the encoder stays usable after Reset, like the reference's)."""
        self._finish_stream()
        with self._lock:
            spare, self._spare = self._spare, []
            c, self._ctx = self._ctx, None
        for x in spare:
            x.close()
        if c is not None:
            c.close()


def NewWriter(w, *opts, **kw):
    """zstd.NewWriter(w, opts...) (encoder.go:71).  w (may be None for block-API use) receives the stream on Close."""
    return Encoder(*opts, w=w, **kw)
