"""ctypes binding of the CPU oracle (oracle/libkcoracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ODIR, "libkcoracle.so")


class ZstdOpts(C.Structure):
    _fields_ = [
        ("level", C.c_int32), ("window_size", C.c_int32), ("block_size", C.c_int32), ("crc", C.c_int32),
        ("single", C.c_int32), ("full_zero", C.c_int32), ("no_entropy", C.c_int32), ("all_lit_entropy", C.c_int32),
        ("low_mem", C.c_int32), ("dict_id", C.c_uint32), ("dict", C.c_char_p), ("dict_len", C.c_uint64),
        ("dict_full", C.c_int32), ("concurrent", C.c_int32),
    ]


def build(force=False):
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_ODIR, f)) > os.path.getmtime(_SO)
            for f in os.listdir(_ODIR) if f.endswith((".h", ".cpp"))):
        subprocess.check_call(["make", "-C", _ODIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u64p, u32p = C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.kco_zstd_encoder_new.restype = C.c_void_p
        L.kco_zstd_encoder_new.argtypes = [C.POINTER(ZstdOpts)]
        L.kco_zstd_encoder_free.argtypes = [C.c_void_p]
        L.kco_zstd_encode_all.restype = C.c_int64
        L.kco_zstd_encode_all.argtypes = [C.c_void_p, u8p, C.c_uint64, u8p, C.c_uint64]
        L.kco_zstd_max_encoded_size.restype = C.c_int64
        L.kco_zstd_max_encoded_size.argtypes = [C.POINTER(ZstdOpts), C.c_int64]
        L.kco_zstd_encode_units.restype = C.c_int64
        L.kco_zstd_encode_units.argtypes = [C.POINTER(ZstdOpts), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.kco_zstd_parse_unit.restype = C.c_int64
        L.kco_zstd_parse_unit.argtypes = [C.POINTER(ZstdOpts), u8p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
        L.kco_xxh64.restype = C.c_uint64
        L.kco_xxh64.argtypes = [u8p, C.c_uint64]
        L.kco_xxh64_chunked.restype = C.c_uint64
        L.kco_xxh64_chunked.argtypes = [u8p, C.c_uint64, C.c_uint64]
        L.kco_zstd_matchlen.restype = C.c_int32
        L.kco_zstd_matchlen.argtypes = [u8p, C.c_uint64, u8p]
        L.kco_zstd_hashlen.restype = C.c_uint32
        L.kco_zstd_hashlen.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.kco_huff0_compress.restype = C.c_int64
        L.kco_huff0_compress.argtypes = [u8p, C.c_uint64, C.c_int, C.c_int, u8p, C.c_uint64]
        L.kco_fse_compress.restype = C.c_int64
        L.kco_fse_compress.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
        L.kco_s2_max_encoded_len.restype = C.c_int64
        L.kco_s2_max_encoded_len.argtypes = [C.c_int64]
        for name in ("kco_s2_encode", "kco_s2_encode_block", "kco_s2_decode"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
        L.kco_s2_emit_literal.restype = C.c_int64
        L.kco_s2_emit_literal.argtypes = [u8p, u8p, C.c_uint64]
        L.kco_s2_emit_copy.restype = C.c_int64
        L.kco_s2_emit_copy.argtypes = [u8p, C.c_int64, C.c_int64]
        L.kco_s2_emit_repeat.restype = C.c_int64
        L.kco_s2_emit_repeat.argtypes = [u8p, C.c_int64, C.c_int64]
        L.kco_s2_crc.restype = C.c_uint32
        L.kco_s2_crc.argtypes = [u8p, C.c_uint64]
        L.kco_s2_encode_stream.restype = C.c_int64
        L.kco_s2_encode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.kco_s2_decode_stream.restype = C.c_int64
        L.kco_s2_decode_stream.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
        for name in ("kco_s2_encode_blocks", "kco_s2_encode_blocks_better", "kco_s2_encode_blocks_snappy", "kco_s2_encode_blocks_snappy_better"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        for name in ("kco_s2_encode_better", "kco_s2_encode_snappy", "kco_s2_encode_snappy_better", "kco_s2_encode_best", "kco_s2_encode_snappy_best"):
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
        _lib = L
    return _lib


def make_opts(level=2, window_size=None, block_size=None, crc=True, single=None, full_zero=True,
              no_entropy=False, all_lit_entropy=None, low_mem=False, dict_id=0, dict_content=None, dict_blob=None, concurrent=0):
    """Resolved options.  Defaults follow encoderOptions.setDefault + WithEncoderLevel
    (zstd/encoder_options.go:36-48,236-266)."""
    if window_size is None:
        window_size = (4 << 20) if level == 1 else (8 << 20)
    if block_size is None:
        block_size = (1 << 16) if level == 1 else (128 << 10)
        block_size = min(block_size, window_size)
    if all_lit_entropy is None:
        all_lit_entropy = level > 2
    o = ZstdOpts()
    o.level, o.window_size, o.block_size, o.crc = level, window_size, block_size, int(crc)
    o.single = -1 if single is None else int(single)
    o.full_zero, o.no_entropy, o.all_lit_entropy, o.low_mem = int(full_zero), int(no_entropy), int(all_lit_entropy), int(low_mem)
    o.dict_id = dict_id
    o.concurrent = int(concurrent)  # WithEncoderConcurrency: 1 = the synchronous nextBlock form (dictionary streams differ)
    o._keep = dict_content
    o.dict = dict_content
    o.dict_len = len(dict_content) if dict_content else 0
    if dict_blob is not None:  # WithEncoderDict: full-format dictionary, parsed by the oracle's loadDict
        o._keep = dict_blob
        o.dict = dict_blob
        o.dict_len = len(dict_blob)
        o.dict_full = 1
    return o


class ZstdOracle:
    """One persistent reference-encoder state (like one pooled Go encoder)."""

    def __init__(self, **kw):
        self.opts = make_opts(**kw)
        self.h = lib().kco_zstd_encoder_new(C.byref(self.opts))

    def __del__(self):
        if getattr(self, "h", None):
            lib().kco_zstd_encoder_free(self.h)
            self.h = None

    def max_encoded_size(self, n):
        return lib().kco_zstd_max_encoded_size(C.byref(self.opts), n)

    def encode_stream(self, src: bytes, flush_at=()) -> bytes:
        """NewWriter(w); Write(src) with Flush() at the given input positions; Close()."""
        import numpy as np
        L = lib()
        L.kco_zstd_encode_stream.restype = C.c_int64
        L.kco_zstd_encode_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        cuts = np.ascontiguousarray(sorted(flush_at), dtype=np.uint64)
        cap = self.max_encoded_size(len(src)) + 64 + 3 * (len(cuts) + 2)
        buf = C.create_string_buffer(cap)
        r = L.kco_zstd_encode_stream(self.h, src, len(src), cuts.ctypes.data if len(cuts) else None, len(cuts), buf, cap)
        if r < 0:
            raise RuntimeError("oracle encode_stream failed: %d" % r)
        return buf.raw[:r]

    def encode_jobs(self, src: bytes, flush_at=()) -> bytes:
        """NewWriter(w, WithConcurrentBlocks(true), concurrency > 1); Write(src) with Flush at flush_at; Close()."""
        import numpy as np
        L = lib()
        L.kco_zstd_encode_jobs.restype = C.c_int64
        L.kco_zstd_encode_jobs.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        cuts = np.ascontiguousarray(sorted(flush_at), dtype=np.uint64)
        cap = len(src) + len(src) // 64 + 3 * (len(src) // 1024 + len(cuts)) + 4096
        buf = C.create_string_buffer(cap)
        r = L.kco_zstd_encode_jobs(self.h, src, len(src), cuts.ctypes.data if len(cuts) else None, len(cuts), buf, cap)
        if r < 0:
            raise RuntimeError("oracle encode_jobs failed: %d" % r)
        return buf.raw[:r]

    def encode_all(self, src: bytes) -> bytes:
        cap = self.max_encoded_size(len(src)) + 64
        buf = C.create_string_buffer(cap)
        r = lib().kco_zstd_encode_all(self.h, src, len(src), buf, cap)
        if r < 0:
            raise RuntimeError("oracle encode_all failed: %d" % r)
        return buf.raw[:r]


def zstd_encode_units(src, unit_off, threads=1, **kw):
    """src: bytes/numpy u8; unit_off: numpy u64 [n+1]. Returns (bytes, out_off numpy)."""
    import numpy as np
    opts = make_opts(**kw)
    src = np.ascontiguousarray(np.frombuffer(src, dtype=np.uint8) if isinstance(src, (bytes, bytearray)) else src)
    unit_off = np.ascontiguousarray(unit_off, dtype=np.uint64)
    n = len(unit_off) - 1
    sizes = np.diff(unit_off.astype(np.int64)) if n else np.zeros(0, dtype=np.int64)
    mes = {int(z): lib().kco_zstd_max_encoded_size(C.byref(opts), int(z)) for z in np.unique(sizes)}
    cap = int(sum(mes[int(z)] * int(c) for z, c in zip(*np.unique(sizes, return_counts=True)))) + 64
    dst = np.empty(cap, dtype=np.uint8)
    out_off = np.empty(n + 1, dtype=np.uint64)
    r = lib().kco_zstd_encode_units(C.byref(opts), src.ctypes.data, unit_off.ctypes.data, n, dst.ctypes.data, cap, out_off.ctypes.data, threads)
    if r < 0:
        raise RuntimeError("oracle encode_units failed: %d" % r)
    return dst[:r], out_off


def zstd_decode(enc: bytes, cap: int, dict_content: bytes = None, dict_blob: bytes = None) -> bytes:
    """The oracle's own zstd decoder (oracle/kco_zstd_dec.h): concatenated frames -> content.  Raises on malformed input."""
    L = lib()
    L.kco_zstd_decode.restype = C.c_int64
    L.kco_zstd_decode.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int]
    buf = C.create_string_buffer(max(cap, 1))
    d = dict_blob if dict_blob is not None else dict_content
    r = L.kco_zstd_decode(enc, len(enc), buf, cap, d, len(d) if d else 0, 1 if dict_blob is not None else 0)
    if r < 0:
        raise RuntimeError("oracle zstd decode failed: %d" % r)
    return buf.raw[:r]


def zstd_load_dict(blob: bytes):
    """loadDict as the encoder sees it: dict(id, offsets, val[256], nbits[256], huf_len, huf_log, content_off) or None."""
    L = lib()
    L.kco_zstd_load_dict.restype = C.c_int
    L.kco_zstd_load_dict.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_int32 * 3), C.POINTER(C.c_uint16 * 256),
                                     C.POINTER(C.c_uint8 * 256), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    id_, offs, val, nb = C.c_uint32(), (C.c_int32 * 3)(), (C.c_uint16 * 256)(), (C.c_uint8 * 256)()
    hl, hg, co = C.c_int32(), C.c_int32(), C.c_uint64()
    if L.kco_zstd_load_dict(blob, len(blob), C.byref(id_), C.byref(offs), C.byref(val), C.byref(nb), C.byref(hl), C.byref(hg), C.byref(co)) != 0:
        return None
    return {"id": id_.value, "offsets": list(offs), "val": list(val), "nbits": list(nb), "huf_len": hl.value, "huf_log": hg.value,
            "content_off": co.value}


def zstd_parse_unit(src: bytes, **kw):
    """Match-finder intermediates: list of (seqs[n,3] u32, literals bytes) per block."""
    import numpy as np
    opts = make_opts(**kw)
    n = len(src)
    seqs = np.empty((n // 3 + 16, 3), dtype=np.uint32)
    lits = np.empty(n + 16, dtype=np.uint8)
    mb = n // max(1, opts.block_size) + 2
    nseq = np.empty(mb, dtype=np.uint32)
    nlit = np.empty(mb, dtype=np.uint32)
    nb = lib().kco_zstd_parse_unit(C.byref(opts), src, n, seqs.ctypes.data, len(seqs), lits.ctypes.data, len(lits),
                                   nseq.ctypes.data, nlit.ctypes.data, mb)
    if nb < 0:
        raise RuntimeError("oracle parse_unit failed: %d" % nb)
    out, so, lo = [], 0, 0
    for b in range(nb):
        out.append((seqs[so:so + nseq[b]].copy(), lits[lo:lo + nlit[b]].tobytes()))
        so += int(nseq[b])
        lo += int(nlit[b])
    return out


def s2_encode(src: bytes) -> bytes:
    cap = lib().kco_s2_max_encoded_len(len(src))
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_encode(src, len(src), buf, cap)
    if r < 0:
        raise RuntimeError("s2 encode failed %d" % r)
    return buf.raw[:r]


def s2_encode_better(src: bytes) -> bytes:
    """s2.EncodeBetter(nil, src) (s2/encode.go:117)."""
    cap = lib().kco_s2_max_encoded_len(len(src))
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_encode_better(src, len(src), buf, cap)
    if r < 0:
        raise RuntimeError("s2 encode_better failed %d" % r)
    return buf.raw[:r]


def s2_encode_snappy(src: bytes) -> bytes:
    """s2.EncodeSnappy(nil, src) (s2/encode.go:204): Snappy-compatible block (no repeat tags)."""
    cap = lib().kco_s2_max_encoded_len(len(src))
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_encode_snappy(src, len(src), buf, cap)
    if r < 0:
        raise RuntimeError("s2 encode_snappy failed %d" % r)
    return buf.raw[:r]


def s2_encode_best(src: bytes) -> bytes:
    """s2.EncodeBest(nil, src) (s2/encode.go:161): restated for the next device level; no device kernel yet."""
    cap = lib().kco_s2_max_encoded_len(len(src))
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_encode_best(src, len(src), buf, cap)
    if r < 0:
        raise RuntimeError("s2 encode_best failed %d" % r)
    return buf.raw[:r]


def s2_encode_snappy_best(src: bytes) -> bytes:
    """s2.EncodeSnappyBest(nil, src) (s2/encode.go:292): oracle only."""
    cap = lib().kco_s2_max_encoded_len(len(src))
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_encode_snappy_best(src, len(src), buf, cap)
    if r < 0:
        raise RuntimeError("s2 encode_snappy_best failed %d" % r)
    return buf.raw[:r]


def s2_encode_snappy_better(src: bytes) -> bytes:
    """s2.EncodeSnappyBetter(nil, src) (s2/encode.go:248): the better parse, Snappy-compatible output."""
    cap = lib().kco_s2_max_encoded_len(len(src))
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_encode_snappy_better(src, len(src), buf, cap)
    if r < 0:
        raise RuntimeError("s2 encode_snappy_better failed %d" % r)
    return buf.raw[:r]


def s2_encode_blocks(src, blk_off, threads=1, better=False, snappy=False, level=None):
    """N x s2.Encode (or s2.EncodeBetter) on host threads.  src: numpy u8; blk_off: numpy u64 [n+1].  Returns (numpy u8, out_off numpy u64[n+1])."""
    import numpy as np
    src = np.ascontiguousarray(src, dtype=np.uint8)
    blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
    n = len(blk_off) - 1
    sizes = np.diff(blk_off.astype(np.int64)) if n else np.zeros(0, dtype=np.int64)
    cap = int(sum(lib().kco_s2_max_encoded_len(int(z)) * int(c) for z, c in zip(*np.unique(sizes, return_counts=True)))) + 64
    dst = np.empty(cap, dtype=np.uint8)
    out_off = np.empty(n + 1, dtype=np.uint64)
    L = lib()
    if level is not None:  # 0..5 (4: s2.EncodeBest, 5: s2.EncodeSnappyBest)
        L.kco_s2_encode_blocks_level.restype = C.c_int64
        L.kco_s2_encode_blocks_level.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int]
        r = L.kco_s2_encode_blocks_level(src.ctypes.data, blk_off.ctypes.data, n, dst.ctypes.data, cap, out_off.ctypes.data, int(threads), int(level))
        if r < 0:
            raise RuntimeError("oracle s2_encode_blocks failed: %d" % r)
        return dst[:r], out_off
    fn = (L.kco_s2_encode_blocks_snappy_better if (better and snappy) else L.kco_s2_encode_blocks_better if better
          else L.kco_s2_encode_blocks_snappy if snappy else L.kco_s2_encode_blocks)
    r = fn(src.ctypes.data, blk_off.ctypes.data, n, dst.ctypes.data, cap, out_off.ctypes.data, int(threads))
    if r < 0:
        raise RuntimeError("oracle s2_encode_blocks failed: %d" % r)
    return dst[:r], out_off


def s2_encode_block(src: bytes) -> bytes:
    cap = lib().kco_s2_max_encoded_len(len(src))
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_encode_block(src, len(src), buf, cap)
    if r < 0:
        raise RuntimeError("s2 encode_block failed %d" % r)
    return buf.raw[:r]


def s2_encode_asm(src: bytes, snappy=False, better=False) -> bytes:
    """s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter as an amd64 build of the reference writes them: the oracle's
    restatement of the assembly encoders (oracle/kco_s2_asm.h), pinned against the assembly itself by tests/test_ref_s2asm.py."""
    L = lib()
    L.kco_s2_encode_asm_level.restype = C.c_int64
    L.kco_s2_encode_asm_level.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int]
    cap = L.kco_s2_max_encoded_len(len(src)) + 64
    buf = C.create_string_buffer(cap)
    r = L.kco_s2_encode_asm_level(src, len(src), buf, cap, (2 if snappy else 0) + (1 if better else 0))
    if r < 0:
        raise RuntimeError("s2 encode_asm failed %d" % r)
    return buf.raw[:r]


def s2_emit_literal(lit: bytes) -> bytes:
    buf = C.create_string_buffer(len(lit) + 16)
    r = lib().kco_s2_emit_literal(buf, lit, len(lit))
    return buf.raw[:r]


def s2_emit_copy(offset: int, length: int) -> bytes:
    buf = C.create_string_buffer(64)
    r = lib().kco_s2_emit_copy(buf, offset, length)
    return buf.raw[:r]


def s2_emit_repeat(offset: int, length: int) -> bytes:
    buf = C.create_string_buffer(64)
    r = lib().kco_s2_emit_repeat(buf, offset, length)
    return buf.raw[:r]


def s2_decode(enc: bytes, cap: int) -> bytes:
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_decode(enc, len(enc), buf, cap)
    if r < 0:
        raise RuntimeError("s2 decode failed %d" % r)
    return buf.raw[:r]


# ---- independent zstd decoder: system libzstd 1.4.8 (runtime only) ----
_zstd = None


def libzstd():
    global _zstd
    if _zstd is None:
        Z = C.CDLL("libzstd.so.1")
        Z.ZSTD_decompress.restype = C.c_size_t
        Z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        Z.ZSTD_isError.restype = C.c_uint
        Z.ZSTD_isError.argtypes = [C.c_size_t]
        Z.ZSTD_getErrorName.restype = C.c_char_p
        Z.ZSTD_getErrorName.argtypes = [C.c_size_t]
        Z.ZSTD_createDCtx.restype = C.c_void_p
        Z.ZSTD_freeDCtx.argtypes = [C.c_void_p]
        Z.ZSTD_decompress_usingDict.restype = C.c_size_t
        Z.ZSTD_decompress_usingDict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        _zstd = Z
    return _zstd


def zstd_decompress(enc: bytes, cap: int, dict_content: bytes = None) -> bytes:
    """Independent decoder for round trips: the system libzstd when it is installed, cross-checked against the in-repo decoder
    (oracle/kco_zstd_dec.h); the in-repo decoder alone when it is not."""
    is_full = bool(dict_content) and dict_content[:4] == b"\x37\xa4\x30\xec"
    own = zstd_decode(enc, cap, dict_content=None if is_full else dict_content, dict_blob=dict_content if is_full else None)
    try:
        Z = libzstd()
    except OSError:
        return own
    buf = C.create_string_buffer(max(cap, 1))
    if dict_content:
        ctx = Z.ZSTD_createDCtx()
        r = Z.ZSTD_decompress_usingDict(ctx, buf, cap, enc, len(enc), dict_content, len(dict_content))
        Z.ZSTD_freeDCtx(ctx)
    else:
        r = Z.ZSTD_decompress(buf, cap, enc, len(enc))
    if not Z.ZSTD_isError(r) and buf.raw[:r] != own:
        raise RuntimeError("in-repo zstd decoder and libzstd disagree")
    if Z.ZSTD_isError(r):
        raise RuntimeError("libzstd: " + Z.ZSTD_getErrorName(r).decode())
    return buf.raw[:r]


def s2_encode_stream(src, blk_off, with_stream_id=True, level=0):
    """Reference s2.Writer framing of the given blocks: (numpy u8 stream, out_off[n+1]).  level: 0 default, 1 better, 2 Snappy
    compatible, 3 both, 4 best, 5 best + Snappy compatible."""
    import numpy as np
    src = np.ascontiguousarray(src, dtype=np.uint8)
    blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
    n = len(blk_off) - 1
    cap = int(blk_off[n] - blk_off[0]) + 16 * n + 64
    dst = np.empty(cap, dtype=np.uint8)
    oo = np.empty(n + 1, dtype=np.uint64)
    L = lib()
    if level:
        L.kco_s2_encode_stream_level.restype = C.c_int64
        L.kco_s2_encode_stream_level.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int]
        r = L.kco_s2_encode_stream_level(src.ctypes.data, blk_off.ctypes.data, n, dst.ctypes.data, cap, oo.ctypes.data, int(with_stream_id), int(level))
        if r < 0:
            raise RuntimeError("s2 encode_stream failed %d" % r)
        return dst[:r], oo
    r = lib().kco_s2_encode_stream(src.ctypes.data, blk_off.ctypes.data, n, dst.ctypes.data, cap, oo.ctypes.data, int(with_stream_id))
    if r < 0:
        raise RuntimeError("s2 encode_stream failed %d" % r)
    return dst[:r], oo


def s2_index(block_size, adds, uncomp_total, comp_total) -> bytes:
    """s2.Index after add(comp, uncomp) for each pair in `adds`, serialised by appendTo (s2/index.go:57-236)."""
    import numpy as np
    L = lib()
    L.kco_s2_index.restype = C.c_int64
    L.kco_s2_index.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_char_p, C.c_uint64]
    comp = np.ascontiguousarray([a[0] for a in adds], dtype=np.int64)
    unc = np.ascontiguousarray([a[1] for a in adds], dtype=np.int64)
    cap = 64 + 20 * (len(adds) + 4)
    buf = C.create_string_buffer(cap)
    r = L.kco_s2_index(block_size, comp.ctypes.data, unc.ctypes.data, len(adds), uncomp_total, comp_total, buf, cap)
    if r < 0:
        raise RuntimeError("s2 index failed %d" % r)
    return buf.raw[:r]


def s2_decode_stream(enc: bytes, cap: int) -> bytes:
    buf = C.create_string_buffer(max(cap, 1))
    r = lib().kco_s2_decode_stream(enc, len(enc), buf, cap)
    if r < 0:
        raise RuntimeError("s2 decode_stream failed %d" % r)
    return buf.raw[:r]
