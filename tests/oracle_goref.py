"""ctypes binding of oracle/_ref/libzstdref.so — TEST INFRASTRUCTURE ONLY: the reference's OWN pure-Go encoders, translated.

The library is the reference's Go source for the zstd encode path (encoder.go encodeAll, enc_fast / enc_dfast / enc_better /
enc_best, blockenc, fse_encoder, huff0, fse, xxhash) and for its portable-Go S2 block encoders (encode_all / encode_better /
encode_best, encode_go), translated statement by statement into C++ at build time (oracle/ref_go/go2cpp.py; Go's integer,
slice and array semantics in oracle/ref_go/gort.h) and compiled here (oracle/Makefile `ref`).  It is built where /root/reference
exists (this container) and travels to the GPU box as a built file; nothing of the reference is stored in the repository."""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ODIR, "_ref", "libzstdref.so")
# The same translation in the reference's amd64 build flavour: the Go halves of its assembly routines + the assembly itself
# (zstd/seqdec_amd64.s, fse_decoder_amd64.s, matchlen_amd64.s, huff0/decompress_amd64.s) — what an x86-64 user of the package runs.
_SO_AMD64 = os.path.join(_ODIR, "_ref", "libzstdref_amd64.so")
_REFSRC = "/root/reference/zstd/encoder.go"
_libs = {}
FLAVOURS = ("generic", "amd64", "amd64-nobmi")   # amd64-nobmi: the amd64 flavour with its BMI1 / BMI2 routines switched off
_flavour = "generic"


class flavour:
    """with oracle_goref.flavour("amd64"): ... — which build of the reference the calls inside go to."""
    def __init__(self, name):
        assert name in FLAVOURS, name
        self.name = name

    def __enter__(self):
        global _flavour
        self.prev, _flavour = _flavour, self.name
        return self

    def __exit__(self, *a):
        global _flavour
        _flavour = self.prev


def available():
    return os.path.exists(_SO) or os.path.exists(_REFSRC)


def amd64_available():
    import platform
    return platform.machine() in ("x86_64", "AMD64") and (os.path.exists(_SO_AMD64) or os.path.exists(_REFSRC))


def lib():
    key = "generic" if _flavour == "generic" else "amd64"
    L = _libs.get(key)
    if L is None:
        so = _SO if key == "generic" else _SO_AMD64
        if os.path.exists(_REFSRC):
            subprocess.check_call(["make", "-C", _ODIR, "-s", "_ref/" + os.path.basename(so)])
        L = C.CDLL(so)
        L.goref_force_bmi.restype = None
        L.goref_force_bmi.argtypes = [C.c_int]
        L.goref_zstd_encode_all.restype = C.c_longlong
        L.goref_zstd_encode_all.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong] + [C.c_int] * 8 + [C.c_char_p, C.c_longlong, C.c_uint,
                                            C.c_char_p, C.c_int]
        L.goref_zstd_encode_stream.restype = C.c_longlong
        L.goref_zstd_encode_stream.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong] + [C.c_int] * 7 + [C.c_char_p, C.c_longlong, C.c_uint,
                                               C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
        L.goref_zstd_encode_all_reuse.restype = C.c_longlong
        L.goref_zstd_encode_all_reuse.argtypes = [C.c_char_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_char_p,
                                                  C.c_longlong, C.c_uint, C.c_char_p, C.c_int]
        L.goref_zstd_next_stream_readfrom.restype = None
        L.goref_zstd_next_stream_readfrom.argtypes = [C.c_longlong]
        L.goref_zstd_decode_all.restype = C.c_longlong
        L.goref_zstd_decode_all.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
        L.goref_zstd_decode_all_dict.restype = C.c_longlong
        L.goref_zstd_decode_all_dict.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_char_p, C.c_longlong, C.c_uint, C.c_char_p, C.c_int]
        L.goref_zstd_max_encoded_size.restype = C.c_longlong
        L.goref_zstd_max_encoded_size.argtypes = [C.c_longlong, C.c_int, C.c_int]
        L.goref_zstd_job_geometry.restype = C.c_longlong
        L.goref_zstd_job_geometry.argtypes = [C.c_int, C.c_int, C.c_int]
        L.goref_calc_skippable_frame.restype = C.c_longlong
        L.goref_calc_skippable_frame.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
        L.goref_s2_max_encoded_len.restype = C.c_longlong
        L.goref_s2_max_encoded_len.argtypes = [C.c_longlong]
        L.goref_s2_next_stream_readfrom.restype = None
        L.goref_s2_next_stream_readfrom.argtypes = [C.c_longlong]
        L.goref_s2_next_stream_encode_buffer.restype = None
        L.goref_s2_next_stream_encode_buffer.argtypes = [C.c_longlong, C.c_longlong]
        L.goref_s2_stream.restype = C.c_longlong
        L.goref_s2_stream.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong] + [C.c_int] * 6 + [C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
        L.goref_s2_read_stream.restype = C.c_longlong
        L.goref_s2_read_stream.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.goref_s2_encode.restype = C.c_longlong
        L.goref_s2_encode.argtypes = [C.c_int, C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
        _libs[key] = L
    if key == "amd64":
        L.goref_force_bmi(0 if _flavour == "amd64-nobmi" else -1)
    return L


def _flag(v):
    return -1 if v is None else int(bool(v))


FULL_DICT = 0xFFFFFFFF  # dict_id value that marks dict_content as a full-format dictionary (WithEncoderDict: it carries its own id)


def zstd_encode_all(src: bytes, level=1, window_size=None, crc=None, single=None, full_zero=None, no_entropy=None, all_lit_entropy=None,
                    low_mem=False, dict_id=0, dict_content=None, dict_blob=None) -> bytes:
    """zstd.NewWriter(nil, WithEncoderLevel(level), <the options given>).EncodeAll(src, nil) of the reference (None: its default).
    dict_content + dict_id: WithEncoderDictRaw; dict_blob: WithEncoderDict (a full-format dictionary, parsed by the reference's loadDict)."""
    if dict_blob is not None:
        dict_id, dict_content = FULL_DICT, dict_blob
    src = bytes(src)
    cap = len(src) + (len(src) >> 6) + 1024
    out = C.create_string_buffer(cap)
    err = C.create_string_buffer(256)
    d = bytes(dict_content) if dict_content else None
    n = lib().goref_zstd_encode_all(src, len(src), out, cap, int(level), int(window_size or 0), _flag(crc), _flag(single), _flag(full_zero),
                                    _flag(no_entropy), _flag(all_lit_entropy), int(bool(low_mem)), d, len(d) if d else 0, int(dict_id), err, 256)
    if n < 0:
        raise RuntimeError("translated reference failed (%d): %s" % (n, err.value.decode(errors="replace")))
    return out.raw[:n]


def zstd_encode_stream(src: bytes, flush_at=(), level=1, window_size=None, crc=None, no_entropy=None, all_lit_entropy=None, low_mem=False,
                       concurrent=0, dict_id=0, dict_content=None, dict_blob=None, jobs=False, readfrom_at=None) -> bytes:
    """w := new(bytes.Buffer); e := zstd.NewWriter(w, WithEncoderLevel(level), <the options given>); e.Write(src[..cut]) and e.Flush()
    at every position of flush_at; e.Close(); w.Bytes() — the reference's streaming writer (encoder.go Write / nextBlock / Flush /
    Close).  concurrent=1: WithEncoderConcurrency(1), the synchronous nextBlock; 0: the reference's default, the asynchronous one
    (its two goroutines per block run to completion where they are started: the schedule their WaitGroups allow).
    jobs=True: WithConcurrentBlocks(true) (enc_jobs.go; needs concurrent > 1 like the reference): every job compressed and written where
    it is dispatched.  readfrom_at=a: the input from offset a on (behind the Flush points before it) is handed over with
    e.ReadFrom(bytes.NewReader(src[a:])) instead of Write."""
    import numpy as np
    if dict_blob is not None:
        dict_id, dict_content = FULL_DICT, dict_blob
    src = bytes(src)
    cuts = np.ascontiguousarray(sorted(int(x) for x in flush_at), dtype=np.int64)
    cap = len(src) + (len(src) >> 6) + 1024 + 4 * (len(cuts) + 2)
    out = C.create_string_buffer(cap)
    err = C.create_string_buffer(256)
    d = bytes(dict_content) if dict_content else None
    if readfrom_at is not None:
        lib().goref_zstd_next_stream_readfrom(int(readfrom_at))
    n = lib().goref_zstd_encode_stream(src, len(src), out, cap, int(level), int(window_size or 0), _flag(crc), _flag(no_entropy), _flag(all_lit_entropy),
                                       int(bool(low_mem)), int(concurrent) | (int(bool(jobs)) << 16), d, len(d) if d else 0, int(dict_id),
                                       cuts.ctypes.data if len(cuts) else None, len(cuts), err, 256)
    if n < 0:
        raise RuntimeError("translated reference failed (%d): %s" % (n, err.value.decode(errors="replace")))
    return out.raw[:n]


def zstd_encode_all_reuse(units, level=1, window_size=None, dict_id=0, dict_content=None, dict_blob=None):
    """e := zstd.NewWriter(nil, ...); for every unit e.EncodeAll(unit, nil) — on ONE encoder object of the reference, re-used from call to
    call without a reset in between (what a goroutine calling EncodeAll repeatedly gets from the Encoder's pool).  Returns the frames."""
    import numpy as np
    if dict_blob is not None:
        dict_id, dict_content = FULL_DICT, dict_blob
    off = np.zeros(len(units) + 1, dtype=np.int64)
    for i, u in enumerate(units):
        off[i + 1] = off[i] + len(u)
    src = b"".join(bytes(u) for u in units)
    cap = len(src) + (len(src) >> 6) + 1024 * (len(units) + 1)
    out = C.create_string_buffer(cap)
    oo = np.zeros(len(units) + 1, dtype=np.int64)
    err = C.create_string_buffer(256)
    d = bytes(dict_content) if dict_content else None
    n = lib().goref_zstd_encode_all_reuse(src, off.ctypes.data, len(units), out, cap, oo.ctypes.data, int(level), int(window_size or 0), d, len(d) if d else 0,
                                          int(dict_id), err, 256)
    if n < 0:
        raise RuntimeError("translated reference failed (%d): %s" % (n, err.value.decode(errors="replace")))
    raw = out.raw[:n]  # (one copy: .raw copies the whole buffer every time it is read)
    return [raw[int(oo[i]):int(oo[i + 1])] for i in range(len(units))]


def zstd_encode_units(src, unit_off, **kw):
    """N x EncodeAll: (bytes, offsets) like oracle_lib.zstd_encode_units."""
    import numpy as np
    buf = bytes(memoryview(np.ascontiguousarray(src, dtype=np.uint8)))
    outs, off = [], [0]
    for i in range(len(unit_off) - 1):
        f = zstd_encode_all(buf[int(unit_off[i]):int(unit_off[i + 1])], **kw)
        outs.append(f)
        off.append(off[-1] + len(f))
    return b"".join(outs), np.array(off, dtype=np.uint64)


def zstd_decode_all(frames: bytes, max_out: int, dict_id=0, dict_content=None, dict_blob=None) -> bytes:
    """zstd.NewReader(nil).DecodeAll(frames, nil) of the reference — its own decoder in the pure-Go form (what noasm / non-amd64 builds
    run), translated like the encoders: the judge of a frame's validity (block and literal section types, both Huffman forms, the FSE
    tables and their modes, sequence execution, window and size checks, the content checksum).  dict_content + dict_id:
    WithDecoderDictRaw; dict_blob: WithDecoderDicts (a full-format dictionary).  Raises on the decoder's error, with its message."""
    if dict_blob is not None:
        dict_id, dict_content = FULL_DICT, dict_blob
    frames = bytes(frames)
    out = C.create_string_buffer(max_out + 64)
    err = C.create_string_buffer(256)
    d = bytes(dict_content) if dict_content else None
    n = lib().goref_zstd_decode_all_dict(frames, len(frames), out, max_out + 64, d, len(d) if d else 0, int(dict_id), err, 256)
    if n < 0:
        raise ValueError("reference decoder: %s (%d)" % (err.value.decode(errors="replace"), n))
    return out.raw[:n]


def s2_encode(src: bytes, level=0) -> bytes:
    """s2.Encode (0) / EncodeBetter (1) / EncodeSnappy (2) / EncodeSnappyBetter (3) / EncodeBest (4) / EncodeSnappyBest (5) of a build
    of the reference without its assembly (the portable Go encoders)."""
    src = bytes(src)
    cap = len(src) + len(src) // 6 + 64
    out = C.create_string_buffer(cap)
    err = C.create_string_buffer(256)
    n = lib().goref_s2_encode(int(level), src, len(src), out, cap, err, 256)
    if n < 0:
        raise RuntimeError("translated reference failed (%d): %s" % (n, err.value.decode(errors="replace")))
    return out.raw[:n]


def s2_stream(src: bytes, flush_at=(), level=0, snappy=False, block_size=0, add_index=False, padding=0, flush_on_write=False, encode_buffer=None, readfrom_at=None) -> bytes:
    """w := new(bytes.Buffer); e := s2.NewWriter(w, WriterConcurrency(1), <options>); e.Write(src[..cut]) + e.Flush() at every position of
    flush_at; e.Close(); w.Bytes() — the reference's own stream writer in its synchronous form (s2/writer.go Write / writeSync / Flush /
    closeIndex, s2/index.go add / appendTo, skippableFrame), translated.  level: 0 default, 1 WriterBetterCompression, 2
    WriterBestCompression, 3 WriterUncompressed; snappy: WriterSnappyCompat; padding: WriterPadding(n) with zero bytes as the padding
    source; the chunk bodies are the portable Go block encoders' (s2_encode).  encode_buffer=(a, b): instead of the Write / Flush
    sequence, e.Write(src[:a]); e.EncodeBuffer(src[a:b]); e.Write(src[b:]); e.Close()."""
    import numpy as np
    src = bytes(src)
    if encode_buffer is not None:
        lib().goref_s2_next_stream_encode_buffer(int(encode_buffer[0]), int(encode_buffer[1]))
    if readfrom_at is not None:  # behind the Write / Flush sequence: e.ReadFrom(reader over src[a:])
        lib().goref_s2_next_stream_readfrom(int(readfrom_at))
    cuts = np.ascontiguousarray(sorted(int(x) for x in flush_at), dtype=np.int64)
    nblk = len(src) // (block_size or (1 << 20)) + len(cuts) + 2
    cap = len(src) + len(src) // 6 + 32 * nblk + 2 * max(int(padding), 0) + 20 * nblk + 4096
    out = C.create_string_buffer(cap)
    err = C.create_string_buffer(256)
    n = lib().goref_s2_stream(src, len(src), out, cap, int(level), int(bool(snappy)), int(block_size), int(bool(add_index)), int(padding),
                              int(bool(flush_on_write)), cuts.ctypes.data if len(cuts) else None, len(cuts), err, 256)
    if n < 0:
        raise RuntimeError("translated reference failed (%d): %s" % (n, err.value.decode(errors="replace")))
    return out.raw[:n]


def s2_read_stream(stream: bytes, max_out: int, max_block=0, ignore_crc=False) -> bytes:
    """io.ReadAll(s2.NewReader(bytes.NewReader(stream))) of the reference — its own stream reader (sequential Read of s2/reader.go:
    chunk types, masked-CRC check, S2 and Snappy stream identifiers, skippable / padding / index chunks) over its own block decoder in
    the portable Go form (decode_other.go), translated: the judge of a framed stream's validity.  Raises on the reader's error."""
    stream = bytes(stream)
    out = C.create_string_buffer(max_out + 64)
    err = C.create_string_buffer(256)
    n = lib().goref_s2_read_stream(stream, len(stream), out, max_out + 64, int(max_block), int(bool(ignore_crc)), err, 256)
    if n < 0:
        raise ValueError("reference s2.Reader: %s (%d)" % (err.value.decode(errors="replace"), n))
    return out.raw[:n]


def zstd_max_encoded_size(size: int, level=1, window_size=None) -> int:
    """(*zstd.Encoder).MaxEncodedSize(size) of the reference."""
    return int(lib().goref_zstd_max_encoded_size(int(size), int(level), int(window_size or 0)))


def s2_emit(kind: str, a: int, b: int = 0, lit: bytes = b"") -> bytes:
    """The reference's own block-format emitters (s2/encode_go.go), translated: kind 'literal' (lit), 'copy' / 'repeat' / 'copy_norepeat'
    (offset a, length b).  Returns the bytes the emitter wrote."""
    k = {"literal": 0, "copy": 1, "repeat": 2, "copy_norepeat": 3}[kind]
    L = lib()
    L.goref_s2_emit.restype = C.c_longlong
    L.goref_s2_emit.argtypes = [C.c_int, C.c_char_p, C.c_longlong, C.c_longlong, C.c_void_p, C.c_longlong]
    cap = len(lit) + 64
    out = C.create_string_buffer(cap)
    n = L.goref_s2_emit(k, bytes(lit), len(lit) if k == 0 else int(a), int(b), out, cap)
    if n < 0:
        raise ValueError("goref_s2_emit: %d" % n)
    return out.raw[:n]


def s2_decode(block: bytes, max_out: int) -> bytes:
    """s2.Decode(nil, block) of the reference (translated): one block, Snappy blocks included.  Raises on the decoder's error."""
    L = lib()
    L.goref_s2_decode.restype = C.c_longlong
    L.goref_s2_decode.argtypes = [C.c_char_p, C.c_longlong, C.c_void_p, C.c_longlong]
    out = C.create_string_buffer(max_out + 64)
    n = L.goref_s2_decode(bytes(block), len(block), out, max_out + 64)
    if n < 0:
        raise ValueError("reference s2.Decode: %d" % n)
    return out.raw[:n]


def s2_max_encoded_len(n: int) -> int:
    """s2.MaxEncodedLen(n) of the reference (-1: too large)."""
    return int(lib().goref_s2_max_encoded_len(int(n)))


def zstd_job_geometry(level=1, window_size=None):
    """(encoderOptions.jobSize(), encoderOptions.overlapSize()) of the reference for a level and a window."""
    L = lib()
    return int(L.goref_zstd_job_geometry(0, int(level), int(window_size or 0))), int(L.goref_zstd_job_geometry(1, int(level), int(window_size or 0)))


def calc_skippable_frame(written: int, want_multiple: int, s2=False) -> int:
    """calcSkippableFrame of the reference's zstd package (frameenc.go) or — s2=True — of its s2 package (writer.go); -1: it panicked."""
    return int(lib().goref_calc_skippable_frame(int(bool(s2)), int(written), int(want_multiple)))
