"""ctypes binding of oracle/_ref/libs2ref.so — TEST INFRASTRUCTURE ONLY: the reference's OWN amd64 S2 block encoders.

The library is the reference's generated Plan 9 assembly (s2/encodeblock_amd64.s) re-spelt for the GNU assembler and assembled
here (oracle/Makefile `ref`, oracle/ref_s2asm/): what an amd64 user of s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter
runs.  It is built where /root/reference exists (this container) and travels to the GPU box as a built file; available() says
whether it can be used (x86-64 host and the file present or buildable)."""
import ctypes as C
import os
import platform
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ODIR, "_ref", "libs2ref.so")
_REFSRC = "/root/reference/s2/encodeblock_amd64.s"
_lib = None


def available():
    if platform.machine() not in ("x86_64", "AMD64"):
        return False
    return os.path.exists(_SO) or os.path.exists(_REFSRC)


def lib():
    global _lib
    if _lib is None:
        if os.path.exists(_REFSRC):
            subprocess.check_call(["make", "-C", _ODIR, "-s", "ref"])
        L = C.CDLL(_SO)
        L.s2ref_encode.restype = C.c_int64
        L.s2ref_encode.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.s2ref_encode_block.restype = C.c_int64
        L.s2ref_encode_block.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.s2ref_emit_literal.restype = C.c_int64
        L.s2ref_emit_literal.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        for n in ("s2ref_emit_repeat", "s2ref_emit_copy", "s2ref_emit_copy_norepeat"):
            f = getattr(L, n)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64]
        L.s2ref_match_len.restype = C.c_int64
        L.s2ref_match_len.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.s2ref_encode_blocks_size.restype = C.c_int64
        L.s2ref_encode_blocks_size.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def max_encoded_len(n):
    """s2.MaxEncodedLen (s2/encode.go:309-331)."""
    n = int(n)
    return n + 5 + (n + 5) // 6 + 32  # a safe bound for the scratch this binding allocates (never smaller than MaxEncodedLen)


def encode(src: bytes, level=0) -> bytes:
    """s2.Encode (0) / EncodeBetter (1) / EncodeSnappy (2) / EncodeSnappyBetter (3) of an amd64 build of the reference."""
    cap = max_encoded_len(len(src)) + 64
    buf = C.create_string_buffer(cap)
    sb = C.create_string_buffer(src, len(src) + 16)  # (the encoders never read past src; the slack guards this binding, not them)
    r = lib().s2ref_encode(level, buf, cap, sb, len(src))
    if r < 0:
        raise RuntimeError("s2ref_encode failed: %d" % r)
    return buf.raw[:r]


def encode_blocks(src, blk_off, level=0, threads=1):
    """N x s2.Encode / ... of the amd64 build on `threads` host threads: (numpy u8 of the blocks back to back, out_off[n+1])."""
    import numpy as np
    src = np.ascontiguousarray(src, dtype=np.uint8)
    blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
    n = len(blk_off) - 1
    cap = sum(max_encoded_len(int(blk_off[i + 1] - blk_off[i])) for i in range(n)) + 64 if n < 100000 else int((blk_off[n] - blk_off[0]) * 1.2) + 64 * n
    dst = np.empty(cap, dtype=np.uint8)
    oo = np.zeros(n + 1, dtype=np.uint64)
    L = lib()
    L.s2ref_encode_blocks.restype = C.c_int64
    L.s2ref_encode_blocks.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    pad = np.concatenate([src, np.zeros(16, dtype=np.uint8)])
    r = L.s2ref_encode_blocks(level, pad.ctypes.data, blk_off.ctypes.data, n, dst.ctypes.data, cap, oo.ctypes.data, int(threads))
    if r < 0:
        raise RuntimeError("s2ref_encode_blocks failed: %d" % r)
    return dst[:r], oo


def decode(enc: bytes, cap: int):
    """s2.Decode of an amd64 build of the reference (its assembly block decoder): the bytes, or None for ErrCorrupt."""
    L = lib()
    L.s2ref_decode.restype = C.c_int64
    L.s2ref_decode.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
    buf = C.create_string_buffer(max(int(cap), 1) + 32)
    sb = C.create_string_buffer(enc, len(enc) + 32)
    r = L.s2ref_decode(sb, len(enc), buf, int(cap))
    if r < 0:  # -1: ErrCorrupt; -2: the block states more bytes than `cap` (for a caller that knows the size: corrupt as well)
        return None
    return buf.raw[:r]


def xxh64(b: bytes) -> int:
    """xxhash.Sum64 of an amd64 build of the reference (zstd/internal/xxhash/xxhash_amd64.s): the frame checksum's hash."""
    L = lib()
    L.zref_xxh64_sum.restype = C.c_uint64
    L.zref_xxh64_sum.argtypes = [C.c_char_p, C.c_uint64]
    return int(L.zref_xxh64_sum(b, len(b)))


def emit(kind, offset, length) -> bytes:
    """emitRepeat / emitCopy / emitCopyNoRepeat of the assembly (kind: 'repeat', 'copy', 'copy_norepeat')."""
    buf = C.create_string_buffer(64)
    fn = {"repeat": lib().s2ref_emit_repeat, "copy": lib().s2ref_emit_copy, "copy_norepeat": lib().s2ref_emit_copy_norepeat}[kind]
    r = fn(buf, 64, int(offset), int(length))
    return buf.raw[:r]


def emit_literal(lit: bytes) -> bytes:
    cap = len(lit) + 64
    buf = C.create_string_buffer(cap)
    r = lib().s2ref_emit_literal(buf, cap, lit, len(lit))
    return buf.raw[:r]


def match_len(a: bytes, b: bytes) -> int:
    return int(lib().s2ref_match_len(a, len(a), b, len(b)))


def zstd_match_len(a: bytes, b: bytes) -> int:
    """matchLen of zstd/matchlen_amd64.s (the zstd package's own copy of the routine)."""
    L = lib()
    L.zstdref_match_len.restype = C.c_int64
    L.zstdref_match_len.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    return int(L.zstdref_match_len(a, len(a), b, len(b)))
