"""The LDS-table kernels compiled for the CPU wave emulator (tools/hipemu): build + ctypes wrappers.

Test infrastructure only.  The same .hip sources the product compiles with hipcc are compiled here with g++ against
tools/hipemu/hip/hip_runtime.h, which runs every lane of a wave as a fiber and every cross-lane operation as a
rendezvous; lanes run maximally out of lockstep in between, so an unfenced exchange through LDS shows up as a wrong
result.  This is how the device algorithms are checked against the oracle without a GPU (tests/test_emu_lds.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tools", "hipemu")
CSRC = os.path.join(ROOT, "compress_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "libkcemu.so")
_L = None


def build(force=False):
    srcs = [os.path.join(EMU, "kcemu.cpp"), os.path.join(EMU, "hipemu.cpp")]
    deps = srcs + [os.path.join(EMU, "hip", "hip_runtime.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", EMU, "-Wall", "-Wno-unused-function",
           "-Wno-unused-variable", "-Wno-unknown-pragmas"] + srcs + ["-o", OUT, "-ldl"]
    subprocess.check_call(cmd)
    return OUT


def lib():
    global _L
    if _L is None:
        _L = C.CDLL(build())
        _L.kcemu_s2_encode.restype = C.c_int
        _L.kcemu_s2_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _L.kcemu_collectives.restype = C.c_uint64
        if hasattr(_L, "kcemu_zfast_parse"):
            _L.kcemu_zfast_parse.restype = C.c_int
            _L.kcemu_zfast_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    return _L


def s2_max_encoded_len(n):
    # s2/encode.go:389-418 (the product's kc_s2_max_encoded_len restates the same)
    n = int(n)
    x = n + 5 + 3 * ((n + 65535) // 65536)  # generous: only sizes the staging slots
    return x + 16


def s2_encode_blocks(blocks, level=0, framed=False, spec_w0=8):
    """blocks: list of bytes -> list of bytes (uvarint + body, or the framed chunk)."""
    n = len(blocks)
    off = np.zeros(n + 1, dtype=np.uint64)
    for i, b in enumerate(blocks):
        off[i + 1] = off[i] + len(b)
    src = np.frombuffer(b"".join(blocks) + b"\0" * 64, dtype=np.uint8).copy()
    soff = np.zeros(n + 1, dtype=np.uint64)
    for i, b in enumerate(blocks):
        soff[i + 1] = soff[i] + ((s2_max_encoded_len(len(b)) + 8 + 63) & ~63)
    stage = np.zeros(int(soff[n]) + 64, dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint32)
    r = lib().kcemu_s2_encode(level, int(framed), spec_w0, src.ctypes.data, off.ctypes.data, n, stage.ctypes.data, soff.ctypes.data, sizes.ctypes.data)
    assert r == 0
    return [stage[int(soff[i]):int(soff[i]) + int(sizes[i])].tobytes() for i in range(n)]
