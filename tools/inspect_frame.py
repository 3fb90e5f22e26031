#!/usr/bin/env python3
"""Per-block structure of a zstd frame (oracle decoder's inspection hook): block types, literal types, sequence
compression modes and the first sequences of every compressed block.  Used for the provenance question of the
reference's zstd/testdata/z000028.zst: the klauspost encoders never emit a repeat-offset code before a
block's 4th sequence (`canRepeat := len(blk.sequences) > 2`, enc_fast.go:117, enc_dfast.go:119, enc_better.go:158).

    python tools/inspect_frame.py /root/reference/zstd/testdata/z000028.zst
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib  # noqa: E402

L = oracle_lib.lib()
L.kco_zstd_inspect.restype = C.c_int64
L.kco_zstd_inspect.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
data = open(sys.argv[1], "rb").read()
buf = C.create_string_buffer(1 << 20)
r = L.kco_zstd_inspect(data, len(data), buf, len(buf))
print(buf.value.decode() if r >= 0 else "inspect failed: %d" % r)
