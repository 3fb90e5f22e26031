"""Diagnostic: device parse at SpeedBestCompression against the oracle on the J corpus (first differing sequence per unit)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from compress_amd import zstd
import corpora, oracle_lib
buf = corpora.corpus("J", 96, 131072)
idx = [int(x) for x in sys.argv[1:]] or list(range(16))
units = [buf[i * 131072:(i + 1) * 131072].tobytes() for i in idx]
ubuf, off = corpora.pack_units(units)
d = torch.from_numpy(ubuf).cuda()
for rep in range(2):
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(4))
    blocks = enc.DebugParseDevice(d.data_ptr(), off)
    for ui, u in enumerate(units):
        rseqs, rlits = oracle_lib.zstd_parse_unit(u, level=4)[0]
        gseqs, gextra = blocks[ui]
        L = oracle_lib.lib()
        import ctypes as C
        L.kco_shannon_entropy_bits.restype = C.c_int64
        L.kco_shannon_entropy_bits.argtypes = [C.c_char_p, C.c_uint64]
        bpb_ref = max(L.kco_shannon_entropy_bits(u, len(u)) * 1024 // len(u), 1024)
        bpb_gpu = int(enc.last_block_flags[ui]) >> 8
        if bpb_ref != bpb_gpu:
            print("run", rep, "unit", idx[ui], "bitsPerByte gpu", bpb_gpu, "oracle", bpb_ref, flush=True)
        k = min(len(gseqs), len(rseqs))
        neq = np.nonzero((gseqs[:k] != rseqs[:k]).any(axis=1))[0]
        if len(neq) or len(gseqs) != len(rseqs):
            j = int(neq[0]) if len(neq) else k
            pos = int(rseqs[:j, 0].sum() + rseqs[:j, 1].sum() + 3 * j)
            print("run", rep, "unit", idx[ui], "nseq", len(gseqs), len(rseqs), "first diff at seq", j, "pos", pos, "gpu", gseqs[j:j + 3].tolist(), "oracle", rseqs[j:j + 3].tolist(), flush=True)
    enc.Close()
print("done")
