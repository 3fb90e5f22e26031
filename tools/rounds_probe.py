import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from compress_amd import _lib, zstd
for kind in "TJM":
    n, usz = 64, 131072
    buf = _lib.corpus_fill(kind, 0x5EED0001, 0, n, usz)
    d = torch.from_numpy(buf).cuda()
    enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1))
    off = np.arange(n + 1, dtype=np.uint64) * usz
    blocks = enc.DebugParseDevice(d.data_ptr(), off)
    rounds = enc.last_block_flags >> 8
    nseq = np.array([len(b[0]) for b in blocks])
    print(kind, "blocks", len(blocks), "avg seqs/block %.0f" % nseq.mean(), "avg rounds/block %.0f" % rounds.mean(), "rounds/seq %.2f" % (rounds.sum() / nseq.sum()))
