"""Two contexts alternating kc_zstd_encode_units_submit / kc_wait over 4 GiB host batches (C2 workload): end-to-end GB/s of the
pipelined form against one context back to back.  Run on the GPU box from the repo root."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from compress_amd import zstd, _lib
n, usz = 32768, 131072
buf = _lib.corpus_fill("T", 0x5EED0001, 0, n, usz)
off = (np.arange(n + 1, dtype=np.uint64) * usz)
encs = [zstd.NewWriter(None, zstd.WithEncoderLevel(1)) for _ in range(2)]
cap = n * ((encs[0].MaxEncodedSize(usz) + 15) & ~15) + 64
dsts = [np.zeros(cap, dtype=np.uint8) for _ in range(2)]
for e, d in zip(encs, dsts):
    e.EncodeUnitsSubmit(buf, off, d); e.Wait()   # warm: buffers, pinned slots
K = 6
t0 = time.perf_counter()
encs[0].EncodeUnitsSubmit(buf, off, dsts[0])
for k in range(1, K + 1):
    if k < K:
        encs[k & 1].EncodeUnitsSubmit(buf, off, dsts[k & 1])
    encs[(k - 1) & 1].Wait()
dt = time.perf_counter() - t0
print("two contexts, submit/wait, %d x 4 GiB host batches: %.1f ms per batch = %.2f GB/s end to end" % (K, dt / K * 1e3, K * n * usz / dt / 1e9))
t0 = time.perf_counter()
for k in range(3):
    encs[0].EncodeUnitsSubmit(buf, off, dsts[0]); encs[0].Wait()
dt = time.perf_counter() - t0
print("one context, back to back: %.1f ms per batch = %.2f GB/s" % (dt / 3 * 1e3, 3 * n * usz / dt / 1e9))
