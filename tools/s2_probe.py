import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from compress_amd import _lib, s2
bsz = 65536; nb = 32768
buf = _lib.corpus_fill("J", 0x5EED0003, 0, nb, bsz); d = torch.from_numpy(buf).cuda(); boff = np.arange(nb + 1, dtype=np.uint64) * bsz
e2 = s2.BlockEncoder(); cap = nb * ((s2.MaxEncodedLen(bsz) + 15) & ~15) + 64; dd = torch.empty(cap, dtype=torch.uint8, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); oo = e2.EncodeBlocksDevice(d.data_ptr(), boff, dd.data_ptr(), cap); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("S2 J 2 GiB 64KiB blocks: %.1f ms %.2f GB/s ratio %.4f" % (dt * 1e3, nb * bsz / dt / 1e9, int(oo[nb]) / (nb * bsz)), e2._ctx.timings())
d_out = torch.empty(nb * bsz + 64, dtype=torch.uint8, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); st = e2.DecodeBlocksDevice(dd.data_ptr(), oo, d_out.data_ptr(), boff); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("S2 decode J 2 GiB: %.1f ms %.2f GB/s (of decoded bytes) ok=%s equal=%s" % (dt * 1e3, nb * bsz / dt / 1e9, not st.any(), bool(torch.equal(d_out[:nb * bsz], d))), e2._ctx.timings()["total_ms"])
