"""Host mirror of the reference's S2 block API (s2/encode.go) over include/kcgpu.h."""
import ctypes as C

from . import _lib


def MaxEncodedLen(src_len):
    """s2.MaxEncodedLen (s2/encode.go:389)."""
    return int(_lib.load().kc_s2_max_encoded_len(int(src_len)))


class BlockEncoder:
    def __init__(self, device=0, stream=None):
        self._ctx = _lib.Context(device, stream)

    def EncodeBlocks(self, src, blk_off):
        """N x s2.Encode(nil, block).  Returns (numpy uint8, uint64[n+1] offsets)."""
        import numpy as np
        ctx = self._ctx
        src = np.ascontiguousarray(src, dtype=np.uint8)
        blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
        n = len(blk_off) - 1
        cap = sum(((MaxEncodedLen(int(blk_off[i + 1] - blk_off[i])) + 15) & ~15) for i in range(n)) + 64
        dst = np.empty(cap, dtype=np.uint8)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        ctx.check(ctx.L.kc_s2_encode_blocks(ctx.h, src.ctypes.data, blk_off.ctypes.data, n, dst.ctypes.data, cap, out_off.ctypes.data))
        return dst[:int(out_off[n])], out_off

    def EncodeBlocksDevice(self, d_src_ptr, blk_off, d_dst_ptr, dst_cap):
        import numpy as np
        ctx = self._ctx
        blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
        n = len(blk_off) - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        ctx.check(ctx.L.kc_s2_encode_blocks_dev(ctx.h, d_src_ptr, blk_off.ctypes.data, n, d_dst_ptr, int(dst_cap), out_off.ctypes.data))
        return out_off

    def EncodeStreamDevice(self, d_src_ptr, blk_off, d_dst_ptr, dst_cap, with_stream_id=True):
        """s2.Writer framing of the blocks (stream identifier + chunks).  Returns uint64[n+1] chunk offsets."""
        import numpy as np
        ctx = self._ctx
        blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
        n = len(blk_off) - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        ctx.check(ctx.L.kc_s2_encode_stream_dev(ctx.h, d_src_ptr, blk_off.ctypes.data, n, d_dst_ptr, int(dst_cap), out_off.ctypes.data,
                                                int(with_stream_id)))
        return out_off

    def Encode(self, dst, src):
        """s2.Encode(dst, src) (s2/encode.go:29): uvarint length + block body."""
        import numpy as np
        src = bytes(src)
        out, _ = self.EncodeBlocks(np.frombuffer(src, dtype=np.uint8), np.array([0, len(src)], dtype=np.uint64))
        return out.tobytes()

    def CustomEncoder(self):
        """The function to hand to s2.WriterCustomEncoder (s2/writer.go:1053): fn(dst, src) -> int."""
        ctx = self._ctx

        def fn(dst, src):
            src = bytes(src)
            r = ctx.L.kc_s2_encode_block(ctx.h, (C.c_char * len(dst)).from_buffer(dst), len(dst), src, len(src))
            return int(r)
        return fn

    def Close(self):
        self._ctx.close()
