"""Host mirror of the reference's S2 block API (s2/encode.go) over include/kcgpu.h."""
import ctypes as C
import threading

from . import _lib


def MaxEncodedLen(src_len):
    """s2.MaxEncodedLen (s2/encode.go:389)."""
    return int(_lib.load().kc_s2_max_encoded_len(int(src_len)))


LevelDefault, LevelBetter, LevelSnappy, LevelSnappyBetter = 0, 1, 2, 3  # s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter (s2/encode.go:29, 117, 204, 248)
LevelBest, LevelSnappyBest = 4, 5  # s2.EncodeBest / EncodeSnappyBest (s2/encode.go:146, 278)
LevelUncompressed = 6              # s2.WriterUncompressed (writer.go:951): a level of the stream writer only — every block an uncompressed chunk


class BlockEncoder:
    def __init__(self, device=0, stream=None, level=LevelDefault, path=None, variant=None):
        """path: None / 'auto' (by blocks in flight), 'hbm' or 'lds' — the kernel family of s2.Encode / s2.EncodeSnappy
        (KC_OPT_MATCH_PATH, include/kcgpu.h); both give the reference's bytes.
        variant: None / 'go' — the bytes of the reference's portable Go block encoders (arm64, noasm builds); 'amd64' — the bytes
        of its amd64 assembly encoders (KC_OPT_S2_VARIANT; s2.Encode and s2.EncodeSnappy)."""
        self._ctx = _lib.Context(device, stream)
        if path is not None:
            self._ctx.set_path(path)
        if variant not in (None, "go", "amd64"):
            raise ValueError("variant must be 'go' or 'amd64'")
        if variant == "amd64":
            self._ctx.set_option(20, 1)
        self.level = int(level)
        # the batched calls use the context's stream and scratch: one at a time per encoder (the Go shim's Ctx.mu); the
        # CustomEncoder hook does not take it — the library batches its concurrent callers itself
        self._mu = threading.Lock()

    def ctx(self):
        return self._ctx

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
        with self._mu:
            ctx.check(ctx.L.kc_s2_encode_blocks_lvl(ctx.h, self.level, src.ctypes.data, blk_off.ctypes.data, n, dst.ctypes.data, cap, out_off.ctypes.data))
        return dst[:int(out_off[n])], out_off

    def EncodeBlocksDevice(self, d_src_ptr, blk_off, d_dst_ptr, dst_cap):
        import numpy as np
        ctx = self._ctx
        blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
        n = len(blk_off) - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        with self._mu:
            ctx.check(ctx.L.kc_s2_encode_blocks_lvl_dev(ctx.h, self.level, d_src_ptr, blk_off.ctypes.data, n, d_dst_ptr, int(dst_cap), out_off.ctypes.data))
        return out_off

    def EncodeBlocksDeviceBegin(self, d_src_ptr, blk_off):
        """First half of EncodeBlocksDevice (one device batch, bare blocks): enqueue up to and including the encoder kernel, no wait
        (kc_s2_encode_blocks_lvl_dev_begin)."""
        import numpy as np
        ctx = self._ctx
        blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
        self._pending_n = len(blk_off) - 1
        with self._mu:
            ctx.check(ctx.L.kc_s2_encode_blocks_lvl_dev_begin(ctx.h, self.level, d_src_ptr, blk_off.ctypes.data, self._pending_n))

    def EncodeBlocksDeviceEnd(self, d_dst_ptr, dst_cap):
        """Second half: the blocks are compacted to d_dst_ptr (named only now: the parts of one batch run as several launches, each part
        right behind the previous one), offsets relative to it (uint64[n+1]), wait."""
        import numpy as np
        ctx = self._ctx
        out_off = np.zeros(self._pending_n + 1, dtype=np.uint64)
        with self._mu:
            ctx.check(ctx.L.kc_s2_encode_blocks_lvl_dev_end_at(ctx.h, d_dst_ptr, int(dst_cap), out_off.ctypes.data))
        return out_off

    def EncodeStreamDevice(self, d_src_ptr, blk_off, d_dst_ptr, dst_cap, with_stream_id=True):
        """s2.Writer framing of the blocks (stream identifier + chunks).  Returns uint64[n+1] chunk offsets."""
        import numpy as np
        ctx = self._ctx
        blk_off = np.ascontiguousarray(blk_off, dtype=np.uint64)
        n = len(blk_off) - 1
        out_off = np.zeros(n + 1, dtype=np.uint64)
        with self._mu:
            ctx.check(ctx.L.kc_s2_encode_stream_lvl_dev(ctx.h, self.level, d_src_ptr, blk_off.ctypes.data, n, d_dst_ptr, int(dst_cap), out_off.ctypes.data,
                                                        int(with_stream_id)))
        return out_off

    def DecodeBlocksDevice(self, d_enc_ptr, enc_off, d_dst_ptr, dst_off):
        """N x s2.Decode on the device (verifier): returns uint32[n] status, 0 = block decoded to exactly its stated size."""
        import numpy as np
        ctx = self._ctx
        enc_off = np.ascontiguousarray(enc_off, dtype=np.uint64)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
        n = len(enc_off) - 1
        status = np.zeros(max(n, 1), dtype=np.uint32)
        with self._mu:
            ctx.check(ctx.L.kc_s2_decode_blocks_dev(ctx.h, d_enc_ptr, enc_off.ctypes.data, n, d_dst_ptr, dst_off.ctypes.data, status.ctypes.data))
        return status[:n]

    def Encode(self, dst, src):
        """s2.Encode(dst, src) (s2/encode.go:29) — or s2.EncodeBetter (:117) for a LevelBetter encoder: uvarint length + block body."""
        import numpy as np
        src = bytes(src)
        out, _ = self.EncodeBlocks(np.frombuffer(src, dtype=np.uint8), np.array([0, len(src)], dtype=np.uint64))
        return out.tobytes()

    def CustomEncoder(self, host_first=0):
        """The function to hand to s2.WriterCustomEncoder (s2/writer.go:1053): fn(dst, src) -> int.  The hook (kc_s2_encode_block)
        encodes at the default level, like the reference's built-in encodeBlock of a default Writer.
        host_first: KC_OPT_S2_HOOK_HOST_FIRST — how many callers at a time the hook leaves to the caller's built-in encoder (fn returns
        -1 for them) before the overflow goes to the device.  This Python façade has no built-in encoder, so its default is 0 (every
        caller to the device); the library's own default (None here: the CPUs of the process) is what the Go shim gets."""
        if self.level != LevelDefault:
            raise ValueError("the WriterCustomEncoder hook serves the default level only (s2.Encode); use EncodeBlocks for level %d" % self.level)
        ctx = self._ctx
        if host_first is not None:
            ctx.set_option(_lib.OPT_S2_HOOK_HOST_FIRST, int(host_first))

        def fn(dst, src):
            src = bytes(src)
            r = ctx.L.kc_s2_encode_block(ctx.h, (C.c_char * len(dst)).from_buffer(dst), len(dst), src, len(src))
            return int(r)
        return fn

    def HookStats(self):
        """(calls, device batches) served by the CustomEncoder hook of this encoder so far."""
        a, b = C.c_uint64(), C.c_uint64()
        self._ctx.L.kc_s2_hook_stats(self._ctx.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def Close(self):
        self._ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# s2.Writer (s2/writer.go) over the device path: the reference's chunking rules decide WHERE the stream is cut (they
# depend on the Write/Flush/EncodeBuffer call pattern), the GPU encodes and frames the chunks in batches.
# ---------------------------------------------------------------------------------------------------------------------
_MIN_BLOCK, _MAX_BLOCK, _DEFAULT_BLOCK = 4 << 10, 4 << 20, 1 << 20  # s2/encode.go minBlockSize/maxBlockSize, writer.go defaultBlockSize
_MAGIC = b"\xff\x06\x00\x00S2sTwO"                                   # magicChunk (s2/s2.go)
_MAGIC_SNAPPY = b"\xff\x06\x00\x00sNaPpY"                            # magicChunkSnappy (s2/s2.go:79-80)


def WriterBlockSize(n):
    def apply(w):
        if n > _MAX_BLOCK or n < _MIN_BLOCK:
            raise ValueError("s2: block size too large. Must be <= 4MB and >=4KB")  # writer.go:985
        w.blockSize = int(n)
    return apply


def WriterConcurrency(n):
    def apply(w):
        if n <= 0:
            raise ValueError("concurrency must be at least 1")  # writer.go:911
        w.concurrency = int(n)  # no effect on bytes: the device batches chunks regardless
    return apply


def WriterFlushOnWrite():
    return lambda w: setattr(w, "flushOnWrite", True)


def _unsupported(name):
    def opt(*a, **k):
        def apply(w):
            raise NotImplementedError("s2.%s is not served by the device path; use the reference writer" % name)
        return apply
    return opt


def WriterBetterCompression():
    """s2.WriterBetterCompression (writer.go:931): blocks are encoded with encodeBlockBetter."""
    return lambda w: setattr(w, "level", LevelBetter)


def WriterBestCompression():
    """s2.WriterBestCompression (writer.go:945): blocks are encoded with encodeBlockBest."""
    return lambda w: setattr(w, "level", LevelBest)


def WriterSnappyCompat():
    """s2.WriterSnappyCompat (writer.go:1025-1037): Snappy-compatible output — the blocks through encodeBlockSnappy /
    encodeBlockBetterSnappy / encodeBlockBestSnappy (no repeat codes), the "sNaPpY" stream identifier, blocks of at most 64 KiB - 8."""
    def apply(w):
        w.snappy = True
        if w.blockSize > (64 << 10):
            w.blockSize = (64 << 10) - 8
    return apply


def WriterUncompressed():
    """s2.WriterUncompressed (writer.go:948-956): bypass compression — the stream is uncompressed chunks only (type 0x01 | length |
    masked CRC32C | bytes; the checksum and the copy run on the device).  A level like the others: a later level option replaces it."""
    return lambda w: setattr(w, "level", LevelUncompressed)


def WriterAddIndex():
    """s2.WriterAddIndex (writer.go:921): append the seek index to the stream on Close."""
    return lambda w: setattr(w, "appendIndex", True)


def WriterPadding(n):
    """s2.WriterPadding (writer.go:999): pad the output to a multiple of n with a skippable 0xfe chunk on Close."""
    def apply(w):
        if n <= 0:
            raise ValueError("s2: padding must be at least 1")
        if n > _MAX_BLOCK:
            raise ValueError("s2: padding must less than 4MB")
        w.pad = int(n)
    return apply


def WriterPaddingSrc(reader):
    """s2.WriterPaddingSrc (writer.go:1018): where the padding bytes come from (default: os.urandom, like crypto/rand)."""
    return lambda w: setattr(w, "randSrc", reader)


def calc_skippable_frame(written, want_multiple):
    """calcSkippableFrame (writer.go:858): bytes to add so that `written` becomes a multiple; 0 or >= 4."""
    left = written % want_multiple
    if left == 0:
        return 0
    add = want_multiple - left
    while add < 4:
        add += want_multiple
    return add


class Index:
    """The writer side of s2.Index (s2/index.go:17-236): add / reduce / appendTo."""
    MAX_ENTRIES, MIN_DIST = 1 << 16, 1 << 20

    def __init__(self, max_block):
        self.est = int(max_block)
        self.info = []  # [compressedOffset, uncompressedOffset]

    def add(self, comp, unc):
        if self.info:
            latest = self.info[-1]
            if latest[1] == unc:
                latest[0] = comp
                return
            if latest[1] > unc or latest[0] > comp:
                raise ValueError("internal error: earlier offset received")
            if latest[1] + self.MIN_DIST > unc:
                return
        self.info.append([comp, unc])

    def _reduce(self):
        if len(self.info) < self.MAX_ENTRIES and self.est >= self.MIN_DIST:
            return
        remove_n = (len(self.info) + 1) // self.MAX_ENTRIES
        while self.est * (remove_n + 1) < self.MIN_DIST and len(self.info) // (remove_n + 1) > 1000:
            remove_n += 1
        self.info = self.info[::remove_n + 1]
        self.est += self.est * remove_n

    @staticmethod
    def _varint(x):
        ux = (x << 1) ^ (x >> 63) if x >= 0 else ((~x) << 1) | 1
        ux &= (1 << 64) - 1
        out = bytearray()
        while ux >= 0x80:
            out.append((ux & 0x7F) | 0x80)
            ux >>= 7
        out.append(ux)
        return bytes(out)

    def append_to(self, uncomp_total, comp_total):
        self._reduce()
        b = bytearray(b"\x99\x00\x00\x00s2idx\x00")
        for v in (uncomp_total, comp_total, self.est, len(self.info)):
            b += self._varint(v)
        has_unc = 0
        for i, (_, u) in enumerate(self.info):
            if (i == 0 and u != 0) or (i > 0 and u != self.info[i - 1][1] + self.est):
                has_unc = 1
                break
        b.append(has_unc)
        if has_unc:
            for i, (_, u) in enumerate(self.info):
                b += self._varint(u - (self.info[i - 1][1] + self.est) if i else u)
        c_predict = self.est // 2
        for i, (c, _) in enumerate(self.info):
            c_off = c
            if i:
                c_off -= self.info[i - 1][0] + c_predict
                c_predict += c_off // 2 if c_off >= 0 else -((-c_off) // 2)  # Go integer division truncates toward zero
            b += self._varint(c_off)
        b += (len(b) + 4 + 6).to_bytes(4, "little")
        b += b"\x00xdi2s"
        n = len(b) - 4
        b[1:4] = bytes([n & 0xFF, (n >> 8) & 0xFF, (n >> 16) & 0xFF])
        return bytes(b)


class Writer:
    """s2.Writer: Write / ReadFrom / EncodeBuffer / AddSkippableBlock / Flush / Close / Reset with the reference's chunk
    boundaries (writer.go:182-218, 357-453, 483-571, 741-857); default, better or best level (WriterBetterCompression / WriterBestCompression).  Chunks are queued and
    encoded on the GPU in batches of `batch_bytes`; the bytes written equal the reference's for the same call sequence."""

    def __init__(self, w, *opts, device=0, stream=None, batch_bytes=256 << 20, variant=None):
        self.blockSize = _DEFAULT_BLOCK
        self.concurrency = 1
        self.flushOnWrite = False
        self.appendIndex = False
        self.pad = 0
        self.randSrc = None
        self.level = LevelDefault
        self.snappy = False
        for o in opts:
            o(self)
        if self.snappy:  # (*Writer).encodeBlock, writer.go:1053-1091: the Snappy-compatible block encoder of the level
            if self.blockSize > (64 << 10):
                raise ValueError("s2: block size too large. Must be <= 64K and >=4KB on for snappy compatible output")  # writer.go:982
            self.level = {LevelDefault: LevelSnappy, LevelBetter: LevelSnappyBetter, LevelBest: LevelSnappyBest, LevelUncompressed: LevelUncompressed}[self.level]
        self._enc = BlockEncoder(device, stream, level=self.level, variant=variant)  # variant: see BlockEncoder
        self._device = device
        self._batch = int(batch_bytes)
        self.Reset(w)

    def Reset(self, w):
        self.writer = w
        self._ibuf = bytearray()
        self._queue = []       # ("c", bytes) data chunk | ("r", bytes) raw bytes to pass through (skippable blocks)
        self._queued = 0
        self._wroteHeader = False
        self._closed = False
        self.written = 0
        self.uncompWritten = 0
        self._index = Index(self.blockSize)
        self._flushedUncomp = 0  # uncompressed start offset of the next chunk to be written out

    # -- chunk cutting, exactly as writer.go --
    def _write(self, p):  # writer.go:483 write(): everything in p becomes chunks now, the last one may be short
        mv = memoryview(p)
        for i in range(0, len(mv), self.blockSize):
            self._queue.append(("c", bytes(mv[i:i + self.blockSize])))
        self._queued += len(mv)
        self.uncompWritten += len(mv)
        if self._queued >= self._batch:
            self._drain()

    def Write(self, p):
        if self._closed:
            raise IOError("s2: Writer is closed")
        p = bytes(p)
        if self.flushOnWrite:
            self._write(p)
            return len(p)
        n_ret = 0
        while len(p) > self.blockSize - len(self._ibuf):  # writer.go:190
            if len(self._ibuf) == 0:
                self._write(p)  # large write, empty buffer: all of p, including its tail
                n = len(p)
            else:
                n = self.blockSize - len(self._ibuf)
                self._ibuf += p[:n]
                self._write(self._ibuf)
                self._ibuf = bytearray()
            n_ret += n
            p = p[n:]
        self._ibuf += p
        return n_ret + len(p)

    def EncodeBuffer(self, buf):  # writer.go:357
        if self._closed:
            raise IOError("s2: Writer is closed")
        if self.flushOnWrite:
            self._write(bytes(buf))
            return
        if self._ibuf:
            self._async_flush()
        self._write(bytes(buf))

    def ReadFrom(self, r):  # writer.go:220: blockSize reads until EOF
        if self._closed:
            raise IOError("s2: Writer is closed")
        if self._ibuf:
            self._async_flush()
        n = 0
        while True:
            b = r.read(self.blockSize)
            while b and len(b) < self.blockSize:  # io.ReadFull
                more = r.read(self.blockSize - len(b))
                if not more:
                    break
                b += more
            if not b:
                break
            n += len(b)
            self._write(b)
            if len(b) < self.blockSize:
                break
        return n

    def AddSkippableBlock(self, id, data):  # writer.go:272
        data = bytes(data)
        if len(data) == 0:
            return
        if id < 0x80 or id > 0xfe:
            raise ValueError("invalid skippable block id %x" % id)
        if len(data) > 0xFFFFFF:
            raise ValueError("skippable block excessed maximum size")
        self._queue.append(("r", bytes([id, len(data) & 0xFF, (len(data) >> 8) & 0xFF, (len(data) >> 16) & 0xFF]) + data))

    def _async_flush(self):  # writer.go:741
        if self._ibuf:
            b = bytes(self._ibuf)
            self._ibuf = bytearray()
            self._write(b)

    def Flush(self):
        if self._closed:
            return
        self._async_flush()
        self._drain()

    def Close(self):
        """writer.go:787: Flush, then the index chunk if WriterAddIndex was given."""
        self._close_index(self.appendIndex)

    def CloseIndex(self):
        """writer.go:794: Close and return the index (it is also appended to the stream only with WriterAddIndex)."""
        return self._close_index(True)

    def _close_index(self, want):
        if self._closed:
            return None
        self.Flush()
        self._closed = True
        index = None
        if want:  # writer.go:808-818: the compressed total is unknown to the index when padding follows
            index = self._index.append_to(self.uncompWritten, self.written if self.pad <= 1 else -1)
            if self.appendIndex:
                self.written += len(index)
        if self.pad > 1:  # writer.go:820-836: the padding chunk goes out BEFORE the index, sized as if the index were written
            add = calc_skippable_frame(self.written, self.pad)
            if add:
                if add >= _MAX_BLOCK + 4:
                    raise ValueError("s2: requested skippable frame (%d) >= max 1<<24" % add)
                f = add - 4
                fill = self.randSrc.read(f) if self.randSrc is not None else __import__("os").urandom(f)
                if len(fill) != f:
                    raise IOError("short read from the padding source")
                self.writer.write(bytes([0xfe, f & 0xFF, (f >> 8) & 0xFF, (f >> 16) & 0xFF]) + fill)
        if index is not None and self.appendIndex:
            self.writer.write(index)
        return index

    # -- device batch --
    def _drain(self):
        import numpy as np
        import torch
        out = []  # (bytes, uncompressed start offset) per write to the underlying writer, in stream order
        i = 0
        q = self._queue
        while i < len(q):
            if q[i][0] == "r":
                if not self._wroteHeader:  # the stream identifier precedes the first output of any kind
                    out.append((_MAGIC_SNAPPY if self.snappy else _MAGIC, self._flushedUncomp))
                    self._wroteHeader = True
                out.append((q[i][1], self._flushedUncomp))
                i += 1
                continue
            j = i
            while j < len(q) and q[j][0] == "c":
                j += 1
            chunks = [c[1] for c in q[i:j]]
            off = np.zeros(len(chunks) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(c) for c in chunks])
            src = np.frombuffer(b"".join(chunks), dtype=np.uint8)
            d_src = torch.from_numpy(src.copy()).cuda(self._device)
            cap = sum(((MaxEncodedLen(len(c)) + 8 + 15) & ~15) for c in chunks) + 64
            d_dst = torch.empty(cap, dtype=torch.uint8, device=d_src.device)
            with_id = not self._wroteHeader
            oo = self._enc.EncodeStreamDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap, with_stream_id=with_id and not self.snappy)
            self._wroteHeader = True
            blob = d_dst[:int(oo[-1])].cpu().numpy().tobytes()
            if with_id and self.snappy:
                out.append((_MAGIC_SNAPPY, self._flushedUncomp))
            elif with_id:
                out.append((blob[:int(oo[0])], self._flushedUncomp))
            for k, c in enumerate(chunks):
                out.append((blob[int(oo[k]):int(oo[k + 1])], self._flushedUncomp))
                self._flushedUncomp += len(c)
            i = j
        self._queue = []
        self._queued = 0
        for b, start in out:  # the writer goroutine of writer.go:146-170: index entry, then the bytes
            self._index.add(self.written, start)
            self.writer.write(b)
            self.written += len(b)

    def CloseDevice(self):
        self._enc.Close()


def NewWriter(w, *opts, **kw):
    return Writer(w, *opts, **kw)
