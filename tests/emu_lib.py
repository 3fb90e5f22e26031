"""The LDS-table kernels compiled for the CPU wave emulator (tools/hipemu): build + ctypes wrappers.

Test infrastructure only.  The same .hip sources the product compiles with hipcc are compiled here with g++ against
tools/hipemu/hip/hip_runtime.h, which runs every lane of a wave as a fiber and every cross-lane operation as a
rendezvous; lanes run maximally out of lockstep in between, so an unfenced exchange through LDS shows up as a wrong
result.  This is how the device algorithms are checked against the oracle without a GPU (tests/test_emu_lds.py)."""
import ctypes as C
import functools
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
    # several test processes (pytest -n) may get here at once: one builds, under a file lock, into a name of its own and renames
    # the finished file into place; the others wait for the lock and find it up to date
    import fcntl
    with open(OUT + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
            return OUT
        tmp = "%s.%d.tmp" % (OUT, os.getpid())
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", EMU, "-Wall", "-Wno-unused-function",
               "-Wno-unused-variable", "-Wno-unknown-pragmas"] + srcs + ["-o", tmp, "-ldl"]
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, OUT)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return OUT


def lib():
    global _L
    if _L is None:
        _L = C.CDLL(build())
        _L.kcemu_s2_encode.restype = C.c_int
        _L.kcemu_s2_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _L.kcemu_collectives.restype = C.c_uint64
        _L.kcemu_s2_best.restype = C.c_int
        _L.kcemu_s2_best.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(_L, "kcemu_zfast_parse"):
            _L.kcemu_zfast_parse.restype = C.c_int
            _L.kcemu_zfast_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    return _L


PROCS = int(os.environ.get("KC_EMU_PROCS", str(min(8, os.cpu_count() or 1))))


def _fanned(min_bytes=200000, procs=None, serial_if=None):
    """The units / blocks of one emulated launch are independent work items (one wave each, or one 8-/16-lane group of a wave):
    run contiguous chunks of the list in forked children, one emulator each, and join the results in order — wall-clock time of the
    CPU suite only; KC_EMU_PROCS=1 runs everything in this process.  The wrapped function returns a list (per item, or per block
    in item order), or a tuple of such a list and integer flags (joined with max)."""
    def deco(fn):
        @functools.wraps(fn)
        def run(items, *a, **kw):
            np_ = min(procs or PROCS, PROCS, len(items))
            total = sum(len(x) for x in items)
            if np_ <= 1 or total < min_bytes or (serial_if is not None and serial_if(kw)):
                return fn(items, *a, **kw)
            # contiguous chunks of about equal bytes
            chunks, cur, acc, goal = [], [], 0, total / np_
            for x in items:
                cur.append(x)
                acc += len(x)
                if acc >= goal * (len(chunks) + 1) and len(chunks) < np_ - 1:
                    chunks.append(cur)
                    cur = []
            if cur:
                chunks.append(cur)
            lib()  # built and loaded before the fork
            import multiprocessing as mp
            ctx = mp.get_context("fork")  # (the children inherit fn, its arguments and the loaded emulator: nothing is pickled on the way in)

            def child(conn, chunk):
                try:
                    conn.send(("ok", fn(chunk, *a, **kw)))
                except BaseException as e:
                    conn.send(("err", repr(e)))
                finally:
                    conn.close()

            kids = []
            for c in chunks:
                pr, pw = ctx.Pipe(duplex=False)
                p = ctx.Process(target=child, args=(pw, c), daemon=True)
                p.start()
                pw.close()
                kids.append((p, pr))
            parts, failed = [], False
            for p, pr in kids:
                try:
                    tag, val = pr.recv() if pr.poll(900) else ("err", "timeout")
                except (EOFError, OSError):
                    tag, val = "err", "child died"  # e.g. the emulator's abort on a divergent collective
                failed = failed or tag != "ok"
                parts.append(val)
                p.join(5)
                if p.is_alive():
                    p.kill()
            if failed:
                return fn(items, *a, **kw)  # in this process, where the assertion / the emulator's message is seen
            if isinstance(parts[0], tuple):
                out = [y for p in parts for y in p[0]]
                return (out,) + tuple(max(p[k] for p in parts) for k in range(1, len(parts[0])))
            return [y for p in parts for y in p]
        return run
    return deco


def s2_max_encoded_len(n):
    # s2/encode.go:389-418 (the product's kc_s2_max_encoded_len restates the same)
    n = int(n)
    x = n + 5 + 3 * ((n + 65535) // 65536)  # generous: only sizes the staging slots
    return x + 16


@_fanned()
def s2_encode_blocks(blocks, level=0, framed=False, spec_w0=8, variant=0, stored_only=False):
    """blocks: list of bytes -> list of bytes (uvarint + body, or the framed chunk).  variant 1: KC_S2_VARIANT_AMD64."""
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
    r = lib().kcemu_s2_encode(level | (variant << 8) | (int(stored_only) << 16), int(framed), spec_w0, src.ctypes.data, off.ctypes.data, n, stage.ctypes.data, soff.ctypes.data, sizes.ctypes.data)
    assert r == 0
    return [stage[int(soff[i]):int(soff[i]) + int(sizes[i])].tobytes() for i in range(n)]


@_fanned()
def s2_encode_blocks_hbm(blocks, level=0, framed=False, variant=0, w0=2, w0b=4, grow=1):
    """kc_s2_encode_kernel<level> (HBM tables, 8 lanes per block) over `blocks`: level 0 s2.Encode, 1 EncodeBetter, 2 EncodeSnappy,
    3 EncodeSnappyBetter; variant 1: the amd64 assembly bytes; w0 / w0b / grow: the speculation policy (the library's defaults)."""
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
    L = lib()
    L.kcemu_s2_hbm.restype = C.c_int
    L.kcemu_s2_hbm.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    r = L.kcemu_s2_hbm(level | (variant << 8), int(framed), w0, w0b, grow, src.ctypes.data, off.ctypes.data, n, stage.ctypes.data, soff.ctypes.data,
                       sizes.ctypes.data)
    assert r == 0
    return [stage[int(soff[i]):int(soff[i]) + int(sizes[i])].tobytes() for i in range(n)]


@_fanned()
def zfast_parse(units, block_size=65536, window=4 << 20, spec_w0=8, stream_mode=0):
    """kc_zfast_match_lds_kernel over `units` (list of bytes).  Returns, per block in unit order, (seqs [n,3] u32 as
    (litLen, matchLen-3, offset code), nlit, extra_lits, flags)."""
    n = len(units)
    off = np.zeros(n + 1, dtype=np.uint64)
    blk0 = np.zeros(n + 1, dtype=np.uint32)
    for i, u in enumerate(units):
        off[i + 1] = off[i] + len(u)
        blk0[i + 1] = blk0[i] + (len(u) + block_size - 1) // block_size
    nb = int(blk0[n])
    src = np.frombuffer(b"".join(units) + b"\0" * 64, dtype=np.uint8).copy()
    # the kernels load whole 16-byte granules around readable bytes: keep the buffer 16-byte aligned like a device allocation
    al = np.zeros(len(src) + 32, dtype=np.uint8)
    o = (-al.ctypes.data) % 16
    al[o:o + len(src)] = src
    base = al.ctypes.data + o
    stride = block_size // 4 + 8
    seqs = np.zeros(max(nb, 1) * stride, dtype=np.uint64)
    meta = np.zeros(max(nb, 1) * 8, dtype=np.uint32)
    maxlen = max([len(u) for u in units] + [16])
    pb = 1
    while (1 << pb) <= maxlen + 2:
        pb += 1
    r = lib().kcemu_zfast_parse(base, off.ctypes.data, n, block_size, window, 0, 1, 4, stream_mode, None, seqs.ctypes.data, meta.ctypes.data, stride,
                                blk0.ctypes.data, spec_w0, pb)
    assert r == 0
    out = []
    for b in range(nb):
        m = meta[8 * b:8 * b + 8]
        ns = int(m[0])
        v = seqs[b * stride:b * stride + ns]
        tri = np.stack([(v >> np.uint64(44)).astype(np.uint32), ((v >> np.uint64(24)) & np.uint64(0xFFFFF)).astype(np.uint32),
                        (v & np.uint64(0xFFFFFF)).astype(np.uint32)], axis=1) if ns else np.zeros((0, 3), dtype=np.uint32)
        out.append((tri, int(m[1]), int(m[2]), int(m[3])))
    return out


_grp_tables = {}


def zfast_parse_grp(units, block_size=65536, window=4 << 20, spec_w0=1, spec_grow=1, stream_mode=0, epoch=0, xseg_k=0, slots=None, fresh=False):
    """kc_zfast_match_grp_kernel<8> (HBM tables, 8 lanes per unit) over `units`.  epoch 0: zeroed tables; else the tables of the
    previous calls are kept (as the context keeps its arena between batches) and `epoch` is this launch's stamp."""
    n = len(units)
    off = np.zeros(n + 1, dtype=np.uint64)
    blk0 = np.zeros(n + 1, dtype=np.uint32)
    for i, u in enumerate(units):
        off[i + 1] = off[i] + len(u)
        blk0[i + 1] = blk0[i] + (len(u) + block_size - 1) // block_size
    nb = int(blk0[n])
    src = np.frombuffer(b"".join(units) + b"\0" * 64, dtype=np.uint8).copy()
    al = np.zeros(len(src) + 32, dtype=np.uint8)
    o = (-al.ctypes.data) % 16
    al[o:o + len(src)] = src
    base = al.ctypes.data + o
    stride = block_size // 4 + 8
    seqs = np.zeros(max(nb, 1) * stride, dtype=np.uint64)
    meta = np.zeros(max(nb, 1) * 8, dtype=np.uint32)
    maxlen = max([len(u) for u in units] + [16])
    pb = 1
    while (1 << pb) <= maxlen + 2:
        pb += 1
    if slots is not None:
        pb = slots
    nslot = (n + 7) // 8 * 8
    key = "t"
    if epoch == 0 or fresh or key not in _grp_tables or len(_grp_tables[key]) < nslot << 15:  # (the host clears the arena when it changes hands)
        _grp_tables[key] = np.zeros(nslot << 15, dtype=np.uint32)
    tab = _grp_tables[key]
    L = lib()
    L.kcemu_zfast_parse_grp.restype = C.c_int
    L.kcemu_zfast_parse_grp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int]
    r = L.kcemu_zfast_parse_grp(base, off.ctypes.data, n, block_size, window, stream_mode, seqs.ctypes.data, meta.ctypes.data, stride, blk0.ctypes.data,
                                spec_w0, spec_grow, pb, tab.ctypes.data, epoch, xseg_k)
    assert r == 0
    out = []
    for b in range(nb):
        m = meta[8 * b:8 * b + 8]
        ns = int(m[0])
        v = seqs[b * stride:b * stride + ns]
        tri = np.stack([(v >> np.uint64(44)).astype(np.uint32), ((v >> np.uint64(24)) & np.uint64(0xFFFFF)).astype(np.uint32),
                        (v & np.uint64(0xFFFFFF)).astype(np.uint32)], axis=1) if ns else np.zeros((0, 3), dtype=np.uint32)
        out.append((tri, int(m[1]), int(m[2]), int(m[3]) & 0xFF))
    return out


_zbest_state = {}


def zbest_parse(units, block_size=131072, window=8 << 20, stream_mode=0, n_slots=2, hist=b"", jobs=False, rep=(1, 4, 8), fresh=False):
    """kc_zbest_match_kernel (SpeedBestCompression) over `units` (list of bytes), each preceded by the history `hist` (a dictionary's
    content, or with jobs=True a job's overlap prefix).  The table slots persist between calls like the context's (fresh=True
    starts from zeroed ones).  Returns, per block in unit order, (seqs [n,3] u32, nlit, extra_lits, flags)."""
    import oracle_lib
    n = len(units)
    h = len(hist)
    off = np.zeros(n + 1, dtype=np.uint64)
    blk0 = np.zeros(n + 1, dtype=np.uint32)
    for i, u in enumerate(units):
        off[i + 1] = off[i] + h + len(u)
        blk0[i + 1] = blk0[i] + (len(u) + block_size - 1) // block_size
    nb = int(blk0[n])
    src = np.frombuffer(b"".join(hist + u for u in units) + b"\0" * 64, dtype=np.uint8).copy()
    al = np.zeros(len(src) + 32, dtype=np.uint8)
    o = (-al.ctypes.data) % 16
    al[o:o + len(src)] = src
    base = al.ctypes.data + o
    stride = block_size // 4 + 8
    seqs = np.zeros(max(nb, 1) * stride, dtype=np.uint64)
    meta = np.zeros(max(nb, 1) * 8, dtype=np.uint32)
    st = _zbest_state
    if fresh or st.get("n_slots") != n_slots:
        st["n_slots"] = n_slots
        st["tables"] = np.zeros(n_slots * ((1 << 22) + (1 << 18)), dtype=np.uint64)
        st["cur"] = np.zeros(n_slots, dtype=np.uint32)
        cost = np.zeros(96, dtype=np.int32)
        oracle_lib.lib().kco_zstd_best_costs(C.c_void_p(cost.ctypes.data))
        st["cost"] = cost
    uh = np.full(n, h, dtype=np.uint32)
    jf = np.zeros(n, dtype=np.uint32)
    L = lib()
    L.kcemu_zbest_parse.restype = C.c_int
    L.kcemu_zbest_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    r = L.kcemu_zbest_parse(base, off.ctypes.data, n, block_size, window, 0 if jobs else h, rep[0], rep[1], rep[2], stream_mode, seqs.ctypes.data,
                            meta.ctypes.data, stride, blk0.ctypes.data, uh.ctypes.data if jobs else None, jf.ctypes.data if jobs else None,
                            st["tables"].ctypes.data, st["cur"].ctypes.data, st["cost"].ctypes.data, n_slots)
    assert r == 0
    out = []
    for b in range(nb):
        m = meta[8 * b:8 * b + 8]
        ns = int(m[0])
        v = seqs[b * stride:b * stride + ns]
        tri = np.stack([(v >> np.uint64(44)).astype(np.uint32), ((v >> np.uint64(24)) & np.uint64(0xFFFFF)).astype(np.uint32),
                        (v & np.uint64(0xFFFFFF)).astype(np.uint32)], axis=1) if ns else np.zeros((0, 3), dtype=np.uint32)
        out.append((tri, int(m[1]), int(m[2]), int(m[3])))
    return out


def zbest_parse_fresh(units, **kw):
    """zbest_parse from zeroed slots, the unit list in chunks (each chunk passes its units through n_slots slots of its own)."""
    return _zbest_fresh(units, **kw)


@_fanned(procs=4)
def _zbest_fresh(units, **kw):
    return zbest_parse(units, fresh=True, **kw)


@_fanned(min_bytes=100000)
def s2_best_blocks(blocks, snappy=False):
    """kc_s2_best_kernel (s2.EncodeBest / s2.EncodeSnappyBest) over `blocks` (list of bytes) -> list of bytes (uvarint + body)."""
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
    out = []
    tab = np.zeros((4 << 20 | 512 << 10) // 4, dtype=np.uint32)
    for i in range(n):  # one block per launch: the table arena stays at 4.5 MiB
        tab[:] = 0
        o2 = np.array([off[i], off[i + 1]], dtype=np.uint64) - off[i]
        s2 = np.array([0, soff[i + 1] - soff[i]], dtype=np.uint64)
        r = lib().kcemu_s2_best(5 if snappy else 4, src.ctypes.data + int(off[i]), o2.ctypes.data, 1, stage.ctypes.data + int(soff[i]), s2.ctypes.data,
                                sizes[i:].ctypes.data, tab.ctypes.data)
        assert r == 0
        out.append(stage[int(soff[i]):int(soff[i]) + int(sizes[i])].tobytes())
    return out


def xxh_fin(units, raw_flags, block_size, out_positions, mode, frame_header=9):
    """kc_xxh64_fin_kernel on frames laid out as the entropy stage leaves them: per unit a frame header of `frame_header` bytes, per
    block a 3-byte header + the payload, 4 bytes of checksum.  raw_flags[i]: the unit's frame is raw blocks only (its payloads are
    copied by the kernel).  out_positions[i]: where the frame starts in dst.  Returns (dst, stage, stage_off, sizes, xxh)."""
    n = len(units)
    off = np.zeros(n + 1, dtype=np.uint64)
    blk0 = np.zeros(n + 1, dtype=np.uint32)
    sizes = np.zeros(n, dtype=np.uint32)
    soff = np.zeros(n + 1, dtype=np.uint64)
    raw = []
    for i, u in enumerate(units):
        off[i + 1] = off[i] + len(u)
        nb = (len(u) + block_size - 1) // block_size
        blk0[i + 1] = blk0[i] + nb
        pos = frame_header
        for b in range(nb):
            sz = min(block_size, len(u) - b * block_size)
            raw.append((pos + 3, b * block_size, sz, 0) if raw_flags[i] else (0, 0, 0, 0))
            pos += 3 + sz
        sizes[i] = pos + 4 if len(u) else 0
        soff[i + 1] = soff[i] + ((int(sizes[i]) + 15) & ~15) + 16
    rawdef = np.array(raw if raw else [(0, 0, 0, 0)], dtype=np.uint32).reshape(-1, 4)
    rawdef = np.ascontiguousarray(np.vstack([rawdef, np.zeros((1, 4), dtype=np.uint32)]))
    src = np.frombuffer(b"".join(units) + b"\0" * 64, dtype=np.uint8).copy()
    stage = np.full(int(soff[n]) + 64, 0x55, dtype=np.uint8)
    oo = np.ascontiguousarray(list(out_positions) + [0], dtype=np.uint64)
    dst = np.full(int(max(int(oo[i]) + int(sizes[i]) for i in range(n))) + 64, 0xAA, dtype=np.uint8)
    flags = np.ascontiguousarray([1 if f else 0 for f in raw_flags], dtype=np.uint32)
    xxh = np.zeros(n, dtype=np.uint64)
    L = lib()
    L.kcemu_xxh_fin.restype = C.c_int
    L.kcemu_xxh_fin.argtypes = [C.c_void_p] * 2 + [C.c_uint32] + [C.c_void_p] * 9 + [C.c_int]
    r = L.kcemu_xxh_fin(src.ctypes.data, off.ctypes.data, n, stage.ctypes.data, soff.ctypes.data, sizes.ctypes.data, oo.ctypes.data, dst.ctypes.data,
                        flags.ctypes.data, rawdef.ctypes.data, blk0.ctypes.data, xxh.ctypes.data, mode)
    assert r == 0
    return dst, stage, soff, sizes, xxh


@_fanned(serial_if=lambda kw: kw.get("fused") is not None)  # (batch_end's sequence is one batch: its counters and the contiguous output)
def zstd_frames(units, block_size=None, window=None, crc=True, single=-1, full_zero=True, stream_mode=0, use_grp=False, tuned=0,
                max_encoded_size=None, level=1, no_entropy=False, all_lit_entropy=False, fused=None):
    """The device's whole SpeedFastest EncodeAll pipeline on the emulator (checksum, match finder, entropy stage): one frame per unit.
    Returns (list of frames, error flag, re-run flag)."""
    if level != 1:
        use_grp = level  # (selects the level's match finder in kcemu_zstd_frames)
    if window is None:
        window = (4 << 20) if level == 1 else (8 << 20)  # encoder_options.go:254-266
    if block_size is None:
        block_size = min(65536 if level == 1 else 131072, window)
    n = len(units)
    off = np.zeros(n + 1, dtype=np.uint64)
    soff = np.zeros(n + 1, dtype=np.uint64)
    for i, u in enumerate(units):
        off[i + 1] = off[i] + len(u)
        mes = max_encoded_size(len(u)) if max_encoded_size else len(u) + 64 + 3 * (len(u) // block_size + 2)
        soff[i + 1] = soff[i] + ((mes + 15) & ~15)
    src = np.frombuffer(b"".join(units) + b"\0" * 64, dtype=np.uint8).copy()
    al = np.zeros(len(src) + 32, dtype=np.uint8)
    o = (-al.ctypes.data) % 16
    al[o:o + len(src)] = src
    stage = np.zeros(int(soff[n]) + 64, dtype=np.uint8)
    sizes = np.zeros(n + 1, dtype=np.uint32)
    err = np.zeros(4, dtype=np.uint32)
    dst = np.full(int(soff[n]) + 64, 0xAA, dtype=np.uint8) if fused is not None else None
    doff = np.zeros(n + 2, dtype=np.uint64)
    L = lib()
    L.kcemu_zstd_frames.restype = C.c_int
    L.kcemu_zstd_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_int] * 9 + [C.c_void_p] * 6 + [C.c_int]
    r = L.kcemu_zstd_frames(al.ctypes.data + o, off.ctypes.data, n, block_size, window, int(crc), single, int(full_zero), stream_mode, int(use_grp), tuned,
                            int(no_entropy) | (int(all_lit_entropy) << 1), stage.ctypes.data, soff.ctypes.data, sizes.ctypes.data, err.ctypes.data,
                            dst.ctypes.data if dst is not None else None, doff.ctypes.data, int(fused or 0))
    assert r == 0
    if fused is not None:  # batch_end's sequence: size scan, checksum (+ raw payloads), compaction -> the contiguous output
        assert int(doff[n]) == int(sizes[:n].sum()) and np.all(dst[int(doff[n]):] == 0xAA)
        if tuned & 0x100:  # (with the pre-scan: how many units it settled)
            return [dst[int(doff[i]):int(doff[i + 1])].tobytes() for i in range(n)], int(err[0]), int(err[1]), int(err[2]), int(err[3])
        return [dst[int(doff[i]):int(doff[i + 1])].tobytes() for i in range(n)], int(err[0]), int(err[1]), int(err[2])
    return [stage[int(soff[i]):int(soff[i]) + int(sizes[i])].tobytes() for i in range(n)], int(err[0]), int(err[1])


def zstd_prime(level, prefixes, pos_bits, reverse=False, unit_list=None):
    """kc_zstd_prime_kernel: one table slot per entry of unit_list (default: every unit), primed from the unit's prefix.  Returns the
    slots as a uint32 array [n_slots, table_words]."""
    tb = {1: 4 << 15, 2: (4 << 17) + (4 << 15), 3: (8 << 19) + (4 << 13)}[level]
    n = len(prefixes)
    off = np.zeros(n + 1, dtype=np.uint64)
    for i, b in enumerate(prefixes):
        off[i + 1] = off[i] + len(b) + 16  # (a unit is prefix + job bytes: 16 bytes of "job" behind each prefix)
    src = np.zeros(int(off[n]) + 64, dtype=np.uint8)
    for i, b in enumerate(prefixes):
        src[int(off[i]):int(off[i]) + len(b)] = np.frombuffer(b, dtype=np.uint8)
        src[int(off[i]) + len(b):int(off[i + 1])] = 0xA5
    hist = np.array([len(b) for b in prefixes], dtype=np.uint32)
    ul = None if unit_list is None else np.array(unit_list, dtype=np.uint32)
    ns = n if ul is None else len(ul)
    tabs = np.zeros((ns, tb // 4), dtype=np.uint32)
    L = lib()
    L.kcemu_zstd_prime.restype = C.c_int
    L.kcemu_zstd_prime.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    r = L.kcemu_zstd_prime(level, pos_bits, int(reverse), src.ctypes.data, off.ctypes.data, hist.ctypes.data, None if ul is None else ul.ctypes.data,
                           ns, tabs.ctypes.data, tb)
    assert r == 0
    return tabs
