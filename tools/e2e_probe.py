"""Host-buffer (PCIe-inclusive) rates of the entry points a cgo caller uses, beside this box's ceilings.  GPU box, repo root.

    python tools/e2e_probe.py [C2|C3|C4|C5 ...] [--steps K] [--trace]

Per configuration: one call (kc_zstd_encode_units / kc_s2_encode_blocks_lvl from pageable memory into pageable memory), then the
steady state of two contexts alternating submit / wait over K batches (the next batch stages while the running one encodes and
drains), against the device-resident rate of the same batch.  First the ceilings: kc_probe_pcie.  One JSON object per line."""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from compress_amd import _lib, zstd, s2  # noqa: E402
import bench  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if a in bench.CONFIGS]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 6
    trace = "--trace" in sys.argv
    only_two = "--only-two" in sys.argv  # skip the single calls (kernel-trace runs of the steady state)
    nctx = int(sys.argv[sys.argv.index("--ctx") + 1]) if "--ctx" in sys.argv else 2  # calls kept in flight in the steady state
    pinned = "--pinned" in sys.argv  # source and destinations in page-locked memory (kc_host_alloc): no staging copies
    configs = args or ["C2"]
    ctx = _lib.Context(0)
    print(json.dumps({"probe_pcie": ctx.probe_pcie(1 << 30)}), flush=True)
    ctx.close()
    for name in configs:
        cfg = bench.CONFIGS[name]
        usz = cfg["unit"]
        n = int(cfg["gib"] * (1 << 30)) // usz
        kind = cfg["kind"]
        buf = _lib.corpus_fill(kind, bench.SEEDS[kind], 0, n, usz)
        if pinned:
            keep = [_lib.PinnedBuffer(buf.size)]
            keep[0].a[:] = buf
            buf = keep[0].a
        off = np.arange(n + 1, dtype=np.uint64) * usz
        is_s2 = cfg["codec"] == "s2"
        if is_s2:
            encs = [s2.BlockEncoder(device=0, variant=cfg.get("variant")) for _ in range(nctx)]
            slot = (s2.MaxEncodedLen(usz) + 15) & ~15
        else:
            zo = [zstd.WithEncoderLevel(cfg["level"])]
            if cfg["dict_kib"]:
                zo.append(zstd.WithEncoderDictRaw(1, _lib.corpus_fill("T", bench.DICT_SEED, 0, 1, cfg["dict_kib"] << 10).tobytes()))
            encs = [zstd.NewWriter(None, *zo, device=0) for _ in range(nctx)]
            slot = (encs[0].MaxEncodedSize(usz) + 15) & ~15
        cap = n * slot + 64
        if pinned:
            keep += [_lib.PinnedBuffer(cap) for _ in range(nctx)]
            dsts = [k.a for k in keep[1:]]
        else:
            dsts = [np.zeros(cap, dtype=np.uint8) for _ in range(nctx)]
        offs = [np.zeros(n + 1, dtype=np.uint64) for _ in range(nctx)]

        def ctx_of(e):
            return e._ctx if is_s2 else e.ctx()

        def call(i, submit):
            c = ctx_of(encs[i])
            L = c.L
            import ctypes as C
            if is_s2:
                f = L.kc_s2_encode_blocks_lvl_submit if submit else L.kc_s2_encode_blocks_lvl
                c.check(f(c.h, 0, buf.ctypes.data, off.ctypes.data, n, dsts[i].ctypes.data, cap, offs[i].ctypes.data))
            else:
                f = L.kc_zstd_encode_units_submit if submit else L.kc_zstd_encode_units
                c.check(f(c.h, C.byref(encs[i].o), buf.ctypes.data, off.ctypes.data, n, dsts[i].ctypes.data, cap, offs[i].ctypes.data))

        def wait(i):
            c = ctx_of(encs[i])
            c.check(c.L.kc_wait(c.h))

        # device-resident rate of the same batch (one context, back to back)
        d_src = torch.from_numpy(buf).cuda()
        d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            if is_s2:
                encs[0].EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
            else:
                encs[0].EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            if is_s2:
                ro = encs[0].EncodeBlocksDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
            else:
                ro = encs[0].EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t0) / 3 * 1e3
        ref = d_dst[:int(ro[n])].cpu().numpy()
        del d_src, d_dst
        torch.cuda.empty_cache()
        for i in range(nctx):  # warm: buffers, pinned slots
            call(i, False)
        if trace:
            ctx_of(encs[0]).set_option(_lib.OPT_HOST_TRACE, 1)
        one = []
        for _ in range(1 if only_two else 3):
            t0 = time.perf_counter()
            call(0, False)
            one.append((time.perf_counter() - t0) * 1e3)
        if trace:
            ctx_of(encs[0]).set_option(_lib.OPT_HOST_TRACE, 0)
        same = bool(np.array_equal(offs[0], ro) and np.array_equal(dsts[0][:int(ro[n])], ref))
        for w in range(2):  # warm: every lane of the engine has held a sub-batch of this shape (scratch allocated) before the clock starts
            for i in range(nctx):
                call(i, True)
            for i in range(nctx):
                wait(i)
        t0 = time.perf_counter()
        sub = 0
        for k in range(steps):  # nctx calls in flight: submit call k + nctx - 1 before waiting for call k
            while sub < steps and sub < k + nctx:
                call(sub % nctx, True)
                sub += 1
            wait(k % nctx)
        two_ms = (time.perf_counter() - t0) / steps * 1e3
        same2 = all(bool(np.array_equal(offs[i], ro) and np.array_equal(dsts[i][:int(ro[n])], ref)) for i in range(nctx))
        gb = n * usz / 1e9
        print(json.dumps({"config": name, "GiB": cfg["gib"], "ratio": round(int(ro[n]) / (n * usz), 4), "device_resident_ms": round(dev_ms, 2),
                          "device_resident_GBps": round(gb / dev_ms * 1e3, 2),
                          "one_call_ms": [round(x, 1) for x in one], "one_call_GBps": round(gb / min(one) * 1e3, 2),
                          "calls_in_flight": nctx, "pinned_buffers": pinned, "two_contexts_ms_per_batch": round(two_ms, 1), "two_contexts_GBps": round(gb / two_ms * 1e3, 2),
                          "frac_one_call": round(dev_ms / min(one), 3), "frac_two_contexts": round(dev_ms / two_ms, 3),
                          "same_bytes": same and same2}), flush=True)
        for e in encs:
            e.Close()


if __name__ == "__main__":
    main()
