#!/usr/bin/env python3
"""Measurement only: does s2.Encode gain from launches that overlap (as zstd SpeedFastest / SpeedDefault do, bench.py --split)?
One 2 GiB batch of 64 KiB JSON blocks per call on one context, against T host threads each encoding 2 GiB / P per call on a context
and stream of its own (ctypes releases the GIL: the calls run side by side), outputs in separate buffers.
    python tools/s2_split_probe.py"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from compress_amd import s2, _lib

UNIT = 64 << 10
N = (2 << 30) // UNIT
host = _lib.corpus_fill("J", 0x5EED0003, 0, N, UNIT)
d_src = torch.from_numpy(host).cuda()
off = np.arange(N + 1, dtype=np.uint64) * UNIT


def run(parts, threads, passes):
    """`passes` passes over the 2 GiB batch, each as `parts` calls of N / parts blocks, spread over `threads` threads"""
    cuts = [N * h // parts for h in range(parts + 1)]
    streams = [torch.cuda.Stream() for _ in range(threads)]
    encs = [s2.BlockEncoder(device=0, stream=st.cuda_stream) for st in streams]
    cap = (N // parts + 1) * (s2.MaxEncodedLen(UNIT) + 16) + 64
    dsts = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(threads)]
    jobs = [(p, h) for p in range(passes) for h in range(parts)]
    nxt = [0]
    lock = threading.Lock()
    outb = [0] * threads

    def work(t):
        while True:
            with lock:
                if nxt[0] >= len(jobs):
                    return
                p, h = jobs[nxt[0]]
                nxt[0] += 1
            a, b = cuts[h], cuts[h + 1]
            o = encs[t].EncodeBlocksDevice(d_src.data_ptr(), off[a:b + 1], dsts[t].data_ptr(), cap)
            outb[t] += int(o[b - a])

    for warm in (True, False):
        nxt[0] = 0
        if warm:
            saved, jobs[:] = list(jobs), [(0, h) for h in range(parts)] * 2
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if warm:
            jobs[:] = saved
            outb[:] = [0] * threads
    for e in encs:
        e.Close()
    return dt / passes * 1e3, sum(outb) / passes


for parts, threads in ((1, 1), (2, 2), (2, 3), (4, 4), (1, 2), (2, 2), (1, 1)):
    ms, ob = run(parts, threads, 12)
    print("parts %d threads %d: %.2f ms per 2 GiB  (%.0f MB/s)  out %.0f bytes" % (parts, threads, ms, (2 << 30) / ms / 1e3, ob), flush=True)
