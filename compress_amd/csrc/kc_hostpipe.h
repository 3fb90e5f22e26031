// kc_hostpipe.h — internal: the pipelines of the host-buffer entry points (templates over the encode call), shared by
// kc_zstd_host.cpp and kc_s2_api.cpp.
#pragma once
#include "kc_host.h"

// ---------------------------------------------------------------------------------------
// Host-buffer path (what the cgo shim calls): a three-stage pipeline over sub-batches —
//   stager thread : pageable source -> pinned slot (parallel memcpy) -> device (copy stream)
//   caller thread : the device encode of the sub-batch (context stream)
//   drainer thread: device -> pinned slot (copy-back stream) -> caller's dst (parallel memcpy)
// so the PCIe transfers and the host copies of sub-batch k+1 / k-1 run under the kernels of sub-batch k.
// Two slots per direction; the reference's own threading seam is EncodeAll being safe for concurrent use
// (zstd/encoder.go:722-729) — here the concurrency is inside one call.
// ---------------------------------------------------------------------------------------
namespace kci {

struct HostPipe {
    uint8_t* pin_in[2] = {nullptr, nullptr};
    uint8_t* pin_out[2] = {nullptr, nullptr};
    size_t in_cap = 0, out_cap = 0;
    DevBuf d_in[2], d_out[2];
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    std::vector<hipStream_t> kstreams;  // kernel streams of the chunk-fed batch
    std::vector<hipEvent_t> events;
    ~HostPipe() {
        for (hipStream_t t : kstreams) (void)hipStreamDestroy(t);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (int i = 0; i < 2; i++) {
            if (pin_in[i]) (void)hipHostFree(pin_in[i]);
            if (pin_out[i]) (void)hipHostFree(pin_out[i]);
            if (d_in[i].p) (void)hipFree(d_in[i].p);
            if (d_out[i].p) (void)hipFree(d_out[i].p);
        }
        if (s_h2d) (void)hipStreamDestroy(s_h2d);
        if (s_d2h) (void)hipStreamDestroy(s_d2h);
    }
};


inline int host_copy_threads() {
    static int n = [] {
        int t = (int)std::thread::hardware_concurrency();
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2 quota: more runnable threads only get throttled
            long long q = 0, per = 0;
            char qs[32];
            if (fscanf(f, "%31s %lld", qs, &per) == 2 && strcmp(qs, "max") != 0 && per > 0) {
                q = atoll(qs);
                const int lim = (int)((q + per - 1) / per);
                if (lim >= 1 && lim < t) t = lim;
            }
            fclose(f);
        }
        return t < 1 ? 1 : (t > 16 ? 16 : t);
    }();
    return n;
}
inline int host_copy_threads(const kc_ctx* c) {
    const int64_t t = c->cfg.host_copy_threads;
    return t >= 1 ? (int)(t > 16 ? 16 : t) : host_copy_threads();
}

inline void parallel_memcpy(uint8_t* dst, const uint8_t* src, size_t n, int threads) {
    if (n < ((size_t)8 << 20) || threads <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n / (size_t)threads) + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; t++) {
        const size_t a = (size_t)t * per;
        if (a >= n) break;
        const size_t len = a + per < n ? per : n - a;
        th.emplace_back([=] { memcpy(dst + a, src + a, len); });
    }
    for (auto& t : th) t.join();
}

// enc(d_in, rel_off, n, d_out, out_cap, out_off_rel) runs one sub-batch on the device (synchronous); max_out(len) bounds a unit's output.
template <class EncFn, class MaxFn>
kc_status host_pipeline(kc_ctx* c, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units, uint8_t* dst, uint64_t dst_cap,
                        uint64_t* out_off, uint64_t sub_bytes, EncFn enc, MaxFn max_out) {
    // ---- cut into sub-batches of ~sub_bytes of input ----
    std::vector<uint32_t> cut{0};
    std::vector<uint64_t> need;  // device output capacity per sub-batch
    {
        uint64_t acc = 0, nd = 0;
        for (uint32_t i = 0; i < n_units; i++) {
            const uint64_t len = unit_off[i + 1] - unit_off[i];
            if (i > cut.back() && acc + len > sub_bytes) { cut.push_back(i); need.push_back(nd); acc = 0; nd = 0; }
            acc += len;
            nd += ((uint64_t)max_out(len) + 15) & ~(uint64_t)15;
        }
        cut.push_back(n_units);
        need.push_back(nd);
    }
    const size_t nsub = cut.size() - 1;
    uint64_t max_in = 0, max_need = 0;
    for (size_t k = 0; k < nsub; k++) {
        max_in = std::max<uint64_t>(max_in, unit_off[cut[k + 1]] - unit_off[cut[k]]);
        max_need = std::max<uint64_t>(max_need, need[k]);
    }
    if (!c->hpipe) c->hpipe = new HostPipe();
    HostPipe* hp = (HostPipe*)c->hpipe;
    if (!hp->s_h2d) {
        // high priority: the runtime keeps a separate pool of hardware queues per priority, so the copies never share a queue
        // with (and wait in line behind) a match-finder launch; with the default 4 queues per pool and 7+ streams in the
        // process they did (measured: the third chunk's copy landed 80 ms late)
        int prLo = 0, prHi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&prLo, &prHi));
        HIPCHK(c, hipStreamCreateWithPriority(&hp->s_h2d, hipStreamNonBlocking, prHi));
        HIPCHK(c, hipStreamCreateWithPriority(&hp->s_d2h, hipStreamNonBlocking, prHi));
    }
    if (hp->in_cap < max_in + 64) {
        for (int i = 0; i < 2; i++) { if (hp->pin_in[i]) (void)hipHostFree(hp->pin_in[i]); hp->pin_in[i] = nullptr; }
        hp->in_cap = 0;
        for (int i = 0; i < 2; i++) HIPCHK(c, hipHostMalloc((void**)&hp->pin_in[i], max_in + 64, hipHostMallocDefault));
        hp->in_cap = max_in + 64;
    }
    if (hp->out_cap < max_need + 64) {
        for (int i = 0; i < 2; i++) { if (hp->pin_out[i]) (void)hipHostFree(hp->pin_out[i]); hp->pin_out[i] = nullptr; }
        hp->out_cap = 0;
        for (int i = 0; i < 2; i++) HIPCHK(c, hipHostMalloc((void**)&hp->pin_out[i], max_need + 64, hipHostMallocDefault));
        hp->out_cap = max_need + 64;
    }
    kc_status s;
    for (int i = 0; i < 2; i++)
        if ((s = ensure(c, hp->d_in[i], max_in + 64)) || (s = ensure(c, hp->d_out[i], max_need + 64))) return s;

    const int T = host_copy_threads(c);
    std::mutex m;
    std::condition_variable cv;
    size_t staged = 0, encoded = 0, drained = 0;  // sub-batches that passed each stage
    bool fail = false;
    std::string ferr;
    std::vector<uint64_t> produced(nsub, 0), pos(nsub + 1, 0);
    const int dev = c->device;

    std::thread stager([&] {
        (void)hipSetDevice(dev);
        for (size_t k = 0; k < nsub; k++) {
            {   // slot k&1 was last read by the encode of sub-batch k-2
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return fail || k < 2 || encoded >= k - 1; });
                if (fail) return;
            }
            const uint64_t a = unit_off[cut[k]], len = unit_off[cut[k + 1]] - a;
            parallel_memcpy(hp->pin_in[k & 1], src + a, (size_t)len, T);
            hipError_t e = hipMemcpyAsync(hp->d_in[k & 1].p, hp->pin_in[k & 1], (size_t)len, hipMemcpyHostToDevice, hp->s_h2d);
            if (e == hipSuccess) e = hipStreamSynchronize(hp->s_h2d);
            std::lock_guard<std::mutex> lk(m);
            if (e != hipSuccess) { fail = true; ferr = std::string("host pipeline H2D: ") + hipGetErrorString(e); }
            else staged = k + 1;
            cv.notify_all();
            if (fail) return;
        }
    });
    std::thread drainer([&] {
        (void)hipSetDevice(dev);
        for (size_t k = 0; k < nsub; k++) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return fail || encoded > k; });
                if (fail) return;
            }
            hipError_t e = hipMemcpyAsync(hp->pin_out[k & 1], hp->d_out[k & 1].p, (size_t)produced[k], hipMemcpyDeviceToHost, hp->s_d2h);
            if (e == hipSuccess) e = hipStreamSynchronize(hp->s_d2h);
            if (e == hipSuccess) parallel_memcpy(dst + pos[k], hp->pin_out[k & 1], (size_t)produced[k], T);
            std::lock_guard<std::mutex> lk(m);
            if (e != hipSuccess) { fail = true; ferr = std::string("host pipeline D2H: ") + hipGetErrorString(e); }
            else drained = k + 1;
            cv.notify_all();
            if (fail) return;
        }
    });
    kc_status rs = KC_OK;
    std::vector<uint64_t> rel, oo;
    for (size_t k = 0; k < nsub && rs == KC_OK; k++) {
        {   // input staged; output slot k&1 drained from sub-batch k-2
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return fail || (staged > k && (k < 2 || drained >= k - 1)); });
            if (fail) break;
        }
        const uint32_t u0 = cut[k], nu = cut[k + 1] - cut[k];
        rel.resize(nu + 1);
        oo.resize(nu + 1);
        for (uint32_t i = 0; i <= nu; i++) rel[i] = unit_off[u0 + i] - unit_off[u0];
        rs = enc((const uint8_t*)hp->d_in[k & 1].p, rel.data(), nu, (uint8_t*)hp->d_out[k & 1].p, need[k], oo.data());
        std::lock_guard<std::mutex> lk(m);
        if (rs != KC_OK) { fail = true; }
        else if (pos[k] + oo[nu] > dst_cap) { fail = true; rs = KC_ERR_DST_TOO_SMALL; c->err = "dst_cap too small"; }
        else {
            produced[k] = oo[nu];
            pos[k + 1] = pos[k] + oo[nu];
            for (uint32_t i = 0; i <= nu; i++) out_off[u0 + i] = pos[k] + oo[i];
            encoded = k + 1;
        }
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return fail || drained == nsub; });
        cv.notify_all();
    }
    stager.join();
    drainer.join();
    if (rs != KC_OK) return rs;
    if (fail) { if (!ferr.empty()) c->err = ferr; return KC_ERR_HIP; }
    return KC_OK;
}

// One device batch whose source arrives in chunks (a quarter of the batch each).  The stager thread copies pageable source ->
// pinned slot -> device; the caller thread sets the batch up and, per chunk, launches the whole encode of the chunk's units on the
// chunk's own stream behind its copy (enq: batch_begin / s2_encode_dev with a ChunkFeed), then drains chunk by chunk (device ->
// pinned -> dst) as each finishes.  All units of the batch end up in flight together (zstd SpeedFastest: a 1 GiB batch encodes at
// 58 ms/GiB, the 4 GiB batch at 41) and both transfers hide under the kernels of the other chunks.
//   need        bytes of c->tmp_dst the batch may write (sum of the aligned per-unit bounds)
//   enq(feed, d_in, rel_off, d_out)   enqueue everything; chunk k's output goes to d_out + region_of(feed.cut[k]), its local
//                                     offsets to feed.loc_off[cut[k] + k ...]
//   fin(&redo)  end of the batch on the context's stream; redo = encode again the plain way (returned as KC_ERR_UNSUPPORTED, no text)
template <class Enq, class RegionOf, class Fin>
kc_status host_chunk_fed(kc_ctx* c, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units, uint8_t* dst, uint64_t dst_cap,
                         uint64_t* out_off, uint64_t need, Enq enq, RegionOf region_of, Fin fin) {
    const uint64_t total = unit_off[n_units] - unit_off[0];
    // kernel chunks: a quarter of the batch each, one per kernel stream so that none waits behind another.  Measured on the 4 GiB
    // SpeedFastest batch (ms, pageable source to pageable frames): 4 x 1 GiB 215, 512M/512M/1G/2G 226, 1G/1G/2G 229, 2 x 2 GiB 235,
    // 6 x 768 MiB 244 (two chunks queue behind others), 8 x 512 MiB 253; the plain sub-batch pipeline 305
    std::vector<uint64_t> sched = {std::max<uint64_t>((total + 3) / 4, (uint64_t)64 << 20)};
    if (!c->cfg.host_chunks.empty()) sched = c->cfg.host_chunks;
    ChunkFeed feed;
    feed.cut.push_back(0);
    {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < n_units; i++) {
            const uint64_t len = unit_off[i + 1] - unit_off[i];
            const uint64_t lim = sched[std::min(feed.cut.size() - 1, sched.size() - 1)];
            if (i > feed.cut.back() && acc + len > lim) { feed.cut.push_back(i); acc = 0; }
            acc += len;
        }
        feed.cut.push_back(n_units);
    }
    const uint64_t piece = std::min<uint64_t>((uint64_t)256 << 20, std::max<uint64_t>(sched[0] / 2, 1 << 16));  // staging granularity: pageable -> pinned slot -> device
    const uint64_t max_chunk = piece;
    const size_t nchunk = feed.cut.size() - 1;
    kc_status s;
    if ((s = ensure(c, c->tmp_src, total + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
    if (!c->hpipe) c->hpipe = new HostPipe();
    HostPipe* hp = (HostPipe*)c->hpipe;
    if (!hp->s_h2d) {
        // high priority: the runtime keeps a separate pool of hardware queues per priority, so the copies never share a queue
        // with (and wait in line behind) a match-finder launch; with the default 4 queues per pool and 7+ streams in the
        // process they did (measured: the third chunk's copy landed 80 ms late)
        int prLo = 0, prHi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&prLo, &prHi));
        HIPCHK(c, hipStreamCreateWithPriority(&hp->s_h2d, hipStreamNonBlocking, prHi));
        HIPCHK(c, hipStreamCreateWithPriority(&hp->s_d2h, hipStreamNonBlocking, prHi));
    }
    while (hp->kstreams.size() < 4) {  // low priority: a queue pool of their own again, one hardware queue per stream, so the chunk kernels overlap
        hipStream_t t = nullptr;
        int prLo = 0, prHi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&prLo, &prHi));
        HIPCHK(c, hipStreamCreateWithPriority(&t, hipStreamNonBlocking, prLo));
        hp->kstreams.push_back(t);
    }
    const size_t npiece_max = (size_t)(total / piece) + nchunk + 1;
    while (hp->events.size() < 2 * nchunk + 2 + npiece_max) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hp->events.push_back(e);
    }
    const uint64_t out_chunk = (uint64_t)256 << 20;
    if (hp->in_cap < max_chunk + 64) {
        for (int i = 0; i < 2; i++) { if (hp->pin_in[i]) (void)hipHostFree(hp->pin_in[i]); hp->pin_in[i] = nullptr; }
        hp->in_cap = 0;
        for (int i = 0; i < 2; i++) HIPCHK(c, hipHostMalloc((void**)&hp->pin_in[i], max_chunk + 64, hipHostMallocDefault));
        hp->in_cap = max_chunk + 64;
    }
    if (hp->out_cap < out_chunk) {
        for (int i = 0; i < 2; i++) { if (hp->pin_out[i]) (void)hipHostFree(hp->pin_out[i]); hp->pin_out[i] = nullptr; }
        hp->out_cap = 0;
        for (int i = 0; i < 2; i++) HIPCHK(c, hipHostMalloc((void**)&hp->pin_out[i], out_chunk, hipHostMallocDefault));
        hp->out_cap = out_chunk;
    }
    for (size_t k = 0; k < nchunk; k++) { feed.landed.push_back(hp->events[2 * k]); feed.done.push_back(hp->events[2 * k + 1]); }
    feed.streams = hp->kstreams;
    const int T = host_copy_threads(c);
    const bool trace = c->cfg.host_trace != 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::mutex m;
    std::condition_variable cv;
    size_t recorded = 0;
    bool fail = false;
    const int dev = c->device;
    uint8_t* d_in = (uint8_t*)c->tmp_src.p;
    const uint64_t base0 = unit_off[0];
    hipEvent_t* pe = hp->events.data() + 2 * nchunk + 2;  // per staged piece: its pinned slot is free again
    std::thread stager([&] {
        (void)hipSetDevice(dev);
        size_t j = 0;
        for (size_t k = 0; k < nchunk; k++) {
            hipError_t e = hipSuccess;
            const uint64_t a0 = unit_off[feed.cut[k]], a1 = unit_off[feed.cut[k + 1]];
            for (uint64_t a = a0; a < a1 && e == hipSuccess; a += piece, j++) {
                const uint64_t len = std::min(piece, a1 - a);
                if (j >= 2) e = hipEventSynchronize(pe[j - 2]);
                if (e != hipSuccess) break;
                parallel_memcpy(hp->pin_in[j & 1], src + a, (size_t)len, T);
                e = hipMemcpyAsync(d_in + (a - base0), hp->pin_in[j & 1], (size_t)len, hipMemcpyHostToDevice, hp->s_h2d);
                if (e == hipSuccess) e = hipEventRecord(pe[j], hp->s_h2d);
            }
            if (e == hipSuccess) e = hipEventRecord(feed.landed[k], hp->s_h2d);
            if (trace) fprintf(stderr, "[kc host] chunk %zu (%llu MiB) staged at %.1f ms\n", k, (unsigned long long)((a1 - a0) >> 20), ms_now());
            std::lock_guard<std::mutex> lk(m);
            if (e != hipSuccess) fail = true; else recorded = k + 1;
            cv.notify_all();
            if (fail) return;
        }
    });
    feed.wait_recorded = [&](size_t k) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return fail || recorded > k; });
        return !fail;
    };
    std::vector<uint64_t> rel(n_units + 1);
    for (uint32_t i = 0; i <= n_units; i++) rel[i] = unit_off[i] - base0;
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    s = enq(feed, (const uint8_t*)d_in, (const uint64_t*)rel.data(), (uint8_t*)c->tmp_dst.p);
    { std::lock_guard<std::mutex> lk(m); }
    stager.join();  // enq returns after the last chunk was staged, or early on an error (then the stager runs out on its own buffers)
    if (s != KC_OK) {
        (void)hipStreamSynchronize(hp->s_h2d);
        (void)hipDeviceSynchronize();
        if (c->pend) { delete (Pending*)c->pend; c->pend = nullptr; }
        return s;
    }
    if (trace) fprintf(stderr, "[kc host] batch enqueued at %.1f ms\n", ms_now());
    // drain chunk by chunk as each finishes: local frame offsets, then device -> pinned -> dst in pieces, the DMA of a piece
    // under the host copy of the one before
    const uint8_t* d_out = (const uint8_t*)c->tmp_dst.p;
    hipEvent_t evo[2] = {hp->events[2 * nchunk], hp->events[2 * nchunk + 1]};
    struct Piece { uint64_t host_off, len; };
    Piece fly[2];
    size_t n_sub = 0, n_ret = 0;
    hipError_t herr = hipSuccess;
    auto retire = [&] {
        const Piece& q = fly[n_ret & 1];
        hipError_t e = hipEventSynchronize(evo[n_ret & 1]);
        if (e != hipSuccess) herr = e;
        else parallel_memcpy(dst + q.host_off, hp->pin_out[n_ret & 1], (size_t)q.len, T);
        n_ret++;
    };
    uint64_t running = 0;
    std::vector<uint64_t> loc;
    kc_status ds = KC_OK;
    for (size_t k = 0; k < nchunk && ds == KC_OK && herr == hipSuccess; k++) {
        const uint32_t u0 = feed.cut[k], nk = feed.cut[k + 1] - u0;
        loc.resize((size_t)nk + 1);
        while (n_ret < n_sub && herr == hipSuccess) retire();  // host copies of the previous chunk while this one is still encoding
        if (herr != hipSuccess) break;
        if ((herr = hipEventSynchronize(feed.done[k])) != hipSuccess) break;
        // on the copy-back stream, not the null stream: hipMemcpy would first wait for every blocking stream, i.e. for a context
        // stream created by kc_ctx_create, which is already waiting for the LAST chunk
        if ((herr = hipMemcpyAsync(loc.data(), feed.loc_off + u0 + k, ((size_t)nk + 1) * 8, hipMemcpyDeviceToHost, hp->s_d2h)) != hipSuccess) break;
        if ((herr = hipStreamSynchronize(hp->s_d2h)) != hipSuccess) break;
        const uint64_t Lk = loc[nk];
        if (running + Lk > dst_cap) { c->err = "dst_cap too small"; ds = KC_ERR_DST_TOO_SMALL; break; }
        for (uint32_t i = 0; i < nk; i++) out_off[u0 + i] = running + loc[i];
        const uint8_t* d_chunk = d_out + region_of(u0);
        for (uint64_t a = 0; a < Lk && herr == hipSuccess; a += out_chunk) {
            const uint64_t len = std::min<uint64_t>(out_chunk, Lk - a);
            if (n_sub - n_ret == 2) retire();
            if (herr != hipSuccess) break;
            herr = hipMemcpyAsync(hp->pin_out[n_sub & 1], d_chunk + a, (size_t)len, hipMemcpyDeviceToHost, hp->s_d2h);
            if (herr == hipSuccess) herr = hipEventRecord(evo[n_sub & 1], hp->s_d2h);
            fly[n_sub & 1] = Piece{running + a, len};
            n_sub++;
        }
        running += Lk;
        if (trace) fprintf(stderr, "[kc host] chunk %zu done, drain queued at %.1f ms\n", k, ms_now());
    }
    while (n_ret < n_sub && herr == hipSuccess) retire();
    out_off[n_units] = running;
    bool redo = false;
    s = fin(&redo);  // synchronises the context's stream behind every chunk
    if (c->cfg.test_feed_redo) redo = true;  // diagnostics (KC_OPT_TEST_FEED_REDO): exercise the fallback below
    if (herr != hipSuccess) { (void)hipDeviceSynchronize(); c->err = std::string("HIP error: ") + hipGetErrorString(herr); return KC_ERR_HIP; }
    if (s != KC_OK) return s;
    if (ds != KC_OK) return ds;
    if (trace) fprintf(stderr, "[kc host] drained at %.1f ms (produced %llu%s)\n", ms_now(), (unsigned long long)running, redo ? ", speculation redo: encoding again" : "");
    if (redo) { c->err.clear(); return KC_ERR_UNSUPPORTED; }  // rare (batch_end's speculation check): the sub-batch pipeline encodes it again
    return KC_OK;
}

// kc_zstd_encode_units as one chunk-fed batch.  KC_ERR_UNSUPPORTED with an empty error text: not a batch of this kind (dictionary:
// the prefixed work buffer is built from the whole source; SpeedBetter: its scratch budget wants small batches; more than
// max_batch_bytes or than the scratch budget) or a unit needed the speculation re-run - the sub-batch pipeline serves it.
inline kc_status host_overlapped_zstd(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                               uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    const uint64_t total = unit_off[n_units] - unit_off[0];
    if (o->dict != nullptr || o->level == KC_SPEED_BETTER || total > c->max_batch_bytes) { c->err.clear(); return KC_ERR_UNSUPPORTED; }
    uint64_t need = 0, scratch = 0;
    for (uint32_t i = 0; i < n_units; i++) {
        need += ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)(unit_off[i + 1] - unit_off[i])) + 15) & ~(uint64_t)15;
        scratch += zstd_unit_scratch(o, unit_off[i + 1] - unit_off[i]);
    }
    if (scratch + (scratch >> 3) + total + need > scratch_budget(c)) { c->err.clear(); return KC_ERR_UNSUPPORTED; }  // many small units: several batches
    auto enq = [&](ChunkFeed& feed, const uint8_t* d_in, const uint64_t* rel, uint8_t* d_out) {
        return batch_begin(c, o, d_in, rel, n_units, d_out, need, &feed);
    };
    auto region = [&](uint32_t u0) { return c->plan.stage_off[u0]; };
    auto fin = [&](bool* redo) { return feed_finish(c, redo); };
    return host_chunk_fed(c, src, unit_off, n_units, dst, dst_cap, out_off, need, enq, region, fin);
}

// Sub-batch of the host pipeline.  The device encode wants many units in flight (C2, ms per GiB: 4 GiB batch 42, 2 GiB 48, 1 GiB 58),
// the pipeline wants several stages: measured PCIe-inclusive on 4 GiB of C2 — 256 MiB 4.0, 512 MiB 6.8, 1 GiB 10.8, 2 GiB 13.8 GB/s.
// 2 GiB sub-batches pin 2 x (2 + 2.1) GiB of host memory per context; KC_HOST_PIPE_MIB overrides.
inline uint64_t host_sub_bytes(const kc_ctx* c, uint64_t total) {
    if (c->cfg.host_pipe_mib >= 16) return (uint64_t)c->cfg.host_pipe_mib << 20;
    return total >= ((uint64_t)4 << 30) ? ((uint64_t)2 << 30) : ((uint64_t)1 << 30);  // at least two stages from 2 GiB of input on
}

}  // namespace kci

