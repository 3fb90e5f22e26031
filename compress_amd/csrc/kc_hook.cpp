// kc_hook.cpp — kc_s2_encode_block, the s2.WriterCustomEncoder hook: concurrent callers micro-batched onto device lanes.
#include "kc_hostpipe.h"
#include <deque>

// ---------------------------------------------------------------------------------------
// kc_s2_encode_block: the s2.WriterCustomEncoder hook (s2/writer.go:1053-1064).
// "The function should expect to be called concurrently" — s2.Writer calls it from one goroutine per block
// (writer.go:455-460).  Concurrent callers on ONE context are micro-batched (group commit): a caller appends its block
// to the open slot and copies its bytes into the slot's pinned input; the first caller of a slot is its leader, which
// takes the device lock (while the previous slot still runs, later callers keep joining this one), closes the slot and
// runs ONE H2D -> kernel -> D2H for all blocks of the slot; every caller then copies its own block out.  An idle
// context adds no waiting: a lone caller's slot closes at once.
// ---------------------------------------------------------------------------------------


namespace kci {

struct S2Hook {
    struct Slot {
        uint8_t* h_in = nullptr;   // pinned
        uint8_t* h_out = nullptr;  // pinned
        std::vector<uint64_t> in_off, out_off;
        uint32_t n = 0, copied = 0, left = 0;
        bool open = false, closed = false, done = false;
        kc_status status = KC_OK;
        std::condition_variable cv;  // the slot's own callers: its leader (all bytes staged?) and its followers (done?)
    };
    static constexpr int kMaxLanes = 8;
    static constexpr int kSlots = kMaxLanes + 2;  // one per lane on the device, one filling, one draining
    std::mutex m;
    std::condition_variable cv;
    // Lanes: contexts of the hook's own (stream + scratch each), so that several slots are on the device at once — a slot of a few
    // blocks keeps a few CUs busy for the ~3 ms of one block, and a caller that arrives meanwhile need not wait for it to finish.
    kc_ctx* lanes[kMaxLanes] = {nullptr};
    bool lane_busy[kMaxLanes] = {false};
    int n_lanes = 1;
    Slot slots[kSlots];
    int n_slots = 3;
    int cur = -1;
    size_t in_cap = (size_t)8 << 20, out_cap = 0;
    uint32_t max_n = 256;
    int wait_us = 0;
    bool ok = false;
    std::atomic<uint64_t> n_calls{0}, n_batches{0}, n_declined{0};
    // host first: the deadlines (steady clock, ns) until which the callers that were sent back to the built-in encoder are taken to be
    // busy with their block; ascending (equal block sizes) or nearly so — expired ones are dropped from the front
    int host_cores = 1;

    bool init(const KcCfg& g, int device) {
        wait_us = (int)g.hook_wait_us;
        // the hardware threads of the host, NOT the cgroup's CPU quota: measured on the GPU boxes (cpu.max = 16 CPUs of a 256-thread
        // host), 64 callers of the built-in encoder reach 60 GB/s — the quota is not what bounds short bursts, and a caller sent to the
        // device on its account was 3.5x slower than left alone (gpurun_out/r6o)
        host_cores = (int)std::thread::hardware_concurrency();
        if (host_cores < 1) host_cores = 1;
        max_n = (uint32_t)std::max<int64_t>(1, g.hook_batch);
        n_lanes = (int)std::min<int64_t>(kMaxLanes, std::max<int64_t>(1, g.hook_lanes));
        n_slots = n_lanes + 2;
        out_cap = in_cap + (size_t)32 * max_n + 64;
        for (int i = 0; i < n_lanes; i++)
            if (kc_ctx_create(&lanes[i], device, nullptr) != KC_OK) return false;
        for (int i = 0; i < n_slots; i++) {
            Slot& sl = slots[i];
            if (hipHostMalloc((void**)&sl.h_in, in_cap, hipHostMallocDefault) != hipSuccess) return false;
            if (hipHostMalloc((void**)&sl.h_out, out_cap, hipHostMallocDefault) != hipSuccess) return false;
            sl.in_off.assign(max_n + 1, 0);
            sl.out_off.assign(max_n + 1, 0);
        }
        ok = true;
        return true;
    }
    ~S2Hook() {
        for (auto& sl : slots) {
            if (sl.h_in) (void)hipHostFree(sl.h_in);
            if (sl.h_out) (void)hipHostFree(sl.h_out);
        }
        for (kc_ctx* l : lanes)
            if (l) kc_ctx_destroy(l);
    }
};

void s2_hook_free(void* h) { delete (S2Hook*)h; }

// Host-first bookkeeping, process-wide (the host's hardware threads are one pool whatever the number of hooks).  A thread that was
// sent back to the built-in encoder holds a booking until it calls again (then it has finished that block) or until the time a slow
// core needs for the block has passed (2 ns per byte = 500 MB/s; the reference's assembly encoder does 1.2 GB/s on JSON).  The common
// case — a booked thread back for its next block — touches only the thread's own record: 64 callers make ~10^6 calls per second
// together, and one mutex for them cost a third of the built-in encoder's rate (gpurun_out/r6o).  Expired bookings of threads that
// never came back are reclaimed by the first caller that finds every place taken.
struct HostBooking {
    std::atomic<int64_t> deadline{0};
    std::atomic<bool> booked{false};
};
static std::atomic<int64_t> g_host_booked{0};
static std::mutex g_host_reg_m;
static std::vector<HostBooking*> g_host_reg;  // every thread's record (never freed: a few bytes per thread that ever called a hook)

static bool host_first_book(int64_t places, uint64_t src_len) {
    thread_local HostBooking* tb = nullptr;
    if (!tb) {
        tb = new HostBooking();
        std::lock_guard<std::mutex> g(g_host_reg_m);
        g_host_reg.push_back(tb);
    }
    const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const int64_t until = now + (int64_t)src_len * 2;
    if (tb->booked.load(std::memory_order_relaxed)) { tb->deadline.store(until, std::memory_order_relaxed); return true; }
    for (int attempt = 0; attempt < 2; attempt++) {
        if (g_host_booked.fetch_add(1, std::memory_order_acq_rel) < places) {
            tb->deadline.store(until, std::memory_order_relaxed);
            tb->booked.store(true, std::memory_order_release);
            return true;
        }
        g_host_booked.fetch_sub(1, std::memory_order_acq_rel);
        if (attempt) break;
        int64_t freed = 0;
        std::lock_guard<std::mutex> g(g_host_reg_m);
        for (HostBooking* b : g_host_reg)
            if (b != tb && b->booked.load(std::memory_order_acquire) && b->deadline.load(std::memory_order_relaxed) <= now && b->booked.exchange(false)) freed++;
        if (freed) g_host_booked.fetch_sub(freed, std::memory_order_acq_rel);
        else break;
    }
    return false;  // every place is taken by a thread that is (taken to be) encoding on the host: this caller is the overflow
}

// one slot through the device: pinned input -> tmp_src, N x s2.Encode, tmp_dst -> pinned output
kc_status s2_hook_run(kc_ctx* c, S2Hook::Slot& sl) {
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = sl.n;
    const uint64_t total = sl.in_off[n];
    uint64_t need = 0;
    for (uint32_t i = 0; i < n; i++) need += ((uint64_t)kc_s2_max_encoded_len((int64_t)(sl.in_off[i + 1] - sl.in_off[i])) + 15) & ~(uint64_t)15;
    kc_status s;
    if ((s = ensure(c, c->tmp_src, total + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
    HIPCHK(c, hipMemcpyAsync(c->tmp_src.p, sl.h_in, total, hipMemcpyHostToDevice, c->stream));
    s = kc_s2_encode_blocks_dev(c, (const uint8_t*)c->tmp_src.p, sl.in_off.data(), n, (uint8_t*)c->tmp_dst.p, need, sl.out_off.data());
    if (s != KC_OK) return s;
    HIPCHK(c, hipMemcpyAsync(sl.h_out, c->tmp_dst.p, sl.out_off[n], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KC_OK;
}

}  // namespace kci

extern "C" {

int64_t kc_s2_encode_block(kc_ctx* c, uint8_t* dst, uint64_t dst_cap, const uint8_t* src, uint64_t src_len) {
    // WriterCustomEncoder contract (s2/writer.go:1053-1064): no varint header; 0 = incompressible; <0 = use built-in.
    if (!c || !dst || !src) return -1;
    if (src_len > (4u << 20) || src_len == 0) return -1;
    if (src_len < 32) return 0;  // encodeBlock: len < minNonLiteralBlockSize -> 0 (stored by the writer)
    std::call_once(c->hook_once, [c] {
        S2Hook* h = new S2Hook();
        if (hipSetDevice(c->device) != hipSuccess || !h->init(c->cfg, c->device)) { delete h; return; }
        c->hook = h;
    });
    S2Hook* h = (S2Hook*)c->hook;
    if (!h) return -1;
    h->n_calls.fetch_add(1, std::memory_order_relaxed);
    {   // host first (include/kcgpu.h): while the host has a hardware thread that is not booked, the built-in encoder serves the caller faster
        const int64_t hf = c->cfg.hook_host_first < 0 ? h->host_cores : c->cfg.hook_host_first;
        if (hf > 0 && host_first_book(hf, src_len)) { h->n_declined.fetch_add(1, std::memory_order_relaxed); return -1; }
    }
    std::unique_lock<std::mutex> lk(h->m);
    S2Hook::Slot* sl = nullptr;
    for (;;) {
        if (h->cur >= 0) {
            S2Hook::Slot& cs = h->slots[h->cur];
            if (!cs.closed && cs.n < h->max_n && cs.in_off[cs.n] + src_len <= h->in_cap) { sl = &cs; break; }
            cs.closed = true;  // full: its leader will run it as it is
            h->cur = -1;
            cs.cv.notify_all();
        }
        int fr = -1;
        for (int i = 0; i < h->n_slots; i++)
            if (!h->slots[i].open) { fr = i; break; }
        if (fr < 0) { h->cv.wait(lk); continue; }
        S2Hook::Slot& ns = h->slots[fr];
        ns.open = true; ns.closed = false; ns.done = false; ns.n = 0; ns.copied = 0; ns.left = 0; ns.status = KC_OK;
        ns.in_off[0] = 0;
        h->cur = fr;
    }
    const uint32_t idx = sl->n++;
    const uint64_t off = sl->in_off[idx];
    sl->in_off[idx + 1] = off + src_len;
    sl->left++;
    const bool leader = idx == 0;
    lk.unlock();
    memcpy(sl->h_in + off, src, src_len);  // callers stage their own bytes in parallel
    lk.lock();
    sl->copied++;
    if (!leader && (sl->closed || sl->n >= h->max_n)) sl->cv.notify_all();  // (the leader may be waiting for the last bytes, or for a full slot)
    if (leader) {
        // take a lane; while all of them are on the device, callers keep joining this slot
        int ln = -1;
        h->cv.wait(lk, [&] {
            for (int i = 0; i < h->n_lanes; i++)
                if (!h->lane_busy[i]) { ln = i; return true; }
            return false;
        });
        h->lane_busy[ln] = true;
        if (h->wait_us > 0 && !sl->closed && sl->n < h->max_n)
            sl->cv.wait_for(lk, std::chrono::microseconds(h->wait_us), [&] { return sl->closed || sl->n >= h->max_n; });
        sl->closed = true;
        if (h->cur >= 0 && &h->slots[h->cur] == sl) h->cur = -1;
        sl->cv.wait(lk, [&] { return sl->copied == sl->n; });
        kc_ctx* const lc = h->lanes[ln];
        // the caller's options as they are now — only the scalar fields the S2 block path reads (not the whole KcCfg: it holds a
        // vector, and another thread may be in kc_ctx_set_option on c), and the caller's scratch ceiling
        lc->cfg.s2_variant = c->cfg.s2_variant;
        lc->cfg.match_path = c->cfg.match_path;
        lc->cfg.s2_lds_max_blocks = c->cfg.s2_lds_max_blocks;
        lc->cfg.s2_lds_spec_w0 = c->cfg.s2_lds_spec_w0;
        lc->cfg.spec_w0 = c->cfg.spec_w0;
        lc->cfg.spec_grow = c->cfg.spec_grow;
        lc->max_scratch_bytes = c->max_scratch_bytes;
        lk.unlock();
        const kc_status st = s2_hook_run(lc, *sl);
        h->n_batches++;
        lk.lock();
        if (st != KC_OK) c->err = lc->err;
        h->lane_busy[ln] = false;
        sl->status = st;
        sl->done = true;
        sl->cv.notify_all();
        h->cv.notify_all();  // a lane is free
    } else {
        sl->cv.wait(lk, [&] { return sl->done; });
    }
    int64_t ret = -1;
    const uint8_t* enc = nullptr;
    uint64_t body = 0;
    if (sl->status == KC_OK) {
        enc = sl->h_out + sl->out_off[idx];
        const uint64_t elen = sl->out_off[idx + 1] - sl->out_off[idx];
        size_t hdr = 0;  // strip the uvarint(len) header
        while (enc[hdr] & 0x80) hdr++;
        hdr++;
        body = elen - hdr;
        enc += hdr;
        const uint64_t nm1 = src_len - 1;  // emitLiteral header size depends on len-1 (encode_go.go:86-113)
        const uint64_t storedLen = src_len + (nm1 < 60 ? 1 : (nm1 < (1 << 8) ? 2 : (nm1 < (1 << 16) ? 3 : (nm1 < (1 << 24) ? 4 : 5))));
        if (body == storedLen) ret = 0;  // a block stored as one literal run: encodeBlock returned 0
        else if (body > dst_cap) ret = -1;
        else ret = (int64_t)body;
    }
    lk.unlock();
    if (ret > 0) memcpy(dst, enc, body);
    lk.lock();
    if (--sl->left == 0) {
        sl->open = false;
        h->cv.notify_all();
    }
    return ret;
}

// diagnostics of the hook's micro-batcher: calls served and device batches run so far
void kc_s2_hook_stats(const kc_ctx* c, uint64_t* calls, uint64_t* batches) {
    const S2Hook* h = c ? (const S2Hook*)c->hook : nullptr;
    if (calls) *calls = h ? h->n_calls.load() - h->n_declined.load() : 0;  // calls served on the device
    if (batches) *batches = h ? h->n_batches.load() : 0;
}
// calls answered -1 by the host-first rule (the caller's built-in encoder took them)
uint64_t kc_s2_hook_declined(const kc_ctx* c) {
    const S2Hook* h = c ? (const S2Hook*)c->hook : nullptr;
    return h ? h->n_declined.load() : 0;
}

}  // extern "C"

