// kc_roll.cpp — the rolling host pipeline: what the host-buffer entry points (kc_zstd_encode_units, kc_s2_encode_blocks_lvl and
// their submit forms — the calls a cgo binding makes with Go slices, zstd/encoder.go:716-729, s2/encode.go:29-56) run large inputs
// through since round 6.
//
// One engine per device, shared by every context of the process.  A call is cut into sub-batches (a quarter of the input, at most
// 1 GiB, at unit boundaries); the sub-batches of all calls in flight move through three stages in arrival order:
//   stager thread    pageable source -> pinned ring (parallel memcpy, 256 MiB pieces) -> the sub-batch's device slot, copy stream
//   8 encoder lanes  a context of the engine's own each (scratch + streams): the whole device encode of one sub-batch
//                    (kc_zstd_encode_units_dev / kc_s2_encode_blocks_lvl_dev behind the slot's "landed" event)
//   drainer thread   device slot -> pinned ring -> the caller's dst at the running output position, frame offsets rebased
// Four kernel streams, two lanes on each: HIP multiplexes a priority level's streams onto four hardware queues (a fifth stream only
// waits in line behind one of the four), and four sub-batches of a quarter of a 4 GiB batch are exactly one residency of the match
// finder on the chip.  A lane's first stage (checksum, table preparation, match finder / S2 encoder) runs on its pair's stream, its
// second stage (zstd: entropy coding, size scan, compaction) on a stream of the lane's own, so while one lane of a pair is in its
// second stage or back on the host, its partner's match finder is already running in the pair's queue: four match finders stay
// resident.  With one lane per queue the queue idled through every second stage (35 ms of a 165 ms cycle per GiB: 0.89 of the
// device-resident rate on C2, gpurun_out/r6d).  Ten device slots: eight encoding, one landing, one draining.  The stages never wait
// for a call to end: the next call's first sub-batch lands while the previous call's last ones encode and drain, so a caller that
// keeps calls in flight (contexts, submit / wait) sees close to the device-resident rate, not the sum of transfer and encode (round
// 5's chunk-fed path: one device batch per call, 0.74 of the device-resident rate on C2; profiles/README_r06.md).
// Output bytes do not depend on the cut: every unit is an independent frame / block.
#include "kc_hostpipe.h"
#include <deque>

namespace kci {

namespace {

struct RollCall;

struct RollJob {
    RollCall* call = nullptr;
    uint32_t k = 0;       // sub-batch index within the call
    int slot = -1;
    bool encoded = false;
    std::vector<uint64_t> rel, oo;
};

struct RollCall {
    kc_ctx* owner = nullptr;
    RollEncFn enc;
    const uint8_t* src = nullptr;
    const uint64_t* unit_off = nullptr;
    uint8_t* dst = nullptr;
    uint64_t dst_cap = 0;
    uint64_t* out_off = nullptr;
    std::vector<uint32_t> cut;
    std::vector<uint64_t> need;
    std::vector<RollJob> jobs;
    uint64_t pos = 0;   // output bytes of the sub-batches drained so far
    size_t done = 0;    // sub-batches that have left the drainer
    bool fail = false;
    kc_status st = KC_OK;
    std::string err;
    int last_path = 0;
    bool trace = false;
    bool src_pinned = false, dst_pinned = false;  // the caller's buffers are page-locked (kc_host_alloc / hipHostMalloc / hipHostRegister): DMA straight from / into them
    std::chrono::steady_clock::time_point t0;
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

struct RollEngine {
    static constexpr int kQueues = 4;
    static constexpr int kEnc = 2 * kQueues;
    static constexpr int kSlots = kEnc + 2;
    static constexpr uint64_t kPiece = (uint64_t)256 << 20;
    int device = 0;
    bool ok = false;
    std::mutex m;
    std::condition_variable cv;
    std::deque<RollJob*> q_stage, q_enc, q_drain;
    struct Slot {
        DevBuf in, out;
        hipEvent_t landed = nullptr;
        bool busy = false;
    } slots[kSlots];
    kc_ctx* lanes[kEnc] = {nullptr};
    hipStream_t pair_stream[kQueues] = {nullptr};  // first stage of lanes q and q + kQueues
    hipStream_t lane_stream2[kEnc] = {nullptr};    // second stage, one per lane
    kc_ctx* holder = nullptr;  // owns nothing but the error text / oom flag of the slot allocations
    uint8_t* pin_in[2] = {nullptr, nullptr};
    uint8_t* pin_out[2] = {nullptr, nullptr};
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    uint64_t n_in = 0;  // pieces staged so far (ring position)
    bool lane_busy[kEnc] = {false};
    RollJob* lane_job[kEnc] = {nullptr};  // a staged sub-batch handed to an idle lane (else it waits in q_enc for the first lane to finish)

    // m held.  An idle lane for a staged sub-batch: one whose pair's queue is idle if there is one — two lanes of one pair run their
    // match finders one after the other, so a lone call's four sub-batches must land on four different queues.
    int pick_lane() const {
        int best = -1, best_load = 99;
        for (int e = 0; e < kEnc; e++) {
            if (lane_busy[e]) continue;
            const int load = lane_busy[(e + kQueues) % kEnc] ? 1 : 0;
            if (load < best_load) { best = e; best_load = load; }
        }
        return best;
    }

    bool init(int dev) {
        device = dev;
        if (hipSetDevice(dev) != hipSuccess) return false;
        int prLo = 0, prHi = 0;
        if (hipDeviceGetStreamPriorityRange(&prLo, &prHi) != hipSuccess) return false;
        // high priority: a queue pool of their own, so a copy never waits in line behind a match-finder launch (kc_hostpipe.h)
        if (hipStreamCreateWithPriority(&s_h2d, hipStreamNonBlocking, prHi) != hipSuccess) return false;
        if (hipStreamCreateWithPriority(&s_d2h, hipStreamNonBlocking, prHi) != hipSuccess) return false;
        for (int i = 0; i < 2; i++) {
            if (hipHostMalloc((void**)&pin_in[i], kPiece, hipHostMallocDefault) != hipSuccess) return false;
            if (hipHostMalloc((void**)&pin_out[i], kPiece, hipHostMallocDefault) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&ev_in[i], hipEventDisableTiming) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&ev_out[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        for (auto& s : slots)
            if (hipEventCreateWithFlags(&s.landed, hipEventDisableTiming) != hipSuccess) return false;
        // low priority: HIP keeps a pool of four hardware queues per priority level and deals a level's streams onto them round robin.
        // At the default level the lanes share the pool with every other stream of the process (the callers' contexts, PyTorch's) and
        // two lanes end up behind each other in one queue (measured: two of four sub-batches finishing together, 35 ms after the
        // other two); nothing else in the library uses the low level, so its four queues are one per lane.
        for (int q = 0; q < kQueues; q++)
            if (hipStreamCreateWithPriority(&pair_stream[q], hipStreamNonBlocking, prLo) != hipSuccess) return false;
        for (int i = 0; i < kEnc; i++) {
            if (kc_ctx_create(&lanes[i], dev, (void*)pair_stream[i % kQueues]) != KC_OK) return false;
            // (default priority: the second stages share that level's queues with the rest of the process; what they wait for there
            // is short, and they keep out of the match finders' queues)
            if (hipStreamCreateWithFlags(&lane_stream2[i], hipStreamNonBlocking) != hipSuccess) return false;
            lanes[i]->stream2 = lane_stream2[i];
            lanes[i]->lane_preclear = true;
        }
        if (kc_ctx_create(&holder, dev, nullptr) != KC_OK) return false;
        ok = true;
        std::thread([this] { stage_loop(); }).detach();
        for (int e = 0; e < kEnc; e++) std::thread([this, e] { enc_loop(e); }).detach();
        std::thread([this] { drain_loop(); }).detach();
        return true;
    }

    void fail_call(RollCall* c, kc_status st, const std::string& text) {  // m held
        if (c->fail) return;
        c->fail = true;
        c->st = st;
        c->err = text;
    }

    void stage_loop() {
        (void)hipSetDevice(device);
        const int T = host_copy_threads();
        for (;;) {
            RollJob* j = nullptr;
            int sl = -1;
            bool skip = false;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] {
                    if (q_stage.empty()) return false;
                    for (int i = 0; i < kSlots; i++) if (!slots[i].busy) return true;
                    return false;
                });
                j = q_stage.front();
                q_stage.pop_front();
                for (int i = 0; i < kSlots; i++) if (!slots[i].busy) { sl = i; break; }
                slots[sl].busy = true;
                j->slot = sl;
                skip = j->call->fail;
            }
            RollCall* c = j->call;
            if (!skip) {
                const uint64_t a0 = c->unit_off[c->cut[j->k]], a1 = c->unit_off[c->cut[j->k + 1]];
                Slot& S = slots[sl];
                holder->err.clear();
                holder->oom = false;
                kc_status s = ensure(holder, S.in, (size_t)(a1 - a0) + 64);
                if (s == KC_OK) s = ensure(holder, S.out, (size_t)c->need[j->k] + 64);
                hipError_t e = hipSuccess;
                if (s == KC_OK) {
                    const int Tc = c->owner->cfg.host_copy_threads >= 1 ? host_copy_threads(c->owner) : T;
                    if (c->src_pinned) {  // no staging copy: one DMA from the caller's pinned pages
                        e = hipMemcpyAsync(S.in.p, c->src + a0, (size_t)(a1 - a0), hipMemcpyHostToDevice, s_h2d);
                    } else
                    for (uint64_t a = a0; a < a1 && e == hipSuccess; a += kPiece, n_in++) {
                        const uint64_t len = std::min(kPiece, a1 - a);
                        if (n_in >= 2) e = hipEventSynchronize(ev_in[n_in & 1]);
                        if (e != hipSuccess) break;
                        parallel_memcpy(pin_in[n_in & 1], c->src + a, (size_t)len, Tc);
                        e = hipMemcpyAsync((uint8_t*)S.in.p + (a - a0), pin_in[n_in & 1], (size_t)len, hipMemcpyHostToDevice, s_h2d);
                        if (e == hipSuccess) e = hipEventRecord(ev_in[n_in & 1], s_h2d);
                    }
                    if (e == hipSuccess) e = hipEventRecord(S.landed, s_h2d);
                }
                if (c->trace) fprintf(stderr, "[kc roll] sub-batch %u (%llu MiB) staged into slot %d at %.1f ms\n", j->k, (unsigned long long)((a1 - a0) >> 20), sl, c->ms());
                if (s != KC_OK || e != hipSuccess) {
                    std::lock_guard<std::mutex> lk(m);
                    if (s != KC_OK) fail_call(c, s, holder->err);
                    else fail_call(c, KC_ERR_HIP, std::string("rolling pipeline H2D: ") + hipGetErrorString(e));
                }
            }
            std::lock_guard<std::mutex> lk(m);
            const int e = pick_lane();
            if (e >= 0) { lane_busy[e] = true; lane_job[e] = j; }
            else q_enc.push_back(j);
            q_drain.push_back(j);
            cv.notify_all();
        }
    }

    void enc_loop(int e) {
        (void)hipSetDevice(device);
        kc_ctx* lane = lanes[e];
        for (;;) {
            RollJob* j = nullptr;
            bool skip = false;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return lane_job[e] != nullptr || (!lane_busy[e] && !q_enc.empty()); });
                if (lane_job[e] != nullptr) { j = lane_job[e]; lane_job[e] = nullptr; }
                else { j = q_enc.front(); q_enc.pop_front(); lane_busy[e] = true; }
                skip = j->call->fail;
            }
            RollCall* c = j->call;
            if (!skip) {
                Slot& S = slots[j->slot];
                const uint32_t nu = c->cut[j->k + 1] - c->cut[j->k];
                // the caller's tunables for this sub-batch (the caller is blocked in its call: its context does not change)
                lane->cfg = c->owner->cfg;
                lane->max_scratch_bytes = c->owner->max_scratch_bytes;
                lane->max_batch_bytes = c->owner->max_batch_bytes;
                kc_status s = KC_OK;
                std::string text;
                const hipError_t he = hipStreamWaitEvent(lane->stream, S.landed, 0);
                if (he != hipSuccess) { s = KC_ERR_HIP; text = std::string("rolling pipeline: ") + hipGetErrorString(he); }
                else {
                    s = c->enc(lane, (const uint8_t*)S.in.p, j->rel.data(), nu, (uint8_t*)S.out.p, c->need[j->k], j->oo.data());
                    if (s != KC_OK) text = lane->err;
                }
                if (c->trace) fprintf(stderr, "[kc roll] sub-batch %u encoded on lane %d at %.1f ms\n", j->k, e, c->ms());
                std::lock_guard<std::mutex> lk(m);
                if (s != KC_OK) fail_call(c, s, text);
                c->last_path = lane->last_path;
            }
            std::lock_guard<std::mutex> lk(m);
            j->encoded = true;
            lane_busy[e] = false;
            cv.notify_all();
        }
    }

    void drain_loop() {
        (void)hipSetDevice(device);
        const int T = host_copy_threads();
        for (;;) {
            RollJob* j = nullptr;
            bool skip = false;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !q_drain.empty() && q_drain.front()->encoded; });
                j = q_drain.front();
                q_drain.pop_front();
                skip = j->call->fail;
            }
            RollCall* c = j->call;
            if (!skip) {
                Slot& S = slots[j->slot];
                const uint32_t u0 = c->cut[j->k], nu = c->cut[j->k + 1] - u0;
                const uint64_t L = j->oo[nu];
                if (c->pos + L > c->dst_cap) {
                    std::lock_guard<std::mutex> lk(m);
                    fail_call(c, KC_ERR_DST_TOO_SMALL, "dst_cap too small");
                } else {
                    for (uint32_t i = 0; i <= nu; i++) c->out_off[u0 + i] = c->pos + j->oo[i];
                    const int Tc = c->owner->cfg.host_copy_threads >= 1 ? host_copy_threads(c->owner) : T;
                    // device -> pinned -> dst in pieces, the DMA of a piece under the host copy of the one before (a pinned dst: one DMA)
                    hipError_t e = hipSuccess;
                    if (c->dst_pinned) {
                        if (L) e = hipMemcpyAsync(c->dst + c->pos, S.out.p, (size_t)L, hipMemcpyDeviceToHost, s_d2h);
                        if (e == hipSuccess) e = hipStreamSynchronize(s_d2h);
                    } else {
                    uint64_t q_off[2] = {0, 0}, q_len[2] = {0, 0};
                    size_t n_sub = 0, n_ret = 0;
                    auto retire = [&] {
                        const hipError_t r = hipEventSynchronize(ev_out[n_ret & 1]);
                        if (r != hipSuccess) e = r;
                        else parallel_memcpy(c->dst + c->pos + q_off[n_ret & 1], pin_out[n_ret & 1], (size_t)q_len[n_ret & 1], Tc);
                        n_ret++;
                    };
                    for (uint64_t a = 0; a < L && e == hipSuccess; a += kPiece) {
                        const uint64_t len = std::min(kPiece, L - a);
                        if (n_sub - n_ret == 2) retire();
                        if (e != hipSuccess) break;
                        e = hipMemcpyAsync(pin_out[n_sub & 1], (const uint8_t*)S.out.p + a, (size_t)len, hipMemcpyDeviceToHost, s_d2h);
                        if (e == hipSuccess) e = hipEventRecord(ev_out[n_sub & 1], s_d2h);
                        q_off[n_sub & 1] = a;
                        q_len[n_sub & 1] = len;
                        n_sub++;
                    }
                    while (n_ret < n_sub && e == hipSuccess) retire();
                    }
                    if (e != hipSuccess) {
                        (void)hipStreamSynchronize(s_d2h);
                        std::lock_guard<std::mutex> lk(m);
                        fail_call(c, KC_ERR_HIP, std::string("rolling pipeline D2H: ") + hipGetErrorString(e));
                    } else {
                        c->pos += L;
                    }
                    if (c->trace) fprintf(stderr, "[kc roll] sub-batch %u drained at %.1f ms\n", j->k, c->ms());
                }
            }
            std::lock_guard<std::mutex> lk(m);
            slots[j->slot].busy = false;
            c->done++;
            cv.notify_all();
        }
    }
};

std::mutex g_eng_m;
RollEngine* g_eng[64] = {nullptr};
bool g_eng_tried[64] = {false};

// The engine of a device: created by the first large host-buffer call, lives to the end of the process (its threads sleep on the
// queue; deliberately never destroyed: no destructor may run against threads that wait on the engine's condition variable).
RollEngine* engine_for(int device) {
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_eng_m);
    if (g_eng_tried[device]) return g_eng[device];
    g_eng_tried[device] = true;
    RollEngine* e = new RollEngine();
    if (!e->init(device)) {
        (void)hipGetLastError();
        return nullptr;  // (what it allocated stays: a failed start-up is not retried)
    }
    g_eng[device] = e;
    return e;
}

}  // namespace

// kc_device_trim: with no call in flight, give the engine's device memory back (slots and the lanes' scratch: ~10 GiB per lane for
// SpeedFastest sub-batches of 1 GiB); the next large call allocates again.  KC_ERR_BAD_ARG while calls are in flight.
kc_status host_roll_trim(int device) {
    RollEngine* E = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_eng_m);
        if (device < 0 || device >= 64) return KC_ERR_BAD_ARG;
        E = g_eng[device];
    }
    if (!E) return KC_OK;  // never started: nothing held
    std::lock_guard<std::mutex> lk(E->m);  // (the stages take jobs under this lock: holding it keeps the engine idle)
    if (!E->q_stage.empty() || !E->q_enc.empty() || !E->q_drain.empty()) return KC_ERR_BAD_ARG;
    for (int e = 0; e < RollEngine::kEnc; e++) if (E->lane_busy[e] || E->lane_job[e]) return KC_ERR_BAD_ARG;
    for (auto& sl : E->slots) if (sl.busy) return KC_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return KC_ERR_HIP;
    for (auto& sl : E->slots) {
        if (sl.in.p) (void)hipFree(sl.in.p);
        if (sl.out.p) (void)hipFree(sl.out.p);
        sl.in = DevBuf();
        sl.out = DevBuf();
    }
    kc_status rs = KC_OK;
    for (kc_ctx* l : E->lanes) { const kc_status s = kc_ctx_trim(l); if (s != KC_OK) rs = s; }
    return rs;
}

uint64_t host_roll_sub_bytes(const kc_ctx* c, uint64_t total) {
    if (c->cfg.host_roll_mib >= 1) return (uint64_t)c->cfg.host_roll_mib << 20;
    const uint64_t q = (total + 3) / 4;
    return std::min<uint64_t>((uint64_t)1 << 30, std::max<uint64_t>(q, (uint64_t)64 << 20));
}

kc_status host_rolling(kc_ctx* c, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units, uint8_t* dst, uint64_t dst_cap,
                       uint64_t* out_off, const RollEncFn& enc, const std::function<uint64_t(uint64_t)>& max_out, uint64_t sub_bytes) {
    RollEngine* E = engine_for(c->device);
    if (!E) { c->err.clear(); return KC_ERR_UNSUPPORTED; }  // no engine on this device (start-up failed): the caller's older paths serve the call
    RollCall call;
    call.owner = c;
    call.enc = enc;
    call.src = src;
    call.unit_off = unit_off;
    call.dst = dst;
    call.dst_cap = dst_cap;
    call.out_off = out_off;
    call.trace = c->cfg.host_trace != 0;
    call.t0 = std::chrono::steady_clock::now();
    {   // page-locked caller buffers need no staging copies (hipPointerGetAttributes fails on ordinary memory: that is the answer "no")
        auto pinned = [](const void* p, uint64_t len) {
            hipPointerAttribute_t a0, a1;
            if (len == 0) return false;
            if (hipPointerGetAttributes(&a0, p) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (hipPointerGetAttributes(&a1, (const uint8_t*)p + len - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
            return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost;
        };
        call.src_pinned = pinned(src + unit_off[0], unit_off[n_units] - unit_off[0]);
        call.dst_pinned = pinned(dst, dst_cap);
    }
    const uint64_t total = unit_off[n_units] - unit_off[0];
    const uint64_t sub = sub_bytes ? sub_bytes : host_roll_sub_bytes(c, total);
    call.cut.push_back(0);
    {
        uint64_t acc = 0, nd = 0;
        for (uint32_t i = 0; i < n_units; i++) {
            const uint64_t len = unit_off[i + 1] - unit_off[i];
            if (i > call.cut.back() && acc + len > sub) { call.cut.push_back(i); call.need.push_back(nd); acc = 0; nd = 0; }
            acc += len;
            nd += (max_out(len) + 15) & ~(uint64_t)15;
        }
        call.cut.push_back(n_units);
        call.need.push_back(nd);
    }
    const size_t nsub = call.cut.size() - 1;
    call.jobs.resize(nsub);
    for (size_t k = 0; k < nsub; k++) {
        RollJob& j = call.jobs[k];
        j.call = &call;
        j.k = (uint32_t)k;
        const uint32_t u0 = call.cut[k], nu = call.cut[k + 1] - u0;
        j.rel.resize((size_t)nu + 1);
        j.oo.assign((size_t)nu + 1, 0);
        for (uint32_t i = 0; i <= nu; i++) j.rel[i] = unit_off[u0 + i] - unit_off[u0];
    }
    {
        std::unique_lock<std::mutex> lk(E->m);
        for (size_t k = 0; k < nsub; k++) E->q_stage.push_back(&call.jobs[k]);
        E->cv.notify_all();
        E->cv.wait(lk, [&] { return call.done == nsub; });
    }
    c->last_path = call.last_path;
    c->last_batches = (int)nsub;
    if (call.fail) {
        // device memory exhausted in a lane or a slot (eight lanes' scratch is more than one context's): not an error of the request —
        // the caller's older paths (one context, smaller footprint) serve the call, and only if they cannot does it go back to the host
        if (call.st == KC_ERR_UNSUPPORTED) { c->err.clear(); return KC_ERR_UNSUPPORTED; }
        c->err = call.err;
        return call.st;
    }
    out_off[n_units] = call.pos;
    if (call.trace) fprintf(stderr, "[kc roll] call of %zu sub-batches done at %.1f ms (produced %llu)\n", nsub, call.ms(), (unsigned long long)call.pos);
    return KC_OK;
}

}  // namespace kci
