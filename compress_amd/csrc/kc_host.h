#pragma once
// kc_host.h — internal header of the host side of the C ABI (include/kcgpu.h): the context, its scratch buffers and the batch
// records shared by the translation units kc_ctx.cpp (options, context), kc_batch.cpp (the zstd device pipeline), kc_zstd_host.cpp
// (host-buffer entry points), kc_jobs.cpp (WithConcurrentBlocks), kc_s2_api.cpp (S2) and kc_hook.cpp (the WriterCustomEncoder hook).
// Not installed: the boundary is include/kcgpu.h.
// There is deliberately NO CPU fallback in this library: when the device path cannot serve a
// request it returns KC_ERR_UNSUPPORTED / KC_ERR_NO_DEVICE and the caller (the Go shim)
// decides to use the reference's own encoder.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "../../include/kcgpu.h"
#include "kc_kernels.h"


namespace kci {

const int kMinWindowSize = 1 << 10;          // zstd/decoder_options.go MinWindowSize
const int kMaxWindowSize = 1 << 29;          // zstd MaxWindowSize
const int kMaxCompressedBlockSize = 128 << 10;  // zstd/blockdec.go:40

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// A batch in flight between its two halves: everything up to and including the match finder is enqueued by batch_begin
// (no host synchronisation unless a dictionary has to be staged), the entropy stage, the speculation check, compaction and
// the copy of the offsets by batch_end.  Two contexts on two streams can therefore pipeline consecutive batches: the match
// finder of batch i+1 (random-access HBM bound, waves mostly parked) runs under the entropy stage of batch i.
struct Pending {
    kc_zstd_opts o;
    KcMatchParams mp;
    KcEntropyParams ep;
    std::vector<uint64_t> unit_off;
    uint32_t n_units = 0;
    uint8_t* d_dst = nullptr;
    uint64_t need = 0;   // sum of the units' staging slots: what dst must hold (checked again when _end_at names another place)
    bool fed = false;    // chunk-fed batch: its frames went to dst while the chunks ran — _end_at cannot move them
    int bs = 0;
    bool k2prof = false;
};


// The source of a batch that arrives in chunks over PCIe (host path): the checksum + match-finder kernels of a chunk are
// launched on their own stream as soon as the chunk's H2D copy has landed, so the transfers run under the kernels of the
// chunks before while all units of the batch end up in flight together.
// A unit is parsed by one lane group, block after block: positions are 32-bit with a few tag bits to spare.
static const uint64_t KC_MAX_UNIT_BYTES = (uint64_t)1 << 30;

struct ChunkFeed {
    std::vector<uint32_t> cut;                   // unit index boundaries, nchunk + 1
    std::vector<hipEvent_t> landed;              // recorded on the copy stream behind chunk k's H2D
    std::vector<hipEvent_t> done;                // recorded behind chunk k's kernels
    std::vector<hipStream_t> streams;            // kernels of chunk k run on streams[k % size]
    std::function<bool(size_t)> wait_recorded;   // blocks until landed[k] HAS BEEN RECORDED (waiting on an unrecorded event is a no-op)
    // chunk k's streams also run its entropy stage and compact its frames to d_dst + stage_off[cut[k]] (the chunk's worst-case
    // region), local frame offsets in loc_off[cut[k] + k ... cut[k+1] + k]; the caller drains chunk by chunk and finishes with
    // feed_finish() instead of batch_end()
    uint64_t* loc_off = nullptr;
};

// Host-side layout of one device batch.  Lives in the context: the H2D copies of its arrays are asynchronous, so the arrays must
// outlive batch_begin (they are overwritten by the next batch of the same context, after batch_end synchronised the stream).
struct Plan {
    uint32_t n_units = 0, n_blocks = 0;
    std::vector<uint32_t> blk0;       // n+1
    std::vector<uint64_t> stage_off;  // n+1
    std::vector<uint32_t> blk_start, unit_flags;  // streams with Flush points: per block / per unit (see KcMatchParams)
    std::vector<uint64_t> stage64;    // S2: n+1 staging slot offsets (64-byte aligned)
    std::vector<uint64_t> rel_off;    // n+1 unit offsets relative to the batch base
    uint32_t seq_stride = 0, lit_stride = 0;
    uint64_t max_unit_bytes = 0;      // longest unit of the batch
};

}  // namespace kci
using namespace kci;

// Tunables of one context.  Initialised ONCE, in kc_ctx_create, from the KC_* environment variables listed in
// include/kcgpu.h (kc_option); changed afterwards only through kc_ctx_set_option.  No entry point reads the environment.
struct KcCfg {
    int64_t match_path = KC_PATH_AUTO;
    int64_t zfast_lds_max_units = 768;    // auto: SpeedFastest batches up to this many units take the LDS-table kernel (profiles/r03_crossover_zfast.csv)
    int64_t s2_lds_max_blocks = 1280;     // auto: s2.Encode / EncodeSnappy batches up to this many blocks (profiles/r04_crossover_s2.csv: the LDS kernel 2.8 ms per 256 blocks, the HBM kernel ~16.5 ms up to 2 048)
    int64_t spec_w0 = -1, spec_grow = -1; // HBM-table kernels: speculation width after a match / growth policy; -1 = the per-level defaults
    int64_t lds_spec_w0 = 16;             // SpeedFastest LDS-table kernel: probe steps per round after a match (doubles on a miss up to 64); 0 = units up to 128 KiB without history through the instantiation with the source in a 64 KiB LDS ring (untagged 17-bit table): same time on text
    int64_t s2_lds_spec_w0 = 0;           // S2 LDS-table kernel: the same; blocks held in LDS: 0 = the fused wave-uniform step (two LDS round trips per step), 1 = its first form
    int64_t host_serial = 0, host_pipe_mib = 0, host_overlap_min_mib = -1, host_copy_threads = 0, host_trace = 0;
    int64_t host_roll = 1;                // large host-buffer calls go through the device's rolling pipeline (kc_roll.cpp); 0: round 5's one-batch chunk-fed path
    int64_t host_roll_mib = 0;            // rolling pipeline: sub-batch size (0: a quarter of the call's input — SpeedBetter and S2: half —, 64 MiB .. 1 GiB)
    std::vector<uint64_t> host_chunks;    // chunk-fed host path: chunk sizes in bytes (empty: a quarter of the batch each)
    int64_t k2_prof = 0;
    int64_t hook_wait_us = 0, hook_batch = 256, hook_lanes = 4;
    int64_t hook_host_first = -1;         // kc_s2_encode_block: callers the host's built-in encoder is assumed to serve at a time (they get -1, "use the built-in");
                                          // only callers beyond that go to the device.  -1: the host's hardware threads; 0: every caller to the device
    int64_t test_feed_redo = 0;           // diagnostics: force the chunk-fed path's re-encode fallback
    int64_t better_dict_epoch = 0;        // SpeedBetterCompression with a dictionary: epoch-stamped tables + shared dictionary table instead of the per-batch copy
    int64_t s2_variant = 0;               // S2 levels 0 / 2: 0 = the portable Go encoders' bytes, 1 = the amd64 assembly encoders' bytes
    int64_t best_slots = 6144;            // SpeedBestCompression: table slots (34 MiB each) = units encoded at a time (round 6: 2048 -> 6144 = 204 GiB when a batch has that many units and the device the room — the level is one unit's latency whatever is in flight: 1.49 -> 3.17 GB/s, gpurun_out/r7a)
    int64_t zfast_epoch = 1;              // SpeedFastest HBM-table kernel without a dictionary: epoch-stamped tables instead of clearing 128 KiB per unit per batch
    int64_t zfast_xseg_k = 0;             // SpeedFastest HBM-table kernel, tuned form: probe rounds cross skip-segment boundaries once (s - nextEmit) >> 5 reaches this (0: always)
    int64_t zfast_variant = -1;           // SpeedFastest HBM-table kernel: 0 the plain form, 1 the form for input without matches (cross-segment rounds + empty-group
                                          // filter), -1 (default) chosen per batch: the tuned form when the context's previous batch did not compress (ratio >= 0.98)
    int64_t zfast_filter = 1;             // SpeedFastest HBM-table kernel: "nothing written there yet" filter in the idle sequence buffer (units without a sequence so far)
    int64_t xxh_fin_mode = 1;             // kc_xxh64_fin_kernel: how the payload of raw-only frames is stored (KcXxhFinParams.mode)
    int64_t zfast_prescan = -1;           // SpeedFastest: the no-match pre-scan (kc_zstd_prescan.hip): 0 off, 1 on, -1 when the previous batch did not compress
    int64_t job_prime = 1;                // jobs of a WithConcurrentBlocks stream: tables primed from the overlap prefix on the device (0: on the host)
    int64_t fuse_raw_xxh = 1;             // frames made of raw blocks only: checksum and payload copy in one pass over the source (kc_xxh64_fin_kernel)
};

struct kc_ctx {
    KcCfg cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t s2_pend_n = 0;  // blocks of the S2 batch between kc_s2_encode_blocks_lvl_dev_begin and _end_at (0: none in flight)
    // lanes of the rolling host pipeline (kc_roll.cpp) only: the S2 HBM-table path zeroes the table arena for the NEXT batch on
    // stream2 as soon as this batch's encoder is done with it (behind ev[1]) - under the partner lane's encoder instead of 4 ms in
    // front of its own (the arena is 64 KiB per block whatever the block holds).  preclear_bytes of tables.p are then zero once
    // ev_preclear has fired; every other user of the arena waits for the event and drops the claim (tables_claim).
    bool lane_preclear = false;
    hipEvent_t ev_preclear = nullptr;
    void* preclear_ptr = nullptr;
    size_t preclear_bytes = 0;
    hipStream_t stream2 = nullptr;   // KC_OPT_STAGE2_STREAM: device-resident zstd batches run the entropy stage and everything behind it here (null: on `stream`)
    bool own_stream = false;
    std::string err;
    bool oom = false;                // the last failure was "device memory exhausted" (KC_ERR_UNSUPPORTED to the caller): the batch cutters retry at half the scratch budget on THIS flag, never on the message text
    hipDeviceProp_t prop;
    DevBuf blk_start, unit_flags, redo_blk, pop_blk;
    DevBuf unit_off, unit_blk0, stage_off, seqs, aux, lits, meta, stage, out_size, xxh, redo, popmask, unit_list, out_off,
        predef, errflag, tmp_src, tmp_dst, tables, prof, work, work_off, dictbuf, proto, dicthuf;
    bool predef_ready = false;
    // SpeedBetterCompression epoch stamps: what the table arena holds (kc_zstd_match_better.hip)
    int tab_owner = 0;          // 1: the arena holds better-level tables of tab_units units, stamped up to tab_ep, written with tab_pb position bits
    int tab_pb = 0;
    uint32_t tab_units = 0, tab_ep = 0, better_epoch_now = 0, fast_epoch_now = 0;  // (tab_owner 2: SpeedFastest tables, kc_zstd_match.hip)
    void* tab_ptr = nullptr;
    uint64_t proto_key = 0;     // the dictionary tables in c->proto were built for this (content hash, level, position bits, stamp mode)
    void* proto_ptr = nullptr;  // ... in these allocations (a re-grown buffer is rebuilt)
    void* dictbuf_ptr = nullptr;
    DevBuf best_tables, best_cur, best_cost;  // SpeedBestCompression: persistent table slots, their position-space counters, the bit costs
    uint32_t best_n = 0;                      // slots allocated (and zeroed) so far
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [6]: batch prepared (chunk-fed launches wait on it); [7]: tables prepared
    kc_timings last = {0, 0, 0, 0, 0, 0};
    size_t max_batch_bytes = (size_t)8 << 30;  // input bytes per device batch (scratch is ~6x this for full-size units)
    uint64_t max_scratch_bytes = (uint64_t)160 << 30;  // scratch per device batch (tables + per-block strides), further capped by the free device memory
    int stream_mode = 0;             // set for the duration of kc_zstd_encode_streams_dev
    std::thread job;                 // kc_*_submit: the host-buffer call running on its own thread until kc_wait
    bool job_active = false;
    kc_status job_status = KC_OK;
    const uint64_t* cut_off = nullptr;  // streams with Flush points (kc_zstd_encode_streams_cuts*): per stream the range of its cuts,
    const uint64_t* cuts = nullptr;     // the cut positions (bytes written before the Flush), for the duration of the call
    uint32_t cut_unit0 = 0;             // index of the running batch's first unit in cut_off
    // a batch whose units are the jobs of ONE WithConcurrentBlocks stream (kc_zstd_encode_jobs), for the duration of that call:
    const uint32_t* job_hist = nullptr;     // host, per unit: bytes of overlap prefix in front of the unit in the source buffer
    const uint32_t* job_flags = nullptr;    // host, per unit: bit 0 = final job
    const uint8_t* job_tables = nullptr;    // host or null: the units' tables primed from their prefixes (ResetPrefix), device entry format
    bool job_primed = false;                // the units' tables start primed from their prefixes: by kc_zstd_prime_kernel, or from job_tables
    DevBuf d_job_hist, d_job_flags, rawdef, unit_raw;
    DevBuf unit_done, probe_rel;         // no-match pre-scan (kc_zstd_prescan.hip): per-unit verdicts; the probe positions of one block
    int probe_bs = 0;                    // block size probe_rel was built for
    uint32_t probe_n = 0;
    bool prescan_ran = false;            // the batch in flight ran the pre-scan (its verdicts are counted at the batch's end)
    int64_t last_prescan_units = 0;      // units of the last batch the pre-scan settled
    std::vector<uint32_t> job_redo_list;    // units of the speculation re-run in progress (their tables are re-primed)
    void* pend = nullptr;            // batch between kc_zstd_encode_units_dev_begin and _end (Pending)
    kc_ctx* chain_after = nullptr;   // pipelining: this context's match finder waits for that context's last one
    void* hpipe = nullptr;           // pinned staging ring + streams of the pipelined host path (HostPipe), created on first use
    Plan plan;                       // layout arrays of the batch in flight (sources of asynchronous H2D copies)
    std::once_flag hook_once;        // kc_s2_encode_block: micro-batcher of concurrent callers (S2Hook), created on first use
    void* hook = nullptr;
    bool ev7_valid = false;          // ev[7] was recorded for the batch in flight
    int last_batches = 0;            // device batches the last zstd / S2 _dev call was cut into (scratch budget)
    int last_path = 0;               // KC_PATH_HBM / KC_PATH_LDS: what the last batch's match finder / S2 encoder ran on
    bool last_incompressible = false;  // the previous zstd batch of this context came out at >= 98 % of its input (picks the match finder's form)
    std::vector<uint8_t> up_unit_off, up_blk0, up_stage_off;  // zstd batches: what unit_off / unit_blk0 / stage_off hold on the device ...
    const void* up_ptr[3] = {nullptr, nullptr, nullptr};     // ... and in which allocation (re-uploaded only when they change)
};

// s2.Encode takes any input MaxEncodedLen accepts (~4 GiB, s2/encode.go:29-56: above 64 KiB encodeBlockGo, on amd64 encodeBlockAsm from
// 4 MiB on); the device kernels keep positions in 31 bits and sizes in 32: blocks up to 1 GiB are served, larger ones are refused
// (KC_ERR_UNSUPPORTED: the Go shim then calls the reference encoder).  s2.Writer never cuts blocks above 4 MiB (s2.maxBlockSize).
#define KC_S2_MAX_BLOCK ((uint64_t)1 << 30)
#define KC_S2_MAX_FRAMED_BLOCK ((uint64_t)4 << 20)  // s2.maxBlockSize: the largest block of a framed stream

namespace kci {

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                            \
            return KC_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

inline kc_status ensure(kc_ctx* c, DevBuf& b, size_t bytes) {
    if (b.cap >= bytes) return KC_OK;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + (bytes >> 3) + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e == hipErrorOutOfMemory) {  // try the exact size before giving up
        (void)hipGetLastError();
        want = bytes + 256;
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        if (e == hipErrorOutOfMemory) {
            // not an error of the request: the device path cannot serve it now, the caller uses the reference encoder
            c->err = "device memory exhausted (" + std::to_string(want >> 20) + " MiB of scratch wanted)";
            c->oom = true;
            return KC_ERR_UNSUPPORTED;
        }
        c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return KC_ERR_HIP;
    }
    b.cap = want;
    return KC_OK;
}

// Before anything but the S2 pre-clear's owner touches c->tables on stream st: wait for a pending pre-clear, forget it.
inline kc_status tables_claim(kc_ctx* c, hipStream_t st) {
    if (c->preclear_bytes != 0) {
        c->preclear_bytes = 0;
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_preclear, 0));
    }
    return KC_OK;
}

inline int bitsLen32(uint32_t v) { return v == 0 ? 0 : 32 - __builtin_clz(v); }

void s2_hook_free(void* h);  // S2Hook (kc_s2_encode_block's micro-batcher), defined with it
void host_pipe_free(void* h); // HostPipe (kc_zstd_encode_units / kc_s2_encode_blocks), defined with it

}  // namespace kci
using namespace kci;

// ---- functions shared between the translation units (defined where the comment says) ----
namespace kci {
// kc_batch.cpp: the zstd device pipeline
kc_status check_supported(kc_ctx* c, const kc_zstd_opts* o);
size_t match_table_bytes(int level);
kc_status launch_match(kc_ctx* c, const KcMatchParams& mp, const uint64_t* unit_off, uint32_t n_units, uint32_t n_launch, int bs, hipStream_t st, int level);
uint32_t plan_stream_blocks(uint64_t bs, uint64_t len, const uint64_t* cuts, uint64_t n_cuts, std::vector<uint32_t>* starts, uint32_t* flags);
kc_status batch_begin(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src_base, const uint64_t* unit_off, uint32_t n_units,
                      uint8_t* d_dst, uint64_t dst_cap, ChunkFeed* feed = nullptr);
kc_status batch_end(kc_ctx* c, uint64_t* out_off_host, uint64_t* produced);
uint64_t zstd_unit_scratch(const kc_zstd_opts* o, uint64_t len, uint64_t n_cuts = 0);
uint64_t scratch_budget(kc_ctx* c);
kc_status feed_finish(kc_ctx* c, bool* redo_needed);
kc_status run_batch(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src_base, const uint64_t* unit_off, uint32_t n_units,
                    uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off_host, uint64_t* produced);
kc_status validate_units(kc_ctx* c, const kc_zstd_opts* o, const uint64_t* unit_off, uint32_t n_units);
// kc_roll.cpp: the rolling host pipeline (one engine per device).  enc runs one sub-batch on a lane context of the engine's
// (synchronous, device pointers, offsets relative to the sub-batch); max_out bounds one unit's output.  KC_ERR_UNSUPPORTED with an
// empty error text: no engine on this device - the caller's older paths serve the call.
typedef std::function<kc_status(kc_ctx* lane, const uint8_t* d_in, const uint64_t* rel_off, uint32_t n, uint8_t* d_out, uint64_t cap, uint64_t* out_off_rel)> RollEncFn;
kc_status host_rolling(kc_ctx* c, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units, uint8_t* dst, uint64_t dst_cap,
                       uint64_t* out_off, const RollEncFn& enc, const std::function<uint64_t(uint64_t)>& max_out, uint64_t sub_bytes = 0);  // sub_bytes 0: host_roll_sub_bytes
uint64_t host_roll_sub_bytes(const kc_ctx* c, uint64_t total);
kc_status host_roll_trim(int device);  // kc_device_trim: free the idle engine's device slots and its lanes' scratch
}  // namespace kci
