// kc_api.cpp — host side of the C ABI declared in include/kcgpu.h: option resolution
// (mirrors zstd/encoder_options.go), device scratch management, kernel orchestration.
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


namespace {

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

}  // namespace

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
    std::vector<uint64_t> host_chunks;    // chunk-fed host path: chunk sizes in bytes (empty: a quarter of the batch each)
    int64_t k2_prof = 0;
    int64_t hook_wait_us = 0, hook_batch = 256, hook_lanes = 4;
    int64_t test_feed_redo = 0;           // diagnostics: force the chunk-fed path's re-encode fallback
    int64_t better_dict_epoch = 0;        // SpeedBetterCompression with a dictionary: epoch-stamped tables + shared dictionary table instead of the per-batch copy
    int64_t s2_variant = 0;               // S2 levels 0 / 2: 0 = the portable Go encoders' bytes, 1 = the amd64 assembly encoders' bytes
    int64_t best_slots = 2048;            // SpeedBestCompression: table slots (34 MiB each) = units encoded at a time
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

namespace {

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                            \
            return KC_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

kc_status ensure(kc_ctx* c, DevBuf& b, size_t bytes) {
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

inline int bitsLen32(uint32_t v) { return v == 0 ? 0 : 32 - __builtin_clz(v); }

void s2_hook_free(void* h);  // S2Hook (kc_s2_encode_block's micro-batcher), defined with it
void host_pipe_free(void* h); // HostPipe (kc_zstd_encode_units / kc_s2_encode_blocks), defined with it

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------
// options (zstd/encoder_options.go)
// ---------------------------------------------------------------------------------------
void kc_zstd_opts_default(kc_zstd_opts* o) {  // setDefault :36-48
    memset(o, 0, sizeof(*o));
    o->level = KC_SPEED_DEFAULT;
    o->window_size = 8 << 20;
    o->block_size = kMaxCompressedBlockSize;
    o->crc = 1;
    o->single = -1;
    o->full_zero = 1;
    o->no_entropy = 0;
    o->all_lit_entropy = 0;
    o->low_mem = 0;
    o->dict_offsets[0] = 1; o->dict_offsets[1] = 4; o->dict_offsets[2] = 8;
    o->concurrent = 0;
}

int kc_zstd_opts_concurrency(kc_zstd_opts* o, int n) {  // WithEncoderConcurrency :76-87
    if (n < 1) return KC_ERR_BAD_ARG;
    o->concurrent = n;
    return KC_OK;
}

int kc_zstd_opts_level(kc_zstd_opts* o, int l) {  // WithEncoderLevel :236-266
    if (l < KC_SPEED_FASTEST || l > 4) return KC_ERR_BAD_ARG;  // speedNotSet < l < speedLast
    o->level = l;
    if (!o->custom_window) {
        switch (l) {
        case KC_SPEED_FASTEST:
            o->window_size = 4 << 20;
            if (!o->custom_block) o->block_size = 1 << 16;
            break;
        default:
            o->window_size = 8 << 20;
            break;
        }
    }
    if (!o->custom_alent) o->all_lit_entropy = l > KC_SPEED_DEFAULT;
    return KC_OK;
}

int kc_zstd_opts_window(kc_zstd_opts* o, int n) {  // WithWindowSize :110-133
    if (n < kMinWindowSize || n > kMaxWindowSize || (n & (n - 1)) != 0) return KC_ERR_BAD_ARG;
    o->window_size = n;
    o->custom_window = 1;
    if (o->block_size > o->window_size) {
        o->block_size = o->window_size;
        o->custom_block = 1;
    }
    return KC_OK;
}
int kc_zstd_opts_crc(kc_zstd_opts* o, int b) { o->crc = b != 0; return KC_OK; }
int kc_zstd_opts_zero_frames(kc_zstd_opts* o, int b) { o->full_zero = b != 0; return KC_OK; }
int kc_zstd_opts_no_entropy(kc_zstd_opts* o, int b) { o->no_entropy = b != 0; return KC_OK; }
int kc_zstd_opts_all_lit_entropy(kc_zstd_opts* o, int b) { o->custom_alent = 1; o->all_lit_entropy = b != 0; return KC_OK; }
int kc_zstd_opts_single_segment(kc_zstd_opts* o, int b) { o->single = b != 0; return KC_OK; }
int kc_zstd_opts_dict_raw(kc_zstd_opts* o, uint32_t id, const uint8_t* content, uint64_t len) {  // :398-406
    if (len > ((uint64_t)1 << 31)) return KC_ERR_BAD_ARG;
    o->dict_id = id;
    o->dict = content;
    o->dict_len = len;
    o->dict_offsets[0] = 1; o->dict_offsets[1] = 4; o->dict_offsets[2] = 8;  // offsets: [3]int{1, 4, 8}, no litEnc
    o->dict_huf_len = 0;
    o->dict_huf_log = 0;
    return KC_OK;
}

int64_t kc_zstd_max_encoded_size(const kc_zstd_opts* o, int64_t size) {  // encoder.go:843-873
    int64_t frameHeader = 4 + 2;
    if (o->dict != nullptr || o->dict_id != 0) frameHeader += 4;
    if (size < 256) frameHeader++;
    else if (size < 65536 + 256) frameHeader += 2;
    else if (size < 0x7fffffff) frameHeader += 4;
    else frameHeader += 8;
    if (o->crc) frameHeader += 4;
    const int64_t blocks = (size + o->block_size) / o->block_size;
    return frameHeader + 3 * blocks + size;
}

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
kc_status kc_ctx_create(kc_ctx** out, int device, void* stream) {
    if (!out) return KC_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return KC_ERR_NO_DEVICE;
    kc_ctx* c = new kc_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
        delete c;
        return KC_ERR_NO_DEVICE;
    }
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return KC_ERR_HIP; }
        c->own_stream = true;
    }
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete c; return KC_ERR_HIP; }
    // every tunable is a field of the context with its default in KcCfg; kc_ctx_set_option is the only way to change one (the library
    // reads no environment variable: measurement harnesses map their KC_* variables to options above the C ABI, compress_amd/_lib.py)
    *out = c;
    return KC_OK;
}

kc_status kc_ctx_set_option(kc_ctx* c, int key, int64_t v) {
    if (!c) return KC_ERR_BAD_ARG;
    KcCfg& g = c->cfg;
    switch (key) {
        case KC_OPT_MATCH_PATH: if (v < KC_PATH_AUTO || v > KC_PATH_LDS) return KC_ERR_BAD_ARG; g.match_path = v; break;
        case KC_OPT_ZFAST_LDS_MAX_UNITS: g.zfast_lds_max_units = v; break;
        case KC_OPT_S2_LDS_MAX_BLOCKS: g.s2_lds_max_blocks = v; break;
        case KC_OPT_SPEC_W0: g.spec_w0 = v; break;
        case KC_OPT_SPEC_GROW: g.spec_grow = v; break;
        case KC_OPT_LDS_SPEC_W0: g.lds_spec_w0 = v; break;
        case KC_OPT_S2_LDS_SPEC_W0: g.s2_lds_spec_w0 = v; break;
        case KC_OPT_HOST_SERIAL: g.host_serial = v; break;
        case KC_OPT_HOST_PIPE_MIB: g.host_pipe_mib = v; break;
        case KC_OPT_HOST_OVERLAP_MIN_MIB: g.host_overlap_min_mib = v; break;
        case KC_OPT_HOST_COPY_THREADS: g.host_copy_threads = v; break;
        case KC_OPT_HOST_TRACE: g.host_trace = v; break;
        case KC_OPT_HOST_CHUNK_MIB: g.host_chunks.clear(); if (v > 0) g.host_chunks.push_back((uint64_t)v << 20); break;
        case KC_OPT_HOST_CHUNK_MIB_APPEND: if (v < 1) return KC_ERR_BAD_ARG; g.host_chunks.push_back((uint64_t)v << 20); break;
        case KC_OPT_K2_PROF: g.k2_prof = v; break;
        case KC_OPT_S2_HOOK_WAIT_US: g.hook_wait_us = v; break;
        case KC_OPT_S2_HOOK_BATCH: g.hook_batch = v < 1 ? 1 : v; break;
        case KC_OPT_S2_HOOK_LANES: g.hook_lanes = v < 1 ? 1 : (v > 8 ? 8 : v); break;
        case KC_OPT_TEST_FEED_REDO: g.test_feed_redo = v; break;
        case KC_OPT_MAX_SCRATCH_MIB: if (v < 1) return KC_ERR_BAD_ARG; c->max_scratch_bytes = (uint64_t)v << 20; break;
        case KC_OPT_BEST_SLOTS: if (v < 1 || v > 8192) return KC_ERR_BAD_ARG; g.best_slots = v; break;
        case KC_OPT_S2_VARIANT: if (v != KC_S2_VARIANT_GO && v != KC_S2_VARIANT_AMD64) return KC_ERR_BAD_ARG; g.s2_variant = v; break;
        case KC_OPT_BETTER_DICT_EPOCH: g.better_dict_epoch = v != 0; break;
        case KC_OPT_ZFAST_EPOCH: g.zfast_epoch = v != 0; break;
        case KC_OPT_ZFAST_XSEG_K: if (v < 0) return KC_ERR_BAD_ARG; g.zfast_xseg_k = v > (1 << 30) ? (1 << 30) : v; break;
        case KC_OPT_FUSE_RAW_XXH: g.fuse_raw_xxh = v != 0; break;
        case KC_OPT_ZFAST_FILTER: g.zfast_filter = v != 0; break;
        case KC_OPT_ZFAST_VARIANT: if (v < -1 || v > 1) return KC_ERR_BAD_ARG; g.zfast_variant = v; break;
        case KC_OPT_ZFAST_PRESCAN: if (v < -1 || v > 1) return KC_ERR_BAD_ARG; g.zfast_prescan = v; break;
        case KC_OPT_XXH_FIN_MODE: if (v < 0 || v > 3) return KC_ERR_BAD_ARG; g.xxh_fin_mode = v; break;
        case KC_OPT_JOB_PRIME: g.job_prime = v != 0; break;
        case KC_OPT_STAGE2_STREAM: if (c->pend) return KC_ERR_BAD_ARG; c->stream2 = (hipStream_t)(intptr_t)v; break;
        default: return KC_ERR_BAD_ARG;
    }
    return KC_OK;
}

int64_t kc_ctx_get_option(const kc_ctx* c, int key) {
    if (!c) return -1;
    const KcCfg& g = c->cfg;
    switch (key) {
        case KC_OPT_MATCH_PATH: return g.match_path;
        case KC_OPT_ZFAST_LDS_MAX_UNITS: return g.zfast_lds_max_units;
        case KC_OPT_S2_LDS_MAX_BLOCKS: return g.s2_lds_max_blocks;
        case KC_OPT_SPEC_W0: return g.spec_w0;
        case KC_OPT_SPEC_GROW: return g.spec_grow;
        case KC_OPT_LDS_SPEC_W0: return g.lds_spec_w0;
        case KC_OPT_S2_LDS_SPEC_W0: return g.s2_lds_spec_w0;
        case KC_OPT_HOST_SERIAL: return g.host_serial;
        case KC_OPT_HOST_PIPE_MIB: return g.host_pipe_mib;
        case KC_OPT_HOST_OVERLAP_MIN_MIB: return g.host_overlap_min_mib;
        case KC_OPT_HOST_COPY_THREADS: return g.host_copy_threads;
        case KC_OPT_HOST_TRACE: return g.host_trace;
        case KC_OPT_HOST_CHUNK_MIB: return g.host_chunks.empty() ? 0 : (int64_t)(g.host_chunks[0] >> 20);
        case KC_OPT_K2_PROF: return g.k2_prof;
        case KC_OPT_S2_HOOK_WAIT_US: return g.hook_wait_us;
        case KC_OPT_S2_HOOK_BATCH: return g.hook_batch;
        case KC_OPT_S2_HOOK_LANES: return g.hook_lanes;
        case KC_OPT_TEST_FEED_REDO: return g.test_feed_redo;
        case KC_OPT_MAX_SCRATCH_MIB: return (int64_t)(c->max_scratch_bytes >> 20);
        case KC_OPT_BEST_SLOTS: return g.best_slots;
        case KC_OPT_S2_VARIANT: return g.s2_variant;
        case KC_OPT_BETTER_DICT_EPOCH: return g.better_dict_epoch;
        case KC_OPT_ZFAST_EPOCH: return g.zfast_epoch;
        case KC_OPT_ZFAST_XSEG_K: return g.zfast_xseg_k;
        case KC_OPT_FUSE_RAW_XXH: return g.fuse_raw_xxh;
        case KC_OPT_ZFAST_FILTER: return g.zfast_filter;
        case KC_OPT_ZFAST_VARIANT: return g.zfast_variant;
        case KC_OPT_ZFAST_PRESCAN: return g.zfast_prescan;
        case KC_OPT_XXH_FIN_MODE: return g.xxh_fin_mode;
        case KC_OPT_JOB_PRIME: return g.job_prime;
        case KC_OPT_STAGE2_STREAM: return (int64_t)(intptr_t)c->stream2;
        case KC_OPT_LAST_PATH: return c->last_path;
        case KC_OPT_LAST_PRESCAN_UNITS: return c->last_prescan_units;
        case KC_OPT_LAST_BATCHES: return c->last_batches;
        default: return -1;
    }
}

void kc_ctx_destroy(kc_ctx* c) {
    if (!c) return;
    if (c->job_active && c->job.joinable()) c->job.join();
    (void)hipSetDevice(c->device);
    DevBuf* bufs[] = {&c->unit_off, &c->unit_blk0, &c->stage_off, &c->seqs, &c->aux, &c->lits, &c->meta, &c->stage, &c->out_size, &c->xxh,
                      &c->redo, &c->popmask, &c->unit_list, &c->out_off, &c->blk_start, &c->unit_flags, &c->redo_blk, &c->pop_blk, &c->predef, &c->errflag, &c->tmp_src, &c->tmp_dst, &c->tables, &c->prof, &c->work, &c->work_off, &c->dictbuf, &c->proto, &c->dicthuf,
                      &c->d_job_hist, &c->d_job_flags, &c->rawdef, &c->unit_raw, &c->unit_done, &c->probe_rel, &c->best_tables, &c->best_cur, &c->best_cost};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->pend) { delete (Pending*)c->pend; c->pend = nullptr; }
    if (c->hook) { s2_hook_free(c->hook); c->hook = nullptr; }
    if (c->hpipe) { host_pipe_free(c->hpipe); c->hpipe = nullptr; }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* kc_last_error(const kc_ctx* c) { return c ? c->err.c_str() : "null context"; }

kc_status kc_device_info(const kc_ctx* c, int32_t* n_cu, int32_t* lds_per_cu, int32_t* clock_khz, char* name, size_t name_cap) {
    if (!c) return KC_ERR_BAD_ARG;
    if (n_cu) *n_cu = c->prop.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (int32_t)c->prop.maxSharedMemoryPerMultiProcessor;
    if (clock_khz) *clock_khz = c->prop.clockRate;
    if (name && name_cap) { strncpy(name, c->prop.gcnArchName, name_cap - 1); name[name_cap - 1] = 0; }
    return KC_OK;
}

kc_status kc_last_timings(const kc_ctx* c, kc_timings* t) {
    if (!c || !t) return KC_ERR_BAD_ARG;
    *t = c->last;
    return KC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// zstd device pipeline
// ---------------------------------------------------------------------------------------
namespace {


// Host-side construction of the dictionary-primed tables of betterFastEncoderDict.Reset
// (zstd/enc_better.go:1114-1183) in the device entry format (position+1 | tag << pos_bits).
void build_better_dict_tables(const uint8_t* dict, size_t len, int pos_bits, uint8_t* out, int reserved_bits = 0) {
    const int TB = (32 - pos_bits - reserved_bits) > 16 ? 16 : (32 - pos_bits - reserved_bits);  // reserved: the epoch stamp's bits (kc_zstd_match_better.hip)
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    auto mk = [&](uint64_t pos, uint32_t val) -> uint32_t { return ((uint32_t)pos + 1u) | (tagOf(val) << pos_bits); };
    auto ld64h = [&](size_t i) -> uint64_t { uint64_t v; memcpy(&v, dict + i, 8); return v; };
    uint32_t* ltab = (uint32_t*)out;  // pairs {offset, prev}
    uint32_t* stab = (uint32_t*)(out + ((size_t)8 << 19));
    if (len < 8) return;
    // short table: every position i, i+1, i+2, i+3 for i stepping by 4 while i < len-8 (:1126-1152)
    for (size_t i = 0; i + 8 < len; i += 4) {
        const uint64_t cv = ld64h(i);
        for (int k = 0; k < 4; k++) {
            const uint64_t v = cv >> (8 * k);
            const uint32_t h = (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 13));
            stab[h] = mk(i + k, (uint32_t)v);
        }
    }
    // long table: every position 0 .. len-9, chained (:1161-1183)
    for (size_t i = 0; i + 8 < len || i == 0; i++) {
        const uint64_t cv = ld64h(i);
        const uint32_t h = (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 19));
        const uint32_t old = ltab[2 * h];
        ltab[2 * h] = mk(i, (uint32_t)cv);
        ltab[2 * h + 1] = old;
        if (i + 8 >= len) break;
    }
}

// fastEncoderDict.Reset (zstd/enc_fast.go:813-845): 2^15 table, 6-byte hash, positions i and i+1 for i stepping by 2.
// Also the SHORT table of doubleFastEncoderDict, which embeds fastEncoderDict and keeps this priming although its
// lookups use the 5-byte hash (enc_dfast.go:1053-1056).
void build_fast_dict_table(const uint8_t* dict, size_t len, int pos_bits, uint32_t* tab) {
    const int TB = (32 - pos_bits) > 16 ? 16 : (32 - pos_bits);
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    auto mk = [&](uint64_t pos, uint32_t val) -> uint32_t { return ((uint32_t)pos + 1u) | (tagOf(val) << pos_bits); };
    if (len < 8) return;
    for (size_t i = 0; i + 8 < len; i += 2) {
        uint64_t cv;
        memcpy(&cv, dict + i, 8);
        const uint32_t h0 = (uint32_t)(((cv << 16) * 227718039650203ULL) >> (64 - 15));
        const uint32_t h1 = (uint32_t)((((cv >> 8) << 16) * 227718039650203ULL) >> (64 - 15));
        tab[h0] = mk(i, (uint32_t)cv);
        tab[h1] = mk(i + 1, (uint32_t)(cv >> 8));
    }
}
// doubleFastEncoderDict.Reset long table (zstd/enc_dfast.go:1060-1083): every position 0 .. len-9, 8-byte hash, 2^17.
void build_dfast_dict_long(const uint8_t* dict, size_t len, int pos_bits, uint32_t* ltab) {
    const int TB = (32 - pos_bits) > 16 ? 16 : (32 - pos_bits);
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    if (len < 8) return;
    for (size_t i = 0; i == 0 || i + 8 < len; i++) {
        uint64_t cv;
        memcpy(&cv, dict + i, 8);
        const uint32_t h = (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 17));
        ltab[h] = ((uint32_t)i + 1u) | (tagOf((uint32_t)cv) << pos_bits);
        if (i + 8 >= len) break;
    }
}

kc_status check_supported(kc_ctx* c, const kc_zstd_opts* o) {
    if (o->level < KC_SPEED_FASTEST || o->level > KC_SPEED_BEST) { c->err = "unknown encoder level"; return KC_ERR_UNSUPPORTED; }
    // (the reference takes any dictionary below 2 GiB as history, zstd/dict.go:27, enc_base.go:160-198; here the dictionary is staged in
    // front of every unit of a batch, which the scratch budget accounts for — 64 MiB keeps positions inside the LDS kernel's field too)
    if (o->dict_len > ((uint64_t)64 << 20)) { c->err = "dictionary larger than 64 MiB not served by the device path"; return KC_ERR_UNSUPPORTED; }
    if (o->block_size < 1024 || o->block_size > kMaxCompressedBlockSize || o->window_size < kMinWindowSize) { c->err = "bad block/window size"; return KC_ERR_BAD_ARG; }
    return KC_OK;
}


// Match finders: sub-wave groups (8 lanes per unit), per-unit hash tables in an HBM arena that is zeroed (or primed from the
// dictionary tables) before every launch.
size_t match_table_bytes(int level) {
    if (level == KC_SPEED_BEST) return 0;  // persistent slots (ensure_best_slots), not per unit
    return level == KC_SPEED_BETTER ? kc_zbetter_table_bytes() : (level == KC_SPEED_DEFAULT ? kc_zdfast_table_bytes() : kc_zfast_table_bytes());
}

// SpeedFastest: which kernel serves a launch of n_launch units.  The LDS-table kernel (one wave per unit, a few ms per unit
// whatever the batch) wins while the units in flight cannot cover the HBM-table kernel's latency; the crossover is measured
// (profiles/r03_crossover_zfast.csv) and set by KC_OPT_ZFAST_LDS_MAX_UNITS; KC_OPT_MATCH_PATH forces a path.  Units (with their
// dictionary history) of 256 KiB and more only fit the HBM path's position field: the HBM kernel takes those, the LDS kernel the rest.
bool zfast_use_lds(const kc_ctx* c, const KcMatchParams& mp, uint32_t n_launch, int level) {
    (void)mp;
    if (level != KC_SPEED_FASTEST) return false;
    if (c->cfg.match_path == KC_PATH_HBM) return false;
    if (c->cfg.match_path == KC_PATH_LDS) return true;
    return (int64_t)n_launch <= c->cfg.zfast_lds_max_units;
}
// with the LDS path chosen: some unit of the batch does not fit its position field and goes through the HBM-table kernel
bool zfast_lds_needs_hbm(const kc_ctx* c, const KcMatchParams& mp) {
    return (uint64_t)mp.hist0 + c->plan.max_unit_bytes > KC_ZFAST_LDS_MAX_UNIT;
}

// per-unit tables of n_launch units: zeroed, or primed from the dictionary tables
// SpeedBestCompression: min(n_launch, KC_OPT_BEST_SLOTS) persistent table slots, zeroed when allocated; a unit starts from whatever
// the slot's earlier units left, past which its position space has moved (kc_zstd_match_best.hip)
kc_status ensure_best_slots(kc_ctx* c, uint32_t n_launch, hipStream_t st) {
    uint32_t want = (uint32_t)std::min<int64_t>((int64_t)n_launch, c->cfg.best_slots);
    if (want < 1) want = 1;
    if (!c->best_cost.p) {
        kc_status s = ensure(c, c->predef, kc_fse_predef_bytes());
        if (s != KC_OK) return s;
        if (!c->predef_ready) {
            kc_launch_fse_predef_init(c->predef.p, st);
            c->predef_ready = true;
        }
        if ((s = ensure(c, c->best_cost, 96 * 4)) != KC_OK) return s;
        kc_launch_zbest_cost(c->predef.p, (int32_t*)c->best_cost.p, st);
    }
    if (want <= c->best_n) return KC_OK;
    // grow in powers of two so that a sequence of growing batches re-allocates a few times at most
    uint32_t n = 1;
    while (n < want) n <<= 1;
    if ((int64_t)n > c->cfg.best_slots) n = (uint32_t)c->cfg.best_slots;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        const uint64_t room = (uint64_t)((double)(fr + c->best_tables.cap) * 0.8) / kc_zbest_table_bytes();
        if (room < 1) { c->err = "device memory exhausted: no room for one SpeedBestCompression table slot (34 MiB)"; c->oom = true; return KC_ERR_UNSUPPORTED; }
        if ((uint64_t)n > room) n = (uint32_t)room;
    } else (void)hipGetLastError();
    if (n <= c->best_n) return KC_OK;
    HIPCHK(c, hipStreamSynchronize(st));
    if (c->best_tables.p) HIPCHK(c, hipFree(c->best_tables.p));
    c->best_tables.p = nullptr;
    c->best_tables.cap = 0;
    c->best_n = 0;
    {   // exactly n slots (ensure() rounds up by an eighth: 4 GiB at 1024 slots)
        const hipError_t e = hipMalloc(&c->best_tables.p, (size_t)n * kc_zbest_table_bytes());
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->best_tables.p = nullptr;
            c->err = "device memory exhausted (" + std::to_string(((size_t)n * kc_zbest_table_bytes()) >> 20) + " MiB of SpeedBestCompression tables wanted)";
            c->oom = true;
            return KC_ERR_UNSUPPORTED;
        }
        c->best_tables.cap = (size_t)n * kc_zbest_table_bytes();
    }
    kc_status s;
    if ((s = ensure(c, c->best_cur, (size_t)8192 * 4)) != KC_OK) return s;
    HIPCHK(c, hipMemsetAsync(c->best_tables.p, 0, (size_t)n * kc_zbest_table_bytes(), st));
    HIPCHK(c, hipMemsetAsync(c->best_cur.p, 0, (size_t)8192 * 4, st));
    c->best_n = n;
    return KC_OK;
}

// SpeedBetterCompression batches whose tables carry epoch stamps instead of being cleared per launch: not the jobs of a
// WithConcurrentBlocks stream (their tables are primed per unit on the host), only while the stamp fits above position and tag, and
// — measured — not with a dictionary: there every lookup has to read the shared dictionary table beside the unit's own bucket, and
// on C5 that costs more (match finder 52.9 -> 63.7 ms per GiB) than the 7.7 ms of copying the dictionary tables it saves
// (KC_OPT_BETTER_DICT_EPOCH = 1 turns it on for measurements).
bool better_epoch_mode(const kc_ctx* c, int level, int pos_bits, int hist0) {
    return level == KC_SPEED_BETTER && !c->job_primed && c->job_hist == nullptr && pos_bits <= 22 && (hist0 == 0 || c->cfg.better_dict_epoch != 0);
}

kc_status prepare_tables(kc_ctx* c, const KcMatchParams& mp, uint32_t n_launch, hipStream_t st, int level) {
    if (level == KC_SPEED_BEST) return ensure_best_slots(c, n_launch, st);
    c->better_epoch_now = 0;
    if (better_epoch_mode(c, level, mp.pos_bits, mp.hist0)) {
        const size_t tbb = match_table_bytes(level);
        kc_status se = ensure(c, c->tables, (size_t)n_launch * tbb);
        if (se != KC_OK) return se;
        const bool fresh = c->tab_owner != 1 || c->tab_pb != mp.pos_bits || n_launch > c->tab_units || c->tab_ep >= 15u || c->tab_ptr != c->tables.p;
        if (fresh) {
            HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tbb, st));
            c->tab_owner = 1;
            c->tab_pb = mp.pos_bits;
            c->tab_units = n_launch;
            c->tab_ep = 1;
            c->tab_ptr = c->tables.p;
        } else {
            c->tab_ep++;
        }
        c->better_epoch_now = c->tab_ep;
        return KC_OK;  // no dictionary copy either: the kernel reads the shared dictionary tables for buckets it has not written
    }
    if (zfast_use_lds(c, mp, n_launch, level) && !zfast_lds_needs_hbm(c, mp) && !c->job_primed) return KC_OK;  // the tables live in LDS (the arena is not touched)
    const bool fastEpoch = level == KC_SPEED_FASTEST && c->cfg.zfast_epoch != 0 && mp.hist0 == 0 && !c->job_primed && mp.pos_bits + KC_ZF_EPOCH_BITS + 4 <= 32;
    if (!fastEpoch) c->tab_owner = 0;  // (whatever follows rewrites the arena)
    c->fast_epoch_now = 0;
    const size_t tb = match_table_bytes(level);
    if (c->job_primed) {  // jobs: slot i holds the table ResetPrefix leaves from the prefix of unit i (or, in a re-run, of unit list[i])
        kc_status sj = ensure(c, c->tables, (size_t)n_launch * tb);
        if (sj != KC_OK) return sj;
        if (c->job_tables == nullptr) {  // primed here, from the prefix bytes already in the source buffer
            HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tb, st));
            KcPrimeParams pp;
            pp.src = mp.src;
            pp.unit_off = mp.unit_off;
            pp.unit_hist = mp.unit_hist;
            pp.unit_list = mp.unit_list;
            pp.unit_base = mp.unit_base;
            pp.n_launch = n_launch;
            pp.level = level;
            pp.pos_bits = mp.pos_bits;
            pp.tables = (uint8_t*)c->tables.p;
            pp.table_bytes = tb;
            kc_launch_zstd_prime(pp, st);
            return KC_OK;
        }
        if (mp.unit_list == nullptr) {
            HIPCHK(c, hipMemcpyAsync(c->tables.p, c->job_tables, (size_t)n_launch * tb, hipMemcpyHostToDevice, st));
            return KC_OK;
        }
        if (c->job_redo_list.size() < n_launch) { c->err = "job re-run without its unit list"; return KC_ERR_INTERNAL; }
        for (uint32_t i = 0; i < n_launch; i++)
            HIPCHK(c, hipMemcpyAsync((uint8_t*)c->tables.p + (size_t)i * tb, c->job_tables + (size_t)c->job_redo_list[i] * tb, tb, hipMemcpyHostToDevice, st));
        return KC_OK;
    }
    kc_status s = ensure(c, c->tables, (size_t)n_launch * tb);
    if (s != KC_OK) return s;
    if (mp.hist0 > 0) kc_launch_bcast((const uint8_t*)c->proto.p, (uint8_t*)c->tables.p, tb, n_launch, st);
    else if (fastEpoch) {
        // SpeedFastest, no dictionary: the entries carry this launch's stamp (kc_zstd_match.hip), so what earlier launches left in
        // the slots reads as empty and nothing is cleared (4 GiB of stores per 4 GiB batch: 0.65 ms) until the stamp wraps
        const bool fresh = c->tab_owner != 2 || c->tab_pb != mp.pos_bits || n_launch > c->tab_units || c->tab_ep >= (1u << KC_ZF_EPOCH_BITS) - 1u || c->tab_ptr != c->tables.p;
        if (fresh) {
            HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tb, st));
            c->tab_owner = 2;
            c->tab_pb = mp.pos_bits;
            c->tab_units = n_launch;
            c->tab_ep = 1;
            c->tab_ptr = c->tables.p;
        } else {
            c->tab_ep++;
        }
        c->fast_epoch_now = c->tab_ep;
    }
    else HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n_launch * tb, st));
    return KC_OK;
}

// the match finder over n_launch units whose tables start at table slot `slot0` (unit = mp.unit_base + i or mp.unit_list[i])
void launch_match_kernel(kc_ctx* c, const KcMatchParams& mp, uint32_t slot0, uint32_t n_launch, hipStream_t st, int level, bool lds = false) {
    if (lds) {
        KcMatchParams ml = mp;
        ml.spec_w0 = (int32_t)c->cfg.lds_spec_w0;
        ml.lds_any_big = c->plan.max_unit_bytes > (uint64_t)131072 ? 1 : 0;
        if (c->job_primed)  // jobs: slot i of the arena holds the table primed from unit i's prefix (prepare_tables)
            kc_launch_zfast_match_lds(ml, (const uint32_t*)((uint8_t*)c->tables.p + (size_t)slot0 * match_table_bytes(level)), 1u << 15, n_launch, st);
        else
            kc_launch_zfast_match_lds(ml, mp.hist0 > 0 ? (const uint32_t*)c->proto.p : nullptr, 0u, n_launch, st);
        c->last_path = KC_PATH_LDS;
        if (zfast_lds_needs_hbm(c, mp)) {  // the units beyond the LDS kernel's position field
            ml = mp;
            ml.lds_split = 1;
            ml.epoch = c->fast_epoch_now;
            ml.xseg_k = (int32_t)c->cfg.zfast_xseg_k;
            ml.empty_filter = c->cfg.zfast_filter != 0;
            ml.tuned = c->cfg.zfast_variant < 0 ? (c->last_incompressible ? 1 : 0) : (int32_t)c->cfg.zfast_variant;
            kc_launch_zfast_match_grp(ml, (uint32_t*)((uint8_t*)c->tables.p + (size_t)slot0 * match_table_bytes(level)), n_launch, st);
        }
        return;
    }
    c->last_path = KC_PATH_HBM;
    if (level == KC_SPEED_BEST) {
        kc_launch_zbest_match(mp, (uint64_t*)c->best_tables.p, (uint32_t*)c->best_cur.p, (const int32_t*)c->best_cost.p, n_launch, c->best_n, st);
        return;
    }
    uint8_t* tab = (uint8_t*)c->tables.p + (size_t)slot0 * match_table_bytes(level);
    if (level == KC_SPEED_BETTER) {
        KcMatchParams mb = mp;
        mb.epoch = c->better_epoch_now;
        mb.proto = (mb.epoch != 0u && mp.hist0 > 0) ? (const uint8_t*)c->proto.p : nullptr;
        kc_launch_zbetter_match_grp(mb, tab, n_launch, mp.hist0 > 0, st);
    }
    else if (level == KC_SPEED_DEFAULT) kc_launch_zdfast_match_grp(mp, (uint32_t*)tab, n_launch, st);
    else {
        KcMatchParams mf = mp;
        mf.epoch = c->fast_epoch_now;
        mf.xseg_k = (int32_t)c->cfg.zfast_xseg_k;
        mf.empty_filter = c->cfg.zfast_filter != 0;
        mf.tuned = c->cfg.zfast_variant < 0 ? (c->last_incompressible ? 1 : 0) : (int32_t)c->cfg.zfast_variant;
        kc_launch_zfast_match_grp(mf, (uint32_t*)tab, n_launch, st);
    }
}

kc_status launch_match(kc_ctx* c, const KcMatchParams& mp, const uint64_t* unit_off, uint32_t n_units, uint32_t n_launch, int bs, hipStream_t st, int level) {
    (void)unit_off; (void)n_units; (void)bs;
    kc_status s = prepare_tables(c, mp, n_launch, st, level);
    if (s != KC_OK) return s;
    HIPCHK(c, hipEventRecord(c->ev[7], st));
    c->ev7_valid = true;
    launch_match_kernel(c, mp, 0, n_launch, st, level, zfast_use_lds(c, mp, n_launch, level));
    return KC_OK;
}

// Blocks of one stream with Flush points (zstd/encoder.go): writeBlocks cuts a block every blockSize bytes after the last Flush
// (:226-253); a Flush that finds nothing buffered does nothing (:552).  Close: a single block still buffered with no header
// written yet is the EncodeAll frame (:272-288), otherwise the stream frame, with an empty last block when nothing is buffered
// (:315-329).  Appends the block starts to *starts; *flags: bit 0 stream frame, bit 1 empty last block.  Returns the block count.
uint32_t plan_stream_blocks(uint64_t bs, uint64_t len, const uint64_t* cuts, uint64_t n_cuts, std::vector<uint32_t>* starts, uint32_t* flags) {
    uint64_t pos = 0, ci = 0, lastStart = 0;
    uint32_t ub = 0;
    while (pos < len) {
        uint64_t e = pos + bs;
        while (ci < n_cuts && cuts[ci] <= pos) ci++;
        if (ci < n_cuts && cuts[ci] < e) e = cuts[ci];
        if (e > len) e = len;
        if (starts) starts->push_back((uint32_t)pos);
        lastStart = pos;
        pos = e;
        ub++;
    }
    const bool flushedAtEnd = n_cuts > 0 && cuts[n_cuts - 1] >= len;
    const bool tailBuffered = ub > 0 && !flushedAtEnd && (len - lastStart) < bs;
    const bool streamU = len > 0 && !(ub == 1 && tailBuffered);
    *flags = (streamU ? 1u : 0u) | ((streamU && !tailBuffered) ? 2u : 0u);
    return ub;
}

kc_status batch_begin(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src_base, const uint64_t* unit_off, uint32_t n_units,
                      uint8_t* d_dst, uint64_t dst_cap, ChunkFeed* feed = nullptr) {
    hipStream_t st = c->stream;
    if (c->pend) { c->err = "a batch is already in flight on this context"; return KC_ERR_BAD_ARG; }
    const int bs = o->block_size;
    Plan& pl = c->plan;
    pl.n_units = n_units;
    pl.blk0.resize(n_units + 1);
    pl.stage_off.resize(n_units + 1);
    pl.rel_off.resize(n_units + 1);
    uint64_t so = 0;
    uint32_t nb = 0;
    const bool irregular = c->cuts != nullptr;
    pl.blk_start.clear();
    pl.unit_flags.clear();
    for (uint32_t i = 0; i < n_units; i++) {
        const uint64_t len = unit_off[i + 1] - unit_off[i];
        pl.blk0[i] = nb;
        pl.stage_off[i] = so;
        pl.rel_off[i] = unit_off[i] - unit_off[0];
        uint32_t ub = (uint32_t)((len - (c->job_hist ? c->job_hist[i] : 0) + bs - 1) / bs);  // (a job's overlap prefix is history, not blocks)
        if (irregular) {
            const uint64_t* cp = c->cuts + c->cut_off[c->cut_unit0 + i];
            const uint64_t nc = c->cut_off[c->cut_unit0 + i + 1] - c->cut_off[c->cut_unit0 + i];
            uint32_t fl = 0;
            ub = plan_stream_blocks((uint64_t)bs, len, cp, nc, &pl.blk_start, &fl);
            pl.unit_flags.push_back(fl);
        }
        nb += ub;
        // every block costs a 3-byte header: Flush points add blocks that MaxEncodedSize(len) does not count
        so += ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)len) + (irregular ? 3ull * (c->cut_off[c->cut_unit0 + i + 1] - c->cut_off[c->cut_unit0 + i]) + 3ull : 0ull) + 15) & ~(uint64_t)15;
    }
    pl.blk0[n_units] = nb;
    pl.stage_off[n_units] = so;
    pl.rel_off[n_units] = unit_off[n_units] - unit_off[0];
    pl.n_blocks = nb;
    pl.seq_stride = (uint32_t)(bs / 4 + 8);
    pl.lit_stride = (uint32_t)(bs + 64);
    if (so > dst_cap) { c->err = "dst_cap smaller than the sum of MaxEncodedSize(unit)"; return KC_ERR_DST_TOO_SMALL; }

    kc_status s;
    if ((s = ensure(c, c->unit_off, (n_units + 1) * 8)) || (s = ensure(c, c->unit_blk0, (n_units + 1) * 4)) ||
        (s = ensure(c, c->stage_off, (n_units + 1) * 8)) || (s = ensure(c, c->out_off, (n_units + 1 + (feed ? feed->cut.size() : 0)) * 8)) ||
        (s = ensure(c, c->seqs, (size_t)nb * pl.seq_stride * 8)) || (s = ensure(c, c->aux, (size_t)nb * pl.seq_stride * 8)) ||
        (s = ensure(c, c->lits, (size_t)nb * pl.lit_stride)) || (s = ensure(c, c->meta, (size_t)nb * sizeof(KcBlkMeta))) ||
        (s = ensure(c, c->stage, so + 64)) || (s = ensure(c, c->out_size, (size_t)n_units * 4)) ||
        (s = ensure(c, c->xxh, (size_t)n_units * 8)) || (s = ensure(c, c->redo, (size_t)n_units * 4)) ||
        (s = ensure(c, c->redo_blk, (size_t)nb + 1)) || (s = ensure(c, c->pop_blk, (size_t)nb + 1)) || (s = ensure(c, c->unit_list, (size_t)n_units * 4)) ||
        (s = ensure(c, c->predef, kc_fse_predef_bytes())) || (s = ensure(c, c->errflag, 64)))
        return s;
    if (!c->predef_ready) {
        kc_launch_fse_predef_init(c->predef.p, st);
        c->predef_ready = true;
    }
    if (irregular) {
        if ((s = ensure(c, c->blk_start, ((size_t)nb + 1) * 4)) || (s = ensure(c, c->unit_flags, (size_t)n_units * 4))) return s;
        if (nb) HIPCHK(c, hipMemcpyAsync(c->blk_start.p, pl.blk_start.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->unit_flags.p, pl.unit_flags.data(), (size_t)n_units * 4, hipMemcpyHostToDevice, st));
    }
    {   // the layout arrays: batches of equal-sized units repeat them exactly (every step of a fixed-size workload), and a copy from
        // pageable memory stalls the host for tens of microseconds each — upload only what differs from what the device holds
        auto same = [](const std::vector<uint8_t>& held, const void* p, size_t n, const void* dev, const void* held_dev) {
            return dev == held_dev && held.size() == n && memcmp(held.data(), p, n) == 0;
        };
        auto keep = [](std::vector<uint8_t>& held, const void* p, size_t n) { held.assign((const uint8_t*)p, (const uint8_t*)p + n); };
        const size_t n8 = (size_t)(n_units + 1) * 8, n4 = (size_t)(n_units + 1) * 4;
        if (!same(c->up_unit_off, pl.rel_off.data(), n8, c->unit_off.p, c->up_ptr[0])) {
            HIPCHK(c, hipMemcpyAsync(c->unit_off.p, pl.rel_off.data(), n8, hipMemcpyHostToDevice, st));
            keep(c->up_unit_off, pl.rel_off.data(), n8);
            c->up_ptr[0] = c->unit_off.p;
        }
        if (!same(c->up_blk0, pl.blk0.data(), n4, c->unit_blk0.p, c->up_ptr[1])) {
            HIPCHK(c, hipMemcpyAsync(c->unit_blk0.p, pl.blk0.data(), n4, hipMemcpyHostToDevice, st));
            keep(c->up_blk0, pl.blk0.data(), n4);
            c->up_ptr[1] = c->unit_blk0.p;
        }
        if (!same(c->up_stage_off, pl.stage_off.data(), n8, c->stage_off.p, c->up_ptr[2])) {
            HIPCHK(c, hipMemcpyAsync(c->stage_off.p, pl.stage_off.data(), n8, hipMemcpyHostToDevice, st));
            keep(c->up_stage_off, pl.stage_off.data(), n8);
            c->up_ptr[2] = c->stage_off.p;
        }
    }
    // (the flag arrays of the batch are zeroed by one launch further down: kc_launch_clear)

    const uint8_t* d_src = d_src_base + unit_off[0];
    // ---- dictionary (raw content, WithEncoderDictRaw): history = dict || unit (enc_base.go:189-198) ----
    const bool useDict = o->dict != nullptr && o->dict_len > 0;
    const int hist0 = useDict ? (int)o->dict_len : 0;
    uint64_t maxLen = 16;
    for (uint32_t i = 0; i < n_units; i++) maxLen = std::max<uint64_t>(maxLen, unit_off[i + 1] - unit_off[i]);
    pl.max_unit_bytes = maxLen;
    int pos_bits = 1;
    while (((uint64_t)1 << pos_bits) <= (uint64_t)hist0 + maxLen + 2) pos_bits++;
    // the packed sequences keep offset + 3 in 24 bits: fine for the default windows (4 / 8 MiB) and for any unit below 16 MiB
    if (std::min<uint64_t>((uint64_t)o->window_size, (uint64_t)hist0 + maxLen) + 3 > 0xFFFFFFull) {
        c->err = "window above 8 MiB with units above 16 MiB: offsets beyond the device path's 24-bit sequence field";
        return KC_ERR_UNSUPPORTED;
    }
    const uint8_t* k_src = d_src;               // what the match finder / entropy kernels read
    const uint64_t* k_off = (const uint64_t*)c->unit_off.p;
    if (useDict) {
        std::vector<uint64_t> woff(n_units + 1);
        for (uint32_t i = 0; i <= n_units; i++) woff[i] = pl.rel_off[i] + (uint64_t)i * (uint64_t)hist0;
        if ((s = ensure(c, c->work, woff[n_units] + 64)) || (s = ensure(c, c->work_off, (n_units + 1) * 8)) ||
            (s = ensure(c, c->dictbuf, (size_t)hist0 + 64)) || (s = ensure(c, c->proto, kc_zbetter_table_bytes())))
            return s;
        HIPCHK(c, hipMemcpyAsync(c->work_off.p, woff.data(), (n_units + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->dictbuf.p, o->dict, (size_t)hist0, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));  // woff is a local
        kc_launch_prefix_units(d_src, (const uint64_t*)c->unit_off.p, (const uint64_t*)c->work_off.p, (const uint8_t*)c->dictbuf.p,
                               (uint32_t)hist0, (uint8_t*)c->work.p, n_units, st);
        // pristine dictionary tables (betterFastEncoderDict.Reset, enc_better.go:1114-1183) in the device entry format
        std::vector<uint8_t> proto(kc_zbetter_table_bytes(), 0);
        if (o->level == KC_SPEED_BEST) { /* the kernel indexes the dictionary itself, per unit (bestFastEncoder.Reset) */ }
        else if (o->level == KC_SPEED_BETTER) build_better_dict_tables(o->dict, (size_t)hist0, pos_bits, proto.data(), better_epoch_mode(c, o->level, pos_bits, hist0) ? 4 : 0);
        else if (o->level == KC_SPEED_DEFAULT) {
            build_dfast_dict_long(o->dict, (size_t)hist0, pos_bits, (uint32_t*)proto.data());
            build_fast_dict_table(o->dict, (size_t)hist0, pos_bits, (uint32_t*)(proto.data() + ((size_t)4 << 17)));
        } else build_fast_dict_table(o->dict, (size_t)hist0, pos_bits, (uint32_t*)proto.data());
        HIPCHK(c, hipMemcpyAsync(c->proto.p, proto.data(), proto.size(), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
        k_src = (const uint8_t*)c->work.p;
        k_off = (const uint64_t*)c->work_off.p;
    }
    if (c->job_hist) {
        if (useDict) { c->err = "jobs of a WithConcurrentBlocks stream take no dictionary"; return KC_ERR_INTERNAL; }
        if ((s = ensure(c, c->d_job_hist, (size_t)n_units * 4)) || (s = ensure(c, c->d_job_flags, (size_t)n_units * 4))) return s;
        HIPCHK(c, hipMemcpyAsync(c->d_job_hist.p, c->job_hist, (size_t)n_units * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->d_job_flags.p, c->job_flags, (size_t)n_units * 4, hipMemcpyHostToDevice, st));
    }
    KcMatchParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.unit_hist = c->job_hist ? (const uint32_t*)c->d_job_hist.p : nullptr;
    mp.job_flags = c->job_hist ? (const uint32_t*)c->d_job_flags.p : nullptr;
    mp.src = k_src;
    mp.src_end = k_src + (useDict ? pl.rel_off[n_units] + (uint64_t)n_units * (uint64_t)hist0 : pl.rel_off[n_units]);
    mp.unit_off = k_off;
    mp.hist0 = hist0;
    mp.pos_bits = pos_bits;
    mp.stream_mode = c->stream_mode;
    mp.rep1 = (int32_t)o->dict_offsets[0];
    mp.rep2 = (int32_t)o->dict_offsets[1];
    mp.rep3 = (int32_t)o->dict_offsets[2];
    if (mp.rep1 <= 0 || mp.rep2 <= 0 || mp.rep3 <= 0) { mp.rep1 = 1; mp.rep2 = 4; mp.rep3 = 8; }  // opts not initialised through kc_zstd_opts_default
    mp.unit_blk0 = (const uint32_t*)c->unit_blk0.p;
    mp.seqs = (uint64_t*)c->seqs.p;
    mp.meta = (KcBlkMeta*)c->meta.p;
    mp.pop_blk = nullptr;
    mp.unit_list = nullptr;
    mp.blk_start = irregular ? (const uint32_t*)c->blk_start.p : nullptr;
    mp.unit_flags = irregular ? (const uint32_t*)c->unit_flags.p : nullptr;
    mp.seq_stride = pl.seq_stride;
    mp.block_size = bs;
    mp.max_match_off = o->window_size;
    // SpeedDefault, round 2 (ms per launch): at 4 GiB (32768 units: DRAM-transaction bound, wasted probes cost) width 2 / 3 / 4 then doubling
    // 265.9 / 266.7 / 274.0, 2 then +1 264.1; at 2 GiB (latency bound) 163.6 / - / 155.4, fixed 1: 285.5.  The BASELINE size is 4 GiB.
    // SpeedBetterCompression (1 GiB = 8192 units: latency-bound, wide speculation pays): width 1 / 2 / 4 then doubling 98.6 / 91.5 / 86.0,
    // fixed 4: 101.4, fixed 8: 80.4 ms with 8 lanes per unit; 16 lanes per unit, fixed 16: 61.1 ms (one probe per round, round 1: 181 ms)
    mp.spec_w0 = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : (o->level == KC_SPEED_DEFAULT ? 2 : (o->level == KC_SPEED_BETTER ? 16 : 1));
    // measured on C2 (ms per 4 GiB): width 1 then +1 per miss 137, fixed 2 136.5, 1 then doubling 140, fixed 1 167, fixed 4 157
    mp.spec_grow = c->cfg.spec_grow >= 0 ? (int)c->cfg.spec_grow : (o->level == KC_SPEED_FASTEST ? 1 : (o->level == KC_SPEED_BETTER ? 0 : 2));
    if (mp.spec_w0 < 1) mp.spec_w0 = 1;
    if (mp.spec_w0 > 64) mp.spec_w0 = 64;  // the kernels clamp to their group size

    KcEntropyParams ep;
    memset(&ep, 0, sizeof(ep));
    ep.unit_hist = mp.unit_hist;
    ep.job_flags = mp.job_flags;
    ep.src = k_src;
    ep.unit_off = k_off;
    ep.hist0 = hist0;
    ep.stream_mode = c->stream_mode;
    ep.stream_sync = o->concurrent == 1;
    ep.dict_huf = nullptr;
    ep.dict_huf_len = 0;
    ep.dict_huf_log = 0;
    if (o->dict_huf_len > 0) {  // dictionary literal table -> prevTable of every unit's first block
        uint8_t blobh[768];
        memcpy(blobh, o->dict_huf_val, 512);
        memcpy(blobh + 512, o->dict_huf_nbits, 256);
        if ((s = ensure(c, c->dicthuf, 768))) return s;
        HIPCHK(c, hipMemcpyAsync(c->dicthuf.p, blobh, 768, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));  // blobh is a stack buffer
        ep.dict_huf = (const uint8_t*)c->dicthuf.p;
        ep.dict_huf_len = o->dict_huf_len;
        ep.dict_huf_log = o->dict_huf_log;
    }
    ep.unit_blk0 = mp.unit_blk0;
    ep.seqs = mp.seqs;
    ep.meta = mp.meta;
    ep.lits = (uint8_t*)c->lits.p;
    ep.aux = (uint64_t*)c->aux.p;
    ep.stage = (uint8_t*)c->stage.p;
    ep.stage_off = (const uint64_t*)c->stage_off.p;
    ep.out_size = (uint32_t*)c->out_size.p;
    ep.xxh = (const uint64_t*)c->xxh.p;
    ep.redo_mask = (uint32_t*)c->redo.p;
    ep.redo_blk = (uint8_t*)c->redo_blk.p;
    ep.blk_start = mp.blk_start;
    ep.unit_flags = mp.unit_flags;
    ep.unit_list = nullptr;
    ep.predef = c->predef.p;
    ep.seq_stride = pl.seq_stride;
    ep.lit_stride = pl.lit_stride;
    ep.block_size = bs;
    ep.window_size = o->window_size;
    ep.crc = o->crc;
    ep.single = o->single;
    ep.no_entropy = o->no_entropy;
    ep.all_lit_entropy = o->all_lit_entropy;
    ep.full_zero = o->full_zero;
    ep.dict_id = o->dict_id;
    ep.err_flag = (uint32_t*)c->errflag.p;
    // raw blocks are copied once, by the compaction, from the source (KcRawDef): the entries of blocks that are not raw stay zero
    if ((s = ensure(c, c->rawdef, ((size_t)nb + 1) * sizeof(KcRawDef))) != KC_OK) return s;
    ep.rawdef = (KcRawDef*)c->rawdef.p;
    // The checksum moves behind the entropy stage (kc_xxh64_fin_kernel) where the batch has a regular block grid and no history in
    // front of its units: frames that turn out to be raw blocks only then get their payload copied by the pass that hashes it.
    const bool fuse_xxh = c->cfg.fuse_raw_xxh != 0 && o->crc && !c->job_hist && !useDict && hist0 == 0 && !irregular && (bs % 256) == 0 && feed == nullptr;
    ep.unit_raw = nullptr;
    if (fuse_xxh) {
        if ((s = ensure(c, c->unit_raw, (size_t)n_units * 4 + 4)) != KC_OK) return s;
        ep.unit_raw = (uint32_t*)c->unit_raw.p;
        ep.xxh = nullptr;
    }
    {   // the batch's flag arrays, zeroed by one launch (every DevBuf has at least 256 bytes of slack behind the size asked for)
        KcClearList cl;
        memset(&cl, 0, sizeof(cl));
        auto add = [&](void* q, size_t bytes) { cl.p[cl.count] = q; cl.n16[cl.count] = (bytes + 15) / 16; cl.count++; };
        add(c->redo.p, (size_t)n_units * 4);
        add(c->redo_blk.p, (size_t)nb + 1);
        add(c->errflag.p, 64);
        add(c->rawdef.p, ((size_t)nb + 1) * sizeof(KcRawDef));
        if (fuse_xxh) add(c->unit_raw.p, (size_t)n_units * 4);
        kc_launch_clear(cl, st);
    }
    ep.prof = nullptr;
    const bool k2prof = c->cfg.k2_prof != 0;
    if (k2prof) {
        if ((s = ensure(c, c->prof, 48 * 8)) != KC_OK) return s;
        HIPCHK(c, hipMemsetAsync(c->prof.p, 0, 48 * 8, st));
        ep.prof = (unsigned long long*)c->prof.p;
        mp.prof = (unsigned long long*)c->prof.p + 32;
    }

    if (c->chain_after) HIPCHK(c, hipStreamWaitEvent(st, c->chain_after->ev[2], 0));  // pipelined contexts: one match finder at a time
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    // The no-match pre-scan (kc_zstd_prescan.hip): SpeedFastest EncodeAll batches whose frames take the deferred-payload path
    // (fuse_xxh: checksum on, regular block grid, no dictionary / job prefix / chunk feed), literal-only blocks going out raw
    // (rawAllLits, the default below SpeedBetterCompression).  On by option, or per batch when the context's previous batch did not
    // compress (the same signal that picks the match finder's form for such input).
    c->prescan_ran = false;
    {
        const bool tuned_now = c->cfg.zfast_variant < 0 ? c->last_incompressible : c->cfg.zfast_variant == 1;
        const bool want = c->cfg.zfast_prescan > 0 || (c->cfg.zfast_prescan < 0 && tuned_now);
        if (want && o->level == KC_SPEED_FASTEST && fuse_xxh && !c->stream_mode && !o->all_lit_entropy && bs >= 16 && n_units > 0) {
            if (c->probe_bs != bs || c->probe_rel.p == nullptr) {
                std::vector<uint32_t> rel(4096);
                uint32_t n = kc_zfast_probe_positions(bs, rel.data(), (uint32_t)rel.size());
                if (n > rel.size()) { rel.resize(n); n = kc_zfast_probe_positions(bs, rel.data(), (uint32_t)rel.size()); }
                if ((s = ensure(c, c->probe_rel, (size_t)n * 4 + 16)) != KC_OK) return s;
                HIPCHK(c, hipMemcpyAsync(c->probe_rel.p, rel.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
                HIPCHK(c, hipStreamSynchronize(st));  // rel is a local
                c->probe_bs = bs;
                c->probe_n = n;
            }
            if ((s = ensure(c, c->unit_done, (size_t)n_units * 4 + 16)) != KC_OK) return s;
            KcPrescanParams pp;
            memset(&pp, 0, sizeof(pp));
            pp.src = k_src;
            pp.unit_off = k_off;
            pp.unit_blk0 = mp.unit_blk0;
            pp.n_units = n_units;
            pp.block_size = bs;
            pp.probe_rel = (const uint32_t*)c->probe_rel.p;
            pp.n_probe = c->probe_n;
            pp.rep1 = mp.rep1;
            pp.rep2 = mp.rep2;
            pp.meta = mp.meta;
            pp.unit_done = (uint32_t*)c->unit_done.p;
            pp.stage = ep.stage;
            pp.stage_off = ep.stage_off;
            pp.out_size = ep.out_size;
            pp.rawdef = ep.rawdef;
            pp.unit_raw = ep.unit_raw;
            pp.window_size = o->window_size;
            pp.crc = o->crc;
            pp.single = o->single;
            pp.dict_id = o->dict_id;
            kc_launch_zfast_prescan(pp, st);
            mp.unit_done = pp.unit_done;
            ep.unit_done = pp.unit_done;
            c->prescan_ran = true;
        }
    }
    mp.unit_base = 0;
    ep.unit_base = 0;
    if (feed == nullptr) {
        if (o->crc && !c->job_hist && !fuse_xxh) kc_launch_xxh64(d_src, (const uint64_t*)c->unit_off.p, n_units, (uint64_t*)c->xxh.p, st);  // (a job stream's checksum is the host's)
        HIPCHK(c, hipEventRecord(c->ev[1], st));
        if ((s = launch_match(c, mp, unit_off, n_units, n_units, bs, st, o->level)) != KC_OK) return s;
    } else {
        // the source is still arriving: per chunk, checksum + match finder on the chunk's stream behind its H2D copy
        if (useDict) { c->err = "chunk feed does not take dictionaries"; return KC_ERR_INTERNAL; }
        HIPCHK(c, hipEventRecord(c->ev[1], st));
        if ((s = prepare_tables(c, mp, n_units, st, o->level)) != KC_OK) return s;
        const bool feed_lds = zfast_use_lds(c, mp, n_units, o->level);
        HIPCHK(c, hipEventRecord(c->ev[6], st));  // everything the chunk kernels need from this stream (offset arrays, tables)
        const size_t nchunk = feed->cut.size() - 1;
        for (size_t k = 0; k < nchunk; k++) {
            if (!feed->wait_recorded(k)) { c->err = "host pipeline: staging failed"; return KC_ERR_HIP; }
            hipStream_t sk = feed->streams[k % feed->streams.size()];
            const uint32_t u0 = feed->cut[k], nk = feed->cut[k + 1] - feed->cut[k];
            HIPCHK(c, hipStreamWaitEvent(sk, c->ev[6], 0));
            HIPCHK(c, hipStreamWaitEvent(sk, feed->landed[k], 0));
            if (o->crc) kc_launch_xxh64(d_src, (const uint64_t*)c->unit_off.p + u0, nk, (uint64_t*)c->xxh.p + u0, sk);
            KcMatchParams mk = mp;
            mk.unit_base = u0;
            launch_match_kernel(c, mk, u0, nk, sk, o->level, feed_lds);
            KcEntropyParams ek = ep;
            ek.unit_base = u0;
            kc_launch_zstd_entropy(ek, nk, sk);
            feed->loc_off = (uint64_t*)c->out_off.p;
            kc_launch_scan_sizes((const uint32_t*)c->out_size.p + u0, nk, feed->loc_off + u0 + k, sk);
            kc_launch_compact((const uint8_t*)c->stage.p, (const uint64_t*)c->stage_off.p + u0, (const uint32_t*)c->out_size.p + u0,
                              feed->loc_off + u0 + k, d_dst + pl.stage_off[u0], nk, sk, ep.src, ep.unit_off + u0, ep.unit_blk0 + u0, ep.rawdef);
            HIPCHK(c, hipEventRecord(feed->done[k], sk));
        }
        for (size_t k = 0; k < nchunk; k++) HIPCHK(c, hipStreamWaitEvent(st, feed->done[k], 0));
    }
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    HIPCHK(c, hipGetLastError());
    Pending* P = new Pending();
    P->o = *o;
    P->mp = mp;
    P->ep = ep;
    P->unit_off.assign(unit_off, unit_off + n_units + 1);
    P->n_units = n_units;
    P->d_dst = d_dst;
    P->bs = bs;
    P->k2prof = k2prof;
    c->pend = P;
    return KC_OK;
}

kc_status batch_end(kc_ctx* c, uint64_t* out_off_host, uint64_t* produced) {
    if (!c->pend) { c->err = "no batch in flight on this context"; return KC_ERR_BAD_ARG; }
    std::unique_ptr<Pending> P((Pending*)c->pend);
    c->pend = nullptr;
    hipStream_t st = c->stream;
    if (c->stream2 != nullptr) {  // the second stage on a stream of its own (e.g. one restricted to other CUs than the match finder's)
        st = c->stream2;
        HIPCHK(c, hipStreamWaitEvent(st, c->ev[2], 0));
    }
    const kc_zstd_opts* o = &P->o;
    KcMatchParams& mp = P->mp;
    KcEntropyParams& ep = P->ep;
    const uint64_t* unit_off = P->unit_off.data();
    const uint32_t n_units = P->n_units;
    uint8_t* d_dst = P->d_dst;
    const int bs = P->bs;
    const bool k2prof = P->k2prof;
    kc_status s;
    kc_launch_zstd_entropy(ep, n_units, st);
    HIPCHK(c, hipEventRecord(c->ev[3], st));
    HIPCHK(c, hipGetLastError());

    // sizes -> offsets, checksum (+ payload of the raw-only frames), compaction.  Enqueued right behind the entropy stage, before the
    // host has looked at the re-run flags: a re-run is rare, and when one happens the pass simply runs again behind it (it rewrites
    // every byte of dst) — so the device does not idle through the flag read-back of every batch.
    auto finish_pass = [&]() -> kc_status {
        kc_launch_scan_sizes((const uint32_t*)c->out_size.p, n_units, (uint64_t*)c->out_off.p, st);
        if (ep.unit_raw != nullptr) {  // the checksum, and with it the payload of the frames that are raw blocks only
            KcXxhFinParams xf;
            xf.src = ep.src;
            xf.unit_off = ep.unit_off;
            xf.n_units = n_units;
            xf.stage = (uint8_t*)c->stage.p;
            xf.stage_off = (const uint64_t*)c->stage_off.p;
            xf.out_size = (const uint32_t*)c->out_size.p;
            xf.out_off = (const uint64_t*)c->out_off.p;
            xf.dst = d_dst;
            xf.unit_raw = ep.unit_raw;
            xf.rawdef = ep.rawdef;
            xf.unit_blk0 = ep.unit_blk0;
            xf.xxh_out = (uint64_t*)c->xxh.p;
            xf.mode = (int32_t)c->cfg.xxh_fin_mode;
            kc_launch_xxh64_fin(xf, st);
        }
        kc_launch_compact((const uint8_t*)c->stage.p, (const uint64_t*)c->stage_off.p, (const uint32_t*)c->out_size.p,
                          (const uint64_t*)c->out_off.p, d_dst, n_units, st, ep.src, ep.unit_off, ep.unit_blk0, ep.rawdef, ep.unit_raw);
        HIPCHK(c, hipGetLastError());
        return KC_OK;
    };
    HIPCHK(c, hipEventRecord(c->ev[4], st));
    if ((s = finish_pass()) != KC_OK) return s;
    HIPCHK(c, hipEventRecord(c->ev[5], st));
    // Speculation check: a block that fell back to raw only after entropy coding (blockenc.go:811-817)
    // pops the repeat offsets; if the following block was parsed with the un-popped offsets the unit is
    // re-run with that verdict forced.  Rare (needs a compressible-looking block that ends larger than raw).
    uint32_t redo_units = 0;
    {
        const Plan& pl = c->plan;  // this batch's layout (one batch in flight per context)
        const uint32_t nb = pl.n_blocks;
        uint32_t maxBlocks = 1;
        for (uint32_t i = 0; i < n_units; i++) maxBlocks = std::max(maxBlocks, pl.blk0[i + 1] - pl.blk0[i]);
        std::vector<uint32_t> redo(n_units), list;
        std::vector<uint8_t> redo_blk, pop_blk;
        uint32_t errv[16];
        for (uint32_t iter = 0;; iter++) {
            HIPCHK(c, hipMemcpyAsync(redo.data(), c->redo.p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(errv, c->errflag.p, 64, hipMemcpyDeviceToHost, st));
            std::vector<uint32_t> doneh;
            if (iter == 0 && c->prescan_ran) {
                doneh.resize(n_units);
                HIPCHK(c, hipMemcpyAsync(doneh.data(), c->unit_done.p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
            }
            HIPCHK(c, hipStreamSynchronize(st));
            if (iter == 0) {
                c->last_prescan_units = 0;
                for (uint32_t v : doneh) c->last_prescan_units += v != 0u;
            }
            if (errv[0] != 0) {
                char b[96];
                snprintf(b, sizeof(b), "device invariant violated (code %u)", errv[0]);
                c->err = b;
                return errv[0] == 100u ? KC_ERR_UNSUPPORTED : KC_ERR_INTERNAL;
            }
            list.clear();
            for (uint32_t i = 0; i < n_units; i++)
                if (redo[i]) list.push_back(i);
            if (list.empty()) break;
            if (iter > maxBlocks + 1) { c->err = "speculation re-run did not converge"; return KC_ERR_INTERNAL; }  // every pass settles one more block per unit
            redo_blk.resize(nb);
            if (pop_blk.empty()) pop_blk.assign(nb, 0);
            HIPCHK(c, hipMemcpyAsync(redo_blk.data(), c->redo_blk.p, nb, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            for (uint32_t i : list)
                for (uint32_t b = pl.blk0[i]; b < pl.blk0[i + 1]; b++)
                    if (redo_blk[b]) { pop_blk[b] = 1; break; }  // only the lowest flagged block is trustworthy
            redo_units += (uint32_t)list.size();
            HIPCHK(c, hipMemcpyAsync(c->pop_blk.p, pop_blk.data(), nb, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->unit_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemsetAsync(c->redo.p, 0, (size_t)n_units * 4, st));
            HIPCHK(c, hipMemsetAsync(c->redo_blk.p, 0, (size_t)nb + 1, st));
            c->job_redo_list = list;
            mp.pop_blk = (const uint8_t*)c->pop_blk.p;
            mp.unit_list = (const uint32_t*)c->unit_list.p;
            ep.unit_list = mp.unit_list;
            if ((s = launch_match(c, mp, unit_off, n_units, (uint32_t)list.size(), bs, st, o->level)) != KC_OK) return s;
            kc_launch_zstd_entropy(ep, (uint32_t)list.size(), st);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(st));  // pop_blk / list are host vectors: the copies above must have been taken before the next pass rewrites them
        }
    }
    if (redo_units != 0) {  // the frames of the re-run units changed: sizes, offsets and everything behind them
        HIPCHK(c, hipEventRecord(c->ev[4], st));
        if ((s = finish_pass()) != KC_OK) return s;
        HIPCHK(c, hipEventRecord(c->ev[5], st));
    }
    HIPCHK(c, hipMemcpyAsync(out_off_host, c->out_off.p, (n_units + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t45 = 0, t05 = 0, tk2 = 0;
    (void)hipEventElapsedTime(&t01, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&t23, c->ev[2], c->ev[3]);
    (void)hipEventElapsedTime(&t34, c->ev[3], c->ev[4]);
    (void)hipEventElapsedTime(&t45, c->ev[4], c->ev[5]);
    (void)hipEventElapsedTime(&t05, c->ev[0], c->ev[5]);
    if (c->ev7_valid) {  // (the speculation re-run records it again: then it brackets the re-run's preparation, a few units)
        float t17 = 0;
        if (hipEventElapsedTime(&t17, c->ev[1], c->ev[7]) == hipSuccess && t17 >= 0 && t17 <= t12) c->last.prep_ms += t17;
        c->ev7_valid = false;
    }
    tk2 = t23;
    c->last.match_ms += t12;       // all match-finder launches (with overlap: includes time shared with entropy kernels)
    c->last.entropy_ms += tk2;     // first to last entropy launch on its stream
    c->last.other_ms += t01 + t34 + t45;
    c->last.total_ms += t05;
    c->last.redo_units += redo_units;
    if (k2prof) {
        unsigned long long pv[48];
        HIPCHK(c, hipMemcpy(pv, c->prof.p, sizeof(pv), hipMemcpyDeviceToHost));
        {
            unsigned long long lt = 0;
            for (int i = 32; i < 40; i++) lt += pv[i];
            if (lt && o->level == KC_SPEED_DEFAULT) {  // -DKC_ZD_STATS build of kc_zstd_match_dfast.hip
                fprintf(stderr, "[dfast stats] per unit: probes looked up %.0f, committed %.0f, candidate / repeat 16-byte loads %.0f, long lookups at s+1 %.0f, matches %.0f, offset-2 matches %.0f, ring refills (128 B) %.0f\n",
                        (double)pv[32] / n_units, (double)pv[33] / n_units, (double)pv[34] / n_units, (double)pv[35] / n_units, (double)pv[36] / n_units, (double)pv[37] / n_units, (double)pv[38] / n_units);
            } else if (lt) {
                fprintf(stderr, "[LDS match prof] shader clocks per phase (window, probe bytes, table, candidates issued, verdicts, commit, -, round tail):");
                for (int i = 32; i < 40; i++) fprintf(stderr, " %.1f%%", 100.0 * (double)pv[i] / (double)lt);
                fprintf(stderr, "  (total %.4g cycles over %u units)\n", (double)lt, n_units);
            }
        }
        {
            unsigned long long ft = 0;
            for (int i = 40; i < 48; i++) ft += pv[i];
            if (ft) {
                fprintf(stderr, "[K2 fine] wave 0, shader clocks per unit; -DKC_K2_FINE=1: gather (pass A, barrier, step: sequences + scan, step: literal loads + ORs, step: flush + histogram, step: carry + zero, tail, closing barrier); =2: (Huffman size pass, payload zero fill + barrier, stream emit, barrier, payload copy + headers, code staging, chains, pack):");
                for (int i = 40; i < 48; i++) fprintf(stderr, " %.0f", (double)pv[i] / (double)(n_units ? n_units : 1));
                fprintf(stderr, "\n");
            }
        }
        unsigned long long tot = 0;
        for (int i = 0; i < 16; i++) tot += pv[i];
        fprintf(stderr, "[K2 prof] shader-clock share per phase:");
        for (int i = 0; i < 14; i++) fprintf(stderr, " p%d=%.1f%%", i, tot ? 100.0 * (double)pv[i] / (double)tot : 0.0);
        fprintf(stderr, "  (total %.3g cycles over %u units)\n", (double)tot, n_units);
        fprintf(stderr, "[K2 prof] tANS chains (counted with -DKC_CHAIN_STATS): %llu chunk-streams (%llu RLE, %llu predefined), %llu repair passes (LL %llu, OF %llu, ML %llu; %llu on RLE tables), %llu segments re-encoded\n",
                pv[16], pv[26], pv[27], pv[17], pv[18], pv[19], pv[20], pv[25], pv[24]);
    }
    *produced = out_off_host[n_units];
    {
        const uint64_t in_total = unit_off[n_units] - unit_off[0];
        c->last_incompressible = in_total >= (1u << 20) && (double)*produced >= 0.98 * (double)in_total;
    }
    return KC_OK;
}

// A batch is bounded by its input bytes AND by the device scratch it needs: tables are per unit, sequences / literals /
// staging are per block at a fixed stride whatever the block's actual length, so many small units need far more than the
// "~6x input" of full-size units (1M x 4 KiB units at SpeedDefault would ask for hundreds of GiB in one batch).
uint64_t zstd_unit_scratch(const kc_zstd_opts* o, uint64_t len, uint64_t n_cuts = 0) {
    const uint64_t bsz = (uint64_t)o->block_size;
    const uint64_t table_b = o->level == KC_SPEED_BEST ? 0 : o->level == KC_SPEED_BETTER ? kc_zbetter_table_bytes() : (o->level == KC_SPEED_DEFAULT ? kc_zdfast_table_bytes() : kc_zfast_table_bytes());
    const uint64_t per_block = 2 * (bsz / 4 + 8) * 8 + (bsz + 64) + sizeof(KcBlkMeta);
    const uint64_t hist0 = (o->dict != nullptr) ? o->dict_len : 0;
    const uint64_t blocks = (len + bsz - 1) / bsz + n_cuts;  // every Flush point can add a block, at the full per-block strides
    const uint64_t enc = ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)len) + 3 * n_cuts + 3 + 15) & ~(uint64_t)15;
    return table_b + blocks * per_block + enc + (hist0 ? hist0 + len : 0) + 64;
}

// Scratch a batch may ask for: the configured ceiling, or 85 % of what is free plus what this context already owns (re-used).
uint64_t scratch_budget(kc_ctx* c) {
    uint64_t budget = c->max_scratch_bytes;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        uint64_t held = 0;
        const DevBuf* bufs[] = {&c->seqs, &c->aux, &c->lits, &c->meta, &c->stage, &c->tables, &c->work};
        for (const DevBuf* b : bufs) held += b->cap;
        const uint64_t avail = (uint64_t)((double)(fr + held) * 0.85);
        if (avail < budget) budget = avail;
    } else {
        (void)hipGetLastError();
    }
    return budget;
}

// End of a chunk-fed batch (batch_begin with a ChunkFeed): every chunk has already been entropy coded and compacted on its own
// stream.  *redo is set when a unit needs the speculation re-run (see batch_end): the caller encodes the batch again the plain way.
kc_status feed_finish(kc_ctx* c, bool* redo_needed) {
    if (!c->pend) { c->err = "no batch in flight on this context"; return KC_ERR_BAD_ARG; }
    std::unique_ptr<Pending> P((Pending*)c->pend);
    c->pend = nullptr;
    hipStream_t st = c->stream;
    const uint32_t n_units = P->n_units;
    std::vector<uint32_t> redo(n_units);
    uint32_t errv[16];
    HIPCHK(c, hipMemcpyAsync(redo.data(), c->redo.p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(errv, c->errflag.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    if (errv[0] != 0) {
        char b[96];
        snprintf(b, sizeof(b), "device invariant violated (code %u)", errv[0]);
        c->err = b;
        return errv[0] == 100u ? KC_ERR_UNSUPPORTED : KC_ERR_INTERNAL;
    }
    *redo_needed = false;
    for (uint32_t i = 0; i < n_units; i++) if (redo[i]) *redo_needed = true;
    return KC_OK;
}

kc_status run_batch(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src_base, const uint64_t* unit_off, uint32_t n_units,
                    uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off_host, uint64_t* produced) {
    kc_status s = batch_begin(c, o, d_src_base, unit_off, n_units, d_dst, dst_cap);
    if (s != KC_OK) return s;
    return batch_end(c, out_off_host, produced);
}

}  // namespace

extern "C" {

kc_status kc_zstd_encode_units_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                   uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off || !out_off || (n_units && (!d_src || !d_dst))) return KC_ERR_BAD_ARG;
    c->err.clear();
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    for (uint32_t i = 0; i < n_units; i++) {
        if (unit_off[i + 1] < unit_off[i]) { c->err = "unit_off not ascending"; return KC_ERR_BAD_ARG; }
        if (unit_off[i + 1] - unit_off[i] > KC_MAX_UNIT_BYTES) {
            c->err = "unit larger than 1 GiB: not served by the device path";
            return KC_ERR_UNSUPPORTED;
        }
    }
    out_off[0] = 0;
    c->last_batches = 0;
    uint64_t pos = 0;
    uint32_t i0 = 0;
    std::vector<uint64_t> tmp;
    auto unit_scratch = [&](uint32_t i) {
        const uint64_t nc = c->cuts ? c->cut_off[i + 1] - c->cut_off[i] : 0;
        return zstd_unit_scratch(o, unit_off[i + 1] - unit_off[i], nc);
    };
    uint64_t budget = scratch_budget(c);
    for (int attempt = 0;; attempt++) {
        bool oom = false;
        while (i0 < n_units) {
            uint32_t i1 = i0;
            const uint64_t cap_bytes = o->level == KC_SPEED_BETTER ? ((uint64_t)1 << 30) : c->max_batch_bytes;  // better: 4 MiB of tables per unit
            const uint32_t cap_units = o->level == KC_SPEED_BETTER ? 16384u : 0xFFFFFFFFu;
            uint64_t scratch = 0;
            while (i1 < n_units) {
                const uint64_t us = unit_scratch(i1);
                // ensure() over-allocates by 1/8
                if (i1 > i0 && (unit_off[i1 + 1] - unit_off[i0] > cap_bytes || i1 - i0 >= cap_units || (scratch + us) + ((scratch + us) >> 3) > budget)) break;
                scratch += us;
                i1++;
            }
            const uint32_t nb = i1 - i0;
            tmp.resize(nb + 1);
            uint64_t produced = 0;
            c->cut_unit0 = i0;
            c->oom = false;
            s = run_batch(c, o, d_src, unit_off + i0, nb, d_dst + pos, dst_cap - pos, tmp.data(), &produced);
            if (s == KC_ERR_UNSUPPORTED && c->oom && nb > 1 && attempt < 6) { oom = true; break; }
            if (s != KC_OK) return s;
            c->last_batches++;
            for (uint32_t k = 0; k <= nb; k++) out_off[i0 + k] = pos + tmp[k];
            pos += produced;
            i0 = i1;
        }
        if (!oom) break;
        budget /= 2;  // another process took device memory since hipMemGetInfo: retry this batch at half the size
        c->err.clear();
    }
    if (n_units == 0) out_off[0] = 0;
    return KC_OK;
}

static kc_status validate_units(kc_ctx* c, const kc_zstd_opts* o, const uint64_t* unit_off, uint32_t n_units) {
    for (uint32_t i = 0; i < n_units; i++) {
        if (unit_off[i + 1] < unit_off[i]) { c->err = "unit_off not ascending"; return KC_ERR_BAD_ARG; }
        if (unit_off[i + 1] - unit_off[i] > KC_MAX_UNIT_BYTES) {
            c->err = "unit larger than 1 GiB: not served by the device path";
            return KC_ERR_UNSUPPORTED;
        }
    }
    return KC_OK;
}

kc_status kc_zstd_encode_units_dev_begin(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                         uint8_t* d_dst, uint64_t dst_cap) {
    if (!c || !o || !unit_off || n_units == 0 || !d_src || !d_dst) return KC_ERR_BAD_ARG;
    c->err.clear();
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    if ((s = validate_units(c, o, unit_off, n_units)) != KC_OK) return s;
    const uint64_t cap_bytes = o->level == KC_SPEED_BETTER ? ((uint64_t)1 << 30) : c->max_batch_bytes;
    if (unit_off[n_units] - unit_off[0] > cap_bytes || (o->level == KC_SPEED_BETTER && n_units > 16384u)) {
        c->err = "begin/end serves one device batch; use kc_zstd_encode_units_dev for larger inputs";
        return KC_ERR_UNSUPPORTED;
    }
    return batch_begin(c, o, d_src, unit_off, n_units, d_dst, dst_cap);
}

kc_status kc_zstd_encode_units_dev_end(kc_ctx* c, uint64_t* out_off) {
    if (!c || !out_off) return KC_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t produced = 0;
    return batch_end(c, out_off, &produced);
}

void kc_ctx_chain_after(kc_ctx* c, kc_ctx* prev) {
    if (c) c->chain_after = prev;
}

kc_status kc_zstd_encode_streams_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                     uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o) return KC_ERR_BAD_ARG;
    c->stream_mode = 1;
    const kc_status s = kc_zstd_encode_units_dev(c, o, d_src, unit_off, n_units, d_dst, dst_cap, out_off);
    c->stream_mode = 0;
    return s;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Host-buffer path (what the cgo shim calls): a three-stage pipeline over sub-batches —
//   stager thread : pageable source -> pinned slot (parallel memcpy) -> device (copy stream)
//   caller thread : the device encode of the sub-batch (context stream)
//   drainer thread: device -> pinned slot (copy-back stream) -> caller's dst (parallel memcpy)
// so the PCIe transfers and the host copies of sub-batch k+1 / k-1 run under the kernels of sub-batch k.
// Two slots per direction; the reference's own threading seam is EncodeAll being safe for concurrent use
// (zstd/encoder.go:722-729) — here the concurrency is inside one call.
// ---------------------------------------------------------------------------------------
namespace {

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

void host_pipe_free(void* h) { delete (HostPipe*)h; }

int host_copy_threads() {
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
int host_copy_threads(const kc_ctx* c) {
    const int64_t t = c->cfg.host_copy_threads;
    return t >= 1 ? (int)(t > 16 ? 16 : t) : host_copy_threads();
}

void parallel_memcpy(uint8_t* dst, const uint8_t* src, size_t n, int threads) {
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
kc_status host_overlapped_zstd(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
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
uint64_t host_sub_bytes(const kc_ctx* c, uint64_t total) {
    if (c->cfg.host_pipe_mib >= 16) return (uint64_t)c->cfg.host_pipe_mib << 20;
    return total >= ((uint64_t)4 << 30) ? ((uint64_t)2 << 30) : ((uint64_t)1 << 30);  // at least two stages from 2 GiB of input on
}

}  // namespace

extern "C" {

kc_status kc_zstd_encode_units(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                               uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off || !out_off || (n_units && (!src || !dst))) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_supported(c, o);  // before any byte moves: an unsupported request must not pay the PCIe copy
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    if (n_units == 0) { out_off[0] = 0; return KC_OK; }
    if ((s = validate_units(c, o, unit_off, n_units)) != KC_OK) return s;
    const uint64_t total = unit_off[n_units] - unit_off[0];
    const uint64_t ov_min = c->cfg.host_overlap_min_mib >= 0 ? (uint64_t)c->cfg.host_overlap_min_mib << 20 : (uint64_t)1 << 30;
    if (total >= ov_min && !c->cfg.host_serial && c->cfg.host_pipe_mib < 16 && c->cuts == nullptr) {
        s = host_overlapped_zstd(c, o, src, unit_off, n_units, dst, dst_cap, out_off);
        if (s != KC_ERR_UNSUPPORTED || !c->err.empty()) return s;  // UNSUPPORTED with no message: shape not served by the one-batch path
    }
    const uint64_t sub = host_sub_bytes(c, total);
    if (total >= 2 * sub && !c->cfg.host_serial && c->cuts == nullptr) {  // (Flush points are indexed by unit: one batch loop)
        auto enc = [&](const uint8_t* d_in, const uint64_t* rel, uint32_t nu, uint8_t* d_out, uint64_t cap, uint64_t* oo) {
            return kc_zstd_encode_units_dev(c, o, d_in, rel, nu, d_out, cap, oo);
        };
        auto mx = [&](uint64_t len) { return (uint64_t)kc_zstd_max_encoded_size(o, (int64_t)len); };
        return host_pipeline(c, src, unit_off, n_units, dst, dst_cap, out_off, sub, enc, mx);
    }
    uint64_t need = 0;
    for (uint32_t i = 0; i < n_units; i++)
        need += ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)(unit_off[i + 1] - unit_off[i])) + (c->cuts ? 3 * (c->cut_off[i + 1] - c->cut_off[i]) + 3 : 0) + 15) & ~(uint64_t)15;
    if ((s = ensure(c, c->tmp_src, total + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
    HIPCHK(c, hipMemcpyAsync(c->tmp_src.p, src + unit_off[0], total, hipMemcpyHostToDevice, c->stream));
    std::vector<uint64_t> rel(n_units + 1);
    for (uint32_t i = 0; i <= n_units; i++) rel[i] = unit_off[i] - unit_off[0];
    s = kc_zstd_encode_units_dev(c, o, (const uint8_t*)c->tmp_src.p, rel.data(), n_units, (uint8_t*)c->tmp_dst.p, need, out_off);
    if (s != KC_OK) return s;
    const uint64_t outn = out_off[n_units];
    if (outn > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
    HIPCHK(c, hipMemcpyAsync(dst, c->tmp_dst.p, outn, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KC_OK;
}

kc_status kc_zstd_encode_streams(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                                 uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o) return KC_ERR_BAD_ARG;
    c->stream_mode = 1;
    const kc_status s = kc_zstd_encode_units(c, o, src, unit_off, n_units, dst, dst_cap, out_off);
    c->stream_mode = 0;
    return s;
}

// Streams with Flush points.  cut_off: n_units+1 indices into cuts; cuts[cut_off[i] .. cut_off[i+1]) = for stream i, ascending, the
// number of bytes that had been written when Flush was called.
static kc_status check_cuts(kc_ctx* c, const uint64_t* unit_off, uint32_t n_units, const uint64_t* cut_off, const uint64_t* cuts) {
    if (!cut_off || (cut_off[n_units] > cut_off[0] && !cuts)) return KC_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n_units; i++) {
        if (cut_off[i + 1] < cut_off[i]) { c->err = "cut_off not ascending"; return KC_ERR_BAD_ARG; }
        for (uint64_t k = cut_off[i]; k + 1 < cut_off[i + 1]; k++)
            if (cuts[k + 1] < cuts[k]) { c->err = "cuts of a stream not ascending"; return KC_ERR_BAD_ARG; }
    }
    (void)unit_off;
    return KC_OK;
}

// The block plan of one stream with Flush points, as the device path lays it out (host logic only; tests without a GPU).
int64_t kc_zstd_plan_stream_blocks(int32_t block_size, uint64_t len, const uint64_t* cuts, uint64_t n_cuts, uint32_t* starts, uint64_t starts_cap,
                                   uint32_t* flags) {
    if (block_size <= 0 || !flags || (n_cuts && !cuts)) return -1;
    std::vector<uint32_t> st;
    const uint32_t n = plan_stream_blocks((uint64_t)block_size, len, cuts, n_cuts, &st, flags);
    if (starts) { if (st.size() > starts_cap) return -2; for (size_t i = 0; i < st.size(); i++) starts[i] = st[i]; }
    return (int64_t)n;
}

kc_status kc_zstd_encode_streams_cuts_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                          const uint64_t* cut_off, const uint64_t* cuts, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_cuts(c, unit_off, n_units, cut_off, cuts);
    if (s != KC_OK) return s;
    static const uint64_t none = 0;
    c->cut_off = cut_off;
    c->cuts = cuts ? cuts : &none;
    c->cut_unit0 = 0;
    s = kc_zstd_encode_streams_dev(c, o, d_src, unit_off, n_units, d_dst, dst_cap, out_off);
    c->cuts = nullptr;
    c->cut_off = nullptr;
    return s;
}

kc_status kc_zstd_encode_streams_cuts(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                                      const uint64_t* cut_off, const uint64_t* cuts, uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o || !unit_off) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_cuts(c, unit_off, n_units, cut_off, cuts);
    if (s != KC_OK) return s;
    static const uint64_t none = 0;
    c->cut_off = cut_off;
    c->cuts = cuts ? cuts : &none;
    c->cut_unit0 = 0;
    s = kc_zstd_encode_streams(c, o, src, unit_off, n_units, dst, dst_cap, out_off);
    c->cuts = nullptr;
    c->cut_off = nullptr;
    return s;
}

// ---------------------------------------------------------------------------------------
// WithConcurrentBlocks (zstd/enc_jobs.go, encoder.go:214-247, 585-597, 652-700): ONE stream cut into jobs of
// max(4 * window, 512 KiB) input bytes, each encoded on a freshly reset encoder whose history is the last overlapSize bytes of
// the previous job's input (ResetPrefix), the job outputs concatenated behind one frame header.  The jobs are independent
// units for the device: unit k = [overlap prefix || job input] in a work buffer, its table primed from the prefix on the host
// exactly as ResetPrefix does (enc_fast.go:800-811, enc_dfast.go:1040-1050, enc_better.go:1099-1112).
// ---------------------------------------------------------------------------------------
}  // extern "C"

namespace {

uint64_t xxh64_host(const uint8_t* p, size_t len) {  // zstd/internal/xxhash/xxhash.go:27-230, seed 0
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    auto rol = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto rd64 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { return rol(acc + in * P2, 31) * P1; };
    auto merge = [&](uint64_t acc, uint64_t v) { return (acc ^ round(0, v)) * P1 + P4; };
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        for (; p + 32 <= end; p += 32) { v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24)); }
        h = rol(v1, 1) + rol(v2, 7) + rol(v3, 12) + rol(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = P5;
    }
    h += (uint64_t)len;
    for (; p + 8 <= end; p += 8) { h ^= round(0, rd64(p)); h = rol(h, 27) * P1 + P4; }
    if (p + 4 <= end) { uint32_t v; memcpy(&v, p, 4); h ^= (uint64_t)v * P1; h = rol(h, 23) * P2 + P3; p += 4; }
    for (; p < end; p++) { h ^= (uint64_t)(*p) * P5; h = rol(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// One job's tables as ResetPrefix leaves them, in the device entry format ((position + 1) | tag << pos_bits, position counted
// from the start of the prefix).  `out` = the unit's table slot (match_table_bytes(level)), zeroed by the caller.
void build_prefix_tables(int level, const uint8_t* prefix, size_t n, int pos_bits, uint8_t* out) {
    const int TB = (32 - pos_bits) > 16 ? 16 : (32 - pos_bits);
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    auto mk = [&](size_t pos, uint32_t val) -> uint32_t { return ((uint32_t)pos + 1u) | (tagOf(val) << pos_bits); };
    auto ld = [&](size_t i) -> uint64_t { uint64_t v; memcpy(&v, prefix + i, 8); return v; };
    if (n < 8) return;
    const size_t end = n - 8;
    if (level == KC_SPEED_BETTER) {  // enc_better.go:1099-1112: i = 0, 2, ... : long table with its chain, short table one byte on
        uint32_t* ltab = (uint32_t*)out;  // pairs {offset, prev}
        uint32_t* stab = (uint32_t*)(out + ((size_t)8 << 19));
        for (size_t i = 0; i < end; i += 2) {
            const uint64_t cv = ld(i);
            const uint32_t h = (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 19));
            const uint32_t old = ltab[2 * h];
            ltab[2 * h] = mk(i, (uint32_t)cv);
            ltab[2 * h + 1] = old;
            const uint64_t v = cv >> 8;
            stab[(uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 13))] = mk(i + 1, (uint32_t)v);
        }
        return;
    }
    // fastEncoder.ResetPrefix (enc_fast.go:800-811): every 4th position from 1, 6-byte hash, 2^15 entries.  doubleFastEncoder embeds
    // it (enc_dfast.go:1040-1041): the same entries land in ITS short table, although its lookups hash 5 bytes — kept as it is.
    uint32_t* ftab = level == KC_SPEED_DEFAULT ? (uint32_t*)(out + ((size_t)4 << 17)) : (uint32_t*)out;
    for (size_t i = 1; i < end; i += 4) {
        const uint64_t cv = ld(i);
        ftab[(uint32_t)(((cv << 16) * 227718039650203ULL) >> (64 - 15))] = mk(i, (uint32_t)cv);
    }
    if (level == KC_SPEED_DEFAULT) {  // enc_dfast.go:1042-1050: every 2nd position from 1 into the long table
        uint32_t* ltab = (uint32_t*)out;
        for (size_t i = 1; i < end; i += 2) {
            const uint64_t cv = ld(i);
            ltab[(uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 17))] = mk(i, (uint32_t)cv);
        }
    }
}

}  // namespace

extern "C" {

int64_t kc_zstd_job_size(const kc_zstd_opts* o) {  // encoderOptions.jobSize, encoder_options.go:356-359
    return o ? std::max<int64_t>((int64_t)o->window_size * 4, (int64_t)512 << 10) : -1;
}
int64_t kc_zstd_overlap_size(const kc_zstd_opts* o) {  // encoderOptions.overlapSize, encoder_options.go:362-371
    if (!o) return -1;
    return o->level == KC_SPEED_BEST ? o->window_size / 2 : (o->level == KC_SPEED_BETTER ? o->window_size / 4 : o->window_size / 8);
}

kc_status kc_zstd_encode_jobs(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, uint64_t len, const uint64_t* cuts, uint64_t n_cuts,
                              uint8_t* dst, uint64_t dst_cap, uint64_t* out_len) {
    if (!c || !o || !out_len || (len && (!src || !dst)) || (n_cuts && !cuts)) return KC_ERR_BAD_ARG;
    if (c->job_active) return KC_ERR_BAD_ARG;  // a submitted call is still in flight on this context: kc_wait first (c->err belongs to its thread)
    c->err.clear();
    *out_len = 0;
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    if (o->dict != nullptr && o->dict_len > 0) {
        c->err = "the reference switches WithConcurrentBlocks off when a dictionary is set (zstd/encoder.go:81,174): use kc_zstd_encode_streams";
        return KC_ERR_UNSUPPORTED;
    }
    for (uint64_t i = 1; i < n_cuts; i++) if (cuts[i] < cuts[i - 1]) { c->err = "flush points not ascending"; return KC_ERR_BAD_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t jobSize = (uint64_t)kc_zstd_job_size(o), overlap = (uint64_t)kc_zstd_overlap_size(o);
    // jobs dispatched before Close: `filling` reached jobSize during Write (encoder.go:239-244), or a Flush found bytes in it (:587-591)
    std::vector<uint64_t> lo, hi;
    uint64_t pos = 0, ci = 0;
    for (;;) {
        while (ci < n_cuts && cuts[ci] <= pos) ci++;
        uint64_t e = pos + jobSize;
        bool dispatched = e <= len;
        if (ci < n_cuts && cuts[ci] < e && cuts[ci] <= len) { e = cuts[ci]; dispatched = true; }
        if (!dispatched) break;
        lo.push_back(pos);
        hi.push_back(e);
        pos = e;
    }
    const uint64_t tail = len - pos;  // what Close finds in `filling`
    if (lo.empty()) {  // dispatchJob(true) before any header was written (enc_jobs.go:263-289)
        if (tail > 0 && tail <= (uint64_t)o->block_size) {  // single block: the EncodeAll frame
            const uint64_t uo[2] = {0, len};
            uint64_t oo[2] = {0, 0};
            s = kc_zstd_encode_units(c, o, src, uo, 1, dst, dst_cap, oo);
            if (s == KC_OK) *out_len = oo[1];
            return s;
        }
        if (tail == 0 && !o->full_zero) return KC_OK;
    }
    lo.push_back(pos);
    hi.push_back(len);  // the final job (possibly empty)
    const uint32_t nj = (uint32_t)lo.size();
    // frame header (enc_jobs.go:291-304): no content size, window = the encoder's, not single segment, no dictionary id
    uint8_t hdr[8];
    int hl = 0;
    hdr[hl++] = 0x28; hdr[hl++] = 0xb5; hdr[hl++] = 0x2f; hdr[hl++] = 0xfd;
    hdr[hl++] = o->crc ? (uint8_t)(1 << 2) : (uint8_t)0;
    hdr[hl++] = (uint8_t)((bitsLen32((uint32_t)o->window_size - 1) - 10) << 3);
    // work buffer: unit k = [prefix_k || job_k]; prefix_k = the last min(overlap, len(job k-1)) bytes of job k-1 (enc_jobs.go:325-331)
    std::vector<uint32_t> jhist(nj), jflags(nj);
    std::vector<uint64_t> jlen(nj);
    for (uint32_t k = 0; k < nj; k++) {
        const uint64_t ov = k == 0 ? 0 : std::min<uint64_t>(overlap, hi[k - 1] - lo[k - 1]);
        jhist[k] = (uint32_t)ov;
        jflags[k] = k + 1 == nj ? 1u : 0u;
        jlen[k] = ov + (hi[k] - lo[k]);
        if (jlen[k] > KC_MAX_UNIT_BYTES) { c->err = "job larger than 1 GiB: not served by the device path"; return KC_ERR_UNSUPPORTED; }
    }
    const size_t tb = match_table_bytes(o->level);
    uint64_t crc = 0;
    std::thread crcT;
    if (o->crc) crcT = std::thread([&] { crc = xxh64_host(src, (size_t)len); });  // under the device work
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } joinCrc{crcT};
    struct Unhook { kc_ctx* c; ~Unhook() { c->job_hist = nullptr; c->job_flags = nullptr; c->job_tables = nullptr; c->job_primed = false; c->job_redo_list.clear(); } } unhook{c};
    if ((uint64_t)hl > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
    memcpy(dst, hdr, (size_t)hl);
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    c->last_batches = 0;
    // The jobs go to the device in batches bounded like kc_zstd_encode_units_dev's: by input bytes and by the scratch their
    // tables and per-block strides ask for (a long stream at a small window is thousands of jobs).  Each batch's tables are
    // primed for the batch only (on the device; with KC_OPT_JOB_PRIME 0 on the host, whose memory the same budget then bounds).
    uint64_t budget = scratch_budget(c);
    uint64_t done = 0;  // frame bytes behind the header so far
    std::vector<uint8_t> tabs;
    std::vector<uint64_t> boff, oo;
    uint32_t k0 = 0;
    for (int attempt = 0; k0 < nj;) {
        uint32_t k1 = k0;
        uint64_t scratch = 0, bytes = 0, need = 0, maxUnit = 16;
        while (k1 < nj) {
            const uint64_t us = zstd_unit_scratch(o, jlen[k1]);
            if (k1 > k0 && (bytes + jlen[k1] > c->max_batch_bytes || (scratch + us) + ((scratch + us) >> 3) > budget)) break;
            scratch += us;
            bytes += jlen[k1];
            need += ((uint64_t)kc_zstd_max_encoded_size(o, (int64_t)jlen[k1]) + 15) & ~(uint64_t)15;
            maxUnit = std::max(maxUnit, jlen[k1]);
            k1++;
        }
        const uint32_t nb = k1 - k0;
        int pos_bits = 1;
        while (((uint64_t)1 << pos_bits) <= maxUnit + 2) pos_bits++;  // batch_begin's, for this batch
        const bool primeHost = tb != 0 && c->cfg.job_prime == 0;  // (default: kc_zstd_prime_kernel, from the prefix bytes staged below)
        if (primeHost) {
            try { tabs.assign((size_t)nb * tb, 0); } catch (...) { c->err = "host memory for the jobs' tables"; return KC_ERR_UNSUPPORTED; }
            const int T = std::max(1, std::min<int>(host_copy_threads(c), (int)nb));
            std::vector<std::thread> th;
            std::atomic<uint32_t> next{0};
            for (int t = 0; t < T; t++)
                th.emplace_back([&] {
                    for (uint32_t k = next++; k < nb; k = next++)
                        if (jhist[k0 + k]) build_prefix_tables(o->level, src + lo[k0 + k] - jhist[k0 + k], jhist[k0 + k], pos_bits, tabs.data() + (size_t)k * tb);
                });
            for (auto& x : th) x.join();
        }
        if ((s = ensure(c, c->tmp_src, bytes + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
        boff.assign(nb + 1, 0);
        for (uint32_t k = 0; k < nb; k++) {
            boff[k + 1] = boff[k] + jlen[k0 + k];
            if (jlen[k0 + k]) HIPCHK(c, hipMemcpyAsync((uint8_t*)c->tmp_src.p + boff[k], src + lo[k0 + k] - jhist[k0 + k], jlen[k0 + k], hipMemcpyHostToDevice, c->stream));
        }
        oo.assign(nb + 1, 0);
        uint64_t produced = 0;
        c->job_hist = jhist.data() + k0;
        c->job_flags = jflags.data() + k0;
        c->job_tables = primeHost ? tabs.data() : nullptr;
        c->job_primed = tb != 0;  // (SpeedBestCompression: the kernel indexes each job's prefix itself)
        c->oom = false;
        s = run_batch(c, o, (const uint8_t*)c->tmp_src.p, boff.data(), nb, (uint8_t*)c->tmp_dst.p, need, oo.data(), &produced);
        c->job_redo_list.clear();
        if (s == KC_ERR_UNSUPPORTED && c->oom && nb > 1 && attempt < 6) {
            attempt++;
            budget /= 2;  // another process took device memory since hipMemGetInfo: this batch again at half the size
            c->err.clear();
            continue;
        }
        if (s != KC_OK) return s;
        c->last_batches++;
        if ((uint64_t)hl + done + produced + (o->crc ? 4 : 0) > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
        if (produced) HIPCHK(c, hipMemcpyAsync(dst + hl + done, c->tmp_dst.p, produced, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (tmp_src / tmp_dst and the host tables are the next batch's)
        done += produced;
        k0 = k1;
        if (attempt) { attempt = 0; budget = scratch_budget(c); }  // the squeeze was this batch's: later batches start from what is free now
    }
    if (crcT.joinable()) crcT.join();
    const uint64_t total = (uint64_t)hl + done + (o->crc ? 4 : 0);
    if (o->crc) for (int k = 0; k < 4; k++) dst[hl + done + k] = (uint8_t)(crc >> (8 * k));
    *out_len = total;
    return KC_OK;
}

// Asynchronous form of the host-buffer entry points: submit returns at once, the call runs on a thread of its own (staging,
// kernels and drain of a batch are already overlapped inside one call; with two contexts a caller also overlaps consecutive
// batches: submit(A, batch k+1) while wait(B) drains batch k).  One job per context; every buffer, and the option struct's
// dictionary, must stay valid until kc_wait returns the job's status.
kc_status kc_zstd_encode_units_submit(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off, uint32_t n_units,
                                      uint8_t* dst, uint64_t dst_cap, uint64_t* out_off) {
    if (!c || !o) return KC_ERR_BAD_ARG;
    if (c->job_active) return KC_ERR_BAD_ARG;  // a submitted job is still in flight: kc_wait first (c->err belongs to the job's thread)
    const kc_zstd_opts oc = *o;
    c->job_active = true;
    c->job = std::thread([=] { c->job_status = kc_zstd_encode_units(c, &oc, src, unit_off, n_units, dst, dst_cap, out_off); });
    return KC_OK;
}

kc_status kc_s2_encode_blocks_lvl_submit(kc_ctx* c, int level, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* dst,
                                         uint64_t dst_cap, uint64_t* out_off) {
    if (!c) return KC_ERR_BAD_ARG;
    if (c->job_active) return KC_ERR_BAD_ARG;  // a submitted job is still in flight: kc_wait first (c->err belongs to the job's thread)
    c->job_active = true;
    c->job = std::thread([=] { c->job_status = kc_s2_encode_blocks_lvl(c, level, src, blk_off, n, dst, dst_cap, out_off); });
    return KC_OK;
}

kc_status kc_wait(kc_ctx* c) {
    if (!c) return KC_ERR_BAD_ARG;
    if (!c->job_active) { c->err = "no submitted job on this context"; return KC_ERR_BAD_ARG; }
    c->job.join();
    c->job_active = false;
    return c->job_status;
}

kc_status kc_xxh64_units_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units, uint64_t* out_hash) {
    if (!c || !unit_off || !out_hash || (n_units && !d_src)) return KC_ERR_BAD_ARG;
    c->err.clear();
    HIPCHK(c, hipSetDevice(c->device));
    if (n_units == 0) return KC_OK;
    kc_status s;
    if ((s = ensure(c, c->unit_off, (n_units + 1) * 8)) || (s = ensure(c, c->xxh, (size_t)n_units * 8))) return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, unit_off, (n_units + 1) * 8, hipMemcpyHostToDevice, c->stream));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    kc_launch_xxh64(d_src, (const uint64_t*)c->unit_off.p, n_units, (uint64_t*)c->xxh.p, c->stream);
    HIPCHK(c, hipMemcpyAsync(out_hash, c->xxh.p, (size_t)n_units * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return KC_OK;
}

kc_status kc_zstd_debug_parse_dev(kc_ctx* c, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units,
                                  uint32_t* seqs, uint64_t seq_cap, uint64_t* blk_first_seq, uint32_t* blk_extra_lits, uint32_t* blk_flags,
                                  uint32_t blk_cap, uint32_t* n_blocks_out) {
    if (!c || !o || !unit_off || !seqs || !blk_first_seq || !blk_extra_lits || !n_blocks_out) return KC_ERR_BAD_ARG;
    c->err.clear();
    kc_status s = check_supported(c, o);
    if (s != KC_OK) return s;
    HIPCHK(c, hipSetDevice(c->device));
    const int bs = o->block_size;
    std::vector<uint32_t> blk0(n_units + 1);
    uint32_t nb = 0;
    for (uint32_t i = 0; i < n_units; i++) { blk0[i] = nb; nb += (uint32_t)((unit_off[i + 1] - unit_off[i] + bs - 1) / bs); }
    blk0[n_units] = nb;
    if (nb > blk_cap) return KC_ERR_DST_TOO_SMALL;
    const uint32_t seq_stride = (uint32_t)(bs / 4 + 8);
    if ((s = ensure(c, c->unit_off, (n_units + 1) * 8)) || (s = ensure(c, c->unit_blk0, (n_units + 1) * 4)) ||
        (s = ensure(c, c->seqs, (size_t)nb * seq_stride * 8)) || (s = ensure(c, c->meta, (size_t)nb * sizeof(KcBlkMeta))))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, unit_off, (n_units + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->unit_blk0.p, blk0.data(), (n_units + 1) * 4, hipMemcpyHostToDevice, c->stream));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    KcMatchParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.src = d_src;
    mp.src_end = d_src + unit_off[n_units];
    mp.unit_off = (const uint64_t*)c->unit_off.p;
    mp.unit_blk0 = (const uint32_t*)c->unit_blk0.p;
    mp.seqs = (uint64_t*)c->seqs.p;
    mp.meta = (KcBlkMeta*)c->meta.p;
    mp.seq_stride = seq_stride;
    mp.block_size = bs;
    mp.max_match_off = o->window_size;
    mp.spec_w0 = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : 1;
    mp.spec_grow = c->cfg.spec_grow >= 0 ? (int)c->cfg.spec_grow : 2;
    if (mp.spec_w0 < 1) mp.spec_w0 = 1;
    if (mp.spec_w0 > 8) mp.spec_w0 = 8;
    mp.hist0 = 0;
    mp.rep1 = 1;
    mp.rep2 = 4;
    mp.rep3 = 8;
    {
        uint64_t maxLen = 16;
        for (uint32_t i = 0; i < n_units; i++) maxLen = std::max<uint64_t>(maxLen, unit_off[i + 1] - unit_off[i]);
        c->plan.max_unit_bytes = maxLen;
        int pb = 1;
        while (((uint64_t)1 << pb) <= maxLen + 2) pb++;
        mp.pos_bits = pb;
    }
    if (o->dict != nullptr && o->dict_len > 0) { c->err = "debug parse does not take dictionaries"; return KC_ERR_UNSUPPORTED; }
    if ((s = launch_match(c, mp, unit_off, n_units, n_units, bs, c->stream, o->level)) != KC_OK) return s;
    std::vector<KcBlkMeta> meta(nb);
    HIPCHK(c, hipMemcpyAsync(meta.data(), c->meta.p, (size_t)nb * sizeof(KcBlkMeta), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    uint64_t total = 0;
    std::vector<uint64_t> packed;
    for (uint32_t b = 0; b < nb; b++) {
        blk_first_seq[b] = total;
        blk_extra_lits[b] = meta[b].extra_lits;
        if (blk_flags) blk_flags[b] = meta[b].flags;
        const uint32_t n = meta[b].nseq;
        if (total + n > seq_cap) return KC_ERR_DST_TOO_SMALL;
        packed.resize(n);
        if (n) HIPCHK(c, hipMemcpy(packed.data(), (const uint64_t*)c->seqs.p + (size_t)b * seq_stride, (size_t)n * 8, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < n; k++) {
            const uint64_t v = packed[k];
            seqs[3 * (total + k) + 0] = (uint32_t)(v >> 44);
            seqs[3 * (total + k) + 1] = (uint32_t)((v >> 24) & 0xFFFFF);
            seqs[3 * (total + k) + 2] = (uint32_t)(v & 0xFFFFFF);
        }
        total += n;
    }
    blk_first_seq[nb] = total;
    *n_blocks_out = nb;
    return KC_OK;
}

// ---------------------------------------------------------------------------------------
// S2 (kernels in kc_s2.hip)
// ---------------------------------------------------------------------------------------
int64_t kc_s2_max_encoded_len(int64_t srcLen) {  // s2/encode.go:389-418 (64-bit int)
    uint64_t n = (uint64_t)srcLen;
    if (n > 0xffffffffULL) return -1;
    int lb = n == 0 ? 0 : 64 - __builtin_clzll(n);
    n = n + (uint64_t)((lb + 7) / 7);
    int64_t extra = srcLen == 0 ? 0 : (srcLen < 60 ? 1 : (srcLen < (1 << 8) ? 2 : (srcLen < (1 << 16) ? 3 : (srcLen < (1 << 24) ? 4 : 5))));
    n += (uint64_t)extra;
    if (n > 0xffffffffULL) return -1;
    return (int64_t)n;
}

static kc_status s2_encode_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                               uint64_t dst_cap, uint64_t* out_off, int framed, int with_stream_id, int level = KC_S2_LEVEL_DEFAULT,
                               ChunkFeed* feed = nullptr) {
    if (!c || !blk_off || !out_off || (n && (!d_src || !d_dst))) return KC_ERR_BAD_ARG;
    if (feed && (framed || n == 0)) { c->err = "chunk feed: bare blocks only"; return KC_ERR_INTERNAL; }
    // s2.WriterUncompressed: a level of the writer (writer.go:951; encodeBlock returns 0 for it, :455-480): framed only, every block one
    // uncompressed chunk — served by the LDS-table kernels' stored path (wave-parallel CRC32C + copy), whatever the batch
    const bool stored_only = level == KC_S2_LEVEL_UNCOMPRESSED;
    if (stored_only) {
        if (!framed) { c->err = "KC_S2_LEVEL_UNCOMPRESSED is a level of the framed stream (s2.WriterUncompressed)"; return KC_ERR_BAD_ARG; }
        level = KC_S2_LEVEL_DEFAULT;
    }
    if (level < KC_S2_LEVEL_DEFAULT || level > KC_S2_LEVEL_SNAPPY_BEST) { c->err = "unknown S2 level"; return KC_ERR_UNSUPPORTED; }
    if (level >= KC_S2_LEVEL_BEST && feed) { c->err = "the best levels are not chunk-fed"; return KC_ERR_UNSUPPORTED; }
    c->err.clear();
    c->last = kc_timings{0, 0, 0, 0, 0, 0};
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t lead = (framed && with_stream_id) ? 10 : 0;
    if (lead) {
        static const uint8_t magic[10] = {0xff, 0x06, 0x00, 0x00, 'S', '2', 's', 'T', 'w', 'O'};  // magicChunk, s2/s2.go:79
        if (dst_cap < 10) return KC_ERR_DST_TOO_SMALL;
        HIPCHK(c, hipMemcpyAsync(d_dst, magic, 10, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (n == 0) { out_off[0] = lead; return KC_OK; }
    hipStream_t st = c->stream;
    std::vector<uint64_t>& rel = c->plan.rel_off;   // in the context: the asynchronous copies below outlive this call when chunk-fed
    std::vector<uint64_t>& so = c->plan.stage64;     // staging slots: whole 64-byte lines (the kernel stores its output line by line)
    std::vector<uint64_t>& reg = c->plan.stage_off;  // 16-byte aligned bounds: what dst_cap is checked against, chunk regions when chunk-fed
    rel.resize(n + 1);
    so.resize(n + 1);
    reg.resize(n + 1);
    uint64_t acc = 0, acc16 = 0, maxLen = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (blk_off[i + 1] < blk_off[i]) { c->err = "blk_off not ascending"; return KC_ERR_BAD_ARG; }
        const uint64_t len = blk_off[i + 1] - blk_off[i];
        maxLen = std::max(maxLen, len);
        if (len > KC_S2_MAX_BLOCK) { c->err = "S2 block larger than 1 GiB not served by the device path"; return KC_ERR_UNSUPPORTED; }
        // a framed block is a chunk of the stream format: its header holds a 24-bit length and the reference's Reader refuses chunks of
        // blocks above s2.maxBlockSize (4 MiB; s2/s2.go, writer.go:986-989 WriterBlockSize) — the 1 GiB bound is for bare blocks only
        if (framed && len > KC_S2_MAX_FRAMED_BLOCK) { c->err = "framed S2 block larger than 4 MiB (s2.maxBlockSize): not a valid chunk of the stream format"; return KC_ERR_BAD_ARG; }
        rel[i] = blk_off[i] - blk_off[0];
        so[i] = acc;
        reg[i] = acc16;
        acc += ((uint64_t)kc_s2_max_encoded_len((int64_t)len) + (framed ? 8 : 0) + 63) & ~(uint64_t)63;
        acc16 += ((uint64_t)kc_s2_max_encoded_len((int64_t)len) + (framed ? 8 : 0) + 15) & ~(uint64_t)15;
    }
    rel[n] = blk_off[n] - blk_off[0];
    so[n] = acc;
    reg[n] = acc16;
    if (acc16 + lead > dst_cap) { c->err = "dst_cap smaller than the sum of MaxEncodedLen(block)"; return KC_ERR_DST_TOO_SMALL; }
    // s2.Encode / s2.EncodeSnappy: the LDS-table kernel (one wave per block, ~1 ms per 64 KiB block whatever the batch) while the
    // blocks in flight cannot cover the HBM-table kernel's latency (measured crossover: profiles/r03_crossover_s2.csv)
    // The best levels are pure Go in the reference — one form on every platform (s2/encode_best.go) — so the variant does not
    // apply to them: an amd64 context (the Go shim's default on amd64 builds) encodes them like any other.
    const int s2var = level >= KC_S2_LEVEL_BEST ? KC_S2_VARIANT_GO : (int)c->cfg.s2_variant;
    // (the LDS kernel keeps positions in 24 bits: a batch with a block of 16 MiB or more goes through the HBM-table kernel whole)
    const bool lds = stored_only || ((level == KC_S2_LEVEL_DEFAULT || level == KC_S2_LEVEL_SNAPPY) && feed == nullptr && c->cfg.match_path != KC_PATH_HBM &&
                     maxLen < ((uint64_t)1 << 24) && (c->cfg.match_path == KC_PATH_LDS || (int64_t)n <= c->cfg.s2_lds_max_blocks));
    c->last_path = lds ? KC_PATH_LDS : KC_PATH_HBM;
    kc_status s;
    if ((s = ensure(c, c->unit_off, (n + 1) * 8)) || (s = ensure(c, c->stage_off, (n + 1) * 8)) ||
        (s = ensure(c, c->out_off, (n + 1 + (feed ? feed->cut.size() : 0)) * 8)) || (s = ensure(c, c->stage, acc + 64)) || (s = ensure(c, c->out_size, (size_t)n * 4)) ||
        (!lds && (s = ensure(c, c->tables, (size_t)n * kc_s2_table_bytes(level, maxLen, s2var)))))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->stage_off.p, so.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    if (!lds) { c->tab_owner = 0; HIPCHK(c, hipMemsetAsync(c->tables.p, 0, (size_t)n * kc_s2_table_bytes(level, maxLen, s2var), st)); }
    KcS2Params P;
    P.src = d_src + blk_off[0];
    P.blk_off = (const uint64_t*)c->unit_off.p;
    P.stage_off = (const uint64_t*)c->stage_off.p;
    P.stage = (uint8_t*)c->stage.p;
    P.out_size = (uint32_t*)c->out_size.p;
    P.tables = (uint32_t*)c->tables.p;
    P.n_blocks = n;
    P.framed = framed;
    P.level = level;
    P.spec_w0 = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : 2;
    P.spec_w0b = c->cfg.spec_w0 >= 0 ? (int)c->cfg.spec_w0 : 4;
    P.spec_grow = c->cfg.spec_grow >= 0 ? (int)c->cfg.spec_grow : 1;
    if (P.spec_w0 < 1) P.spec_w0 = 1;
    if (P.spec_w0b < 1) P.spec_w0b = 1;
    P.table_stride = (uint32_t)(kc_s2_table_bytes(level, maxLen, s2var) / 4);
    P.variant = (int32_t)s2var;
    P.stored_only = stored_only ? 1 : 0;
    if (feed) {
        // the source is still arriving: per chunk, encode + compaction on the chunk's stream behind its H2D copy; frames of chunk k
        // at d_dst + reg[cut[k]], local offsets in out_off[cut[k] + k ...].  The caller synchronises (s2_feed_finish).
        HIPCHK(c, hipEventRecord(c->ev[6], st));
        feed->loc_off = (uint64_t*)c->out_off.p;
        const size_t nchunk = feed->cut.size() - 1;
        for (size_t k = 0; k < nchunk; k++) {
            if (!feed->wait_recorded(k)) { c->err = "host pipeline: staging failed"; return KC_ERR_HIP; }
            hipStream_t sk = feed->streams[k % feed->streams.size()];
            const uint32_t u0 = feed->cut[k], nk = feed->cut[k + 1] - u0;
            HIPCHK(c, hipStreamWaitEvent(sk, c->ev[6], 0));
            HIPCHK(c, hipStreamWaitEvent(sk, feed->landed[k], 0));
            KcS2Params Pk = P;
            Pk.blk_off += u0;
            Pk.stage_off += u0;
            Pk.out_size += u0;
            Pk.tables += (size_t)u0 * P.table_stride;
            Pk.n_blocks = nk;
            kc_launch_s2_encode(Pk, sk);
            kc_launch_scan_sizes(Pk.out_size, nk, feed->loc_off + u0 + k, sk);
            kc_launch_compact((const uint8_t*)c->stage.p, Pk.stage_off, Pk.out_size, feed->loc_off + u0 + k, d_dst + reg[u0], nk, sk);
            HIPCHK(c, hipEventRecord(feed->done[k], sk));
        }
        for (size_t k = 0; k < nchunk; k++) HIPCHK(c, hipStreamWaitEvent(st, feed->done[k], 0));
        HIPCHK(c, hipGetLastError());
        return KC_OK;
    }
    if (level >= KC_S2_LEVEL_BEST) {
        kc_launch_s2_best(P, st);
    } else if (lds) {
        bool any_small = false, any_big = false;
        for (uint32_t i = 0; i < n; i++) ((blk_off[i + 1] - blk_off[i]) <= ((uint64_t)64 << 10) ? any_small : any_big) = true;
        P.spec_w0 = (int32_t)c->cfg.s2_lds_spec_w0;
        kc_launch_s2_encode_lds(P, any_small, any_big, st);
    } else {
        kc_launch_s2_encode(P, st);
    }
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    kc_launch_scan_sizes((const uint32_t*)c->out_size.p, n, (uint64_t*)c->out_off.p, st);
    kc_launch_compact((const uint8_t*)c->stage.p, (const uint64_t*)c->stage_off.p, (const uint32_t*)c->out_size.p,
                      (const uint64_t*)c->out_off.p, d_dst + lead, n, st);
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    HIPCHK(c, hipMemcpyAsync(out_off, c->out_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    for (uint32_t i = 0; i <= n; i++) out_off[i] += lead;
    float t01 = 0, t12 = 0;
    (void)hipEventElapsedTime(&t01, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&t12, c->ev[1], c->ev[2]);
    c->last.match_ms = t01;
    c->last.other_ms = t12;
    c->last.total_ms = t01 + t12;
    return KC_OK;
}

// zstd frame decode over N units on the device (verifier): decode, then XXH64 of the output against the stored checksum.
kc_status kc_zstd_decode_units_dev(kc_ctx* c, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n, uint8_t* d_dst,
                                   const uint64_t* dst_off, uint32_t* status) {
    return kc_zstd_decode_units_dict_dev(c, d_enc, enc_off, n, d_dst, dst_off, status, nullptr, 0);
}

kc_status kc_zstd_decode_units_dict_dev(kc_ctx* c, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n, uint8_t* d_dst,
                                        const uint64_t* dst_off, uint32_t* status, const uint8_t* dict, uint64_t dict_len) {
    if (!c || !enc_off || !dst_off || !status || (n && (!d_enc || !d_dst))) return KC_ERR_BAD_ARG;
    if (dict_len > ((uint64_t)1 << 30)) return KC_ERR_BAD_ARG;
    c->err.clear();
    if (n == 0) return KC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    kc_status s;
    const uint32_t lit_stride = (128u << 10) + 64u;
    if ((s = ensure(c, c->unit_off, (size_t)(n + 1) * 8)) || (s = ensure(c, c->stage_off, (size_t)(n + 1) * 8)) ||
        (s = ensure(c, c->out_size, (size_t)n * 4)) || (s = ensure(c, c->redo, (size_t)n * 4)) || (s = ensure(c, c->popmask, (size_t)n * 4)) ||
        (s = ensure(c, c->xxh, (size_t)n * 8)) || (s = ensure(c, c->lits, (size_t)n * lit_stride)))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, enc_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->stage_off.p, dst_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    KcZstdDecParams P;
    P.enc = d_enc;
    P.enc_off = (const uint64_t*)c->unit_off.p;
    P.dst = d_dst;
    P.dst_off = (const uint64_t*)c->stage_off.p;
    P.lits = (uint8_t*)c->lits.p;
    P.lit_stride = lit_stride;
    P.status = (uint32_t*)c->out_size.p;
    P.crc_stored = (uint32_t*)c->redo.p;
    P.has_crc = (uint32_t*)c->popmask.p;
    P.n_units = n;
    P.dict = nullptr;
    P.dict_len = 0;
    if (dict != nullptr && dict_len > 0) {
        if ((s = ensure(c, c->dictbuf, (size_t)dict_len + 64))) return s;
        HIPCHK(c, hipMemcpyAsync(c->dictbuf.p, dict, (size_t)dict_len, hipMemcpyHostToDevice, st));
        P.dict = (const uint8_t*)c->dictbuf.p;
        P.dict_len = (uint32_t)dict_len;
    }
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    kc_launch_zstd_decode(P, st);
    kc_launch_xxh64(d_dst, (const uint64_t*)c->stage_off.p, n, (uint64_t*)c->xxh.p, st);
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    std::vector<uint32_t> stored(n), has(n);
    std::vector<uint64_t> hashes(n);
    HIPCHK(c, hipMemcpyAsync(status, c->out_size.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(stored.data(), c->redo.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(has.data(), c->popmask.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(hashes.data(), c->xxh.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    for (uint32_t i = 0; i < n; i++)
        if (status[i] == 0 && has[i] && (uint32_t)hashes[i] != stored[i]) status[i] = 30;  // checksum mismatch (framedec.go:310-325)
    float t = 0;
    (void)hipEventElapsedTime(&t, c->ev[0], c->ev[1]);
    c->last = kc_timings{t, t, 0, 0, 0};
    return KC_OK;
}

// s2.Decode over N blocks on the device (verifier).  status[i] (host) receives 0 or the first error of block i.
kc_status kc_s2_decode_blocks_dev(kc_ctx* c, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n, uint8_t* d_dst,
                                  const uint64_t* dst_off, uint32_t* status) {
    if (!c || !enc_off || !dst_off || !status || (n && (!d_enc || !d_dst))) return KC_ERR_BAD_ARG;
    c->err.clear();
    if (n == 0) return KC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    kc_status s;
    if ((s = ensure(c, c->unit_off, (size_t)(n + 1) * 8)) || (s = ensure(c, c->stage_off, (size_t)(n + 1) * 8)) ||
        (s = ensure(c, c->out_size, (size_t)n * 4)))
        return s;
    HIPCHK(c, hipMemcpyAsync(c->unit_off.p, enc_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->stage_off.p, dst_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    c->up_ptr[0] = c->up_ptr[1] = c->up_ptr[2] = nullptr;  // (the zstd batch path re-uploads its layout arrays)
    KcS2DecParams P;
    P.enc = d_enc;
    P.enc_off = (const uint64_t*)c->unit_off.p;
    P.dst = d_dst;
    P.dst_off = (const uint64_t*)c->stage_off.p;
    P.status = (uint32_t*)c->out_size.p;
    P.n_blocks = n;
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    kc_launch_s2_decode(P, st);
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipMemcpyAsync(status, c->out_size.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    float t = 0;
    (void)hipEventElapsedTime(&t, c->ev[0], c->ev[1]);
    c->last = kc_timings{t, t, 0, 0, 0};
    return KC_OK;
}

// s2_encode_dev in batches that fit the scratch budget: the HBM path keeps a table per block (64 KiB default / snappy, 288-576 KiB
// better), whatever the block's length, plus a MaxEncodedLen staging slot — half a million small blocks would ask for > 100 GiB.
static kc_status s2_encode_dev_budgeted(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                        uint64_t dst_cap, uint64_t* out_off, int framed, int with_stream_id, int level) {
    if (!c || !blk_off || !out_off) return KC_ERR_BAD_ARG;
    if (level < KC_S2_LEVEL_DEFAULT || level > KC_S2_LEVEL_SNAPPY_BEST || n == 0) return s2_encode_dev(c, d_src, blk_off, n, d_dst, dst_cap, out_off, framed, with_stream_id, level);
    uint64_t maxLen = 0;
    for (uint32_t i = 0; i < n; i++) if (blk_off[i + 1] >= blk_off[i]) maxLen = std::max(maxLen, blk_off[i + 1] - blk_off[i]);
    const uint64_t tb = kc_s2_table_bytes(level, maxLen, level >= KC_S2_LEVEL_BEST ? KC_S2_VARIANT_GO : (int)c->cfg.s2_variant);
    uint64_t budget = scratch_budget(c);
    std::vector<uint64_t> tmp;
    uint64_t pos = 0;
    uint32_t i0 = 0;
    while (i0 < n) {
        uint32_t i1 = i0;
        uint64_t scratch = 0;
        while (i1 < n) {
            const uint64_t len = blk_off[i1 + 1] >= blk_off[i1] ? blk_off[i1 + 1] - blk_off[i1] : 0;
            const uint64_t us = tb + (((uint64_t)std::max<int64_t>(0, kc_s2_max_encoded_len((int64_t)len)) + 8 + 63) & ~(uint64_t)63);
            if (i1 > i0 && (scratch + us) + ((scratch + us) >> 3) > budget) break;
            scratch += us;
            i1++;
        }
        if (i0 == 0 && i1 == n) {  // the usual case: one batch
            const kc_status s1 = s2_encode_dev(c, d_src, blk_off, n, d_dst, dst_cap, out_off, framed, with_stream_id, level);
            c->last_batches = s1 == KC_OK ? 1 : 0;  // (0: nothing was encoded on the device — the Go shim's tests tell a fallback by it)
            return s1;
        }
        if (i0 == 0) c->last_batches = 0;
        const uint32_t nb = i1 - i0;
        tmp.resize(nb + 1);
        c->oom = false;
        kc_status s = s2_encode_dev(c, d_src, blk_off + i0, nb, d_dst + pos, dst_cap - pos, tmp.data(), framed, i0 == 0 ? with_stream_id : 0, level);
        if (s == KC_ERR_UNSUPPORTED && c->oom && nb > 1 && budget > ((uint64_t)64 << 20)) {
            budget /= 2;  // another process took device memory since hipMemGetInfo
            c->err.clear();
            continue;
        }
        if (s != KC_OK) return s;
        for (uint32_t k = 0; k <= nb; k++) out_off[i0 + k] = pos + tmp[k];
        pos += tmp[nb];
        i0 = i1;
        c->last_batches++;
    }
    return KC_OK;
}

kc_status kc_s2_encode_blocks_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                  uint64_t dst_cap, uint64_t* out_off) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 0, 0, KC_S2_LEVEL_DEFAULT);
}

kc_status kc_s2_encode_stream_dev(kc_ctx* c, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                  uint64_t dst_cap, uint64_t* out_off, int with_stream_id) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 1, with_stream_id, KC_S2_LEVEL_DEFAULT);
}

kc_status kc_s2_encode_blocks_lvl_dev(kc_ctx* c, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                      uint64_t dst_cap, uint64_t* out_off) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 0, 0, level);
}

kc_status kc_s2_encode_stream_lvl_dev(kc_ctx* c, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n, uint8_t* d_dst,
                                      uint64_t dst_cap, uint64_t* out_off, int with_stream_id) {
    return s2_encode_dev_budgeted(c, d_src, blk_off, n, d_dst, dst_cap, out_off, 1, with_stream_id, level);
}

kc_status kc_s2_encode_blocks(kc_ctx* c, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* dst, uint64_t dst_cap,
                              uint64_t* out_off) {
    return kc_s2_encode_blocks_lvl(c, KC_S2_LEVEL_DEFAULT, src, blk_off, n, dst, dst_cap, out_off);
}

kc_status kc_s2_encode_blocks_lvl(kc_ctx* c, int level, const uint8_t* src, const uint64_t* blk_off, uint32_t n, uint8_t* dst, uint64_t dst_cap,
                                  uint64_t* out_off) {
    if (!c || !blk_off || !out_off || (n && (!src || !dst))) return KC_ERR_BAD_ARG;
    if (level < KC_S2_LEVEL_DEFAULT || level > KC_S2_LEVEL_SNAPPY_BEST) { c->err = "unknown S2 level"; return KC_ERR_UNSUPPORTED; }
    c->err.clear();
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) { out_off[0] = 0; return KC_OK; }
    for (uint32_t i = 0; i < n; i++) {  // before any byte moves
        if (blk_off[i + 1] < blk_off[i]) { c->err = "blk_off not ascending"; return KC_ERR_BAD_ARG; }
        if (blk_off[i + 1] - blk_off[i] > KC_S2_MAX_BLOCK) { c->err = "S2 block larger than 1 GiB not served by the device path"; return KC_ERR_UNSUPPORTED; }
    }
    const uint64_t total = blk_off[n] - blk_off[0];
    const uint64_t ov_min = c->cfg.host_overlap_min_mib >= 0 ? (uint64_t)c->cfg.host_overlap_min_mib << 20 : (uint64_t)512 << 20;
    if (total >= ov_min && total <= c->max_batch_bytes && !c->cfg.host_serial && c->cfg.host_pipe_mib < 16 && level < KC_S2_LEVEL_BEST) {  // (the best levels: 4.5 MiB of tables per block, several device batches)
        uint64_t need = 0;
        for (uint32_t i = 0; i < n; i++) need += ((uint64_t)kc_s2_max_encoded_len((int64_t)(blk_off[i + 1] - blk_off[i])) + 15) & ~(uint64_t)15;
        auto enq = [&](ChunkFeed& feed, const uint8_t* d_in, const uint64_t* rel, uint8_t* d_out) {
            return s2_encode_dev(c, d_in, rel, n, d_out, need, out_off, 0, 0, level, &feed);
        };
        auto region = [&](uint32_t u0) { return c->plan.stage_off[u0]; };
        auto fin = [&](bool* redo) {
            *redo = false;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipGetLastError());
            return KC_OK;
        };
        return host_chunk_fed(c, src, blk_off, n, dst, dst_cap, out_off, need, enq, region, fin);
    }
    if (total >= 2 * host_sub_bytes(c, total) && !c->cfg.host_serial) {
        auto enc = [&](const uint8_t* d_in, const uint64_t* rel, uint32_t nu, uint8_t* d_out, uint64_t cap, uint64_t* oo) {
            return kc_s2_encode_blocks_lvl_dev(c, level, d_in, rel, nu, d_out, cap, oo);
        };
        auto mx = [&](uint64_t len) { return (uint64_t)kc_s2_max_encoded_len((int64_t)len); };
        return host_pipeline(c, src, blk_off, n, dst, dst_cap, out_off, host_sub_bytes(c, total), enc, mx);
    }
    uint64_t need = 0;
    for (uint32_t i = 0; i < n; i++) need += ((uint64_t)kc_s2_max_encoded_len((int64_t)(blk_off[i + 1] - blk_off[i])) + 15) & ~(uint64_t)15;
    kc_status s;
    if ((s = ensure(c, c->tmp_src, total + 64)) || (s = ensure(c, c->tmp_dst, need + 64))) return s;
    HIPCHK(c, hipMemcpyAsync(c->tmp_src.p, src + blk_off[0], total, hipMemcpyHostToDevice, c->stream));
    std::vector<uint64_t> rel(n + 1);
    for (uint32_t i = 0; i <= n; i++) rel[i] = blk_off[i] - blk_off[0];
    s = kc_s2_encode_blocks_lvl_dev(c, level, (const uint8_t*)c->tmp_src.p, rel.data(), n, (uint8_t*)c->tmp_dst.p, need, out_off);
    if (s != KC_OK) return s;
    const uint64_t outn = out_off[n];
    if (outn > dst_cap) { c->err = "dst_cap too small"; return KC_ERR_DST_TOO_SMALL; }
    HIPCHK(c, hipMemcpyAsync(dst, c->tmp_dst.p, outn, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KC_OK;
}

// ---------------------------------------------------------------------------------------
// kc_s2_encode_block: the s2.WriterCustomEncoder hook (s2/writer.go:1053-1064).
// "The function should expect to be called concurrently" — s2.Writer calls it from one goroutine per block
// (writer.go:455-460).  Concurrent callers on ONE context are micro-batched (group commit): a caller appends its block
// to the open slot and copies its bytes into the slot's pinned input; the first caller of a slot is its leader, which
// takes the device lock (while the previous slot still runs, later callers keep joining this one), closes the slot and
// runs ONE H2D -> kernel -> D2H for all blocks of the slot; every caller then copies its own block out.  An idle
// context adds no waiting: a lone caller's slot closes at once.
// ---------------------------------------------------------------------------------------
}  // extern "C"

namespace {

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
    std::atomic<uint64_t> n_calls{0}, n_batches{0};

    bool init(const KcCfg& g, int device) {
        wait_us = (int)g.hook_wait_us;
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

}  // namespace

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
    h->n_calls++;
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
    if (calls) *calls = h ? h->n_calls.load() : 0;
    if (batches) *batches = h ? h->n_batches.load() : 0;
}

}  // extern "C"
