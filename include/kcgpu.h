/* kcgpu.h — C ABI of the MI355X-native block-parallel zstd / S2 encode engine.
 *
 * This is the drop-in boundary for the encode hot path of klauspost/compress: every entry
 * point below is what a cgo binding in the reference would call instead of its pure-Go
 * encoder.  Plain pointers and sizes only; no exceptions cross this boundary; every call
 * returns a kc_status (0 == KC_OK, negative == error) and never aborts the host process
 * (the reference turns encoder panics into errors at its goroutine boundaries,
 * zstd/encoder.go:397-403,421-427).
 *
 * Reference seams replaced (file:line under the reference tree):
 *   kc_zstd_encode_units[_dev]  == N x (*Encoder).EncodeAll(unit, nil)   zstd/encoder.go:722-839
 *   kc_zstd_max_encoded_size    == (*Encoder).MaxEncodedSize             zstd/encoder.go:843-873
 *   kc_s2_encode_blocks[_dev]   == N x s2.Encode(nil, block)             s2/encode.go:29-56
 *   kc_zstd_encode_streams[_dev]      == N x NewWriter(w); Write(unit); Close()                zstd/encoder.go:154-428, 589-649
 *   kc_zstd_encode_streams_cuts[_dev] == the same with Flush() at given byte counts             zstd/encoder.go:547-570
 *   kc_s2_encode_blocks_lvl[_dev]     == N x s2.EncodeBetter / EncodeSnappy / EncodeSnappyBetter(nil, block)   s2/encode.go:117-276
 *   kc_s2_encode_block          == the s2.WriterCustomEncoder callback   s2/writer.go:1053-1064
 *   kc_s2_max_encoded_len       == s2.MaxEncodedLen                      s2/encode.go:389-418
 *   kc_s2_encode_stream_dev     == s2.Writer.EncodeBuffer framing        s2/writer.go:357-451
 *   kc_zstd_decode_units_dev    == N x Decoder.DecodeAll (verifier)       zstd/framedec.go, blockdec.go, seqdec_generic.go
 *   kc_s2_decode_blocks_dev     == N x s2.Decode (verifier)              s2/decode.go:58, decode_other.go:22
 *   kc_xxh64_units_dev          == xxhash.Digest over each unit          zstd/internal/xxhash/xxhash.go:27-230
 */
#ifndef KCGPU_H
#define KCGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kc_ctx kc_ctx; /* opaque: device buffers, stream, scratch */

typedef enum {
    KC_OK = 0,
    KC_ERR_BAD_ARG = -1,       /* null pointer, unsorted offsets, zero units ... */
    KC_ERR_DST_TOO_SMALL = -2, /* dst_cap < sum of MaxEncodedSize(unit) */
    KC_ERR_HIP = -3,           /* a HIP runtime call failed; kc_last_error() has the text */
    KC_ERR_UNSUPPORTED = -4,   /* options outside what the device path implements; caller falls back to the CPU encoder */
    KC_ERR_NO_DEVICE = -5,     /* no gfx950 device visible */
    KC_ERR_INTERNAL = -6       /* device-side invariant violated (reported, never silently ignored) */
} kc_status;

/* Which build of the reference the S2 default / Snappy-compatible levels are byte-identical to.  The reference has two block
 * encoders for them: portable Go (s2/encode_all.go: arm64, noasm and every non-amd64 build) and generated amd64 assembly
 * (s2/encode_amd64.go + encodeblock_amd64.s: other table sizes, hash lengths and skip rates per input size class, matches extended
 * to the very end of the block).  Both write valid, different streams.  KC_OPT_S2_VARIANT picks the one to match. */
#define KC_S2_VARIANT_GO 0
#define KC_S2_VARIANT_AMD64 1

/* zstd.EncoderLevel (zstd/encoder_options.go:163-179) */
typedef enum { KC_SPEED_FASTEST = 1, KC_SPEED_DEFAULT = 2, KC_SPEED_BETTER = 3, KC_SPEED_BEST = 4 } kc_level;

/* Resolved zstd encoder options == the fields of encoderOptions (zstd/encoder_options.go:18-34)
 * that change output bytes.  Use kc_zstd_opts_default() + kc_zstd_opts_* helpers, which apply the
 * reference's EOption functions with the same order-dependent side effects. */
typedef struct {
    int32_t level;           /* o.level */
    int32_t window_size;     /* o.windowSize */
    int32_t block_size;      /* o.blockSize */
    int32_t crc;             /* o.crc            (WithEncoderCRC) */
    int32_t single;          /* o.single: -1 nil, 0 false, 1 true (WithSingleSegment) */
    int32_t full_zero;       /* o.fullZero       (WithZeroFrames) */
    int32_t no_entropy;      /* o.noEntropy      (WithNoEntropyCompression) */
    int32_t all_lit_entropy; /* o.allLitEntropy  (WithAllLitEntropyCompression) */
    int32_t low_mem;         /* o.lowMem         (WithLowerEncoderMem; no effect on bytes) */
    int32_t custom_window, custom_block, custom_alent; /* o.customWindow / customBlockSize / customALEntropy */
    uint32_t dict_id;        /* WithEncoderDictRaw id (0 = no dictionary) */
    const uint8_t* dict;     /* dictionary content (host pointer), used as initial history */
    uint64_t dict_len;
    /* full-format dictionaries only (kc_zstd_opts_dict); raw dictionaries keep {1,4,8} and no literal table */
    uint32_t dict_offsets[3];     /* dict.offsets -> blk.recentOffsets (zstd/enc_base.go:189-195) */
    int32_t dict_huf_len;         /* len(dict.litEnc.prevTable); 0 = no literal table */
    int32_t dict_huf_log;         /* dict.litEnc.prevTableLog */
    uint16_t dict_huf_val[256];   /* cTableEntry.val  (huff0/decompress.go:142-165) */
    uint8_t dict_huf_nbits[256];  /* cTableEntry.nBits */
    int32_t concurrent;           /* o.concurrent (WithEncoderConcurrency); 0 = the default (> 1).  Only "1 or not" changes bytes, and
                                   * only of dictionary streams: with 1 the reference takes nextBlock's synchronous form
                                   * (zstd/encoder.go:364-391), whose blk.reset(nil) drops the dictionary's literal table before the
                                   * first block; otherwise the first block starts from it (kc_zstd_encode_streams*) */
} kc_zstd_opts;

/* encoderOptions.setDefault, zstd/encoder_options.go:36-48 */
void kc_zstd_opts_default(kc_zstd_opts* o);
/* WithEncoderLevel, zstd/encoder_options.go:236-266 */
int kc_zstd_opts_level(kc_zstd_opts* o, int level);
/* WithWindowSize, zstd/encoder_options.go:110-133 */
int kc_zstd_opts_window(kc_zstd_opts* o, int n);
/* WithEncoderCRC / WithZeroFrames / WithNoEntropyCompression / WithAllLitEntropyCompression / WithSingleSegment */
int kc_zstd_opts_crc(kc_zstd_opts* o, int b);
int kc_zstd_opts_zero_frames(kc_zstd_opts* o, int b);
int kc_zstd_opts_no_entropy(kc_zstd_opts* o, int b);
int kc_zstd_opts_all_lit_entropy(kc_zstd_opts* o, int b);
int kc_zstd_opts_single_segment(kc_zstd_opts* o, int b);
/* WithEncoderConcurrency, zstd/encoder_options.go:76-87 (n < 1: error) */
int kc_zstd_opts_concurrency(kc_zstd_opts* o, int n);
/* WithEncoderDictRaw, zstd/encoder_options.go:398-406 */
int kc_zstd_opts_dict_raw(kc_zstd_opts* o, uint32_t id, const uint8_t* content, uint64_t len);
/* WithEncoderDict, zstd/encoder_options.go:382-391: a dictionary in the "zstd --train" format.  Parses it like loadDict
 * (zstd/dict.go:71-150): ID, literal Huffman table (becomes huff0 prevTable of each unit's first block,
 * zstd/blockenc.go:518-522), repeat offsets, content.  `blob` must outlive every encode call using `o`
 * (o->dict points into it).  Returns 0, or -1 when the reference's loadDict would return an error. */
int kc_zstd_opts_dict(kc_zstd_opts* o, const uint8_t* blob, uint64_t len);

/* (*Encoder).MaxEncodedSize, zstd/encoder.go:843-873 (padding off) */
int64_t kc_zstd_max_encoded_size(const kc_zstd_opts* o, int64_t size);

/* ---- context ---- */
/* device: HIP device ordinal.  stream: a hipStream_t to launch on (NULL = the context's own stream).
 * The context owns all device scratch: one call at a time per context (kc_s2_encode_block excepted, see there); any number of
 * contexts may work on one device at the same time from different host threads — how the drop-ins make EncodeAll safe for concurrent
 * callers on ONE encoder like the reference's (zstd/encoder.go:90-99, 722-729: a channel of encoder states; here a pool of contexts:
 * shim/go/zstdgpu, compress_amd/zstd.py; tests/test_zz_gpu_threads.py runs 8 threads on one encoder on the device). */
kc_status kc_ctx_create(kc_ctx** out, int device, void* stream);
void kc_ctx_destroy(kc_ctx* ctx);
/* Why the last kc_ctx_create of the calling thread failed ("" after a success).  The first kc_ctx_create on a device runs a
 * known-answer launch (256 bytes through the XXH64 kernel) on a thread of its own and waits for it with a 90 s deadline: a device
 * that faults, hangs or miscomputes makes kc_ctx_create return KC_ERR_HIP / KC_ERR_INTERNAL with the reason here instead of hanging
 * or aborting the host (the promise at the top of this file); the verdict is remembered per device for the life of the process. */
const char* kc_create_error(void);
/* Give device memory back without giving the handles up (long-lived hosts, several processes on one device).  kc_ctx_trim frees
 * the context's device scratch (it grows back with the next call; KC_ERR_BAD_ARG while a begin / submit is in flight on it).
 * kc_device_trim frees what the device's rolling host pipeline holds (kc_zstd_encode_units / kc_s2_encode_blocks_lvl on large inputs:
 * ten device slots of a sub-batch each and the scratch of its eight encoder lanes, ~100 GiB after 1 GiB SpeedFastest sub-batches);
 * KC_ERR_BAD_ARG while host-buffer calls are in flight on that device, KC_OK when the pipeline never started. */
kc_status kc_ctx_trim(kc_ctx* ctx);
kc_status kc_device_trim(int device);
/* Page-locked host memory for callers that own their buffers (a Go caller: C memory wrapped in a slice with unsafe.Slice).  The
 * host-buffer entry points recognise page-locked source / destination buffers — these, hipHostMalloc'ed or hipHostRegister'ed ones —
 * and DMA straight from / into them: the rolling pipeline's staging copies (16 host threads at ~100 GB/s while a call runs) are not
 * made at all.  Pageable buffers work as before.  kc_host_alloc: KC_ERR_UNSUPPORTED when the host cannot lock that much memory. */
kc_status kc_host_alloc(void** out, uint64_t bytes);
void kc_host_free(void* p);
const char* kc_last_error(const kc_ctx* ctx);
/* ONE stream with WithConcurrentBlocks(true) (zstd/encoder_options.go:340-353; zstd/enc_jobs.go): the bytes equal
 *   enc, _ := zstd.NewWriter(w, opts..., zstd.WithConcurrentBlocks(true))   (with WithEncoderConcurrency > 1)
 *   enc.Write(src[...]) with enc.Flush() after cuts[0], cuts[1], ... bytes (ascending; a Write followed by ReadFrom is such a point, encoder.go:500-507); enc.Close()
 * The reference cuts the stream into jobs of kc_zstd_job_size() input bytes (a Flush ends a job early), encodes each on a freshly
 * reset encoder whose history is the last kc_zstd_overlap_size() bytes of the previous job's input (ResetPrefix,
 * zstd/enc_fast.go:800-811, enc_dfast.go:1040-1050, enc_better.go:1099-1112) and concatenates the outputs behind one frame header:
 * the jobs are independent units, which is what this engine wants: this is the reference's own way of turning ONE stream into
 * device work.  src / dst are HOST buffers; dst_cap >= sum over jobs of kc_zstd_max_encoded_size(job + overlap) + 16.  A stream of
 * at most one block is the EncodeAll frame (enc_jobs.go:263-279).  With a dictionary the reference switches the option off
 * (zstd/encoder.go:81,174): KC_ERR_UNSUPPORTED.  The frame is assembled in dst batch by batch: on any error *out_len is 0 and the
 * bytes already written to dst (the frame header, earlier batches) are unspecified — dst is clobbered, not rolled back. */
kc_status kc_zstd_encode_jobs(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* src, uint64_t len, const uint64_t* cuts, uint64_t n_cuts,
                              uint8_t* dst, uint64_t dst_cap, uint64_t* out_len);
int64_t kc_zstd_job_size(const kc_zstd_opts* o);     /* encoderOptions.jobSize, zstd/encoder_options.go:356-359 */
int64_t kc_zstd_overlap_size(const kc_zstd_opts* o); /* encoderOptions.overlapSize, zstd/encoder_options.go:362-371 */

/* ---- context options ----
 * Every tunable of the library is a field of the context with a built-in default; kc_ctx_set_option is the only way to change one.
 * The library reads NO environment variable (round 5).  The name beside each key is the variable the measurement harness above the
 * C ABI maps to it (compress_amd/_lib.py: Context applies KC_* variables as options after kc_ctx_create) — a convenience of the
 * Python tooling, not of the shipped library.
 *
 * Two kernel families serve SpeedFastest and s2.Encode / s2.EncodeSnappy (KC_OPT_MATCH_PATH):
 *   KC_PATH_HBM  per-unit hash tables in an HBM arena, 8 units per wave: the throughput path, needs ~10^4 units in flight;
 *   KC_PATH_LDS  the unit's hash table (and, for S2 blocks up to 64 KiB, the block) in the CU's LDS, one wave per unit: the
 *                latency path — one unit in a few ms instead of ~30 ms; also the faster one for sparse scans (high-entropy input);
 *   KC_PATH_AUTO by units in flight, against the measured crossovers (profiles/r03_crossover_*.csv).
 * Both produce the same bytes (the GPU parity tests run on both). */
enum { KC_PATH_AUTO = 0, KC_PATH_HBM = 1, KC_PATH_LDS = 2 };
typedef enum {
    KC_OPT_MATCH_PATH = 1,           /* KC_MATCH_PATH             KC_PATH_* */
    KC_OPT_ZFAST_LDS_MAX_UNITS = 2,  /* KC_ZFAST_LDS_MAX_UNITS    auto: SpeedFastest batches of at most this many units take KC_PATH_LDS */
    KC_OPT_S2_LDS_MAX_BLOCKS = 3,    /* KC_S2_LDS_MAX_BLOCKS      auto: s2 block batches of at most this many blocks take KC_PATH_LDS */
    KC_OPT_SPEC_W0 = 4,              /* KC_SPEC_W0                HBM kernels: probe steps per round after a match (-1: per-level default) */
    KC_OPT_SPEC_GROW = 5,            /* KC_SPEC_GROW              HBM kernels: width after a miss: 0 keep, 1 +1, 2 double (-1: default) */
    KC_OPT_LDS_SPEC_W0 = 6,          /* KC_LDS_SPEC_W0            SpeedFastest LDS kernel: probe steps per round after a match (doubles on a miss, max 64; default 16); 0 = units up to 128 KiB without history through the source-ring instantiation at 16 */
    KC_OPT_S2_LDS_SPEC_W0 = 17,      /* KC_S2_LDS_SPEC_W0         S2 LDS kernel: the same; blocks up to 64 KiB: 0 (default) the fused wave-uniform step, 1 its first form */
    KC_OPT_HOST_SERIAL = 7,          /* KC_HOST_SERIAL            host-buffer entry points: no pipelining (copy, encode, copy) */
    KC_OPT_HOST_PIPE_MIB = 8,        /* KC_HOST_PIPE_MIB          host-buffer entry points: sub-batch size of the three-stage pipeline */
    KC_OPT_HOST_OVERLAP_MIN_MIB = 9, /* KC_HOST_OVERLAP_MIN_MIB   smallest input that takes the chunk-fed path (-1: default) */
    KC_OPT_HOST_COPY_THREADS = 10,   /* KC_HOST_COPY_THREADS      threads of the pageable <-> pinned copies (0: the cgroup's CPUs, max 16) */
    KC_OPT_HOST_TRACE = 11,          /* KC_HOST_TRACE             timeline of the chunk-fed path on stderr */
    KC_OPT_HOST_CHUNK_MIB = 12,      /* KC_HOST_CHUNKS_MIB (list) chunk size of the chunk-fed path (0: a quarter of the batch) */
    KC_OPT_K2_PROF = 13,             /* KC_K2_PROF                per-phase shader clocks of the entropy kernel on stderr */
    KC_OPT_S2_HOOK_WAIT_US = 14,     /* KC_S2_HOOK_WAIT_US        kc_s2_encode_block: time a batch leader waits for more callers */
    KC_OPT_S2_HOOK_BATCH = 15,       /* KC_S2_HOOK_BATCH          kc_s2_encode_block: most blocks per device batch */
    KC_OPT_TEST_FEED_REDO = 16,      /* (no variable)             diagnostics: force the chunk-fed path's re-encode fallback */
    KC_OPT_MAX_SCRATCH_MIB = 18,     /* (no variable)             ceiling of the device scratch one batch may take (default 160 GiB, and 85 % of the free memory): larger calls are cut into several batches */
    KC_OPT_BEST_SLOTS = 19,          /* (no variable)             SpeedBestCompression: table slots of 34 MiB = units encoded at a time (default 6144 = 204 GiB; allocated on demand for the units a batch has, never more than 80 % of the free device memory) */
    KC_OPT_S2_VARIANT = 20,          /* (no variable)             s2.Encode / s2.EncodeSnappy: KC_S2_VARIANT_GO (default) or KC_S2_VARIANT_AMD64 */
    KC_OPT_BETTER_DICT_EPOCH = 21,   /* (no variable)             SpeedBetterCompression with a dictionary: 1 = epoch-stamped tables + shared dictionary table (measurements; default 0: per-batch copy) */
    KC_OPT_ZFAST_EPOCH = 22,         /* KC_ZFAST_EPOCH            SpeedFastest HBM-table kernel, no dictionary: 1 (default) = epoch-stamped table entries, the arena is cleared every 15 batches instead of every batch */
    KC_OPT_ZFAST_XSEG_K = 23,        /* KC_ZFAST_XSEG_K           SpeedFastest HBM-table kernel: a probe round crosses skip-segment boundaries once (s - nextEmit) >> 5 has reached this value (default 0: always; large: never); acts in the kernel form KC_OPT_ZFAST_VARIANT selects */
    KC_OPT_FUSE_RAW_XXH = 24,        /* KC_FUSE_RAW_XXH           frames made of raw blocks only: 1 (default) = XXH64 and the payload copy in one pass over the source */
    KC_OPT_ZFAST_FILTER = 25,        /* KC_ZFAST_FILTER           SpeedFastest HBM-table kernel: 1 (default) = units that have emitted no sequence yet skip the table loads of bucket groups they have not written (a 512-bit map in their idle sequence buffer) */
    KC_OPT_XXH_FIN_MODE = 26,        /* KC_XXH_FIN_MODE           checksum-and-copy kernel, payload of raw-only frames: 0 stored from the registers, 1 the same software-pipelined, 2 through an LDS ring as aligned stores, 3 like 1 with eight 16-byte loads per lane in flight instead of four */
    KC_OPT_ZFAST_VARIANT = 27,       /* KC_ZFAST_VARIANT          SpeedFastest HBM-table kernel: 0 the plain form, 1 the form for input without matches (KC_OPT_ZFAST_XSEG_K, KC_OPT_ZFAST_FILTER), -1 (default) per batch: form 1 when the context's previous batch did not compress */
    KC_OPT_ZFAST_PRESCAN = 28,       /* KC_ZFAST_PRESCAN          SpeedFastest, EncodeAll batches without dictionary: 1 = a pre-scan proves units free of matches from their probe positions alone and writes their (raw-block) frames, the match finder and the entropy stage skip them; 0 off; -1 (default) per batch: on when the context's previous batch did not compress */
    KC_OPT_S2_HOOK_LANES = 29,       /* KC_S2_HOOK_LANES          kc_s2_encode_block: batches of concurrent callers on the device at once (own stream and scratch each; default 4, at most 8).  Footprint: on the first hook call the context creates `lanes` contexts and (lanes + 2) slots of 2 x 8 MiB pinned host memory (96 MiB at the default), each lane's device scratch grows to its largest batch (a few MiB per 256 blocks of 64 KiB); a lane takes the caller's variant / kernel-family options and scratch ceiling per batch */
    KC_OPT_JOB_PRIME = 30,           /* KC_JOB_PRIME              kc_zstd_encode_jobs: where a job's tables are primed from its overlap prefix (ResetPrefix): 1 (default) on the device (kc_zstd_prime.hip), 0 on the host, uploaded per batch */
    KC_OPT_HOST_CHUNK_MIB_APPEND = 32, /* KC_HOST_CHUNKS_MIB (list) one more chunk size behind KC_OPT_HOST_CHUNK_MIB's: an uneven chunk schedule (the last size repeats) */
    KC_OPT_STAGE2_STREAM = 31,       /* (no variable)             a hipStream_t handle (0: none): kc_zstd_encode_units_dev[_begin/_end] run the entropy stage and everything behind it on this stream, behind an event of the match finder's — for callers that give the two stages different CU masks (hipExtStreamCreateWithCUMask).  Scope: the device-resident zstd entry points only (the host-buffer entry points and S2 ignore it).  Lifetime: the handle must belong to the context's device and outlive the option — reset it to 0 before destroying the stream; the library never destroys it */
    KC_OPT_HOST_ROLL = 33,           /* KC_HOST_ROLL              host-buffer entry points, large inputs: 1 (default) = the device's rolling pipeline (sub-batches of all calls in flight staged, encoded on four lanes and drained in arrival order: consecutive calls overlap), 0 = one chunk-fed device batch per call (round 5) */
    KC_OPT_HOST_ROLL_MIB = 34,       /* KC_HOST_ROLL_MIB          rolling pipeline: sub-batch size (0: a quarter of the call's input — SpeedBetterCompression and S2: half —, 64 MiB .. 1 GiB) */
    KC_OPT_S2_HOOK_HOST_FIRST = 35,  /* KC_S2_HOOK_HOST_FIRST     kc_s2_encode_block: how many callers at a time are left to the host's built-in encoder (they get -1) before the overflow goes to the device: -1 (default) the host's hardware threads, 0 every caller to the device (see kc_s2_encode_block) */
    KC_OPT_LAST_PRESCAN_UNITS = 102, /* read-only: units of the last batch the pre-scan settled */
    KC_OPT_LAST_PATH = 100,          /* read-only: KC_PATH_HBM / KC_PATH_LDS the last batch ran on */
    KC_OPT_LAST_BATCHES = 101        /* read-only: device batches the last kc_zstd_encode_units_dev / kc_s2_encode_*_dev call was cut into */
} kc_option;
kc_status kc_ctx_set_option(kc_ctx* ctx, int key, int64_t value);
int64_t kc_ctx_get_option(const kc_ctx* ctx, int key);
/* Device properties as probed (CU count, LDS bytes per CU, clock, name) */
kc_status kc_device_info(const kc_ctx* ctx, int32_t* n_cu, int32_t* lds_per_cu, int32_t* clock_khz, char* name, size_t name_cap);

/* ---- zstd: N independent units, each == EncodeAll(unit, nil) ----
 * src:       all units, concatenated (host memory for the plain call, device memory for _dev)
 * unit_off:  n_units+1 ascending byte offsets into src (HOST memory in both variants)
 * dst:       output buffer; unit i's frame is written at dst + out_off[i]
 * dst_cap:   must be >= sum_i kc_zstd_max_encoded_size(unit_i)
 * out_off:   n_units+1 offsets of the compacted frames (HOST memory in both variants)
 * The _dev variant leaves the frames in device memory (dst is a device pointer) and only
 * copies the n_units+1 offsets back; it synchronises the stream before returning.
 * A unit may have any number of blocks up to 1 GiB (beyond: KC_ERR_UNSUPPORTED); its blocks are parsed one after the other by
 * one lane group, so the device pays with thousands of units in flight, not with a few long ones. */
kc_status kc_zstd_encode_units(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off,
                               uint32_t n_units, uint8_t* dst, uint64_t dst_cap, uint64_t* out_off);
kc_status kc_zstd_encode_units_dev(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off,
                                   uint32_t n_units, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off);
/* N independent STREAMS: unit i's output equals NewWriter(w).Write(unit_i) ... Close() (zstd/encoder.go:154-428, 567-649;
 * no Flush in between — see kc_zstd_encode_streams_cuts): below one block that is the EncodeAll frame; from one block on, a frame without content size whose
 * blocks all see the history, with the `last` flag on a short final block or else a trailing empty raw block.  Same limits as
 * kc_zstd_encode_units_dev (a stream of more than 1 GiB is KC_ERR_UNSUPPORTED); dictionaries are KC_ERR_UNSUPPORTED. */
kc_status kc_zstd_encode_streams_dev(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off,
                                     uint32_t n_units, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off);
/* host-buffer form (src / dst in host memory), like kc_zstd_encode_units */
kc_status kc_zstd_encode_streams(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off,
                                 uint32_t n_units, uint8_t* dst, uint64_t dst_cap, uint64_t* out_off);
/* The same with Flush points (zstd/encoder.go:547-570: Flush ends the block being filled; a Flush that finds nothing buffered does
 * nothing): cuts[cut_off[i] .. cut_off[i+1]) lists, ascending, how many bytes of stream i had been written at each Flush.  A Write
 * that precedes ReadFrom is such a point too (ReadFrom first ends the block being filled, :482-486).  The output is the
 * concatenation of what the reference's writer received, delivered when the call returns.
 * dst_cap: >= sum_i (kc_zstd_max_encoded_size(unit_i) + 3 * ncuts_i + 3, rounded up to 16). */
kc_status kc_zstd_encode_streams_cuts_dev(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off,
                                          uint32_t n_units, const uint64_t* cut_off, const uint64_t* cuts, uint8_t* d_dst,
                                          uint64_t dst_cap, uint64_t* out_off);
kc_status kc_zstd_encode_streams_cuts(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off,
                                      uint32_t n_units, const uint64_t* cut_off, const uint64_t* cuts, uint8_t* dst,
                                      uint64_t dst_cap, uint64_t* out_off);
/* Host logic only: the block plan of one stream with Flush points as kc_zstd_encode_streams_cuts lays it out — block starts
 * (starts may be NULL), *flags bit 0 = stream frame (a block was written before Close), bit 1 = Close found nothing buffered
 * (empty last block).  Returns the number of blocks, -1 bad argument, -2 starts_cap too small. */
int64_t kc_zstd_plan_stream_blocks(int32_t block_size, uint64_t len, const uint64_t* cuts, uint64_t n_cuts, uint32_t* starts,
                                   uint64_t starts_cap, uint32_t* flags);
/* Asynchronous form of the host-buffer entry points (SURVEY §8(b) "async submit/wait"): submit returns at once and the call runs
 * on a thread of its own; kc_wait blocks until it is done and returns ITS status (kc_last_error for the text).  One job per
 * context.  Inside one call the source staging, the kernels and the drain of the frames already overlap chunk by chunk; with
 * two contexts the caller also overlaps consecutive batches (submit(A, batch k+1); wait(B) ...).  src, unit_off / blk_off, dst,
 * out_off and a dictionary referenced by *o must stay valid until kc_wait returns; *o itself is copied. */
kc_status kc_zstd_encode_units_submit(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* src, const uint64_t* unit_off,
                                      uint32_t n_units, uint8_t* dst, uint64_t dst_cap, uint64_t* out_off);
kc_status kc_s2_encode_blocks_lvl_submit(kc_ctx* ctx, int level, const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks,
                                         uint8_t* dst, uint64_t dst_cap, uint64_t* out_off);
kc_status kc_wait(kc_ctx* ctx);
/* Split form of kc_zstd_encode_units_dev for ONE device batch (<= 8 GiB): _begin enqueues everything up to and including
 * the match finder and returns without waiting; _end enqueues the entropy stage, waits, and returns the offsets.  With two
 * contexts (two streams, two sets of scratch) a caller pipelines consecutive batches:
 *     begin(A, batch0); begin(B, batch1); end(A); begin(A, batch2); end(B); ...
 * so that the match finder of batch i+1 runs under the entropy stage of batch i.  kc_ctx_chain_after(B, A) makes B's match
 * finder wait for A's (and vice versa), which keeps one match finder on the device at a time: right for SpeedFastest and
 * SpeedDefault, whose kernels are one residency of the chip per 4 GiB batch.  A SpeedBetterCompression batch of 1 GiB fills
 * 8 of 12 wave slots per CU and every unit is a chain of dependent table trips: with three contexts chained two apart
 * (chain_after(C, A), chain_after(A, B), chain_after(B, C)) — or not chained at all — consecutive batches' match finders share the
 * chip (measured +16-20 %: DESIGN.md 4.2).  d_src, d_dst and the options' dictionary must stay valid until _end returns; unit_off
 * is copied. */
kc_status kc_zstd_encode_units_dev_begin(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off,
                                         uint32_t n_units, uint8_t* d_dst, uint64_t dst_cap);
kc_status kc_zstd_encode_units_dev_end(kc_ctx* ctx, uint64_t* out_off);
/* _end with the place of the frames named only now (d_dst of _begin is then just the capacity check): the frames of a batch leave
 * their staging slots in _end, so a caller that runs ONE EncodeAll batch as several smaller launches — two halves of a 4 GiB batch
 * on three contexts, chained two apart, are 4-8 % faster than one launch (the second half's units fill the first half's tail:
 * DESIGN.md 4.1) — puts each part's frames right behind the previous part's and gets the same contiguous output, out_off relative
 * to the d_dst given here.  KC_ERR_DST_TOO_SMALL if dst_cap is below the sum of MaxEncodedSize(unit) of this batch. */
kc_status kc_zstd_encode_units_dev_end_at(kc_ctx* ctx, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off);
void kc_ctx_chain_after(kc_ctx* ctx, kc_ctx* prev);

/* XXH64(seed 0) of every unit (the frame checksum primitive), device-resident input, host output. */
kc_status kc_xxh64_units_dev(kc_ctx* ctx, const uint8_t* d_src, const uint64_t* unit_off, uint32_t n_units, uint64_t* out_hash);

/* Debug/inspection (parity of intermediates): run only the match finder on device-resident units and
 * return, for every block, the sequence list as (litLen, matchLen, offset) u32 triples.
 * blk_first_seq has n_blocks+1 entries; blk_flags (may be NULL) receives KC_BF_* verdict bits in bits 0..7 and
 * the number of speculative probe rounds in bits 8..31.  Buffers are HOST memory. */
kc_status kc_zstd_debug_parse_dev(kc_ctx* ctx, const kc_zstd_opts* o, const uint8_t* d_src, const uint64_t* unit_off,
                                  uint32_t n_units, uint32_t* seqs, uint64_t seq_cap, uint64_t* blk_first_seq,
                                  uint32_t* blk_extra_lits, uint32_t* blk_flags, uint32_t blk_cap, uint32_t* n_blocks_out);

/* ---- S2: N independent blocks, each == s2.Encode(nil, block) (varint length + body) ---- */
int64_t kc_s2_max_encoded_len(int64_t src_len);
kc_status kc_s2_encode_blocks(kc_ctx* ctx, const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks,
                              uint8_t* dst, uint64_t dst_cap, uint64_t* out_off);
kc_status kc_s2_encode_blocks_dev(kc_ctx* ctx, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n_blocks,
                                  uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off);
/* The same at a chosen level: KC_S2_LEVEL_DEFAULT == s2.Encode (encodeBlockGo / encodeBlockGo64K, s2/encode_all.go:72-500),
 * KC_S2_LEVEL_BETTER == s2.EncodeBetter (s2/encode.go:117-144: encodeBlockBetterGo / encodeBlockBetterGo64K,
 * s2/encode_better.go:50-307 / 485-730), KC_S2_LEVEL_SNAPPY == s2.EncodeSnappy (s2/encode.go:204-232: encodeBlockSnappyGo /
 * encodeBlockSnappyGo64K, s2/encode_all.go:502-690 / 692-880 — blocks a Snappy decoder reads: no repeat tags),
 * KC_S2_LEVEL_SNAPPY_BETTER == s2.EncodeSnappyBetter, KC_S2_LEVEL_BEST == s2.EncodeBest (s2/encode.go:146-202: encodeBlockBest,
 * s2/encode_best.go:22-455), KC_S2_LEVEL_SNAPPY_BEST == s2.EncodeSnappyBest (s2/encode.go:278-305: encodeBlockBestSnappy,
 * s2/encode_best.go:457-710).  The best levels take 4.5 MiB of tables per block: large calls are cut into batches. */
#define KC_S2_LEVEL_DEFAULT 0
#define KC_S2_LEVEL_BETTER 1
#define KC_S2_LEVEL_SNAPPY 2
#define KC_S2_LEVEL_BEST 4
#define KC_S2_LEVEL_SNAPPY_BEST 5
#define KC_S2_LEVEL_UNCOMPRESSED 6 /* kc_s2_encode_stream_lvl_dev only: s2.WriterUncompressed (s2/writer.go:951) — every block an uncompressed chunk */
#define KC_S2_LEVEL_SNAPPY_BETTER 3   /* s2.EncodeSnappyBetter (s2/encode.go:248-276: encodeBlockBetterSnappyGo / ...64K, s2/encode_better.go:310-483 / 733-900) */
kc_status kc_s2_encode_blocks_lvl(kc_ctx* ctx, int level, const uint8_t* src, const uint64_t* blk_off, uint32_t n_blocks,
                                  uint8_t* dst, uint64_t dst_cap, uint64_t* out_off);
kc_status kc_s2_encode_blocks_lvl_dev(kc_ctx* ctx, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n_blocks,
                                      uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off);
/* Split form of kc_s2_encode_blocks_lvl_dev for ONE device batch of bare blocks below the best levels (the S2 sibling of
 * kc_zstd_encode_units_dev_begin / _end_at): _begin enqueues everything up to and including the encoder kernel and returns without
 * waiting; _end_at compacts the blocks to the d_dst named there, copies the offsets (relative to it) and waits.  A 2 GiB batch of
 * 64 KiB blocks is exactly one residency of the chip; run as two launches on three contexts (three host threads, or begin / begin /
 * end_at / ...) the next launch's blocks take the wave slots as the previous launch's leave them: measured +5-6 %.
 * d_src must stay valid until _end_at returns; blk_off is copied.  KC_ERR_UNSUPPORTED: framed, best levels, more than one batch. */
kc_status kc_s2_encode_blocks_lvl_dev_begin(kc_ctx* ctx, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n_blocks);
kc_status kc_s2_encode_blocks_lvl_dev_end_at(kc_ctx* ctx, uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off);
/* kc_s2_encode_stream_dev at a chosen level (s2.WriterBetterCompression, s2/writer.go:931) */
kc_status kc_s2_encode_stream_lvl_dev(kc_ctx* ctx, int level, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n_blocks,
                                      uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off, int with_stream_id);
/* s2.Writer framing on the device (s2/writer.go:394-451): every block becomes one chunk
 * `type(1) | len24 | masked CRC32C(4) | body` (compressed 0x00: uvarint(len) + block; incompressible 0x01: raw bytes),
 * optionally preceded by the stream identifier `ff 06 00 00 "S2sTwO"`.  The result is a complete, concatenable .s2
 * stream; out_off[i] is the start of chunk i (out_off[0] == 10 with the identifier).  dst_cap >= sum(MaxEncodedLen+8)+10.
 * A block of a framed stream is at most 4 MiB (s2.maxBlockSize: the chunk header holds a 24-bit length and the reference's
 * Reader refuses larger chunks): a larger block is KC_ERR_BAD_ARG (bare blocks: kc_s2_encode_blocks*, up to 1 GiB). */
kc_status kc_s2_encode_stream_dev(kc_ctx* ctx, const uint8_t* d_src, const uint64_t* blk_off, uint32_t n_blocks,
                                  uint8_t* d_dst, uint64_t dst_cap, uint64_t* out_off, int with_stream_id);
/* s2.Decode (s2/decode.go:58 -> s2Decode, s2/decode_other.go:22) over N encoded blocks (uvarint length + body), for
 * on-device round-trip verification: block i decodes to d_dst + dst_off[i] and must produce exactly dst_off[i+1]-dst_off[i]
 * bytes.  enc_off / dst_off / status are host arrays; status[i] = 0 or the first error met in block i (corrupt input is
 * reported, never written out of bounds). */
kc_status kc_s2_decode_blocks_dev(kc_ctx* ctx, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n_blocks, uint8_t* d_dst,
                                  const uint64_t* dst_off, uint32_t* status);
/* zstd frame decoder (zstd/framedec.go:65-330, blockdec.go:227-690, seqdec_generic.go) over N units, one frame each, for
 * on-device round-trip verification: unit i decodes to d_dst + dst_off[i] and must produce exactly dst_off[i+1]-dst_off[i]
 * bytes; the frame checksum, if present, is checked against XXH64 of the decoded bytes (status 30 on mismatch).  All block,
 * literal and sequence modes; dictionary frames need kc_zstd_decode_units_dict_dev (status 20 otherwise).  enc_off / dst_off / status are host arrays. */
kc_status kc_zstd_decode_units_dev(kc_ctx* ctx, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n_units, uint8_t* d_dst,
                                   const uint64_t* dst_off, uint32_t* status);
/* the same with a dictionary's CONTENT (host pointer) as history in front of every frame: frames written with
 * WithEncoderDictRaw, or with WithEncoderDict as long as they do not reuse the dictionary's entropy tables */
kc_status kc_zstd_decode_units_dict_dev(kc_ctx* ctx, const uint8_t* d_enc, const uint64_t* enc_off, uint32_t n_units, uint8_t* d_dst,
                                        const uint64_t* dst_off, uint32_t* status, const uint8_t* dict, uint64_t dict_len);
/* Single-block form with the WriterCustomEncoder contract (s2/writer.go:1053-1064): no varint header;
 * returns bytes used, 0 = incompressible (store raw), <0 = fall back to the built-in encoder.
 * "The function should expect to be called concurrently" (writer.go:1058; s2.Writer calls it from one goroutine per block,
 * writer.go:455-460): this entry point — and only this one — is safe to call from any number of host threads on ONE
 * context.  Concurrent callers are micro-batched: the blocks that arrive while the previous batch is on the device share
 * one H2D copy, one kernel launch and one D2H copy (pinned staging; KC_S2_HOOK_WAIT_US adds an optional collection window,
 * KC_S2_HOOK_BATCH caps the blocks per launch, default 256).  A context serving this hook must not be used for other calls
 * at the same time.
 * Host first (round 6): one block through the device takes ~3 ms (copy in, one wave on one CU, copy out) where the reference's own
 * encoder takes ~0.1 ms on a host core, so a caller the host could serve is never faster here (measured: 23 MB/s per lone caller,
 * 669 MB/s at 64 callers against ~600 MB/s per core for the built-in encoder; profiles/r04_hook_bench_final.json).  The hook
 * therefore answers -1 ("use the built-in encoder", writer.go:455-460) while fewer callers than the host has hardware threads are
 * busy encoding on the host — it books every caller it sends back as busy until that thread calls again, or for the time a slow core
 * needs for the block (len / 500 MB/s) — and takes only the overflow: callers that arrive while every hardware thread is booked.
 * Wiring the hook is so never slower than not wiring it (tools/hook_bench.cpp, profiles/r06_hook_bench.json: within 4 % of the
 * built-in encoder at 1 / 4 / 16 / 64 callers; forced to the device 23 / 92 / 265 / 630 MB/s against 1 250 / 4 970 / 19 400 /
 * 60 000).  KC_OPT_S2_HOOK_HOST_FIRST: the number of callers left to the host (default: its hardware threads), 0 = every caller to
 * the device (what the parity tests and a host without spare CPUs want). */
int64_t kc_s2_encode_block(kc_ctx* ctx, uint8_t* dst, uint64_t dst_cap, const uint8_t* src, uint64_t src_len);
/* hook diagnostics: calls served and device batches run on this context so far */
void kc_s2_hook_stats(const kc_ctx* ctx, uint64_t* calls, uint64_t* batches);
/* calls the hook answered -1 by the host-first rule (left to the caller's built-in encoder) */
uint64_t kc_s2_hook_declined(const kc_ctx* ctx);

/* ---- timing of the last call (HIP events recorded on the launch stream) ---- */
typedef struct {
    float total_ms;   /* all kernels of the last encode call */
    float match_ms;   /* match-finder kernel(s) */
    float entropy_ms; /* histogram + table build + bitstream emit kernel(s) */
    float other_ms;   /* checksum, scan, compaction */
    uint32_t redo_units; /* units re-run because a speculated block verdict was wrong */
    float prep_ms;    /* part of match_ms: zeroing / dictionary-priming of the per-unit hash tables before the match finder */
} kc_timings;
kc_status kc_last_timings(const kc_ctx* ctx, kc_timings* t);

/* ---- probes (measurement only; nothing on the encode path calls them) ----
 * kc_probe_table_pattern: the match finders' hash-table traffic on THIS device — scattered 4-byte accesses into n_tables tables of
 * table_bytes each (8 lanes per table, 8 tables per wave, two independent accesses per lane per iteration: the two buckets a
 * fastEncoder step looks up and overwrites, zstd/enc_fast.go:147-207).  out3[0] = read + write-back pairs per second, out3[1] = plain
 * reads per second, out3[2] = plain stores per second (each counts one request per access).  bench.py prices the match finder's
 * measured transactions per unit at these rates (roofline.floor).  Allocates and frees n_tables * table_bytes of device memory.
 * kc_probe_pcie: one pinned copy of `bytes` each way.  out7[0] H2D GB/s, [1] D2H GB/s, [2] / [3] H2D / D2H GB/s with both in flight,
 * [4] pageable -> pinned host copy GB/s with the context's copy threads (KC_OPT_HOST_COPY_THREADS), [5] pinned -> pageable, [6] the
 * thread count: the ceilings of the host-buffer entry points (bench.py end_to_end.pcie_ceiling). */
kc_status kc_probe_table_pattern(kc_ctx* ctx, uint32_t n_tables, uint32_t table_bytes, uint32_t waves, uint32_t iters, double* out3);
kc_status kc_probe_pcie(kc_ctx* ctx, uint64_t bytes, double* out7);

/* ---- synthetic corpora: deterministic, per-unit seeded, host-side generator ----
 * kind: 'T' enwik-style text, 'H' high-entropy, 'J' JSON records, 'M' mixed.  Fills
 * n_units * unit_size bytes at dst (host memory) using `threads` host threads. */
kc_status kc_corpus_fill(int kind, uint64_t seed, uint64_t first_unit, uint32_t n_units, uint32_t unit_size, uint8_t* dst, int threads);

#ifdef __cplusplus
}
#endif
#endif /* KCGPU_H */
