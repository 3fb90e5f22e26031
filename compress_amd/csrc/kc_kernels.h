// kc_kernels.h — host-visible parameter blocks and launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kc_dev.h"

// ---- match finders (kc_zstd_match.hip) ----
struct KcMatchParams {
    const uint8_t* src;         // device: concatenated units
    const uint8_t* src_end;     // device: one past the last readable byte of src (kernels never load at or beyond it, nor below src,
                                // except inside the 16-byte aligned granule of a readable byte)
    const uint64_t* unit_off;   // device: n_units+1
    const uint32_t* unit_blk0;  // device: n_units+1, first global block index of each unit
    uint64_t* seqs;             // device scratch: seq_stride packed sequences per block
    KcBlkMeta* meta;            // device: one record per block
    const uint32_t* blk_start;  // device or null: per block, its start inside the unit (streams with Flush points: irregular blocks)
    const uint32_t* unit_flags; // device, with blk_start: bit 0 stream frame (a block was written before Close), bit 1 Close found nothing buffered
    const uint8_t* pop_blk;     // device or null: per block (global index), 1 = its offsets must be popped (re-run)
    const uint32_t* unit_list;  // device or null: indirection for re-runs
    uint32_t unit_base;         // first unit of this launch when unit_list is null (chunked launches)
    uint32_t seq_stride;
    int32_t block_size;
    int32_t max_match_off;
    int32_t spec_w0;            // initial speculation width after a match (group kernels)
    int32_t spec_grow;          // speculation width after a round without a match: 0 keep, 1 +1, 2 double (default)
    int32_t hist0;              // bytes of dictionary content prepended to every unit in `src` (0: no dictionary)
    int32_t pos_bits;           // bits reserved for position+1 in tagged table entries (better level)
    int32_t rep1, rep2;         // initial recentOffsets[0..1]: {1,4} unless a full-format dictionary supplies its own
    int32_t rep3;               // recentOffsets[2] (SpeedBestCompression only)
    int32_t stream_mode;        // units are Write+Close streams: a unit of >= one block is parsed with Encode (history) from its first block
    const uint32_t* unit_hist;  // device or null: per-unit history bytes in front of the unit in `src` (jobs: the overlap prefix), replaces hist0
    const uint32_t* job_flags;  // device or null: units are the jobs of ONE WithConcurrentBlocks stream (enc_jobs.go): bit 0 = final job
    unsigned long long* prof;   // device or null: per-phase shader-clock totals of the LDS-table kernel (built with -DKC_LDS_PROF, KC_OPT_K2_PROF)
    uint32_t epoch;             // SpeedBetterCompression, SpeedFastest (HBM-table kernel): 0 = the tables were fully initialised by the
                                // host; else this launch's stamp (entries with another stamp read as empty — at SpeedBetterCompression
                                // with a dictionary as the dictionary's entry: kc_zstd_match_better.hip, kc_zstd_match.hip)
    const uint8_t* proto;       // device or null: with epoch != 0 and a dictionary, the dictionary's tables (long then short)
    int32_t lds_any_big;        // SpeedFastest LDS kernels: 1 = the launch has a unit above 128 KiB (the first form runs beside the fused one)
    int32_t lds_split;          // SpeedFastest HBM kernel: 1 = skip the units the LDS-table kernel takes (those that fit KC_ZFAST_LDS_MAX_UNIT)
    int32_t empty_filter;       // SpeedFastest HBM kernel: 1 = skip the table loads of bucket groups the unit has not written yet, while it has
                                // emitted no sequence (kc_zstd_match.hip; the tables must start empty: no dictionary, no job prefix)
    int32_t tuned;              // SpeedFastest HBM kernel: 1 = the form compiled with cross-segment rounds (xseg_k) and the empty-group filter
    const uint32_t* unit_done;  // device or null: per unit, 1 = the pre-scan (kc_zstd_prescan.hip) proved that no probe of the unit finds a
                                // match and wrote the unit's block records: the match finder skips it
    int32_t xseg_k;             // SpeedFastest HBM kernel: a probe round continues across a skip-segment boundary once (s - nextEmit) >> 5
                                // has reached this value (0: always; a large value: never, round 2's rounds)
};
// SpeedFastest: 8 lanes per unit, tables = n_launch x 2^15 x u32 in HBM, zeroed by the caller
void kc_launch_zfast_match_grp(const KcMatchParams& P, uint32_t* tables, uint32_t n_launch, hipStream_t st);
static inline size_t kc_zfast_table_bytes() { return (size_t)4 << 15; }
#define KC_ZF_EPOCH_BITS 4  // bits of the launch stamp in a table entry (stamps 1..15; KcMatchParams.epoch 0 = tables cleared per launch)
// SpeedFastest, LDS-table path (kc_zstd_match_lds.hip): one wave per unit, the 2^15 x u32 table in LDS; units (with their
// history) below KC_ZFAST_LDS_MAX_UNIT bytes.  proto: null or primed tables in the HBM entry format (P.pos_bits): one for all units
// (dictionary; proto_stride 0) or one per launch slot (jobs: proto_stride = 2^15 entries)
void kc_launch_zfast_match_lds(const KcMatchParams& P, const uint32_t* proto, uint32_t proto_stride, uint32_t n_launch, hipStream_t st);
#define KC_ZFAST_LDS_MAX_UNIT ((1u << 26) - 4u)
// SpeedDefault: long (2^17) + short (2^15) u32 tables per unit in HBM, zeroed by the caller
void kc_launch_zdfast_match_grp(const KcMatchParams& P, uint32_t* tables, uint32_t n_launch, hipStream_t st);
static inline size_t kc_zdfast_table_bytes() { return ((size_t)4 << 17) + ((size_t)4 << 15); }
// SpeedBetterCompression: long 2^19 x {offset,prev} + short 2^13 x u32 per unit
void kc_launch_zbetter_match_grp(const KcMatchParams& P, uint8_t* tables, uint32_t n_launch, bool dict, hipStream_t st);
static inline size_t kc_zbetter_table_bytes() { return ((size_t)8 << 19) + ((size_t)4 << 13); }
// SpeedBestCompression (kc_zstd_match_best.hip): one wave per unit over n_slots persistent table slots of long 2^22 + short 2^18
// {offset, prev} pairs (34 MiB each, zeroed once); slot_cur: per slot, where its position space stands (0: fresh); cost: 96 int32
// built by kc_launch_zbest_cost from the predefined FSE tables
void kc_launch_zbest_match(const KcMatchParams& P, uint64_t* tables, uint32_t* slot_cur, const int32_t* cost, uint32_t n_launch, uint32_t n_slots,
                           hipStream_t st);
void kc_launch_zbest_cost(const void* d_predef, int32_t* d_cost, hipStream_t st);
static inline size_t kc_zbest_table_bytes() { return ((size_t)8 << 22) + ((size_t)8 << 18); }

// A raw block whose payload the compaction copies straight from the source (KcEntropyParams.rawdef)
struct KcRawDef {
    uint32_t frame_pos;  // position of the payload in the unit's frame
    uint32_t src_pos;    // position of the block in the unit (history included)
    uint32_t size;       // 0: nothing deferred for this block
    uint32_t pad;
};

// ---- entropy + emit (kc_zstd_entropy.hip) ----
struct KcFsePredef;  // opaque device blob built by kc_launch_fse_predef_init
struct KcEntropyParams {
    const uint8_t* src;
    const uint64_t* unit_off;
    const uint32_t* unit_blk0;
    const uint64_t* seqs;
    const KcBlkMeta* meta;
    uint8_t* lits;          // device scratch: lit_stride bytes per block (gathered literals, later seq bitstream staging)
    uint64_t* aux;          // device scratch: seq_stride u64 per block (huffman stream staging, then FSE state bits)
    uint8_t* stage;         // device: per-unit staging area for the encoded frame
    const uint64_t* stage_off;  // device: n_units+1 offsets into stage (16-byte aligned)
    uint32_t* out_size;     // device: encoded size per unit
    const uint64_t* xxh;    // device: XXH64 per unit (only read when crc != 0)
    uint32_t* redo_mask;    // device: per unit, non-zero when a block's late raw fallback invalidated carried offsets
    uint8_t* redo_blk;      // device: per block (global index), 1 = that block
    const uint32_t* blk_start;  // as in KcMatchParams
    const uint32_t* unit_flags;
    const uint32_t* unit_list;
    uint32_t unit_base;     // first unit of this launch when unit_list is null
    const void* predef;     // device: KcFsePredef
    uint32_t seq_stride;
    uint32_t lit_stride;
    int32_t block_size;
    int32_t window_size;
    int32_t crc, single, no_entropy, all_lit_entropy, full_zero;
    uint32_t dict_id;
    int32_t hist0;          // bytes of dictionary content prepended to every unit in `src`
    const uint32_t* unit_hist;  // as in KcMatchParams
    const uint32_t* job_flags;  // as in KcMatchParams: the unit's output is its blocks alone (no frame header, no checksum), `last` only
                                // on the final job's last block; an empty final job is one empty raw last block (enc_jobs.go:96-103)
    const uint8_t* dict_huf;   // device or null: dictionary literal table, 256 x u16 val then 256 x u8 nBits (KcHufTable layout)
    int32_t dict_huf_len, dict_huf_log;
    int32_t stream_mode;       // streaming frame layout for units >= one block (zstd/encoder.go:257-428): no content size, no single
                               // segment, last flag only on a short final block, else an empty raw last block
    int32_t stream_sync;       // WithEncoderConcurrency(1): the synchronous nextBlock form (zstd/encoder.go:364-391) resets the block
                               // before its first Encode too, so a stream frame never sees the dictionary literal table
    uint32_t* unit_raw;     // device or null: per unit, 1 = every block of the frame is raw with its payload deferred (rawdef) — then
                            // kc_xxh64_fin_kernel copies the payloads; with xxh == null (and crc) the checksum field is left to it too
    KcRawDef* rawdef;       // device or null: per block (global index), where a RAW block's payload goes in the frame and where it comes
                            // from in the unit: the entropy kernel then writes the 3-byte header only and kc_compact_kernel copies
                            // the payload once, from the source (instead of source -> staging slot -> output)
    uint32_t* err_flag;     // device: set non-zero on a device-side invariant violation
    unsigned long long* prof;  // device or null: per-phase shader-clock totals (diagnostics, KC_K2_PROF=1)
    const uint32_t* unit_done;  // device or null: per unit, 1 = the pre-scan wrote the unit's whole frame (raw blocks only): nothing to do here
};
size_t kc_fse_predef_bytes();
void kc_launch_fse_predef_init(void* d_predef, hipStream_t st);
void kc_launch_zstd_entropy(const KcEntropyParams& P, uint32_t grid, hipStream_t st);

// ---- SpeedFastest pre-scan for input without matches (kc_zstd_prescan.hip) ----
// While a block has produced no sequence, the probe positions of fastEncoder.Encode are a function of the block start alone
// (s += 2 + ((s - nextEmit) >> 5), enc_fast.go:207), and a probe can only find a candidate that an EARLIER probe of the unit
// inserted into the same bucket with the same 4 bytes (tableEntry.val == uint32(cv), enc_fast.go:176,188; no repeat check before
// the third sequence).  So "no two probe inserts of the unit share (bucket, 4 bytes)" proves that the whole unit parses to zero
// sequences — without a table in HBM.  Such a unit's frame is fully determined (every block raw: blockenc.go:337-352 with
// rawAllLits): the kernel writes the block records, the frame header, the block headers and the raw-payload descriptors, and
// flags the unit done; the match finder and the entropy kernel skip it, kc_xxh64_fin_kernel hashes and copies the payload.
// Units with a possible match (or too many probes for the on-chip set) are left to the regular kernels, untouched.
struct KcPrescanParams {
    const uint8_t* src;
    const uint64_t* unit_off;   // n_units + 1
    const uint32_t* unit_blk0;  // n_units + 1
    uint32_t n_units;
    int32_t block_size;
    const uint32_t* probe_rel;  // device: the probe positions of a block of block_size bytes relative to its start, ascending
    uint32_t n_probe;           // how many (those below block_size - 8)
    int32_t rep1, rep2;         // recentOffsets the unit starts with (carried through blocks without sequences)
    KcBlkMeta* meta;
    uint32_t* unit_done;        // out, per unit: 1 = done here
    uint8_t* stage;
    const uint64_t* stage_off;
    uint32_t* out_size;
    KcRawDef* rawdef;
    uint32_t* unit_raw;
    int32_t window_size, crc, single;
    uint32_t dict_id;
};
#define KC_PRESCAN_MAX_KEYS 1024  // (bucket, value) pairs one unit may insert: two per probe; the on-chip set has twice as many slots
void kc_launch_zfast_prescan(const KcPrescanParams& P, hipStream_t st);
// the probe positions of a block of `block_size` bytes relative to its start (host helper: fills rel[], returns the count)
uint32_t kc_zfast_probe_positions(int block_size, uint32_t* rel, uint32_t cap);

// ---- job tables primed from the overlap prefix (kc_zstd_prime.hip) ----
// encoder.ResetPrefix of a WithConcurrentBlocks job (enc_jobs.go:325-331 -> enc_fast.go:800-811, enc_dfast.go:1040-1050,
// enc_better.go:1099-1112): the unit's first unit_hist[u] bytes are the previous job's tail; their positions go into the unit's
// table slot (zeroed by the caller) in the entry format of the match finders, (position + 1) | tag << pos_bits.
struct KcPrimeParams {
    const uint8_t* src;
    const uint64_t* unit_off;   // device: n_units + 1
    const uint32_t* unit_hist;  // device: per unit, bytes of prefix
    const uint32_t* unit_list;  // device or null: slot i primes unit unit_list[i] (re-runs), else unit_base + i
    uint32_t unit_base;
    uint32_t n_launch;          // table slots
    int32_t level;              // KC_SPEED_FASTEST / DEFAULT / BETTER_COMPRESSION's values (1, 2, 3)
    int32_t pos_bits;
    uint8_t* tables;            // device: n_launch slots of table_bytes
    size_t table_bytes;
};
void kc_launch_zstd_prime(const KcPrimeParams& P, hipStream_t st);

// ---- S2 block encoder (kc_s2.hip) ----
struct KcS2Params {
    const uint8_t* src;
    const uint64_t* blk_off;    // device, n+1
    const uint64_t* stage_off;  // device, n+1 (16-byte aligned slots of MaxEncodedLen)
    uint8_t* stage;
    uint32_t* out_size;
    uint32_t* tables;           // n x table_stride u32, zeroed by the caller
    uint32_t table_stride;      // u32 entries per block: kc_s2_table_bytes(level, max block length) / 4
    uint32_t n_blocks;
    int32_t level;              // 0: s2.Encode, 1: s2.EncodeBetter, 2: s2.EncodeSnappy, 3: s2.EncodeSnappyBetter, 4: s2.EncodeBest, 5: s2.EncodeSnappyBest
    int32_t spec_w0, spec_w0b;  // speculation width after a match (default / better parse)
    int32_t spec_grow;          // after a round without a match: 0 keep, 1 +1, 2 double
    int32_t framed;             // 1: emit s2.Writer chunks (type | len24 | masked CRC32C | body), s2/writer.go:414-451
    int32_t stored_only;        // framed mode, s2.WriterUncompressed (s2/writer.go:951): every block becomes an uncompressed chunk (type 0x01 | len | CRC32C | bytes)
    int32_t variant;            // levels 0 and 2: 0 = the bytes of the portable Go encoders (encode_all.go; arm64 and noasm builds),
                                // 1 = the bytes of the amd64 assembly encoders (encode_amd64.go + encodeblock_amd64.s)
};
void kc_launch_s2_encode(const KcS2Params& P, hipStream_t st);
// s2.EncodeBest (level 4) / s2.EncodeSnappyBest (level 5): kc_s2_best.hip, one wave per block, 4.5 MiB of {cur, prev} tables per block
void kc_launch_s2_best(const KcS2Params& P, hipStream_t st);
// LDS-table path (kc_s2_lds.hip): one wave per block, levels 0 and 2; any_small / any_big: the batch has blocks <= / > 64 KiB
void kc_launch_s2_encode_lds(const KcS2Params& P, bool any_small, bool any_big, hipStream_t st);
struct KcS2DecParams {
    const uint8_t* enc;         // encoded blocks (uvarint length + body each)
    const uint64_t* enc_off;    // device, n+1
    uint8_t* dst;
    const uint64_t* dst_off;    // device, n+1: where each block decodes to, and how long it must be
    uint32_t* status;           // device, n: 0 ok, else the first error met
    uint32_t n_blocks;
};
void kc_launch_s2_decode(const KcS2DecParams& P, hipStream_t st);
struct KcZstdDecParams {
    const uint8_t* enc;         // one frame per unit
    const uint64_t* enc_off;    // device, n+1
    uint8_t* dst;
    const uint64_t* dst_off;    // device, n+1: where each frame decodes to, and how long its content must be
    uint8_t* lits;              // device scratch: lit_stride bytes per unit (Huffman-decoded literals of the current block)
    uint32_t lit_stride;
    uint32_t* status;           // device, n: 0 ok, else the first error met
    uint32_t* crc_stored;       // device, n: the frame's stored XXH64 low word
    uint32_t* has_crc;          // device, n
    uint32_t n_units;
    const uint8_t* dict;        // device or null: raw dictionary content = history in front of every frame
    uint32_t dict_len;
};
void kc_launch_zstd_decode(const KcZstdDecParams& P, hipStream_t st);
// default: 2^14 entries; better: long 2^17 + short 2^14 (blocks > 64 KiB), long 2^16 + short 2^13 (all blocks <= 64 KiB)
static inline size_t kc_s2_table_bytes(int level, uint64_t max_block_len, int variant = 0) {
    // the assembly forms of the better levels take 2^17 + 2^14 entries from 16 KiB on (Snappy-compatible: above 64 KiB; 2^16 + 2^13 below)
    if (variant == 1 && level == 1 && max_block_len >= ((uint64_t)16 << 10)) return ((size_t)4 << 17) + ((size_t)4 << 14);
    if (variant == 1 && level == 3) return max_block_len > ((uint64_t)64 << 10) ? (((size_t)4 << 17) + ((size_t)4 << 14)) : (((size_t)4 << 16) + ((size_t)4 << 13));
    if (level >= 4) return ((size_t)8 << 19) + ((size_t)8 << 16);  // best: long 2^19 + short 2^16 entries of {cur, prev}
    if (level == 3) return max_block_len > ((uint64_t)64 << 10) ? (((size_t)4 << 16) + ((size_t)4 << 14)) : (((size_t)4 << 15) + ((size_t)4 << 13));
    if (level != 1) return (size_t)4 << 14;
    return max_block_len > ((uint64_t)64 << 10) ? (((size_t)4 << 17) + ((size_t)4 << 14)) : (((size_t)4 << 16) + ((size_t)4 << 13));
}

// ---- misc (kc_misc.hip) ----
// work[work_off[i] ..] = dict (dict_len bytes) || src[unit_off[i] .. unit_off[i+1])
void kc_launch_prefix_units(const uint8_t* src, const uint64_t* unit_off, const uint64_t* work_off, const uint8_t* dict, uint32_t dict_len,
                            uint8_t* work, uint32_t n, hipStream_t st);
// dst[i*bytes ..] = proto[0 .. bytes) for i < n  (bytes multiple of 16)
void kc_launch_bcast(const uint8_t* proto, uint8_t* dst, size_t bytes, uint32_t n, hipStream_t st);
void kc_launch_xxh64(const uint8_t* src, const uint64_t* unit_off, uint32_t n_units, uint64_t* out, hipStream_t st);
// XXH64 behind the entropy stage and the size scan (kc_misc.hip): writes the frame's checksum field into its staging slot and, for
// the units flagged in unit_raw (every block raw with a deferred payload: KcEntropyParams.unit_raw), copies the payloads from the
// source into dst while hashing them.  Regular block grid, block size a multiple of 256, no history in front of the units.
struct KcXxhFinParams {
    const uint8_t* src;
    const uint64_t* unit_off;   // n_units + 1
    uint32_t n_units;
    uint8_t* stage;
    const uint64_t* stage_off;
    const uint32_t* out_size;   // frame sizes (checksum field included)
    const uint64_t* out_off;    // frame positions in dst (the size scan's result)
    uint8_t* dst;
    const uint32_t* unit_raw;   // or null: no payload is copied here
    const KcRawDef* rawdef;
    const uint32_t* unit_blk0;
    uint64_t* xxh_out;          // or null
    int32_t mode;               // payload copy of the raw-only frames: 0 stored straight from the registers, 1 the same with the next step's
                                // loads issued before the current step's stores, 2 through an LDS ring as 16-byte aligned stores
};
void kc_launch_xxh64_fin(const KcXxhFinParams& P, hipStream_t st);
// up to 8 device ranges zeroed by one launch: p[k] 16-byte aligned, n16[k] 16-byte words
struct KcClearList { void* p[8]; uint64_t n16[8]; int count; };
void kc_launch_clear(const KcClearList& L, hipStream_t st);
// exclusive scan of sizes (u32) into offsets (u64, n+1 entries)
void kc_launch_scan_sizes(const uint32_t* sizes, uint32_t n, uint64_t* out_off, hipStream_t st);
// dst[out_off[i] .. ) = stage[stage_off[i] .. +sizes[i]); with rawdef: except the deferred raw-block payloads, which come from
// src[unit_off[i] + src_pos ..) (unit_blk0[i] .. unit_blk0[i+1] are unit i's entries of rawdef)
void kc_launch_compact(const uint8_t* stage, const uint64_t* stage_off, const uint32_t* sizes, const uint64_t* out_off,
                       uint8_t* dst, uint32_t n, hipStream_t st, const uint8_t* src = nullptr, const uint64_t* unit_off = nullptr,
                       const uint32_t* unit_blk0 = nullptr, const KcRawDef* rawdef = nullptr, const uint32_t* unit_raw = nullptr);
