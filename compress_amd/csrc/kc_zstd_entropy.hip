// kc_zstd_entropy.hip — per-unit block serialisation for gfx950: literal gather + 256-bin
// histogram, huff0 table build and 1/4-stream Huffman emit, sequence code histograms,
// FSE table build / mode choice / NCount headers, the interleaved FSE bitstream, block and
// frame headers, raw / RLE fallbacks and the frame checksum.
//
// Replaces blockEnc.encode / encodeLits / encodeRLE / encodeRawTo (zstd/blockenc.go:310-827),
// huff0.Compress1X/4X (huff0/compress.go:14-302), frameHeader.appendTo
// (zstd/frameenc.go:25-92) and the per-frame glue of (*Encoder).encodeAll
// (zstd/encoder.go:731-839).
//
// One 256-thread workgroup per unit; blocks of a unit are processed in order because
// they share entropy state (huff0 prevTable + Reuse policy, FSE "prev" tables).  Within a
// block the O(n) passes are data parallel:
//   * literals are gathered from the source through a tile-wise exclusive scan of the
//     sequence list (the match finder stores no literal bytes);
//   * Huffman streams: one wave per stream, per-lane bit counts -> wave scan -> lanes OR
//     their codes into a zeroed, word-aligned staging area;
//   * FSE: the three state chains are inherently serial (tANS) and run on three lanes of
//     three different waves over LDS-resident code chunks; bit packing of state bits and
//     extra bits is parallel over sequences with a block-wide scan of bit lengths.
// The O(alphabet) table constructions run on one lane per table out of LDS.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_fse_dev.h"
#include "kc_huf_dev.h"
#include "kc_frame_dev.h"

#define ET 256            // threads per workgroup
#ifndef SEQ_CHUNK
#define SEQ_CHUNK 1024    // sequences staged in LDS per FSE chain chunk
#endif
#ifndef KC_PACK_REUSE
#define KC_PACK_REUSE 1   // the bit packing takes the chunk's sequences from the registers the code staging loaded them into (four per thread, kept across the chain phase) instead of loading them again: 18.83 -> 18.53 ms (profiles/r05_entropy_pack_reuse.txt)
#endif
#ifndef KC_HCOPIES
#define KC_HCOPIES 2   // 2: 19.19-19.26 vs 19.36-19.41 ms on one box, SQ_LDS_ADDR_CONFLICT 581 M -> 504 M per launch (profiles/r05_entropy_hist_copies.txt)
#endif
#ifndef KC_K2_WGS
#define KC_K2_WGS 4
#endif
#ifndef KC_CHAIN_WARM
#define KC_CHAIN_WARM 48  // warm-up symbols per tANS chain segment (speculation, verified; any multiple of 16 up to 48 is exact)
#endif
#define CH_PAD 16         // front pad of the chain slots (the init-only element of a block's first chunk sits at CH_PAD - 1)
#define CH_SEG 16         // stream elements per lane segment of a tANS chain
#define CH_WB (KC_CHAIN_WARM / CH_SEG)  // warm-up, in segments
static_assert(KC_CHAIN_WARM % CH_SEG == 0 && CH_WB >= 1 && CH_WB <= 3, "warm-up is whole segments");
static_assert(SEQ_CHUNK % CH_SEG == 0 && SEQ_CHUNK / CH_SEG <= 64, "one segment per lane");
#define LONG_RUN 32       // literal runs longer than this are copied cooperatively by the wave (64 x LONG_RUN fits one LDS window)

static_assert(sizeof(KcFsePredefBlob) == 3 * sizeof(KcFseT), "blob layout");
size_t kc_fse_predef_bytes() { return sizeof(KcFsePredefBlob); }

// ---------------------------------------------------------------------------------------
// predefined encoders (zstd/fse_predefined.go:112-155, encoder half)
// ---------------------------------------------------------------------------------------
__global__ void kc_fse_predef_init_kernel(KcFsePredefBlob* B) {
    __shared__ uint8_t tsym[3][256];
    __shared__ int16_t cumul[3][66];
    const int i = threadIdx.x;
    if (i >= 3) return;
    const int16_t llNorm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
    const int16_t ofNorm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
    const int16_t mlNorm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
    KcFseT* f = &B->t[i];
    for (int k = 0; k < 64; k++) { f->norm[k] = 0; f->dnb[k] = 0; f->dfs[k] = 0; f->outBits[k] = 0; }
    for (int k = 0; k < 256; k++) f->st[k] = 0;
    if (i == 0) { for (int k = 0; k < 36; k++) f->norm[k] = llNorm[k]; f->symbolLen = 36; f->tableLog = 6; }
    if (i == 1) { for (int k = 0; k < 29; k++) f->norm[k] = ofNorm[k]; f->symbolLen = 29; f->tableLog = 5; }
    if (i == 2) { for (int k = 0; k < 53; k++) f->norm[k] = mlNorm[k]; f->symbolLen = 53; f->tableLog = 6; }
    f->useRLE = 0; f->rleVal = 0; f->reUsed = 0; f->stLen1 = 0;
    fse_build_core<int16_t>(f->norm, f->symbolLen, f->tableLog, tsym[i], cumul[i], f->st, f->dnb, f->dfs);
    for (int k = 0; k < (int)f->symbolLen; k++)
        f->outBits[k] = (uint8_t)(i == 0 ? kc_ll_bits(k) : (i == 1 ? (uint32_t)k : kc_ml_bits(k)));
    f->preDefined = 1;
}
void kc_launch_fse_predef_init(void* d_predef, hipStream_t st) {
    hipLaunchKernelGGL(kc_fse_predef_init_kernel, dim3(1), dim3(64), 0, st, (KcFsePredefBlob*)d_predef);
}

// ---------------------------------------------------------------------------------------
// workgroup primitives
// ---------------------------------------------------------------------------------------
// Two independent 32-bit sums packed in one value: neither half may overflow (callers: byte and bit counts of one block).
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v, int lane) {
    const uint32_t lo = wave_incl_scan((uint32_t)v, lane), hi = wave_incl_scan((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
// Exclusive scan over the 256 threads; *total receives the block sum.  wsum: 4 x u64 LDS scratch.
__device__ __forceinline__ uint64_t block_excl_scan64(uint64_t v, uint64_t* wsum, uint64_t* total, int tid) {  // tid: the kernel's (rotated) thread index
    const int lane = tid & 63, w = tid >> 6;
    const uint64_t inc = wave_incl_scan64(v, lane);
    __syncthreads();  // protect wsum reuse
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint64_t base = 0;
    for (int k = 0; k < w; k++) base += wsum[k];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - v;
}

// fseEncoder.approxSize (zstd/fse_encoder.go:603-660) with one lane per symbol: the sum is a wrapping uint32 sum, so
// the lane order does not matter; any "impossible" symbol makes the whole estimate MaxUint32 like the serial code.
__device__ __forceinline__ uint32_t wave_approx_size(const KcFseT* f, const uint32_t* hist, int histLen, int lane) {
    if ((int)f->symbolLen < histLen) return 0xFFFFFFFFu;
    if (f->useRLE) return 0xFFFFFFFFu;
    const uint32_t kAccuracyLog = 8;
    const uint32_t badCost = ((uint32_t)f->tableLog + 1) << kAccuracyLog;
    uint32_t term = 0;
    bool bad = false;
    if (lane < histLen) {
        const uint32_t v = hist[lane];
        if (v != 0) {
            if (f->norm[lane] == 0) bad = true;
            else {
                const uint32_t minNbBits = f->dnb[lane] >> 16;
                const uint32_t threshold = (minNbBits + 1) << 16;
                const uint32_t tableSize = 1u << f->tableLog;
                const uint32_t deltaFromThreshold = threshold - (f->dnb[lane] + tableSize);
                const uint32_t normalizedDelta = (deltaFromThreshold << kAccuracyLog) >> f->tableLog;
                const uint32_t bc = (minNbBits + 1) * (1u << kAccuracyLog) - normalizedDelta;
                if (bc > badCost) bad = true;
                term = v * bc;
            }
        }
    }
    if (__ballot(bad) != 0ull) return 0xFFFFFFFFu;
    return wave_reduce_sum(term) >> kAccuracyLog;
}

// ---------------------------------------------------------------------------------------
// shared state
// ---------------------------------------------------------------------------------------
struct HufState {  // huff0.Scratch fields that persist across the blocks of a unit
    KcHufTable prev;   // prevTable
    int prevLen;       // len(prevTable)
    uint8_t prevLog;   // prevTableLog
    int reuse;         // Reuse policy: 0 Allow, 2 None (only these two occur on this path)
};

struct Shared {
    // --- literal histogram / huffman build ---
    uint32_t whist[4 * KC_HCOPIES][256];   // literal histogram copies: per wave, and per lane parity inside a wave when KC_HCOPIES is 2 (same-address LDS atomics serialise)
    uint32_t cnt[256];
    KcHufNodes nodes;
    KcHufTable cur;     // freshly built cTable
    HufState huf;
    uint8_t weights[256];
    KcWeightFse wfse;
    KcHufWaveTmp hwt;   // work arrays of the wave-cooperative buildCTable
    uint8_t tdesc[192];  // serialised Huffman table description
    // --- sequences ---
    KcFseT fse[9];       // 0..5: ll/of/ml cur+prev pool, 6..8: predefined LL/OF/ML
    uint8_t curIdx[3], prevIdx[3];  // pool index per table kind (0 LL, 1 OF, 2 ML)
    uint8_t useIdx[3];   // encoder chosen for this block
    uint32_t shist[3][64];
    uint32_t smax[3];
    uint8_t tsym[3][256];
    int16_t cumul[3][66];
    int16_t posx[3][66];
    uint8_t seqhdr[224];
    uint32_t asz[3][3];   // approxSize of {new, predefined, previous} encoder per stream
    uint8_t nc[3][72];    // NCount bytes per stream (maxHeaderSize <= ((53 * 9) >> 3) + 3)
    int ncLen[3];
    uint8_t seqMode;
    int seqhdrLen;
    // Chain slots (round 5): the staged element j of a chunk (stream order) lives at index CH_PAD + j - jb, jb = 1 in a block's first chunk
    // (whose element 0 only initialises the states) — so that chain position x = j - jb of lane x / 16 starts a 16-byte aligned row:
    // a lane fetches the 16 codes of its segment with ONE ds_read_b128 and leaves its 16 results with two ds_write_b128
    alignas(16) uint8_t codes[3][CH_PAD + SEQ_CHUNK + 16];     // ll / of / ml code per staged sequence
    alignas(16) uint16_t sbits[3][CH_PAD + SEQ_CHUNK + 16];    // state bits emitted for that sequence: nb<<12 | value
    uint32_t cpk[3][64];             // per code of the block's three encoders: deltaNbBits << 12 | (deltaFindState & 0xFFF), one lookup per symbol
    uint16_t state[3];               // running FSE states (ll, of, ml)
    // --- scan / misc ---
    uint64_t wsum[4];
    uint32_t wtot[4];
    int ivar[24];  // broadcast slots
    unsigned long long profAcc[24];  // KC_OPT_K2_PROF: shader clocks per phase (0..15) and per sub-step of a measurement build (16..23)
};

enum { IV_SYMLEN = 0, IV_MAXCNT, IV_CANREUSE, IV_LITMODE, IV_USEPREV, IV_TABLOG, IV_DESCLEN, IV_DATALEN, IV_LITSEC,
       IV_OK, IV_SEQBYTES, IV_RAW, IV_TMP0, IV_TMP1 };

// ---------------------------------------------------------------------------------------
// byte helpers
// ---------------------------------------------------------------------------------------
// literalsHeader.setSize (blockenc.go:150): raw / RLE literal sections. Returns header size.
__device__ __forceinline__ int lit_header_size1(int regenLen) {
    const int inBits = bits_len32((uint32_t)regenLen);
    return inBits < 5 ? 1 : (inBits < 12 ? 2 : 3);
}
__device__ __forceinline__ int put_lit_header1(uint8_t* p, uint32_t type, int regenLen) {
    const int inBits = bits_len32((uint32_t)regenLen);
    uint64_t lh = type;
    int sz;
    if (inBits < 5) { lh |= ((uint64_t)regenLen << 3); sz = 1; }
    else if (inBits < 12) { lh |= (1 << 2) | ((uint64_t)regenLen << 4); sz = 2; }
    else { lh |= (3 << 2) | ((uint64_t)regenLen << 4); sz = 3; }
    for (int i = 0; i < sz; i++) p[i] = (uint8_t)(lh >> (8 * i));
    return sz;
}
// literalsHeader.setSizes (blockenc.go:178): compressed literal sections.
__device__ __forceinline__ int lit_header_size2(int compLen, int inLen) {
    const int compBits = bits_len32((uint32_t)compLen), inBits = bits_len32((uint32_t)inLen);
    if (compBits <= 10 && inBits <= 10) return 3;
    if (compBits <= 14 && inBits <= 14) return 4;
    return 5;
}
__device__ __forceinline__ int put_lit_header2(uint8_t* p, uint32_t type, int compLen, int inLen, bool single) {
    const int compBits = bits_len32((uint32_t)compLen), inBits = bits_len32((uint32_t)inLen);
    uint64_t lh = type;
    int sz;
    if (compBits <= 10 && inBits <= 10) {
        if (!single) lh |= 1 << 2;
        lh |= ((uint64_t)inLen << 4) | ((uint64_t)compLen << (10 + 4));
        sz = 3;
    } else if (compBits <= 14 && inBits <= 14) {
        lh |= (2 << 2) | ((uint64_t)inLen << 4) | ((uint64_t)compLen << (14 + 4));
        sz = 4;
    } else {
        lh |= (3 << 2) | ((uint64_t)inLen << 4) | ((uint64_t)compLen << (18 + 4));
        sz = 5;
    }
    for (int i = 0; i < sz; i++) p[i] = (uint8_t)(lh >> (8 * i));
    return sz;
}

// cooperative byte copy (global -> global), 256 threads
__device__ __forceinline__ void wg_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int n) {
    const int tid = threadIdx.x;
    // head to 4-byte dst alignment is not attempted: source alignment is arbitrary; go bytewise in
    // 4-byte units with unaligned loads and byte-exact tail.
    const int n4 = n >> 2;
    if ((((uintptr_t)dst) & 3) == 0) {
        for (int i = tid; i < n4; i += ET) ((uint32_t*)dst)[i] = ld32(src + 4 * i);
        for (int i = (n4 << 2) + tid; i < n; i += ET) dst[i] = src[i];
    } else {
        for (int i = tid; i < n; i += ET) dst[i] = src[i];
    }
}

// OR `nbits` (<= 57) bits of `value` at absolute bit position `bitpos` of a zeroed,
// 4-byte aligned global buffer.
__device__ __forceinline__ void or_bits(uint32_t* words, uint64_t bitpos, uint64_t value, int nbits) {
    if (nbits == 0) return;
    const uint32_t w = (uint32_t)(bitpos >> 5);
    const int sh = (int)(bitpos & 31);
    const uint64_t lo = value << sh;                    // bits for words w, w+1
    atomicOr(&words[w], (uint32_t)lo);
    if (sh + nbits > 32) atomicOr(&words[w + 1], (uint32_t)(lo >> 32));
    if (sh + nbits > 64) atomicOr(&words[w + 2], (uint32_t)(value >> (64 - sh)));
}

// same into an LDS word buffer (ds_or_b32)
__device__ __forceinline__ void lds_or_bits(uint32_t* words, uint64_t bitpos, uint64_t value, int nbits) {
    if (nbits == 0) return;
    const uint32_t w = (uint32_t)(bitpos >> 5);
    const int sh = (int)(bitpos & 31);
    const uint64_t lo = value << sh;
    atomicOr(&words[w], (uint32_t)lo);
    if (sh + nbits > 32) atomicOr(&words[w + 1], (uint32_t)(lo >> 32));
    if (sh + nbits > 64) atomicOr(&words[w + 2], (uint32_t)(value >> (64 - sh)));
}

// ---------------------------------------------------------------------------------------
// Huffman stream emit: one wave per stream.
// Stream symbols are encoded last-to-first (huff0/compress.go:233-266) followed by the end
// mark bit; lane l encodes reversed positions [l*chunk, (l+1)*chunk).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t huf_lane_bits(const uint8_t* __restrict__ seg, int segLen, const KcHufTable* T, int lane, int chunk) {
    uint32_t bits = 0;
    const int r0 = lane * chunk;
    int r1 = r0 + chunk;
    if (r1 > segLen) r1 = segLen;
    // forward positions hi..lo, walked downwards; 8 symbols per global load (a byte load per symbol costs one
    // 64-address memory instruction each — the texture/L1 path, not the ALU, was the limiter)
    int p = segLen - 1 - r0;
    const int lo = segLen - r1;
    for (; p - 15 >= lo; p -= 16) {  // two loads in flight per round trip (the literals were just written: they come from the L2)
        const uint64_t v = ld64(seg + p - 7), v2 = ld64(seg + p - 15);
#pragma unroll
        for (int b = 7; b >= 0; b--) bits += T->nb[(uint32_t)(v >> (8 * b)) & 0xFFu];
#pragma unroll
        for (int b = 7; b >= 0; b--) bits += T->nb[(uint32_t)(v2 >> (8 * b)) & 0xFFu];
    }
    for (; p - 7 >= lo; p -= 8) {
        const uint64_t v = ld64(seg + p - 7);
#pragma unroll
        for (int b = 7; b >= 0; b--) bits += T->nb[(uint32_t)(v >> (8 * b)) & 0xFFu];
    }
    for (; p >= lo; p--) bits += T->nb[seg[p]];
    return bits;
}
__device__ __forceinline__ void huf_lane_emit(const uint8_t* __restrict__ seg, int segLen, const KcHufTable* T, int lane, int chunk,
                                              uint32_t* words, uint64_t bitpos) {
    const int r0 = lane * chunk;
    int r1 = r0 + chunk;
    if (r1 > segLen) r1 = segLen;
    // The lane owns the bit range [bitpos, bitpos + laneBits): only its first and last word can be shared with the
    // neighbouring lanes (atomicOr); every word in between is written whole with a plain store.
    uint32_t* wp = words + (bitpos >> 5);
    uint64_t acc = 0;
    int nb = (int)(bitpos & 31);
    bool firstWord = true;
    auto put = [&](uint32_t sym) {
        acc |= (uint64_t)T->val[sym] << nb;
        nb += T->nb[sym];
        if (nb >= 32) {
            if (firstWord) { atomicOr(wp, (uint32_t)acc); firstWord = false; }
            else *wp = (uint32_t)acc;
            wp++;
            acc >>= 32;
            nb -= 32;
        }
    };
    int p = segLen - 1 - r0;
    const int lo = segLen - r1;
    for (; p - 15 >= lo; p -= 16) {
        const uint64_t v = ld64(seg + p - 7), v2 = ld64(seg + p - 15);
#pragma unroll
        for (int b = 7; b >= 0; b--) put((uint32_t)(v >> (8 * b)) & 0xFFu);
#pragma unroll
        for (int b = 7; b >= 0; b--) put((uint32_t)(v2 >> (8 * b)) & 0xFFu);
    }
    for (; p - 7 >= lo; p -= 8) {
        const uint64_t v = ld64(seg + p - 7);
#pragma unroll
        for (int b = 7; b >= 0; b--) put((uint32_t)(v >> (8 * b)) & 0xFFu);
    }
    for (; p >= lo; p--) put(seg[p]);
    if (nb > 0 && (uint32_t)acc != 0u) atomicOr(wp, (uint32_t)acc);
}

// ---------------------------------------------------------------------------------------
// tANS chain over one 16-symbol segment held in registers (blockenc.go:757-787, fse_encoder.go cState.encode): pk[q] is the packed
// constant of the q-th symbol (cpk above), tbl the encoder's state table.  Only the state-table read is on the dependent chain:
// t = (st << 12) + pk = (st + deltaNbBits) << 12 | dfs12, nbBitsOut = t >> 28, st' = tbl[(st >> nbBitsOut) + deltaFindState].
// EMIT: the state bits of step q (nb << 12 | value) packed two per register.  Returns the state after `cnt` steps (cnt <= 16:
// the slots behind a chunk's last element hold a valid code, their steps run and are ignored).
// ---------------------------------------------------------------------------------------
template <bool EMIT>
__device__ __forceinline__ uint32_t chain_seg16(uint32_t st, const uint32_t (&pk)[CH_SEG], int cnt, const uint16_t* __restrict__ tbl, uint32_t (&sbp)[CH_SEG / 2]) {
    uint32_t out = st;
#pragma unroll
    for (int q = 0; q < CH_SEG; q++) {
        const uint32_t w = pk[q];
        const uint32_t t = (st << 12) + w;
        const uint32_t nb = t >> 28;
        const char* pb = (const char*)tbl + (((int32_t)(w << 20)) >> 19);  // &tbl[deltaFindState], off the dependent chain
        if (EMIT) {
            const uint32_t v = (nb << 12) | (st & ((1u << nb) - 1u));
            if (q & 1) sbp[q >> 1] |= v << 16; else sbp[q >> 1] = v;
        }
        st = *(const uint16_t*)(pb + ((st >> nb) << 1));
        if (q + 1 == cnt) out = st;
    }
    return cnt >= CH_SEG ? st : out;
}
// the 16 codes at cod[0..16) (16-byte aligned: one ds_read_b128) -> their packed constants
__device__ __forceinline__ void chain_load16(const uint8_t* __restrict__ cod, const uint32_t* __restrict__ cpk, uint32_t (&pk)[CH_SEG]) {
    const uint4 c4 = *(const uint4*)cod;
    const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int q = 0; q < CH_SEG; q++) pk[q] = cpk[(cw[q >> 2] >> (8 * (q & 3))) & 0xFFu];
}

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(ET, KC_K2_WGS) void kc_zstd_entropy_kernel(KcEntropyParams P) {
    __shared__ Shared S;
#ifdef KC_K2_PAD
    __shared__ uint32_t padLds[KC_K2_PAD / 4];  // occupancy experiment only
    if (P.block_size == -12345) padLds[threadIdx.x] = 1;
#endif
    // (Rotating the thread roles by whole waves per workgroup — every workgroup's "wave 0" with its single-wave phases on another SIMD —
    // was measured in round 5 and changes nothing: profiles/r05_entropy_rotation.txt.)
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t u = P.unit_list ? P.unit_list[blockIdx.x] : P.unit_base + blockIdx.x;
    if (P.unit_done != nullptr && P.unit_done[u] != 0u) return;  // the pre-scan proved the unit free of matches and wrote its frame (kc_zstd_prescan.hip)
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int hist0 = P.unit_hist != nullptr ? (int)P.unit_hist[u] : P.hist0;  // history in front of the unit (dictionary content, or a job's overlap prefix): not part of the output
    const bool JOB = P.job_flags != nullptr;  // the unit is a job of a WithConcurrentBlocks stream: blocks only (compressJob, enc_jobs.go:88-124)
    const bool finalJob = JOB && (P.job_flags[u] & 1u) != 0;
    const int ulen = (int)(P.unit_off[u + 1] - P.unit_off[u]) - hist0;
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
    const int nblk = UB.nblk;
    uint8_t* __restrict__ outp = P.stage + P.stage_off[u];
    const bool rawAllLits = !P.all_lit_entropy;  // blk.encode(src, noEntropy, !allLitEntropy) (encoder.go:795)
    const bool noEntropy = P.no_entropy != 0;

    // ---- per-unit state init: Reset -> initNewEncode (blockenc.go:77-82) ----
    {
        const uint32_t* pb = (const uint32_t*)P.predef;
        uint32_t* dstp = (uint32_t*)&S.fse[6];
        for (int i = tid; i < (int)(sizeof(KcFsePredefBlob) / 4); i += ET) dstp[i] = pb[i];
        if (tid < 6) {
            KcFseT* f = &S.fse[tid];
            f->symbolLen = 0; f->tableLog = 0; f->useRLE = 0; f->rleVal = 0; f->reUsed = 0; f->preDefined = 0; f->stLen1 = 0;
        }
        if (tid == 0) {
            for (int k = 0; k < 3; k++) { S.curIdx[k] = (uint8_t)(2 * k); S.prevIdx[k] = (uint8_t)(2 * k + 1); }
            S.huf.prevLen = 0;
            S.huf.prevLog = 0;
            S.huf.reuse = 2;  // ReusePolicyNone
        }
    }
    int opos = 0;  // bytes written to the staging area (wave-uniform, tracked by every thread)
    long long pt0 = 0;
    // (summed in LDS by the one profiling thread and added to the buffer once, at the kernel's end: a global atomic per mark would be
    // waited for by the next barrier's vmcnt(0) and show up as the phases' time)
#define PROF_MARK(i) do { if (P.prof && tid == 0) { const long long t1__ = clock64(); S.profAcc[i] += (unsigned long long)(t1__ - pt0); pt0 = t1__; } } while (0)
    if (P.prof && tid == 0) for (int i = 0; i < 24; i++) S.profAcc[i] = 0;
    if (P.prof && tid == 0) pt0 = clock64();
    // measurement builds only (-DKC_K2_FINE): the gather's sub-steps on wave 0, every mark behind a full s_waitcnt so that a step's
    // memory latency is charged to the step that waits for it (slots 40..47 of the profile buffer)
#ifdef KC_K2_FINE
    long long pf0 = 0;
#define PROF_FINE(i) do { if (P.prof && tid == 0) { __builtin_amdgcn_s_waitcnt(0); const long long t1__ = clock64(); S.profAcc[16 + (i)] += (unsigned long long)(t1__ - pf0); pf0 = t1__; } } while (0)
#define PROF_FINE0() do { if (P.prof && tid == 0) { __builtin_amdgcn_s_waitcnt(0); pf0 = clock64(); } } while (0)
#else
#define PROF_FINE(i) do { } while (0)
#define PROF_FINE0() do { } while (0)
#endif
#if defined(KC_K2_FINE) && KC_K2_FINE == 2   // second set: the phases behind the gather
#define PROF_FINB(i) do { if (P.prof && tid == 0) { __builtin_amdgcn_s_waitcnt(0); const long long t1__ = clock64(); S.profAcc[16 + (i)] += (unsigned long long)(t1__ - pf0); pf0 = t1__; } } while (0)
#define PROF_FINB0() do { if (P.prof && tid == 0) { __builtin_amdgcn_s_waitcnt(0); pf0 = clock64(); } } while (0)
#undef PROF_FINE
#undef PROF_FINE0
#define PROF_FINE(i) do { } while (0)
#define PROF_FINE0() do { } while (0)
#else
#define PROF_FINB(i) do { } while (0)
#define PROF_FINB0() do { } while (0)
#endif
    // ---- frame header (frameenc.go:25-92; encoder.go:756-772) ----
    // Streaming layout (Write ... Close, zstd/encoder.go:257-428) for units of at least one block: frame header without content
    // size or single segment, window = the encoder's, `last` only on a short final block, otherwise an empty raw last block.
    const bool streamU = UB.streamU;
    if (ulen > 0 && !JOB) {
        uint8_t hdr[16];
        const int h = kc_frame_header(hdr, ulen, P.window_size, P.single, P.crc, P.dict_id, streamU);
        if (tid == 0) for (int i = 0; i < h; i++) outp[i] = hdr[i];
        opos = h;
    }
    __syncthreads();
    if (ulen == 0 && JOB) {
        // an empty job: nothing, or (final job) the stream's empty raw last block (enc_jobs.go:96-103)
        if (tid == 0) {
            if (finalJob) { outp[0] = 0x01; outp[1] = 0x00; outp[2] = 0x00; }
            P.out_size[u] = finalJob ? 3u : 0u;
        }
        return;
    }
    if (ulen == 0) {
        // zero-length input: optional 9-byte frame (encoder.go:732-752, App. A-18)
        if (tid == 0) {
            if (P.full_zero && P.stream_mode) {
                // Close on an empty stream (encoder.go:266-329): streaming header, empty raw last block, checksum of nothing
                int h = 0;
                outp[h++] = 0x28; outp[h++] = 0xb5; outp[h++] = 0x2f; outp[h++] = 0xfd;
                uint8_t fhd = P.crc ? (1 << 2) : 0;
                const uint32_t did = P.dict_id;
                int didLen = 0;
                if (did > 0) { if (did < 256) { fhd |= 1; didLen = 1; } else if (did < (1u << 16)) { fhd |= 2; didLen = 2; } else { fhd |= 3; didLen = 4; } }
                outp[h++] = fhd;
                outp[h++] = (uint8_t)((bits_len32((uint32_t)P.window_size - 1) - 10) << 3);
                for (int i = 0; i < didLen; i++) outp[h++] = (uint8_t)(did >> (8 * i));
                outp[h++] = 0x01; outp[h++] = 0x00; outp[h++] = 0x00;
                if (P.crc) { outp[h++] = 0x99; outp[h++] = 0xe9; outp[h++] = 0xd8; outp[h++] = 0x51; }  // XXH64("") & 0xffffffff
                P.out_size[u] = (uint32_t)h;
            } else if (P.full_zero) {
                const uint8_t z[9] = {0x28, 0xb5, 0x2f, 0xfd, 0x20, 0x00, 0x01, 0x00, 0x00};
                for (int i = 0; i < 9; i++) outp[i] = z[i];
                P.out_size[u] = 9;
            } else {
                P.out_size[u] = 0;
            }
        }
        return;
    }

    int nRawDef = 0;  // blocks that went out raw with their payload left to the copy kernels
    for (int b = 0; b < nblk; b++) {
        const KcBlkMeta m = P.meta[blk0 + (uint32_t)b];
        const int blkStart = hist0 + kc_blk_begin(P.blk_start, blk0, b, bs);
        const int blkEnd = hist0 + kc_blk_end(P.blk_start, blk0, b, nblk, bs, ulen);
        const int size = blkEnd - blkStart;
        const bool last = JOB ? (b == nblk - 1 && finalJob) : (b == nblk - 1 && !UB.emptyLast);  // jobs: blk.last = len(data) == 0 && job.last
        const uint8_t* __restrict__ org = base + blkStart;
        const uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        uint8_t* __restrict__ lits = P.lits + (size_t)(blk0 + (uint32_t)b) * P.lit_stride;
        uint32_t* __restrict__ auxw = (uint32_t*)(P.aux + (size_t)(blk0 + (uint32_t)b) * P.seq_stride);
        const int nseq = (int)m.nseq;
        const int nlit = (int)m.nlit;
        uint8_t* bout = outp + opos;  // block header goes here
        if (P.rawdef != nullptr && tid == 0) P.rawdef[blk0 + (uint32_t)b].size = 0;
        // a raw block's payload: copied here, or (rawdef) left to the compaction, which takes it from the source once
        auto raw_payload = [&]() {
            if (P.rawdef != nullptr) {
                nRawDef++;
                if (tid == 0) {
                    KcRawDef r;
                    r.frame_pos = (uint32_t)(opos + 3); r.src_pos = (uint32_t)blkStart; r.size = (uint32_t)size; r.pad = 0;
                    P.rawdef[blk0 + (uint32_t)b] = r;
                }
            } else {
                wg_copy(bout + 3, org, size);
            }
        };

        // ---------- literals-only / RLE / incompressible verdicts (blockenc.go:482-503) ----------
        bool litsOnly = false;  // encodeLits path: no sequences, or saved < 16 (offsets popped by the match finder)
        if (nseq == 0) {
            litsOnly = true;  // encodeLits(b.literals, rawAllLits): with no sequences the literals are the whole block
        } else {
            const uint64_t s0 = sq[0];
            if (nseq == 1 && nlit <= 1 && (int)seq_ll(s0) == nlit && seq_of(s0) - 3u == 1u) {
                // encodeRLE(org[0], matchLen + zstdMinMatch + litLen) (blockenc.go:485-493)
                if (tid == 0) {
                    put_block_header(bout, last, 1u, seq_ml(s0) + 3u + seq_ll(s0));
                    bout[3] = org[0];
                }
                opos += 4;
                __syncthreads();
                continue;
            }
            if (m.flags & KC_BF_POP_A) litsOnly = true;  // saved < 16: popOffsets + encodeLits(org, rawAllLits)
        }
        // encodeLits (blockenc.go:337-352): extremely small blocks and rawAllLits go out as raw blocks
        // blk.dictLitEnc: first block only (reset clears it, blockenc.go:97); in a stream frame written by the synchronous nextBlock
        // form not even that one (blk.reset(nil) before the first Encode, encoder.go:371)
        const bool dictLit = b == 0 && P.dict_huf != nullptr && !(streamU && P.stream_sync);
        if (litsOnly && (rawAllLits || size < (dictLit ? 8 : 32))) {
            if (tid == 0) put_block_header(bout, last, 0u, (uint32_t)size);
            raw_payload();
            opos += 3 + size;
            __syncthreads();
            continue;
        }
        // literal source for the entropy stage: gathered literals, or the block itself on the encodeLits path
        const uint8_t* __restrict__ L = litsOnly ? org : lits;
        const int nlitE = litsOnly ? size : nlit;

        PROF_MARK(0);
        // ==================== compressed block attempt ====================
        // ---------- 1. gather literals + histogram ----------
        for (int i = tid; i < 4 * KC_HCOPIES * 256; i += ET) ((uint32_t*)S.whist)[i] = 0;
        if (!litsOnly) for (int i = tid; i < (int)((sizeof(S.codes) + sizeof(S.sbits)) / 16); i += ET) ((uint4*)&S.codes[0][0])[i] = make_uint4(0, 0, 0, 0);  // gather window
        if (!litsOnly) {  // the sequence histograms may be filled early, under the Huffman tree build (see 2.)
            for (int i = tid; i < 3 * 64; i += ET) ((uint32_t*)S.shist)[i] = 0;
            if (tid < 3) S.smax[tid] = 0;
        }
        __syncthreads();
        if (litsOnly) {
            for (int k = tid; k < size; k += ET) atomicAdd(&S.whist[wv * KC_HCOPIES + (lane & (KC_HCOPIES - 1))][org[k]], 1u);
        } else
        {
            // Literal gather = compaction of the literal runs of all sequences into `lits` (the match finder stores no literal
            // bytes).  Round 5: every WAVE gathers its own contiguous quarter of the sequence list, without a workgroup barrier:
            //   pass A  the wave sums litLen / litLen + matchLen of its quarter (and, while the sequences are in registers, adds
            //           their codes to the three sequence histograms: genCodes, blockenc.go:831-893) -> one barrier -> the
            //           quarter's first literal offset and first source position;
            //   pass B  64 sequences per step: a wave scan gives every run's place, the runs are ORed into the wave's own window
            //           of LDS (ds_or_b64; a global byte store per lane is a 64-address memory instruction), complete 16-byte
            //           words are flushed to `lits` and counted into the wave's histogram copy, the open word stays as the head
            //           of the next step's window.  The next step's sequences are loaded under the current step's work.
            // Runs longer than LONG_RUN are copied by the whole wave, 4 bytes per lane.  A quarter's first and last 16-byte word
            // may be shared with the neighbouring wave's: their bytes are stored one by one.
            uint8_t* __restrict__ tile = &S.codes[0][0];  // codes + sbits: unused until the sequence phase
            constexpr int TWB = 2304;   // bytes of a wave's window region: 2048 flushed per pass + the open word + slack
            constexpr int TWC = 2048;
            static_assert(sizeof(S.codes) + sizeof(S.sbits) >= 4 * TWB, "four gather windows");
            uint8_t* __restrict__ tw = tile + wv * TWB;
            const uint8_t* __restrict__ bsrc = base + blkStart;
            const int per = (((nseq + 3) >> 2) + 63) & ~63;  // sequences per wave: whole steps of 64
            const int q0 = wv * per < nseq ? wv * per : nseq;
            const int q1 = q0 + per < nseq ? q0 + per : nseq;
            // ---- pass A ----
            PROF_FINE0();
            {
                uint64_t acc = 0;  // lo32: literal bytes, hi32: source bytes
                uint32_t mll = 0, mof = 0, mml = 0;
                for (int i0 = q0 + lane; i0 < q1; i0 += 8 * 64) {
                    uint64_t sv[8];  // eight loads in flight: a round trip to memory is what this pass waits for
#pragma unroll
                    for (int r = 0; r < 8; r++) sv[r] = i0 + 64 * r < q1 ? sq[i0 + 64 * r] : 0ull;
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        if (i0 + 64 * r < q1) {
                            const uint32_t ll = seq_ll(sv[r]), ml = seq_ml(sv[r]), of = seq_of(sv[r]);
                            acc += (uint64_t)ll | ((uint64_t)(ll + ml + 3u) << 32);
                            const uint32_t cl = kc_ll_code(ll), co = kc_of_code(of), cm = kc_ml_code(ml);
                            atomicAdd(&S.shist[0][cl], 1u);
                            atomicAdd(&S.shist[1][co], 1u);
                            atomicAdd(&S.shist[2][cm], 1u);
                            mll = cl > mll ? cl : mll;
                            mof = co > mof ? co : mof;
                            mml = cm > mml ? cm : mml;
                        }
                    }
                }
                mll = wave_reduce_max(mll); mof = wave_reduce_max(mof); mml = wave_reduce_max(mml);
                const uint64_t tot = wave_incl_scan64(acc, lane);
                if (lane == 63) S.wsum[wv] = tot;
                if (lane == 0) { atomicMax(&S.smax[0], mll); atomicMax(&S.smax[1], mof); atomicMax(&S.smax[2], mml); }
            }
            PROF_FINE(0);
            __syncthreads();
            PROF_FINE(1);
            uint64_t run = 0;  // lo32: literal bytes in front of the wave's next step, hi32: source bytes (wave-uniform)
            for (int k = 0; k < wv; k++) run += S.wsum[k];
            const uint32_t firstLo = (uint32_t)run;        // the quarter's first literal offset
            const uint32_t fw = firstLo & ~15u;            // its first 16-byte word: the bytes below firstLo are the previous wave's
            const uint32_t hs = firstLo & 15u;
            // ---- pass B ----
            // A global round trip costs this kernel ~4 000 shader clocks (measured, -DKC_K2_FINE: 5 000 clocks per step for the literal
            // loads + ORs, 1 000 for the scan, 1 100 for flush + histogram, 300 for carry + zero), and the step's literal loads are one.
            // Both ways of hiding it were built and measured, and both LOST on this kernel, which sits at its 128-VGPR ceiling: the
            // next step's 32 bytes per lane held in registers while this step is processed (19.9 vs 19.1 ms: more spills than hidden
            // latency), and touching one byte at each end of the next step's runs a step ahead (19.6 ms: the touch is waited for too).
            // The step's sequences are loaded one step ahead.
            struct GStep { uint32_t ll, lo, sp; uint64_t tot, longMask; };
            auto place = [&](uint64_t sx, bool have, uint64_t at, GStep& g) {  // the step's runs: a wave scan from the literal / source offsets `at`
                uint32_t ll = 0, adv = 0;
                if (have) { ll = seq_ll(sx); adv = ll + seq_ml(sx) + 3u; }
                const uint64_t v = (uint64_t)ll | ((uint64_t)adv << 32);
                const uint64_t inc = wave_incl_scan64(v, lane);
                g.tot = bcast64(inc, 63);
                const uint64_t ex = inc - v + at;
                g.ll = ll; g.lo = (uint32_t)ex; g.sp = (uint32_t)(ex >> 32);
                g.longMask = ballot64(ll > LONG_RUN);
            };
            auto fetch = [&](const GStep& g, uint64_t (&vv)[4]) {  // a run of up to LONG_RUN (32) bytes: four 8-byte loads
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const uint32_t k = 8u * (uint32_t)kk;
                    vv[kk] = 0;
                    if (g.ll <= LONG_RUN && k < g.ll) {
                        const uint32_t n8 = g.ll - k < 8u ? g.ll - k : 8u;
                        if ((int)(g.sp + k) + 8 <= size) vv[kk] = ld64(bsrc + g.sp + k);
                        else { uint64_t t = 0; for (uint32_t q = 0; q < n8; q++) t |= (uint64_t)bsrc[g.sp + k + q] << (8 * q); vv[kk] = t; }
                    }
                }
            };
            GStep cur;
            uint64_t vv[4];
            uint64_t sq1 = q0 + lane < q1 ? sq[q0 + lane] : 0ull;
            for (int t0 = q0; t0 < q1; t0 += 64) {
                const int i = t0 + lane;
                place(sq1, i < q1, run, cur);
                sq1 = i + 64 < q1 ? sq[i + 64] : 0ull;
                PROF_FINE(2);
                fetch(cur, vv);
                const uint32_t ll = cur.ll, lo = cur.lo, sp = cur.sp;
                const uint64_t tot = cur.tot, longMask = cur.longMask;
                const bool mine = ll > 0 && ll <= LONG_RUN;
                const uint32_t begLo = (uint32_t)run;
                const uint32_t endLo = begLo + (uint32_t)tot;  // literal bytes after this step
                for (uint32_t winBase = begLo & ~15u; winBase < endLo; winBase += TWC) {
                    if (mine && lo < winBase + TWC && lo + ll > winBase) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const uint32_t k = 8u * (uint32_t)kk;
                            if (k < ll) {
                                const uint32_t n8 = ll - k < 8u ? ll - k : 8u;
                                const uint64_t x = n8 < 8u ? vv[kk] & ((1ull << (8u * n8)) - 1ull) : vv[kk];
                                const uint32_t o = lo + k - winBase;  // wraps for bytes in front of the window
                                if (lo + k >= winBase && o + n8 <= (uint32_t)TWC) {
                                    // the window is zero where nothing has been written yet: up to 8 bytes at any alignment are
                                    // one or two 64-bit ORs (ds_or_b64) instead of 8 byte writes
                                    unsigned long long* pq = (unsigned long long*)(tw + (o & ~7u));
                                    const uint32_t sh = (o & 7u) * 8u;
                                    atomicOr(pq, (unsigned long long)(x << sh));
                                    if (sh != 0u && (x >> (64u - sh)) != 0ull) atomicOr(pq + 1, (unsigned long long)(x >> (64u - sh)));
                                } else {  // run cut by a window edge (only when a step holds more than TWC literal bytes)
                                    for (uint32_t q = 0; q < n8; q++) {
                                        const uint32_t oo = lo + k + q - winBase;
                                        if (oo < (uint32_t)TWC) atomicOr((uint32_t*)(tw + (oo & ~3u)), (uint32_t)((x >> (8u * q)) & 0xFFu) << (8u * (oo & 3u)));
                                    }
                                }
                            }
                        }
                    }
                    for (uint64_t lm = longMask; lm != 0ull; lm &= lm - 1ull) {  // wave-uniform: the long runs, one after the other
                        const int e = ctz64(lm);
                        const uint32_t lsp = bcast32(sp, e), llo = bcast32(lo, e), lln = bcast32(ll, e);
                        const uint32_t o0 = llo > winBase ? llo : winBase;
                        const uint32_t o1 = llo + lln < winBase + TWC ? llo + lln : winBase + TWC;
                        for (uint32_t o = o0 + 4u * (uint32_t)lane; o < o1; o += 4u * 64u) {
                            const uint32_t n4 = o1 - o < 4u ? o1 - o : 4u;
                            const uint32_t si = lsp + (o - llo);
                            uint32_t x;
                            if ((int)si + 4 <= size) x = ld32(bsrc + si);
                            else { x = 0; for (uint32_t q = 0; q < n4; q++) x |= (uint32_t)bsrc[si + q] << (8u * q); }
                            if (n4 < 4u) x &= (1u << (8u * n4)) - 1u;
                            const uint32_t oo = o - winBase;
                            unsigned long long* pq = (unsigned long long*)(tw + (oo & ~7u));
                            const uint32_t sh = (oo & 7u) * 8u;
                            atomicOr(pq, (unsigned long long)x << sh);
                            if (sh > 32u && (x >> (64u - sh)) != 0u) atomicOr(pq + 1, (unsigned long long)(x >> (64u - sh)));
                        }
                    }
                    KC_WAVE_SYNC();
                    PROF_FINE(3);
                    const uint32_t wEnd = endLo < winBase + TWC ? endLo : winBase + TWC;
                    const uint32_t nfull = (wEnd - winBase) >> 4;
                    const bool shared0 = hs != 0u && winBase == fw && nfull > 0u;  // word 0 holds bytes of the previous wave's quarter
                    if (shared0 && (uint32_t)lane >= hs && lane < 16) { const uint8_t c = tw[lane]; lits[fw + lane] = c; atomicAdd(&S.whist[wv * KC_HCOPIES][c], 1u); }
                    for (uint32_t w = (shared0 ? 1u : 0u) + (uint32_t)lane; w < nfull; w += 64u) ((uint4*)(lits + winBase))[w] = ((const uint4*)tw)[w];
                    // literal histogram from the flushed words, four bytes per lane and step
                    for (uint32_t d = (shared0 ? 4u : 0u) + (uint32_t)lane; d < 4u * nfull; d += 64u) {
                        const uint32_t x = ((const uint32_t*)tw)[d];
                        uint32_t* const hc = S.whist[wv * KC_HCOPIES + (lane & (KC_HCOPIES - 1))];
                        atomicAdd(&hc[x & 0xFFu], 1u);
                        atomicAdd(&hc[(x >> 8) & 0xFFu], 1u);
                        atomicAdd(&hc[(x >> 16) & 0xFFu], 1u);
                        atomicAdd(&hc[x >> 24], 1u);
                    }
                    const uint32_t tail = (wEnd - winBase) & 15u;  // only the last window of a step has a tail
                    uint8_t carry = 0;
                    if (nfull > 0u && lane < (int)tail) carry = tw[(nfull << 4) + lane];
                    KC_WAVE_SYNC();
                    PROF_FINE(4);
                    if (nfull > 0u) {  // the flushed words become zero again; the tail moves to the window's first word
                        for (uint32_t w = 1u + (uint32_t)lane; w <= nfull; w += 64u) ((uint4*)tw)[w] = make_uint4(0, 0, 0, 0);
                        if (lane < 16) tw[lane] = carry;
                    }
                    KC_WAVE_SYNC();
                    PROF_FINE(5);
                }
                run += tot;
            }
            // the quarter's last bytes still in LDS (less than one 16-byte word; if the quarter never left its first word: not the
            // previous wave's part of it)
            {
                const uint32_t total = (uint32_t)run;
                const uint32_t rem = total & 15u;
                const uint32_t lowT = ((total & ~15u) == fw) ? hs : 0u;
                if ((uint32_t)lane >= lowT && (uint32_t)lane < rem) { const uint8_t c = tw[lane]; lits[(total & ~15u) + lane] = c; atomicAdd(&S.whist[wv * KC_HCOPIES][c], 1u); }
            }
            // trailing literals after the last sequence
            {
                const uint32_t len = m.extra_lits, spos = (uint32_t)size - len, lo = (uint32_t)nlit - len;
                for (uint32_t k = tid; k < len; k += ET) {
                    const uint8_t c = bsrc[spos + k];
                    lits[lo + k] = c;
                    atomicAdd(&S.whist[wv * KC_HCOPIES][c], 1u);
                }
            }
            PROF_FINE(6);
        }
        __syncthreads();
        PROF_FINE(7);
        {
            // reduce histogram (huff0 countSimple, compress.go:351): maxCount, symbolLen
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < 4 * KC_HCOPIES; k++) c += S.whist[k][tid];
            S.cnt[tid] = c;
            const uint32_t wm = wave_reduce_max(c);
            const uint32_t wl = wave_reduce_max(c ? (uint32_t)tid + 1u : 0u);
            if (lane == 0) { S.wtot[wv] = wm; S.wsum[wv] = wl; }
        }
        __syncthreads();  // also makes the gathered literals visible workgroup-wide

        PROF_MARK(1);
        // Sequence-side preparation that does not depend on the literals.  When a Huffman table is built for this block the
        // two long single-lane phases of huff0 (tree build, weight-table FSE) run on wave 0, and waves 1-3 use that time:
        // histograms under the tree build, normalizeCount + buildCTable under the table description.
        auto seq_build = [&](int k) {  // whole wave: normalizeCount on lane 0, buildCTable on all lanes
            KcFseT* f = &S.fse[S.curIdx[k]];
            int doBuild = 0;
            if (lane == 0) {
                const int maxSym = (int)S.smax[k];
                uint32_t maxCount = 0;
                for (int i = 0; i <= maxSym; i++) if (S.shist[k][i] > maxCount) maxCount = S.shist[k][i];
                f->symbolLen = (uint16_t)(maxSym + 1);  // HistogramFinished
                if (!f->reUsed) {  // normalizeCount returns early for reused encoders (fse_encoder.go:260)
                    f->tableLog = fse_optimal_table_log(nseq, f->symbolLen);
                    f->stLen1 = 0;
                    if ((int)maxCount == nseq) {
                        f->useRLE = 1;
                    } else {
                        f->useRLE = 0;
                        if (fse_normalize_core(S.shist[k], f->norm, f->symbolLen, nseq, f->tableLog)) doBuild = 1;
                        else atomicExch(P.err_flag, 2u);
                    }
                }
            }
            doBuild = __shfl(doBuild, 0, 64);
            if (doBuild) {
                if (!fse_build_wave(f->norm, f->symbolLen, f->tableLog, S.tsym[k], S.cumul[k], S.posx[k], f->st, f->dnb, f->dfs, lane)) {
                    if (lane == 0) atomicExch(P.err_flag, 2u);
                }
            }
        };
        bool seqHistDone = !litsOnly, seqBuildDone = false;  // workgroup-uniform (the histograms: filled by the gather's pass A)
        // ---------- 2. huff0.compress decisions (compress.go:43-163) ----------
        const bool wantHuf = litsOnly ? (nlitE > 16) : (!noEntropy && nlitE > 16);  // encodeLits ignores noEntropy (encoder.go:795)
        const bool four = nlitE >= 1024;
        // litMode: 0 raw literals, 1 RLE literals, 2 compressed (new table), 3 compressed (reused table)
        if (tid == 0) {
            int litMode = 0;
            uint32_t maxCount = 0, symbolLen = 0;
            for (int k = 0; k < 4; k++) { if (S.wtot[k] > maxCount) maxCount = S.wtot[k]; if ((uint32_t)S.wsum[k] > symbolLen) symbolLen = (uint32_t)S.wsum[k]; }
            S.ivar[IV_SYMLEN] = (int)symbolLen;
            S.ivar[IV_MAXCNT] = (int)maxCount;
            S.ivar[IV_OK] = 0;
            if (dictLit) {  // TransferCTable(dictLitEnc) + Reuse = Allow, before the size checks (blockenc.go:358-362, 518-522)
                const uint16_t* dv = (const uint16_t*)P.dict_huf;
                const uint8_t* dn = P.dict_huf + 512;
                for (int k = 0; k < P.dict_huf_len; k++) { S.huf.prev.val[k] = dv[k]; S.huf.prev.nb[k] = dn[k]; }
                S.huf.prevLen = P.dict_huf_len;
                S.huf.prevLog = (uint8_t)P.dict_huf_log;
                S.huf.reuse = 0;
            }
            if (wantHuf) {
                if (S.huf.reuse == 2) S.huf.prevLen = 0;  // ReusePolicyNone nukes prevTable (compress.go:45)
                if ((int)maxCount >= nlitE) litMode = (nlitE == 1) ? 0 : 1;              // single symbol -> RLE
                else if (maxCount == 1 || (int)maxCount < (nlitE >> 7)) litMode = 0;    // ErrIncompressible
                else S.ivar[IV_OK] = 1;                                              // go on and build a table
            }
            S.ivar[IV_LITMODE] = litMode;
        }
        __syncthreads();
        int litSecLen = 0;  // bytes of the literals section (header included), wave-uniform
        bool single = false;
        if (S.ivar[IV_OK]) {
            const int symbolLen = S.ivar[IV_SYMLEN];
            // canReuse (countSimple): every present symbol has a code in prevTable
            {
                bool bad = false;
                if (S.huf.prevLen > 0) { if (S.cnt[tid] != 0 && (tid >= S.huf.prevLen || S.huf.prev.nb[tid] == 0)) bad = true; }
                else bad = true;
                const int anyBad = __syncthreads_or(bad ? 1 : 0);
                if (tid == 0) S.ivar[IV_CANREUSE] = anyBad ? 0 : 1;
            }
            PROF_MARK(2);
            // huffSort as a parallel rank: stable by (count desc, symbol asc)
            if (tid < symbolLen) {
                const uint32_t c = S.cnt[tid];
                int rank = 0;
                for (int j = 0; j < symbolLen; j++) {
                    const uint32_t cj = S.cnt[j];
                    rank += (cj > c || (cj == c && j < tid)) ? 1 : 0;
                }
                S.nodes.count[rank + 1] = c;
                S.nodes.symbol[rank + 1] = (uint8_t)tid;
            }
            __syncthreads();
            if (wv == 0) {  // buildCTable on wave 0 (only the two-queue merge is single-lane), waves 1-3 fill the sequence histograms
                const uint8_t tl = huf_build_wave(&S.nodes, &S.cur, &S.hwt, symbolLen, nlitE, lane);
                if (lane == 0) {
                    if (tl == 0xFF) atomicExch(P.err_flag, 1u);
                    S.ivar[IV_TABLOG] = tl;
                }
            }
            seqHistDone = !litsOnly;
            __syncthreads();
            PROF_MARK(3);
            // estimateSize for old/new tables (huff0.go:308) — block reduction of nBits*count
            {
                const uint32_t c = tid < symbolLen ? S.cnt[tid] : 0u;
                const uint32_t nn = wave_reduce_sum(c * (uint32_t)(tid < symbolLen ? S.cur.nb[tid] : 0));
                const uint32_t no = wave_reduce_sum(c * (uint32_t)((tid < symbolLen && tid < S.huf.prevLen) ? S.huf.prev.nb[tid] : 0));
                if (lane == 0) { S.wtot[wv] = nn; S.wsum[wv] = no; }
            }
            __syncthreads();
            if (wv == 0) {  // wave 0: reuse decision + table description (cTable.write) as a wave; waves 1-3 build the sequence tables
                const uint32_t newBits = 7 + S.wtot[0] + S.wtot[1] + S.wtot[2] + S.wtot[3];
                const uint32_t oldBits = 7 + (uint32_t)(S.wsum[0] + S.wsum[1] + S.wsum[2] + S.wsum[3]);
                const int newSize = (int)(newBits >> 3), oldSize = (int)(oldBits >> 3);
                const int wantSize = nlitE - (nlitE >> 4);  // WantLogLess = 4 (blockenc.go:72)
                int usePrev = 0;
                // ReusePolicyAllow && canReuse: hSize == len(s.Out) == 0 at this point (App. A-12)
                if (S.huf.reuse == 0 && S.ivar[IV_CANREUSE]) {
                    if (oldSize <= 0 + newSize || 0 + 12 >= wantSize) usePrev = 1;
                }
                int descLen = 0;
                if (!usePrev) {  // wave-uniform
                    huf_weights_wave(&S.cur, symbolLen, (uint8_t)S.ivar[IV_TABLOG], S.weights, &S.wfse, lane);
                    descLen = huf_write_table(symbolLen, S.weights, &S.wfse, S.tdesc, (int)sizeof(S.tdesc), lane);
                }
                if (lane == 0) {
                    S.ivar[IV_USEPREV] = usePrev;
                    S.ivar[IV_DESCLEN] = descLen;  // -1: cTable.write failed (ErrIncompressible)
                }
            }
            if (wv >= 1 && seqHistDone) seq_build(wv - 1);
            seqBuildDone = seqHistDone;
            __syncthreads();
            const bool usePrev = S.ivar[IV_USEPREV] != 0;
            const KcHufTable* T = usePrev ? &S.huf.prev : &S.cur;
            const int descLen = S.ivar[IV_DESCLEN];
            PROF_MARK(4);
            // ---- size pass: exact stream sizes with table T ----
            const int nstreams = four ? 4 : 1;
            const int segSize = four ? (nlitE + 3) / 4 : nlitE;
            int segStart = 0, segLen = 0, chunk = 1;
            uint32_t laneBits = 0, laneOff = 0, streamBits = 0;
            if (wv < nstreams && descLen >= 0) {
                segStart = wv * segSize;
                segLen = nlitE - segStart;
                if (segLen > segSize) segLen = segSize;
                if (segLen < 0) segLen = 0;
                chunk = (segLen + 63) / 64;
                if (chunk < 1) chunk = 1;
                PROF_FINB0();
                laneBits = huf_lane_bits(L + segStart, segLen, T, lane, chunk);
                PROF_FINB(0);
                const uint32_t inc = wave_incl_scan(laneBits, lane);
                laneOff = inc - laneBits;
                streamBits = (uint32_t)__shfl((int)inc, 63, 64);
            }
            if (lane == 0) S.wtot[wv] = (wv < nstreams) ? ((streamBits + 1 + 7) >> 3) : 0u;  // + end mark, byte aligned
            __syncthreads();
            if (tid == 0) {
                int litMode = 0;
                int wantSize = nlitE - (nlitE >> 4);
                int dataLen = (int)(S.wtot[0] + S.wtot[1] + S.wtot[2] + S.wtot[3]) + (four ? 6 : 0);
                bool ok = descLen >= 0;
                if (ok && four) for (int k = 0; k < 4; k++) if (S.wtot[k] > 65535u) ok = false;  // jump table limit (compress.go:288)
                int outLen = dataLen + (usePrev ? 0 : descLen);
                if (ok && outLen >= wantSize) ok = false;
                if (ok && !usePrev) {
                    // Move current table into previous (compress.go:160): happens before blockenc's own checks.
                    for (int k = 0; k < symbolLen; k++) { S.huf.prev.val[k] = S.cur.val[k]; S.huf.prev.nb[k] = S.cur.nb[k]; }
                    S.huf.prevLen = symbolLen;
                    S.huf.prevLog = (uint8_t)S.ivar[IV_TABLOG];
                }
                if (ok && outLen + 5 > nlitE) {
                    // close call: compare with raw including header sizes (blockenc.go:534-544; encodeLits :374-381)
                    const int szRaw = litsOnly ? 0 : lit_header_size1(nlitE);
                    const int szComp = lit_header_size2(outLen, nlitE);
                    if (outLen + szComp >= nlitE + szRaw) ok = false;
                }
                if (ok) {
                    litMode = usePrev ? 3 : 2;
                    S.huf.reuse = 0;  // b.litEnc.Reuse = ReusePolicyAllow (blockenc.go:588)
                }
                S.ivar[IV_LITMODE] = litMode;
                S.ivar[IV_DATALEN] = outLen;
            }
            __syncthreads();
            PROF_MARK(5);
            const int litMode = S.ivar[IV_LITMODE];
            if (litMode >= 2) {
                single = !four;
                const int outLen = S.ivar[IV_DATALEN];
                const int hsz = lit_header_size2(outLen, nlitE);
                // stream byte offsets inside the section payload
                const int tabLen = usePrev ? 0 : descLen;
                const uint32_t sb0 = S.wtot[0], sb1 = S.wtot[1], sb2 = S.wtot[2];
                uint32_t myByteOff = (uint32_t)(tabLen + (four ? 6 : 0));
                if (wv >= 1) myByteOff += sb0;
                if (wv >= 2) myByteOff += sb1;
                if (wv >= 3) myByteOff += sb2;
                // Emit streams into the zeroed, word-aligned staging area (aux scratch), each stream
                // at its final byte offset relative to the payload start.
                const int payloadWords = (outLen + 3 + 4) >> 2;
                PROF_FINB0();
                for (int i = tid; i < payloadWords; i += ET) auxw[i] = 0;
                __syncthreads();
                PROF_FINB(1);
                if (wv < nstreams) {
                    const uint64_t bitBase = (uint64_t)myByteOff * 8 + laneOff;
                    huf_lane_emit(L + segStart, segLen, T, lane, chunk, auxw, bitBase);
                    if (lane == 63) or_bits(auxw, (uint64_t)myByteOff * 8 + streamBits, 1, 1);  // end mark
                }
                PROF_FINB(2);
                __syncthreads();
                PROF_FINB(3);
                // copy payload to its final place and write the byte-granular headers
                uint8_t* lsec = bout + 3;
                wg_copy(lsec + hsz + tabLen + (four ? 6 : 0), (const uint8_t*)auxw + tabLen + (four ? 6 : 0), outLen - tabLen - (four ? 6 : 0));
                if (tid == 0) {
                    put_lit_header2(lsec, litMode == 3 ? 3u : 2u, outLen, nlitE, single);
                    for (int k = 0; k < tabLen; k++) lsec[hsz + k] = S.tdesc[k];
                    if (four) {
                        uint8_t* jt = lsec + hsz + tabLen;
                        jt[0] = (uint8_t)sb0; jt[1] = (uint8_t)(sb0 >> 8);
                        jt[2] = (uint8_t)sb1; jt[3] = (uint8_t)(sb1 >> 8);
                        jt[4] = (uint8_t)sb2; jt[5] = (uint8_t)(sb2 >> 8);
                    }
                }
                litSecLen = hsz + outLen;
                PROF_FINB(4);
            }
        }
        if (litsOnly) {
            // ---------- encodeLits block assembly (blockenc.go:382-427) ----------
            const int litMode = S.ivar[IV_LITMODE];
            if (litMode == 0) {  // ErrIncompressible -> raw block
                if (tid == 0) put_block_header(bout, last, 0u, (uint32_t)size);
                raw_payload();
                opos += 3 + size;
            } else if (litMode == 1) {  // ErrUseRLE -> RLE block
                if (tid == 0) { put_block_header(bout, last, 1u, (uint32_t)size); bout[3] = org[0]; }
                opos += 4;
            } else {  // compressed block: literals section + "no sequences" byte
                if (tid == 0) { put_block_header(bout, last, 2u, (uint32_t)(litSecLen + 1)); bout[3 + litSecLen] = 0; }
                opos += 3 + litSecLen + 1;
            }
            __syncthreads();
            PROF_MARK(6);
            continue;
        }
        {
            const int litMode = S.ivar[IV_LITMODE];
            if (litMode == 0) {  // raw literals (blockenc.go:546-552)
                const int hsz = lit_header_size1(nlit);
                if (tid == 0) put_lit_header1(bout + 3, 0u, nlit);
                wg_copy(bout + 3 + hsz, lits, nlit);
                litSecLen = hsz + nlit;
            } else if (litMode == 1) {  // RLE literals (blockenc.go:553-560)
                const int hsz = lit_header_size1(nlit);
                if (tid == 0) { put_lit_header1(bout + 3, 1u, nlit); bout[3 + hsz] = lits[0]; }
                litSecLen = hsz + 1;
            }
        }
        __syncthreads();

        PROF_MARK(6);
        // ---------- 3. sequence codes + histograms (genCodes, blockenc.go:831-893): filled by the gather's pass A ----------
        PROF_MARK(7);
        // ---------- 4. normalizeCount (one lane) + buildCTable (whole wave) for the three "cur" encoders, one wave each ----------
        if (!seqBuildDone && wv < 3) seq_build(wv);
        __syncthreads();
        PROF_MARK(8);
        // ---------- 5. mode choice, mode byte, NCount headers (blockenc.go:633-722) ----------
        // 5a. approxSize of the new / predefined / previous encoder for each stream: one wave per stream, one lane per symbol
        if (wv < 3) {
            const int k = wv;
            const KcFseT* cur = &S.fse[S.curIdx[k]];
            if (!cur->useRLE) {
                const uint32_t* hist = S.shist[k];
                const int histLen = cur->symbolLen;
                const uint32_t v0 = wave_approx_size(cur, hist, histLen, lane);
                const uint32_t v1 = wave_approx_size(&S.fse[6 + k], hist, histLen, lane);
                const uint32_t v2 = wave_approx_size(&S.fse[S.prevIdx[k]], hist, histLen, lane);
                if (lane == 0) { S.asz[k][0] = v0; S.asz[k][1] = v1; S.asz[k][2] = v2; }
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint8_t mode = 0;
            const uint32_t firstCodes[3] = {kc_ll_code(seq_ll(sq[0])), kc_of_code(seq_of(sq[0])), kc_ml_code(seq_ml(sq[0]))};
            const int shifts[3] = {6, 4, 2};
            for (int k = 0; k < 3; k++) {
                KcFseT* cur = &S.fse[S.curIdx[k]];
                int use = S.curIdx[k];
                uint32_t mk;
                if (cur->useRLE) {
                    mk = 1;  // compModeRLE; setRLE (fse_encoder.go:208)
                    const uint32_t v = firstCodes[k];
                    cur->tableLog = 0;
                    cur->stLen1 = 1;
                    cur->dfs[v] = 0; cur->dnb[v] = 0; cur->outBits[v] = 0;
                    cur->st[0] = 0;  // cState.init pins state 0 for a 1-entry table (fse_encoder.go:685-690)
                    cur->rleVal = (uint8_t)v;
                } else {
                    uint32_t nSize = S.asz[k][0] + fse_max_header_size(cur);
                    const uint32_t predefSize = S.asz[k][1];
                    const uint32_t prevSize = S.asz[k][2];
                    nSize = nSize + ((nSize + 2 * 8 * 16) >> 4);
                    if (predefSize <= prevSize && predefSize <= nSize) { mk = 0; use = 6 + k; }
                    else if (prevSize <= nSize) { mk = 3; use = S.prevIdx[k]; }
                    else mk = 2;
                }
                S.useIdx[k] = (uint8_t)use;
                mode |= (uint8_t)(mk << shifts[k]);
            }
            S.seqMode = mode;
        }
        __syncthreads();
        // 5b. NCount of the encoders built this block (one lane per stream) and setBits (one lane per code)
        if (wv < 3) {
            const int k = wv;
            KcFseT* f = &S.fse[S.useIdx[k]];
            if (lane == 0) {
                int n = 0;
                if (f->useRLE) { S.nc[k][0] = f->rleVal; n = 1; }
                else if (!(f->preDefined || f->reUsed)) {
                    n = fse_write_ncount(f->norm, f->symbolLen, f->tableLog, S.nc[k]);
                    if (n < 0) { atomicExch(P.err_flag, 3u); n = 0; }
                }
                S.ncLen[k] = n;
            }
            // setBits (fse_encoder.go:225): extra-bit counts per code for encoders built this block
            if (!(f->reUsed || f->preDefined)) {
                if (f->useRLE) {
                    if (lane == 0) { const uint32_t v = f->rleVal; f->outBits[v] = (uint8_t)(k == 0 ? kc_ll_bits(v) : (k == 1 ? v : kc_ml_bits(v))); }
                } else if (lane < (int)f->symbolLen) {
                    f->outBits[lane] = (uint8_t)(k == 0 ? kc_ll_bits((uint32_t)lane) : (k == 1 ? (uint32_t)lane : kc_ml_bits((uint32_t)lane)));
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint8_t* h = S.seqhdr;
            int hp = 0;
            if (nseq < 128) h[hp++] = (uint8_t)nseq;
            else if (nseq < 0x7f00) { h[hp++] = (uint8_t)(128 + (uint8_t)(nseq >> 8)); h[hp++] = (uint8_t)nseq; }
            else { const int n = nseq - 0x7f00; h[hp++] = 255; h[hp++] = (uint8_t)n; h[hp++] = (uint8_t)(n >> 8); }
            h[hp++] = S.seqMode;
            for (int k = 0; k < 3; k++)  // writeCount order: LL, OF, ML
                for (int i = 0; i < S.ncLen[k]; i++) h[hp++] = S.nc[k][i];
            S.seqhdrLen = hp;
        }
        __syncthreads();
        const KcFseT* E[3] = {&S.fse[S.useIdx[0]], &S.fse[S.useIdx[1]], &S.fse[S.useIdx[2]]};
        const int seqhdrLen = S.seqhdrLen;
        if (tid < 192) {  // the chains' per-code constants of this block's three encoders (an RLE encoder: all zero, the state stays 0)
            const KcFseT* f = &S.fse[S.useIdx[tid >> 6]];
            const int c = tid & 63;
            S.cpk[tid >> 6][c] = c < (int)f->symbolLen || f->stLen1 ? ((f->dnb[c] << 12) | ((uint32_t)(int32_t)f->dfs[c] & 0xFFFu)) : 0u;
        }
        PROF_MARK(9);
        // ---------- 6. FSE state chains + bit packing, chunk by chunk from the last sequence ----------
        // Stream element order (blockenc.go:725-807): element 0 = last sequence (extra bits only, states
        // initialised), elements 1..n-1 = sequences n-2..0 (OF, ML, LL state bits then extra bits with LL
        // lowest), element n = final states ML, OF, LL + end mark.
        const int seqBudgetBytes = size - litSecLen - seqhdrLen;  // stream must be smaller than this to beat raw
        uint32_t* __restrict__ sw = (uint32_t*)lits;               // literals are consumed: reuse as bit staging
        uint64_t bitRun = 0;
        uint32_t carry = 0;  // the partially filled staging word at bitRun >> 5 (workgroup-uniform); it is stored once it is complete
        bool overflow = seqBudgetBytes <= 0;
        __syncthreads();
        for (int hiSeq = nseq; hiSeq > 0 && !overflow; hiSeq -= SEQ_CHUNK) {
            const int loSeq = hiSeq - SEQ_CHUNK > 0 ? hiSeq - SEQ_CHUNK : 0;
            const int cn = hiSeq - loSeq;
            const int jb = hiSeq == nseq ? 1 : 0;  // very first stream element: cState.init, no state bits
            const int nrem = cn - jb;              // elements the chains encode: chain position x = j - jb
            const int slot0 = CH_PAD - jb;         // staged element j <-> index slot0 + j
            PROF_FINB0();
            // stage codes of sequences [loSeq, hiSeq), stored in stream order: element j <-> seq hiSeq-1-j; the positions up to the
            // end of the last 16-element segment repeat the last code (a valid symbol: their chain steps run and are ignored)
            constexpr int R = SEQ_CHUNK / ET;  // all loads of the chunk are issued before the first code is computed
            uint64_t sv[R];                     // (KC_PACK_REUSE: kept across the chain phase for the bit packing below)
            {
                const int cpad = jb + ((nrem + CH_SEG - 1) & ~(CH_SEG - 1));  // (up to SEQ_CHUNK + 1: the rows have the room)
#pragma unroll
                for (int r = 0; r < R; r++) { const int j = tid + r * ET; sv[r] = j < cn ? sq[hiSeq - 1 - j] : 0ull; }
                const uint64_t svl = tid < cpad - cn ? sq[loSeq] : 0ull;  // the chunk's last element (stream order)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int j = tid + r * ET;
                    if (j < cn) {
                        S.codes[0][slot0 + j] = (uint8_t)kc_ll_code(seq_ll(sv[r]));
                        S.codes[1][slot0 + j] = (uint8_t)kc_of_code(seq_of(sv[r]));
                        S.codes[2][slot0 + j] = (uint8_t)kc_ml_code(seq_ml(sv[r]));
                    }
                }
                if (tid < cpad - cn) {
                    S.codes[0][slot0 + cn + tid] = (uint8_t)kc_ll_code(seq_ll(svl));
                    S.codes[1][slot0 + cn + tid] = (uint8_t)kc_of_code(seq_of(svl));
                    S.codes[2][slot0 + cn + tid] = (uint8_t)kc_ml_code(seq_ml(svl));
                }
            }
            __syncthreads();
            PROF_FINB(5);
            PROF_MARK(10);
            if (wv < 3) {
                // tANS state chains, one wave per stream (LL, OF, ML).  The chain st -> stateTable[...] -> st is serial, but
                // chains started from different states coalesce after a few symbols, so the chunk is cut into segments of 16:
                // lane i warms up over the KC_CHAIN_WARM symbols before its segment from an arbitrary valid state (the first
                // lanes: from the chunk's true entry state), encodes its segment, and the wave then VERIFIES that every lane's
                // assumed entry state equals its predecessor's exit state.  A lane that guessed wrong re-encodes its segment
                // from the proven state, so the result is exactly the sequential chain of blockenc.go:757-787.
                // Round 5: a segment lives in registers — its 16 codes arrive with one aligned 16-byte LDS read, each symbol's
                // two constants with one lookup (cpk), the state bits leave with two 16-byte writes after the last repair; the
                // only LDS access on the dependent chain (and in a repair pass the only one at all) is the state-table read.
                const int k = wv;
                const KcFseT* f = &S.fse[S.useIdx[k]];  // derived from S directly: keeps the LDS address space (ds_read, not flat_load)
                const uint16_t* __restrict__ tbl = f->st;
                const uint32_t* __restrict__ cpk = S.cpk[k];
                const uint8_t* __restrict__ cod = &S.codes[k][CH_PAD];  // chain position x at cod[x]
                const uint32_t trueIn = jb ? (uint32_t)fse_init_state(f, S.codes[k][CH_PAD - 1]) : (uint32_t)S.state[k];
                if (jb && lane == 0) S.sbits[k][CH_PAD - 1] = 0;
                const int nL = (nrem + CH_SEG - 1) / CH_SEG;  // lanes with a non-empty segment
                const bool act = lane < nL;
                const int x0 = lane * CH_SEG;
                const int cnt = nrem - x0 < CH_SEG ? nrem - x0 : CH_SEG;
                uint32_t pk[CH_SEG], sbp[CH_SEG / 2];
#pragma unroll
                for (int q = 0; q < CH_SEG / 2; q++) sbp[q] = 0;
                uint32_t st = trueIn, assumed = trueIn, endSt = trueIn;
                if (act) {
                    if (lane > CH_WB) st = tbl[0];  // any table entry is a valid state; lanes 0..CH_WB run from the chunk's entry state, exactly
#pragma unroll
                    for (int b = CH_WB; b >= 1; b--) {
                        if (lane >= b) {
                            chain_load16(cod + x0 - b * CH_SEG, cpk, pk);
                            st = chain_seg16<false>(st, pk, CH_SEG, tbl, sbp);
                        }
                    }
                    assumed = st;
                    chain_load16(cod + x0, cpk, pk);
                    endSt = chain_seg16<true>(st, pk, cnt, tbl, sbp);
                }
                for (;;) {
                    uint32_t prevEnd = (uint32_t)__shfl_up((int)endSt, 1, 64);
                    if (lane == 0) prevEnd = trueIn;
                    const unsigned long long bad = __ballot(act && assumed != prevEnd);
                    if (bad == 0ull) break;
#ifdef KC_CHAIN_STATS
                    if (P.prof && lane == 0) { atomicAdd(&P.prof[17], 1ull); atomicAdd(&P.prof[18 + k], 1ull); atomicAdd(&P.prof[24], (unsigned long long)__popcll(bad & ~(bad << 1))); if (f->useRLE) atomicAdd(&P.prof[25], 1ull); }
#endif
                    // Of every run of consecutive lanes whose entry state disagrees with the predecessor's exit state, only the FIRST
                    // re-encodes its segment in this pass: its predecessor is consistent, while the exit states of the others'
                    // predecessors are about to change (re-running them too replaces a warm-up guess that may well be right by a
                    // state that is certainly stale, and the error then cascades down the wave: measured 10 passes x 6.6 segments
                    // per offset-code chunk, against ~10 genuinely wrong guesses).  The lowest wrong lane of the wave has a proven
                    // predecessor, so each pass fixes at least that lane for good; on exit every entry state equals the
                    // predecessor's exit state, i.e. the sequential chain.
                    if ((bad & ~(bad << 1)) >> lane & 1ull) {
                        assumed = prevEnd;
                        endSt = chain_seg16<true>(prevEnd, pk, cnt, tbl, sbp);
                    }
                }
                if (act) {  // the segment's state bits: 16 x u16, 32-byte aligned
                    uint4* o = (uint4*)&S.sbits[k][CH_PAD + x0];
                    o[0] = make_uint4(sbp[0], sbp[1], sbp[2], sbp[3]);
                    o[1] = make_uint4(sbp[4], sbp[5], sbp[6], sbp[7]);
                }
#ifdef KC_CHAIN_STATS
                if (P.prof && lane == 0) { atomicAdd(&P.prof[16], 1ull); if (f->useRLE) atomicAdd(&P.prof[26], 1ull); if (f->preDefined) atomicAdd(&P.prof[27], 1ull); }
#endif
                const uint32_t fin = (uint32_t)__shfl((int)endSt, nL > 0 ? nL - 1 : 0, 64);
                if (lane == 0) S.state[k] = (uint16_t)(nL > 0 ? fin : trueIn);
            }
            __syncthreads();
            PROF_FINB(6);
            PROF_MARK(11);
            // pack this chunk
            for (int t0 = 0; t0 < cn; t0 += ET) {
                const int j = t0 + tid;
                uint64_t fv0 = 0, fv1 = 0;  // up to 27 state bits, then up to 63 extra bits
                int fb0 = 0, fb1 = 0;
                if (j < cn) {
#if KC_PACK_REUSE
                    static_assert(R == 4, "the select below is written for four staged sequences per thread");
                    const int rr = t0 / ET;
                    const uint64_t s = rr == 0 ? sv[0] : (rr == 1 ? sv[1] : (rr == 2 ? sv[2] : sv[3]));
#else
                    const uint64_t s = sq[hiSeq - 1 - j];
#endif
                    const uint32_t cl = S.codes[0][slot0 + j], co = S.codes[1][slot0 + j], cm = S.codes[2][slot0 + j];
                    const int lb = E[0]->outBits[cl] & 31, ob = E[1]->outBits[co] & 31, mb = E[2]->outBits[cm] & 31;
                    const uint32_t ll = seq_ll(s), ml = seq_ml(s), of = seq_of(s);
                    const uint64_t lv = ll & (lb ? (0xFFFFFFFFu >> (32 - lb)) : 0u);
                    const uint64_t mv = ml & (mb ? (0xFFFFFFFFu >> (32 - mb)) : 0u);
                    const uint64_t ov = of & (ob ? (0xFFFFFFFFu >> (32 - ob)) : 0u);
                    // state bits: OF, ML, LL (blockenc.go:757-787)
                    const uint32_t so = S.sbits[1][slot0 + j], sm = S.sbits[2][slot0 + j], sl = S.sbits[0][slot0 + j];
                    fv0 = (uint64_t)(so & 0xFFF); fb0 = (int)(so >> 12);
                    fv0 |= (uint64_t)(sm & 0xFFF) << fb0; fb0 += (int)(sm >> 12);
                    fv0 |= (uint64_t)(sl & 0xFFF) << fb0; fb0 += (int)(sl >> 12);
                    // extra bits: LL lowest, then ML, then OF (first element: LL, ML, OF — same order)
                    fv1 = lv | (mv << lb) | (ov << (lb + mb));
                    fb1 = lb + mb + ob;
                }
                uint64_t tot;
                const uint64_t ex = block_excl_scan64((uint64_t)(fb0 + fb1), S.wsum, &tot, tid) + bitRun;
                if ((int64_t)((bitRun + tot + 7) >> 3) >= (int64_t)seqBudgetBytes) { overflow = true; break; }
                // Pack this step's fields in LDS (ds_or), then flush the complete words to the staging stream with coalesced stores.
                // The word a step ends in the middle of is not stored: it is carried into the next step's first LDS word, so no
                // staging word is ever written twice (no zero-fill of the staging area, no global atomics).
                uint32_t* __restrict__ pk = (uint32_t*)S.whist;  // 1024 words, free during the sequence phase
                const uint32_t wbase = (uint32_t)(bitRun >> 5);
                const int nwords = (int)((((uint32_t)bitRun & 31u) + (uint32_t)tot + 31u) >> 5);
                for (int i = tid; i < nwords + 2; i += ET) pk[i] = i == 0 ? carry : 0u;
                __syncthreads();
                if (j < cn) {
                    const uint64_t rel = ex - ((uint64_t)wbase << 5);
                    lds_or_bits(pk, rel, fv0, fb0);
                    if (fb1 <= 32) lds_or_bits(pk, rel + fb0, fv1, fb1);
                    else { lds_or_bits(pk, rel + fb0, fv1 & 0xFFFFFFFFull, 32); lds_or_bits(pk, rel + fb0 + 32, fv1 >> 32, fb1 - 32); }
                }
                __syncthreads();
                const bool lastPartial = (((uint32_t)bitRun + (uint32_t)tot) & 31u) != 0u;
                const int nstore = lastPartial ? nwords - 1 : nwords;
                for (int i = tid; i < nstore; i += ET) sw[wbase + i] = pk[i];
                carry = lastPartial ? pk[nwords - 1] : 0u;  // pk is rewritten only after the next step's scan barriers
                bitRun += tot;
            }
            __syncthreads();
            PROF_FINB(7);
        }
        PROF_MARK(12);
        int seqStreamBytes = 0;
        if (!overflow) {
            // final states: ml.flush, of.flush, ll.flush, end mark (blockenc.go:804-807)
            const int tb = (int)E[2]->tableLog + (int)E[1]->tableLog + (int)E[0]->tableLog + 1;
            if ((int64_t)((bitRun + tb + 7) >> 3) >= (int64_t)seqBudgetBytes) overflow = true;
            else {
                if (tid == 0) {
                    uint64_t bp = bitRun;
                    sw[bitRun >> 5] = carry;       // the open word, then a clean one: the final states + end mark (<= 27 bits) may reach into it
                    sw[(bitRun >> 5) + 1] = 0;
                    const int mlL = E[2]->tableLog, ofL = E[1]->tableLog, llL = E[0]->tableLog;
                    or_bits(sw, bp, (uint64_t)(S.state[2] & ((1u << mlL) - 1u)), mlL); bp += mlL;
                    or_bits(sw, bp, (uint64_t)(S.state[1] & ((1u << ofL) - 1u)), ofL); bp += ofL;
                    or_bits(sw, bp, (uint64_t)(S.state[0] & ((1u << llL) - 1u)), llL); bp += llL;
                    or_bits(sw, bp, 1, 1);
                }
                seqStreamBytes = (int)((bitRun + tb + 7) >> 3);
            }
        }
        __syncthreads();
        const int bodyLen = litSecLen + seqhdrLen + seqStreamBytes;
        if (overflow || bodyLen >= size) {
            // ---------- raw fallback (blockenc.go:811-817) ----------
            __syncthreads();
            if (tid == 0) {
                put_block_header(bout, last, 0u, (uint32_t)size);
                S.huf.reuse = 2;  // litEnc.Reuse = ReusePolicyNone
                if (!last) {
                    if (!(m.flags & KC_BF_FORCED) && (m.o1_out != m.o1_in || m.o2_out != m.o2_in)) { P.redo_blk[blk0 + (uint32_t)b] = 1; atomicOr(&P.redo_mask[u], 1u); }
                }
            }
            raw_payload();
            opos += 3 + size;
            __syncthreads();
            continue;
        }
        if ((m.flags & KC_BF_FORCED) && tid == 0) atomicExch(P.err_flag, 4u);  // host forced a pop that did not recur
        // ---------- assemble: sequence header bytes + stream, patch block header, setPrev ----------
        wg_copy(bout + 3 + litSecLen + seqhdrLen, (const uint8_t*)sw, seqStreamBytes);
        if (tid == 0) {
            uint8_t* p = bout + 3 + litSecLen;
            for (int k = 0; k < seqhdrLen; k++) p[k] = S.seqhdr[k];
            put_block_header(bout, last, 2u, (uint32_t)bodyLen);
            // seqCoders.setPrev(ll, ml, of) (seqenc.go:21-42)
            for (int k = 0; k < 3; k++) {
                const int used = S.useIdx[k];
                if (used == S.curIdx[k]) {
                    const uint8_t t = S.prevIdx[k]; S.prevIdx[k] = S.curIdx[k]; S.curIdx[k] = t;
                    S.fse[S.curIdx[k]].reUsed = 0;
                    S.fse[S.prevIdx[k]].reUsed = 1;
                } else if (used != S.prevIdx[k]) {
                    S.fse[S.prevIdx[k]].symbolLen = 0;  // ensure we cannot reuse by accident
                }
            }
        }
        opos += 3 + bodyLen;
        __syncthreads();
        PROF_MARK(13);
    }

    if (UB.emptyLast && !JOB) {  // Close found nothing buffered: final block without data (encoder.go:315-329)
        if (tid == 0) { outp[opos] = 0x01; outp[opos + 1] = 0x00; outp[opos + 2] = 0x00; }
        opos += 3;
    }
    // ---- checksum (enc_base.go:34-38) ----
    if (ulen > 0 && P.crc && !JOB) {  // (a job stream's checksum covers the whole stream: the host appends it)
        if (tid == 0 && P.xxh != nullptr) {  // (null: kc_xxh64_fin_kernel fills the field in, behind this kernel)
            const uint64_t h = P.xxh[u];
            outp[opos] = (uint8_t)h; outp[opos + 1] = (uint8_t)(h >> 8); outp[opos + 2] = (uint8_t)(h >> 16); outp[opos + 3] = (uint8_t)(h >> 24);
        }
        opos += 4;
    }
    if (P.prof && tid == 0) {
        for (int i = 0; i < 16; i++) atomicAdd(&P.prof[i], S.profAcc[i]);
        for (int i = 0; i < 8; i++) atomicAdd(&P.prof[40 + i], S.profAcc[16 + i]);
    }
    if (tid == 0) {
        P.out_size[u] = (uint32_t)opos;
        if (P.unit_raw != nullptr) P.unit_raw[u] = (nblk > 0 && nRawDef == nblk && !JOB && hist0 == 0) ? 1u : 0u;
    }
}

void kc_launch_zstd_entropy(const KcEntropyParams& P, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL(kc_zstd_entropy_kernel, dim3(grid), dim3(ET), 0, st, P);
}
