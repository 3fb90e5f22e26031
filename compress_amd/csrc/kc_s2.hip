// kc_s2.hip — S2 block encoder for gfx950: N independent blocks, each == s2.Encode(nil, block).
//
// Replaces s2.Encode (s2/encode.go:29-56), encodeBlockGo / encodeBlockGo64K
// (s2/encode_all.go:72-284 / 287-500) and emitLiteral / emitRepeat / emitCopy
// (s2/encode_go.go:80-234).  Parity target is the portable Go encoder (build tag noasm), NOT the
// amd64 assembly variant.
//
// Same execution scheme as the zstd match finder (kc_zstd_match.hip, v3): 8 lanes per block,
// 8 blocks per wave, speculative probing of the next probe steps of the current skip segment with
// ordered commit, hash table (2^14 x u32) per block in an HBM scratch arena.  S2 has no entropy
// stage: the group emits the tag bytes and literal runs directly into the block's staging slot.
// Differences from the zstd parse that are reproduced literally:
//   * three positions are hashed per probe step (s, s+1, s+2); the s+2 bucket is read AFTER the
//     s and s+1 buckets were written (encode_all.go:327-401), and is written only on some paths;
//   * table entries carry no validity: an empty (zero) slot means "candidate = position 0" and is
//     verified on the bytes only (App. A-20b);
//   * the repeat check at s+1 is always armed (repeat starts at 1), its forward extension stops at
//     sLimit, the regular one at len-8, both in whole 8-byte steps with no byte tail;
//   * incompressible bail-outs against dstLimit = len - len>>5 - 5 (App. A-20).
// Table entry layout: position in the low PB bits, a tag of the 4 source bytes at that position
// above; a zero entry is the empty slot and is never filtered by its tag.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_s2_dev.h"
#include "kc_wave.h"

#define S2G 8
// Source window of the default / Snappy levels (round 6; the zstd match finders' ring): the block's bytes around the parse position in a
// per-block ring in LDS, refilled 128 bytes at a time by the group's lanes, so the 8 bytes at a probe position (and near candidates)
// are an LDS read instead of the first of three dependent round trips.  MEASURED AND LEFT OFF (profiles/r06_ab_kernels.txt, C4 same
// box): without 43.6 ms per 2 GiB; with it 61.3 ms — the kernel sits at 127 VGPRs and the window's state takes it to 139, i.e. from
// 16 to 12 waves per CU; held at 128 VGPRs (S2_WPE=4: 28 bytes of scratch) 44.8 ms.  This kernel runs at 0.9 of the DRAM-request floor
// of its tables (bench.py roofline.floor): the round trip the ring saves is not what bounds it, occupancy is.  Kept for measurement
// builds (KC_EXTRA_FLAGS=-DS2_RING=1).
#ifndef S2_RING
#define S2_RING 0
#endif
#define S2_RB 512
#define S2_MIRROR 32
#define S2_STRIDE (S2_RB + S2_MIRROR)
#ifndef S2_AHEAD
#define S2_AHEAD 224
#endif

__device__ __forceinline__ uint32_t s2g_ballot(bool p, int grp) { return (uint32_t)((ballot64(p) >> (grp * S2G)) & 0xFFull); }
__device__ __forceinline__ uint32_t s2g_bcast32(uint32_t v, int grp, int srcLig) { return (uint32_t)__shfl((int)v, grp * S2G + srcLig, 64); }
__device__ __forceinline__ uint64_t s2g_bcast64(uint64_t v, int grp, int srcLig) {
    const int src = grp * S2G + srcLig;
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Forward extension in whole 8-byte steps: positions a (ahead) and b (behind) advance together while
// a <= limit (encode_all.go:353-360 with limit = sLimit, :435-442 with limit = len-8).  Returns the new a.
__device__ __forceinline__ int s2_extend(const uint8_t* __restrict__ base, int a, int b, int limit, int lig, int grp) {
    for (;;) {
        const int pa = a + 8 * lig, pb = b + 8 * lig;
        const bool inb = pa <= limit;
        uint64_t diff = 0;
        if (inb) diff = ld64(base + pa) ^ ld64(base + pb);
        const uint32_t oob = s2g_ballot(!inb, grp);       // lanes past the limit (a suffix)
        const uint32_t dm = s2g_ballot(inb && diff != 0, grp);
        const int firstOob = oob ? __builtin_ctz(oob) : S2G;
        if (dm) {
            const int fl = __builtin_ctz(dm);  // necessarily < firstOob
            const uint64_t d = s2g_bcast64(diff, grp, fl);
            return a + 8 * fl + (ctz64(d) >> 3);
        }
        if (firstOob < S2G) return a + 8 * firstOob;
        a += 8 * S2G;
        b += 8 * S2G;
    }
}


// The assembly's matchLen (s2/_generate/gen.go:2778-2880): the exact common prefix, up to the end of the block.  Whole 8-byte
// steps while 8 bytes are left, then the tail byte by byte.  Returns the new a.
__device__ __forceinline__ int s2_extend_exact(const uint8_t* __restrict__ base, int a, int b, int len, int lig, int grp) {
    for (;;) {
        const int pa = a + 8 * lig, pb = b + 8 * lig;
        const bool inb = pa + 8 <= len;
        uint64_t diff = 0;
        if (inb) diff = ld64(base + pa) ^ ld64(base + pb);
        const uint32_t oob = s2g_ballot(!inb, grp);
        const uint32_t dm = s2g_ballot(inb && diff != 0, grp);
        const int firstOob = oob ? __builtin_ctz(oob) : S2G;
        if (dm) {
            const int fl = __builtin_ctz(dm);
            const uint64_t d = s2g_bcast64(diff, grp, fl);
            return a + 8 * fl + (ctz64(d) >> 3);
        }
        if (firstOob < S2G) {
            const int t = a + 8 * firstOob, tb = b + 8 * firstOob;
            const int rem = len - t;  // 0..7
            const bool ne = lig < rem && base[t + lig] != base[tb + lig];
            const uint32_t nm = s2g_ballot(ne, grp);
            const int k = nm ? __builtin_ctz(nm) : rem;
            return t + (k < rem ? k : rem);
        }
        a += 8 * S2G;
        b += 8 * S2G;
    }
}

#ifdef S2_WPE
#define S2_KATTR __attribute__((amdgpu_waves_per_eu(S2_WPE, S2_WPE)))
#else
#define S2_KATTR
#endif
template <int LEVEL>  // 0: s2.Encode (encodeBlockGo / ...64K), 1: s2.EncodeBetter (encodeBlockBetterGo / ...64K), 2: s2.EncodeSnappy (encodeBlockSnappyGo / ...64K), 3: s2.EncodeSnappyBetter (encodeBlockBetterSnappyGo / ...64K)
__global__ __launch_bounds__(64) S2_KATTR void kc_s2_encode_kernel(KcS2Params P) {
    constexpr int G = S2G;
    __shared__ uint32_t crcT[4][256];
    if (P.framed) {
        for (int i = (int)threadIdx.x; i < 256; i += 64) {
            uint32_t c = (uint32_t)i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            crcT[0][i] = c;
        }
        __syncthreads();
        for (int i = (int)threadIdx.x; i < 256; i += 64) {
            uint32_t c = crcT[0][i];
            for (int t = 1; t < 4; t++) { c = crcT[0][c & 0xFF] ^ (c >> 8); crcT[t][i] = c; }
        }
        __syncthreads();
    }
    const int lane = (int)threadIdx.x;
    const int lig = lane % G, grp = lane / G;
    const uint32_t bi = blockIdx.x * (64 / G) + (uint32_t)grp;
    const bool gact = bi < P.n_blocks;
    const uint32_t bq = gact ? bi : 0u;
    const uint8_t* __restrict__ src = P.src + P.blk_off[bq];
    const int len = gact ? (int)(P.blk_off[bq + 1] - P.blk_off[bq]) : 0;
    uint8_t* __restrict__ slot = P.stage + P.stage_off[bq];
    uint8_t* __restrict__ out = slot + (P.framed ? 8 : 0);  // chunk header (type, len24, crc) goes in front
    uint32_t* __restrict__ tab = P.tables + (size_t)bi * P.table_stride;  // u32 entries per block
    if (!gact) return;  // whole group leaves together

    // uvarint(len) header (encode.go:39)
    int hdr = 0;
    {
        uint64_t x = (uint64_t)len;
        while (x >= 0x80) { if (lig == 0) out[hdr] = (uint8_t)x | 0x80; hdr++; x >>= 7; }
        if (lig == 0) out[hdr] = (uint8_t)x;
        hdr++;
    }
    uint8_t* __restrict__ dst = out + hdr;
    int d = 0;
    bool stored = false;  // encodeBlock returned 0 -> emit everything as one literal
    if (len == 0 && !P.framed) { if (lig == 0) P.out_size[bi] = (uint32_t)hdr; return; }
    if (len == 0) stored = true;
    if (len < 32) stored = true;  // minNonLiteralBlockSize

    // ---- output staging (s2.EncodeBetter) ----
    // Tag bytes and short literals are a few bytes each; stored straight to the slot every one of them is a partial-line write
    // that the L2 (flooded by the table traffic) evicts before the next one arrives: a DRAM write transaction per emit.  The
    // group therefore assembles the stream in an LDS ring addressed by the byte position in the slot (slots are 64-byte
    // aligned) and stores whole 64-byte lines, 8 bytes per lane; literals longer than 128 bytes go line by line from the source.
    // Measured on C4 (2 GiB JSON): s2.EncodeBetter 122 -> 106.6 ms; s2.Encode 43.4 -> 47.7 ms (fewer, longer matches: the extra
    // instructions per emit, paid once per group of the wave, cost more than the partial writes), so s2.Encode keeps the direct
    // stores, as does the Snappy variant (one copy can be hundreds of 3-byte operations).
    constexpr bool RING = LEVEL == 1;
    constexpr int ORING = 256;
    __shared__ __attribute__((aligned(16))) uint8_t oring_all[RING ? (64 / G) * ORING : 16];
    uint8_t* const oring = oring_all + (RING ? grp * ORING : 0);
    const int q0 = (P.framed ? 8 : 0) + hdr;  // position of dst in the slot
    int flushedQ = 0;                         // slot bytes [0, flushedQ) are in memory, [flushedQ, q0 + d) in the ring
    if (RING && !stored) {
        *(uint4*)(oring + 32 * lig) = make_uint4(0, 0, 0, 0);
        *(uint4*)(oring + 32 * lig + 16) = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
        if (lig == 0) for (int k = 0; k < hdr; k++) oring[(P.framed ? 8 : 0) + k] = out[k];
        __builtin_amdgcn_wave_barrier();
    }
    // The ring is zero wherever nothing has been written yet, so up to 8 bytes at any alignment go in as one or two 64-bit ORs
    // (ds_or_b64): with 8 groups of a wave running their emits at different times, every instruction here is paid 8 times.
    auto ring_or = [&](int q, uint64_t v) {  // v's non-zero bytes -> slot positions q, q+1, ...
        const uint32_t sh = ((uint32_t)q & 7u) * 8u;
        atomicOr((unsigned long long*)(oring + (q & (ORING - 8))), (unsigned long long)(v << sh));
        if (sh != 0u && (v >> (64u - sh)) != 0ull) atomicOr((unsigned long long*)(oring + ((q + 8) & (ORING - 8))), (unsigned long long)(v >> (64u - sh)));
    };
    auto ring_flush = [&](int qend) {  // group-uniform: store every complete line below qend, and clear it in the ring
        while (flushedQ + 64 <= qend) {
            __builtin_amdgcn_wave_barrier();
            unsigned long long* w = (unsigned long long*)(oring + ((flushedQ + 8 * lig) & (ORING - 1)));
            st64(slot + flushedQ + 8 * lig, *w);
            *w = 0ull;
            flushedQ += 64;
            __builtin_amdgcn_wave_barrier();
        }
    };
    auto ring_copy = [&](const uint8_t* __restrict__ p, int cnt, int q) {  // p[0, cnt) -> ring at slot position q (cnt <= 128)
        for (int k = lig * 8; k < cnt; k += 8 * G) {
            const int n8 = cnt - k < 8 ? cnt - k : 8;
            uint64_t v = 0;
            if (n8 == 8) v = ld64(p + k);
            else for (int b = 0; b < n8; b++) v |= (uint64_t)p[k + b] << (8 * b);
            ring_or(q + k, v);
        }
    };
    auto emit_lit = [&](int from, int n) -> int {  // emitLiteral(dst[d:], src[from:from+n])
        if (!RING) return s2_emit_literal<S2G>(dst + d, src + from, n, lig);
        if (n == 0) return 0;
        const uint32_t m = (uint32_t)(n - 1);
        int q = q0 + d, i;
        uint64_t tag;
        if (m < 60) { i = 1; tag = m << 2; }
        else if (m < (1u << 8)) { i = 2; tag = (60u << 2) | ((uint64_t)m << 8); }
        else if (m < (1u << 16)) { i = 3; tag = (61u << 2) | ((uint64_t)m << 8); }
        else if (m < (1u << 24)) { i = 4; tag = (62u << 2) | ((uint64_t)m << 8); }
        else { i = 5; tag = (63u << 2) | ((uint64_t)m << 8); }
        if (lig == 0) ring_or(q, tag);
        q += i;
        const uint8_t* __restrict__ lit = src + from;
        int done = 0;
        if (n > 128) {
            const int h = (64 - (q & 63)) & 63;  // through the ring up to the next line boundary, then whole lines from the source
            ring_copy(lit, h, q);
            q += h;
            ring_flush(q);
            const int nl = (n - h) >> 6;
            for (int j = 0; j < nl; j++) st64(slot + q + 64 * j + 8 * lig, ld64(lit + h + 64 * j + 8 * lig));
            q += 64 * nl;
            flushedQ = q;
            done = h + 64 * nl;
        }
        ring_copy(lit + done, n - done, q);
        ring_flush(q + n - done);
        return i + n;
    };
    // an operation (2..10 bytes) is assembled in two registers and ORed in by lane 0
    auto emit_op = [&](uint64_t lo, uint64_t hi, int n) {
        const int q = q0 + d;
        if (lig == 0) { ring_or(q, lo); if (n > 8) ring_or(q + 8, hi); }
        ring_flush(q + n);
    };
    bool smallRep = false;  // the assembly's emitRepeat as generated into encodeBlockAsm8B (see below)
    auto emit_repeat = [&](int offset, int length) -> int {
        if (smallRep && length > 8 && length < 12) {  // no two-byte offset form there: gen.go:1991-1994 leaves its test (and jump) out
            if (RING) emit_op((uint64_t)(5 << 2 | 1) | ((uint64_t)(length - 8) << 16), 0, 3);
            else if (lig == 0) { dst[d] = (uint8_t)(5 << 2 | 1); dst[d + 1] = 0; dst[d + 2] = (uint8_t)(length - 8); }
            return 3;
        }
        if (!RING) { if (lig == 0) s2_emit_repeat1(dst + d, offset, length); return s2_repeat_size(offset, length); }
        uint64_t lo = 0, hi = 0;
        const int n = s2_put_repeat([&](int k, uint8_t v) { if (k < 8) lo |= (uint64_t)v << (8 * k); else hi |= (uint64_t)v << (8 * (k - 8)); }, offset, length);
        emit_op(lo, hi, n);
        return n;
    };
    auto emit_copy = [&](int offset, int length) -> int {
        if (!RING) { if (lig == 0) s2_emit_copy1(dst + d, offset, length); return s2_copy_size(offset, length); }
        uint64_t lo = 0, hi = 0;
        const int n = s2_put_copy([&](int k, uint8_t v) { if (k < 8) lo |= (uint64_t)v << (8 * k); else hi |= (uint64_t)v << (8 * (k - 8)); }, offset, length);
        emit_op(lo, hi, n);
        return n;
    };

    if ((LEVEL == 0 || LEVEL == 2) && !stored) {
        constexpr bool SNAPPY = LEVEL == 2;
        // ---- source window ----
        __shared__ __attribute__((aligned(16))) uint8_t sring_all[S2_RING ? (64 / G) * S2_STRIDE : 16];
        uint8_t* const sring = sring_all + (S2_RING ? grp * S2_STRIDE : 0);
        const int boff = (int)((uintptr_t)src & 15);
        const uint8_t* __restrict__ abase = src - boff;
        const uint8_t* const srcHi = src + len;  // (the block's own end: the bytes behind it belong to the next block or to the buffer's padding)
        int wlo = 0, whi = 0;
        bool pend = false;
        uint4 rf = make_uint4(0, 0, 0, 0);
        auto window = [&](int sp) {  // top of every probe round (group-uniform)
            if (!S2_RING) return;
            if (pend) {
                const int ro = (whi + 16 * lig) & (S2_RB - 1);
                *(uint4*)(sring + ro) = rf;
                if (ro < S2_MIRROR) *(uint4*)(sring + S2_RB + ro) = rf;
                whi += 16 * G;
                if (whi - wlo > S2_RB) wlo = whi - S2_RB;
                pend = false;
            }
            KC_EMU_SYNC();
            const int sa = sp + boff;
            if (sa >= whi || sa < wlo) { const int w0 = sa & ~15; wlo = whi = w0; }
            if (whi - sa < S2_AHEAD) {
                const uint8_t* q = abase + whi + 16 * lig;
                rf = make_uint4(0, 0, 0, 0);
                if (q < srcHi) rf = *(const uint4*)q;  // aligned: never leaves the 16-byte granule of a readable byte
                pend = true;
            }
        };
        auto rd64 = [&](int pos) -> uint64_t {
            const int a = pos + boff, a4 = a & ~3;
            if (S2_RING && a4 >= wlo && a4 + 12 <= whi) {
                const uint32_t* r = (const uint32_t*)(sring + (a4 & (S2_RB - 1)));
                const uint32_t r0 = r[0], r1 = r[1], r2 = r[2];
                const uint32_t sh = (uint32_t)(a & 3);
                return (uint64_t)__builtin_amdgcn_alignbyte(r1, r0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(r2, r1, sh) << 32);
            }
            return ld64(src + pos);
        };
        auto rd32 = [&](int pos) -> uint32_t {
            const int a = pos + boff, a4 = a & ~3;
            if (S2_RING && a4 >= wlo && a4 + 8 <= whi) {
                const uint32_t* r = (const uint32_t*)(sring + (a4 & (S2_RB - 1)));
                return __builtin_amdgcn_alignbyte(r[1], r[0], (uint32_t)(a & 3));
            }
            return ld32(src + pos);
        };  // encode_all.go:502 / :692: the same parse, every copy through emitCopyNoRepeat
        int SKIP = len <= (64 << 10) ? 5 : 6;  // encodeBlockGo64K vs encodeBlockGo (encode_go.go:23-26)
        // P.variant 1: the bytes of the amd64 ASSEMBLY encoders (s2/encode_amd64.go:23-87, 176-239; generator s2/_generate/gen.go:170-855)
        // — the same algorithm with, per size class, another table size / hash length / skip rate, matches extended to the very end
        // of the block, `nextS >= sLimit`, output margin 9 and the literal header's worst case in every bail-out test.
        const bool AX = P.variant == 1;
        int HSHL = 16, HSHR = 64 - S2_TABLE_BITS, LITOVH = 0;
        uint64_t HPRIME = KC_PRIME6;
        if (AX) {
            const bool top = SNAPPY ? len > 65536 : len >= (4 << 20);
            if (top || len >= (16 << 10)) { SKIP = 6; LITOVH = top ? 5 : (SNAPPY ? 3 : 4); }                                    // ...Asm / ...Asm4MB / ...Asm64K
            else if (len >= (4 << 10)) { SKIP = 5; HSHL = 24; HPRIME = KC_PRIME5; HSHR = 64 - 12; LITOVH = 3; }                   // ...Asm12B
            else if (len >= 512) { SKIP = 5; HSHL = 32; HPRIME = (uint64_t)KC_PRIME4; HSHR = 64 - 10; LITOVH = 3; }              // ...Asm10B
            else { SKIP = 4; HSHL = 32; HPRIME = (uint64_t)KC_PRIME4; HSHR = 64 - 8; LITOVH = 3; smallRep = !SNAPPY; }            // ...Asm8B
        }
        auto hashOf = [&](uint64_t v) -> uint32_t { return (uint32_t)(((v << HSHL) * HPRIME) >> HSHR); };
        const int PB = bits_len32((uint32_t)len);
        const int TB = (32 - PB) > 16 ? 16 : (32 - PB);
        const uint32_t posMask = (1u << PB) - 1u;
        auto tagOf = [&](uint32_t v) -> uint32_t { return (v * 2654435761u) >> (32 - TB); };
        auto mk = [&](int pos, uint32_t val) -> uint32_t { return (uint32_t)pos | (tagOf(val) << PB); };
        const int sLimit = len - 8;
        const int sLimT = AX ? sLimit - 1 : sLimit;                     // a step is taken while nextS <= sLimT
        const int dstLimit = AX ? (len - 9) - (len >> 5) : len - (len >> 5) - 5;
        const int bailLim = AX ? dstLimit - LITOVH - 1 : dstLimit;      // literals of n bytes are refused when d + n > bailLim
        const int cpLim = AX ? dstLimit - 1 : dstLimit;                 // after a copy: d > cpLim
        int nextEmit = 0, s = 1, repeat = 1;
        bool fin = false;   // goto emitRemainder
        int W = G;  // speculation width: every speculative probe step costs three table lines from HBM, and matches come every few steps
        while (!fin && !stored) {
            KC_EMU_SYNC();  // (lane 0's table stores behind a match precede the next round's lookups)
            window(s);
            // ---------------- speculative probe round ----------------
            const int d0 = s - nextEmit;
            const int k0 = d0 >> SKIP;
            const int step = 4 + k0;
            const int p = s + lig * step;
            // lane i is a real probe step iff all earlier steps stayed in the skip segment and nextS(p) <= sLimit
            const bool inseg = lig == 0 || ((d0 + (lig - 1) * step) >> SKIP) == k0;
            const int nextS = p + ((p - nextEmit) >> SKIP) + 4;
            const bool valid = lig < W && inseg && nextS <= sLimT;
            const bool term = inseg && nextS > sLimT;  // this step would `goto emitRemainder`
            uint64_t cv = 0;
            uint32_t h0 = 0xFFFFFFF0u, h1 = 0xFFFFFFF1u, h2 = 0xFFFFFFF2u, e0 = 0, e1 = 0, e2 = 0;
            if (valid) {
                cv = rd64(p);
                h0 = hashOf(cv); h1 = hashOf(cv >> 8); h2 = hashOf(cv >> 16);
                e0 = tab[h0]; e1 = tab[h1]; e2 = tab[h2];
            }
            bool dep = false;
#pragma unroll
            for (int dd = 1; dd < G; dd++) {
                const uint32_t a0 = (uint32_t)__shfl_up((int)h0, dd, G), a1 = (uint32_t)__shfl_up((int)h1, dd, G), a2 = (uint32_t)__shfl_up((int)h2, dd, G);
                if (lig >= dd && (a0 == h0 || a0 == h1 || a0 == h2 || a1 == h0 || a1 == h1 || a1 == h2 || a2 == h0 || a2 == h1 || a2 == h2)) dep = true;
            }
            int kind = 0, cand = 0;  // 1 repeat at s+1, 2 match at s, 3 match at s+1, 4 match at s+2
            if (valid) {
                // the s+2 bucket is read after the s / s+1 buckets were written (encode_all.go:401)
                uint32_t e2v = e2;
                if (h2 == h1) e2v = mk(p + 1, (uint32_t)(cv >> 8));
                else if (h2 == h0) e2v = mk(p, (uint32_t)cv);
                const int c0 = (int)(e0 & posMask), c1 = (int)(e1 & posMask), c2 = (int)(e2v & posMask);
                const bool ok0 = e0 == 0 || (e0 >> PB) == tagOf((uint32_t)cv);
                const bool ok1 = e1 == 0 || (e1 >> PB) == tagOf((uint32_t)(cv >> 8));
                const bool ok2 = e2v == 0 || (e2v >> PB) == tagOf((uint32_t)(cv >> 16));
                const uint32_t wr = rd32(p - repeat + 1);
                const uint32_t w0 = ok0 ? rd32(c0) : ~(uint32_t)cv;
                const uint32_t w1 = ok1 ? rd32(c1) : ~(uint32_t)(cv >> 8);
                const uint32_t w2 = ok2 ? rd32(c2) : ~(uint32_t)(cv >> 16);
                if ((uint32_t)(cv >> 8) == wr) kind = 1;
                else if ((uint32_t)cv == w0) { kind = 2; cand = c0; }
                else if ((uint32_t)(cv >> 8) == w1) { kind = 3; cand = c1; }
                else if ((uint32_t)(cv >> 16) == w2) { kind = 4; cand = c2; }
            }
            const uint32_t vm = s2g_ballot(valid, grp);
            const uint32_t tm = s2g_ballot(term, grp);
            const uint32_t depm = s2g_ballot(valid && dep, grp);
            const uint32_t hm = s2g_ballot(kind != 0, grp);
            const int nvalid = __popc(vm);  // valid lanes form a prefix; the terminating step (if any) is lane nvalid
            const int c = depm ? __builtin_ctz(depm) : G;
            const uint32_t hmc = hm & ((1u << c) - 1u);
            const bool found = hmc != 0;
            const int f = found ? __builtin_ctz(hmc) : 0;
            const int commitUpTo = found ? f : ((c < nvalid ? c : nvalid) - 1);
            if (valid && lig <= commitUpTo) {
                const bool winner = found && lig == f;
                tab[h0] = mk(p, (uint32_t)cv);
                tab[h1] = mk(p + 1, (uint32_t)(cv >> 8));
                // table[hash2] = s+2 is skipped when the step ends on the repeat or on the match at s
                if (!(winner && (kind == 1 || kind == 2))) tab[h2] = mk(p + 2, (uint32_t)(cv >> 16));
            }
            if (!found) {
                W = P.spec_grow == 0 ? W : (P.spec_grow == 1 ? (W + 1 < G ? W + 1 : G) : ((2 * W < G) ? 2 * W : G));
                if (c < nvalid) {
                    s = s + c * step;  // first dependent lane restarts as lane 0
                } else if (nvalid < G && (tm & (1u << nvalid))) {
                    fin = true;        // the step after the last committed one hits `nextS > sLimit`
                } else {
                    const int pl = s + (nvalid - 1) * step;  // nvalid >= 1 here
                    s = pl + ((pl - nextEmit) >> SKIP) + 4;
                }
                continue;
            }
            W = P.spec_w0;
            const int mkind = (int)s2g_bcast32((uint32_t)kind, grp, f);
            int candidate = (int)s2g_bcast32((uint32_t)cand, grp, f);
            const int ps = s + f * step;
            if (mkind == 1) {
                // ---------------- repeat at s+1 (encode_all.go:336-384) ----------------
                int base = ps + 1;
                {
                    // extend back: i = base - repeat; while base > nextEmit && i > 0 && src[i-1] == src[base-1]
                    int i0 = base - repeat;
                    int kmax = base - nextEmit;
                    if (i0 < kmax) kmax = i0;
                    const int back = grp_backlen<S2G>(src, base, i0, kmax, lig, grp);
                    base -= back;
                }
                if (d + (base - nextEmit) > bailLim) { stored = true; continue; }
                d += emit_lit(nextEmit, base - nextEmit);
                const int cand2 = ps - repeat + 4 + 1;
                s = AX ? s2_extend_exact(src, ps + 4 + 1, cand2, len, lig, grp) : s2_extend(src, ps + 4 + 1, cand2, sLimit, lig, grp);
                if (SNAPPY) { if (lig == 0) s2_emit_copy_nr1(dst + d, repeat, s - base); d += s2_copy_nr_size(repeat, s - base); }
                else if (nextEmit > 0) d += emit_repeat(repeat, s - base);
                else d += emit_copy(repeat, s - base);
                nextEmit = s;
                if (s >= sLimit) fin = true;  // (the assembly has no such test: its next step's nextS = s + 4 >= sLimit says the same)
                continue;
            }
            // ---------------- regular match (encode_all.go:387-489) ----------------
            s = ps + (mkind - 2);
            {
                int kmax = candidate;  // candidate > 0
                if (s - nextEmit < kmax) kmax = s - nextEmit;
                const int back = grp_backlen<S2G>(src, s, candidate, kmax, lig, grp);
                candidate -= back;
                s -= back;
            }
            if (d + (s - nextEmit) > bailLim) { stored = true; continue; }
            d += emit_lit(nextEmit, s - nextEmit);
            for (;;) {
                const int base = s;
                repeat = base - candidate;
                s = AX ? s2_extend_exact(src, s + 4, candidate + 4, len, lig, grp) : s2_extend(src, s + 4, candidate + 4, len - 8, lig, grp);
                if (SNAPPY) { if (lig == 0) s2_emit_copy_nr1(dst + d, repeat, s - base); d += s2_copy_nr_size(repeat, s - base); }
                else d += emit_copy(repeat, s - base);
                nextEmit = s;
                if (s >= sLimit) { fin = true; break; }
                if (d > cpLim) { stored = true; break; }
                // check for an immediate match, otherwise start the search at s+1 (:474-488)
                const uint64_t x = rd64(s - 2);
                const uint32_t m2Hash = hashOf(x), currHash = hashOf(x >> 16);
                const uint32_t ec = tab[currHash];
                KC_EMU_SYNC();
                if (lig == 0) { tab[m2Hash] = mk(s - 2, (uint32_t)x); tab[currHash] = mk(s, (uint32_t)(x >> 16)); }
                // make the writes visible to the group's next reads of these buckets (same wave: program order)
                candidate = (int)(ec & posMask);
                const bool okc = ec == 0 || (ec >> PB) == tagOf((uint32_t)(x >> 16));
                if (!okc || (uint32_t)(x >> 16) != ld32(src + candidate)) { s++; break; }
            }
        }
        if (!stored) {
            // emitRemainder (:491-499)
            if (AX || nextEmit < len) {  // (the assembly tests the bail-out even when nothing is left to emit)
                if (d + len - nextEmit > bailLim) stored = true;
                else if (nextEmit < len) d += emit_lit(nextEmit, len - nextEmit);
            }
        }
    }
    if ((LEVEL == 1 || LEVEL == 3) && !stored) {
        // LEVEL 3 = s2.EncodeSnappyBetter: encodeBlockBetterSnappyGo / ...64K (s2/encode_better.go:310-483 / 733-900): tables 2^16 + 2^14
        // (2^15 + 2^13 up to 64 KiB), the skip capped at maxSkip = 100, candidates accepted on 4 equal bytes only, every copy
        // through emitCopyNoRepeat.
        constexpr bool SNB = LEVEL == 3;
        // ---------------- s2.EncodeBetter: encodeBlockBetterGo (> 64 KiB) / encodeBlockBetterGo64K (s2/encode_better.go:50-307 / 485-730) ----------------
        // Long table (7-byte hash, 2^17 / 2^16 entries) + short table (4-byte hash, 2^14 / 2^13), every position probed (step 1, skip >>7 / >>6),
        // candidates accepted on 8 equal bytes (long, then short), then on 4 (long, then short with a lazy long lookup at s+1);
        // after a match: s+1 / end-2 into both tables, then the long table sparsely from two starting points.  The repeat check
        // inside the probe loop is dead code in the reference (`if false && ...`).  Entries: position | tag(4 bytes) << PB, 0 = empty
        // (== candidate 0, verified on the bytes only, as the reference does).
        const bool big = len > (64 << 10);
        int LB = SNB ? (big ? 16 : 15) : (big ? 17 : 16), SB = big ? 14 : 13, SKIP = big ? 7 : 6;
        int MAXSKIP = SNB ? 100 : 0, LSHL = 8, LITOVH = 0, OM = 6;
        uint64_t LPRIME = 58295818150454627ULL;  // hash7
        bool bigoff = big;
        // P.variant 1: the amd64 assembly forms (s2/encode_amd64.go:99-166, 249-316; generator gen.go:873-1655): per size class other
        // table sizes / long-hash length / skip rate, the skip capped at 100 in the large classes of BOTH levels, the 8-byte
        // candidate tests also at the Snappy-compatible level, output margin 6 (9 Snappy-compatible) with the literal header's
        // worst case in every bail-out test, `nextS >= sLimit`.
        const bool AX = P.variant == 1;
        if (AX) {
            OM = SNB ? 9 : 6;
            const bool top = SNB ? len > 65536 : len > (4 << 20);
            if (top) { LB = 17; SB = 14; SKIP = 7; MAXSKIP = 100; LITOVH = 5; bigoff = true; }
            else if (len >= (16 << 10)) {
                if (SNB) { LB = 16; SB = 13; SKIP = 7; MAXSKIP = 0; LITOVH = 3; bigoff = false; }       // encodeSnappyBetterBlockAsm64K
                else { LB = 17; SB = 14; SKIP = 7; MAXSKIP = 100; LITOVH = 4; bigoff = true; }          // encodeBetterBlockAsm4MB
            }
            else if (len >= (4 << 10)) { LB = 14; SB = 12; SKIP = 6; MAXSKIP = 0; LITOVH = 3; bigoff = false; LSHL = 16; LPRIME = KC_PRIME6; }
            else if (len >= 512) { LB = 12; SB = 10; SKIP = 5; MAXSKIP = 0; LITOVH = 3; bigoff = false; LSHL = 16; LPRIME = KC_PRIME6; }
            else { LB = 10; SB = 8; SKIP = 4; MAXSKIP = 0; LITOVH = 3; bigoff = false; LSHL = 16; LPRIME = KC_PRIME6; smallRep = !SNB; }
        }
        auto skipOf = [&](int dist) -> int { const int k = (dist >> SKIP) + 1; return MAXSKIP != 0 && k > MAXSKIP ? MAXSKIP : k; };  // nextS - s
        uint32_t* __restrict__ ltab = tab;
        uint32_t* __restrict__ stab = tab + (1u << LB);
        const int PB = bits_len32((uint32_t)len);
        const int TB = (32 - PB) > 16 ? 16 : (32 - PB);
        const uint32_t posMask = (1u << PB) - 1u;
        auto tagOf = [&](uint32_t v) -> uint32_t { return (v * 2654435761u) >> (32 - TB); };
        auto mk = [&](int pos, uint32_t val) -> uint32_t { return (uint32_t)pos | (tagOf(val) << PB); };
        auto hL = [&](uint64_t v) -> uint32_t { return (uint32_t)(((v << LSHL) * LPRIME) >> (64 - LB)); };  // hash7 (hash6 in the small assembly classes)
        auto hS = [&](uint64_t v) -> uint32_t { return ((uint32_t)v * KC_PRIME4) >> (32 - SB); };                      // hash4
        const int sLimit = len - 8;
        const int sLimT = AX ? sLimit - 1 : sLimit;
        const int dstLimit = (len - OM) - (len >> 5);
        const int bailLim = AX ? dstLimit - LITOVH - 1 : dstLimit;
        const int cpLim = AX ? dstLimit - 1 : dstLimit;
        int nextEmit = 0, s = 1, repeat = 0;
        bool fin = false;
        int W = G;
        while (!fin && !stored) {
            KC_EMU_SYNC();  // (lane 0's table stores behind a match precede the next round's lookups)
            const int d0 = s - nextEmit;
            const int k0 = d0 >> SKIP;
            const int step = skipOf(d0);
            const int p = s + lig * step;
            const bool inseg = lig == 0 || ((d0 + (lig - 1) * step) >> SKIP) == k0;
            const int nextS = p + skipOf(p - nextEmit);
            const bool valid = lig < W && inseg && nextS <= sLimT;
            const bool term = inseg && nextS > sLimT;  // this step would `goto emitRemainder`
            uint64_t cv = 0;
            uint32_t hl = 0xFFFFFFF0u, hs = 0xFFFFFFF1u, eL = 0, eS = 0;
            if (valid) {
                cv = ld64(src + p);
                hl = hL(cv);
                hs = hS(cv);
                eL = ltab[hl];
                eS = stab[hs];
            }
            bool dep = false;
#pragma unroll
            for (int dd = 1; dd < G; dd++) {
                const uint32_t al = (uint32_t)__shfl_up((int)hl, dd, G), as = (uint32_t)__shfl_up((int)hs, dd, G);
                if (lig >= dd && (al == hl || as == hs)) dep = true;
            }
            int kind = 0, cand = 0;  // 1: 8 bytes long, 2: 8 bytes short, 3: 4 bytes long, 4: 4 bytes short (lazy long lookup at s+1 follows)
            if (valid) {
                const int cL = (int)(eL & posMask), cS = (int)(eS & posMask);
                const bool okL = eL == 0 || (eL >> PB) == tagOf((uint32_t)cv);
                const bool okS = eS == 0 || (eS >> PB) == tagOf((uint32_t)cv);
                const uint64_t vL = okL ? ld64(src + cL) : ~cv;
                const uint64_t vS = okS ? ld64(src + cS) : ~cv;
                if ((AX || !SNB) && cv == vL) { kind = 1; cand = cL; }
                else if ((AX || !SNB) && cv == vS) { kind = 2; cand = cS; }
                else if ((uint32_t)cv == (uint32_t)vL) { kind = 3; cand = cL; }
                else if ((uint32_t)cv == (uint32_t)vS) { kind = 4; cand = cS; }
            }
            const uint32_t vm = s2g_ballot(valid, grp);
            const uint32_t tm = s2g_ballot(term, grp);
            const uint32_t depm = s2g_ballot(valid && dep, grp);
            const uint32_t hm = s2g_ballot(kind != 0, grp);
            const int nvalid = __popc(vm);
            const int c = depm ? __builtin_ctz(depm) : G;
            const uint32_t hmc = hm & ((1u << c) - 1u);
            const bool found = hmc != 0;
            const int f = found ? __builtin_ctz(hmc) : 0;
            const int commitUpTo = found ? f : ((c < nvalid ? c : nvalid) - 1);
            if (valid && lig <= commitUpTo) {
                const uint32_t e = mk(p, (uint32_t)cv);
                ltab[hl] = e;
                stab[hs] = e;
            }
            if (!found) {
                W = P.spec_grow == 0 ? W : (P.spec_grow == 1 ? (W + 1 < G ? W + 1 : G) : ((2 * W < G) ? 2 * W : G));
                if (c < nvalid) {
                    s = s + c * step;
                } else if (nvalid < G && (tm & (1u << nvalid))) {
                    fin = true;
                } else {
                    const int pl = s + (nvalid - 1) * step;  // nvalid >= 1: lane 0 is valid or terminates
                    s = pl + skipOf(pl - nextEmit);
                }
                continue;
            }
            W = P.spec_w0b;
            const int mkind = (int)s2g_bcast32((uint32_t)kind, grp, f);
            int candidate = (int)s2g_bcast32((uint32_t)cand, grp, f);
            const int ps = s + f * step;
            const int nextSw = ps + skipOf(ps - nextEmit);
            s = ps;
            if (mkind == 4) {
                // try a long candidate at s+1 (:186-196); the lookup stores s+1 and observes this round's committed writes
                const uint64_t cv1 = s2g_bcast64(cv, grp, f) >> 8;
                const uint32_t hn = hL(cv1);
                const uint32_t en = ltab[hn];
                KC_EMU_SYNC();
                if (lig == 0) ltab[hn] = mk(s + 1, (uint32_t)cv1);
                const int cn = (int)(en & posMask);
                const bool okn = en == 0 || (en >> PB) == tagOf((uint32_t)cv1);
                if (okn && ld32(src + cn) == (uint32_t)cv1) { s++; candidate = cn; }
            }
            {
                int kmax = candidate;  // candidateL > 0 && s > nextEmit
                if (s - nextEmit < kmax) kmax = s - nextEmit;
                const int back = grp_backlen<S2G>(src, s, candidate, kmax, lig, grp);
                candidate -= back;
                s -= back;
            }
            if (d + (s - nextEmit) > bailLim) { stored = true; continue; }
            const int base = s;
            const int offset = base - candidate;
            const int l = 4 + grp_matchlen<S2G>(src, s + 4, candidate + 4, len - (s + 4), lig, grp);
            s = base + l;
            if (bigoff && offset > 65535 && l <= 5 && ((AX && SNB) || repeat != offset)) {  // the match is equal or worse to the encoding (:221-229)
                s = nextSw + 1;
                if (!AX && s >= sLimit) fin = true;  // (the assembly goes back to the search loop, whose nextS test ends the block)
                continue;
            }
            d += emit_lit(nextEmit, base - nextEmit);
            if (SNB) {
                if (lig == 0) s2_emit_copy_nr1(dst + d, offset, l);
                d += s2_copy_nr_size(offset, l);
                repeat = offset;
            } else if (repeat == offset) {
                d += emit_repeat(offset, l);
            } else {
                d += emit_copy(offset, l);
                repeat = offset;
            }
            nextEmit = s;
            if (s >= sLimit) { fin = true; continue; }
            if (d > cpLim) { stored = true; continue; }
            // index short & long at base+1 and s-2 (:252-262), in program order on one lane
            int index0 = base + 1, index1 = s - 2;
            {
                const uint64_t cv0 = ld64(src + index0), cv1 = ld64(src + index1);
                if (lig == 0) {
                    ltab[hL(cv0)] = mk(index0, (uint32_t)cv0);
                    stab[hS(cv0 >> 8)] = mk(index0 + 1, (uint32_t)(cv0 >> 8));
                    ltab[hL(cv1)] = mk(index1, (uint32_t)cv1);
                    stab[hS(cv1 >> 8)] = mk(index1 + 1, (uint32_t)(cv1 >> 8));
                }
            }
            index0 += 1;
            index1 -= 1;
            // long values sparsely in between, from two starting points (:266-274): the j-th store of the reference's loop is
            // position (j odd ? index2 : index0) + 2*(j/2).  G stores per pass, one per lane; inside a pass only the LAST store to a
            // bucket is issued, passes follow in program order: the table ends as after the sequential loop.
            const int index2 = (index0 + index1 + 1) >> 1;
            const int iters = index2 < index1 ? (index1 - index2 + 1) >> 1 : 0;
            for (int j0 = 0; j0 < 2 * iters; j0 += G) {
                const int j = j0 + lig;
                const bool act = j < 2 * iters;
                const int pos = ((j & 1) ? index2 : index0) + (j >> 1) * 2;
                uint64_t v = 0;
                uint32_t h = 0xFFFFFF00u + (uint32_t)lig;
                if (act) { v = ld64(src + pos); h = hL(v); }
                bool later = false;
#pragma unroll
                for (int dd = 1; dd < G; dd++) {
                    const uint32_t bh = (uint32_t)__shfl_down((int)h, dd, G);
                    if (lig + dd < G && bh == h) later = true;
                }
                if (act && !later) ltab[h] = mk(pos, (uint32_t)v);
            }
        }
        if (!stored) {
            if (AX || nextEmit < len) {  // emitRemainder (:277-284); the assembly tests the bail-out even with nothing left
                if (d + len - nextEmit > bailLim) stored = true;
                else if (nextEmit < len) d += emit_lit(nextEmit, len - nextEmit);
            }
        }
    }
    if (RING && !stored && flushedQ < q0 + d) {  // the open line: the slot is a whole number of lines, bytes past the stream are never read
        __builtin_amdgcn_wave_barrier();
        st64(slot + flushedQ + 8 * lig, *(const uint64_t*)(oring + ((flushedQ + 8 * lig) & (ORING - 1))));
    }
    KC_EMU_SYNC();  // (an abandoned attempt's stores, lane 0's among them, precede the stored form's: program order on the hardware)
    if (!P.framed) {
        if (stored) d = s2_emit_literal<S2G>(dst, src, len, lig);  // encode.go:44-55: not compressible -> one literal
        if (lig == 0) P.out_size[bi] = (uint32_t)(hdr + d);
        return;
    }
    // ---- s2.Writer chunk (s2/writer.go:414-451) ----
    uint32_t chunkLen;
    uint8_t chunkType;
    if (stored) {  // encodeBlock returned 0: uncompressed chunk, raw copy
        for (int k = lig; k < len; k += G) out[k] = src[k];
        chunkType = 0x01;
        chunkLen = 4u + (uint32_t)len;
    } else {
        chunkType = 0x00;
        chunkLen = 4u + (uint32_t)hdr + (uint32_t)d;
    }
    if (lig == 0) {
        const uint32_t c = s2_crc32c(src, len, crcT);
        const uint32_t checksum = ((c >> 15) | (c << 17)) + 0xa282ead8u;
        slot[0] = chunkType;
        slot[1] = (uint8_t)chunkLen; slot[2] = (uint8_t)(chunkLen >> 8); slot[3] = (uint8_t)(chunkLen >> 16);
        slot[4] = (uint8_t)checksum; slot[5] = (uint8_t)(checksum >> 8); slot[6] = (uint8_t)(checksum >> 16); slot[7] = (uint8_t)(checksum >> 24);
        P.out_size[bi] = 4u + chunkLen;
    }
}

void kc_launch_s2_encode(const KcS2Params& P, hipStream_t st) {
    if (P.n_blocks == 0) return;
    if (P.level == 1) hipLaunchKernelGGL(kc_s2_encode_kernel<1>, dim3((P.n_blocks + 7) / 8), dim3(64), 0, st, P);
    else if (P.level == 2) hipLaunchKernelGGL(kc_s2_encode_kernel<2>, dim3((P.n_blocks + 7) / 8), dim3(64), 0, st, P);
    else if (P.level == 3) hipLaunchKernelGGL(kc_s2_encode_kernel<3>, dim3((P.n_blocks + 7) / 8), dim3(64), 0, st, P);
    else hipLaunchKernelGGL(kc_s2_encode_kernel<0>, dim3((P.n_blocks + 7) / 8), dim3(64), 0, st, P);
}
