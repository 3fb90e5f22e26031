// kc_dev.h — device-side helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KC_WAVE 64

// ---- unaligned little-endian loads (reference: internal/le/unsafe_enabled.go) ----
struct __attribute__((packed)) kc_u64u { uint64_t v; };
struct __attribute__((packed)) kc_u32u { uint32_t v; };
struct __attribute__((packed)) kc_u16u { uint16_t v; };
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { return ((const kc_u64u*)p)->v; }
__device__ __forceinline__ void st64(uint8_t* p, uint64_t v) { ((kc_u64u*)p)->v = v; }  // unaligned 8-byte store
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return ((const kc_u32u*)p)->v; }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { ((kc_u32u*)p)->v = v; }  // unaligned 4-byte store
__device__ __forceinline__ uint16_t ld16(const uint8_t* p) { return ((const kc_u16u*)p)->v; }
struct __attribute__((packed)) kc_u128u { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 ld128u(const uint8_t* p) {  // unaligned 16-byte load
    const kc_u128u v = *(const kc_u128u*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Lanes of one wave communicating through LDS: the hardware runs a wave's LDS instructions in program order, all lanes
// of one instruction before the next, so no instruction is needed — only the compiler has to keep the order (and the
// CPU emulator of the tests, tools/hipemu, which runs lanes out of lockstep, has to synchronise them here).
#ifdef KC_HIPEMU
#define KC_WAVE_SYNC() hipemu::wave_sync()
#else
#define KC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

// ---- wave primitives ----
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ int ctz64(uint64_t v) { return __builtin_ctzll(v); }
// Block partition of a unit.  Regular: a block every block_size bytes (EncodeAll; Write ... Close).  With blk_start (streams
// with Flush points, zstd/encoder.go:547-570: Flush ends the block being filled) the host lists every block's start and says
// per unit whether a block was written before Close (stream frame) and whether Close found nothing buffered (empty last block).
struct KcUnitBlocks { int nblk; bool streamU; bool emptyLast; };
__device__ __forceinline__ KcUnitBlocks kc_unit_blocks(const uint32_t* blk_start, const uint32_t* unit_flags, const uint32_t* unit_blk0,
                                                       uint32_t u, int ulen, int bs, int stream_mode) {
    KcUnitBlocks r;
    if (blk_start != nullptr) {
        const uint32_t f = unit_flags[u];
        r.nblk = (int)(unit_blk0[u + 1] - unit_blk0[u]);
        r.streamU = (f & 1u) != 0u;
        r.emptyLast = (f & 2u) != 0u;
    } else {
        r.nblk = (ulen + bs - 1) / bs;
        r.streamU = stream_mode != 0 && ulen >= bs;
        r.emptyLast = r.streamU && (ulen % bs) == 0;
    }
    return r;
}
__device__ __forceinline__ int kc_blk_begin(const uint32_t* blk_start, uint32_t blk0, int b, int bs) {
    return blk_start != nullptr ? (int)blk_start[blk0 + (uint32_t)b] : b * bs;
}
__device__ __forceinline__ int kc_blk_end(const uint32_t* blk_start, uint32_t blk0, int b, int nblk, int bs, int ulen) {
    if (blk_start != nullptr) return b + 1 < nblk ? (int)blk_start[blk0 + (uint32_t)b + 1u] : ulen;
    const int e = (b + 1) * bs;
    return e < ulen ? e : ulen;
}

__device__ __forceinline__ uint32_t bcast32(uint32_t v, int srcLane) { return (uint32_t)__shfl((int)v, srcLane, 64); }
__device__ __forceinline__ uint64_t bcast64(uint64_t v, int srcLane) {
    uint32_t lo = bcast32((uint32_t)v, srcLane), hi = bcast32((uint32_t)(v >> 32), srcLane);
    return ((uint64_t)hi << 32) | lo;
}
// Value of lane `srcLane` when srcLane is the SAME in every lane (wave-uniform): v_readlane_b32, a few cycles, and the result is
// a scalar to the compiler — state derived from it stays in SGPRs and branches on it are scalar branches.  (bcast32/__shfl is a
// ds_bpermute round trip and its result counts as divergent.)
__device__ __forceinline__ uint32_t rdlane32(uint32_t v, int srcLane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, srcLane); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, int srcLane) {
    return (uint64_t)rdlane32((uint32_t)v, srcLane) | ((uint64_t)rdlane32((uint32_t)(v >> 32), srcLane) << 32);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// bits.Len32(v)
__device__ __forceinline__ int bits_len32(uint32_t v) { return v == 0 ? 0 : 32 - __builtin_clz(v); }
// highBit: uint32(bits.Len32(v) - 1); highBit(0) == 0xFFFFFFFF (zstd/seqenc.go:44)
__device__ __forceinline__ uint32_t high_bit(uint32_t v) { return (uint32_t)(bits_len32(v) - 1); }

// ---- zstd hashLen (zstd/hash.go:20-35) ----
#define KC_PRIME4 2654435761u
#define KC_PRIME5 889523592379ULL
#define KC_PRIME6 227718039650203ULL
#define KC_PRIME8 0xcf1bbcdcb7a56463ULL
__device__ __forceinline__ uint32_t hash4(uint32_t u, int bits) { return (u * KC_PRIME4) >> (32 - bits); }
__device__ __forceinline__ uint32_t hash5(uint64_t u, int bits) { return (uint32_t)(((u << 24) * KC_PRIME5) >> (64 - bits)); }
__device__ __forceinline__ uint32_t hash6(uint64_t u, int bits) { return (uint32_t)(((u << 16) * KC_PRIME6) >> (64 - bits)); }
__device__ __forceinline__ uint32_t hash8(uint64_t u, int bits) { return (uint32_t)((u * KC_PRIME8) >> (64 - bits)); }

// ---- sequence packing in HBM scratch: of:24 | ml:20 | ll:20 ----
// of = offset code value (real offset + 3, or 1 for repeat), ml = matchLen - 3, ll = litLen
__device__ __forceinline__ uint64_t seq_pack(uint32_t ll, uint32_t ml, uint32_t of) {
    return (uint64_t)of | ((uint64_t)ml << 24) | ((uint64_t)ll << 44);
}
__device__ __forceinline__ uint32_t seq_of(uint64_t s) { return (uint32_t)(s & 0xFFFFFF); }
__device__ __forceinline__ uint32_t seq_ml(uint64_t s) { return (uint32_t)((s >> 24) & 0xFFFFF); }
__device__ __forceinline__ uint32_t seq_ll(uint64_t s) { return (uint32_t)(s >> 44); }

// ---- wave-cooperative common-prefix length (zstd/matchlen_generic.go:17-37) ----
// a has `left` readable bytes, b (earlier in the buffer) is at least as long.
// All 64 lanes must call; returns the same value in every lane.
__device__ __forceinline__ int wave_matchlen(const uint8_t* a, const uint8_t* b, int left, int lane) {
    int n = 0;
    int width = 8;  // first probe 8 lanes (64 B): most matches are short; then full wave
    for (;;) {
        int words = (left - n) >> 3;
        int active = words < width ? words : width;
        uint64_t diff = 0;
        if (lane < active) diff = ld64(a + n + 8 * lane) ^ ld64(b + n + 8 * lane);
        uint64_t m = ballot64(diff != 0);
        if (m) {
            int fl = ctz64(m);
            uint64_t d = rdlane64(diff, fl);
            return n + 8 * fl + (ctz64(d) >> 3);
        }
        n += 8 * active;
        if (active < width) break;
        width = 64;
    }
    int tail = left - n;  // < 8 bytes
    bool ne = lane < tail && a[n + lane] != b[n + lane];
    uint64_t m = ballot64(ne);
    return n + (m ? ctz64(m) : tail);
}

// Number of consecutive k = 1..kmax with base[t-k] == base[s-k] (backward extension loops,
// e.g. zstd/enc_fast.go:230-234).  Wave-uniform result.
__device__ __forceinline__ int wave_backlen(const uint8_t* base, int s, int t, int kmax, int lane) {
    int cnt = 0;
    while (cnt < kmax) {
        int k = cnt + lane + 1;
        bool ne = true;
        if (k <= kmax) ne = base[t - k] != base[s - k];
        uint64_t m = ballot64(ne);
        int c = m ? ctz64(m) : 64;
        cnt += c;
        if (c < 64) break;
    }
    return cnt < kmax ? cnt : kmax;
}

// ---- sub-wave group primitives: G lanes per work item, 64/G items per wave ----
template <int G>
__device__ __forceinline__ uint32_t gballot(bool p, int grp) {
    return (uint32_t)((ballot64(p) >> (grp * G)) & ((1ull << G) - 1ull));
}
template <int G>
__device__ __forceinline__ uint64_t gbcast64(uint64_t v, int grp, int srcLig) {
    const int src = grp * G + srcLig;
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
template <int G>
__device__ __forceinline__ uint32_t gbcast32(uint32_t v, int grp, int srcLig) {
    return (uint32_t)__shfl((int)v, grp * G + srcLig, 64);
}
template <int G>
__device__ __forceinline__ int grp_matchlen(const uint8_t* __restrict__ base, int a, int b, int left, int lig, int grp) {
    int n = 0;
    for (;;) {
        const int words = (left - n) >> 3;
        const int active = words < G ? words : G;
        uint64_t diff = 0;
        if (lig < active) diff = ld64(base + a + n + 8 * lig) ^ ld64(base + b + n + 8 * lig);
        const uint32_t m = gballot<G>(diff != 0, grp);
        if (m) {
            const int fl = __builtin_ctz(m);
            const uint64_t d = gbcast64<G>(diff, grp, fl);
            return n + 8 * fl + (ctz64(d) >> 3);
        }
        n += 8 * active;
        if (active < G) break;
    }
    const int tail = left - n;  // < 8
    for (int k0 = 0; k0 < tail; k0 += G) {
        const int k = k0 + lig;
        const bool ne = k < tail && base[a + n + k] != base[b + n + k];
        const uint32_t m = gballot<G>(ne, grp);
        if (m) return n + k0 + __builtin_ctz(m);
    }
    return n + tail;
}
template <int G>
__device__ __forceinline__ int grp_backlen(const uint8_t* __restrict__ base, int s, int t, int kmax, int lig, int grp) {
    int cnt = 0;
    while (cnt < kmax) {
        const int k = cnt + lig + 1;
        bool ne = true;
        if (k <= kmax) ne = base[t - k] != base[s - k];
        const uint32_t m = gballot<G>(ne, grp);
        const int c = m ? __builtin_ctz(m) : G;
        cnt += c;
        if (c < G) break;
    }
    return cnt < kmax ? cnt : kmax;
}


// Per-block record written by the match finders and consumed by the entropy kernel.
struct KcBlkMeta {
    uint32_t nseq;       // sequences in this block
    uint32_t nlit;       // len(blk.literals) == sum(litLen) + extraLits
    uint32_t extra_lits; // trailing literals after the last sequence
    uint32_t flags;      // KC_BF_*
    uint32_t o1_in, o2_in;   // recentOffsets[0..1] before the block (what popOffsets restores)
    uint32_t o1_out, o2_out; // recentOffsets[0..1] the match finder carried into the next block
};
#define KC_BF_POP_A 1u   // block took the saved<16 literals-only path (blockenc.go:496-503): offsets popped
#define KC_BF_FORCED 2u  // pop forced by the host for a re-run (block re-emitted raw, blockenc.go:811-817)
