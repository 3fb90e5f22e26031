// kc_zfast_dev.h — helpers shared by the SpeedFastest match finders (kc_zstd_match.hip: HBM tables, 8 lanes per unit;
// kc_zstd_match_lds.hip: LDS tables, one wave per unit).
#pragma once
#include "kc_dev.h"

// The emulator only (tools/hipemu runs lanes out of lockstep): an exchange through LDS that the hardware orders by itself, in a
// kernel where KC_WAVE_SYNC's compiler fence is not wanted.
#ifndef KC_EMU_SYNC
#ifdef KC_HIPEMU
#define KC_EMU_SYNC() hipemu::wave_sync()
#else
#define KC_EMU_SYNC() do { } while (0)
#endif
#endif

#define ZF_TABLE_BITS 15
#define ZF_MAX_MATCH_LENGTH 131074  // enc_fast.go:18

// c = 16 bytes at [t-4, t+12), p0..p3 = the 16 bytes at [p'-4, p'+12):
// fwd = equal bytes from t / p' on (0..12), back = equal bytes going down from t-1 / p'-1 (0..4).
__device__ __forceinline__ void zf_cmp16(const uint4 c, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, int& fwd, int& back) {
    const uint32_t x0 = c.x ^ p0, x1 = c.y ^ p1, x2 = c.z ^ p2, x3 = c.w ^ p3;
    back = x0 ? (__builtin_clz(x0) >> 3) : 4;
    fwd = x1 ? (__builtin_ctz(x1) >> 3) : (x2 ? 4 + (__builtin_ctz(x2) >> 3) : (x3 ? 8 + (__builtin_ctz(x3) >> 3) : 12));
}

// 4 bytes at q, with bytes outside [lo, hi) read as zero (edges of the caller's buffer only).
static __device__ __noinline__ uint32_t zf_edge_dword(const uint8_t* q, const uint8_t* lo, const uint8_t* hi) {
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) {
        const uint8_t* a = q + k;
        if (a >= lo && a < hi) v |= (uint32_t)(*a) << (8 * k);
    }
    return v;
}

