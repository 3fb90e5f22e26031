// kc_s2_dev.h — S2 emit helpers shared by the S2 kernels (kc_s2.hip: HBM tables, 8 lanes per block; kc_s2_lds.hip:
// LDS tables, one wave per block).  emitLiteral / emitRepeat / emitCopy / emitCopyNoRepeat: s2/encode_go.go:80-290.
#pragma once
#include "kc_dev.h"

#define S2_TABLE_BITS 14
// ---- emit helpers (group-uniform arguments; lane 0 writes tag bytes, all lanes copy literals) ----
template <int G>
__device__ __forceinline__ int s2_emit_literal(uint8_t* __restrict__ dst, const uint8_t* __restrict__ lit, int len, int lig) {
    if (len == 0) return 0;
    const uint32_t n = (uint32_t)(len - 1);
    int i;
    if (n < 60) { i = 1; if (lig == 0) dst[0] = (uint8_t)(n << 2); }
    else if (n < (1u << 8)) { i = 2; if (lig == 0) { dst[0] = 60 << 2; dst[1] = (uint8_t)n; } }
    else if (n < (1u << 16)) { i = 3; if (lig == 0) { dst[0] = 61 << 2; dst[1] = (uint8_t)n; dst[2] = (uint8_t)(n >> 8); } }
    else if (n < (1u << 24)) { i = 4; if (lig == 0) { dst[0] = 62 << 2; dst[1] = (uint8_t)n; dst[2] = (uint8_t)(n >> 8); dst[3] = (uint8_t)(n >> 16); } }
    else { i = 5; if (lig == 0) { dst[0] = 63 << 2; dst[1] = (uint8_t)n; dst[2] = (uint8_t)(n >> 8); dst[3] = (uint8_t)(n >> 16); dst[4] = (uint8_t)(n >> 24); } }
    // 8 bytes per lane and pass (unaligned 8-byte loads/stores), then the tail bytewise: a byte per lane would be one
    // memory instruction per 8 bytes of literals
    const int body = len & ~7;
    for (int k = lig * 8; k < body; k += G * 8) st64(dst + i + k, ld64(lit + k));
    for (int k = body + lig; k < len; k += G) dst[i + k] = lit[k];
    return i + len;
}
// emitRepeat (encode_go.go:118) through a byte sink put(i, byte); one lane calls it.  Returns bytes.
template <class Put>
__device__ __forceinline__ int s2_put_repeat(Put put, int offset, int length) {
    int total = 0;
    for (;;) {
        length -= 4;
        if (length <= 4) { put(total + 0, (uint8_t)((uint32_t)length << 2 | 1)); put(total + 1, (uint8_t)0); return total + 2; }
        if (length < 8 && offset < 2048) { put(total + 1, (uint8_t)offset); put(total + 0, (uint8_t)((uint32_t)(offset >> 8) << 5 | (uint32_t)length << 2 | 1)); return total + 2; }
        if (length < (1 << 8) + 4) { length -= 4; put(total + 2, (uint8_t)length); put(total + 1, (uint8_t)0); put(total + 0, (uint8_t)(5 << 2 | 1)); return total + 3; }
        if (length < (1 << 16) + (1 << 8)) { length -= 1 << 8; put(total + 3, (uint8_t)(length >> 8)); put(total + 2, (uint8_t)length); put(total + 1, (uint8_t)0); put(total + 0, (uint8_t)(6 << 2 | 1)); return total + 4; }
        const int maxRepeat = (1 << 24) - 1;
        length -= 1 << 16;
        int left = 0;
        if (length > maxRepeat) { left = length - maxRepeat + 4; length = maxRepeat - 4; }
        put(total + 4, (uint8_t)(length >> 16)); put(total + 3, (uint8_t)(length >> 8)); put(total + 2, (uint8_t)length); put(total + 1, (uint8_t)0); put(total + 0, (uint8_t)(7 << 2 | 1));
        total += 5;
        if (left <= 0) return total;
        length = left;  // tail call emitRepeat(dst[5:], offset, left)
    }
}
// emitCopy (encode_go.go:172) through a byte sink; one lane calls it.
template <class Put>
__device__ __forceinline__ int s2_put_copy(Put put, int offset, int length) {
    if (offset >= 65536) {
        int i = 0;
        if (length > 64) {
            put(4, (uint8_t)(offset >> 24)); put(3, (uint8_t)(offset >> 16)); put(2, (uint8_t)(offset >> 8)); put(1, (uint8_t)offset); put(0, (uint8_t)(63 << 2 | 3));
            length -= 64;
            if (length >= 4) return 5 + s2_put_repeat([&](int k, uint8_t v) { put(5 + k, v); }, offset, length);
            i = 5;
        }
        if (length == 0) return i;
        put(i + 0, (uint8_t)((uint32_t)(length - 1) << 2 | 3));
        put(i + 1, (uint8_t)offset); put(i + 2, (uint8_t)(offset >> 8)); put(i + 3, (uint8_t)(offset >> 16)); put(i + 4, (uint8_t)(offset >> 24));
        return i + 5;
    }
    if (length > 64) {
        int off = 3;
        if (offset < 2048) {
            put(1, (uint8_t)offset); put(0, (uint8_t)((uint32_t)(offset >> 8) << 5 | (uint32_t)(8 - 4) << 2 | 1));
            length -= 8;
            off = 2;
        } else {
            put(2, (uint8_t)(offset >> 8)); put(1, (uint8_t)offset); put(0, (uint8_t)(59 << 2 | 2));
            length -= 60;
        }
        return off + s2_put_repeat([&](int k, uint8_t v) { put(off + k, v); }, offset, length);
    }
    if (length >= 12 || offset >= 2048) {
        put(2, (uint8_t)(offset >> 8)); put(1, (uint8_t)offset); put(0, (uint8_t)((uint32_t)(length - 1) << 2 | 2));
        return 3;
    }
    put(1, (uint8_t)offset); put(0, (uint8_t)((uint32_t)(offset >> 8) << 5 | (uint32_t)(length - 4) << 2 | 1));
    return 2;
}
__device__ inline int s2_emit_repeat1(uint8_t* dst, int offset, int length) { return s2_put_repeat([&](int i, uint8_t v) { dst[i] = v; }, offset, length); }
__device__ inline int s2_emit_copy1(uint8_t* dst, int offset, int length) { return s2_put_copy([&](int i, uint8_t v) { dst[i] = v; }, offset, length); }
// emitCopyNoRepeat (encode_go.go:241): the Snappy-compatible copy encoding; single lane writes.
__device__ inline int s2_emit_copy_nr1(uint8_t* dst, int offset, int length) {
    int total = 0;
    for (;;) {
        if (offset >= 65536) {
            int i = 0;
            if (length > 64) {
                dst[4] = (uint8_t)(offset >> 24); dst[3] = (uint8_t)(offset >> 16); dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset; dst[0] = 63 << 2 | 3;
                length -= 64;
                if (length >= 4) { dst += 5; total += 5; continue; }  // tail call emitCopyNoRepeat(dst[5:], offset, length)
                i = 5;
            }
            if (length == 0) return total + i;
            dst[i + 0] = (uint8_t)((uint32_t)(length - 1) << 2 | 3);
            dst[i + 1] = (uint8_t)offset; dst[i + 2] = (uint8_t)(offset >> 8); dst[i + 3] = (uint8_t)(offset >> 16); dst[i + 4] = (uint8_t)(offset >> 24);
            return total + i + 5;
        }
        if (length > 64) {
            dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset; dst[0] = 59 << 2 | 2;
            length -= 60;
            dst += 3; total += 3;
            continue;
        }
        if (length >= 12 || offset >= 2048) {
            dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset; dst[0] = (uint8_t)((uint32_t)(length - 1) << 2 | 2);
            return total + 3;
        }
        dst[1] = (uint8_t)offset; dst[0] = (uint8_t)((uint32_t)(offset >> 8) << 5 | (uint32_t)(length - 4) << 2 | 1);
        return total + 2;
    }
}
__device__ inline int s2_copy_nr_size(int offset, int length) {
    int total = 0;
    for (;;) {
        if (offset >= 65536) {
            int i = 0;
            if (length > 64) { length -= 64; if (length >= 4) { total += 5; continue; } i = 5; }
            if (length == 0) return total + i;
            return total + i + 5;
        }
        if (length > 64) { length -= 60; total += 3; continue; }
        if (length >= 12 || offset >= 2048) return total + 3;
        return total + 2;
    }
}
// sizes without writing (every lane of the group needs the byte count; only lane 0 writes)
__device__ inline int s2_repeat_size(int offset, int length) {
    int total = 0;
    for (;;) {
        length -= 4;
        if (length <= 4) return total + 2;
        if (length < 8 && offset < 2048) return total + 2;
        if (length < (1 << 8) + 4) return total + 3;
        if (length < (1 << 16) + (1 << 8)) return total + 4;
        const int maxRepeat = (1 << 24) - 1;
        length -= 1 << 16;
        int left = 0;
        if (length > maxRepeat) { left = length - maxRepeat + 4; length = maxRepeat - 4; }
        total += 5;
        if (left <= 0) return total;
        length = left;
    }
}
__device__ inline int s2_copy_size(int offset, int length) {
    if (offset >= 65536) {
        int i = 0;
        if (length > 64) { length -= 64; if (length >= 4) return 5 + s2_repeat_size(offset, length); i = 5; }
        if (length == 0) return i;
        return i + 5;
    }
    if (length > 64) {
        if (offset < 2048) return 2 + s2_repeat_size(offset, length - 8);
        return 3 + s2_repeat_size(offset, length - 60);
    }
    if (length >= 12 || offset >= 2048) return 3;
    return 2;
}

__device__ __forceinline__ uint32_t s2_hash6(uint64_t u) { return (uint32_t)(((u << 16) * KC_PRIME6) >> (64 - S2_TABLE_BITS)); }

// CRC32C (Castagnoli, reflected 0x82F63B78), slicing-by-4 tables in LDS; masked as s2.crc (s2/s2.go:120-125).
__device__ __forceinline__ uint32_t s2_crc32c(const uint8_t* __restrict__ p, int n, const uint32_t (*T)[256]) {
    uint32_t c = 0xFFFFFFFFu;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        c ^= ld32(p + i);
        c = T[3][c & 0xFF] ^ T[2][(c >> 8) & 0xFF] ^ T[1][(c >> 16) & 0xFF] ^ T[0][c >> 24];
    }
    for (; i < n; i++) c = T[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
