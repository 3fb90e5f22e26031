// oracle/kco_s2.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates the portable-Go ("noasm") S2 block encoder: s2/encode.go:29-56,389-418,
// s2/encode_all.go:17-30,72-500 (encodeBlockGo / encodeBlockGo64K), s2/encode_go.go:19-27,
// 80-234 (emitLiteral/emitRepeat/emitCopy), s2/s2.go:120-147 (crc, literalExtraSize),
// plus a bounds-checked block decoder following s2/decode_other.go:22-260 used only to
// verify that oracle/GPU output round-trips.
// PARITY TARGET: encodeBlockGo64K / encodeBlockGo (build tag noasm) — NOT the amd64 asm
// variant.
#pragma once
#include "kco_common.h"

namespace kco {
namespace s2 {

constexpr int tagLiteral = 0x00, tagCopy1 = 0x01, tagCopy2 = 0x02, tagCopy4 = 0x03;
constexpr int inputMargin = 8, minNonLiteralBlockSize = 32;

// s2/s2.go:129 literalExtraSize
static inline int64_t literalExtraSize(int64_t n) {
    if (n == 0) return 0;
    if (n < 60) return 1;
    if (n < (1 << 8)) return 2;
    if (n < (1 << 16)) return 3;
    if (n < (1 << 24)) return 4;
    return 5;
}
// s2/encode.go:389 MaxEncodedLen (64-bit int)
static inline int64_t MaxEncodedLen(int64_t srcLen) {
    uint64_t n = (uint64_t)srcLen;
    if (n > 0xffffffffULL) return -1;
    n = n + (uint64_t)((bitsLen64(n) + 7) / 7);
    n += (uint64_t)literalExtraSize(srcLen);
    if (n > 0xffffffffULL) return -1;
    return (int64_t)n;
}
static inline int putUvarint(uint8_t* buf, uint64_t x) {
    int i = 0;
    while (x >= 0x80) { buf[i++] = (uint8_t)x | 0x80; x >>= 7; }
    buf[i] = (uint8_t)x;
    return i + 1;
}

// s2/encode_go.go:80 emitLiteral
static inline int emitLiteral(uint8_t* dst, const uint8_t* lit, size_t len) {
    if (len == 0) return 0;
    int i = 0;
    uint64_t n = (uint64_t)(len - 1);
    if (n < 60) { dst[0] = (uint8_t)((uint8_t)n << 2 | tagLiteral); i = 1; }
    else if (n < (1 << 8)) { dst[1] = (uint8_t)n; dst[0] = 60 << 2 | tagLiteral; i = 2; }
    else if (n < (1 << 16)) { dst[2] = (uint8_t)(n >> 8); dst[1] = (uint8_t)n; dst[0] = 61 << 2 | tagLiteral; i = 3; }
    else if (n < (1 << 24)) { dst[3] = (uint8_t)(n >> 16); dst[2] = (uint8_t)(n >> 8); dst[1] = (uint8_t)n; dst[0] = 62 << 2 | tagLiteral; i = 4; }
    else { dst[4] = (uint8_t)(n >> 24); dst[3] = (uint8_t)(n >> 16); dst[2] = (uint8_t)(n >> 8); dst[1] = (uint8_t)n; dst[0] = 63 << 2 | tagLiteral; i = 5; }
    memcpy(dst + i, lit, len);
    return i + (int)len;
}

// s2/encode_go.go:118 emitRepeat
static inline int emitRepeat(uint8_t* dst, int offset, int length) {
    length -= 4;
    if (length <= 4) { dst[0] = (uint8_t)((uint8_t)length << 2 | tagCopy1); dst[1] = 0; return 2; }
    if (length < 8 && offset < 2048) {
        dst[1] = (uint8_t)offset;
        dst[0] = (uint8_t)((uint8_t)(offset >> 8) << 5 | (uint8_t)length << 2 | tagCopy1);
        return 2;
    }
    if (length < (1 << 8) + 4) {
        length -= 4;
        dst[2] = (uint8_t)length; dst[1] = 0; dst[0] = 5 << 2 | tagCopy1;
        return 3;
    }
    if (length < (1 << 16) + (1 << 8)) {
        length -= 1 << 8;
        dst[3] = (uint8_t)(length >> 8); dst[2] = (uint8_t)length; dst[1] = 0; dst[0] = 6 << 2 | tagCopy1;
        return 4;
    }
    const int maxRepeat = (1 << 24) - 1;
    length -= 1 << 16;
    int left = 0;
    if (length > maxRepeat) { left = length - maxRepeat + 4; length = maxRepeat - 4; }
    dst[4] = (uint8_t)(length >> 16); dst[3] = (uint8_t)(length >> 8); dst[2] = (uint8_t)length; dst[1] = 0; dst[0] = 7 << 2 | tagCopy1;
    if (left > 0) return 5 + emitRepeat(dst + 5, offset, left);
    return 5;
}

// s2/encode_go.go:172 emitCopy
static inline int emitCopy(uint8_t* dst, int offset, int length) {
    if (offset >= 65536) {
        int i = 0;
        if (length > 64) {
            dst[4] = (uint8_t)(offset >> 24); dst[3] = (uint8_t)(offset >> 16); dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset;
            dst[0] = 63 << 2 | tagCopy4;
            length -= 64;
            if (length >= 4) return 5 + emitRepeat(dst + 5, offset, length);
            i = 5;
        }
        if (length == 0) return i;
        dst[i + 0] = (uint8_t)((uint8_t)(length - 1) << 2 | tagCopy4);
        dst[i + 1] = (uint8_t)offset; dst[i + 2] = (uint8_t)(offset >> 8); dst[i + 3] = (uint8_t)(offset >> 16); dst[i + 4] = (uint8_t)(offset >> 24);
        return i + 5;
    }
    if (length > 64) {
        int off = 3;
        if (offset < 2048) {
            dst[1] = (uint8_t)offset;
            dst[0] = (uint8_t)((uint8_t)(offset >> 8) << 5 | (uint8_t)(8 - 4) << 2 | tagCopy1);
            length -= 8;
            off = 2;
        } else {
            dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset; dst[0] = 59 << 2 | tagCopy2;
            length -= 60;
        }
        return off + emitRepeat(dst + off, offset, length);
    }
    if (length >= 12 || offset >= 2048) {
        dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset; dst[0] = (uint8_t)((uint8_t)(length - 1) << 2 | tagCopy2);
        return 3;
    }
    dst[1] = (uint8_t)offset;
    dst[0] = (uint8_t)((uint8_t)(offset >> 8) << 5 | (uint8_t)(length - 4) << 2 | tagCopy1);
    return 2;
}

// s2/encode_all.go:27 hash6
static inline uint32_t hash6(uint64_t u, uint8_t h) {
    const uint64_t prime6bytes = 227718039650203ULL;
    return (uint32_t)(((u << (64 - 48)) * prime6bytes) >> ((64 - h) & 63));
}

// s2/encode_go.go:241 emitCopyNoRepeat: the Snappy-compatible copy encoding (no repeat tags)
static inline int emitCopyNoRepeat(uint8_t* dst, int offset, int length) {
    if (offset >= 65536) {
        int i = 0;
        if (length > 64) {
            dst[4] = (uint8_t)(offset >> 24); dst[3] = (uint8_t)(offset >> 16); dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset;
            dst[0] = 63 << 2 | 3;
            length -= 64;
            if (length >= 4) return 5 + emitCopyNoRepeat(dst + 5, offset, length);
            i = 5;
        }
        if (length == 0) return i;
        dst[i + 0] = (uint8_t)((uint8_t)(length - 1) << 2 | 3);
        dst[i + 1] = (uint8_t)offset; dst[i + 2] = (uint8_t)(offset >> 8); dst[i + 3] = (uint8_t)(offset >> 16); dst[i + 4] = (uint8_t)(offset >> 24);
        return i + 5;
    }
    if (length > 64) {
        dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset; dst[0] = 59 << 2 | 2;
        length -= 60;
        return 3 + emitCopyNoRepeat(dst + 3, offset, length);
    }
    if (length >= 12 || offset >= 2048) {
        dst[2] = (uint8_t)(offset >> 8); dst[1] = (uint8_t)offset; dst[0] = (uint8_t)((uint8_t)(length - 1) << 2 | 2);
        return 3;
    }
    dst[1] = (uint8_t)offset;
    dst[0] = (uint8_t)((uint8_t)(offset >> 8) << 5 | (uint8_t)(length - 4) << 2 | 1);
    return 2;
}

// s2/encode_all.go:72 encodeBlockGo (T=uint32_t, SKIP=6) / :287 encodeBlockGo64K (T=uint16_t, SKIP=5);
// SNAPPY: :502 encodeBlockSnappyGo / :692 encodeBlockSnappyGo64K — the same parse, every copy through emitCopyNoRepeat
template <typename T, int SKIP, bool SNAPPY = false>
static int encodeBlockGoT(uint8_t* dst, const uint8_t* src, size_t srcLen) {
    const uint8_t tableBits = 14;
    const int maxTableSize = 1 << tableBits;
    std::vector<T> table((size_t)maxTableSize, (T)0);
    const int len = (int)srcLen;
    int sLimit = len - inputMargin;
    int dstLimit = len - (len >> 5) - 5;
    int nextEmit = 0;
    int s = 1;
    uint64_t cv = load64(src, s);
    int repeat = 1;
    int d = 0;
    for (;;) {
        int candidate = 0;
        for (;;) {
            int nextS = s + ((s - nextEmit) >> SKIP) + 4;
            if (nextS > sLimit) goto emitRemainder;
            {
                uint32_t hash0 = hash6(cv, tableBits);
                uint32_t hash1 = hash6(cv >> 8, tableBits);
                candidate = (int)table[hash0];
                int candidate2 = (int)table[hash1];
                table[hash0] = (T)s;
                table[hash1] = (T)(s + 1);
                uint32_t hash2 = hash6(cv >> 16, tableBits);
                const int checkRep = 1;
                if ((uint32_t)(cv >> (checkRep * 8)) == load32(src, s - repeat + checkRep)) {
                    int base = s + checkRep;
                    for (int i = base - repeat; base > nextEmit && i > 0 && src[i - 1] == src[base - 1];) { i--; base--; }
                    if (d + (base - nextEmit) > dstLimit) return 0;
                    d += emitLiteral(dst + d, src + nextEmit, (size_t)(base - nextEmit));
                    int cand = s - repeat + 4 + checkRep;
                    s += 4 + checkRep;
                    while (s <= sLimit) {
                        uint64_t diff = load64(src, s) ^ load64(src, cand);
                        if (diff != 0) { s += tz64(diff) >> 3; break; }
                        s += 8;
                        cand += 8;
                    }
                    if (SNAPPY) d += emitCopyNoRepeat(dst + d, repeat, s - base);
                    else if (nextEmit > 0) d += emitRepeat(dst + d, repeat, s - base);
                    else d += emitCopy(dst + d, repeat, s - base);
                    nextEmit = s;
                    if (s >= sLimit) goto emitRemainder;
                    cv = load64(src, s);
                    continue;
                }
                if ((uint32_t)cv == load32(src, candidate)) break;
                candidate = (int)table[hash2];
                if ((uint32_t)(cv >> 8) == load32(src, candidate2)) {
                    table[hash2] = (T)(s + 2);
                    candidate = candidate2;
                    s++;
                    break;
                }
                table[hash2] = (T)(s + 2);
                if ((uint32_t)(cv >> 16) == load32(src, candidate)) { s += 2; break; }
                cv = load64(src, nextS);
                s = nextS;
            }
        }
        while (candidate > 0 && s > nextEmit && src[candidate - 1] == src[s - 1]) { candidate--; s--; }
        if (d + (s - nextEmit) > dstLimit) return 0;
        d += emitLiteral(dst + d, src + nextEmit, (size_t)(s - nextEmit));
        for (;;) {
            int base = s;
            repeat = base - candidate;
            s += 4;
            candidate += 4;
            while (s <= len - 8) {
                uint64_t diff = load64(src, s) ^ load64(src, candidate);
                if (diff != 0) { s += tz64(diff) >> 3; break; }
                s += 8;
                candidate += 8;
            }
            d += SNAPPY ? emitCopyNoRepeat(dst + d, repeat, s - base) : emitCopy(dst + d, repeat, s - base);
            nextEmit = s;
            if (s >= sLimit) goto emitRemainder;
            if (d > dstLimit) return 0;
            uint64_t x = load64(src, s - 2);
            uint32_t m2Hash = hash6(x, tableBits);
            uint32_t currHash = hash6(x >> 16, tableBits);
            candidate = (int)table[currHash];
            table[m2Hash] = (T)(s - 2);
            table[currHash] = (T)s;
            if ((uint32_t)(x >> 16) != load32(src, candidate)) {
                cv = load64(src, s + 1);
                s++;
                break;
            }
        }
    }
emitRemainder:
    if (nextEmit < len) {
        if (d + len - nextEmit > dstLimit) return 0;
        d += emitLiteral(dst + d, src + nextEmit, (size_t)(len - nextEmit));
    }
    return d;
}

// s2/encode_go.go:19 encodeBlock
static inline int encodeBlock(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n < (size_t)minNonLiteralBlockSize) return 0;
    if (n <= (64 << 10)) return encodeBlockGoT<uint16_t, 5>(dst, src, n);
    return encodeBlockGoT<uint32_t, 6>(dst, src, n);
}

// s2/encode.go:29 Encode; returns bytes written or -1 (too large) / -2 (dst too small)
static inline int64_t Encode(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    int k = encodeBlock(dst + d, src, n);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

// ---- s2.EncodeBetter (s2/encode.go:117-144): encodeBlockBetterGo / encodeBlockBetterGo64K (s2/encode_better.go:50-307 / 485-730) ----
// s2/encode_better.go:15-40
static inline uint32_t hash4(uint64_t u, uint8_t h) { return ((uint32_t)u * 2654435761u) >> ((32 - h) & 31); }
static inline uint32_t hash7(uint64_t u, uint8_t h) { return (uint32_t)(((u << (64 - 56)) * 58295818150454627ULL) >> ((64 - h) & 63)); }

// One body for both variants: T = uint32/uint16 tables, LBITS/SBITS = 17/14 or 16/13, SKIP = 7 or 6,
// BIG = the "offset > 65535 && s-base <= 5 && repeat != offset" bail of the > 64 KiB variant (:221-229).
// The repeat check inside the probe loop is disabled in the reference (`if false && ...`, :124 / :561) and is not restated.
// SNAPPY = encodeBlockBetterSnappyGo / ...64K (s2/encode_better.go:310-483 / 733-900): skip capped at maxSkip = 100, candidates
// accepted on 4 equal bytes only (long, then short with the lazy long lookup), every copy through emitCopyNoRepeat.
template <typename T, int LBITS, int SBITS, int SKIP, bool BIG, bool SNAPPY = false>
static int encodeBlockBetterGoT(uint8_t* dst, const uint8_t* src, size_t srcLen) {
    const int len = (int)srcLen;
    const int sLimit = len - inputMargin;
    if (len < minNonLiteralBlockSize) return 0;
    std::vector<T> lTable((size_t)1 << LBITS, (T)0), sTable((size_t)1 << SBITS, (T)0);
    const int dstLimit = len - (len >> 5) - 6;
    int nextEmit = 0;
    int s = 1;
    uint64_t cv = load64(src, s);
    int repeat = 0;
    int d = 0;
    for (;;) {
        int candidateL = 0;
        int nextS = 0;
        for (;;) {
            nextS = s + ((s - nextEmit) >> SKIP) + 1;
            if (SNAPPY && nextS > s + 100) nextS = s + 100;  // maxSkip (:348, :356 / :775)
            if (nextS > sLimit) goto emitRemainder;
            {
                uint32_t hashL = hash7(cv, LBITS);
                const uint32_t hashS = hash4(cv, SBITS);
                candidateL = (int)lTable[hashL];
                const int candidateS = (int)sTable[hashS];
                lTable[hashL] = (T)s;
                sTable[hashS] = (T)s;
                const uint64_t valLong = load64(src, candidateL);
                const uint64_t valShort = load64(src, candidateS);
                if (!SNAPPY && cv == valLong) break;               // long matches at least 8 bytes
                if (!SNAPPY && cv == valShort) { candidateL = candidateS; break; }
                if ((uint32_t)cv == (uint32_t)valLong) break;      // long likely matches 7
                if ((uint32_t)cv == (uint32_t)valShort) {          // short candidate: try a long candidate at s+1 first
                    hashL = hash7(cv >> 8, LBITS);
                    candidateL = (int)lTable[hashL];
                    lTable[hashL] = (T)(s + 1);
                    if ((uint32_t)(cv >> 8) == load32(src, candidateL)) { s++; break; }
                    candidateL = candidateS;
                    break;
                }
            }
            cv = load64(src, nextS);
            s = nextS;
        }
        // extend backwards
        while (candidateL > 0 && s > nextEmit && src[candidateL - 1] == src[s - 1]) { candidateL--; s--; }
        if (d + (s - nextEmit) > dstLimit) return 0;
        {
            const int base = s;
            const int offset = base - candidateL;
            s += 4;
            candidateL += 4;
            while (s < len) {
                if (len - s < 8) {
                    if (src[s] == src[candidateL]) { s++; candidateL++; continue; }
                    break;
                }
                const uint64_t diff = load64(src, s) ^ load64(src, candidateL);
                if (diff != 0) { s += tz64(diff) >> 3; break; }
                s += 8;
                candidateL += 8;
            }
            if (BIG && offset > 65535 && s - base <= 5 && repeat != offset) {  // the match is equal or worse to the encoding
                s = nextS + 1;
                if (s >= sLimit) goto emitRemainder;
                cv = load64(src, s);
                continue;
            }
            d += emitLiteral(dst + d, src + nextEmit, (size_t)(base - nextEmit));
            if (SNAPPY) { d += emitCopyNoRepeat(dst + d, offset, s - base); repeat = offset; }
            else if (repeat == offset) d += emitRepeat(dst + d, offset, s - base);
            else { d += emitCopy(dst + d, offset, s - base); repeat = offset; }
            nextEmit = s;
            if (s >= sLimit) goto emitRemainder;
            if (d > dstLimit) return 0;
            // index short & long at base+1 and s-2, then long values sparsely in between from two starting points
            int index0 = base + 1;
            int index1 = s - 2;
            const uint64_t cv0 = load64(src, index0);
            const uint64_t cv1 = load64(src, index1);
            lTable[hash7(cv0, LBITS)] = (T)index0;
            sTable[hash4(cv0 >> 8, SBITS)] = (T)(index0 + 1);
            lTable[hash7(cv1, LBITS)] = (T)index1;
            sTable[hash4(cv1 >> 8, SBITS)] = (T)(index1 + 1);
            index0 += 1;
            index1 -= 1;
            cv = load64(src, s);
            int index2 = (index0 + index1 + 1) >> 1;
            while (index2 < index1) {
                lTable[hash7(load64(src, index0), LBITS)] = (T)index0;
                lTable[hash7(load64(src, index2), LBITS)] = (T)index2;
                index0 += 2;
                index2 += 2;
            }
        }
    }
emitRemainder:
    if (nextEmit < len) {
        if (d + len - nextEmit > dstLimit) return 0;
        d += emitLiteral(dst + d, src + nextEmit, (size_t)(len - nextEmit));
    }
    return d;
}

// s2/encode_go.go:36 encodeBlockBetter
static inline int encodeBlockBetter(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n <= ((size_t)64 << 10)) return encodeBlockBetterGoT<uint16_t, 16, 13, 6, false>(dst, src, n);
    return encodeBlockBetterGoT<uint32_t, 17, 14, 7, true>(dst, src, n);
}

// s2/encode_go.go:45 encodeBlockBetterSnappy: tables 2^16 + 2^14 (hashtable_pool.go:13-14), 2^15 + 2^13 up to 64 KiB
static inline int encodeBlockBetterSnappy(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n <= ((size_t)64 << 10)) return encodeBlockBetterGoT<uint16_t, 15, 13, 6, false, true>(dst, src, n);
    return encodeBlockBetterGoT<uint32_t, 16, 14, 7, true, true>(dst, src, n);
}

// s2/encode.go:248 EncodeSnappyBetter; returns bytes written or -1 / -2
static inline int64_t EncodeSnappyBetter(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    int k = encodeBlockBetterSnappy(dst + d, src, n);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

// ---- s2.EncodeBest (s2/encode_best.go:22-455 encodeBlockBest with dict == nil; size estimates :718-797) ----
// Restated for the next device level (§8(f)); no device kernel uses it yet.  prime8bytes hash (encode_better.go:37-43).
static inline uint32_t hash8(uint64_t u, uint8_t h) { return (uint32_t)((u * 0xcf1bbcdcb7a56463ULL) >> ((64 - h) & 63)); }

static inline int emitRepeatSize(int offset, int length) {  // :776-797
    if (length <= 4 + 4 || (length < 8 + 4 && offset < 2048)) return 2;
    if (length < (1 << 8) + 4 + 4) return 3;
    if (length < (1 << 16) + (1 << 8) + 4) return 4;
    const int maxRepeat = (1 << 24) - 1;
    length -= (1 << 16) - 4;
    int left = 0;
    if (length > maxRepeat) left = length - maxRepeat + 4;
    if (left > 0) return 5 + emitRepeatSize(offset, left);
    return 5;
}
static inline int emitCopySize(int offset, int length) {  // :718-749
    if (offset >= 65536) {
        int i = 0;
        if (length > 64) {
            length -= 64;
            if (length >= 4) return 5 + emitRepeatSize(offset, length);
            i = 5;
        }
        if (length == 0) return i;
        return i + 5;
    }
    if (length > 64) {
        if (offset < 2048) return 2 + emitRepeatSize(offset, length - 8);
        return 3 + emitRepeatSize(offset, length - 60);
    }
    if (length >= 12 || offset >= 2048) return 3;
    return 2;
}

static int encodeBlockBest(uint8_t* dst, const uint8_t* src, size_t srcLen) {
    const int lTableBits = 19, sTableBits = 16;  // bestLongTableBits / bestShortTableBits (hashtable_pool.go:16-20)
    const int inputMarginBest = 8 + 2;
    const int len = (int)srcLen;
    const int sLimit = len - inputMarginBest;
    if (len < minNonLiteralBlockSize) return 0;
    std::vector<uint64_t> lTable((size_t)1 << lTableBits, 0), sTable((size_t)1 << sTableBits, 0);
    const int dstLimit = len - 5;
    int nextEmit = 0;
    int s = 1;
    int repeat = 1;
    uint64_t cv = load64(src, s);
    int d = 0;
    auto getCur = [](uint64_t x) { return (int)(x & 0xffffffffULL); };
    auto getPrev = [](uint64_t x) { return (int)(x >> 32); };
    const int maxSkip = 64;
    struct match { int offset = 0, s = 0, length = 0, score = 0; bool rep = false; };
    for (;;) {
        match best;
        for (;;) {
            int nextS = ((s - nextEmit) >> 8) + 1;
            if (nextS > maxSkip) nextS = s + maxSkip; else nextS += s;
            if (nextS > sLimit) goto emitRemainder;
            {
                const uint32_t hashL = hash8(cv, lTableBits);
                const uint32_t hashS = hash4(cv, sTableBits);
                const uint64_t candidateL = lTable[hashL];
                const uint64_t candidateS = sTable[hashS];
                auto score = [&](const match& m) {
                    int sc = m.length - m.s;            // matches that start later are penalised: the bytes before go out as literals
                    if (nextEmit == m.s) sc++;          // no literals to emit: one byte saved
                    const int offset = m.s - m.offset;
                    if (m.rep) return sc - emitRepeatSize(offset, m.length);
                    return sc - emitCopySize(offset, m.length);
                };
                auto matchAt = [&](int offset, int s_, uint32_t first, bool rep) {
                    match m;
                    m.offset = offset;
                    m.s = s_;
                    if (best.length != 0 && best.s - best.offset == s_ - offset) return m;  // same offset: not retested
                    if (load32(src, offset) != first) return m;
                    m.length = 4 + offset;
                    m.rep = rep;
                    int sp = s_ + 4;
                    while (sp < len) {
                        if (len - sp < 8) {
                            if (src[sp] == src[m.length]) { m.length++; sp++; continue; }
                            break;
                        }
                        const uint64_t diff = load64(src, sp) ^ load64(src, m.length);
                        if (diff != 0) { m.length += tz64(diff) >> 3; break; }
                        sp += 8;
                        m.length += 8;
                    }
                    m.length -= offset;
                    m.score = score(m);
                    if (m.score <= -m.s) m.length = 0;  // no savings: maybe a better one turns up
                    return m;
                };
                auto bestOf = [](const match& a, const match& b) {
                    if (b.length == 0) return a;
                    if (a.length == 0) return b;
                    const int as = a.score + b.s, bs = b.score + a.s;
                    return as >= bs ? a : b;
                };
                if (s > 0) {
                    best = bestOf(matchAt(getCur(candidateL), s, (uint32_t)cv, false), matchAt(getPrev(candidateL), s, (uint32_t)cv, false));
                    best = bestOf(best, matchAt(getCur(candidateS), s, (uint32_t)cv, false));
                    best = bestOf(best, matchAt(getPrev(candidateS), s, (uint32_t)cv, false));
                }
                if (repeat > 0) best = bestOf(best, matchAt(s - repeat + 1, s + 1, (uint32_t)(cv >> 8), true));
                if (best.length > 0) {
                    uint32_t hS = hash4(cv >> 8, sTableBits);
                    uint64_t nextShort = sTable[hS];       // s+1
                    int s1 = s + 1;
                    uint64_t cv1 = load64(src, s1);
                    uint32_t hL = hash8(cv1, lTableBits);
                    uint64_t nextLong = lTable[hL];
                    best = bestOf(best, matchAt(getCur(nextShort), s1, (uint32_t)cv1, false));
                    best = bestOf(best, matchAt(getPrev(nextShort), s1, (uint32_t)cv1, false));
                    best = bestOf(best, matchAt(getCur(nextLong), s1, (uint32_t)cv1, false));
                    best = bestOf(best, matchAt(getPrev(nextLong), s1, (uint32_t)cv1, false));
                    {   // s+2
                        hS = hash4(cv1 >> 8, sTableBits);
                        nextShort = sTable[hS];
                        s1++;
                        cv1 = load64(src, s1);
                        hL = hash8(cv1, lTableBits);
                        nextLong = lTable[hL];
                        if (repeat > 0) best = bestOf(best, matchAt(s1 - repeat, s1, (uint32_t)cv1, true));  // repeat at +2
                        best = bestOf(best, matchAt(getCur(nextShort), s1, (uint32_t)cv1, false));
                        best = bestOf(best, matchAt(getPrev(nextShort), s1, (uint32_t)cv1, false));
                        best = bestOf(best, matchAt(getCur(nextLong), s1, (uint32_t)cv1, false));
                        best = bestOf(best, matchAt(getPrev(nextLong), s1, (uint32_t)cv1, false));
                    }
                    // a match at the end of the best match, shifted back over it (:313-345)
                    const int skipBeginning = 2, skipEnd = 1;
                    const int sAt = best.s + best.length - skipEnd;
                    if (sAt < sLimit) {
                        const int sBack = best.s + skipBeginning - skipEnd;
                        const int backL = best.length - skipBeginning;
                        const uint64_t cvb = load64(src, sBack);
                        const uint64_t next = lTable[hash8(load64(src, sAt), lTableBits)];
                        int checkAt = getCur(next) - backL;
                        if (checkAt > 0) best = bestOf(best, matchAt(checkAt, sBack, (uint32_t)cvb, false));
                        checkAt = getPrev(next) - backL;
                        if (checkAt > 0) best = bestOf(best, matchAt(checkAt, sBack, (uint32_t)cvb, false));
                    }
                }
                // update tables
                lTable[hashL] = (uint64_t)(uint32_t)s | candidateL << 32;
                sTable[hashS] = (uint64_t)(uint32_t)s | candidateS << 32;
            }
            if (best.length > 0) break;
            cv = load64(src, nextS);
            s = nextS;
        }
        // extend backwards (not for repeats)
        s = best.s;
        if (!best.rep) {
            while (best.offset > 0 && s > nextEmit && src[best.offset - 1] == src[s - 1]) { best.offset--; best.length++; s--; }
        }
        if (d + (s - nextEmit) > dstLimit) return 0;
        {
            const int base = s;
            const int offset = s - best.offset;
            s += best.length;
            if (offset > 65535 && s - base <= 5 && !best.rep) {  // equal or worse than the encoding
                s = best.s + 1;
                if (s >= sLimit) goto emitRemainder;
                cv = load64(src, s);
                continue;
            }
            d += emitLiteral(dst + d, src + nextEmit, (size_t)(base - nextEmit));
            if (best.rep) {
                if (nextEmit > 0) d += emitRepeat(dst + d, offset, best.length);
                else d += emitCopy(dst + d, offset, best.length);  // the first match cannot be a repeat
            } else {
                d += emitCopy(dst + d, offset, best.length);
            }
            repeat = offset;
            nextEmit = s;
            if (s >= sLimit) goto emitRemainder;
            if (d > dstLimit) return 0;
            for (int i = best.s + 1; i < s; i++) {  // fill tables: every position of the match
                const uint64_t cv0 = load64(src, i);
                const uint32_t long0 = hash8(cv0, lTableBits), short0 = hash4(cv0, sTableBits);
                lTable[long0] = (uint64_t)(uint32_t)i | lTable[long0] << 32;
                sTable[short0] = (uint64_t)(uint32_t)i | sTable[short0] << 32;
            }
            cv = load64(src, s);
        }
    }
emitRemainder:
    if (nextEmit < len) {
        if (d + len - nextEmit > dstLimit) return 0;
        d += emitLiteral(dst + d, src + nextEmit, (size_t)(len - nextEmit));
    }
    return d;
}

static inline int emitCopyNoRepeatSize(int offset, int length) {  // s2/encode_best.go:757-771 (an estimate, like the two above)
    if (offset >= 65536) return 5 + 5 * (length / 64);
    if (length > 64) return 3 + 3 * (length / 60);
    if (length >= 12 || offset >= 2048) return 3;
    return 2;
}

// s2/encode_best.go:457-710 encodeBlockBestSnappy: the best parse with Snappy-compatible output.  Differences from encodeBlockBest
// that are restated literally: candidates are scored with emitCopyNoRepeatSize, extension runs in whole 8-byte steps while
// s <= sLimit (no byte tail), the repeat candidate is always tried (also at +2, from the s+1 state), the end-of-match probe uses
// no skips, every match is extended backwards, the long-offset bail ignores repeats, copies go through emitCopyNoRepeat.
static int encodeBlockBestSnappy(uint8_t* dst, const uint8_t* src, size_t srcLen) {
    const int lTableBits = 19, sTableBits = 16;
    const int len = (int)srcLen;
    const int sLimit = len - (8 + 2);
    if (len < minNonLiteralBlockSize) return 0;
    std::vector<uint64_t> lTable((size_t)1 << lTableBits, 0), sTable((size_t)1 << sTableBits, 0);
    const int dstLimit = len - 5;
    int nextEmit = 0;
    int s = 1;
    uint64_t cv = load64(src, s);
    int repeat = 1;
    int d = 0;
    auto getCur = [](uint64_t x) { return (int)(x & 0xffffffffULL); };
    auto getPrev = [](uint64_t x) { return (int)(x >> 32); };
    const int maxSkip = 64;
    struct match { int offset = 0, s = 0, length = 0, score = 0; };
    for (;;) {
        match best;
        for (;;) {
            int nextS = ((s - nextEmit) >> 8) + 1;
            if (nextS > maxSkip) nextS = s + maxSkip; else nextS += s;
            if (nextS > sLimit) goto emitRemainder;
            {
                const uint32_t hashL = hash8(cv, lTableBits);
                const uint32_t hashS = hash4(cv, sTableBits);
                const uint64_t candidateL = lTable[hashL];
                const uint64_t candidateS = sTable[hashS];
                auto score = [&](const match& m) {
                    int sc = m.length - m.s;
                    if (nextEmit == m.s) sc++;
                    return sc - emitCopyNoRepeatSize(m.s - m.offset, m.length);
                };
                auto matchAt = [&](int offset, int s_, uint32_t first) {
                    match m;
                    m.offset = offset;
                    m.s = s_;
                    if (best.length != 0 && best.s - best.offset == s_ - offset) return m;
                    if (load32(src, offset) != first) return m;
                    m.length = 4 + offset;
                    int sp = s_ + 4;
                    while (sp <= sLimit) {
                        const uint64_t diff = load64(src, sp) ^ load64(src, m.length);
                        if (diff != 0) { m.length += tz64(diff) >> 3; break; }
                        sp += 8;
                        m.length += 8;
                    }
                    m.length -= offset;
                    m.score = score(m);
                    if (m.score <= -m.s) m.length = 0;
                    return m;
                };
                auto bestOf = [](const match& a, const match& b) {
                    if (b.length == 0) return a;
                    if (a.length == 0) return b;
                    const int as = a.score + b.s, bs = b.score + a.s;
                    return as >= bs ? a : b;
                };
                best = bestOf(matchAt(getCur(candidateL), s, (uint32_t)cv), matchAt(getPrev(candidateL), s, (uint32_t)cv));
                best = bestOf(best, matchAt(getCur(candidateS), s, (uint32_t)cv));
                best = bestOf(best, matchAt(getPrev(candidateS), s, (uint32_t)cv));
                best = bestOf(best, matchAt(s - repeat + 1, s + 1, (uint32_t)(cv >> 8)));
                if (best.length > 0) {
                    uint64_t nextShort = sTable[hash4(cv >> 8, sTableBits)];  // s+1
                    int s1 = s + 1;
                    uint64_t cv1 = load64(src, s1);
                    uint64_t nextLong = lTable[hash8(cv1, lTableBits)];
                    best = bestOf(best, matchAt(getCur(nextShort), s1, (uint32_t)cv1));
                    best = bestOf(best, matchAt(getPrev(nextShort), s1, (uint32_t)cv1));
                    best = bestOf(best, matchAt(getCur(nextLong), s1, (uint32_t)cv1));
                    best = bestOf(best, matchAt(getPrev(nextLong), s1, (uint32_t)cv1));
                    best = bestOf(best, matchAt(s1 - repeat + 1, s1 + 1, (uint32_t)(cv1 >> 8)));  // repeat at +2
                    nextShort = sTable[hash4(cv1 >> 8, sTableBits)];                              // s+2
                    s1++;
                    cv1 = load64(src, s1);
                    nextLong = lTable[hash8(cv1, lTableBits)];
                    best = bestOf(best, matchAt(getCur(nextShort), s1, (uint32_t)cv1));
                    best = bestOf(best, matchAt(getPrev(nextShort), s1, (uint32_t)cv1));
                    best = bestOf(best, matchAt(getCur(nextLong), s1, (uint32_t)cv1));
                    best = bestOf(best, matchAt(getPrev(nextLong), s1, (uint32_t)cv1));
                    const int sAt = best.s + best.length;  // a match at the end of the best match
                    if (sAt < sLimit) {
                        const int sBack = best.s, backL = best.length;
                        const uint64_t cvb = load64(src, sBack);
                        const uint64_t next = lTable[hash8(load64(src, sAt), lTableBits)];
                        int checkAt = getCur(next) - backL;
                        if (checkAt > 0) best = bestOf(best, matchAt(checkAt, sBack, (uint32_t)cvb));
                        checkAt = getPrev(next) - backL;
                        if (checkAt > 0) best = bestOf(best, matchAt(checkAt, sBack, (uint32_t)cvb));
                    }
                }
                lTable[hashL] = (uint64_t)(uint32_t)s | candidateL << 32;
                sTable[hashS] = (uint64_t)(uint32_t)s | candidateS << 32;
            }
            if (best.length > 0) break;
            cv = load64(src, nextS);
            s = nextS;
        }
        s = best.s;
        while (best.offset > 0 && s > nextEmit && src[best.offset - 1] == src[s - 1]) { best.offset--; best.length++; s--; }
        if (d + (s - nextEmit) > dstLimit) return 0;
        {
            const int base = s;
            const int offset = s - best.offset;
            s += best.length;
            if (offset > 65535 && s - base <= 5) {
                s = best.s + 1;
                if (s >= sLimit) goto emitRemainder;
                cv = load64(src, s);
                continue;
            }
            d += emitLiteral(dst + d, src + nextEmit, (size_t)(base - nextEmit));
            d += emitCopyNoRepeat(dst + d, offset, best.length);
            repeat = offset;
            nextEmit = s;
            if (s >= sLimit) goto emitRemainder;
            if (d > dstLimit) return 0;
            for (int i = best.s + 1; i < s; i++) {
                const uint64_t cv0 = load64(src, i);
                const uint32_t long0 = hash8(cv0, lTableBits), short0 = hash4(cv0, sTableBits);
                lTable[long0] = (uint64_t)(uint32_t)i | lTable[long0] << 32;
                sTable[short0] = (uint64_t)(uint32_t)i | sTable[short0] << 32;
            }
            cv = load64(src, s);
        }
    }
emitRemainder:
    if (nextEmit < len) {
        if (d + len - nextEmit > dstLimit) return 0;
        d += emitLiteral(dst + d, src + nextEmit, (size_t)(len - nextEmit));
    }
    return d;
}

// s2/encode.go:292 EncodeSnappyBest; returns bytes written or -1 / -2
static inline int64_t EncodeSnappyBest(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    int k = encodeBlockBestSnappy(dst + d, src, n);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

// s2/encode.go:161 EncodeBest; returns bytes written or -1 / -2
static inline int64_t EncodeBest(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    int k = encodeBlockBest(dst + d, src, n);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

// s2/encode.go:204 EncodeSnappy (s2/encode_go.go:27 encodeBlockSnappy); returns bytes written or -1 / -2
static inline int64_t EncodeSnappy(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    int k = n <= ((size_t)64 << 10) ? encodeBlockGoT<uint16_t, 5, true>(dst + d, src, n) : encodeBlockGoT<uint32_t, 6, true>(dst + d, src, n);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

// s2/encode.go:117 EncodeBetter; returns bytes written or -1 (too large) / -2 (dst too small)
static inline int64_t EncodeBetter(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    int k = encodeBlockBetter(dst + d, src, n);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

// CRC32C (Castagnoli) + s2/s2.go:120 crc masking: ((c>>15)|(c<<17)) + 0xa282ead8
static inline uint32_t crc32c(const uint8_t* p, size_t n) {
    static uint32_t tab[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            tab[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
static inline uint32_t crc(const uint8_t* p, size_t n) {
    uint32_t c = crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// Block decoder (format per s2/decode_other.go:22-260); verifier only.
// Returns decoded length or -1 on corruption / -2 if dst too small.
static inline int64_t Decode(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    size_t s = 0;
    uint64_t dLen = 0;
    int shift = 0;
    for (;;) {
        if (s >= n || shift > 63) return -1;
        uint8_t b = src[s++];
        dLen |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) break;
        shift += 7;
    }
    if (dLen > cap) return -2;
    uint64_t d = 0;
    uint64_t offset = 0;
    while (s < n) {
        uint64_t length;
        uint8_t tag = src[s];
        switch (tag & 3) {
        case tagLiteral: {
            uint32_t x = tag >> 2;
            if (x < 60) { s += 1; }
            else if (x == 60) { if (s + 2 > n) return -1; x = src[s + 1]; s += 2; }
            else if (x == 61) { if (s + 3 > n) return -1; x = load16(src, (int64_t)s + 1); s += 3; }
            else if (x == 62) { if (s + 4 > n) return -1; x = (uint32_t)src[s + 1] | (uint32_t)src[s + 2] << 8 | (uint32_t)src[s + 3] << 16; s += 4; }
            else { if (s + 5 > n) return -1; x = load32(src, (int64_t)s + 1); s += 5; }
            length = (uint64_t)x + 1;
            if (length > dLen - d || length > n - s) return -1;
            memcpy(dst + d, src + s, length);
            d += length;
            s += length;
            continue;
        }
        case tagCopy1: {
            if (s + 2 > n) return -1;
            uint64_t toffset = ((uint64_t)(tag & 0xe0) << 3) | src[s + 1];
            length = (tag >> 2) & 7;
            s += 2;
            if (toffset == 0) {
                switch (length) {
                case 5: if (s + 1 > n) return -1; length = (uint64_t)src[s] + 4; s += 1; break;
                case 6: if (s + 2 > n) return -1; length = (uint64_t)load16(src, (int64_t)s) + (1 << 8); s += 2; break;
                case 7: if (s + 3 > n) return -1; length = ((uint64_t)src[s] | (uint64_t)src[s + 1] << 8 | (uint64_t)src[s + 2] << 16) + (1 << 16); s += 3; break;
                default: break;
                }
            } else {
                offset = toffset;
            }
            length += 4;
            break;
        }
        case tagCopy2:
            if (s + 3 > n) return -1;
            offset = load16(src, (int64_t)s + 1);
            length = 1 + (uint64_t)(tag >> 2);
            s += 3;
            break;
        default:
            if (s + 5 > n) return -1;
            offset = load32(src, (int64_t)s + 1);
            length = 1 + (uint64_t)(tag >> 2);
            s += 5;
            break;
        }
        if (offset == 0 || d < offset || length > dLen - d) return -1;
        for (uint64_t i = 0; i < length; i++) dst[d + i] = dst[d + i - offset];
        d += length;
    }
    if (d != dLen) return -1;
    return (int64_t)d;
}

// s2.Writer chunk for one block (s2/writer.go:414-451): type(1) | len24 | masked CRC32C(4) | body.
// Compressed body = uvarint(len) + encodeBlock output; encodeBlock == 0 -> uncompressed chunk (type 0x01, raw bytes).
// Returns bytes written to dst (needs len + 8 + 5).
static inline int64_t EncodeChunk(uint8_t* dst, const uint8_t* src, size_t n) {
    const uint32_t checksum = crc(src, n);
    uint8_t chunkType = 0x01;
    size_t chunkLen = 4 + n;
    const int v = putUvarint(dst + 8, (uint64_t)n);
    const int n2 = encodeBlock(dst + 8 + v, src, n);
    if (n2 > 0) { chunkType = 0x00; chunkLen = 4 + (size_t)v + (size_t)n2; }
    else memcpy(dst + 8, src, n);
    dst[0] = chunkType;
    dst[1] = (uint8_t)chunkLen; dst[2] = (uint8_t)(chunkLen >> 8); dst[3] = (uint8_t)(chunkLen >> 16);
    dst[4] = (uint8_t)checksum; dst[5] = (uint8_t)(checksum >> 8); dst[6] = (uint8_t)(checksum >> 16); dst[7] = (uint8_t)(checksum >> 24);
    return (int64_t)(4 + chunkLen);
}
// Stream decoder (verifier): stream identifier + compressed / uncompressed chunks, CRC checked. Returns decoded size or -1.
static inline int64_t DecodeStream(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n) {
    static const uint8_t magic[10] = {0xff, 0x06, 0x00, 0x00, 'S', '2', 's', 'T', 'w', 'O'};
    size_t s = 0;
    uint64_t d = 0;
    while (s + 4 <= n) {
        const uint8_t t = src[s];
        const size_t cl = (size_t)src[s + 1] | (size_t)src[s + 2] << 8 | (size_t)src[s + 3] << 16;
        s += 4;
        if (s + cl > n) return -1;
        if (t == 0xff) { if (cl != 6 || memcmp(src + s - 4, magic, 10) != 0) return -1; s += cl; continue; }
        if (t >= 0x80 && t <= 0xfe) { s += cl; continue; }  // skippable chunks / padding (s2/reader.go: chunkTypePadding 0xfe, 0x80-0xfd skippable)
        if (t != 0x00 && t != 0x01) return -1;
        if (cl < 4) return -1;
        const uint32_t want = load32(src, (int64_t)s);
        int64_t got;
        if (t == 0x01) {
            got = (int64_t)cl - 4;
            if (d + (uint64_t)got > cap) return -2;
            memcpy(dst + d, src + s + 4, (size_t)got);
        } else {
            got = Decode(dst + d, cap - d, src + s + 4, cl - 4);
            if (got < 0) return -1;
        }
        if (crc(dst + d, (size_t)got) != want) return -3;
        d += (uint64_t)got;
        s += cl;
    }
    return s == n ? (int64_t)d : -1;
}

// s2/index.go:17-236: Index.add / reduce / appendTo (the writer side of the seek index).
struct Index {
    struct Info { int64_t compressedOffset, uncompressedOffset; };
    std::vector<Info> info;
    int64_t estBlockUncomp = 0;
    static constexpr int64_t maxIndexEntries = 1 << 16, minIndexDist = 1 << 20;
    void reset(int maxBlock) { estBlockUncomp = maxBlock; info.clear(); }  // :35
    int add(int64_t compressedOffset, int64_t uncompressedOffset) {      // :57
        if (!info.empty()) {
            Info& latest = info.back();
            if (latest.uncompressedOffset == uncompressedOffset) { latest.compressedOffset = compressedOffset; return 0; }
            if (latest.uncompressedOffset > uncompressedOffset) return -1;
            if (latest.compressedOffset > compressedOffset) return -1;
            if (latest.uncompressedOffset + minIndexDist > uncompressedOffset) return 0;
        }
        info.push_back(Info{compressedOffset, uncompressedOffset});
        return 0;
    }
    void reduce() {  // :130
        if ((int64_t)info.size() < maxIndexEntries && estBlockUncomp >= minIndexDist) return;
        int64_t removeN = ((int64_t)info.size() + 1) / maxIndexEntries;
        while (estBlockUncomp * (removeN + 1) < minIndexDist && (int64_t)info.size() / (removeN + 1) > 1000) removeN++;
        size_t j = 0;
        for (size_t idx = 0; idx < info.size(); idx++) {
            info[j++] = info[idx];
            idx += (size_t)removeN;
        }
        info.resize(j);
        estBlockUncomp += estBlockUncomp * removeN;
    }
    static void putVarint(Bytes* b, int64_t x) {  // encoding/binary PutVarint: zig-zag then uvarint
        uint64_t ux = (uint64_t)x << 1;
        if (x < 0) ux = ~ux;
        while (ux >= 0x80) { b->push_back((uint8_t)ux | 0x80); ux >>= 7; }
        b->push_back((uint8_t)ux);
    }
    void appendTo(Bytes* b, int64_t uncompTotal, int64_t compTotal) {  // :154
        reduce();
        const size_t initSize = b->size();
        b->push_back(0x99); b->push_back(0); b->push_back(0); b->push_back(0);  // ChunkTypeIndex + length placeholder
        const char hdr[6] = {'s', '2', 'i', 'd', 'x', 0};
        b->insert(b->end(), hdr, hdr + 6);
        putVarint(b, uncompTotal);
        putVarint(b, compTotal);
        putVarint(b, estBlockUncomp);
        putVarint(b, (int64_t)info.size());
        uint8_t hasUncompressed = 0;
        for (size_t idx = 0; idx < info.size(); idx++) {
            if (idx == 0) { if (info[idx].uncompressedOffset != 0) { hasUncompressed = 1; break; } continue; }
            if (info[idx].uncompressedOffset != info[idx - 1].uncompressedOffset + estBlockUncomp) { hasUncompressed = 1; break; }
        }
        b->push_back(hasUncompressed);
        if (hasUncompressed)
            for (size_t idx = 0; idx < info.size(); idx++) {
                int64_t uOff = info[idx].uncompressedOffset;
                if (idx > 0) uOff -= info[idx - 1].uncompressedOffset + estBlockUncomp;
                putVarint(b, uOff);
            }
        int64_t cPredict = estBlockUncomp / 2;
        for (size_t idx = 0; idx < info.size(); idx++) {
            int64_t cOff = info[idx].compressedOffset;
            if (idx > 0) {
                cOff -= info[idx - 1].compressedOffset + cPredict;
                cPredict += cOff / 2;
            }
            putVarint(b, cOff);
        }
        const uint32_t total = (uint32_t)(b->size() - initSize + 4 + 6);
        for (int k = 0; k < 4; k++) b->push_back((uint8_t)(total >> (8 * k)));
        const char trl[6] = {0, 'x', 'd', 'i', '2', 's'};
        b->insert(b->end(), trl, trl + 6);
        const size_t chunkLen = b->size() - initSize - 4;
        (*b)[initSize + 1] = (uint8_t)chunkLen; (*b)[initSize + 2] = (uint8_t)(chunkLen >> 8); (*b)[initSize + 3] = (uint8_t)(chunkLen >> 16);
    }
};

}  // namespace s2
}  // namespace kco
