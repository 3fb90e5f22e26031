// oracle/kco_s2_asm.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates the amd64 ASSEMBLY block encoders of s2 — the code every amd64 user of s2.Encode / s2.EncodeSnappy runs — from their
// generator, s2/_generate/gen.go:170-855 (genEncodeBlockAsm) with its matchLen (:2778-2880), and the size dispatch of
// s2/encode_amd64.go:23-87, 176-239.  The assembly is the same algorithm as encodeBlockGo (s2/encode_all.go) but not the same
// bytes: per size class another table size, hash length and skip rate; matches extended to the end of the block (matchLen takes the
// bytes left, the Go version stops 8 short of the end); `nextS >= sLimit` where the Go version has `>`; no `s >= sLimit` exit
// after a repeat; output margin 9 and the literal header's worst case counted into every bail-out test.
// Unlike the rest of the oracle this restatement is PINNED: tests/test_ref_s2asm.py compares it byte for byte with the assembly
// itself (oracle/_ref, the reference's .s file assembled here).
#pragma once
#include "kco_s2.h"

namespace kco {
namespace s2 {

// hashN (gen.go:1719-1759): the low HB bytes moved to the top, multiplied, the top TB bits kept
template <int HB, int TB>
static inline uint32_t asmHash(uint64_t v) {
    const uint64_t prime = HB == 4 ? 2654435761ULL : (HB == 5 ? 889523592379ULL : 227718039650203ULL);
    return (uint32_t)(((v << (64 - 8 * HB)) * prime) >> (64 - TB));
}

// matchLen (gen.go:2778-2880): the common prefix of a and b, at most n bytes
static inline int asmMatchLen(const uint8_t* a, const uint8_t* b, int n) {
    int m = 0;
    while (n >= 8) {
        const uint64_t diff = load64(a, m) ^ load64(b, m);
        if (diff != 0) return m + (tz64(diff) >> 3);
        m += 8;
        n -= 8;
    }
    while (n > 0 && a[m] == b[m]) { m++; n--; }
    return m;
}

// emitRepeat as generated INTO a block encoder whose maxLen is below 2048 (gen.go:1991-1994: `if o.maxLen >= 2048 { CMPL offset,
// 2048; JB repeat_two_offset }` — with the test left out nothing jumps to the two-byte offset form at all): a repeat of 9..11
// bytes takes the three-byte form there.  Only encodeBlockAsm8B (blocks below 512 bytes) is generated that way.
static inline int emitRepeatAsmSmall(uint8_t* dst, int offset, int length) {
    if (length > 8 && length < 12) {
        dst[0] = (uint8_t)(5 << 2 | 1);
        dst[1] = 0;
        dst[2] = (uint8_t)(length - 8);
        return 3;
    }
    return emitRepeat(dst, offset, length);
}

// genEncodeBlockAsm(name, tableBits TB, skipLog SKIP, hashBytes HB, maxLen): LITOVH = maxLitOverheadFor(maxLen)
template <int TB, int SKIP, int HB, int LITOVH, bool SNAPPY, bool SMALL = false>
static int encodeBlockAsmT(uint8_t* dst, const uint8_t* src, size_t srcLen) {
    std::vector<uint32_t> table((size_t)1 << TB, 0u);
    const int outputMargin = 9;  // gen.go:55, :74
    const int len = (int)srcLen;
    const int sLimit = len - 8;
    const int dstLimit = (len - outputMargin) - (len >> 5);
    int nextEmit = 0, s = 1, repeat = 1, d = 0;
    auto emitLits = [&](int until) {  // emitLiteralsDstP (:1693): nothing for an empty run
        if (until == nextEmit) return;
        d += emitLiteral(dst + d, src + nextEmit, (size_t)(until - nextEmit));
        nextEmit = until;
    };
    for (;;) {  // search_loop
        int candidate;
        {
            const int nextS = s + ((s - nextEmit) >> SKIP) + 4;
            if (nextS >= sLimit) goto emit_remainder;
            uint64_t cv = load64(src, s);
            const uint32_t hash0 = asmHash<HB, TB>(cv), hash1 = asmHash<HB, TB>(cv >> 8);
            candidate = (int)table[hash0];
            const int candidate2 = (int)table[hash1];
            table[hash0] = (uint32_t)s;
            table[hash1] = (uint32_t)(s + 1);
            const uint32_t hash2 = asmHash<HB, TB>(cv >> 16);
            if ((uint32_t)(cv >> 8) == load32(src, s - repeat + 1)) {  // repeat at s+1 (:353-520)
                int base = s + 1;
                const int ne = nextEmit;
                int i = base - repeat;
                if (i != 0) {
                    for (;;) {
                        if (base <= ne) break;
                        if (src[i - 1] != src[base - 1]) break;
                        base--;
                        i--;
                        if (i == 0) break;
                    }
                }
                if (d + (base - nextEmit) + LITOVH >= dstLimit) return 0;
                emitLits(base);
                s += 5;
                s += asmMatchLen(src + s, src + (s - repeat), len - s);
                if (SNAPPY) d += emitCopyNoRepeat(dst + d, repeat, s - base);
                else if (ne == 0) d += emitCopy(dst + d, repeat, s - base);  // the first match of a block cannot be a repeat
                else d += SMALL ? emitRepeatAsmSmall(dst + d, repeat, s - base) : emitRepeat(dst + d, repeat, s - base);
                nextEmit = s;
                continue;  // (s >= sLimit is picked up by the nextS test)
            }
            if (load32(src, candidate) == (uint32_t)cv) goto candidate_match;
            cv >>= 8;
            candidate = (int)table[hash2];
            if (load32(src, candidate2) == (uint32_t)cv) {
                table[hash2] = (uint32_t)(s + 2);
                s++;
                candidate = candidate2;
                goto candidate_match;
            }
            table[hash2] = (uint32_t)(s + 2);
            cv >>= 8;
            if (load32(src, candidate) == (uint32_t)cv) {
                s += 2;
                goto candidate_match;
            }
            s = nextS;
            continue;
        }
    candidate_match:
        if (candidate != 0) {  // extend backwards (:583-603)
            for (;;) {
                if (s <= nextEmit) break;
                if (src[candidate - 1] != src[s - 1]) break;
                s--;
                candidate--;
                if (candidate == 0) break;
            }
        }
        if (d + (s - nextEmit) + LITOVH >= dstLimit) return 0;
        emitLits(s);
        for (;;) {  // match_nolit_loop
            repeat = s - candidate;
            s += 4;
            candidate += 4;
            int length = asmMatchLen(src + s, src + candidate, len - s);
            s += length;
            length += 4;
            nextEmit = s;
            d += SNAPPY ? emitCopyNoRepeat(dst + d, repeat, length) : emitCopy(dst + d, repeat, length);
            if (s >= sLimit) goto emit_remainder;
            const uint64_t x = load64(src, s - 2);
            if (d >= dstLimit) return 0;
            const uint32_t hash0 = asmHash<HB, TB>(x), hash1 = asmHash<HB, TB>(x >> 16);
            candidate = (int)table[hash1];
            table[hash0] = (uint32_t)(s - 2);
            table[hash1] = (uint32_t)s;
            if (load32(src, candidate) == (uint32_t)(x >> 16)) continue;
            s++;
            break;
        }
    }
emit_remainder:
    if (d + (len - nextEmit) + LITOVH >= dstLimit) return 0;
    emitLits(len);
    return d;
}

// s2/encode_amd64.go:23-87 encodeBlock and :176-239 encodeBlockSnappy: the variant by input size
static inline int encodeBlockAsm(uint8_t* dst, const uint8_t* src, size_t n, bool snappy) {
    const size_t limit12B = 16 << 10, limit10B = 4 << 10, limit8B = 512;
    if (!snappy) {
        if (n >= ((size_t)4 << 20)) return encodeBlockAsmT<14, 6, 6, 5, false>(dst, src, n);  // encodeBlockAsm (maxLen MaxUint32)
        if (n >= limit12B) return encodeBlockAsmT<14, 6, 6, 4, false>(dst, src, n);            // encodeBlockAsm4MB
        if (n >= limit10B) return encodeBlockAsmT<12, 5, 5, 3, false>(dst, src, n);            // encodeBlockAsm12B
        if (n >= limit8B) return encodeBlockAsmT<10, 5, 4, 3, false>(dst, src, n);             // encodeBlockAsm10B
        if (n < (size_t)minNonLiteralBlockSize) return 0;
        return encodeBlockAsmT<8, 4, 4, 3, false, true>(dst, src, n);                          // encodeBlockAsm8B
    }
    if (n > 65536) return encodeBlockAsmT<14, 6, 6, 5, true>(dst, src, n);                    // encodeSnappyBlockAsm
    if (n >= limit12B) return encodeBlockAsmT<14, 6, 6, 3, true>(dst, src, n);                 // encodeSnappyBlockAsm64K (maxLen 65535)
    if (n >= limit10B) return encodeBlockAsmT<12, 5, 5, 3, true>(dst, src, n);
    if (n >= limit8B) return encodeBlockAsmT<10, 5, 4, 3, true>(dst, src, n);
    if (n < (size_t)minNonLiteralBlockSize) return 0;
    return encodeBlockAsmT<8, 4, 4, 3, true>(dst, src, n);
}

// s2.Encode / s2.EncodeSnappy of an amd64 build (s2/encode.go:29-57, 204-246)
static inline int64_t EncodeAsm(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n, bool snappy) {
    const int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    const int k = encodeBlockAsm(dst + d, src, n, snappy);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

}  // namespace s2
}  // namespace kco
