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


// genEncodeBetterBlockAsm(name, lTableBits LB, sTableBits SB, skipLog SKIP, lHashBytes LHB, maxLen) (gen.go:873-1655), o.maxSkip
// MAXSKIP (0: none), output margin OM (6; 9 for the Snappy-compatible forms), BIGOFF: maxLen - 1 > 65535
// SMALLREP: maxLen below 2048 (only encodeBetterBlockAsm8B, blocks below 512 bytes): the inlined emitRepeat has no two-byte offset form
// (gen.go:1991-1994, as in encodeBlockAsm8B) — found by tools/fuzz_emu_s2.py: the device kernel and the assembly agreed, this file did not.
template <int LB, int SB, int SKIP, int LHB, int LITOVH, int MAXSKIP, int OM, bool BIGOFF, bool SNAPPY, bool SMALLREP = false>
static int encodeBetterBlockAsmT(uint8_t* dst, const uint8_t* src, size_t srcLen) {
    std::vector<uint32_t> lTable((size_t)1 << LB, 0u), sTable((size_t)1 << SB, 0u);
    const int len = (int)srcLen;
    const int sLimit = len - 8;
    const int dstLimit = (len - OM) - (len >> 5);
    int nextEmit = 0, s = 1, repeat = 0, d = 0;
    auto HL = [](uint64_t v) -> uint32_t {
        const uint64_t prime = LHB == 5 ? 889523592379ULL : (LHB == 6 ? 227718039650203ULL : 58295818150454627ULL);
        return (uint32_t)(((v << (64 - 8 * LHB)) * prime) >> (64 - LB));
    };
    auto HS = [](uint64_t v) -> uint32_t { return (uint32_t)(((v << 32) * 2654435761ULL) >> (64 - SB)); };
    auto emitLits = [&](int until) {
        if (until == nextEmit) return;
        d += emitLiteral(dst + d, src + nextEmit, (size_t)(until - nextEmit));
        nextEmit = until;
    };
    for (;;) {  // search_loop
        int nextS;
        {
            const int t = (s - nextEmit) >> SKIP;
            if (MAXSKIP != 0 && t > MAXSKIP - 1) nextS = s + MAXSKIP;
            else nextS = s + 1 + t;
        }
        if (nextS >= sLimit) break;
        uint64_t cv = load64(src, s);
        const uint32_t hashL = HL(cv), hashS = HS(cv);
        int candidate = (int)lTable[hashL];
        const int candidateS = (int)sTable[hashS];
        lTable[hashL] = (uint32_t)s;
        sTable[hashS] = (uint32_t)s;
        const uint64_t longVal = load64(src, candidate), shortVal = load64(src, candidateS);
        if (longVal == cv) {
        } else if (shortVal == cv) {
            candidate = candidateS;
        } else if ((uint32_t)longVal == (uint32_t)cv) {
        } else if ((uint32_t)shortVal == (uint32_t)cv) {
            // short match at s: try a long candidate at s+1 (:1268-1288)
            cv >>= 8;
            const uint32_t h = HL(cv);
            candidate = (int)lTable[h];
            s++;
            lTable[h] = (uint32_t)s;
            if (load32(src, candidate) != (uint32_t)cv) {
                s--;
                candidate = candidateS;
            }
        } else {
            s = nextS;
            continue;
        }
        // candidate_match: extend backwards
        if (candidate != 0) {
            for (;;) {
                if (s <= nextEmit) break;
                if (src[candidate - 1] != src[s - 1]) break;
                s--;
                candidate--;
                if (candidate == 0) break;
            }
        }
        if (d + (s - nextEmit) + LITOVH >= dstLimit) return 0;
        const int base = s;
        s += 4;
        candidate += 4;
        int length = asmMatchLen(src + s, src + candidate, len - s);
        const int offset = s - candidate;
        if (!SNAPPY && repeat == offset) {
            emitLits(base);
            s += length;
            length += 4;
            nextEmit = s;
            d += SMALLREP ? emitRepeatAsmSmall(dst + d, offset, length) : emitRepeat(dst + d, offset, length);
        } else {
            if (BIGOFF && length <= 1 && offset > 65535) {  // equal or worse than the encoding (:1369-1379)
                s = nextS + 1;
                continue;
            }
            repeat = offset;
            emitLits(base);
            s += length;
            length += 4;
            nextEmit = s;
            d += SNAPPY ? emitCopyNoRepeat(dst + d, offset, length) : emitCopy(dst + d, offset, length);
        }
        if (s >= sLimit) break;
        if (d >= dstLimit) return 0;
        {   // index base+1 / s-2 into both tables, then the long table sparsely from both ends (:1429-1484)
            int64_t index0 = base + 1, index1 = s - 2;
            const uint32_t h0l = HL(load64(src, index0)), h0s = HS(load64(src, index0 + 1));
            const uint32_t h1l = HL(load64(src, index1)), h1s = HS(load64(src, index1 + 1));
            lTable[h0l] = (uint32_t)index0;
            lTable[h1l] = (uint32_t)index1;
            sTable[h0s] = (uint32_t)(index0 + 1);
            sTable[h1s] = (uint32_t)(index1 + 1);
            int64_t index2 = (index0 + index1 + 1) >> 1;
            index0 += 1;
            index1 -= 1;
            while (index2 < index1) {
                const uint32_t a = HL(load64(src, index0)), b = HL(load64(src, index2));
                lTable[a] = (uint32_t)index0;
                lTable[b] = (uint32_t)index2;
                index0 += 2;
                index2 += 2;
            }
        }
    }
    if (d + (len - nextEmit) + LITOVH >= dstLimit) return 0;
    emitLits(len);
    return d;
}

// s2/encode_amd64.go:99-166 encodeBlockBetter and :249-316 encodeBlockBetterSnappy
static inline int encodeBlockBetterAsm(uint8_t* dst, const uint8_t* src, size_t n, bool snappy) {
    const size_t limit12B = 16 << 10, limit10B = 4 << 10, limit8B = 512;
    if (!snappy) {
        if (n > ((size_t)4 << 20)) return encodeBetterBlockAsmT<17, 14, 7, 7, 5, 100, 6, true, false>(dst, src, n);   // encodeBetterBlockAsm
        if (n >= limit12B) return encodeBetterBlockAsmT<17, 14, 7, 7, 4, 100, 6, true, false>(dst, src, n);           // ...4MB
        if (n >= limit10B) return encodeBetterBlockAsmT<14, 12, 6, 6, 3, 0, 6, false, false>(dst, src, n);            // ...12B
        if (n >= limit8B) return encodeBetterBlockAsmT<12, 10, 5, 6, 3, 0, 6, false, false>(dst, src, n);             // ...10B
        if (n < (size_t)minNonLiteralBlockSize) return 0;
        return encodeBetterBlockAsmT<10, 8, 4, 6, 3, 0, 6, false, false, true>(dst, src, n);                          // ...8B
    }
    if (n > 65536) return encodeBetterBlockAsmT<17, 14, 7, 7, 5, 100, 9, true, true>(dst, src, n);                    // encodeSnappyBetterBlockAsm
    if (n >= limit12B) return encodeBetterBlockAsmT<16, 13, 7, 7, 3, 0, 9, false, true>(dst, src, n);                 // ...64K
    if (n >= limit10B) return encodeBetterBlockAsmT<14, 12, 6, 6, 3, 0, 9, false, true>(dst, src, n);
    if (n >= limit8B) return encodeBetterBlockAsmT<12, 10, 5, 6, 3, 0, 9, false, true>(dst, src, n);
    if (n < (size_t)minNonLiteralBlockSize) return 0;
    return encodeBetterBlockAsmT<10, 8, 4, 6, 3, 0, 9, false, true>(dst, src, n);
}

// s2.Encode / s2.EncodeSnappy (better = false), s2.EncodeBetter / s2.EncodeSnappyBetter (true) of an amd64 build
// (s2/encode.go:29-57, 117-144, 204-276)
static inline int64_t EncodeAsm(uint8_t* dst, uint64_t cap, const uint8_t* src, size_t n, bool snappy, bool better = false) {
    const int64_t m = MaxEncodedLen((int64_t)n);
    if (m < 0) return -1;
    if (cap < (uint64_t)m) return -2;
    int d = putUvarint(dst, (uint64_t)n);
    if (n == 0) return d;
    if (n < (size_t)minNonLiteralBlockSize) { d += emitLiteral(dst + d, src, n); return d; }
    const int k = better ? encodeBlockBetterAsm(dst + d, src, n, snappy) : encodeBlockAsm(dst + d, src, n, snappy);
    if (k > 0) return d + k;
    d += emitLiteral(dst + d, src, n);
    return d;
}

}  // namespace s2
}  // namespace kco
