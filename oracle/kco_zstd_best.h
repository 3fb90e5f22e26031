// oracle/kco_zstd_best.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates zstd/enc_best.go: match.estBits (:39-62), bestFastEncoder.Encode (:80-463), EncodeNoHist (:468-471), Reset with a
// dictionary (:474-552), ResetPrefix (:554-568); compress.ShannonEntropyBits (compressible.go:68-85) with Go's math.Log2
// (math/log10.go: frexp, exact for powers of two, else Log(frac) * (1/Ln2) + exp) and math.Log (math/log.go, the FDLIBM e_log
// algorithm) spelled out in IEEE double operations, because the estimate is part of the encoder's decisions.
#pragma once
#include <cmath>
#include "kco_zstd_fast.h"
#include "kco_zstd_better.h"
#include "kco_zstd_fse.h"

namespace kco {

constexpr int bestLongTableBits = 22, bestLongTableSize = 1 << bestLongTableBits, bestLongLen = 8;
constexpr int bestShortTableBits = 18, bestShortTableSize = 1 << bestShortTableBits, bestShortLen = 4;
constexpr int32_t bestMaxMatchLen = 131074;           // blockdec.go:48
constexpr int32_t bestHighScore = bestMaxMatchLen * 8;  // enc_best.go:37

namespace golog {
// math.Log (math/log.go): no fused multiply-add (the Go compiler does not fuse on amd64)
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif
static inline double frexp2(double x, int* e) { return std::frexp(x, e); }  // exact
static inline double log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    const double Sqrt2 = 1.41421356237309504880168872420969808;
    int ki;
    volatile double f1 = frexp2(x, &ki);
    if (f1 < Sqrt2 / 2) { f1 = f1 * 2; ki--; }
    volatile double f = f1 - 1;
    const double k = (double)ki;
    volatile double s = f / (2 + f);
    volatile double s2 = s * s;
    volatile double s4 = s2 * s2;
    volatile double a7 = s4 * L7;
    volatile double a5 = L5 + a7;
    volatile double b5 = s4 * a5;
    volatile double a3 = L3 + b5;
    volatile double b3 = s4 * a3;
    volatile double a1 = L1 + b3;
    volatile double t1 = s2 * a1;
    volatile double c6 = s4 * L6;
    volatile double c4 = L4 + c6;
    volatile double d4 = s4 * c4;
    volatile double c2 = L2 + d4;
    volatile double t2 = s4 * c2;
    volatile double R = t1 + t2;
    volatile double hf = 0.5 * f;
    volatile double hfsq = hf * f;
    volatile double u1 = hfsq + R;
    volatile double u2 = s * u1;
    volatile double u3 = k * Ln2Lo;
    volatile double u4 = u2 + u3;
    volatile double u5 = hfsq - u4;
    volatile double u6 = u5 - f;
    volatile double u7 = k * Ln2Hi;
    return u7 - u6;
}
static inline double log2(double x) {  // math/log10.go log2
    int e;
    const double frac = frexp2(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    const double invLn2 = 1.4426950408889634;  // 1/Ln2 rounded to float64 (0x3FF71547652B82FE)
    volatile double a = log(frac) * invLn2;
    return a + (double)e;
}
#if !defined(__clang__) && defined(__GNUC__)
#pragma GCC pop_options
#endif
}  // namespace golog

// compressible.go:68 ShannonEntropyBits
static inline int ShannonEntropyBits(const uint8_t* b, size_t n) {
    if (n == 0) return 0;
    int hist[256] = {0};
    for (size_t i = 0; i < n; i++) hist[b[i]]++;
    double shannon = 0;
    const double invTotal = 1.0 / (double)n;
    for (int i = 0; i < 256; i++) {
        if (hist[i] > 0) {
            const double v = (double)hist[i];
            volatile double p = v * invTotal;
            volatile double t = -golog::log2(p) * v;
            shannon += std::ceil(t);
        }
    }
    return (int)std::ceil(shannon);
}

struct BestMatch { int32_t offset, s, length, rep, est; };  // enc_best.go:29

// match.estBits (:39)
static inline void bestEstBits(BestMatch* m, int32_t bitsPerByte) {
    const uint8_t mlc = zfse::mlCode((uint32_t)(m->length - zstdMinMatch));
    uint8_t ofc;
    if (m->rep < 0) ofc = zfse::ofCode((uint32_t)(m->s - m->offset) + 3);
    else ofc = zfse::ofCode((uint32_t)m->rep & 3);
    const zfse::SymbolTransform ofTT = zfse::predef().enc[1].symbolTT[ofc], mlTT = zfse::predef().enc[2].symbolTT[mlc];
    m->est = (int32_t)(ofTT.outBits + mlTT.outBits);
    m->est += (int32_t)((ofTT.deltaNbBits >> 16) + (mlTT.deltaNbBits >> 16));
    m->est -= (m->length * bitsPerByte) >> 10;
    if (m->est > 0) {
        m->length = 0;
        m->est = bestHighScore;
    }
}

struct BestFastEncoder : FastBase {  // enc_best.go:71
    std::vector<PrevEntry> table;      // short
    std::vector<PrevEntry> longTable;
    BestFastEncoder() : table(bestShortTableSize, PrevEntry{0, 0}), longTable(bestLongTableSize, PrevEntry{0, 0}) {}

    void Encode(BlockEnc* blk, const uint8_t* src, size_t n) override;
    void EncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t n) override {  // :468
        ensureHist((int)n);
        Encode(blk, src, n);
    }
    // :474 Reset.  The reference builds dictTable / dictLongTable once per dictionary and copies them; the result is the same
    // as building them in place.
    void Reset(const DictO* d, bool singleBlock) override {
        resetBase(d, singleBlock);
        if (d == nullptr) return;
        for (auto& t : table) t = PrevEntry{0, 0};
        for (auto& t : longTable) t = PrevEntry{0, 0};
        const uint8_t* c = d->content.data();
        const int32_t clen = (int32_t)d->content.size();
        {
            const int32_t end = clen - 8 + maxMatchOff;
            for (int32_t i = maxMatchOff; i < end; i += 4) {
                const uint64_t cv = load64(c, i - maxMatchOff);
                const uint32_t h0 = hashLen(cv, bestShortTableBits, bestShortLen), h1 = hashLen(cv >> 8, bestShortTableBits, bestShortLen);
                const uint32_t h2 = hashLen(cv >> 16, bestShortTableBits, bestShortLen), h3 = hashLen(cv >> 24, bestShortTableBits, bestShortLen);
                table[h0] = PrevEntry{i, table[h0].offset};
                table[h1] = PrevEntry{i + 1, table[h1].offset};
                table[h2] = PrevEntry{i + 2, table[h2].offset};
                table[h3] = PrevEntry{i + 3, table[h3].offset};
            }
        }
        if (clen >= 8) {
            uint64_t cv = load64(c, 0);
            uint32_t h = hashLen(cv, bestLongTableBits, bestLongLen);
            longTable[h] = PrevEntry{maxMatchOff, longTable[h].offset};
            const int32_t end = clen - 8 + maxMatchOff;
            int off = 8;
            for (int32_t i = maxMatchOff + 1; i < end; i++) {
                cv = cv >> 8 | ((uint64_t)c[off] << 56);
                h = hashLen(cv, bestLongTableBits, bestLongLen);
                longTable[h] = PrevEntry{i, longTable[h].offset};
                off++;
            }
        }
        cur = maxMatchOff;
    }
    // :554 ResetPrefix: every position of the prefix into both tables, with their chains
    void ResetPrefix(const uint8_t* prefix, size_t n) override {
        resetBasePrefix(prefix, n);
        if (n < 8) return;
        const int32_t end = cur + (int32_t)n - 8;
        for (int32_t i = cur; i < end; i++) {
            const uint64_t cv = load64(prefix, i - cur);
            const uint32_t h = hashLen(cv, bestLongTableBits, bestLongLen);
            longTable[h] = PrevEntry{i, longTable[h].offset};
            const uint32_t h0 = hashLen(cv, bestShortTableBits, bestShortLen);
            table[h0] = PrevEntry{i, table[h0].offset};
        }
    }
};

inline void BestFastEncoder::Encode(BlockEnc* blk, const uint8_t* srcIn, size_t srcLen) {
    const int inputMargin = 8 + 4;
    const int minNonLiteralBlockSize = 16;
    while (cur >= bufferReset - (int32_t)hist.size()) {  // :88-133
        if (hist.empty()) {
            for (auto& t : table) t = PrevEntry{0, 0};
            for (auto& t : longTable) t = PrevEntry{0, 0};
            cur = maxMatchOff;
            break;
        }
        const int32_t minOff = cur + (int32_t)hist.size() - maxMatchOff;
        auto shift = [&](PrevEntry& t) {
            int32_t v = t.offset, v2 = t.prev;
            if (v < minOff) { v = 0; v2 = 0; }
            else { v = v - cur + maxMatchOff; v2 = v2 < minOff ? 0 : v2 - cur + maxMatchOff; }
            t = PrevEntry{v, v2};
        };
        for (auto& t : table) shift(t);
        for (auto& t : longTable) shift(t);
        cur = maxMatchOff;
        break;
    }
    int32_t s = addBlock(srcIn, srcLen);
    blk->size = (int)srcLen;
    if ((int)srcLen > zstdMinMatch) {  // Check RLE first (:143-150)
        const int ml = matchLen(srcIn + 1, srcLen - 1, srcIn);
        if (ml == (int)srcLen - 1) {
            blk->literals.push_back(srcIn[0]);
            Seq sq = {1, (uint32_t)(srcLen - 1) - zstdMinMatch, 1 + 3, 0, 0, 0};
            blk->sequences.push_back(sq);
            return;
        }
    }
    if ((int)srcLen < minNonLiteralBlockSize) {
        blk->extraLits = (int)srcLen;
        blk->literals.assign(srcIn, srcIn + srcLen);
        return;
    }
    // literal cost estimate, scaled by 10 bits (:160-164)
    const int32_t bitsPerByte = std::max((int32_t)(((int64_t)ShannonEntropyBits(srcIn, srcLen) * 1024) / (int64_t)srcLen), (int32_t)1024);

    const uint8_t* src = hist.data();
    const size_t len = hist.size();
    const int32_t sLimit = (int32_t)len - inputMargin;
    const int kSearchStrength = 10;
    int32_t nextEmit = s;
    int32_t offset1 = (int32_t)blk->recentOffsets[0];
    int32_t offset2 = (int32_t)blk->recentOffsets[1];
    int32_t offset3 = (int32_t)blk->recentOffsets[2];
    auto addLiterals = [&](Seq* sq, int32_t until) {
        if (until == nextEmit) return;
        blk->literals.insert(blk->literals.end(), src + nextEmit, src + until);
        sq->litLen = (uint32_t)(until - nextEmit);
    };
    auto HL = [](uint64_t v) { return hashLen(v, bestLongTableBits, bestLongLen); };
    auto HS = [](uint64_t v) { return hashLen(v, bestShortTableBits, bestShortLen); };

    for (;;) {  // encodeLoop
        const bool canRepeat = blk->sequences.size() > 2;
        const int32_t goodEnough = 250;
        uint64_t cv = load64(src, s);
        const uint32_t nextHashL = HL(cv), nextHashS = HS(cv);
        PrevEntry candidateL = longTable[nextHashL];
        PrevEntry candidateS = table[nextHashS];

        // :209-254 improve
        auto improve = [&](BestMatch* m, int32_t offset, int32_t s_, uint32_t first, int32_t rep) {
            const int32_t delta = s_ - offset;
            if (delta >= maxMatchOff || delta <= 0) return;
            if (offset < 0) return;  // (cannot happen: the reference would index out of range)
            if (load32(src, offset) != first) return;
            if (m->length > 16) {  // quick reject against a long match
                const int left = (int)len - (int)(m->s + m->length);
                if (left <= 0) return;
                const int32_t checkLen = m->length - (s_ - m->s) - 8;
                if (left > 2 && checkLen > 4) {
                    if (load32(src, offset + checkLen) != load32(src, s_ + checkLen)) return;
                }
            }
            int32_t l = 4 + matchlen(s_ + 4, offset + 4, src, len);
            if (m->rep <= 0) {  // (the current best's rep, not the candidate's)
                const int32_t tMin = std::max(s_ - maxMatchOff, (int32_t)0);
                while (offset > tMin && s_ > nextEmit && src[offset - 1] == src[s_ - 1] && l < maxMatchLength) {
                    s_--;
                    offset--;
                    l++;
                }
            }
            BestMatch cand = {offset, s_, l, rep, 0};
            bestEstBits(&cand, bitsPerByte);
            if (m->est >= bestHighScore || cand.est - m->est + (((cand.s - m->s) * bitsPerByte) >> 10) < 0) *m = cand;
        };

        BestMatch best = {0, s, 0, 0, bestHighScore};
        improve(&best, candidateL.offset - cur, s, (uint32_t)cv, -1);
        improve(&best, candidateL.prev - cur, s, (uint32_t)cv, -1);
        improve(&best, candidateS.offset - cur, s, (uint32_t)cv, -1);
        improve(&best, candidateS.prev - cur, s, (uint32_t)cv, -1);

        if (canRepeat && best.length < goodEnough) {
            if (s == nextEmit) {  // repeats straight after a match
                improve(&best, s - offset2, s, (uint32_t)cv, 1 | 4);
                improve(&best, s - offset3, s, (uint32_t)cv, 2 | 4);
                if (offset1 > 1) improve(&best, s - (offset1 - 1), s, (uint32_t)cv, 3 | 4);
            }
            if (best.rep <= 0) {  // no match or a non-repeat match: check at +1
                uint32_t cv32 = (uint32_t)(cv >> 8);
                int32_t spp = s + 1;
                improve(&best, spp - offset1, spp, cv32, 1);
                improve(&best, spp - offset2, spp, cv32, 2);
                improve(&best, spp - offset3, spp, cv32, 3);
                if (best.rep < 0) {
                    cv32 = (uint32_t)(cv >> 24);
                    spp += 2;
                    improve(&best, spp - offset1, spp, cv32, 1);
                    improve(&best, spp - offset2, spp, cv32, 2);
                    improve(&best, spp - offset3, spp, cv32, 3);
                }
            }
        }
        // :287-289
        longTable[nextHashL] = PrevEntry{s + cur, candidateL.offset};
        table[nextHashS] = PrevEntry{s + cur, candidateS.offset};
        int32_t index0 = s + 1;

        if (best.length < goodEnough) {
            if (best.length < 4) {  // no match: move forward
                s += 1 + ((s - nextEmit) >> (kSearchStrength - 1));
                if (s >= sLimit) break;
                continue;
            }
            candidateS = table[HS(cv >> 8)];
            cv = load64(src, s + 1);
            const uint64_t cv2 = load64(src, s + 2);
            candidateL = longTable[HL(cv)];
            const PrevEntry candidateL2 = longTable[HL(cv2)];
            improve(&best, candidateS.offset - cur, s + 1, (uint32_t)cv, -1);   // short at s+1
            improve(&best, candidateL.offset - cur, s + 1, (uint32_t)cv, -1);   // long at s+1, s+2
            improve(&best, candidateL.prev - cur, s + 1, (uint32_t)cv, -1);
            improve(&best, candidateL2.offset - cur, s + 2, (uint32_t)cv2, -1);
            improve(&best, candidateL2.prev - cur, s + 2, (uint32_t)cv2, -1);
            const int32_t skipBeginning = 2;
            if (best.s > s - skipBeginning) {  // where the current best ends (:331-345)
                const int32_t sAt = best.s + best.length;
                if (sAt < sLimit) {
                    const PrevEntry candidateEnd = longTable[HL(load64(src, sAt))];
                    const int32_t off = candidateEnd.offset - cur - best.length + skipBeginning;
                    if (off >= 0) {
                        improve(&best, off, best.s + skipBeginning, load32(src, best.s + skipBeginning), -1);
                        const int32_t off2 = candidateEnd.prev - cur - best.length + skipBeginning;
                        if (off2 >= 0) improve(&best, off2, best.s + skipBeginning, load32(src, best.s + skipBeginning), -1);
                    }
                }
            }
        }

        // We have a match (:364)
        s = best.s;
        if (best.rep > 0) {
            Seq seq = {0, 0, 0, 0, 0, 0};
            seq.matchLen = (uint32_t)(best.length - zstdMinMatch);
            addLiterals(&seq, best.s);
            seq.offset = (uint32_t)(best.rep & 3);  // bit 4 set: a repeat straight after a match
            blk->sequences.push_back(seq);
            s = best.s + best.length;
            nextEmit = s;
            const int32_t end = std::min(s, sLimit + 4);  // index skipped (:384-395)
            int32_t off = index0 + cur;
            while (index0 < end) {
                const uint64_t cv0 = load64(src, index0);
                const uint32_t h0 = HL(cv0), h1 = HS(cv0);
                longTable[h0] = PrevEntry{off, longTable[h0].offset};
                table[h1] = PrevEntry{off, table[h1].offset};
                off++;
                index0++;
            }
            switch (best.rep) {
            case 2: case 4 | 1: std::swap(offset1, offset2); break;
            case 3: case 4 | 2: { const int32_t o1 = offset1, o2 = offset2; offset1 = offset3; offset2 = o1; offset3 = o2; break; }
            case 4 | 3: { const int32_t o1 = offset1, o2 = offset2; offset1 = o1 - 1; offset2 = o1; offset3 = o2; break; }
            }
            if (s >= sLimit) break;
            continue;
        }
        // a regular match (:414-455)
        const int32_t t = best.offset;
        offset3 = offset2; offset2 = offset1; offset1 = s - t;
        Seq seq = {0, 0, 0, 0, 0, 0};
        const int32_t l = best.length;
        seq.litLen = (uint32_t)(s - nextEmit);
        seq.matchLen = (uint32_t)(l - zstdMinMatch);
        if (seq.litLen > 0) blk->literals.insert(blk->literals.end(), src + nextEmit, src + s);
        seq.offset = (uint32_t)(s - t) + 3;
        s += l;
        blk->sequences.push_back(seq);
        nextEmit = s;
        const int32_t end = std::min(s, sLimit - 4);
        int32_t off = index0 + cur;
        while (index0 < end) {
            const uint64_t cv0 = load64(src, index0);
            const uint32_t h0 = HL(cv0), h1 = HS(cv0);
            longTable[h0] = PrevEntry{off, longTable[h0].offset};
            table[h1] = PrevEntry{off, table[h1].offset};
            index0++;
            off++;
        }
        if (s >= sLimit) break;
    }
    if ((size_t)nextEmit < len) {
        blk->literals.insert(blk->literals.end(), src + nextEmit, src + len);
        blk->extraLits = (int)len - nextEmit;
    }
    blk->recentOffsets[0] = (uint32_t)offset1;
    blk->recentOffsets[1] = (uint32_t)offset2;
    blk->recentOffsets[2] = (uint32_t)offset3;
}

}  // namespace kco
