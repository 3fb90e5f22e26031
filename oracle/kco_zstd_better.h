// oracle/kco_zstd_better.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates zstd/enc_better.go: betterFastEncoder.Encode (:56-568), EncodeNoHist (:573-576),
// betterFastEncoderDict.Encode (:579-1091; differs by: no RLE pre-check, end-of-match re-search
// without skipBeginning, shard-dirty marks) and Reset (:1092-1243).
#pragma once
#include "kco_zstd_fast.h"

namespace kco {

constexpr int betterLongTableBits = 19, betterLongTableSize = 1 << betterLongTableBits, betterLongLen = 8;
constexpr int betterShortTableBits = 13, betterShortTableSize = 1 << betterShortTableBits, betterShortLen = 5;

struct PrevEntry { int32_t offset, prev; };

struct BetterFastEncoder : FastBase {  // enc_better.go:40
    std::vector<TableEntry> table;      // short
    std::vector<PrevEntry> longTable;
    BetterFastEncoder() : table(betterShortTableSize, TableEntry{0, 0}), longTable(betterLongTableSize, PrevEntry{0, 0}) {}

    template <bool DICT>
    void encodeT(BlockEnc* blk, const uint8_t* srcIn, size_t srcLen);
    void Encode(BlockEnc* blk, const uint8_t* src, size_t n) override { encodeT<false>(blk, src, n); }
    void EncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t n) override {  // :573
        ensureHist((int)n);
        Encode(blk, src, n);
    }
    void Reset(const DictO* d, bool singleBlock) override { resetBase(d, singleBlock); }  // :1092
    // enc_better.go:1099 ResetPrefix: every 2nd position into the long table (with its chain) and, one byte on, the short table
    void ResetPrefix(const uint8_t* prefix, size_t n) override {
        resetBasePrefix(prefix, n);
        if (n < 8) return;
        const int32_t end = cur + (int32_t)n - 8;
        for (int32_t i = cur; i < end; i += 2) {
            const uint64_t cv = load64(prefix, i - cur);
            const uint32_t h = hashLen(cv, betterLongTableBits, betterLongLen);
            longTable[h] = PrevEntry{i, longTable[h].offset};
            table[hashLen(cv >> 8, betterShortTableBits, betterShortLen)] = TableEntry{(uint32_t)(cv >> 8), i + 1};
        }
    }
};

template <bool DICT>
inline void BetterFastEncoder::encodeT(BlockEnc* blk, const uint8_t* srcIn, size_t srcLen) {
    const int inputMargin = 8 + 2;
    const int minNonLiteralBlockSize = 16;
    while (cur >= bufferReset - (int32_t)hist.size()) {
        if (hist.empty()) {
            for (auto& t : table) t = TableEntry{0, 0};
            for (auto& t : longTable) t = PrevEntry{0, 0};
            cur = maxMatchOff;
            break;
        }
        int32_t minOff = cur + (int32_t)hist.size() - maxMatchOff;
        for (auto& t : table) { int32_t v = t.offset; v = v < minOff ? 0 : v - cur + maxMatchOff; t.offset = v; }
        for (auto& t : longTable) {
            int32_t v = t.offset, v2 = t.prev;
            if (v < minOff) { v = 0; v2 = 0; }
            else { v = v - cur + maxMatchOff; v2 = v2 < minOff ? 0 : v2 - cur + maxMatchOff; }
            t = PrevEntry{v, v2};
        }
        cur = maxMatchOff;
        break;
    }
    int32_t s = addBlock(srcIn, srcLen);
    blk->size = (int)srcLen;
    if (!DICT) {  // Check RLE first (:109-117)
        if ((int)srcLen > zstdMinMatch) {
            int ml = matchLen(srcIn + 1, srcLen - 1, srcIn);
            if (ml == (int)srcLen - 1) {
                blk->literals.push_back(srcIn[0]);
                Seq sq = {1, (uint32_t)(srcLen - 1) - zstdMinMatch, 1 + 3, 0, 0, 0};
                blk->sequences.push_back(sq);
                return;
            }
        }
    }
    if ((int)srcLen < minNonLiteralBlockSize) {
        blk->extraLits = (int)srcLen;
        blk->literals.assign(srcIn, srcIn + srcLen);
        return;
    }
    const uint8_t* src = hist.data();
    const size_t len = hist.size();
    int32_t sLimit = (int32_t)len - inputMargin;
    const int stepSize = 1;
    const int kSearchStrength = 9;
    int32_t nextEmit = s;
    uint64_t cv = load64(src, s);
    int32_t offset1 = (int32_t)blk->recentOffsets[0];
    int32_t offset2 = (int32_t)blk->recentOffsets[1];
    auto addLiterals = [&](Seq* sq, int32_t until) {
        if (until == nextEmit) return;
        blk->literals.insert(blk->literals.end(), src + nextEmit, src + until);
        sq->litLen = (uint32_t)(until - nextEmit);
    };
    auto HL = [](uint64_t v) { return hashLen(v, betterLongTableBits, betterLongLen); };
    auto HS = [](uint64_t v) { return hashLen(v, betterShortTableBits, betterShortLen); };
    bool finished = false;
    while (!finished) {  // encodeLoop
        int32_t t = 0;
        bool canRepeat = blk->sequences.size() > 2;
        int32_t matched = 0, index0 = 0;
        for (;;) {
            uint32_t nextHashL = HL(cv);
            uint32_t nextHashS = HS(cv);
            PrevEntry candidateL = longTable[nextHashL];
            TableEntry candidateS = table[nextHashS];
            const int repOff = 1;
            int32_t repIndex = s - offset1 + repOff;
            int32_t off = s + cur;
            longTable[nextHashL] = PrevEntry{off, candidateL.offset};
            table[nextHashS] = TableEntry{(uint32_t)cv, off};
            index0 = s + 1;
            if (canRepeat) {
                if (repIndex >= 0 && load32(src, repIndex) == (uint32_t)(cv >> (repOff * 8))) {
                    Seq seq = {0, 0, 0, 0, 0, 0};
                    int32_t length = 4 + matchlen(s + 4 + repOff, repIndex + 4, src, len);
                    seq.matchLen = (uint32_t)(length - zstdMinMatch);
                    int32_t start = s + repOff;
                    int32_t startLimit = nextEmit + 1;
                    int32_t tMin = std::max(s - maxMatchOff, (int32_t)0);
                    while (repIndex > tMin && start > startLimit && src[repIndex - 1] == src[start - 1] &&
                           seq.matchLen < (uint32_t)(maxMatchLength - zstdMinMatch - 1)) {
                        repIndex--;
                        start--;
                        seq.matchLen++;
                    }
                    addLiterals(&seq, start);
                    seq.offset = 1;
                    blk->sequences.push_back(seq);
                    // Index match start+1 (long) -> s - 1.  (non-dict: a shadowing local index0; dict: the outer one,
                    // which already equals s+1 == s+repOff, :150 vs :154)
                    int32_t idx = s + repOff;
                    s += length + repOff;
                    nextEmit = s;
                    if (s >= sLimit) { finished = true; break; }
                    while (idx < s - 1) {
                        uint64_t cv0 = load64(src, idx);
                        uint64_t cv1 = cv0 >> 8;
                        uint32_t h0 = HL(cv0);
                        int32_t o = idx + cur;
                        longTable[h0] = PrevEntry{o, longTable[h0].offset};
                        table[HS(cv1)] = TableEntry{(uint32_t)cv1, o + 1};
                        idx += 2;
                    }
                    if (DICT) index0 = idx;  // dict variant advances the outer index0 (no shadowing)
                    cv = load64(src, s);
                    continue;
                }
            }
            int32_t coffsetL = candidateL.offset - cur;
            int32_t coffsetLP = candidateL.prev - cur;
            if (s - coffsetL < maxMatchOff && cv == load64(src, coffsetL)) {
                matched = matchlen(s + 8, coffsetL + 8, src, len) + 8;
                t = coffsetL;
                if (s - coffsetLP < maxMatchOff && cv == load64(src, coffsetLP)) {
                    int32_t prevMatch = matchlen(s + 8, coffsetLP + 8, src, len) + 8;
                    if (prevMatch > matched) { matched = prevMatch; t = coffsetLP; }
                }
                break;
            }
            if (s - coffsetLP < maxMatchOff && cv == load64(src, coffsetLP)) {
                matched = matchlen(s + 8, coffsetLP + 8, src, len) + 8;
                t = coffsetLP;
                break;
            }
            int32_t coffsetS = candidateS.offset - cur;
            if (s - coffsetS < maxMatchOff && (uint32_t)cv == candidateS.val) {
                matched = matchlen(s + 4, coffsetS + 4, src, len) + 4;
                const int checkAt = 1;
                uint64_t cv2 = load64(src, s + checkAt);
                nextHashL = HL(cv2);
                candidateL = longTable[nextHashL];
                coffsetL = candidateL.offset - cur;
                longTable[nextHashL] = PrevEntry{s + checkAt + cur, candidateL.offset};
                if (s - coffsetL < maxMatchOff && cv2 == load64(src, coffsetL)) {
                    int32_t matchedNext = matchlen(s + 8 + checkAt, coffsetL + 8, src, len) + 8;
                    if (matchedNext > matched) { t = coffsetL; s += checkAt; matched = matchedNext; break; }
                }
                coffsetL = candidateL.prev - cur;
                if (s - coffsetL < maxMatchOff && cv2 == load64(src, coffsetL)) {
                    int32_t matchedNext = matchlen(s + 8 + checkAt, coffsetL + 8, src, len) + 8;
                    if (matchedNext > matched) { t = coffsetL; s += checkAt; matched = matchedNext; break; }
                }
                t = coffsetS;
                break;
            }
            s += stepSize + ((s - nextEmit) >> (kSearchStrength - 1));
            if (s >= sLimit) { finished = true; break; }
            cv = load64(src, s);
        }
        if (finished) break;
        // Try to find a better match by searching for a long match at the end of the current best match (:419-460)
        if (s + matched < sLimit) {
            const int skipBeginning = DICT ? 0 : 3;
            uint32_t nextHashL = HL(load64(src, s + matched));
            int32_t s2 = s + skipBeginning;
            uint32_t cv4 = load32(src, s2);
            PrevEntry candidateL = longTable[nextHashL];
            int32_t coffsetL = candidateL.offset - cur - matched + skipBeginning;
            if (coffsetL >= 0 && coffsetL < s2 && s2 - coffsetL < maxMatchOff && cv4 == load32(src, coffsetL)) {
                int32_t matchedNext = matchlen(s2 + 4, coffsetL + 4, src, len) + 4;
                if (matchedNext > matched) { t = coffsetL; s = s2; matched = matchedNext; }
            }
            coffsetL = candidateL.prev - cur - matched + skipBeginning;
            // NOTE: the reference computes s2 once, before the first branch may have moved s (:426 / :371 use s at entry).
            if (coffsetL >= 0 && coffsetL < s2 && s2 - coffsetL < maxMatchOff && cv4 == load32(src, coffsetL)) {
                int32_t matchedNext = matchlen(s2 + 4, coffsetL + 4, src, len) + 4;
                if (matchedNext > matched) { t = coffsetL; s = s2; matched = matchedNext; }
            }
        }
        offset2 = offset1;
        offset1 = s - t;
        int32_t l = matched;
        int32_t tMin = std::max(s - maxMatchOff, (int32_t)0);
        while (t > tMin && s > nextEmit && src[t - 1] == src[s - 1] && l < maxMatchLength) { s--; t--; l++; }
        Seq seq = {0, 0, 0, 0, 0, 0};
        seq.litLen = (uint32_t)(s - nextEmit);
        seq.matchLen = (uint32_t)(l - zstdMinMatch);
        if (seq.litLen > 0) blk->literals.insert(blk->literals.end(), src + nextEmit, src + s);
        seq.offset = (uint32_t)(s - t) + 3;
        s += l;
        blk->sequences.push_back(seq);
        nextEmit = s;
        if (s >= sLimit) break;
        {
            int32_t off = index0 + cur;
            while (index0 < s - 1) {
                uint64_t cv0 = load64(src, index0);
                uint64_t cv1 = cv0 >> 8;
                uint32_t h0 = HL(cv0);
                longTable[h0] = PrevEntry{off, longTable[h0].offset};
                table[HS(cv1)] = TableEntry{(uint32_t)cv1, off + 1};
                index0 += 2;
                off += 2;
            }
        }
        cv = load64(src, s);
        if (!canRepeat) continue;
        for (;;) {
            int32_t o2 = s - offset2;
            if (load32(src, o2) != (uint32_t)cv) break;
            uint32_t nextHashL = HL(cv);
            uint32_t nextHashS = HS(cv);
            int32_t l2 = 4 + matchlen(s + 4, o2 + 4, src, len);
            longTable[nextHashL] = PrevEntry{s + cur, longTable[nextHashL].offset};
            table[nextHashS] = TableEntry{(uint32_t)cv, s + cur};
            seq.matchLen = (uint32_t)l2 - zstdMinMatch;
            seq.litLen = 0;
            seq.offset = 1;
            s += l2;
            nextEmit = s;
            blk->sequences.push_back(seq);
            std::swap(offset1, offset2);
            if (s >= sLimit) { finished = true; break; }
            cv = load64(src, s);
        }
    }
    if ((size_t)nextEmit < len) {
        blk->literals.insert(blk->literals.end(), src + nextEmit, src + len);
        blk->extraLits = (int)len - (int)nextEmit;
    }
    blk->recentOffsets[0] = (uint32_t)offset1;
    blk->recentOffsets[1] = (uint32_t)offset2;
}

struct BetterFastEncoderDict : BetterFastEncoder {  // enc_better.go:46
    std::vector<TableEntry> dictTable;
    std::vector<PrevEntry> dictLongTable;
    void Encode(BlockEnc* blk, const uint8_t* src, size_t n) override { encodeT<true>(blk, src, n); }
    void EncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t n) override { ensureHist((int)n); Encode(blk, src, n); }
    // enc_better.go:1114 Reset.  The shard-dirty bookkeeping only avoids copying clean shards; the table state
    // after Reset always equals the pristine dictionary tables (App. A-8), so both tables are copied whole.
    void Reset(const DictO* d, bool singleBlock) override {
        resetBase(d, singleBlock);
        if (d == nullptr) return;
        bool dictChanged = d != lastDict;
        if (dictTable.size() != table.size() || dictChanged) {
            dictTable.assign(table.size(), TableEntry{0, 0});
            int32_t end = (int32_t)d->content.size() - 8 + maxMatchOff;
            for (int32_t i = maxMatchOff; i < end; i += 4) {
                uint64_t cv = load64(d->content.data(), i - maxMatchOff);
                dictTable[hashLen(cv, betterShortTableBits, betterShortLen)] = TableEntry{(uint32_t)cv, i};
                dictTable[hashLen(cv >> 8, betterShortTableBits, betterShortLen)] = TableEntry{(uint32_t)(cv >> 8), i + 1};
                dictTable[hashLen(cv >> 16, betterShortTableBits, betterShortLen)] = TableEntry{(uint32_t)(cv >> 16), i + 2};
                dictTable[hashLen(cv >> 24, betterShortTableBits, betterShortLen)] = TableEntry{(uint32_t)(cv >> 24), i + 3};
            }
        }
        if (dictLongTable.size() != longTable.size() || dictChanged) {
            dictLongTable.assign(longTable.size(), PrevEntry{0, 0});
            if (d->content.size() >= 8) {
                uint64_t cv = load64(d->content.data(), 0);
                uint32_t h = hashLen(cv, betterLongTableBits, betterLongLen);
                dictLongTable[h] = PrevEntry{maxMatchOff, dictLongTable[h].offset};
                int32_t end = (int32_t)d->content.size() - 8 + maxMatchOff;
                int off = 8;
                for (int32_t i = maxMatchOff + 1; i < end; i++) {
                    cv = cv >> 8 | ((uint64_t)d->content[(size_t)off] << 56);
                    h = hashLen(cv, betterLongTableBits, betterLongLen);
                    dictLongTable[h] = PrevEntry{i, dictLongTable[h].offset};
                    off++;
                }
            }
        }
        lastDict = d;
        table = dictTable;
        longTable = dictLongTable;
        cur = maxMatchOff;
    }
};

}  // namespace kco
