// oracle/kco_zstd_dfast.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates zstd/enc_dfast.go: doubleFastEncoder.Encode (:38-367), EncodeNoHist (:372-675),
// doubleFastEncoderDict.Encode (:678-1031; same decisions + shard-dirty marks) and the
// Reset functions (:1033-1115).
#pragma once
#include "kco_zstd_fast.h"

namespace kco {

constexpr int dFastLongTableBits = 17, dFastLongTableSize = 1 << dFastLongTableBits, dFastLongLen = 8;
constexpr int dFastShortTableBits = tableBits, dFastShortLen = 5;

// The encode loops are written once over a "host" that owns base state + both tables,
// so that doubleFastEncoder (embeds fastEncoder) and doubleFastEncoderDict (embeds
// fastEncoderDict) share the restatement, as the reference's two copies differ only in
// dirty-shard bookkeeping (verified by diff).
template <class Base>
struct DoubleFastT : Base {
    std::vector<TableEntry> longTable;
    DoubleFastT() : longTable(dFastLongTableSize, TableEntry{0, 0}) {}

    void dfEncode(BlockEnc* blk, const uint8_t* srcIn, size_t srcLen) {
        const int inputMargin = 8 + 2;
        const int minNonLiteralBlockSize = 16;
        auto& table = this->table;
        int32_t& cur = this->cur;
        const int32_t maxMatchOff = this->maxMatchOff;
        while (cur >= this->bufferReset - (int32_t)this->hist.size()) {
            if (this->hist.empty()) {
                for (auto& t : table) t = TableEntry{0, 0};
                for (auto& t : longTable) t = TableEntry{0, 0};
                cur = maxMatchOff;
                break;
            }
            int32_t minOff = cur + (int32_t)this->hist.size() - maxMatchOff;
            for (auto& t : table) { int32_t v = t.offset; v = v < minOff ? 0 : v - cur + maxMatchOff; t.offset = v; }
            for (auto& t : longTable) { int32_t v = t.offset; v = v < minOff ? 0 : v - cur + maxMatchOff; t.offset = v; }
            cur = maxMatchOff;
            break;
        }
        int32_t s = this->addBlock(srcIn, srcLen);
        blk->size = (int)srcLen;
        if ((int)srcLen < minNonLiteralBlockSize) {
            blk->extraLits = (int)srcLen;
            blk->literals.assign(srcIn, srcIn + srcLen);
            return;
        }
        const uint8_t* src = this->hist.data();
        const size_t len = this->hist.size();
        int32_t sLimit = (int32_t)len - inputMargin;
        const int stepSize = 1;
        const int kSearchStrength = 8;
        int32_t nextEmit = s;
        uint64_t cv = load64(src, s);
        int32_t offset1 = (int32_t)blk->recentOffsets[0];
        int32_t offset2 = (int32_t)blk->recentOffsets[1];
        auto addLiterals = [&](Seq* sq, int32_t until) {
            if (until == nextEmit) return;
            blk->literals.insert(blk->literals.end(), src + nextEmit, src + until);
            sq->litLen = (uint32_t)(until - nextEmit);
        };
        bool finished = false;
        while (!finished) {  // encodeLoop
            int32_t t = 0;
            bool canRepeat = blk->sequences.size() > 2;
            for (;;) {
                uint32_t nextHashL = hashLen(cv, dFastLongTableBits, dFastLongLen);
                uint32_t nextHashS = hashLen(cv, dFastShortTableBits, dFastShortLen);
                TableEntry candidateL = longTable[nextHashL];
                TableEntry candidateS = table[nextHashS];
                const int repOff = 1;
                int32_t repIndex = s - offset1 + repOff;
                TableEntry entry{(uint32_t)cv, s + cur};
                longTable[nextHashL] = entry;
                table[nextHashS] = entry;
                if (canRepeat) {
                    if (repIndex >= 0 && load32(src, repIndex) == (uint32_t)(cv >> (repOff * 8))) {
                        Seq seq = {0, 0, 0, 0, 0, 0};
                        int32_t length = 4 + this->matchlen(s + 4 + repOff, repIndex + 4, src, len);
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
                        s += length + repOff;
                        nextEmit = s;
                        if (s >= sLimit) { finished = true; break; }
                        cv = load64(src, s);
                        continue;
                    }
                }
                int32_t coffsetL = s - (candidateL.offset - cur);
                int32_t coffsetS = s - (candidateS.offset - cur);
                if (coffsetL < maxMatchOff && (uint32_t)cv == candidateL.val) {
                    t = candidateL.offset - cur;
                    break;
                }
                if (coffsetS < maxMatchOff && (uint32_t)cv == candidateS.val) {
                    const int checkAt = 1;
                    uint64_t cv2 = load64(src, s + checkAt);
                    nextHashL = hashLen(cv2, dFastLongTableBits, dFastLongLen);
                    candidateL = longTable[nextHashL];
                    coffsetL = s - (candidateL.offset - cur) + checkAt;
                    longTable[nextHashL] = TableEntry{(uint32_t)cv2, s + checkAt + cur};
                    if (coffsetL < maxMatchOff && (uint32_t)cv2 == candidateL.val) {
                        t = candidateL.offset - cur;
                        s += checkAt;
                        break;
                    }
                    t = candidateS.offset - cur;
                    break;
                }
                s += stepSize + ((s - nextEmit) >> (kSearchStrength - 1));
                if (s >= sLimit) { finished = true; break; }
                cv = load64(src, s);
            }
            if (finished) break;
            offset2 = offset1;
            offset1 = s - t;
            int32_t l = this->matchlen(s + 4, t + 4, src, len) + 4;
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
            int32_t index0 = s - l + 1;
            int32_t index1 = s - 2;
            uint64_t cv0 = load64(src, index0);
            uint64_t cv1 = load64(src, index1);
            TableEntry te0{(uint32_t)cv0, index0 + cur};
            TableEntry te1{(uint32_t)cv1, index1 + cur};
            longTable[hashLen(cv0, dFastLongTableBits, dFastLongLen)] = te0;
            longTable[hashLen(cv1, dFastLongTableBits, dFastLongLen)] = te1;
            cv0 >>= 8;
            cv1 >>= 8;
            te0.offset++;
            te1.offset++;
            te0.val = (uint32_t)cv0;
            te1.val = (uint32_t)cv1;
            table[hashLen(cv0, dFastShortTableBits, dFastShortLen)] = te0;
            table[hashLen(cv1, dFastShortTableBits, dFastShortLen)] = te1;
            cv = load64(src, s);
            if (!canRepeat) continue;
            for (;;) {  // Check offset 2 (:283)
                int32_t o2 = s - offset2;
                if (load32(src, o2) != (uint32_t)cv) break;
                uint32_t nextHashS = hashLen(cv, dFastShortTableBits, dFastShortLen);
                uint32_t nextHashL = hashLen(cv, dFastLongTableBits, dFastLongLen);
                int32_t l2 = 4 + this->matchlen(s + 4, o2 + 4, src, len);
                TableEntry entry{(uint32_t)cv, s + cur};
                longTable[nextHashL] = entry;
                table[nextHashS] = entry;
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

    void dfEncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t len) {
        const int inputMargin = 8 + 2;
        const int minNonLiteralBlockSize = 16;
        auto& table = this->table;
        int32_t& cur = this->cur;
        const int32_t maxMatchOff = this->maxMatchOff;
        if (cur >= this->bufferReset) {
            for (auto& t : table) t = TableEntry{0, 0};
            for (auto& t : longTable) t = TableEntry{0, 0};
            cur = maxMatchOff;
        }
        int32_t s = 0;
        blk->size = (int)len;
        if ((int)len < minNonLiteralBlockSize) {
            blk->extraLits = (int)len;
            blk->literals.assign(src, src + len);
            return;
        }
        int32_t sLimit = (int32_t)len - inputMargin;
        const int stepSize = 1;
        const int kSearchStrength = 8;
        int32_t nextEmit = s;
        uint64_t cv = load64(src, s);
        int32_t offset1 = (int32_t)blk->recentOffsets[0];
        int32_t offset2 = (int32_t)blk->recentOffsets[1];
        auto addLiterals = [&](Seq* sq, int32_t until) {
            if (until == nextEmit) return;
            blk->literals.insert(blk->literals.end(), src + nextEmit, src + until);
            sq->litLen = (uint32_t)(until - nextEmit);
        };
        bool finished = false;
        while (!finished) {
            int32_t t = 0;
            for (;;) {
                uint32_t nextHashL = hashLen(cv, dFastLongTableBits, dFastLongLen);
                uint32_t nextHashS = hashLen(cv, dFastShortTableBits, dFastShortLen);
                TableEntry candidateL = longTable[nextHashL];
                TableEntry candidateS = table[nextHashS];
                const int repOff = 1;
                int32_t repIndex = s - offset1 + repOff;
                TableEntry entry{(uint32_t)cv, s + cur};
                longTable[nextHashL] = entry;
                table[nextHashS] = entry;
                if (blk->sequences.size() > 2) {
                    if (load32(src, repIndex) == (uint32_t)(cv >> (repOff * 8))) {
                        Seq seq = {0, 0, 0, 0, 0, 0};
                        int32_t length = 4 + (int32_t)matchLen(src + s + 4 + repOff, len - (size_t)(s + 4 + repOff), src + repIndex + 4);
                        seq.matchLen = (uint32_t)(length - zstdMinMatch);
                        int32_t start = s + repOff;
                        int32_t startLimit = nextEmit + 1;
                        int32_t tMin = std::max(s - maxMatchOff, (int32_t)0);
                        while (repIndex > tMin && start > startLimit && src[repIndex - 1] == src[start - 1]) {
                            repIndex--;
                            start--;
                            seq.matchLen++;
                        }
                        addLiterals(&seq, start);
                        seq.offset = 1;
                        blk->sequences.push_back(seq);
                        s += length + repOff;
                        nextEmit = s;
                        if (s >= sLimit) { finished = true; break; }
                        cv = load64(src, s);
                        continue;
                    }
                }
                int32_t coffsetL = s - (candidateL.offset - cur);
                int32_t coffsetS = s - (candidateS.offset - cur);
                if (coffsetL < maxMatchOff && (uint32_t)cv == candidateL.val) {
                    t = candidateL.offset - cur;
                    break;
                }
                if (coffsetS < maxMatchOff && (uint32_t)cv == candidateS.val) {
                    const int checkAt = 1;
                    uint64_t cv2 = load64(src, s + checkAt);
                    nextHashL = hashLen(cv2, dFastLongTableBits, dFastLongLen);
                    candidateL = longTable[nextHashL];
                    coffsetL = s - (candidateL.offset - cur) + checkAt;
                    longTable[nextHashL] = TableEntry{(uint32_t)cv2, s + checkAt + cur};
                    if (coffsetL < maxMatchOff && (uint32_t)cv2 == candidateL.val) {
                        t = candidateL.offset - cur;
                        s += checkAt;
                        break;
                    }
                    t = candidateS.offset - cur;
                    break;
                }
                s += stepSize + ((s - nextEmit) >> (kSearchStrength - 1));
                if (s >= sLimit) { finished = true; break; }
                cv = load64(src, s);
            }
            if (finished) break;
            offset2 = offset1;
            offset1 = s - t;
            int32_t l = (int32_t)matchLen(src + s + 4, len - (size_t)(s + 4), src + t + 4) + 4;
            int32_t tMin = std::max(s - maxMatchOff, (int32_t)0);
            while (t > tMin && s > nextEmit && src[t - 1] == src[s - 1]) { s--; t--; l++; }
            Seq seq = {0, 0, 0, 0, 0, 0};
            seq.litLen = (uint32_t)(s - nextEmit);
            seq.matchLen = (uint32_t)(l - zstdMinMatch);
            if (seq.litLen > 0) blk->literals.insert(blk->literals.end(), src + nextEmit, src + s);
            seq.offset = (uint32_t)(s - t) + 3;
            s += l;
            blk->sequences.push_back(seq);
            nextEmit = s;
            if (s >= sLimit) break;
            int32_t index0 = s - l + 1;
            int32_t index1 = s - 2;
            uint64_t cv0 = load64(src, index0);
            uint64_t cv1 = load64(src, index1);
            TableEntry te0{(uint32_t)cv0, index0 + cur};
            TableEntry te1{(uint32_t)cv1, index1 + cur};
            longTable[hashLen(cv0, dFastLongTableBits, dFastLongLen)] = te0;
            longTable[hashLen(cv1, dFastLongTableBits, dFastLongLen)] = te1;
            cv0 >>= 8;
            cv1 >>= 8;
            te0.offset++;
            te1.offset++;
            te0.val = (uint32_t)cv0;
            te1.val = (uint32_t)cv1;
            table[hashLen(cv0, dFastShortTableBits, dFastShortLen)] = te0;
            table[hashLen(cv1, dFastShortTableBits, dFastShortLen)] = te1;
            cv = load64(src, s);
            if (blk->sequences.size() <= 2) continue;
            for (;;) {  // Check offset 2 (:617) — note the cv1>>8 quirk (:630), App. A-7
                int32_t o2 = s - offset2;
                if (load32(src, o2) != (uint32_t)cv) break;
                uint32_t nextHashS = hashLen(cv1 >> 8, dFastShortTableBits, dFastShortLen);
                uint32_t nextHashL = hashLen(cv, dFastLongTableBits, dFastLongLen);
                int32_t l2 = 4 + (int32_t)matchLen(src + s + 4, len - (size_t)(s + 4), src + o2 + 4);
                TableEntry entry{(uint32_t)cv, s + cur};
                longTable[nextHashL] = entry;
                table[nextHashS] = entry;
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
        if (cur < this->bufferReset) cur += (int32_t)len;
    }
};

struct DoubleFastEncoder : DoubleFastT<FastEncoder> {  // enc_dfast.go:25
    void Encode(BlockEnc* blk, const uint8_t* src, size_t n) override { dfEncode(blk, src, n); }
    void EncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t n) override { dfEncodeNoHist(blk, src, n); }
    void Reset(const DictO* d, bool singleBlock) override { FastEncoder::Reset(d, singleBlock); }  // :1033
    // enc_dfast.go:1040 ResetPrefix: fastEncoder.ResetPrefix fills the SHORT table through the fast encoder's 6-byte hash (the
    // lookups of this encoder hash 5 bytes: the entries sit in other buckets than a lookup of the same bytes would read — kept
    // as the reference has it), then every 2nd position goes into the long table
    void ResetPrefix(const uint8_t* prefix, size_t n) override {
        FastEncoder::ResetPrefix(prefix, n);
        if (n < 8) return;
        const int32_t end = cur + (int32_t)n - 8;
        for (int32_t i = cur + 1; i < end; i += 2) {
            const uint64_t cv = load64(prefix, i - cur);
            longTable[hashLen(cv, dFastLongTableBits, dFastLongLen)] = TableEntry{(uint32_t)cv, i};
        }
    }
};

struct DoubleFastEncoderDict : DoubleFastT<FastEncoderDict> {  // enc_dfast.go:30
    std::vector<TableEntry> dictLongTable;
    void Encode(BlockEnc* blk, const uint8_t* src, size_t n) override { dfEncode(blk, src, n); }
    void EncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t n) override { dfEncodeNoHist(blk, src, n); }
    // enc_dfast.go:1053 Reset.  Dirty-shard tracking only avoids copying clean shards; the
    // resulting table always equals the pristine dict tables, so a full copy is equivalent.
    void Reset(const DictO* d, bool singleBlock) override {
        bool dictChanged = d != this->lastDict;
        FastEncoderDict::Reset(d, singleBlock);
        // Our encodeImpl does not record dirty marks for the dfast loops, so force a full
        // short-table restore (equivalent end state; see above).
        if (d != nullptr) this->table = this->dictTable;
        if (d == nullptr) return;
        if (dictLongTable.size() != longTable.size() || dictChanged) {
            dictLongTable.assign(longTable.size(), TableEntry{0, 0});
            if (d->content.size() >= 8) {
                uint64_t cv = load64(d->content.data(), 0);
                dictLongTable[hashLen(cv, dFastLongTableBits, dFastLongLen)] = TableEntry{(uint32_t)cv, maxMatchOff};
                int32_t end = (int32_t)d->content.size() - 8 + maxMatchOff;
                for (int32_t i = maxMatchOff + 1; i < end; i++) {
                    cv = cv >> 8 | ((uint64_t)d->content[(size_t)(i - maxMatchOff + 7)] << 56);
                    dictLongTable[hashLen(cv, dFastLongTableBits, dFastLongLen)] = TableEntry{(uint32_t)cv, i};
                }
            }
        }
        cur = maxMatchOff;
        longTable = dictLongTable;
    }
};

}  // namespace kco
