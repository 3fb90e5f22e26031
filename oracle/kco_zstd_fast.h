// oracle/kco_zstd_fast.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates zstd/enc_base.go (fastBase), zstd/hash.go (hashLen), zstd/matchlen_generic.go
// and zstd/enc_fast.go (fastEncoder.Encode / EncodeNoHist / Reset, fastEncoderDict).
#pragma once
#include "kco_common.h"
#include "kco_xxhash.h"
#include "kco_zstd_block.h"

namespace kco {

// zstd/hash.go:7-35
constexpr uint64_t prime3bytes = 506832829ULL, prime4bytes = 2654435761ULL, prime5bytes = 889523592379ULL,
                   prime6bytes = 227718039650203ULL, prime7bytes = 58295818150454627ULL, prime8bytes = 0xcf1bbcdcb7a56463ULL;
static inline uint32_t hashLen(uint64_t u, uint8_t length, uint8_t mls) {
    switch (mls) {
    case 3: return ((uint32_t)(u << 8) * (uint32_t)prime3bytes) >> (32 - length);
    case 5: return (uint32_t)(((u << (64 - 40)) * prime5bytes) >> (64 - length));
    case 6: return (uint32_t)(((u << (64 - 48)) * prime6bytes) >> (64 - length));
    case 7: return (uint32_t)(((u << (64 - 56)) * prime7bytes) >> (64 - length));
    case 8: return (uint32_t)((u * prime8bytes) >> (64 - length));
    default: return ((uint32_t)u * (uint32_t)prime4bytes) >> (32 - length);
    }
}

// zstd/matchlen_generic.go:17 matchLen(a, b): a is the shorter; lengths given explicitly.
static inline int matchLen(const uint8_t* a, size_t alen, const uint8_t* b) {
    int n = 0;
    size_t left = alen;
    while (left >= 8) {
        uint64_t diff = load64(a, n) ^ load64(b, n);
        if (diff != 0) return n + (tz64(diff) >> 3);
        n += 8;
        left -= 8;
    }
    for (size_t i = 0; i < left; i++) {
        if (a[n] != b[n]) break;
        n++;
    }
    return n;
}

constexpr int dictShardBits = 7;

struct DictO {  // zstd/dict.go:15 (encoder-relevant fields only)
    uint32_t id = 0;
    huff0::Scratch* litEnc = nullptr;
    int offsets[3] = {1, 4, 8};
    Bytes content;
};

struct FastBase {  // zstd/enc_base.go:14
    int32_t cur = 0;
    int32_t maxMatchOff = 0;
    int32_t bufferReset = 0;
    Bytes hist;
    size_t histCap = 0;  // cap(e.hist)
    XXH64 crc;
    BlockEnc blk;
    const DictO* lastDict = nullptr;
    bool lowMem = false;

    virtual ~FastBase() {}
    void setup(int windowSize, bool lowMem_) {  // encoder_options.go:51-73
        maxMatchOff = (int32_t)windowSize;
        bufferReset = (int32_t)(0x7fffffff - (int32_t)(windowSize * 2));
        lowMem = lowMem_;
    }
    // enc_base.go:34 AppendCRC
    void AppendCRC(Bytes* dst) {
        uint64_t s = crc.Sum64();
        dst->push_back((uint8_t)s);
        dst->push_back((uint8_t)(s >> 8));
        dst->push_back((uint8_t)(s >> 16));
        dst->push_back((uint8_t)(s >> 24));
    }
    // enc_base.go:42 WindowSize
    int32_t WindowSize(int64_t size) const {
        if (size > 0 && size < (int64_t)maxMatchOff) {
            int32_t b = (int32_t)1 << (unsigned)bitsLen64((uint64_t)size);
            if (b < 1024) b = 1024;
            return b;
        }
        return maxMatchOff;
    }
    // enc_base.go:84 ensureHist
    void ensureHist(int n) {
        if ((int64_t)histCap >= (int64_t)n) return;
        int32_t l = maxMatchOff;
        if ((lowMem && maxMatchOff > maxCompressedBlockSize) || maxMatchOff <= maxCompressedBlockSize) l += maxCompressedBlockSize;
        else l += maxMatchOff;
        if (l < (1 << 20) && !lowMem) l = 1 << 20;
        if (l < (int32_t)n) l = (int32_t)n;
        hist.clear();
        histCap = (size_t)l;
    }
    // enc_base.go:57 addBlock
    int32_t addBlock(const uint8_t* src, size_t n) {
        if (hist.size() + n > histCap) {
            if (histCap == 0) {
                ensureHist((int)n);
            } else {
                int32_t offset = (int32_t)hist.size() - maxMatchOff;
                memmove(hist.data(), hist.data() + offset, (size_t)maxMatchOff);
                cur += offset;
                hist.resize((size_t)maxMatchOff);
            }
        }
        int32_t s = (int32_t)hist.size();
        hist.insert(hist.end(), src, src + n);
        return s;
    }
    int32_t matchlen(int32_t s, int32_t t, const uint8_t* src, size_t srcLen) const {  // enc_base.go:111
        return (int32_t)matchLen(src + s, srcLen - (size_t)s, src + t);
    }
    // enc_base.go:160 resetBase
    void resetBase(const DictO* d, bool singleBlock) {
        blk.reset(nullptr);
        blk.initNewEncode();
        crc.Reset();
        blk.dictLitEnc = nullptr;
        if (d != nullptr) {
            bool low = lowMem;
            if (singleBlock) lowMem = true;
            ensureHist((int)d->content.size() + maxCompressedBlockSize);
            lowMem = low;
        }
        if (cur < bufferReset) cur += maxMatchOff + (int32_t)hist.size();
        hist.clear();
        if (d != nullptr) {
            for (int i = 0; i < 3; i++) {
                blk.recentOffsets[i] = (uint32_t)d->offsets[i];
                blk.prevRecentOffsets[i] = blk.recentOffsets[i];
            }
            blk.dictLitEnc = d->litEnc;
            hist.insert(hist.end(), d->content.begin(), d->content.end());
        }
    }
    // enc_base.go:134 resetBasePrefix (WithConcurrentBlocks jobs: history = the overlap prefix of the previous job)
    void resetBasePrefix(const uint8_t* prefix, size_t n) {
        blk.reset(nullptr);
        blk.initNewEncode();
        crc.Reset();
        blk.dictLitEnc = nullptr;
        ensureHist((int)n + maxCompressedBlockSize);
        // Bump cur so old table entries fall outside the window (:149-154)
        if (cur < bufferReset) cur += maxMatchOff + (int32_t)hist.size();
        hist.clear();
        hist.insert(hist.end(), prefix, prefix + n);
    }
    // zstd/encoder.go:32 encoder interface
    virtual void ResetPrefix(const uint8_t* prefix, size_t n) = 0;
    virtual void Encode(BlockEnc* b, const uint8_t* src, size_t n) = 0;
    virtual void EncodeNoHist(BlockEnc* b, const uint8_t* src, size_t n) = 0;
    virtual void Reset(const DictO* d, bool singleBlock) = 0;
};

constexpr int tableBits = 15, tableSize = 1 << tableBits, tableFastHashLen = 6;
constexpr int tableShardCnt = 1 << (tableBits - dictShardBits), tableShardSize = tableSize / tableShardCnt;
constexpr int maxMatchLength = 131074;

struct TableEntry { uint32_t val; int32_t offset; };

struct FastEncoder : FastBase {  // zstd/enc_fast.go:26
    std::vector<TableEntry> table;
    FastEncoder() : table(tableSize, TableEntry{0, 0}) {}

    // enc_fast.go:39 Encode; kSearchStrength parameterised for the dict small-input variant (:585)
    void encodeImpl(BlockEnc* blk, const uint8_t* srcIn, size_t srcLen, const int kSearchStrength, bool markDirty);
    void Encode(BlockEnc* blk, const uint8_t* src, size_t n) override { encodeImpl(blk, src, n, 6, false); }
    void EncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t n) override;
    void Reset(const DictO* d, bool singleBlock) override { resetBase(d, singleBlock); }  // :793 (panics on dict)
    // enc_fast.go:800 ResetPrefix: the prefix becomes history, every 4th position of it is indexed
    void ResetPrefix(const uint8_t* prefix, size_t n) override {
        resetBasePrefix(prefix, n);
        if (n < 8) return;
        const int32_t end = cur + (int32_t)n - 8;
        for (int32_t i = cur + 1; i < end; i += 4) {
            const uint64_t cv = load64(prefix, i - cur);
            table[hashLen(cv, tableBits, tableFastHashLen)] = TableEntry{(uint32_t)cv, i};
        }
    }
    virtual void markShardDirty(uint32_t) {}
};

inline void FastEncoder::encodeImpl(BlockEnc* blk, const uint8_t* srcIn, size_t srcLen, const int kSearchStrength, bool markDirty) {
    const int inputMargin = 8;
    const int minNonLiteralBlockSize = 1 + 1 + inputMargin;
    // Protect against e.cur wraparound (:45)
    while (cur >= bufferReset - (int32_t)hist.size()) {
        if (hist.empty()) {
            for (auto& t : table) t = TableEntry{0, 0};
            cur = maxMatchOff;
            break;
        }
        int32_t minOff = cur + (int32_t)hist.size() - maxMatchOff;
        for (auto& t : table) {
            int32_t v = t.offset;
            if (v < minOff) v = 0;
            else v = v - cur + maxMatchOff;
            t.offset = v;
        }
        cur = maxMatchOff;
        break;
    }
    int32_t s = addBlock(srcIn, srcLen);
    blk->size = (int)srcLen;
    if ((int)srcLen < minNonLiteralBlockSize) {
        blk->extraLits = (int)srcLen;
        blk->literals.assign(srcIn, srcIn + srcLen);
        return;
    }
    const uint8_t* src = hist.data();
    const size_t len = hist.size();
    int32_t sLimit = (int32_t)len - inputMargin;
    const int stepSize = 2;
    const uint8_t hashLog = tableBits;
    int32_t nextEmit = s;
    uint64_t cv = load64(src, s);
    int32_t offset1 = (int32_t)blk->recentOffsets[0];
    int32_t offset2 = (int32_t)blk->recentOffsets[1];

    auto addLiterals = [&](Seq* sq, int32_t until) {
        if (until == nextEmit) return;
        blk->literals.insert(blk->literals.end(), src + nextEmit, src + until);
        sq->litLen = (uint32_t)(until - nextEmit);
    };

    for (;;) {  // encodeLoop
        int32_t t = 0;
        bool canRepeat = blk->sequences.size() > 2;
        bool done = false;
        for (;;) {
            uint32_t nextHash = hashLen(cv, hashLog, tableFastHashLen);
            uint32_t nextHash2 = hashLen(cv >> 8, hashLog, tableFastHashLen);
            TableEntry candidate = table[nextHash];
            TableEntry candidate2 = table[nextHash2];
            int32_t repIndex = s - offset1 + 2;
            table[nextHash] = TableEntry{(uint32_t)cv, s + cur};
            table[nextHash2] = TableEntry{(uint32_t)(cv >> 8), s + cur + 1};
            if (markDirty) { markShardDirty(nextHash); markShardDirty(nextHash2); }

            if (canRepeat && repIndex >= 0 && load32(src, repIndex) == (uint32_t)(cv >> 16)) {
                Seq seq = {0, 0, 0, 0, 0, 0};
                int32_t length = 4 + matchlen(s + 6, repIndex + 4, src, len);
                seq.matchLen = (uint32_t)(length - zstdMinMatch);
                int32_t start = s + 2;
                int32_t startLimit = nextEmit + 1;
                int32_t sMin = std::max(s - maxMatchOff, (int32_t)0);
                while (repIndex > sMin && start > startLimit && src[repIndex - 1] == src[start - 1] &&
                       seq.matchLen < (uint32_t)(maxMatchLength - zstdMinMatch)) {
                    repIndex--;
                    start--;
                    seq.matchLen++;
                }
                addLiterals(&seq, start);
                seq.offset = 1;
                blk->sequences.push_back(seq);
                s += length + 2;
                nextEmit = s;
                if (s >= sLimit) { done = true; break; }
                cv = load64(src, s);
                continue;
            }
            int32_t coffset0 = s - (candidate.offset - cur);
            int32_t coffset1 = s - (candidate2.offset - cur) + 1;
            if (coffset0 < maxMatchOff && (uint32_t)cv == candidate.val) {
                t = candidate.offset - cur;
                break;
            }
            if (coffset1 < maxMatchOff && (uint32_t)(cv >> 8) == candidate2.val) {
                t = candidate2.offset - cur;
                s++;
                break;
            }
            s += stepSize + ((s - nextEmit) >> (kSearchStrength - 1));
            if (s >= sLimit) { done = true; break; }
            cv = load64(src, s);
        }
        if (done) break;
        offset2 = offset1;
        offset1 = s - t;
        int32_t l = matchlen(s + 4, t + 4, src, len) + 4;
        int32_t tMin = std::max(s - maxMatchOff, (int32_t)0);
        while (t > tMin && s > nextEmit && src[t - 1] == src[s - 1] && l < maxMatchLength) {
            s--;
            t--;
            l++;
        }
        Seq seq = {0, 0, 0, 0, 0, 0};
        seq.litLen = (uint32_t)(s - nextEmit);
        seq.matchLen = (uint32_t)(l - zstdMinMatch);
        if (seq.litLen > 0) blk->literals.insert(blk->literals.end(), src + nextEmit, src + s);
        seq.offset = (uint32_t)(s - t) + 3;
        s += l;
        blk->sequences.push_back(seq);
        nextEmit = s;
        if (s >= sLimit) break;
        cv = load64(src, s);

        // Check offset 2 (:250)
        int32_t o2 = s - offset2;
        if (canRepeat && load32(src, o2) == (uint32_t)cv) {
            int32_t l2 = 4 + matchlen(s + 4, o2 + 4, src, len);
            uint32_t nextHash = hashLen(cv, hashLog, tableFastHashLen);
            table[nextHash] = TableEntry{(uint32_t)cv, s + cur};
            if (markDirty) markShardDirty(nextHash);
            seq.matchLen = (uint32_t)l2 - zstdMinMatch;
            seq.litLen = 0;
            seq.offset = 1;
            s += l2;
            nextEmit = s;
            blk->sequences.push_back(seq);
            std::swap(offset1, offset2);
            if (s >= sLimit) break;
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

// enc_fast.go:294 EncodeNoHist
inline void FastEncoder::EncodeNoHist(BlockEnc* blk, const uint8_t* src, size_t len) {
    const int inputMargin = 8;
    const int minNonLiteralBlockSize = 1 + 1 + inputMargin;
    if (cur >= bufferReset) {
        for (auto& t : table) t = TableEntry{0, 0};
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
    const int stepSize = 2;
    const uint8_t hashLog = tableBits;
    const int kSearchStrength = 6;
    int32_t nextEmit = s;
    uint64_t cv = load64(src, s);
    int32_t offset1 = (int32_t)blk->recentOffsets[0];
    int32_t offset2 = (int32_t)blk->recentOffsets[1];
    auto addLiterals = [&](Seq* sq, int32_t until) {
        if (until == nextEmit) return;
        blk->literals.insert(blk->literals.end(), src + nextEmit, src + until);
        sq->litLen = (uint32_t)(until - nextEmit);
    };
    for (;;) {
        int32_t t = 0;
        bool done = false;
        for (;;) {
            uint32_t nextHash = hashLen(cv, hashLog, tableFastHashLen);
            uint32_t nextHash2 = hashLen(cv >> 8, hashLog, tableFastHashLen);
            TableEntry candidate = table[nextHash];
            TableEntry candidate2 = table[nextHash2];
            int32_t repIndex = s - offset1 + 2;
            table[nextHash] = TableEntry{(uint32_t)cv, s + cur};
            table[nextHash2] = TableEntry{(uint32_t)(cv >> 8), s + cur + 1};
            if (blk->sequences.size() > 2 && load32(src, repIndex) == (uint32_t)(cv >> 16)) {
                Seq seq = {0, 0, 0, 0, 0, 0};
                int32_t length = 4 + matchlen(s + 6, repIndex + 4, src, len);
                seq.matchLen = (uint32_t)(length - zstdMinMatch);
                int32_t start = s + 2;
                int32_t startLimit = nextEmit + 1;
                int32_t sMin = std::max(s - maxMatchOff, (int32_t)0);
                while (repIndex > sMin && start > startLimit && src[repIndex - 1] == src[start - 1]) {
                    repIndex--;
                    start--;
                    seq.matchLen++;
                }
                addLiterals(&seq, start);
                seq.offset = 1;
                blk->sequences.push_back(seq);
                s += length + 2;
                nextEmit = s;
                if (s >= sLimit) { done = true; break; }
                cv = load64(src, s);
                continue;
            }
            int32_t coffset0 = s - (candidate.offset - cur);
            int32_t coffset1 = s - (candidate2.offset - cur) + 1;
            if (coffset0 < maxMatchOff && (uint32_t)cv == candidate.val) {
                t = candidate.offset - cur;
                break;
            }
            if (coffset1 < maxMatchOff && (uint32_t)(cv >> 8) == candidate2.val) {
                t = candidate2.offset - cur;
                s++;
                break;
            }
            s += stepSize + ((s - nextEmit) >> (kSearchStrength - 1));
            if (s >= sLimit) { done = true; break; }
            cv = load64(src, s);
        }
        if (done) break;
        offset2 = offset1;
        offset1 = s - t;
        int32_t l = matchlen(s + 4, t + 4, src, len) + 4;
        int32_t tMin = std::max(s - maxMatchOff, (int32_t)0);
        while (t > tMin && s > nextEmit && src[t - 1] == src[s - 1]) {
            s--;
            t--;
            l++;
        }
        Seq seq = {0, 0, 0, 0, 0, 0};
        seq.litLen = (uint32_t)(s - nextEmit);
        seq.matchLen = (uint32_t)(l - zstdMinMatch);
        if (seq.litLen > 0) blk->literals.insert(blk->literals.end(), src + nextEmit, src + s);
        seq.offset = (uint32_t)(s - t) + 3;
        s += l;
        blk->sequences.push_back(seq);
        nextEmit = s;
        if (s >= sLimit) break;
        cv = load64(src, s);
        int32_t o2 = s - offset2;
        if (blk->sequences.size() > 2 && load32(src, o2) == (uint32_t)cv) {
            int32_t l2 = 4 + matchlen(s + 4, o2 + 4, src, len);
            uint32_t nextHash = hashLen(cv, hashLog, tableFastHashLen);
            table[nextHash] = TableEntry{(uint32_t)cv, s + cur};
            seq.matchLen = (uint32_t)l2 - zstdMinMatch;
            seq.litLen = 0;
            seq.offset = 1;
            s += l2;
            nextEmit = s;
            blk->sequences.push_back(seq);
            std::swap(offset1, offset2);
            if (s >= sLimit) break;
            cv = load64(src, s);
        }
    }
    if ((size_t)nextEmit < len) {
        blk->literals.insert(blk->literals.end(), src + nextEmit, src + len);
        blk->extraLits = (int)len - (int)nextEmit;
    }
    // We do not store history, so we must offset e.cur to avoid false matches for next user.
    if (cur < bufferReset) cur += (int32_t)len;
}

struct FastEncoderDict : FastEncoder {  // zstd/enc_fast.go:31
    std::vector<TableEntry> dictTable;
    bool tableShardDirty[tableShardCnt];
    bool allDirty = false;
    FastEncoderDict() { memset(tableShardDirty, 0, sizeof(tableShardDirty)); }
    void markShardDirty(uint32_t entryNum) override { tableShardDirty[entryNum / tableShardSize] = true; }
    // enc_fast.go:534 Encode
    void Encode(BlockEnc* blk, const uint8_t* src, size_t n) override {
        if (allDirty || n > (32 << 10)) {
            encodeImpl(blk, src, n, 6, false);
            allDirty = true;
            return;
        }
        encodeImpl(blk, src, n, 7, true);
    }
    // enc_fast.go:813 Reset
    void Reset(const DictO* d, bool singleBlock) override {
        resetBase(d, singleBlock);
        if (d == nullptr) return;
        if (dictTable.size() != table.size() || d != lastDict) {
            dictTable.assign(table.size(), TableEntry{0, 0});
            int32_t end = maxMatchOff + (int32_t)d->content.size() - 8;
            for (int32_t i = maxMatchOff; i < end; i += 2) {
                uint64_t cv = load64(d->content.data(), i - maxMatchOff);
                uint32_t nextHash = hashLen(cv, tableBits, tableFastHashLen);
                uint32_t nextHash1 = hashLen(cv >> 8, tableBits, tableFastHashLen);
                dictTable[nextHash] = TableEntry{(uint32_t)cv, i};
                dictTable[nextHash1] = TableEntry{(uint32_t)(cv >> 8), i + 1};
            }
            lastDict = d;
            allDirty = true;
        }
        cur = maxMatchOff;
        int dirtyShardCnt = 0;
        if (!allDirty)
            for (int i = 0; i < tableShardCnt; i++) if (tableShardDirty[i]) dirtyShardCnt++;
        if (allDirty || dirtyShardCnt > tableShardCnt * 4 / 6) {
            table = dictTable;
            memset(tableShardDirty, 0, sizeof(tableShardDirty));
            allDirty = false;
            return;
        }
        for (int i = 0; i < tableShardCnt; i++) {
            if (!tableShardDirty[i]) continue;
            std::copy(dictTable.begin() + i * tableShardSize, dictTable.begin() + (i + 1) * tableShardSize, table.begin() + i * tableShardSize);
            tableShardDirty[i] = false;
        }
        allDirty = false;
    }
};

}  // namespace kco
