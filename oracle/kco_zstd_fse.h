// oracle/kco_zstd_fse.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates zstd/fse_encoder.go (fseEncoder, cState), zstd/fse_predefined.go:75-158
// (encoder side) and zstd/seqenc.go (code tables, seqCoders.setPrev).
#pragma once
#include "kco_common.h"

namespace kco {
namespace zfse {

constexpr int maxEncTableLog = 8, minEncTablelog = 5;
constexpr int maxLiteralLengthSymbol = 35, maxOffsetLengthSymbol = 30, maxMatchLengthSymbol = 52;

struct SymbolTransform {  // zstd/fse_encoder.go:48
    uint32_t deltaNbBits;
    int16_t deltaFindState;
    uint8_t outBits;
};

// zstd/seqenc.go:48-112 code tables.
static const uint8_t llCodeTable[64] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
    16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
    22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23,
    24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
static const uint8_t llBitsTable[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint8_t mlCodeTable[128] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
    16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
    32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37,
    38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40,
    41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41,
    42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42,
    42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};
static const uint8_t mlBitsTable[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

static inline uint8_t llCode(uint32_t litLength) {  // seqenc.go:70
    if (litLength <= 63) return llCodeTable[litLength & 63];
    return (uint8_t)((uint8_t)highBit(litLength) + 19);
}
static inline uint8_t mlCode(uint32_t mlBase) {  // seqenc.go:100
    if (mlBase <= 127) return mlCodeTable[mlBase & 127];
    return (uint8_t)((uint8_t)highBit(mlBase) + 36);
}
static inline uint8_t ofCode(uint32_t offset) {  // seqenc.go:108
    return (uint8_t)(bitsLen32(offset) - 1);
}

static inline uint32_t tableStep(uint32_t tableSize) { return (tableSize >> 1) + (tableSize >> 3) + 3; }

struct FseEncoder {  // zstd/fse_encoder.go:23
    uint16_t symbolLen = 0;
    uint8_t actualTableLog = 0;
    // cTable (fse_encoder.go:41): slices with tracked lengths
    uint8_t tableSymbol[256];
    uint16_t stateTable[256];
    int stateTableLen = 0;
    SymbolTransform symbolTT[256];
    int maxCount = 0;
    bool zeroBits = false, clearCount = false, useRLE = false, preDefined = false, reUsed = false;
    uint8_t rleVal = 0, maxBits = 0;
    uint32_t count[256];
    int16_t norm[256];

    FseEncoder() {
        memset(tableSymbol, 0, sizeof(tableSymbol));
        memset(stateTable, 0, sizeof(stateTable));
        memset(symbolTT, 0, sizeof(symbolTT));
        memset(count, 0, sizeof(count));
        memset(norm, 0, sizeof(norm));
    }

    // fse_encoder.go:71 HistogramFinished
    void HistogramFinished(uint8_t maxSymbol, int maxCnt) {
        maxCount = maxCnt;
        symbolLen = (uint16_t)maxSymbol + 1;
        clearCount = maxCnt != 0;
    }
    // fse_encoder.go:79 allocCtable
    void allocCtable() { stateTableLen = 1 << actualTableLog; }

    // fse_encoder.go:102 buildCTable; returns false on internal error
    bool buildCTable() {
        uint32_t tableSize = 1u << actualTableLog;
        uint32_t highThreshold = tableSize - 1;
        int16_t cumul[257];
        memset(cumul, 0, sizeof(cumul));
        allocCtable();
        {
            cumul[0] = 0;
            for (int ui = 0; ui < (int)symbolLen - 1; ui++) {
                int16_t v = norm[ui];
                uint8_t u = (uint8_t)ui;
                if (v == -1) {
                    cumul[u + 1] = (int16_t)(cumul[u] + 1);
                    tableSymbol[highThreshold] = u;
                    highThreshold--;
                } else {
                    cumul[u + 1] = (int16_t)(cumul[u] + v);
                }
            }
            int u = (int)symbolLen - 1;
            int16_t v = norm[symbolLen - 1];
            if (v == -1) {
                cumul[u + 1] = (int16_t)(cumul[u] + 1);
                tableSymbol[highThreshold] = (uint8_t)u;
                highThreshold--;
            } else {
                cumul[u + 1] = (int16_t)(cumul[u] + v);
            }
            if ((uint32_t)(int32_t)cumul[symbolLen] != tableSize) return false;
            cumul[symbolLen] = (int16_t)((int16_t)tableSize + 1);
        }
        zeroBits = false;
        {
            uint32_t step = tableStep(tableSize);
            uint32_t tableMask = tableSize - 1;
            uint32_t position = 0;
            int16_t largeLimit = (int16_t)(1 << (actualTableLog - 1));
            for (int ui = 0; ui < (int)symbolLen; ui++) {
                int16_t v = norm[ui];
                uint8_t symbol = (uint8_t)ui;
                if (v > largeLimit) zeroBits = true;
                for (int k = 0; k < (int)v; k++) {
                    tableSymbol[position] = symbol;
                    position = (position + step) & tableMask;
                    while (position > highThreshold) position = (position + step) & tableMask;
                }
            }
            if (position != 0) return false;
        }
        {
            int tsi = (int)tableSize;
            for (int u = 0; u < (int)tableSize; u++) {
                uint8_t v = tableSymbol[u];
                stateTable[cumul[v]] = (uint16_t)(tsi + u);
                cumul[v]++;
            }
        }
        {
            int16_t total = 0;
            uint8_t tableLog = actualTableLog;
            uint32_t tl = ((uint32_t)tableLog << 16) - (1u << tableLog);
            for (int i = 0; i < (int)symbolLen; i++) {
                int16_t v = norm[i];
                switch (v) {
                case 0: break;
                case -1:
                case 1:
                    symbolTT[i].deltaNbBits = tl;
                    symbolTT[i].deltaFindState = (int16_t)(total - 1);
                    total++;
                    break;
                default: {
                    uint32_t maxBitsOut = (uint32_t)tableLog - highBit((uint32_t)(int32_t)(int16_t)(v - 1));
                    uint32_t minStatePlus = (uint32_t)(int32_t)v << maxBitsOut;
                    symbolTT[i].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                    symbolTT[i].deltaFindState = (int16_t)(total - v);
                    total = (int16_t)(total + v);
                }
                }
            }
            if (total != (int16_t)tableSize) return false;
        }
        return true;
    }

    // fse_encoder.go:208 setRLE
    void setRLE(uint8_t val) {
        allocCtable();
        actualTableLog = 0;
        stateTableLen = 1;
        symbolTT[val].deltaFindState = 0;
        symbolTT[val].deltaNbBits = 0;
        symbolTT[val].outBits = 0;
        rleVal = val;
        useRLE = true;
    }
    // fse_encoder.go:225 setBits
    void setBits(const uint8_t* transform) {
        if (reUsed || preDefined) return;
        if (useRLE) {
            if (transform == nullptr) {
                symbolTT[rleVal].outBits = rleVal;
                maxBits = rleVal;
                return;
            }
            maxBits = transform[rleVal];
            symbolTT[rleVal].outBits = maxBits;
            return;
        }
        if (transform == nullptr) {
            for (int i = 0; i < (int)symbolLen; i++) symbolTT[i].outBits = (uint8_t)i;
            maxBits = (uint8_t)(symbolLen - 1);
            return;
        }
        maxBits = 0;
        for (int i = 0; i < (int)symbolLen; i++) {
            uint8_t v = transform[i];
            symbolTT[i].outBits = v;
            if (v > maxBits) maxBits = v;
        }
    }

    // fse_encoder.go:429 optimalTableLog
    void optimalTableLog(int length) {
        uint8_t tableLog = (uint8_t)maxEncTableLog;
        uint32_t minBitsSrc = highBit((uint32_t)length) + 1;
        uint32_t minBitsSymbols = highBit((uint32_t)(uint16_t)(symbolLen - 1)) + 2;
        uint8_t minBits = (uint8_t)minBitsSymbols;
        if (minBitsSrc < minBitsSymbols) minBits = (uint8_t)minBitsSrc;
        uint8_t maxBitsSrc = (uint8_t)((uint8_t)highBit((uint32_t)(length - 1)) - 2);
        if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
        if (minBits > tableLog) tableLog = minBits;
        if (tableLog < minEncTablelog) tableLog = minEncTablelog;
        if (tableLog > maxEncTableLog) tableLog = maxEncTableLog;
        actualTableLog = tableLog;
    }

    // fse_encoder.go:259 normalizeCount; returns false on error
    bool normalizeCount(int length) {
        static const uint32_t rtbTable[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
        if (reUsed) return true;
        optimalTableLog(length);
        uint8_t tableLog = actualTableLog;
        uint64_t scale = 62 - (uint64_t)tableLog;
        uint64_t step = ((uint64_t)1 << 62) / (uint64_t)length;
        uint64_t vStep = (uint64_t)1 << (scale - 20);
        int16_t stillToDistribute = (int16_t)(1 << tableLog);
        int largest = 0;
        int16_t largestP = 0;
        uint32_t lowThreshold = (uint32_t)(length >> tableLog);
        if (maxCount == length) { useRLE = true; return true; }
        useRLE = false;
        for (int i = 0; i < (int)symbolLen; i++) {
            uint32_t cnt = count[i];
            if (cnt == 0) { norm[i] = 0; continue; }
            if (cnt <= lowThreshold) {
                norm[i] = -1;
                stillToDistribute--;
            } else {
                int16_t proba = (int16_t)(((uint64_t)cnt * step) >> scale);
                if (proba < 8) {
                    uint64_t restToBeat = vStep * (uint64_t)rtbTable[proba];
                    uint64_t v = (uint64_t)cnt * step - ((uint64_t)proba << scale);
                    if (v > restToBeat) proba++;
                }
                if (proba > largestP) { largestP = proba; largest = i; }
                norm[i] = proba;
                stillToDistribute = (int16_t)(stillToDistribute - proba);
            }
        }
        if ((int16_t)(-stillToDistribute) >= (norm[largest] >> 1)) {
            if (!normalizeCount2(length)) return false;
            return buildCTable();
        }
        norm[largest] = (int16_t)(norm[largest] + stillToDistribute);
        return buildCTable();
    }

    // fse_encoder.go:334 normalizeCount2
    bool normalizeCount2(int length) {
        const int16_t notYetAssigned = -2;
        uint32_t distributed = 0;
        uint32_t total = (uint32_t)length;
        uint8_t tableLog = actualTableLog;
        uint32_t lowThreshold = total >> tableLog;
        uint32_t lowOne = (total * 3) >> (tableLog + 1);
        for (int i = 0; i < (int)symbolLen; i++) {
            uint32_t cnt = count[i];
            if (cnt == 0) { norm[i] = 0; continue; }
            if (cnt <= lowThreshold) { norm[i] = -1; distributed++; total -= cnt; continue; }
            if (cnt <= lowOne) { norm[i] = 1; distributed++; total -= cnt; continue; }
            norm[i] = notYetAssigned;
        }
        uint32_t toDistribute = (1u << tableLog) - distributed;
        if ((total / toDistribute) > lowOne) {
            lowOne = (total * 3) / (toDistribute * 2);
            for (int i = 0; i < (int)symbolLen; i++) {
                uint32_t cnt = count[i];
                if (norm[i] == notYetAssigned && cnt <= lowOne) { norm[i] = 1; distributed++; total -= cnt; continue; }
            }
            toDistribute = (1u << tableLog) - distributed;
        }
        if (distributed == (uint32_t)symbolLen + 1) {
            int maxV = 0;
            uint32_t maxC = 0;
            for (int i = 0; i < (int)symbolLen; i++)
                if (count[i] > maxC) { maxV = i; maxC = count[i]; }
            norm[maxV] = (int16_t)(norm[maxV] + (int16_t)toDistribute);
            return true;
        }
        if (total == 0) {
            for (uint32_t i = 0; toDistribute > 0; i = (i + 1) % (uint32_t)symbolLen) {
                if (norm[i] > 0) { toDistribute--; norm[i]++; }
            }
            return true;
        }
        uint64_t vStepLog = 62 - (uint64_t)tableLog;
        uint64_t mid = (uint64_t)(((uint64_t)1 << (vStepLog - 1)) - 1);
        uint64_t rStep = ((((uint64_t)1 << vStepLog) * (uint64_t)toDistribute) + mid) / (uint64_t)total;
        uint64_t tmpTotal = mid;
        for (int i = 0; i < (int)symbolLen; i++) {
            if (norm[i] == notYetAssigned) {
                uint64_t end = tmpTotal + (uint64_t)count[i] * rStep;
                uint32_t sStart = (uint32_t)(tmpTotal >> vStepLog);
                uint32_t sEnd = (uint32_t)(end >> vStepLog);
                uint32_t weight = sEnd - sStart;
                if (weight < 1) return false;
                norm[i] = (int16_t)weight;
                tmpTotal = end;
            }
        }
        return true;
    }

    // fse_encoder.go:488 writeCount — appends to out; returns false on error
    bool writeCount(Bytes* outv) {
        if (useRLE) { outv->push_back(rleVal); return true; }
        if (preDefined || reUsed) return true;
        uint8_t tableLog = actualTableLog;
        int tableSize = 1 << tableLog;
        bool previous0 = false;
        uint16_t charnum = 0;
        int maxHeaderSize = (((int)symbolLen * (int)tableLog) >> 3) + 3 + 2;
        uint32_t bitStream = (uint32_t)(tableLog - minEncTablelog);
        unsigned bitCount = 4;
        int16_t remaining = (int16_t)(tableSize + 1);
        int16_t threshold = (int16_t)tableSize;
        unsigned nbBits = (unsigned)(tableLog + 1);
        size_t outP = outv->size();
        size_t base = outP;
        outv->resize(base + (size_t)maxHeaderSize + 8, 0);
        Bytes& out = *outv;
        while (remaining > 1) {
            if (previous0) {
                uint16_t start = charnum;
                while (norm[charnum] == 0) charnum++;
                while (charnum >= start + 24) {
                    start += 24;
                    bitStream += (uint32_t)0xFFFF << bitCount;
                    out[outP] = (uint8_t)bitStream;
                    out[outP + 1] = (uint8_t)(bitStream >> 8);
                    outP += 2;
                    bitStream >>= 16;
                }
                while (charnum >= start + 3) {
                    start += 3;
                    bitStream += (uint32_t)3 << bitCount;
                    bitCount += 2;
                }
                bitStream += (uint32_t)(uint16_t)(charnum - start) << bitCount;
                bitCount += 2;
                if (bitCount > 16) {
                    out[outP] = (uint8_t)bitStream;
                    out[outP + 1] = (uint8_t)(bitStream >> 8);
                    outP += 2;
                    bitStream >>= 16;
                    bitCount -= 16;
                }
            }
            int16_t cnt = norm[charnum];
            charnum++;
            int16_t max = (int16_t)((2 * threshold - 1) - remaining);
            if (cnt < 0) remaining = (int16_t)(remaining + cnt);
            else remaining = (int16_t)(remaining - cnt);
            cnt++;
            if (cnt >= threshold) cnt = (int16_t)(cnt + max);
            bitStream += (uint32_t)(int32_t)cnt << bitCount;
            bitCount += nbBits;
            if (cnt < max) bitCount--;
            previous0 = cnt == 1;
            if (remaining < 1) return false;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
            if (bitCount > 16) {
                out[outP] = (uint8_t)bitStream;
                out[outP + 1] = (uint8_t)(bitStream >> 8);
                outP += 2;
                bitStream >>= 16;
                bitCount -= 16;
            }
        }
        if (outP + 2 > base + (size_t)maxHeaderSize) return false;
        out[outP] = (uint8_t)bitStream;
        out[outP + 1] = (uint8_t)(bitStream >> 8);
        outP += (size_t)((bitCount + 7) / 8);
        if (charnum > symbolLen) return false;
        out.resize(outP);
        return true;
    }

    // fse_encoder.go:603 bitCost
    uint32_t bitCost(uint8_t symbolValue, uint32_t accuracyLog) const {
        uint32_t minNbBits = symbolTT[symbolValue].deltaNbBits >> 16;
        uint32_t threshold = (minNbBits + 1) << 16;
        uint32_t tableSize = 1u << actualTableLog;
        uint32_t deltaFromThreshold = threshold - (symbolTT[symbolValue].deltaNbBits + tableSize);
        uint32_t normalizedDeltaFromThreshold = (deltaFromThreshold << accuracyLog) >> actualTableLog;
        uint32_t bitMultiplier = 1u << accuracyLog;
        return (minNbBits + 1) * bitMultiplier - normalizedDeltaFromThreshold;
    }
    // fse_encoder.go:633 approxSize
    uint32_t approxSize(const uint32_t* hist, int histLen) const {
        if ((int)symbolLen < histLen) return 0xFFFFFFFFu;
        if (useRLE) return 0xFFFFFFFFu;
        const uint32_t kAccuracyLog = 8;
        uint32_t badCost = ((uint32_t)actualTableLog + 1) << kAccuracyLog;
        uint32_t cost = 0;
        for (int i = 0; i < histLen; i++) {
            uint32_t v = hist[i];
            if (v == 0) continue;
            if (norm[i] == 0) return 0xFFFFFFFFu;
            uint32_t bc = bitCost((uint8_t)i, kAccuracyLog);
            if (bc > badCost) return 0xFFFFFFFFu;
            cost += v * bc;
        }
        return cost >> kAccuracyLog;
    }
    // fse_encoder.go:663 maxHeaderSize
    uint32_t maxHeaderSize() const {
        if (preDefined) return 0;
        if (useRLE) return 8;
        return ((((uint32_t)symbolLen * (uint32_t)actualTableLog) >> 3) + 3) * 8;
    }
};

struct CState {  // fse_encoder.go:675
    BitWriter* bw;
    uint16_t* stateTable;
    int stateTableLen;
    uint16_t state;
    void init(BitWriter* w, FseEncoder* enc, SymbolTransform first) {  // :682
        bw = w;
        stateTable = enc->stateTable;
        stateTableLen = enc->stateTableLen;
        if (stateTableLen == 1) {
            stateTable[0] = 0;
            state = 0;
            return;
        }
        uint32_t nbBitsOut = (first.deltaNbBits + (1u << 15)) >> 16;
        int32_t im = (int32_t)((nbBitsOut << 16) - first.deltaNbBits);
        int32_t lu = (im >> nbBitsOut) + (int32_t)first.deltaFindState;
        state = stateTable[lu];
    }
    void flush(uint8_t tableLog) {  // :698
        bw->flush32();
        bw->addBits16NC(state, tableLog);
    }
};

// fse_predefined.go:112-155 (encoder half): predefined LL / OF / ML encoders.
struct Predef {
    FseEncoder enc[3];  // tableLiteralLengths=0, tableOffsets=1, tableMatchLengths=2
    Predef() {
        static const int16_t llNorm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1,
            2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
        static const int16_t ofNorm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1,
            1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
        static const int16_t mlNorm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1,
            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
        memcpy(enc[0].norm, llNorm, sizeof(llNorm)); enc[0].symbolLen = 36; enc[0].actualTableLog = 6;
        memcpy(enc[1].norm, ofNorm, sizeof(ofNorm)); enc[1].symbolLen = 29; enc[1].actualTableLog = 5;
        memcpy(enc[2].norm, mlNorm, sizeof(mlNorm)); enc[2].symbolLen = 53; enc[2].actualTableLog = 6;
        for (int i = 0; i < 3; i++) enc[i].buildCTable();
        enc[0].setBits(llBitsTable);
        enc[1].setBits(nullptr);
        enc[2].setBits(mlBitsTable);
        for (int i = 0; i < 3; i++) enc[i].preDefined = true;
    }
};
static inline Predef& predef() { static Predef p; return p; }

}  // namespace zfse
}  // namespace kco
