// oracle/kco_fse_bytes.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates the compress side of the reference's generic byte FSE package (fse/compress.go,
// fse/fse.go, fse/bitwriter.go).  On the hot path it is used only to compress Huffman
// weights (huff0/huff0.go:211-229).
#pragma once
#include "kco_common.h"

namespace kco {
namespace fseb {

enum Err { OK = 0, ErrIncompressible = 1, ErrUseRLE = 2, ErrInternal = 3 };

constexpr int maxMemoryUsage = 14, defaultMemoryUsage = 13;
constexpr int maxTableLog = maxMemoryUsage - 2, defaultTablelog = defaultMemoryUsage - 2, minTablelog = 5;

struct SymbolTransform {  // fse/compress.go:308
    int32_t deltaFindState;
    uint32_t deltaNbBits;
};

// fse/fse.go:140 tableStep
static inline uint32_t tableStep(uint32_t tableSize) { return (tableSize >> 1) + (tableSize >> 3) + 3; }

struct CState {  // fse/compress.go:80
    BitWriter* bw;
    const uint16_t* stateTable;
    uint16_t state;
    void init(BitWriter* w, const uint16_t* st, SymbolTransform first) {  // :87
        bw = w;
        stateTable = st;
        uint32_t nbBitsOut = (first.deltaNbBits + (1u << 15)) >> 16;
        int32_t im = (int32_t)((nbBitsOut << 16) - first.deltaNbBits);
        int32_t lu = (im >> nbBitsOut) + first.deltaFindState;
        state = stateTable[lu];
    }
    void encode(SymbolTransform tt) {  // :98
        uint32_t nbBitsOut = ((uint32_t)state + tt.deltaNbBits) >> 16;
        int32_t dstState = (int32_t)(state >> (nbBitsOut & 15)) + tt.deltaFindState;
        bw->addBits16NC(state, (uint8_t)nbBitsOut);
        state = stateTable[dstState];
    }
    void encodeZero(SymbolTransform tt) {  // :106
        uint32_t nbBitsOut = ((uint32_t)state + tt.deltaNbBits) >> 16;
        int32_t dstState = (int32_t)(state >> (nbBitsOut & 15)) + tt.deltaFindState;
        bw->addBits16ZeroNC(state, (uint8_t)nbBitsOut);
        state = stateTable[dstState];
    }
    void flush(uint8_t tableLog) {  // :114
        bw->flush32();
        bw->addBits16NC(state, tableLog);
        bw->flush();
    }
};

struct Scratch {  // fse/fse.go:46
    uint32_t count[256] = {0};
    int16_t norm[256] = {0};
    BitWriter bw;
    uint8_t tableSymbol[1 << maxTableLog];
    uint16_t stateTable[1 << maxTableLog];
    SymbolTransform symbolTT[256];
    int maxCount = 0;
    Bytes Out;
    uint16_t symbolLen = 0;
    uint8_t actualTableLog = 0;
    bool zeroBits = false, clearCount = false;
    uint8_t MaxSymbolValue = 0, TableLog = 0;
    int64_t remain = 0;  // s.br.remain() == len(in): byteReader is init'ed at offset 0

    Scratch() { memset(symbolTT, 0, sizeof(symbolTT)); memset(tableSymbol, 0, sizeof(tableSymbol)); memset(stateTable, 0, sizeof(stateTable)); }

    // fse/fse.go:96 HistogramFinished
    void HistogramFinished(uint8_t maxSymbol, int maxCnt) {
        maxCount = maxCnt;
        symbolLen = (uint16_t)maxSymbol + 1;
        clearCount = maxCnt != 0;
    }
    // fse/fse.go:103 prepare
    void prepare(const uint8_t* in, size_t n) {
        (void)in;
        if (MaxSymbolValue == 0) MaxSymbolValue = 255;
        if (TableLog == 0) TableLog = defaultTablelog;
        if (clearCount && maxCount == 0) {
            memset(count, 0, sizeof(count));
            clearCount = false;
        }
        remain = (int64_t)n;
    }
    // fse/compress.go:453 countSimple
    int countSimple(const uint8_t* in, size_t n) {
        for (size_t i = 0; i < n; i++) count[in[i]]++;
        uint32_t m = 0;
        uint16_t symlen = symbolLen;
        for (int i = 0; i < 256; i++) {
            uint32_t v = count[i];
            if (v == 0) continue;
            if (v > m) m = v;
            symlen = (uint16_t)i + 1;
        }
        symbolLen = symlen;
        return (int)m;
    }
    // fse/compress.go:474 minTableLog
    uint8_t minTableLog() {
        uint32_t minBitsSrc = highBit((uint32_t)(remain - 1)) + 1;
        uint32_t minBitsSymbols = highBit((uint32_t)(uint16_t)(symbolLen - 1)) + 2;
        if (minBitsSrc < minBitsSymbols) return (uint8_t)minBitsSrc;
        return (uint8_t)minBitsSymbols;
    }
    // fse/compress.go:484 optimalTableLog
    void optimalTableLog() {
        uint8_t tableLog = TableLog;
        uint8_t minBits = minTableLog();
        uint8_t maxBitsSrc = (uint8_t)((uint8_t)highBit((uint32_t)(remain - 1)) - 2);
        if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
        if (minBits > tableLog) tableLog = minBits;
        if (tableLog < minTablelog) tableLog = minTablelog;
        if (tableLog > maxTableLog) tableLog = maxTableLog;
        actualTableLog = tableLog;
    }
    // fse/compress.go:510 normalizeCount
    Err normalizeCount() {
        static const uint32_t rtbTable[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
        uint8_t tableLog = actualTableLog;
        uint64_t scale = 62 - (uint64_t)tableLog;
        uint64_t step = ((uint64_t)1 << 62) / (uint64_t)remain;
        uint64_t vStep = (uint64_t)1 << (scale - 20);
        int16_t stillToDistribute = (int16_t)(1 << tableLog);
        int largest = 0;
        int16_t largestP = 0;
        uint32_t lowThreshold = (uint32_t)(remain >> tableLog);
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
        if ((int16_t)(-stillToDistribute) >= (norm[largest] >> 1)) return normalizeCount2();
        norm[largest] = (int16_t)(norm[largest] + stillToDistribute);
        return OK;
    }
    // fse/compress.go:557 normalizeCount2
    Err normalizeCount2() {
        const int16_t notYetAssigned = -2;
        uint32_t distributed = 0;
        uint32_t total = (uint32_t)remain;
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
            for (int i = 0; i < (int)symbolLen; i++) {
                if (count[i] > maxC) { maxV = i; maxC = count[i]; }
            }
            norm[maxV] = (int16_t)(norm[maxV] + (int16_t)toDistribute);
            return OK;
        }
        if (total == 0) {
            for (uint32_t i = 0; toDistribute > 0; i = (i + 1) % (uint32_t)symbolLen) {
                if (norm[i] > 0) { toDistribute--; norm[i]++; }
            }
            return OK;
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
                if (weight < 1) return ErrInternal;
                norm[i] = (int16_t)weight;
                tmpTotal = end;
            }
        }
        return OK;
    }
    // fse/compress.go:208 writeCount
    Err writeCount() {
        uint8_t tableLog = actualTableLog;
        int tableSize = 1 << tableLog;
        bool previous0 = false;
        uint16_t charnum = 0;
        int maxHeaderSize = (((int)symbolLen * (int)tableLog + 4 + 2) >> 3) + 3;
        uint32_t bitStream = (uint32_t)(tableLog - minTablelog);
        unsigned bitCount = 4;
        int16_t remaining = (int16_t)(tableSize + 1);
        int16_t threshold = (int16_t)tableSize;
        unsigned nbBits = (unsigned)(tableLog + 1);
        size_t outP = 0;
        Bytes out((size_t)maxHeaderSize + 8, 0);
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
            if (remaining < 1) return ErrInternal;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
            if (bitCount > 16) {
                out[outP] = (uint8_t)bitStream;
                out[outP + 1] = (uint8_t)(bitStream >> 8);
                outP += 2;
                bitStream >>= 16;
                bitCount -= 16;
            }
        }
        out[outP] = (uint8_t)bitStream;
        out[outP + 1] = (uint8_t)(bitStream >> 8);
        outP += (bitCount + 7) / 8;
        if (charnum > symbolLen) return ErrInternal;
        out.resize(outP);
        Out = out;
        return OK;
    }
    // fse/compress.go:349 buildCTable
    Err buildCTable() {
        uint32_t tableSize = 1u << actualTableLog;
        uint32_t highThreshold = tableSize - 1;
        int16_t cumul[256 + 2] = {0};
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
            if ((uint32_t)cumul[symbolLen] != tableSize) return ErrInternal;
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
                for (int k = 0; k < (int)v; k++) {  // Go: `for range v` runs zero times for v<=0
                    tableSymbol[position] = symbol;
                    position = (position + step) & tableMask;
                    while (position > highThreshold) position = (position + step) & tableMask;
                }
            }
            if (position != 0) return ErrInternal;
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
                    symbolTT[i].deltaFindState = (int32_t)(int16_t)(total - 1);
                    total++;
                    break;
                default: {
                    uint32_t maxBitsOut = (uint32_t)tableLog - highBit((uint32_t)(int32_t)(int16_t)(v - 1));
                    uint32_t minStatePlus = (uint32_t)(int32_t)v << maxBitsOut;
                    symbolTT[i].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                    symbolTT[i].deltaFindState = (int32_t)(int16_t)(total - v);
                    total = (int16_t)(total + v);
                }
                }
            }
            if (total != (int16_t)tableSize) return ErrInternal;
        }
        return OK;
    }
    // fse/compress.go:121 compress
    Err compress(const uint8_t* src, size_t n) {
        if (n <= 2) return ErrInternal;
        const SymbolTransform* tt = symbolTT;
        bw.reset(&Out);
        CState c1, c2;
        size_t ip = n;
        if (ip & 1) {
            c1.init(&bw, stateTable, tt[src[ip - 1]]);
            c2.init(&bw, stateTable, tt[src[ip - 2]]);
            c1.encodeZero(tt[src[ip - 3]]);
            ip -= 3;
        } else {
            c2.init(&bw, stateTable, tt[src[ip - 1]]);
            c1.init(&bw, stateTable, tt[src[ip - 2]]);
            ip -= 2;
        }
        if (ip & 2) {
            c2.encodeZero(tt[src[ip - 1]]);
            c1.encodeZero(tt[src[ip - 2]]);
            ip -= 2;
        }
        size_t len = ip;
        // The four loop variants (:153-196) differ only in flush cadence / zero-bit handling and
        // produce the same bits; restated with the most defensive one (flush32 between pairs,
        // encodeZero).
        for (; len >= 4; len -= 4) {
            bw.flush32();
            uint8_t v3 = src[len - 4], v2 = src[len - 3], v1 = src[len - 2], v0 = src[len - 1];
            c2.encodeZero(tt[v0]);
            c1.encodeZero(tt[v1]);
            bw.flush32();
            c2.encodeZero(tt[v2]);
            c1.encodeZero(tt[v3]);
        }
        c2.flush(actualTableLog);
        c1.flush(actualTableLog);
        bw.close();
        return OK;
    }
};

// fse/compress.go:18 Compress.  Output left in s.Out on OK.
static inline Err Compress(const uint8_t* in, size_t n, Scratch* s) {
    if (n <= 1) return ErrIncompressible;
    s->prepare(in, n);
    s->Out.clear();
    int maxCount = s->maxCount;
    if (maxCount == 0) maxCount = s->countSimple(in, n);
    s->clearCount = true;
    s->maxCount = 0;
    if (maxCount == (int)n) return ErrUseRLE;
    if (maxCount == 1 || maxCount < (int)(n >> 7)) return ErrIncompressible;
    s->optimalTableLog();
    Err e = s->normalizeCount();
    if (e != OK) return e;
    e = s->writeCount();
    if (e != OK) return e;
    e = s->buildCTable();
    if (e != OK) return e;
    e = s->compress(in, n);
    if (e != OK) return e;
    if (s->Out.size() >= n) return ErrIncompressible;
    return OK;
}

}  // namespace fseb
}  // namespace kco
