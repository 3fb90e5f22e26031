// oracle/kco_huff0.h — TEST INFRASTRUCTURE ONLY (CPU oracle; see kco_common.h).
// Restates the compress side of huff0: huff0/compress.go:14-742, huff0/huff0.go:125-337,
// huff0/bitwriter.go.  State that persists across calls (count clearing, prevTable,
// nodes scratch, Reuse policy) is kept exactly as in the reference's Scratch.
#pragma once
#include "kco_common.h"
#include "kco_fse_bytes.h"

namespace kco {
namespace huff0 {

constexpr int maxSymbolValue = 255, tableLogMax = 11, tableLogDefault = 11, minTablelog = 5, huffNodesLen = 512;
constexpr int BlockSizeMax = (1 << 18) - 1;
constexpr int huffNodesMask = huffNodesLen - 1;

enum Err { OK = 0, ErrIncompressible = 1, ErrUseRLE = 2, ErrTooBig = 3, ErrInternal = 4 };
enum ReusePolicy { ReusePolicyAllow = 0, ReusePolicyPrefer = 1, ReusePolicyNone = 2, ReusePolicyMust = 3 };

struct CTableEntry { uint16_t val; uint8_t nBits; };

// A Go slice of cTableEntry with capacity 256: len is tracked, storage persists.
struct CTable {
    CTableEntry e[256];
    int len = 0;
    CTable() { memset(e, 0, sizeof(e)); }
    // huff0/huff0.go:308 estimateSize
    int estimateSize(const uint32_t* hist, int histLen) const {
        uint32_t nbBits = 7;
        for (int i = 0; i < histLen; i++) nbBits += (uint32_t)e[i].nBits * hist[i];
        return (int)(nbBits >> 3);
    }
};

// nodeElt (huff0/compress.go:718-742): count u32 | parent u16<<32 | symbol u8<<48 | nbBits u8<<56
typedef uint64_t nodeElt;
static inline nodeElt makeNodeElt(uint32_t count, uint8_t symbol) { return (nodeElt)count | (nodeElt)symbol << 48; }
static inline uint32_t nCount(nodeElt e) { return (uint32_t)e; }
static inline uint16_t nParent(nodeElt e) { return (uint16_t)(e >> 32); }
static inline uint8_t nSymbol(nodeElt e) { return (uint8_t)(e >> 48); }
static inline uint8_t nNbBits(nodeElt e) { return (uint8_t)(e >> 56); }
static inline void setCount(nodeElt& e, uint32_t c) { e = (e & 0xffffffff00000000ULL) | (nodeElt)c; }
static inline void setParent(nodeElt& e, int16_t p) { e = (e & 0xffff0000ffffffffULL) | (nodeElt)(uint16_t)p << 32; }
static inline void setNbBits(nodeElt& e, uint8_t n) { e = (e & 0x00ffffffffffffffULL) | (nodeElt)n << 56; }

struct Scratch {  // huff0/huff0.go:64
    uint32_t count[256] = {0};
    Bytes Out;
    size_t OutTableLen = 0;  // len(OutTable): table bytes at the front of Out (0 = nil)
    int srcLen = 0;
    uint8_t MaxSymbolValue = 0;
    uint8_t TableLog = 0;
    ReusePolicy Reuse = ReusePolicyAllow;
    uint8_t WantLogLess = 0;
    uint16_t symbolLen = 0;
    int maxCount = 0;
    bool clearCount = false;
    uint8_t actualTableLog = 0;
    uint8_t prevTableLog = 0;
    CTable prevTable;
    CTable cTable;
    nodeElt nodes[huffNodesLen + 1];
    fseb::Scratch fse;
    uint8_t huffWeight[256];

    Scratch() { memset(nodes, 0, sizeof(nodes)); memset(huffWeight, 0, sizeof(huffWeight)); }

    // huff0/huff0.go:125 TransferCTable
    void TransferCTable(const Scratch* src) {
        prevTable.len = src->prevTable.len;
        memcpy(prevTable.e, src->prevTable.e, sizeof(CTableEntry) * (size_t)src->prevTable.len);
        prevTableLog = src->prevTableLog;
    }

    // huff0/huff0.go:134 prepare
    Err prepare(size_t n) {
        if (n > (size_t)BlockSizeMax) return ErrTooBig;
        if (MaxSymbolValue == 0) MaxSymbolValue = maxSymbolValue;
        if (TableLog == 0) TableLog = tableLogDefault;
        if (TableLog > tableLogMax || TableLog < minTablelog) return ErrInternal;
        if (clearCount && maxCount == 0) {
            memset(count, 0, sizeof(count));
            clearCount = false;
        }
        Out.clear();
        OutTableLen = 0;
        srcLen = (int)n;
        return OK;
    }

    // huff0/compress.go:351 countSimple
    int countSimple(const uint8_t* in, size_t n, bool* reuse) {
        *reuse = true;
        for (size_t i = 0; i < n; i++) count[in[i]]++;
        uint32_t m = 0;
        if (prevTable.len > 0) {
            for (int i = 0; i < 256; i++) {
                uint32_t v = count[i];
                if (v == 0) continue;
                if (v > m) m = v;
                symbolLen = (uint16_t)i + 1;
                if (i >= prevTable.len) *reuse = false;
                else if (prevTable.e[i].nBits == 0) *reuse = false;
            }
            return (int)m;
        }
        for (int i = 0; i < 256; i++) {
            uint32_t v = count[i];
            if (v == 0) continue;
            if (v > m) m = v;
            symbolLen = (uint16_t)i + 1;
        }
        *reuse = false;
        return (int)m;
    }
    // huff0/compress.go:387 canUseTable
    bool canUseTable(const CTable& c) const {
        if (c.len < (int)symbolLen) return false;
        for (int i = 0; i < (int)symbolLen; i++)
            if (count[i] != 0 && c.e[i].nBits == 0) return false;
        return true;
    }
    // huff0/compress.go:418 minTableLog
    uint8_t minTableLog() const {
        uint32_t minBitsSrc = highBit((uint32_t)srcLen) + 1;
        uint32_t minBitsSymbols = highBit((uint32_t)(uint16_t)(symbolLen - 1)) + 2;
        if (minBitsSrc < minBitsSymbols) return (uint8_t)minBitsSrc;
        return (uint8_t)minBitsSymbols;
    }
    // huff0/compress.go:428 optimalTableLog
    void optimalTableLog() {
        uint8_t tableLog = TableLog;
        uint8_t minBits = minTableLog();
        uint8_t maxBitsSrc = (uint8_t)((uint8_t)highBit((uint32_t)(srcLen - 1)) - 1);
        if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
        if (minBits > tableLog) tableLog = minBits;
        if (tableLog < minTablelog) tableLog = minTablelog;
        if (tableLog > tableLogMax) tableLog = tableLogMax;
        actualTableLog = tableLog;
    }

    // huff0/compress.go:570 huffSort
    void huffSort() {
        struct rankPos { uint32_t base, current; };
        nodeElt* nd = nodes + 1;  // nodes[1 : huffNodesLen+1]
        rankPos rank[32];
        memset(rank, 0, sizeof(rank));
        for (int i = 0; i < (int)symbolLen; i++) {
            uint32_t r = highBit(count[i] + 1) & 31;
            rank[r].base++;
        }
        const int maxBitLength = 18 + 1;
        for (int n = maxBitLength; n > 0; n--) rank[n - 1].base += rank[n].base;
        for (int n = 0; n < maxBitLength; n++) rank[n].current = rank[n].base;
        for (int n = 0; n < (int)symbolLen; n++) {
            uint32_t c = count[n];
            uint32_t r = (highBit(c + 1) + 1) & 31;
            uint32_t pos = rank[r].current;
            rank[r].current++;
            nodeElt prev = nd[(pos - 1) & huffNodesMask];
            while (pos > rank[r].base && c > nCount(prev)) {
                nd[pos & huffNodesMask] = prev;
                pos--;
                prev = nd[(pos - 1) & huffNodesMask];
            }
            nd[pos & huffNodesMask] = makeNodeElt(c, (uint8_t)n);
        }
    }

    // huff0/compress.go:609 setMaxHeight
    uint8_t setMaxHeight(int lastNonNull) {
        uint8_t maxNbBits = actualTableLog;
        nodeElt* huffNode = nodes + 1;
        uint8_t largestBits = nNbBits(huffNode[lastNonNull]);
        if (largestBits <= maxNbBits) return largestBits;
        int totalCost = 0;
        int baseCost = 1 << (largestBits - maxNbBits);
        uint32_t n = (uint32_t)lastNonNull;
        while (nNbBits(huffNode[n]) > maxNbBits) {
            totalCost += baseCost - (1 << (largestBits - nNbBits(huffNode[n])));
            setNbBits(huffNode[n], maxNbBits);
            n--;
        }
        while (nNbBits(huffNode[n]) == maxNbBits) n--;
        totalCost >>= (largestBits - maxNbBits);
        {
            const uint32_t noSymbol = 0xF0F0F0F0;
            uint32_t rankLast[tableLogMax + 2];
            for (int i = 0; i < tableLogMax + 2; i++) rankLast[i] = noSymbol;
            {
                uint8_t currentNbBits = maxNbBits;
                for (int pos = (int)n; pos >= 0; pos--) {
                    if (nNbBits(huffNode[pos]) >= currentNbBits) continue;
                    currentNbBits = nNbBits(huffNode[pos]);
                    rankLast[maxNbBits - currentNbBits] = (uint32_t)pos;
                }
            }
            while (totalCost > 0) {
                uint8_t nBitsToDecrease = (uint8_t)((uint8_t)highBit((uint32_t)totalCost) + 1);
                for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                    uint32_t highPos = rankLast[nBitsToDecrease];
                    uint32_t lowPos = rankLast[nBitsToDecrease - 1];
                    if (highPos == noSymbol) continue;
                    if (lowPos == noSymbol) break;
                    uint32_t highTotal = nCount(huffNode[highPos]);
                    uint32_t lowTotal = 2 * nCount(huffNode[lowPos]);
                    if (highTotal <= lowTotal) break;
                }
                while (nBitsToDecrease <= tableLogMax && rankLast[nBitsToDecrease] == noSymbol) nBitsToDecrease++;
                totalCost -= 1 << (nBitsToDecrease - 1);
                if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                setNbBits(huffNode[rankLast[nBitsToDecrease]], (uint8_t)(1 + nNbBits(huffNode[rankLast[nBitsToDecrease]])));
                if (rankLast[nBitsToDecrease] == 0) {
                    rankLast[nBitsToDecrease] = noSymbol;
                } else {
                    rankLast[nBitsToDecrease]--;
                    if (nNbBits(huffNode[rankLast[nBitsToDecrease]]) != (uint8_t)(maxNbBits - nBitsToDecrease))
                        rankLast[nBitsToDecrease] = noSymbol;
                }
            }
            while (totalCost < 0) {
                if (rankLast[1] == noSymbol) {
                    while (nNbBits(huffNode[n]) == maxNbBits) n--;
                    setNbBits(huffNode[n + 1], (uint8_t)(nNbBits(huffNode[n + 1]) - 1));
                    rankLast[1] = n + 1;
                    totalCost++;
                    continue;
                }
                setNbBits(huffNode[rankLast[1] + 1], (uint8_t)(nNbBits(huffNode[rankLast[1] + 1]) - 1));
                rankLast[1]++;
                totalCost++;
            }
        }
        return maxNbBits;
    }

    // huff0/compress.go:457 buildCTable
    Err buildCTable() {
        optimalTableLog();
        huffSort();
        cTable.len = symbolLen;
        for (int i = 0; i < cTable.len; i++) { cTable.e[i].val = 0; cTable.e[i].nBits = 0; }

        int16_t startNode = (int16_t)symbolLen;
        uint16_t nonNullRank = (uint16_t)(symbolLen - 1);
        int16_t nodeNb = startNode;
        nodeElt* huffNode = nodes + 1;  // huffNode[i] == nodes[i+1]
        nodeElt* huffNode0 = nodes;     // huffNode0[i+1] == huffNode[i]

        while (nCount(huffNode[nonNullRank]) == 0) nonNullRank--;

        int16_t lowS = (int16_t)nonNullRank;
        int16_t nodeRoot = (int16_t)(nodeNb + lowS - 1);
        int16_t lowN = nodeNb;
        setCount(huffNode[nodeNb], nCount(huffNode[lowS]) + nCount(huffNode[lowS - 1]));
        setParent(huffNode[lowS], nodeNb);
        setParent(huffNode[lowS - 1], nodeNb);
        nodeNb++;
        lowS -= 2;
        for (int16_t n = nodeNb; n <= nodeRoot; n++) setCount(huffNode[n], 1u << 30);
        setCount(huffNode0[0], 1u << 31);

        while (nodeNb <= nodeRoot) {
            int16_t n1, n2;
            if (nCount(huffNode0[lowS + 1]) < nCount(huffNode0[lowN + 1])) { n1 = lowS; lowS--; }
            else { n1 = lowN; lowN++; }
            if (nCount(huffNode0[lowS + 1]) < nCount(huffNode0[lowN + 1])) { n2 = lowS; lowS--; }
            else { n2 = lowN; lowN++; }
            setCount(huffNode[nodeNb], nCount(huffNode0[n1 + 1]) + nCount(huffNode0[n2 + 1]));
            setParent(huffNode0[n1 + 1], nodeNb);
            setParent(huffNode0[n2 + 1], nodeNb);
            nodeNb++;
        }

        setNbBits(huffNode[nodeRoot], 0);
        for (int16_t n = (int16_t)(nodeRoot - 1); n >= startNode; n--)
            setNbBits(huffNode[n], (uint8_t)(nNbBits(huffNode[nParent(huffNode[n])]) + 1));
        for (uint16_t n = 0; n <= nonNullRank; n++)
            setNbBits(huffNode[n], (uint8_t)(nNbBits(huffNode[nParent(huffNode[n])]) + 1));
        actualTableLog = setMaxHeight((int)nonNullRank);
        uint8_t maxNbBits = actualTableLog;
        if (maxNbBits > tableLogMax) return ErrInternal;

        uint16_t nbPerRank[tableLogMax + 1];
        uint16_t valPerRank[16];
        memset(nbPerRank, 0, sizeof(nbPerRank));
        memset(valPerRank, 0, sizeof(valPerRank));
        for (int i = 0; i <= (int)nonNullRank; i++) nbPerRank[nNbBits(huffNode[i])]++;
        {
            uint16_t min = 0;
            for (uint8_t n = maxNbBits; n > 0; n--) {
                valPerRank[n] = min;
                min = (uint16_t)(min + nbPerRank[n]);
                min >>= 1;
            }
        }
        for (int i = 0; i <= (int)nonNullRank; i++) cTable.e[nSymbol(huffNode[i])].nBits = nNbBits(huffNode[i]);
        for (int n = 0; n < (int)symbolLen; n++) {
            uint8_t nbits = cTable.e[n].nBits & 15;
            uint16_t v = valPerRank[nbits];
            cTable.e[n].val = v;
            valPerRank[nbits] = (uint16_t)(v + 1);
        }
        return OK;
    }

    // huff0/huff0.go:180 cTable.write — appends the table description to Out.
    Err writeTable(const CTable& c) {
        uint8_t bitsToWeight[tableLogMax + 1];
        uint8_t huffLog = actualTableLog;
        uint8_t maxSym = (uint8_t)(symbolLen - 1);
        const int maxFSETableLog = 6;
        memset(bitsToWeight, 0, sizeof(bitsToWeight));
        for (uint8_t n = 1; n < huffLog + 1; n++) bitsToWeight[n] = (uint8_t)(huffLog + 1 - n);
        uint32_t* hist = fse.count;
        for (int i = 0; i < 16; i++) hist[i] = 0;
        for (int n = 0; n < (int)maxSym; n++) {
            uint8_t v = bitsToWeight[c.e[n].nBits] & 15;
            huffWeight[n] = v;
            hist[v]++;
        }
        if (maxSym >= 2) {
            uint32_t huffMaxCnt = 0;
            uint8_t huffMax = 0;
            for (int i = 0; i < 16; i++) {
                uint32_t v = hist[i];
                if (v == 0) continue;
                huffMax = (uint8_t)i;
                if (v > huffMaxCnt) huffMaxCnt = v;
            }
            fse.HistogramFinished(huffMax, (int)huffMaxCnt);
            fse.TableLog = maxFSETableLog;
            fseb::Err err = fseb::Compress(huffWeight, maxSym, &fse);
            if (err == fseb::OK && (int)fse.Out.size() < (int)(symbolLen >> 1)) {
                Out.push_back((uint8_t)fse.Out.size());
                Out.insert(Out.end(), fse.Out.begin(), fse.Out.end());
                return OK;
            }
        }
        if (maxSym > (256 - 128)) return ErrIncompressible;
        Out.push_back((uint8_t)(128 | (maxSym - 1)));
        huffWeight[maxSym] = 0;
        for (uint16_t n = 0; n < (uint16_t)maxSym; n += 2)
            Out.push_back((uint8_t)((huffWeight[n] << 4) | huffWeight[n + 1]));
        return OK;
    }

    // huff0/compress.go:233 compress1xDo — appends to dst.
    void compress1xDo(Bytes* dst, const uint8_t* src, size_t len) {
        BitWriter bw;
        bw.reset(dst);
        const CTableEntry* ct = cTable.e;
        int64_t n = (int64_t)len;
        n -= n & 3;
        for (int64_t i = (int64_t)(len & 3); i > 0; i--) {
            CTableEntry enc = ct[src[n + i - 1]];
            bw.addBits32Clean(enc.val, enc.nBits);  // bitWriter.encSymbol
        }
        n -= 4;
        for (; n >= 0; n -= 4) {
            const uint8_t* tmp = src + n;
            bw.flush32();
            bw.addBits32Clean(ct[tmp[3]].val, ct[tmp[3]].nBits);
            bw.addBits32Clean(ct[tmp[2]].val, ct[tmp[2]].nBits);
            bw.flush32();
            bw.addBits32Clean(ct[tmp[1]].val, ct[tmp[1]].nBits);
            bw.addBits32Clean(ct[tmp[0]].val, ct[tmp[0]].nBits);
        }
        bw.close();
    }
    // huff0/compress.go:229 compress1X
    Err compress1X(const uint8_t* src, size_t len) { compress1xDo(&Out, src, len); return OK; }
    // huff0/compress.go:269 compress4X
    Err compress4X(const uint8_t* src, size_t len) {
        if (len < 12) return ErrIncompressible;
        size_t segmentSize = (len + 3) / 4;
        size_t offsetIdx = Out.size();
        for (int i = 0; i < 6; i++) Out.push_back(0);
        for (int i = 0; i < 4; i++) {
            size_t toDo = len;
            if (toDo > segmentSize) toDo = segmentSize;
            size_t idx = Out.size();
            compress1xDo(&Out, src, toDo);
            src += toDo;
            len -= toDo;
            if (Out.size() - idx > 65535) return ErrIncompressible;
            if (i < 3) {
                size_t length = Out.size() - idx;
                Out[i * 2 + offsetIdx] = (uint8_t)length;
                Out[i * 2 + offsetIdx + 1] = (uint8_t)(length >> 8);
            }
        }
        return OK;
    }
};

// huff0/compress.go:43 compress.  `four` selects compress4X vs compress1X.
// On OK, s->Out holds (table +) data; *reUsed tells whether prevTable was used.
static inline Err compress(const uint8_t* in, size_t n, Scratch* s, bool four, bool* reUsed) {
    *reUsed = false;
    Err perr = s->prepare(n);
    if (perr != OK) return perr;
    auto compressor = [&](void) -> Err { return four ? s->compress4X(in, n) : s->compress1X(in, n); };

    if (s->Reuse == ReusePolicyNone) s->prevTable.len = 0;
    int maxCount = s->maxCount;
    bool canReuse = false;
    if (maxCount == 0) maxCount = s->countSimple(in, n, &canReuse);
    else canReuse = s->canUseTable(s->prevTable);

    int wantSize = (int)n;
    if (s->WantLogLess > 0) wantSize -= wantSize >> s->WantLogLess;

    s->clearCount = true;
    s->maxCount = 0;
    if (maxCount >= (int)n) {
        if (maxCount > (int)n) return ErrInternal;
        if (n == 1) return ErrIncompressible;
        return ErrUseRLE;
    }
    if (maxCount == 1 || maxCount < (int)(n >> 7)) return ErrIncompressible;
    if (s->Reuse == ReusePolicyMust && !canReuse) return ErrIncompressible;
    if ((s->Reuse == ReusePolicyPrefer || s->Reuse == ReusePolicyMust) && canReuse) {
        CTable keepTable = s->cTable;
        uint8_t keepTL = s->actualTableLog;
        s->cTable = s->prevTable;
        s->actualTableLog = s->prevTableLog;
        Err err = compressor();
        s->cTable = keepTable;
        s->actualTableLog = keepTL;
        if (err == OK && (int)s->Out.size() < wantSize) { *reUsed = true; return OK; }
        if (s->Reuse == ReusePolicyMust) return ErrIncompressible;
        s->prevTable.len = 0;
    }
    Err err = s->buildCTable();
    if (err != OK) return err;

    if (s->Reuse == ReusePolicyAllow && canReuse) {
        int hSize = (int)s->Out.size();
        int oldSize = s->prevTable.estimateSize(s->count, s->symbolLen);
        int newSize = s->cTable.estimateSize(s->count, s->symbolLen);
        if (oldSize <= hSize + newSize || hSize + 12 >= wantSize) {
            CTable keepTable = s->cTable;
            uint8_t keepTL = s->actualTableLog;
            s->cTable = s->prevTable;
            s->actualTableLog = s->prevTableLog;
            err = compressor();
            s->cTable = keepTable;
            s->actualTableLog = keepTL;
            if (err != OK) return err;
            if ((int)s->Out.size() >= wantSize) return ErrIncompressible;
            *reUsed = true;
            return OK;
        }
    }
    err = s->writeTable(s->cTable);
    if (err != OK) { s->OutTableLen = 0; return err; }
    s->OutTableLen = s->Out.size();
    err = compressor();
    if (err != OK) { s->OutTableLen = 0; return err; }
    if ((int)s->Out.size() >= wantSize) { s->OutTableLen = 0; return ErrIncompressible; }
    // Move current table into previous.
    s->prevTable = s->cTable;
    s->prevTableLog = s->actualTableLog;
    s->cTable.len = 0;
    return OK;
}

}  // namespace huff0
}  // namespace kco
