// TEST INFRASTRUCTURE ONLY (see kco_common.h).  CPU restatement of the reference's dictionary loader:
//   zstd/dict.go:71-150        loadDict
//   huff0/decompress.go:29-168 ReadTable  (weights -> cTable "prevTable" used by the encoder)
//   fse/decompress.go:19-330   Decompress (readNCount / buildDtable / decompress) for FSE-compressed weights
//   zstd/fse_decoder.go:52-184 fseDecoder.readNCount (the three sequence tables: parsed for their length + validity)
// Only what the ENCODER consumes is kept (id, litEnc.prevTable/prevTableLog, offsets, content).
#pragma once
#include "kco_common.h"
#include "kco_huff0.h"
#include "kco_zstd_fast.h"

namespace kco {
namespace dictload {

struct ByteReader {  // zstd/bytereader.go / fse/bytereader.go
    const uint8_t* b;
    int len;
    int off = 0;
    int remain() const { return len - off; }
    void advance(unsigned n) { off += (int)n; }
    uint32_t Uint32() const {  // zero-extends past the end like the reference's bounds-checked variant
        uint32_t v = 0;
        for (int k = 0; k < 4; k++)
            if (off + k < len && off + k >= 0) v |= (uint32_t)b[off + k] << (8 * k);
        return v;
    }
};

inline int highBits(uint32_t v) { return 31 - __builtin_clz(v); }

// Shared body of fse.Scratch.readNCount (fse/decompress.go:42-160) and fseDecoder.readNCount
// (zstd/fse_decoder.go:52-184): they differ in the tableLog limit, the symbol limit of the main loop and in the
// trailing checks, passed as parameters.
struct NCount {
    int16_t norm[256];
    uint16_t symbolLen = 0;
    uint8_t actualTableLog = 0;
};
inline bool readNCount(ByteReader* b, NCount* s, unsigned tablelogAbsoluteMax, int maxSymbol, bool zstdVariant) {
    uint16_t charnum = 0;
    bool previous0 = false;
    const int iend = b->remain();
    if (iend < 4) return false;
    const int base = b->off;
    uint32_t bitStream = b->Uint32();
    unsigned nbBits = (bitStream & 0xF) + 5;  // minTablelog
    if (nbBits > tablelogAbsoluteMax) return false;
    bitStream >>= 4;
    unsigned bitCount = 4;
    s->actualTableLog = (uint8_t)nbBits;
    int32_t remaining = (int32_t)((1 << nbBits) + 1);
    int32_t threshold = (int32_t)(1 << nbBits);
    int32_t gotTotal = 0;
    nbBits++;
    auto rem = [&]() { return iend - (b->off - base); };
    while (remaining > 1 && (int)charnum <= maxSymbol) {
        if (previous0) {
            uint16_t n0 = charnum;
            while ((bitStream & 0xFFFF) == 0xFFFF) {
                n0 += 24;
                if (rem() > 5) {
                    b->advance(2);
                    bitStream = b->Uint32() >> bitCount;
                } else {
                    bitStream >>= 16;
                    bitCount += 16;
                }
            }
            while ((bitStream & 3) == 3) {
                n0 += 3;
                bitStream >>= 2;
                bitCount += 2;
            }
            n0 += (uint16_t)(bitStream & 3);
            bitCount += 2;
            if (n0 > 255) return false;
            while (charnum < n0) { s->norm[charnum & 0xff] = 0; charnum++; }
            if (rem() >= 7 || rem() - (int)(bitCount >> 3) >= 4) {
                b->advance(bitCount >> 3);
                bitCount &= 7;
                bitStream = b->Uint32() >> bitCount;
            } else {
                bitStream >>= 2;
            }
        }
        const int32_t max = (2 * threshold - 1) - remaining;
        int32_t count;
        if (((int32_t)bitStream & (threshold - 1)) < max) {
            count = (int32_t)bitStream & (threshold - 1);
            bitCount += nbBits - 1;
        } else {
            count = (int32_t)bitStream & (2 * threshold - 1);
            if (count >= threshold) count -= max;
            bitCount += nbBits;
        }
        count--;
        if (count < 0) { remaining += count; gotTotal -= count; }
        else { remaining -= count; gotTotal += count; }
        s->norm[charnum & 0xff] = (int16_t)count;
        charnum++;
        previous0 = count == 0;
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
        if (rem() >= 7 || rem() - (int)(bitCount >> 3) >= 4) {
            b->advance(bitCount >> 3);
            bitCount &= 7;
        } else {
            bitCount -= (unsigned)(8 * (base + iend - 4 - b->off));
            b->off = base + iend - 4;
        }
        bitStream = b->Uint32() >> (bitCount & 31);
    }
    s->symbolLen = charnum;
    if (s->symbolLen <= 1) return false;
    if (s->symbolLen > 256) return false;
    if (remaining != 1) return false;
    if (bitCount > 32) return false;
    if (gotTotal != (1 << s->actualTableLog)) return false;
    b->advance((bitCount + 7) >> 3);
    (void)zstdVariant;
    return true;
}

struct DecSymbol { uint16_t newState; uint8_t symbol; uint8_t nbBits; };

// fse/decompress.go:186-255 buildDtable (also what zstd's fseDecoder.buildDtable checks: spread must end at 0)
inline bool buildDtable(const NCount& s, std::vector<DecSymbol>* decTable, bool* zeroBits) {
    const uint32_t tableSize = 1u << s.actualTableLog;
    uint32_t highThreshold = tableSize - 1;
    decTable->assign(tableSize, DecSymbol{0, 0, 0});
    uint16_t symbolNext[256];
    *zeroBits = false;
    const int16_t largeLimit = (int16_t)(1 << (s.actualTableLog - 1));
    for (int i = 0; i < s.symbolLen; i++) {
        const int16_t v = s.norm[i];
        if (v == -1) {
            (*decTable)[highThreshold].symbol = (uint8_t)i;
            highThreshold--;
            symbolNext[i] = 1;
        } else {
            if (v >= largeLimit) *zeroBits = true;
            symbolNext[i] = (uint16_t)v;
        }
    }
    const uint32_t tableMask = tableSize - 1;
    const uint32_t step = (tableSize >> 1) + (tableSize >> 3) + 3;
    uint32_t position = 0;
    for (int ss = 0; ss < s.symbolLen; ss++) {
        for (int i = 0; i < (int)s.norm[ss]; i++) {
            (*decTable)[position].symbol = (uint8_t)ss;
            position = (position + step) & tableMask;
            while (position > highThreshold) position = (position + step) & tableMask;
        }
    }
    if (position != 0) return false;
    for (uint32_t u = 0; u < tableSize; u++) {
        const uint8_t symbol = (*decTable)[u].symbol;
        const uint16_t nextState = symbolNext[symbol];
        symbolNext[symbol] = (uint16_t)(nextState + 1);
        if (nextState == 0) return false;
        const uint8_t nBits = (uint8_t)(s.actualTableLog - (uint8_t)highBits(nextState));
        (*decTable)[u].nbBits = nBits;
        const uint16_t newState = (uint16_t)((nextState << nBits) - tableSize);
        if (newState >= tableSize) return false;
        if (newState == (uint16_t)u && nBits == 0) return false;
        (*decTable)[u].newState = newState;
    }
    return true;
}

// fse/bitreader.go: reverse bit reader (init skips the end mark; getBits returns zero-extended bits)
struct BitReader {
    const uint8_t* in;
    int off;  // bytes not yet loaded
    uint64_t value = 0;
    uint8_t bitsRead = 64;
    bool init(const uint8_t* p, int n) {
        if (n < 1) return false;
        in = p; off = n;
        const uint8_t v = p[n - 1];
        if (v == 0) return false;
        bitsRead = 64; value = 0;
        if (n >= 8) fillFast2();
        else fill();
        fill();
        bitsRead += (uint8_t)(8 - highBits(v));
        return true;
    }
    void fillFast2() { fillFastOnce(); fillFastOnce(); }
    void fillFastOnce() {
        if (bitsRead < 32) return;
        uint32_t low = (uint32_t)in[off - 4] | ((uint32_t)in[off - 3] << 8) | ((uint32_t)in[off - 2] << 16) | ((uint32_t)in[off - 1] << 24);
        value = (value << 32) | low;
        bitsRead -= 32;
        off -= 4;
    }
    void fill() {
        if (bitsRead < 32) return;
        if (off > 4) { fillFastOnce(); return; }
        while (off > 0) {
            value = (value << 8) | in[off - 1];
            bitsRead -= 8;
            off--;
        }
    }
    uint16_t getBits(uint8_t n) {
        if (n == 0 || bitsRead >= 64) return 0;
        const uint16_t v = (uint16_t)((value << (bitsRead & 63)) >> ((64 - n) & 63));
        bitsRead += n;
        return v;
    }
    bool finished() const { return bitsRead >= 64 && off == 0; }
    bool overread() const { return bitsRead > 64; }
};

// fse.Decompress for Huffman weights (huff0/decompress.go:57-70), DecompressLimit 255
inline bool fseDecompress(const uint8_t* in, int n, Bytes* out) {
    ByteReader br{in, n};
    NCount nc;
    if (!readNCount(&br, &nc, 15, 255, false)) return false;
    std::vector<DecSymbol> dt;
    bool zeroBits;
    if (!buildDtable(nc, &dt, &zeroBits)) return false;
    BitReader bits;
    if (!bits.init(in + br.off, n - br.off)) return false;
    uint16_t s1 = bits.getBits(nc.actualTableLog), s2 = bits.getBits(nc.actualTableLog);
    out->clear();
    auto next = [&](uint16_t& st) -> uint8_t {
        const DecSymbol& d = dt[st];
        const uint16_t low = bits.getBits(d.nbBits);
        st = (uint16_t)(d.newState + low);
        return d.symbol;
    };
    auto fin = [&](uint16_t st) { return bits.finished() && dt[st].nbBits > 0; };
    for (;;) {  // fse/decompress.go:310-326 (the unrolled main loops decode the same symbol sequence)
        if (fin(s1)) { out->push_back(dt[s1].symbol); out->push_back(dt[s2].symbol); break; }
        bits.fill();
        out->push_back(next(s1));
        if (fin(s2)) { out->push_back(dt[s2].symbol); out->push_back(dt[s1].symbol); break; }
        out->push_back(next(s2));
        if (out->size() >= 255) return false;
    }
    return !bits.overread();
}

// huff0.ReadTable (huff0/decompress.go:29-168), encoder-relevant result only
inline bool ReadTable(const uint8_t* in, int n, huff0::Scratch* s, int* consumed) {
    if (n <= 1) return false;
    int iSize = in[0];
    const uint8_t* p = in + 1;
    int left = n - 1;
    int symbolLen = 0;
    memset(s->huffWeight, 0, sizeof(s->huffWeight));
    if (iSize >= 128) {
        const int oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize > left) return false;
        for (int k = 0; k < oSize; k += 2) {
            const uint8_t v = p[k / 2];
            s->huffWeight[k] = v >> 4;
            if (k + 1 < 256) s->huffWeight[k + 1] = v & 15;
        }
        symbolLen = oSize;
        p += iSize; left -= iSize;
    } else {
        if (left < iSize) return false;
        Bytes w;
        if (!fseDecompress(p, iSize, &w)) return false;
        if (w.size() > 255) return false;
        memcpy(s->huffWeight, w.data(), w.size());
        symbolLen = (int)w.size();
        p += iSize; left -= iSize;
    }
    uint32_t rankStats[16] = {0};
    uint32_t weightTotal = 0;
    for (int k = 0; k < symbolLen; k++) {
        const uint8_t v = s->huffWeight[k];
        if (v > 11) return false;  // tableLogMax
        rankStats[v & 15]++;
        weightTotal += (1u << (v & 15)) >> 1;
    }
    if (weightTotal == 0) return false;
    {
        const uint32_t tableLog = (uint32_t)highBits(weightTotal) + 1;
        if (tableLog > 11) return false;
        s->actualTableLog = (uint8_t)tableLog;
        const uint32_t total = 1u << tableLog;
        const uint32_t rest = total - weightTotal;
        const uint32_t verif = 1u << highBits(rest);
        const uint32_t lastWeight = (uint32_t)highBits(rest) + 1;
        if (verif != rest) return false;
        s->huffWeight[symbolLen] = (uint8_t)lastWeight;
        symbolLen++;
        rankStats[lastWeight]++;
    }
    if (rankStats[1] < 2 || (rankStats[1] & 1) != 0) return false;
    {
        uint32_t nextRankStart = 0;
        for (uint8_t r = 1; r < s->actualTableLog + 1; r++) {
            const uint32_t current = nextRankStart;
            nextRankStart += rankStats[r] << (r - 1);
            rankStats[r] = current;
        }
    }
    s->symbolLen = (uint16_t)symbolLen;
    s->prevTable.len = symbolLen;
    s->prevTableLog = s->actualTableLog;
    for (int k = 0; k < symbolLen; k++) {
        const uint8_t w = s->huffWeight[k];
        if (w == 0) { s->prevTable.e[k] = huff0::CTableEntry{0, 0}; continue; }
        const uint32_t length = (1u << w) >> 1;
        const uint8_t nBits = (uint8_t)(s->actualTableLog + 1 - w);
        s->prevTable.e[k] = huff0::CTableEntry{(uint16_t)(rankStats[w] >> (w - 1)), nBits};
        rankStats[w] += length;
    }
    *consumed = (int)(p - in);
    return true;
}

// zstd/dict.go:71 loadDict.  litEncStore receives the literal encoder; d->litEnc points at it.
inline bool loadDict(const uint8_t* b, size_t len, DictO* d, huff0::Scratch* litEncStore) {
    if (len <= 8 + 3 * 4) return false;
    if (!(b[0] == 0x37 && b[1] == 0xA4 && b[2] == 0x30 && b[3] == 0xEC)) return false;  // dictMagic "\x37\xa4\x30\xec"
    d->id = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
    if (d->id == 0) return false;
    int used = 0;
    if (!ReadTable(b + 8, (int)(len - 8), litEncStore, &used)) return false;
    litEncStore->Reuse = huff0::ReusePolicyMust;
    d->litEnc = litEncStore;
    ByteReader br{b + 8 + used, (int)(len - 8) - used};
    const int maxTableSymbol[3] = {30, 52, 35};  // maxOffsetLengthSymbol = 30 (zstd/fse_predefined.go:45)  // tableOffsets, tableMatchLengths, tableLiteralLengths order of dict.go:120-128
    for (int t = 0; t < 3; t++) {
        NCount nc;
        if (!readNCount(&br, &nc, 9, maxTableSymbol[t], true)) return false;
        if (br.off > br.len) return false;  // br.overread()
        std::vector<DecSymbol> dt;
        bool zb;
        if (!buildDtable(nc, &dt, &zb)) return false;
        // dec.transform(symbolTableX[i]) (fse_decoder.go:274-290): every symbol that owns a decoding cell must be in the alphabet.
        // readNCount itself lets a zero run carry the symbol count past maxSymbol (fse_decoder.go:105-108).
        for (int i = maxTableSymbol[t] + 1; i < nc.symbolLen; i++)
            if (nc.norm[i] != 0) return false;
    }
    if (br.remain() < 12) return false;
    for (int k = 0; k < 3; k++) {
        d->offsets[k] = (int)(int32_t)br.Uint32();
        br.advance(4);
        if (d->offsets[k] <= 0) return false;
    }
    d->content.assign(br.b + br.off, br.b + br.len);
    for (int k = 0; k < 3; k++)
        if (d->offsets[k] > (int)d->content.size()) return false;
    return true;
}

}  // namespace dictload
}  // namespace kco
