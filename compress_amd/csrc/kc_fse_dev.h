// kc_fse_dev.h — device-side FSE table construction for the zstd sequence coders and for
// Huffman-weight compression.  Serial per table (tables have <= 256 states): one lane per
// table, three tables (LL/OF/ML) in flight on three waves.
//
// Follows the arithmetic of zstd/fse_encoder.go (normalizeCount :259, normalizeCount2 :334,
// optimalTableLog :429, buildCTable :102, writeCount :488, approxSize/bitCost :603-660,
// setRLE :208, cState.init :682) and fse/compress.go for the byte coder, including the Go
// integer-width wrap-arounds they rely on.
#pragma once
#include "kc_dev.h"
#include "kc_wave.h"

struct KcFseT {  // one fseEncoder (zstd/fse_encoder.go:23) — alphabet <= 64 symbols, tableLog <= 8
    uint32_t dnb[64];    // symbolTT[].deltaNbBits
    int16_t dfs[64];     // symbolTT[].deltaFindState
    uint16_t st[256];    // ct.stateTable
    int16_t norm[64];
    uint8_t outBits[64];  // symbolTT[].outBits
    uint16_t symbolLen;
    uint8_t tableLog;  // actualTableLog
    uint8_t useRLE;
    uint8_t rleVal;
    uint8_t reUsed;
    uint8_t preDefined;
    uint8_t stLen1;  // len(ct.stateTable) == 1 (RLE)
};

struct KcFsePredefBlob { KcFseT t[3]; };  // LL, OF, ML predefined encoders (zstd/fse_predefined.go)

__device__ __forceinline__ uint32_t fse_table_step(uint32_t tableSize) { return (tableSize >> 1) + (tableSize >> 3) + 3; }

// zstd/seqenc.go code tables
__device__ __forceinline__ uint32_t kc_ll_code(uint32_t litLength) {
    if (litLength <= 63) {
        if (litLength < 16) return litLength;
        // llCodeTable[16..63]
        if (litLength < 24) return 16 + ((litLength - 16) >> 1);
        if (litLength < 32) return 20 + ((litLength - 24) >> 2);
        if (litLength < 48) return 22 + ((litLength - 32) >> 3);
        return 24;
    }
    return high_bit(litLength) + 19;
}
__device__ __forceinline__ uint32_t kc_ml_code(uint32_t mlBase) {
    if (mlBase <= 127) {
        if (mlBase < 32) return mlBase;
        if (mlBase < 40) return 32 + ((mlBase - 32) >> 1);
        if (mlBase < 48) return 36 + ((mlBase - 40) >> 2);
        if (mlBase < 64) return 38 + ((mlBase - 48) >> 3);
        if (mlBase < 96) return 40 + ((mlBase - 64) >> 4);
        return 42;
    }
    return high_bit(mlBase) + 36;
}
__device__ __forceinline__ uint32_t kc_of_code(uint32_t offset) { return (uint32_t)(bits_len32(offset) - 1); }
__device__ __forceinline__ uint32_t kc_ll_bits(uint32_t code) {
    // llBitsTable (zstd/seqenc.go:62)
    if (code < 16) return 0;
    if (code < 20) return 1;
    if (code < 22) return 2;
    if (code < 24) return 3;
    if (code == 24) return 4;
    return code - 19;  // 25->6 ... 35->16
}
__device__ __forceinline__ uint32_t kc_ml_bits(uint32_t code) {
    // mlBitsTable (zstd/seqenc.go:90)
    if (code < 32) return 0;
    if (code < 36) return 1;
    if (code < 38) return 2;
    if (code < 40) return 3;
    if (code < 42) return 4;
    if (code == 42) return 5;
    return code - 36;  // 43->7 ... 52->16
}

// fseEncoder.optimalTableLog (zstd/fse_encoder.go:429)
__device__ inline uint8_t fse_optimal_table_log(int length, uint16_t symbolLen) {
    uint8_t tableLog = 8;  // maxEncTableLog
    uint32_t minBitsSrc = high_bit((uint32_t)length) + 1;
    uint32_t minBitsSymbols = high_bit((uint32_t)(uint16_t)(symbolLen - 1)) + 2;
    uint8_t minBits = (uint8_t)minBitsSymbols;
    if (minBitsSrc < minBitsSymbols) minBits = (uint8_t)minBitsSrc;
    uint8_t maxBitsSrc = (uint8_t)((uint8_t)high_bit((uint32_t)(length - 1)) - 2);
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 8) tableLog = 8;
    return tableLog;
}

// Shared normalisation core of zstd/fse_encoder.go:259-427 and fse/compress.go:510-632
// (the two copies are arithmetically identical).  count/norm: alphabet arrays.
// Returns false on the reference's internal error ("weight < 1").
__device__ inline bool fse_normalize_core(const uint32_t* count, int16_t* norm, int symbolLen, int length, uint8_t tableLog) {
    const uint32_t rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    const uint64_t scale = 62 - (uint64_t)tableLog;
    const uint64_t step = ((uint64_t)1 << 62) / (uint64_t)length;
    const uint64_t vStep = (uint64_t)1 << (scale - 20);
    int16_t stillToDistribute = (int16_t)(1 << tableLog);
    int largest = 0;
    int16_t largestP = 0;
    const uint32_t lowThreshold = (uint32_t)(length >> tableLog);
    for (int i = 0; i < symbolLen; i++) {
        const uint32_t cnt = count[i];
        if (cnt == 0) { norm[i] = 0; continue; }
        if (cnt <= lowThreshold) {
            norm[i] = -1;
            stillToDistribute--;
        } else {
            int16_t proba = (int16_t)(((uint64_t)cnt * step) >> scale);
            if (proba < 8) {
                const uint64_t restToBeat = vStep * (uint64_t)rtb[proba];
                const uint64_t v = (uint64_t)cnt * step - ((uint64_t)proba << scale);
                if (v > restToBeat) proba++;
            }
            if (proba > largestP) { largestP = proba; largest = i; }
            norm[i] = proba;
            stillToDistribute = (int16_t)(stillToDistribute - proba);
        }
    }
    if ((int16_t)(-stillToDistribute) < (int16_t)(norm[largest] >> 1)) {
        norm[largest] = (int16_t)(norm[largest] + stillToDistribute);
        return true;
    }
    // normalizeCount2 — secondary method
    const int16_t notYetAssigned = -2;
    uint32_t distributed = 0;
    uint32_t total = (uint32_t)length;
    uint32_t lowOne = (total * 3) >> (tableLog + 1);
    for (int i = 0; i < symbolLen; i++) {
        const uint32_t cnt = count[i];
        if (cnt == 0) { norm[i] = 0; continue; }
        if (cnt <= lowThreshold) { norm[i] = -1; distributed++; total -= cnt; continue; }
        if (cnt <= lowOne) { norm[i] = 1; distributed++; total -= cnt; continue; }
        norm[i] = notYetAssigned;
    }
    uint32_t toDistribute = (1u << tableLog) - distributed;
    if ((total / toDistribute) > lowOne) {
        lowOne = (total * 3) / (toDistribute * 2);
        for (int i = 0; i < symbolLen; i++) {
            const uint32_t cnt = count[i];
            if (norm[i] == notYetAssigned && cnt <= lowOne) { norm[i] = 1; distributed++; total -= cnt; }
        }
        toDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == (uint32_t)symbolLen + 1) {
        int maxV = 0;
        uint32_t maxC = 0;
        for (int i = 0; i < symbolLen; i++)
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
    const uint64_t vStepLog = 62 - (uint64_t)tableLog;
    const uint64_t mid = (((uint64_t)1 << (vStepLog - 1)) - 1);
    const uint64_t rStep = ((((uint64_t)1 << vStepLog) * (uint64_t)toDistribute) + mid) / (uint64_t)total;
    uint64_t tmpTotal = mid;
    for (int i = 0; i < symbolLen; i++) {
        if (norm[i] == notYetAssigned) {
            const uint64_t end = tmpTotal + (uint64_t)count[i] * rStep;
            const uint32_t sStart = (uint32_t)(tmpTotal >> vStepLog);
            const uint32_t sEnd = (uint32_t)(end >> vStepLog);
            const uint32_t weight = sEnd - sStart;
            if (weight < 1) return false;
            norm[i] = (int16_t)weight;
            tmpTotal = end;
        }
    }
    return true;
}

// Table construction shared by zstd/fse_encoder.go:102-204 and fse/compress.go:349-451.
// tsym: scratch of tableSize bytes; cumul: scratch of symbolLen+2 int16.
// Writes stateTable, deltaNbBits, deltaFindState (as int32 to cover the byte coder).
// Returns false on the reference's internal errors.
template <typename DFS>
__device__ inline bool fse_build_core(const int16_t* norm, int symbolLen, uint8_t tableLog, uint8_t* tsym, int16_t* cumul,
                                      uint16_t* stateTable, uint32_t* dnb, DFS* dfs) {
    const uint32_t tableSize = 1u << tableLog;
    uint32_t highThreshold = tableSize - 1;
    cumul[0] = 0;
    for (int u = 0; u < symbolLen; u++) {
        const int16_t v = norm[u];
        if (v == -1) {
            cumul[u + 1] = (int16_t)(cumul[u] + 1);
            tsym[highThreshold] = (uint8_t)u;
            highThreshold--;
        } else {
            cumul[u + 1] = (int16_t)(cumul[u] + v);
        }
    }
    if ((uint32_t)(int32_t)cumul[symbolLen] != tableSize) return false;
    cumul[symbolLen] = (int16_t)((int16_t)tableSize + 1);
    {
        const uint32_t step = fse_table_step(tableSize);
        const uint32_t tableMask = tableSize - 1;
        uint32_t position = 0;
        for (int ui = 0; ui < symbolLen; ui++) {
            const int v = norm[ui];
            for (int k = 0; k < v; k++) {  // Go `for range v`: zero iterations for v <= 0
                tsym[position] = (uint8_t)ui;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
        if (position != 0) return false;
    }
    for (uint32_t u = 0; u < tableSize; u++) {
        const uint8_t v = tsym[u];
        stateTable[cumul[v]] = (uint16_t)(tableSize + u);
        cumul[v]++;
    }
    {
        int16_t total = 0;
        const uint32_t tl = ((uint32_t)tableLog << 16) - (1u << tableLog);
        for (int i = 0; i < symbolLen; i++) {
            const int16_t v = norm[i];
            if (v == 0) continue;  // symbolTT left stale, as in the reference (:185)
            if (v == -1 || v == 1) {
                dnb[i] = tl;
                dfs[i] = (DFS)(int16_t)(total - 1);
                total++;
            } else {
                const uint32_t maxBitsOut = (uint32_t)tableLog - high_bit((uint32_t)(int32_t)(int16_t)(v - 1));
                const uint32_t minStatePlus = (uint32_t)(int32_t)v << maxBitsOut;
                dnb[i] = (maxBitsOut << 16) - minStatePlus;
                dfs[i] = (DFS)(int16_t)(total - v);
                total = (int16_t)(total + v);
            }
        }
        if (total != (int16_t)tableSize) return false;
    }
    return true;
}

// Wave-parallel form of fse_build_core for the sequence coders (alphabet <= 64 symbols, tableLog 5..8): all 64 lanes of
// one wave call it together and obtain exactly the tables of zstd/fse_encoder.go:102-204.
//  * cumul / symbolTT: one lane per symbol, exclusive wave scan of the normalised counts;
//  * spread: the placement walk visits the cells (t*step) & mask, t = 0,1,2,.. and skips the low-probability area, so the
//    k-th placed cell is the k-th valid cell of that orbit: one lane per orbit index, ballot-prefix for k, binary search of k
//    in the per-symbol prefix sums for the symbol;
//  * stateTable: cells in ascending order take consecutive slots of their symbol: 64 cells per pass, rank among equal
//    symbols by ballot, running per-symbol slot counters in `cumul` exactly like the reference's cumul[v]++.
// posx: scratch of 66 int16.  Returns false (wave-uniform) on the reference's internal error.
__device__ inline bool fse_build_wave(const int16_t* norm, int symbolLen, uint8_t tableLog, uint8_t* tsym, int16_t* cumul, int16_t* posx,
                                      uint16_t* stateTable, uint32_t* dnb, int16_t* dfs, int lane) {
    const uint32_t tableSize = 1u << tableLog;
    const unsigned long long ltMask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const bool isSym = lane < symbolLen;
    const int v = isSym ? (int)norm[lane] : 0;
    const int cnt = v == -1 ? 1 : v;   // states owned by the symbol
    const int pcnt = v > 0 ? v : 0;    // cells placed by the spread walk
    const int incC = (int)wave_incl_scan((uint32_t)cnt, lane), incP = (int)wave_incl_scan((uint32_t)pcnt, lane);
    const int exC = incC - cnt, exP = incP - pcnt;
    const int total = __shfl(incC, 63, 64);
    if ((uint32_t)total != tableSize) return false;
    const unsigned long long lowMask = __ballot(isSym && v == -1);
    const uint32_t nLow = (uint32_t)__popcll(lowMask);
    const uint32_t highThreshold = tableSize - 1 - nLow;
    if (isSym) {
        cumul[lane] = (int16_t)exC;
        posx[lane] = (int16_t)exP;
        if (v == -1) tsym[tableSize - 1 - (uint32_t)__popcll(lowMask & ltMask)] = (uint8_t)lane;
        // symbolTT (fse_encoder.go:176-203): `total` before symbol i is the exclusive prefix of the state counts
        if (v != 0) {
            if (v == -1 || v == 1) {
                dnb[lane] = ((uint32_t)tableLog << 16) - (1u << tableLog);
                dfs[lane] = (int16_t)(exC - 1);
            } else {
                const uint32_t maxBitsOut = (uint32_t)tableLog - high_bit((uint32_t)(v - 1));
                const uint32_t minStatePlus = (uint32_t)v << maxBitsOut;
                dnb[lane] = (maxBitsOut << 16) - minStatePlus;
                dfs[lane] = (int16_t)(exC - v);
            }
        }
    }
    if (lane == 0) { cumul[symbolLen] = (int16_t)((int16_t)tableSize + 1); posx[symbolLen] = (int16_t)(tableSize - nLow); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // spread
    {
        const uint32_t step = fse_table_step(tableSize), mask = tableSize - 1;
        uint32_t placed = 0;
        for (uint32_t t0 = 0; t0 < tableSize; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            const uint32_t pos = (t * step) & mask;
            const bool valid = t < tableSize && pos <= highThreshold;
            const unsigned long long vm = __ballot(valid);
            if (valid) {
                const int k = (int)(placed + (uint32_t)__popcll(vm & ltMask));
                int lo = 0, hi = symbolLen;  // largest s with posx[s] <= k (symbols without cells share a prefix value: take the last)
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if ((int)posx[mid] <= k) lo = mid; else hi = mid;
                }
                tsym[pos] = (uint8_t)lo;
            }
            placed += (uint32_t)__popcll(vm);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // state table
    for (uint32_t u0 = 0; u0 < tableSize; u0 += 64) {
        const uint32_t u = u0 + (uint32_t)lane;
        const bool act = u < tableSize;
        const int sy = act ? (int)tsym[u] : -1;
        unsigned long long rem = __ballot(act);
        int slot = 0;
        while (rem) {
            const int leader = __builtin_ctzll(rem);
            const int ls = __shfl(sy, leader, 64);
            const unsigned long long m = __ballot(act && sy == ls);
            const int base = (int)cumul[ls];
            KC_EMU_SYNC();  // (every lane has read the counter before the leader advances it)
            if (act && sy == ls) slot = base + __popcll(m & ltMask);
            if (lane == leader) cumul[ls] = (int16_t)(base + __popcll(m));
            rem &= ~m;
        }
        if (act) stateTable[slot] = (uint16_t)(tableSize + u);
    }
    return true;
}

// NCount header writer shared by zstd/fse_encoder.go:488-600 and fse/compress.go:208-306.
// tableLogBase = minTablelog (5 in both).  Returns bytes written, or -1 on internal error.
__device__ inline int fse_write_ncount(const int16_t* norm, int symbolLen, uint8_t tableLog, uint8_t* out) {
    const int tableSize = 1 << tableLog;
    bool previous0 = false;
    uint16_t charnum = 0;
    uint32_t bitStream = (uint32_t)(tableLog - 5);
    uint32_t bitCount = 4;
    int16_t remaining = (int16_t)(tableSize + 1);
    int16_t threshold = (int16_t)tableSize;
    uint32_t nbBits = (uint32_t)tableLog + 1;
    int outP = 0;
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
        const int16_t max = (int16_t)((2 * threshold - 1) - remaining);
        if (cnt < 0) remaining = (int16_t)(remaining + cnt);
        else remaining = (int16_t)(remaining - cnt);
        cnt++;
        if (cnt >= threshold) cnt = (int16_t)(cnt + max);
        bitStream += (uint32_t)(int32_t)cnt << bitCount;
        bitCount += nbBits;
        if (cnt < max) bitCount--;
        previous0 = cnt == 1;
        if (remaining < 1) return -1;
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
    outP += (int)((bitCount + 7) / 8);
    if ((int)charnum > symbolLen) return -1;
    return outP;
}

// fseEncoder.bitCost / approxSize (zstd/fse_encoder.go:603-660)
__device__ inline uint32_t fse_approx_size(const KcFseT* f, const uint32_t* hist, int histLen) {
    if ((int)f->symbolLen < histLen) return 0xFFFFFFFFu;
    if (f->useRLE) return 0xFFFFFFFFu;
    const uint32_t kAccuracyLog = 8;
    const uint32_t badCost = ((uint32_t)f->tableLog + 1) << kAccuracyLog;
    uint32_t cost = 0;
    for (int i = 0; i < histLen; i++) {
        const uint32_t v = hist[i];
        if (v == 0) continue;
        if (f->norm[i] == 0) return 0xFFFFFFFFu;
        const uint32_t minNbBits = f->dnb[i] >> 16;
        const uint32_t threshold = (minNbBits + 1) << 16;
        const uint32_t tableSize = 1u << f->tableLog;
        const uint32_t deltaFromThreshold = threshold - (f->dnb[i] + tableSize);
        const uint32_t normalizedDelta = (deltaFromThreshold << kAccuracyLog) >> f->tableLog;
        const uint32_t bc = (minNbBits + 1) * (1u << kAccuracyLog) - normalizedDelta;
        if (bc > badCost) return 0xFFFFFFFFu;
        cost += v * bc;
    }
    return cost >> kAccuracyLog;
}
// fseEncoder.maxHeaderSize (zstd/fse_encoder.go:663)
__device__ inline uint32_t fse_max_header_size(const KcFseT* f) {
    if (f->preDefined) return 0;
    if (f->useRLE) return 8;
    return ((((uint32_t)f->symbolLen * (uint32_t)f->tableLog) >> 3) + 3) * 8;
}
// cState.init (zstd/fse_encoder.go:682): initial state for the first symbol of the stream.
__device__ inline uint16_t fse_init_state(const KcFseT* f, uint32_t code) {
    if (f->stLen1) return 0;
    const uint32_t d = f->dnb[code];
    const uint32_t nbBitsOut = (d + (1u << 15)) >> 16;
    const int32_t im = (int32_t)((nbBitsOut << 16) - d);
    const int32_t lu = (im >> nbBitsOut) + (int32_t)f->dfs[code];
    return f->st[lu];
}
