// kc_huf_dev.h — device-side Huffman (huff0) table construction.
//
// Follows huff0/compress.go: optimalTableLog :428, huffSort :570, buildCTable :457,
// setMaxHeight :609, and huff0/huff0.go cTable.write :180 with fse.Compress of the weights
// (fse/compress.go:18-204).  The symbol sort is computed as a parallel rank (stable by
// count descending, symbol ascending == the reference's bucketed insertion sort); the tree is
// built by one wave (huf_build_wave), the weight-table serialisation is O(256) serial work done
// by one lane on LDS-resident arrays.
#pragma once
#include "kc_dev.h"
#include "kc_fse_dev.h"

#define HUF_TABLELOG_MAX 11
#define HUF_NODES 512

struct KcHufNodes {          // huffNode[-1 .. 511] stored at index+1 (huffNode0 view of the reference)
    uint32_t count[HUF_NODES + 1];
    uint16_t parent[HUF_NODES + 1];
    uint8_t nbits[HUF_NODES + 1];
    uint8_t symbol[HUF_NODES + 1];
};

struct KcHufTable {  // cTable: val/nBits per symbol
    uint16_t val[256];
    uint8_t nb[256];
};

// huff0.Scratch.optimalTableLog (compress.go:428) with TableLog = 11
__device__ inline uint8_t huf_optimal_table_log(int srcLen, int symbolLen) {
    uint8_t tableLog = 11;
    uint32_t minBitsSrc = high_bit((uint32_t)srcLen) + 1;
    uint32_t minBitsSymbols = high_bit((uint32_t)(uint16_t)(symbolLen - 1)) + 2;
    uint8_t minBits = (uint8_t)(minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols);
    uint8_t maxBitsSrc = (uint8_t)((uint8_t)high_bit((uint32_t)(srcLen - 1)) - 1);
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 11) tableLog = 11;
    return tableLog;
}

// N: nodes (index+1 addressing handled by the macros below).
#define HN_CNT(i) N->count[(i) + 1]
#define HN_PAR(i) N->parent[(i) + 1]
#define HN_NB(i) N->nbits[(i) + 1]
#define HN_SYM(i) N->symbol[(i) + 1]

// ---------------------------------------------------------------------------------------
// buildCTable (compress.go:457-567) on a whole wave.  Precondition: nodes 0..symbolLen-1 hold the symbols sorted by
// (count desc, symbol asc) (huffSort).  Only the two-queue merge (inherently sequential, 255 steps) runs on one lane, with the
// two queue heads cached in registers, and everything around it is spread over the 64 lanes:
//   * node depths: nBits(n) = nBits(parent(n)) + 1 is relaxed in parallel until nothing changes (a node's parent has a
//     higher index, so iteration k fixes every node of depth <= k; typically 12-20 iterations instead of a 511-step chain);
//   * nbPerRank / valPerRank / the table fill: ballots per code length give, for a symbol, the number of earlier symbols with
//     the same length — the serial loop's running valPerRank counter;
//   * setMaxHeight (only when the tree is deeper than tableLog) stays on lane 0, its work arrays in LDS instead of scratch.
// All 64 lanes of ONE wave must call; LDS accesses of a wave execute in program order, the wave barriers keep the compiler
// from moving them.  Returns actualTableLog, or 0xFF on internal error (wave-uniform).
// ---------------------------------------------------------------------------------------
struct KcHufWaveTmp {          // LDS scratch of the wave build
    uint32_t rankLast[HUF_TABLELOG_MAX + 2];
    uint16_t nbPerRank[HUF_TABLELOG_MAX + 5];
    uint16_t valPerRank[16];
};

__device__ inline uint8_t huf_set_max_height_lds(KcHufNodes* N, int lastNonNull, uint8_t maxNbBits, uint32_t* rankLast) {
    const uint8_t largestBits = HN_NB(lastNonNull);
    if (largestBits <= maxNbBits) return largestBits;
    int totalCost = 0;
    const int baseCost = 1 << (largestBits - maxNbBits);
    uint32_t n = (uint32_t)lastNonNull;
    while (HN_NB(n) > maxNbBits) {
        totalCost += baseCost - (1 << (largestBits - HN_NB(n)));
        HN_NB(n) = maxNbBits;
        n--;
    }
    while (HN_NB(n) == maxNbBits) n--;
    totalCost >>= (largestBits - maxNbBits);
    const uint32_t noSymbol = 0xF0F0F0F0u;
    for (int i = 0; i < HUF_TABLELOG_MAX + 2; i++) rankLast[i] = noSymbol;
    {
        uint8_t currentNbBits = maxNbBits;
        for (int pos = (int)n; pos >= 0; pos--) {
            const uint8_t nbp = HN_NB(pos);
            if (nbp >= currentNbBits) continue;
            currentNbBits = nbp;
            rankLast[maxNbBits - currentNbBits] = (uint32_t)pos;
        }
    }
    while (totalCost > 0) {
        uint8_t nBitsToDecrease = (uint8_t)((uint8_t)high_bit((uint32_t)totalCost) + 1);
        for (; nBitsToDecrease > 1; nBitsToDecrease--) {
            const uint32_t highPos = rankLast[nBitsToDecrease];
            const uint32_t lowPos = rankLast[nBitsToDecrease - 1];
            if (highPos == noSymbol) continue;
            if (lowPos == noSymbol) break;
            const uint32_t highTotal = HN_CNT(highPos);
            const uint32_t lowTotal = 2 * HN_CNT(lowPos);
            if (highTotal <= lowTotal) break;
        }
        while (nBitsToDecrease <= HUF_TABLELOG_MAX && rankLast[nBitsToDecrease] == noSymbol) nBitsToDecrease++;
        totalCost -= 1 << (nBitsToDecrease - 1);
        if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
        HN_NB(rankLast[nBitsToDecrease]) = (uint8_t)(1 + HN_NB(rankLast[nBitsToDecrease]));
        if (rankLast[nBitsToDecrease] == 0) {
            rankLast[nBitsToDecrease] = noSymbol;
        } else {
            rankLast[nBitsToDecrease]--;
            if (HN_NB(rankLast[nBitsToDecrease]) != (uint8_t)(maxNbBits - nBitsToDecrease)) rankLast[nBitsToDecrease] = noSymbol;
        }
    }
    while (totalCost < 0) {
        if (rankLast[1] == noSymbol) {
            while (HN_NB(n) == maxNbBits) n--;
            HN_NB(n + 1) = (uint8_t)(HN_NB(n + 1) - 1);
            rankLast[1] = n + 1;
            totalCost++;
            continue;
        }
        HN_NB(rankLast[1] + 1) = (uint8_t)(HN_NB(rankLast[1] + 1) - 1);
        rankLast[1]++;
        totalCost++;
    }
    return maxNbBits;
}

// LDS hand-off between the lanes of one wave: KC_WAVE_SYNC (kc_dev.h)

__device__ inline uint8_t huf_build_wave(KcHufNodes* N, KcHufTable* T, KcHufWaveTmp* W, int symbolLen, int srcLen, int lane) {
    const uint8_t tableLog0 = huf_optimal_table_log(srcLen, symbolLen);
    for (int i = lane; i < symbolLen; i += 64) { T->val[i] = 0; T->nb[i] = 0; }
    // nonNullRank: the counts are sorted descending, the non-zero ones form a prefix
    int nz = 0;
    for (int i0 = 0; i0 < symbolLen; i0 += 64) {
        const int i = i0 + lane;
        nz += __popcll(ballot64(i < symbolLen && HN_CNT(i) != 0));
    }
    const int nonNullRank = nz - 1;
    const int startNode = symbolLen;
    const int nodeRoot = startNode + nonNullRank - 1;
    for (int n = startNode + 1 + lane; n <= nodeRoot; n += 64) HN_CNT(n) = 1u << 30;
    if (lane == 0) HN_CNT(-1) = 1u << 31;  // fake entry, strong barrier
    KC_WAVE_SYNC();
    if (lane == 0) {
        // the two-queue merge: leaves from lowS downwards, internal nodes from lowN upwards; heads cached in cS / cN
        int lowS = nonNullRank, lowN = startNode, nodeNb = startNode;
        {
            const uint32_t sum = HN_CNT(lowS) + HN_CNT(lowS - 1);
            HN_CNT(nodeNb) = sum;
            HN_PAR(lowS) = (uint16_t)nodeNb;
            HN_PAR(lowS - 1) = (uint16_t)nodeNb;
            nodeNb++;
            lowS -= 2;
        }
        uint32_t cS = HN_CNT(lowS), cN = HN_CNT(lowN);
        while (nodeNb <= nodeRoot) {
            int n1, n2;
            uint32_t c1, c2;
            if (cS < cN) { n1 = lowS; c1 = cS; lowS--; cS = HN_CNT(lowS); } else { n1 = lowN; c1 = cN; lowN++; cN = HN_CNT(lowN); }
            if (cS < cN) { n2 = lowS; c2 = cS; lowS--; cS = HN_CNT(lowS); } else { n2 = lowN; c2 = cN; lowN++; cN = HN_CNT(lowN); }
            const uint32_t sum = c1 + c2;
            HN_CNT(nodeNb) = sum;
            if (lowN == nodeNb) cN = sum;  // the internal queue's head is the node just created
            HN_PAR(n1) = (uint16_t)nodeNb;
            HN_PAR(n2) = (uint16_t)nodeNb;
            nodeNb++;
        }
        HN_NB(nodeRoot) = 0;
    }
    KC_WAVE_SYNC();
    // depths of the internal nodes: relax nBits(n) = nBits(parent(n)) + 1 until stable
    for (int n = startNode + lane; n < nodeRoot; n += 64) HN_NB(n) = 0xFF;
    KC_WAVE_SYNC();
    for (int iter = 0; iter < HUF_NODES; iter++) {
        bool changed = false;
        uint8_t nv[4];
        int nn = 0;
        for (int n = startNode + lane; n < nodeRoot; n += 64, nn++) {
            const uint8_t pb = HN_NB(HN_PAR(n));
            nv[nn] = pb == 0xFF ? (uint8_t)0xFF : (uint8_t)(pb + 1);
        }
        KC_WAVE_SYNC();
        nn = 0;
        for (int n = startNode + lane; n < nodeRoot; n += 64, nn++) {
            if (HN_NB(n) != nv[nn]) { HN_NB(n) = nv[nn]; changed = true; }
        }
        KC_WAVE_SYNC();
        if (ballot64(changed) == 0) break;
    }
    for (int n = lane; n <= nonNullRank; n += 64) HN_NB(n) = (uint8_t)(HN_NB(HN_PAR(n)) + 1);
    KC_WAVE_SYNC();
    uint8_t maxNbBits = 0;
    if (lane == 0) maxNbBits = huf_set_max_height_lds(N, nonNullRank, tableLog0, W->rankLast);
    maxNbBits = (uint8_t)bcast32(maxNbBits, 0);
    KC_WAVE_SYNC();
    if (maxNbBits > HUF_TABLELOG_MAX) return 0xFF;
    // code lengths per symbol, then canonical values: valPerRank[n] = first value of length n (from nbPerRank), a symbol's
    // value = valPerRank[its length] + number of lower-numbered symbols of the same length
    for (int i = lane; i <= nonNullRank; i += 64) T->nb[HN_SYM(i)] = HN_NB(i);
    if (lane <= HUF_TABLELOG_MAX) W->nbPerRank[lane] = 0;
    KC_WAVE_SYNC();
    uint32_t base[HUF_TABLELOG_MAX + 1];  // running count of symbols per length seen in earlier chunks (lane-uniform)
#pragma unroll
    for (int r = 0; r <= HUF_TABLELOG_MAX; r++) base[r] = 0;
    uint32_t within[4] = {0, 0, 0, 0};   // for my symbol of chunk k: same-length symbols before it (all chunks)
    uint8_t mynb[4] = {0, 0, 0, 0};
    {
        int k = 0;
        for (int i0 = 0; i0 < symbolLen; i0 += 64, k++) {
            const int i = i0 + lane;
            const uint8_t b = i < symbolLen ? (uint8_t)(T->nb[i] & 15) : (uint8_t)0xFF;
            mynb[k] = b;
#pragma unroll
            for (int r = 0; r <= HUF_TABLELOG_MAX; r++) {
                const uint64_t m = ballot64(b == r);
                if (b == r) within[k] = base[r] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                base[r] += (uint32_t)__popcll(m);
            }
        }
    }
    // nbPerRank counts the symbols 0..nonNullRank by length; symbols with a zero count have length 0 in T->nb as well,
    // but the reference counts only ranks <= nonNullRank: length-0 entries never receive a value that is used.
    if (lane == 0) {
        uint16_t minv = 0;
        for (int n = maxNbBits; n > 0; n--) {
            W->valPerRank[n] = minv;
            minv = (uint16_t)(minv + base[n]);
            minv >>= 1;
        }
        W->valPerRank[0] = 0;
    }
    KC_WAVE_SYNC();
    {
        int k = 0;
        for (int i0 = 0; i0 < symbolLen; i0 += 64, k++) {
            const int i = i0 + lane;
            if (i < symbolLen) T->val[i] = (uint16_t)(W->valPerRank[mynb[k] & 15] + within[k]);
        }
    }
    KC_WAVE_SYNC();
    return maxNbBits;
}

// ---- byte FSE of the Huffman weights (fse.Compress with TableLog 6; fse/compress.go:18-204) ----
struct KcWeightFse {  // scratch, LDS
    uint32_t count[16];
    int16_t norm[16];
    uint16_t st[64];
    uint32_t dnb[16];
    int16_t dfs[16];
    uint8_t tsym[64];
    int16_t cumul[18];
    int16_t posx[18];
};

struct KcBitW {  // serial LSB-first bit writer into a byte buffer (fse/bitwriter.go)
    uint64_t acc;
    int nb;
    uint8_t* out;
    int pos;
    __device__ void add(uint32_t value, int bits) {
        if (bits == 0) return;
        acc |= (uint64_t)(value & ((1u << bits) - 1u)) << nb;
        nb += bits;
        while (nb >= 8) { out[pos++] = (uint8_t)acc; acc >>= 8; nb -= 8; }
    }
    __device__ void close() {  // end mark + align
        add(1, 1);
        if (nb > 0) { out[pos++] = (uint8_t)acc; acc = 0; nb = 0; }
    }
};

// Returns compressed size (header + stream) written to out, or -1 when fse.Compress would
// return an error (incompressible / RLE / internal), in which case huff0 falls back to raw
// 4-bit weights.  W: scratch with count[] already holding the weight histogram.
__device__ inline int huf_fse_compress_weights(const uint8_t* w, int n, int huffMax, int huffMaxCnt, KcWeightFse* W, uint8_t* out, int outCap, int lane) {
    // called by all 64 lanes of one wave: checks and tableLog are wave-uniform, normalizeCount / writeCount run on lane 0,
    // buildCTable on the wave (fse_build_wave), the two-state encode loop on lane 0
    if (n <= 1) return -1;
    const int symbolLen = huffMax + 1;
    const int maxCount = huffMaxCnt;
    if (maxCount == n) return -1;                       // ErrUseRLE
    if (maxCount == 1 || maxCount < (n >> 7)) return -1;  // ErrIncompressible
    // optimalTableLog (fse/compress.go:484) with TableLog = 6
    uint8_t tableLog = 6;
    {
        uint32_t minBitsSrc = high_bit((uint32_t)(n - 1)) + 1;
        uint32_t minBitsSymbols = high_bit((uint32_t)(uint16_t)(symbolLen - 1)) + 2;
        uint8_t minBits = (uint8_t)(minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols);
        uint8_t maxBitsSrc = (uint8_t)((uint8_t)high_bit((uint32_t)(n - 1)) - 2);
        if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
        if (minBits > tableLog) tableLog = minBits;
        if (tableLog < 5) tableLog = 5;
        if (tableLog > 12) tableLog = 12;
    }
    if (tableLog > 6) return -2;  // cannot happen for <=256 weights; guarded for the scratch sizes
    int hdr = -1;
    if (lane == 0 && fse_normalize_core(W->count, W->norm, symbolLen, n, tableLog)) hdr = fse_write_ncount(W->norm, symbolLen, tableLog, out);
    hdr = __shfl(hdr, 0, 64);
    KC_WAVE_SYNC();
    if (hdr < 0) return -1;
    if (!fse_build_wave(W->norm, symbolLen, tableLog, W->tsym, W->cumul, W->posx, W->st, W->dnb, W->dfs, lane)) return -1;
    KC_WAVE_SYNC();
    if (n <= 2) return -1;  // compress: "src too small"
    int res = 0;
    if (lane == 0) {
    KcBitW bw;
    bw.acc = 0; bw.nb = 0; bw.out = out; bw.pos = hdr;
    uint16_t c1 = 0, c2 = 0;
    auto init = [&](uint8_t sym) -> uint16_t {
        const uint32_t d = W->dnb[sym];
        const uint32_t nbBitsOut = (d + (1u << 15)) >> 16;
        const int32_t im = (int32_t)((nbBitsOut << 16) - d);
        const int32_t lu = (im >> nbBitsOut) + W->dfs[sym];
        return W->st[lu];
    };
    auto enc = [&](uint16_t& state, uint8_t sym) {
        const uint32_t nbBitsOut = ((uint32_t)state + W->dnb[sym]) >> 16;
        const int32_t dstState = (int32_t)(state >> (nbBitsOut & 15)) + W->dfs[sym];
        bw.add(state, (int)nbBitsOut);
        state = W->st[dstState];
    };
    int ip = n;
    if (ip & 1) {
        c1 = init(w[ip - 1]);
        c2 = init(w[ip - 2]);
        enc(c1, w[ip - 3]);
        ip -= 3;
    } else {
        c2 = init(w[ip - 1]);
        c1 = init(w[ip - 2]);
        ip -= 2;
    }
    if (ip & 2) {
        enc(c2, w[ip - 1]);
        enc(c1, w[ip - 2]);
        ip -= 2;
    }
    for (; ip >= 4; ip -= 4) {
        enc(c2, w[ip - 1]);
        enc(c1, w[ip - 2]);
        enc(c2, w[ip - 3]);
        enc(c1, w[ip - 4]);
    }
    bw.add(c2, tableLog);
    bw.add(c1, tableLog);
    bw.close();
    res = bw.pos > outCap ? -2 : (bw.pos >= n ? -1 : bw.pos);  // "len(s.Out) >= len(in)" → ErrIncompressible
    }
    res = __shfl(res, 0, 64);
    KC_WAVE_SYNC();
    return res;
}

// First half of cTable.write on a whole wave: weights[n] = bitsToWeight[nBits[n]] (huff0.go:186-199) for every symbol but the
// last, and their histogram in W->count[0..15] (ballot counts; a serial loop over ~100 symbols with a dynamically indexed local
// array, i.e. scratch memory, was the most expensive single-lane part of the table description).
__device__ inline void huf_weights_wave(const KcHufTable* T, int symbolLen, uint8_t huffLog, uint8_t* weights, KcWeightFse* W, int lane) {
    const int maxSym = (int)(uint8_t)(symbolLen - 1);
    uint32_t cnt = 0;
    for (int n0 = 0; n0 < maxSym; n0 += 64) {
        const int n = n0 + lane;
        const bool act = n < maxSym;
        const uint32_t nbv = act ? T->nb[n] : 0u;
        const uint32_t v = (nbv >= 1 && nbv <= (uint32_t)huffLog) ? (uint32_t)huffLog + 1u - nbv : 0u;  // bitsToWeight
        if (act) weights[n] = (uint8_t)(v & 15u);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t m = ballot64(act && (v & 15u) == (uint32_t)k);
            if (lane == k) cnt += (uint32_t)__popcll(m);
        }
    }
    if (lane < 16) W->count[lane] = cnt;
    KC_WAVE_SYNC();
}

// cTable.write (huff0/huff0.go:180) on one wave: serialise the weights of the table into out; weights[] and W->count[] come from
// huf_weights_wave.  Returns (wave-uniform) the description length, or -1 for ErrIncompressible (maxSymbolValue > 128 with no
// FSE gain).
__device__ inline int huf_write_table(int symbolLen, uint8_t* weights, KcWeightFse* W, uint8_t* out, int outCap, int lane) {
    const int maxSym = (int)(uint8_t)(symbolLen - 1);
    if (maxSym >= 2) {
        const uint32_t c = lane < 16 ? W->count[lane] : 0u;
        const uint32_t huffMaxCnt = wave_reduce_max(c);
        const uint64_t nzm = ballot64(c != 0u);
        const int huffMax = nzm ? 63 - __builtin_clzll(nzm) : 0;
        const int r = huf_fse_compress_weights(weights, maxSym, huffMax, (int)huffMaxCnt, W, out + 1, outCap - 1, lane);
        if (r >= 0 && r < (symbolLen >> 1)) {
            if (lane == 0) out[0] = (uint8_t)r;
            KC_WAVE_SYNC();
            return 1 + r;
        }
    }
    if (maxSym > (256 - 128)) return -1;
    if (lane == 0) { out[0] = (uint8_t)(128 | (maxSym - 1)); weights[maxSym] = 0; }
    KC_WAVE_SYNC();
    const int np = (maxSym + 1) >> 1;
    for (int j = lane; j < np; j += 64) out[1 + j] = (uint8_t)((weights[2 * j] << 4) | weights[2 * j + 1]);
    KC_WAVE_SYNC();
    return 1 + np;
}
