// kc_zstd_match_dfast.hip — SpeedDefault (double-fast) match finder for gfx950.
//
// Replaces doubleFastEncoder.Encode / EncodeNoHist (zstd/enc_dfast.go:38-367 / 372-675).
// Same execution scheme as the SpeedFastest group kernel (kc_zstd_match.hip, v3): 8 lanes per unit,
// 8 units per wave, speculative probing with ordered commit, tables in an HBM arena:
// long table 2^17 x u32 (8-byte hash) followed by short table 2^15 x u32 (5-byte hash) per unit,
// entry = (position+1) | tag(4 source bytes) << PB.
// Reproduced literally: step 1 / skip >>7 (kSearchStrength 8), inputMargin 10, repeat check at s+1,
// long-before-short priority, the "long match at s+1 after a short match" lookup (which writes the
// long table and must see this round's committed writes, so it is done after the commit), the four
// table inserts after a match (start+1 / end-2 long, start+2 / end-1 short), the offset-2 loop,
// canRepeat snapshot (Encode) vs live (EncodeNoHist), the maxMatchLength caps of Encode, and the
// EncodeNoHist quirk that hashes the short table with the already shifted cv1 in the offset-2 loop
// (enc_dfast.go:630).
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_wave.h"

#define ZD_LONG_BITS 17
#define ZD_SHORT_BITS 15
#define ZD_MAX_MATCH_LENGTH 131074
#ifndef ZD_W0
#define ZD_W0 4
#endif

template <int G>
__global__ __launch_bounds__(64) void kc_zdfast_match_grp_kernel(KcMatchParams P, uint32_t* __restrict__ tables, uint32_t n_launch) {
    constexpr int UPW = 64 / G;
    __shared__ uint64_t sbuf_all[UPW * G];  // per unit: the last (nseq mod G) sequences
    const int lane = (int)threadIdx.x;
    const int lig = lane % G, grp = lane / G;
    uint64_t* const sbuf = sbuf_all + grp * G;
    const uint32_t ui = blockIdx.x * UPW + (uint32_t)grp;
    const bool gact = ui < n_launch;
    const uint32_t u = gact ? (P.unit_list ? P.unit_list[ui] : P.unit_base + ui) : 0u;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    // history in front of the unit: the dictionary content, or (jobs of a WithConcurrentBlocks stream) the unit's own overlap prefix
    const int hist0 = P.unit_hist != nullptr ? (int)P.unit_hist[u] : P.hist0;
    const int ulen = gact ? (int)(P.unit_off[u + 1] - P.unit_off[u]) - hist0 : 0;
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const int mmo = P.max_match_off;
    const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
    const int nblk = gact ? UB.nblk : 0;  // a group without a unit (the launch's tail) does nothing
    const bool HIST = ulen > bs || hist0 > 0 || UB.streamU || P.job_flags != nullptr;  // compressJob always calls Encode (enc_jobs.go:114)
    uint32_t* __restrict__ ltab = tables + (size_t)ui * ((1u << ZD_LONG_BITS) + (1u << ZD_SHORT_BITS));
    uint32_t* __restrict__ stab = ltab + (1u << ZD_LONG_BITS);
    const int PB = P.pos_bits;
    const int TB = (32 - PB) > 16 ? 16 : (32 - PB);
    const uint32_t posMask = (1u << PB) - 1u;
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };
    auto mk = [&](int pos, uint32_t val) -> uint32_t { return ((uint32_t)pos + 1u) | (tagOf(val) << PB); };
    auto hL = [&](uint64_t v) -> uint32_t { return hash8(v, ZD_LONG_BITS); };
    auto hS = [&](uint64_t v) -> uint32_t { return hash5(v, ZD_SHORT_BITS); };

    int o1 = P.rep1, o2 = P.rep2;  // {1,4} (blockenc.go:78) or the dictionary's offsets (enc_base.go:189-195)
    for (int b = 0; b < nblk; b++) {
        const int blkStart = hist0 + kc_blk_begin(P.blk_start, blk0, b, bs);
        const int blkEnd = hist0 + kc_blk_end(P.blk_start, blk0, b, nblk, bs, ulen);
        const int srcLen = blkEnd - blkStart;
        const int o1_in = o1, o2_in = o2;
        uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        int nseq = 0, sumLL = 0;
        uint32_t rounds = 0;
        int nextEmit = blkStart, s = blkStart;
        uint32_t firstLL = 0, firstOf = 0;
        auto emit = [&](int ll, int ml3, uint32_t of) {
            if (nseq == 0) { firstLL = (uint32_t)ll; firstOf = of; }
            if (lig == 0) sbuf[nseq & (G - 1)] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
            nseq++;
            sumLL += ll;
            if ((nseq & (G - 1)) == 0) {  // group-uniform: G sequences buffered in LDS -> one coalesced 64-byte store (an 8-byte scattered store is a DRAM write each)
                __builtin_amdgcn_wave_barrier();
                sq[nseq - G + lig] = sbuf[lig];
                __builtin_amdgcn_wave_barrier();
            }
        };
        if (srcLen >= 16) {  // minNonLiteralBlockSize
            const int sLimit = blkEnd - 10;  // inputMargin = 8 + 2
            bool canRep = false, fin = false;
            int W = G;
            while (!fin) {
                rounds++;
                KC_EMU_SYNC();  // (lane 0's table stores behind the previous match precede this round's lookups)
                const int d0 = s - nextEmit;
                const int k0 = d0 >> 7;  // kSearchStrength-1 == 7
                const int step = 1 + k0;
                const int p = s + lig * step;
                const bool valid = lig < W && (lig == 0 || ((d0 + (lig - 1) * step) >> 7) == k0) && p < sLimit;
                const uint64_t cv = valid ? ld64(base + p) : 0ull;
                uint32_t hl = 0xFFFFFFFFu, hs = 0xFFFFFFFEu, cL = 0, cS = 0;
                if (valid) {
                    hl = hL(cv);
                    hs = hS(cv);
                    cL = ltab[hl];
                    cS = stab[hs];
                }
                bool dep = false;
#pragma unroll
                for (int d = 1; d < G; d++) {
                    const uint32_t al = (uint32_t)__shfl_up((int)hl, d, G), as = (uint32_t)__shfl_up((int)hs, d, G);
                    if (lig >= d && (al == hl || as == hs)) dep = true;
                }
                int kind = 0, t = 0;  // 1 repeat at s+1, 2 long match, 3 short match
                if (valid) {
                    const int repIndex = p - o1 + 1;
                    const bool repOk = canRep && repIndex >= 0;
                    const uint32_t eL = cL & posMask, eS = cS & posMask;
                    const int tL = (int)eL - 1, tS = (int)eS - 1;
                    const bool okL = eL != 0 && (p - tL) < mmo && (cL >> PB) == tagOf((uint32_t)cv);
                    const bool okS = eS != 0 && (p - tS) < mmo && (cS >> PB) == tagOf((uint32_t)cv);
                    const uint32_t wr = ld32(base + (repOk ? repIndex : p));
                    const uint32_t wL = ld32(base + (okL ? tL : p));
                    const uint32_t wS = ld32(base + (okS ? tS : p));
                    if (repOk && wr == (uint32_t)(cv >> 8)) kind = 1;
                    else if (okL && wL == (uint32_t)cv) { kind = 2; t = tL; }
                    else if (okS && wS == (uint32_t)cv) { kind = 3; t = tS; }
                }
                const uint32_t vm = gballot<G>(valid, grp);
                const uint32_t depm = gballot<G>(valid && dep, grp);
                const uint32_t hm = gballot<G>(kind != 0, grp);
                const int nvalid = __popc(vm);
                const int c = depm ? __builtin_ctz(depm) : G;
                const uint32_t hmc = hm & ((1u << c) - 1u);
                const bool found = hmc != 0;
                const int f = found ? __builtin_ctz(hmc) : 0;
                const int commitUpTo = found ? f : ((c < nvalid ? c : nvalid) - 1);
                if (valid && lig <= commitUpTo) {
                    const uint32_t e = mk(p, (uint32_t)cv);
                    ltab[hl] = e;
                    stab[hs] = e;
                }
                if (!found) {
                    W = P.spec_grow == 0 ? W : (P.spec_grow == 1 ? (W + 1 < G ? W + 1 : G) : ((2 * W < G) ? 2 * W : G));
                    if (c < nvalid) {
                        s = s + c * step;
                    } else {
                        const int pl = s + (nvalid - 1) * step;
                        s = pl + 1 + ((pl - nextEmit) >> 7);
                    }
                    if (s >= sLimit) fin = true;
                    continue;
                }
                const int mkd = (int)gbcast32<G>((uint32_t)kind, grp, f);
                const int ps = s + f * step;
                int mt = (int)gbcast32<G>((uint32_t)t, grp, f);
                if (mkd == 1) {
                    // ---------------- repeat at s+1 (enc_dfast.go:137-178 / 443-482) ----------------
                    int repIndex = ps - o1 + 1;
                    const int length = 4 + grp_matchlen<G>(base, ps + 5, repIndex + 4, blkEnd - (ps + 5), lig, grp);
                    int start = ps + 1;
                    const int startLimit = nextEmit + 1;
                    const int tMin = (ps - mmo) > 0 ? (ps - mmo) : 0;
                    int kmax = repIndex - tMin;
                    if (start - startLimit < kmax) kmax = start - startLimit;
                    if (HIST) {  // && seq.matchLen < maxMatchLength-zstdMinMatch-1 (:153)
                        const int cap = (ZD_MAX_MATCH_LENGTH - 3 - 1) - (length - 3);
                        if (cap < kmax) kmax = cap;
                    }
                    if (kmax < 0) kmax = 0;
                    const int back = grp_backlen<G>(base, start, repIndex, kmax, lig, grp);
                    start -= back;
                    emit(start - nextEmit, length - 3 + back, 1u);
                    W = P.spec_w0;
                    s = ps + length + 1;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                s = ps;
                if (mkd == 3) {
                    // short match: see if there is a long match at s+1 (enc_dfast.go:204-233); the lookup
                    // stores s+1 in the long table and observes this round's committed writes.
                    const uint64_t cvn = ld64(base + s + 1);
                    const uint32_t hn = hL(cvn);
                    const uint32_t cn = ltab[hn];
                    KC_EMU_SYNC();
                    if (lig == 0) ltab[hn] = mk(s + 1, (uint32_t)cvn);
                    const uint32_t en = cn & posMask;
                    const int tn = (int)en - 1;
                    // coffsetL = s - (candidateL.offset - e.cur) + checkAt
                    if (en != 0 && (s - tn + 1) < mmo && (cn >> PB) == tagOf((uint32_t)cvn) && ld32(base + tn) == (uint32_t)cvn) {
                        mt = tn;
                        s += 1;
                    }
                }
                o2 = o1;
                o1 = s - mt;
                int l = grp_matchlen<G>(base, s + 4, mt + 4, blkEnd - (s + 4), lig, grp) + 4;
                {
                    const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = mt - tMin;
                    if (s - nextEmit < kmax) kmax = s - nextEmit;
                    if (HIST && (ZD_MAX_MATCH_LENGTH - l) < kmax) kmax = ZD_MAX_MATCH_LENGTH - l;
                    if (kmax < 0) kmax = 0;
                    const int back = grp_backlen<G>(base, s, mt, kmax, lig, grp);
                    s -= back;
                    mt -= back;
                    l += back;
                }
                emit(s - nextEmit, l - 3, (uint32_t)(s - mt) + 3u);
                W = P.spec_w0;
                s += l;
                nextEmit = s;
                const bool canRepO2 = HIST ? canRep : (nseq > 2);
                canRep = nseq > 2;
                if (s >= sLimit) { fin = true; continue; }
                // index match start+1 (long) / start+2 (short) and end-2 (long) / end-1 (short) (:258-272)
                const int index0 = s - l + 1, index1 = s - 2;
                uint64_t cv0 = ld64(base + index0), cv1 = ld64(base + index1);
                if (lig == 0) {
                    ltab[hL(cv0)] = mk(index0, (uint32_t)cv0);
                    ltab[hL(cv1)] = mk(index1, (uint32_t)cv1);
                }
                cv0 >>= 8;
                cv1 >>= 8;
                if (lig == 0) {
                    stab[hS(cv0)] = mk(index0 + 1, (uint32_t)cv0);
                    stab[hS(cv1)] = mk(index1 + 1, (uint32_t)cv1);
                }
                if (!canRepO2) continue;
                // ---------------- offset-2 loop (:283-322 / 617-657) ----------------
                for (;;) {
                    const uint64_t cvs = ld64(base + s);
                    const int o2pos = s - o2;
                    if (ld32(base + o2pos) != (uint32_t)cvs) break;
                    const uint32_t nhS = HIST ? hS(cvs) : hS(cv1 >> 8);
                    const uint32_t nhL = hL(cvs);
                    const int l2 = 4 + grp_matchlen<G>(base, s + 4, o2pos + 4, blkEnd - (s + 4), lig, grp);
                    if (lig == 0) {
                        const uint32_t e = mk(s, (uint32_t)cvs);
                        ltab[nhL] = e;
                        stab[nhS] = e;
                    }
                    emit(0, l2 - 3, 1u);
                    s += l2;
                    nextEmit = s;
                    const int tmp = o1; o1 = o2; o2 = tmp;
                    canRep = nseq > 2;
                    if (s >= sLimit) { fin = true; break; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lig < (nseq & (G - 1))) sq[(nseq & ~(G - 1)) + lig] = sbuf[lig];  // the buffered tail of the sequence list
        const int extra = nextEmit < blkEnd ? blkEnd - nextEmit : 0;
        const int nlit = sumLL + extra;
        const bool rle = nseq == 1 && nlit <= 1 && (int)firstLL == nlit && firstOf - 3u == 1u;
        const int saved = srcLen - nlit - (srcLen >> 6);
        uint32_t flags = 0;
        if (nseq > 0 && !rle && saved < 16) flags |= KC_BF_POP_A;
        if (P.pop_blk != nullptr && P.pop_blk[blk0 + (uint32_t)b] != 0) flags |= KC_BF_FORCED;
        const int o1c = o1, o2c = o2;
        if (flags) { o1 = o1_in; o2 = o2_in; }
        flags |= rounds << 8;
        if (lig == 0) {
            KcBlkMeta m;
            m.nseq = (uint32_t)nseq;
            m.nlit = (uint32_t)nlit;
            m.extra_lits = (uint32_t)extra;
            m.flags = flags;
            m.o1_in = (uint32_t)o1_in; m.o2_in = (uint32_t)o2_in;
            m.o1_out = (uint32_t)o1c; m.o2_out = (uint32_t)o2c;
            P.meta[blk0 + (uint32_t)b] = m;
        }
    }
}

void kc_launch_zdfast_match_grp(const KcMatchParams& P, uint32_t* tables, uint32_t n_launch, hipStream_t st) {
    hipLaunchKernelGGL(kc_zdfast_match_grp_kernel<8>, dim3((n_launch + 7) / 8), dim3(64), 0, st, P, tables, n_launch);
}
