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
// Round 5: what round 2 gave the SpeedFastest kernel, for this one — the source bytes around the parse position come from a per-unit
// ring in LDS (refilled 128 bytes at a time, one round ahead of use) instead of a global load at the head of every round (one
// dependent round trip less per round, and the source lines are no longer re-fetched through the thrashed caches); every candidate
// (repeat, long, short) is verified with ONE 16-byte load of [t-4, t+12), which also yields the forward length below 12 and the
// backward extension below 4; the bytes of the long lookup at s+1 come from the winner's registers; the dependency check of the
// speculative probes runs over DPP row shifts.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_wave.h"
#include "kc_zfast_dev.h"

#define ZD_LONG_BITS 17
#define ZD_SHORT_BITS 15
#define ZD_MAX_MATCH_LENGTH 131074
#ifndef ZD_W0
#define ZD_W0 4
#endif
#define ZD_RB 1024      // ring bytes per unit (power of two)
#define ZD_MIRROR 32    // the first 32 ring bytes again behind the ring: 24-byte reads never wrap
#define ZD_STRIDE (ZD_RB + ZD_MIRROR)
#define ZD_BK 4         // bytes in front of a probe / candidate position kept for the backward extension
#define ZD_AHEAD 320    // refill while fewer than this many bytes are buffered ahead of s

template <int G>
__global__ __launch_bounds__(64) void kc_zdfast_match_grp_kernel(KcMatchParams P, uint32_t* __restrict__ tables, uint32_t n_launch) {
    constexpr int UPW = 64 / G;
    static_assert(G == 8, "the ring refill (16 bytes per lane = 128 per round) and the DPP distances are written for 8-lane groups");
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[UPW * ZD_STRIDE];
    __shared__ uint64_t sbuf_all[UPW * G];  // per unit: the last (nseq mod G) sequences
    const int lane = (int)threadIdx.x;
    const int lig = lane % G, grp = lane / G;
    uint8_t* const ring = ring_all + grp * ZD_STRIDE;
    uint64_t* const sbuf = sbuf_all + grp * G;
    const uint32_t ui = blockIdx.x * UPW + (uint32_t)grp;
    const bool gact = ui < n_launch;
    const uint32_t u = gact ? (P.unit_list ? P.unit_list[ui] : P.unit_base + ui) : 0u;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int boff = (int)((uintptr_t)base & 15);       // window positions are relative to the 16-byte aligned abase
    const uint8_t* __restrict__ abase = base - boff;
    const uint8_t* const srcLo = P.src;
    const uint8_t* const srcHi = P.src_end;
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
#ifdef KC_ZD_STATS   // measurement build: what a unit asks of the memory system, by source (KC_OPT_K2_PROF; slots 32..39 of the buffer)
    unsigned long long zst[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // probes looked up, probes committed, candidate / repeat 16-byte loads, long lookups at s+1, matches, offset-2 matches, ring refills, global fallbacks of the window
#define ZD_STAT(i, n) do { zst[i] += (unsigned long long)(n); } while (0)
#else
#define ZD_STAT(i, n) do { } while (0)
#endif
    int wlo = 0, whi = 0;   // the ring holds the bytes abase[wlo .. whi)
    bool pend = false;      // rf holds the 16*G bytes abase[whi ..) loaded during the previous round
    uint4 rf = make_uint4(0, 0, 0, 0);
    // 8 source bytes at unit position q (group-uniform): from the ring when it holds them, else from memory
    auto src8 = [&](int q) -> uint64_t {
        const int a = q + boff, a4 = a & ~3;
        if (a4 >= wlo && a4 + 12 <= whi) {
            const uint32_t* r = (const uint32_t*)(ring + (a4 & (ZD_RB - 1)));
            const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], sh = (uint32_t)(a & 3);
            return (uint64_t)__builtin_amdgcn_alignbyte(r1, r0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(r2, r1, sh) << 32);
        }
        return ld64(base + q);
    };
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
                // ---------------- source window (LDS ring) ----------------
                if (pend) {  // the refill issued one round ago has landed
                    const int ro = (whi + 16 * lig) & (ZD_RB - 1);
                    *(uint4*)(ring + ro) = rf;
                    if (ro < ZD_MIRROR) *(uint4*)(ring + ZD_RB + ro) = rf;
                    whi += 16 * G;
                    if (whi - wlo > ZD_RB) wlo = whi - ZD_RB;
                    pend = false;
                }
                KC_EMU_SYNC();  // (the ring is written by all lanes of the group and read by all of them; lane 0's table stores behind the previous match precede this round's lookups)
                const int sa = s + boff;
                if (sa >= whi || sa < wlo) {  // block start, or a match jumped past the window: restart it just behind s
                    int w0 = (sa - 16) & ~15;
                    if (w0 < 0) w0 = 0;
                    wlo = whi = w0;
                }
                if (whi - sa < ZD_AHEAD) {
                    const uint8_t* q = abase + whi + 16 * lig;
                    rf = make_uint4(0, 0, 0, 0);
                    if (q < srcHi) rf = *(const uint4*)q;  // aligned: never leaves the 16-byte granule of a readable byte
                    pend = true;
                    ZD_STAT(6, 1);
                }
                const int d0 = s - nextEmit;
                const int k0 = d0 >> 7;  // kSearchStrength-1 == 7
                const int step = 1 + k0;
                const int p = s + lig * step;
                const bool valid = lig < W && (lig == 0 || ((d0 + (lig - 1) * step) >> 7) == k0) && p < sLimit;
                // R = the 20 source bytes [p-4, p+16): D1:D2 = cv, the rest feeds the fused candidate compares
                uint32_t D0 = 0, D1 = 0, D2 = 0, D3 = 0, D4 = 0;
                if (valid) {
                    const int a = p + boff - ZD_BK;
                    const int a4 = a & ~3;
                    if (a4 >= wlo && a4 + 24 <= whi) {
                        const uint32_t* r = (const uint32_t*)(ring + (a4 & (ZD_RB - 1)));
                        const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
                        const uint32_t sh = (uint32_t)(a & 3);
                        D0 = __builtin_amdgcn_alignbyte(r1, r0, sh);
                        D1 = __builtin_amdgcn_alignbyte(r2, r1, sh);
                        D2 = __builtin_amdgcn_alignbyte(r3, r2, sh);
                        D3 = __builtin_amdgcn_alignbyte(r4, r3, sh);
                        D4 = __builtin_amdgcn_alignbyte(r5, r4, sh);
                    } else {
                        const uint8_t* q = base + p - ZD_BK;
                        if (q >= srcLo && q + 20 <= srcHi) {
                            const uint64_t qa = ld64(q), qb = ld64(q + 8);
                            D0 = (uint32_t)qa; D1 = (uint32_t)(qa >> 32); D2 = (uint32_t)qb; D3 = (uint32_t)(qb >> 32); D4 = ld32(q + 16);
                        } else {
                            D0 = zf_edge_dword(q, srcLo, srcHi); D1 = zf_edge_dword(q + 4, srcLo, srcHi); D2 = zf_edge_dword(q + 8, srcLo, srcHi);
                            D3 = zf_edge_dword(q + 12, srcLo, srcHi); D4 = zf_edge_dword(q + 16, srcLo, srcHi);
                        }
                    }
                }
                const uint64_t cv = (uint64_t)D1 | ((uint64_t)D2 << 32);
                // ---------------- round trip 1: both table entries and the repeat candidate ----------------
                uint32_t hl = 0xFFFFFFFFu, hs = 0xFFFFFFFEu, cL = 0, cS = 0;
                if (valid) {
                    hl = hL(cv);
                    hs = hS(cv);
                    cL = ltab[hl];
                    cS = stab[hs];
                }
                const int repIndex0 = p - o1 + 1;
                const bool repOk = valid && canRep && repIndex0 >= 0;
                uint4 cr = make_uint4(0, 0, 0, 0);
                bool repWide = false;
                if (repOk) {
                    const uint8_t* q = base + repIndex0 - ZD_BK;
                    repWide = q >= srcLo && q + 16 <= srcHi;
                    if (repWide) cr = ld128u(q);
                    else cr.y = ld32(base + repIndex0);
                }
                bool dep = false;
#define KC_DEP_STEP(d) do { const uint32_t al = kc_dpp_or0<0x110 + (d), 0xf>(hl), as = kc_dpp_or0<0x110 + (d), 0xf>(hs); \
                            if (lig >= (d) && (al == hl || as == hs)) dep = true; } while (0)
                KC_DEP_STEP(1); KC_DEP_STEP(2); KC_DEP_STEP(3); KC_DEP_STEP(4); KC_DEP_STEP(5); KC_DEP_STEP(6); KC_DEP_STEP(7);
#undef KC_DEP_STEP
                // ---------------- round trip 2: the tagged candidates, one 16-byte load each ----------------
                const uint32_t eL = cL & posMask, eS = cS & posMask;
                const int tL = (int)eL - 1, tS = (int)eS - 1;
                const bool okL = valid && eL != 0 && (p - tL) < mmo && (cL >> PB) == tagOf((uint32_t)cv);
                const bool okS = valid && eS != 0 && (p - tS) < mmo && (cS >> PB) == tagOf((uint32_t)cv);
                uint4 ca = make_uint4(0, 0, 0, 0), cb = make_uint4(0, 0, 0, 0);
                bool wideL = false, wideS = false;
                if (okL) {
                    const uint8_t* q = base + tL - ZD_BK;
                    wideL = q >= srcLo && q + 16 <= srcHi;
                    if (wideL) ca = ld128u(q);
                    else ca.y = ld32(base + tL);
                }
                if (okS) {
                    const uint8_t* q = base + tS - ZD_BK;
                    wideS = q >= srcLo && q + 16 <= srcHi;
                    if (wideS) cb = ld128u(q);
                    else cb.y = ld32(base + tS);
                }
                // per-lane verdict: 1 repeat at s+1, 2 long match, 3 short match (enc_dfast.go:137, 181, 199), with what the 16 bytes
                // say about the lengths: fwd equal bytes from the match position on (0..12), back equal bytes behind it (0..4)
                int kind = 0, t = 0, fwd = 0, back = 0, fa = 0, ba = 0;
                if (repOk) {
                    int f, bk;
                    zf_cmp16(cr, __builtin_amdgcn_alignbyte(D1, D0, 1), __builtin_amdgcn_alignbyte(D2, D1, 1),
                             __builtin_amdgcn_alignbyte(D3, D2, 1), __builtin_amdgcn_alignbyte(D4, D3, 1), f, bk);
                    if (f >= 4) { kind = 1; fwd = repWide ? f : 4; back = repWide ? bk : 0; fa = repWide ? 12 : 4; ba = repWide ? ZD_BK : 0; }
                }
                if (kind == 0 && okL) {
                    int f, bk;
                    zf_cmp16(ca, D0, D1, D2, D3, f, bk);
                    if (f >= 4) { kind = 2; t = tL; fwd = wideL ? f : 4; back = wideL ? bk : 0; fa = wideL ? 12 : 4; ba = wideL ? ZD_BK : 0; }
                }
                if (kind == 0 && okS) {
                    int f, bk;
                    zf_cmp16(cb, D0, D1, D2, D3, f, bk);
                    if (f >= 4) { kind = 3; t = tS; fwd = wideS ? f : 4; back = wideS ? bk : 0; fa = wideS ? 12 : 4; ba = wideS ? ZD_BK : 0; }
                }
                uint32_t vk = 0;  // kind:2 | length final:1 | known forward length:5 | equal bytes behind:3 | bytes behind examined:3
                if (kind != 0) {
                    const int limit = blkEnd - (p + (kind == 1 ? 1 : 0));
                    const bool done = fwd < fa || fwd >= limit;
                    const int fk = fwd < limit ? fwd : limit;
                    vk = (uint32_t)kind | (done ? 4u : 0u) | ((uint32_t)fk << 3) | ((uint32_t)back << 8) | ((uint32_t)ba << 11);
                }
                ZD_STAT(0, __popc(gballot<G>(valid, grp)));
                ZD_STAT(2, __popc(gballot<G>(okL, grp)) + __popc(gballot<G>(okS, grp)) + __popc(gballot<G>(repOk, grp)));
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
                ZD_STAT(1, commitUpTo + 1);
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
                const uint32_t wk = gbcast32<G>(vk, grp, f);
                const int mkd = (int)(wk & 3u);
                const bool fdone = (wk & 4u) != 0;
                const int fk = (int)((wk >> 3) & 31u);
                const int bke = (int)((wk >> 8) & 7u), bav = (int)((wk >> 11) & 7u);
                const int ps = s + f * step;
                int mt = (int)gbcast32<G>((uint32_t)t, grp, f);
                // backward extension given the bke equal bytes found among the bav bytes examined
                auto backlen = [&](int sp, int tp, int kmax) -> int {
                    if (kmax <= 0) return 0;
                    if (bke < bav) return bke < kmax ? bke : kmax;
                    if (kmax <= bav) return kmax;
                    return bav + grp_backlen<G>(base, sp - bav, tp - bav, kmax - bav, lig, grp);
                };
                if (mkd == 1) {
                    // ---------------- repeat at s+1 (enc_dfast.go:137-178 / 443-482) ----------------
                    int repIndex = ps - o1 + 1;
                    int length = fk;  // equal bytes from s+1 / repIndex on: 4 + matchlen(s+5, repIndex+4)
                    if (!fdone) length += grp_matchlen<G>(base, ps + 1 + fk, repIndex + fk, blkEnd - (ps + 1 + fk), lig, grp);
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
                    const int back = backlen(start, repIndex, kmax);
                    start -= back;
                    emit(start - nextEmit, length - 3 + back, 1u);
                    ZD_STAT(4, 1);
                    W = P.spec_w0;
                    s = ps + length + 1;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                s = ps;
                bool fused = true;  // the winner's 16 candidate bytes describe the match that is taken
                ZD_STAT(4, 1);
                if (mkd == 3) {
                    ZD_STAT(3, 1);
                    // short match: see if there is a long match at s+1 (enc_dfast.go:204-233); the lookup
                    // stores s+1 in the long table and observes this round's committed writes.  Its 8 bytes are the winner's.
                    const uint64_t cvn = gbcast64<G>((uint64_t)__builtin_amdgcn_alignbyte(D2, D1, 1) | ((uint64_t)__builtin_amdgcn_alignbyte(D3, D2, 1) << 32), grp, f);
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
                        fused = false;
                    }
                }
                o2 = o1;
                o1 = s - mt;
                int l;
                if (fused) {
                    l = fk;
                    if (!fdone) l += grp_matchlen<G>(base, s + fk, mt + fk, blkEnd - (s + fk), lig, grp);
                } else {
                    l = grp_matchlen<G>(base, s + 4, mt + 4, blkEnd - (s + 4), lig, grp) + 4;
                }
                {
                    const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = mt - tMin;
                    if (s - nextEmit < kmax) kmax = s - nextEmit;
                    if (HIST && (ZD_MAX_MATCH_LENGTH - l) < kmax) kmax = ZD_MAX_MATCH_LENGTH - l;
                    if (kmax < 0) kmax = 0;
                    const int back = fused ? backlen(s, mt, kmax) : grp_backlen<G>(base, s, mt, kmax, lig, grp);
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
                uint64_t cv0 = src8(index0), cv1 = src8(index1);
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
                    const uint64_t cvs = src8(s);
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
                    ZD_STAT(5, 1);
                    s += l2;
                    nextEmit = s;
                    const int tmp = o1; o1 = o2; o2 = tmp;
                    canRep = nseq > 2;
                    if (s >= sLimit) { fin = true; break; }
                }
            }
        }
        pend = false;  // a refill still in flight at the end of a block is dropped; the window itself stays valid
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
#ifdef KC_ZD_STATS
    if (P.prof != nullptr && lig == 0 && gact) for (int i = 0; i < 8; i++) atomicAdd(&P.prof[i], zst[i]);
#endif
}

void kc_launch_zdfast_match_grp(const KcMatchParams& P, uint32_t* tables, uint32_t n_launch, hipStream_t st) {
    hipLaunchKernelGGL(kc_zdfast_match_grp_kernel<8>, dim3((n_launch + 7) / 8), dim3(64), 0, st, P, tables, n_launch);
}
