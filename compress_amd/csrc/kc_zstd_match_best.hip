// kc_zstd_match_best.hip — SpeedBestCompression match finder: one wave per unit, persistent waves over table slots.
//
// Replaces bestFastEncoder.Encode / EncodeNoHist / Reset(dict) / ResetPrefix (zstd/enc_best.go:80-568) with match.estBits
// (:39-62) and compress.ShannonEntropyBits (compressible.go:68-85), behind EncodeAll / Write..Close / WithConcurrentBlocks at
// WithEncoderLevel(SpeedBestCompression).  The encoder keeps a long (8-byte hash, 2^22) and a short (4-byte hash, 2^18) table
// of {offset, prev} pairs — 34 MiB —, tries up to 13 candidates at a position (4 from the tables, 9 repeat-offset forms), after a
// match 5 more at s+1 / s+2 and 2 behind the end of the best match, prices every candidate in bits with the predefined FSE
// tables and an entropy estimate of the block, and indexes every position of every match.
//
// Mapping.  The control flow is the reference's and wave-uniform (state in scalar registers).  The lanes take what is
// independent: every candidate of a phase is checked, extended (forwards, and backwards in both variants improve() can ask for)
// and priced by its own lane; the order-dependent part of improve() — the quick reject against the current best, the "extend
// backwards unless the CURRENT BEST is a repeat" rule and the acceptance test — is then replayed in the reference's order, one
// ballot per accepted candidate.  Table updates after a match take 64 positions per pass, buckets hit twice within a pass applied
// in order (the {offset, prev} chain is order-dependent).
//
// Tables.  34 MiB per unit cannot be zeroed per unit (270 x the unit's bytes).  Like the reference, which never clears its
// tables between frames but moves e.cur past everything indexed so far (enc_base.go:160-175), a wave keeps ONE table slot for
// all the units it encodes and a running `cur`: entries are position + cur, an entry of an earlier unit decodes to an offset
// <= -maxMatchOff and fails the window test exactly as a zero entry does.  The slot is cleared when cur would pass 2^31 minus
// the reference's margin (about every 250 units at the default window).  `cur` lives in device memory between launches.
#include "kc_dev.h"
#include "kc_kernels.h"

#define ZB_LBITS 22
#define ZB_SBITS 18
#define ZB_HIGH (131074 * 8)   // highScore = maxMatchLen * 8 (enc_best.go:37)
#define ZB_GOOD 250            // goodEnough
#define ZB_MAXML 131074        // maxMatchLength

__device__ __forceinline__ uint32_t zb_hashL(uint64_t u) { return hash8(u, ZB_LBITS); }
__device__ __forceinline__ uint32_t zb_hashS(uint64_t u) { return hash4((uint32_t)u, ZB_SBITS); }

__device__ __forceinline__ uint32_t zb_ml_code(uint32_t mlBase) {  // seqenc.go:100 mlCode
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

// math.Log / math.Log2 of the Go runtime (math/log.go: FDLIBM e_log; math/log10.go), in IEEE double operations with no fused
// multiply-add (the Go compiler does not fuse on amd64): the block's entropy estimate decides matches, so its last bit counts.
__device__ inline double zb_golog(double x) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    const double HalfSqrt2 = 1.41421356237309504880168872420969808 / 2;
    int ki;
    double f1 = __builtin_frexp(x, &ki);
    if (f1 < HalfSqrt2) { f1 = f1 * 2; ki--; }
    const double f = f1 - 1;
    const double k = (double)ki;
    const double s = f / (2 + f);
    const double s2 = s * s;
    const double s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
__device__ inline double zb_golog2(double x) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    int e;
    const double frac = __builtin_frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    const double a = zb_golog(frac) * 1.4426950408889634;  // 1/Ln2 as a float64 constant
    return a + (double)e;
}
// one symbol's term of ShannonEntropyBits: ceil(-log2(n * invTotal) * n), an integer
__device__ inline long long zb_shannon_term(uint32_t cnt, double invTotal) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    const double n = (double)cnt;
    const double p = n * invTotal;
    const double t = -zb_golog2(p) * n;
    return (long long)__builtin_ceil(t);
}

// cost[0..31]: offset code -> outBits + deltaNbBits >> 16 of the predefined offset encoder; cost[32..95]: match-length code
// (enc_best.go:48-53).  Built once per context from the predefined tables the entropy stage uses.
struct KcFseTView {  // the leading fields of KcFseT (kc_fse_dev.h)
    uint32_t dnb[64];
    int16_t dfs[64];
    uint16_t st[256];
    int16_t norm[64];
    uint8_t outBits[64];
    uint16_t symbolLen;
    uint8_t tableLog, useRLE, rleVal, reUsed, preDefined, stLen1;
};
__global__ void kc_zbest_cost_kernel(const KcFseTView* predef, int32_t* cost) {
    const int i = (int)threadIdx.x;
    if (i < 32) cost[i] = (int32_t)predef[1].outBits[i] + (int32_t)(predef[1].dnb[i] >> 16);
    if (i < 64) cost[32 + i] = (int32_t)predef[2].outBits[i] + (int32_t)(predef[2].dnb[i] >> 16);
}
void kc_launch_zbest_cost(const void* d_predef, int32_t* d_cost, hipStream_t st) {
    hipLaunchKernelGGL(kc_zbest_cost_kernel, dim3(1), dim3(64), 0, st, (const KcFseTView*)d_predef, d_cost);
}

struct ZbMatch { int offset, s, length, rep, est; };

__global__ __launch_bounds__(64) void kc_zbest_match_kernel(KcMatchParams P, uint64_t* tables, uint32_t* slot_cur, const int32_t* cost_g,
                                                             uint32_t n_launch, uint32_t n_slots) {
    __shared__ uint32_t hist[256];
    __shared__ int32_t cost[96];
    __shared__ unsigned long long shsum;
    __shared__ uint64_t sbuf[64];
    const int lane = (int)threadIdx.x;
    const uint32_t slot = blockIdx.x;
    if (slot >= n_slots) return;
    uint64_t* const lT = tables + (size_t)slot * (((size_t)1 << ZB_LBITS) + ((size_t)1 << ZB_SBITS));
    uint64_t* const sT = lT + ((size_t)1 << ZB_LBITS);
    for (int i = lane; i < 96; i += 64) cost[i] = cost_g[i];
    KC_WAVE_SYNC();
    const int W = P.max_match_off;
    // bufferReset = MaxInt32 - 2 * window (encoder_options.go:51-73)
    const int64_t bufferReset = (int64_t)0x7fffffff - 2 * (int64_t)W;
    int64_t cur64 = (int64_t)slot_cur[slot];  // 0: a fresh slot (zeroed tables); else the cur + length of the last unit encoded here

    for (uint32_t ui = slot; ui < n_launch; ui += n_slots) {
        const uint32_t u = P.unit_list ? P.unit_list[ui] : P.unit_base + ui;
        const uint8_t* __restrict__ src = P.src + P.unit_off[u];
        const int hist0 = P.unit_hist != nullptr ? (int)P.unit_hist[u] : P.hist0;
        const int tlen = (int)(P.unit_off[u + 1] - P.unit_off[u]);  // history included
        const int ulen = tlen - hist0;
        const uint32_t blk0 = P.unit_blk0[u];
        const int bs = P.block_size;
        const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
        const int nblk = UB.nblk;

        // ---- Reset: move cur past everything this slot has indexed (resetBase, enc_base.go:172-175), or clear ----
        cur64 += (int64_t)W;
        if (cur64 + (int64_t)tlen >= bufferReset) {  // (the reference shifts or clears at this point: with no live history, clears)
            for (size_t i = (size_t)lane; i < ((size_t)1 << ZB_LBITS) + ((size_t)1 << ZB_SBITS); i += 64) lT[i] = 0;
#ifndef KC_HIPEMU
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
#endif
            KC_WAVE_SYNC();
            cur64 = (int64_t)W;
        }
        const int cur = (int)cur64;

        // chain insert of the positions [a, bL) into the long table and [a, bS) into the short one, ascending
        auto index_range = [&](int a, int bL, int bS) {
            const int bmax = bL > bS ? bL : bS;
            for (int i0 = a; i0 < bmax; i0 += 64) {
                const int i = i0 + lane;
                const bool actL = i < bL, actS = i < bS;
                uint32_t hl = 0xFFFFFFFFu, hs = 0xFFFFFFFFu;
                if (actL) { const uint64_t cv0 = ld64(src + i); hl = zb_hashL(cv0); if (actS) hs = zb_hashS(cv0); }
                else if (actS) hs = zb_hashS((uint64_t)ld32(src + i));  // (past the long table's range: the last bytes of the history)
                bool dupL = false, dupS = false;
                const int npass = bmax - i0 < 64 ? bmax - i0 : 64;
                for (int k = 0; k + 1 < npass; k++) {
                    const uint32_t kl = rdlane32(hl, k), ks = rdlane32(hs, k);
                    if (k < lane) { dupL = dupL || kl == hl; dupS = dupS || ks == hs; }
                }
                const uint32_t off = (uint32_t)(i + cur);
                if (actL && !dupL) { const uint64_t old = lT[hl]; lT[hl] = (uint64_t)off | (old << 32); }
                if (actS && !dupS) { const uint64_t old = sT[hs]; sT[hs] = (uint64_t)off | (old << 32); }
                uint64_t mL = ballot64(actL && dupL), mS = ballot64(actS && dupS);
                while (mL) {  // in order: each sees the entry its predecessors of the pass left
                    const int k = ctz64(mL);
                    mL &= mL - 1;
                    KC_WAVE_SYNC();
                    if (lane == k) { const uint64_t old = lT[hl]; lT[hl] = (uint64_t)off | (old << 32); }
                }
                while (mS) {
                    const int k = ctz64(mS);
                    mS &= mS - 1;
                    KC_WAVE_SYNC();
                    if (lane == k) { const uint64_t old = sT[hs]; sT[hs] = (uint64_t)off | (old << 32); }
                }
                KC_WAVE_SYNC();
            }
        };
        // ---- history in front of the unit: a dictionary (Reset, :474-552) or a job's overlap prefix (ResetPrefix, :554-568) ----
        if (hist0 >= 8) {
            if (P.job_flags != nullptr) index_range(0, hist0 - 8, hist0 - 8);
            else index_range(0, hist0 > 8 ? hist0 - 8 : 1, (hist0 - 8 + 3) & ~3);  // long: position 0 always; short: four positions per step
        }

        int o1 = P.rep1, o2 = P.rep2, o3 = P.rep3;
        for (int b = 0; b < nblk; b++) {
            const int blkStart = hist0 + kc_blk_begin(P.blk_start, blk0, b, bs);
            const int blkEnd = hist0 + kc_blk_end(P.blk_start, blk0, b, nblk, bs, ulen);
            const int srcLen = blkEnd - blkStart;
            const int o1_in = o1, o2_in = o2;
            uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
            int nseq = 0, sumLL = 0;
            uint32_t firstLL = 0, firstOf = 0, diag = 0;
            int nextEmit = blkStart, s = blkStart;
            auto emit = [&](int ll, int ml3, uint32_t of) {
                if (nseq == 0) { firstLL = (uint32_t)ll; firstOf = of; }
                if (lane == 0) sbuf[nseq & 63] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
                nseq++;
                sumLL += ll;
                if ((nseq & 63) == 0) {
                    KC_WAVE_SYNC();
                    sq[nseq - 64 + lane] = sbuf[lane];
                    KC_WAVE_SYNC();
                }
            };
            // ---- RLE first (:143-150): the whole block is one byte ----
            bool rleBlk = false;
            if (srcLen > 3) {
                rleBlk = true;
                const uint8_t b0 = src[blkStart];
                for (int i0 = blkStart + 1; i0 < blkEnd && rleBlk; i0 += 64) {
                    const int i = i0 + lane;
                    const bool ne = i < blkEnd && src[i] != b0;
                    if (ballot64(ne) != 0) rleBlk = false;
                }
            }
            if (rleBlk) {
                emit(1, srcLen - 1 - 3, 1u + 3u);
                nextEmit = blkEnd;
            } else if (srcLen >= 16) {
                // ---- literal cost: ShannonEntropyBits(block) * 1024 / len, at least 1024 (:160-164) ----
                for (int i = lane; i < 256; i += 64) hist[i] = 0;
                if (lane == 0) shsum = 0;
                KC_WAVE_SYNC();
                for (int i = blkStart + lane * 8; i < blkEnd; i += 512) {
                    if (i + 8 <= blkEnd) {
                        const uint64_t v = ld64(src + i);
                        for (int k = 0; k < 8; k++) atomicAdd(&hist[(v >> (8 * k)) & 0xFF], 1u);
                    } else {
                        for (int k = i; k < blkEnd; k++) atomicAdd(&hist[src[k]], 1u);
                    }
                }
                KC_WAVE_SYNC();
                {
                    const double invTotal = 1.0 / (double)srcLen;
                    long long mine = 0;
                    for (int i = lane; i < 256; i += 64) { const uint32_t c = hist[i]; if (c > 0) mine += zb_shannon_term(c, invTotal); }
                    atomicAdd(&shsum, (unsigned long long)mine);
                }
                KC_WAVE_SYNC();
                const long long bits = (long long)shsum;
                int bpb = (int)((bits * 1024) / (long long)srcLen);
                if (bpb < 1024) bpb = 1024;
                diag = (uint32_t)bpb;

                const int sLimit = blkEnd - 12;  // inputMargin = 8 + 4
                // ---- per-lane candidate: everything of improve() that does not depend on the current best ----
                bool cv_ok = false;   // delta / first-bytes test passed
                int c_off = 0, c_s = 0, c_l = 0, c_bk = 0, c_rep = 0, c_e0 = 0, c_e1 = 0;
                auto est_of = [&](int length, int ofc) -> int {  // match.estBits before its `est > 0` clause
                    int e = cost[ofc] + cost[32 + (int)zb_ml_code((uint32_t)(length - 3))];
                    e -= (length * bpb) >> 10;
                    return e;
                };
                auto eval = [&](bool act, int offset, int sp, uint32_t first, int rep) {
                    cv_ok = false;
                    c_off = offset; c_s = sp; c_rep = rep; c_l = 0; c_bk = 0; c_e0 = 0; c_e1 = 0;
                    if (!act) return;
                    const int delta = sp - offset;
                    if (delta >= W || delta <= 0 || offset < 0) return;
                    if (ld32(src + offset) != first) return;
                    int a = sp + 4, t = offset + 4;  // l = 4 + matchlen(s+4, offset+4, src): bounded by the block's end
                    while (a < blkEnd) {
                        if (blkEnd - a < 8) {
                            if (src[a] == src[t]) { a++; t++; continue; }
                            break;
                        }
                        const uint64_t diff = ld64(src + a) ^ ld64(src + t);
                        if (diff != 0) { a += ctz64(diff) >> 3; break; }
                        a += 8; t += 8;
                    }
                    const int l = a - sp;
                    int bk = 0;  // the backward extension improve() applies when the current best is not a repeat
                    {
                        const int tMin = sp - W > 0 ? sp - W : 0;
                        while (offset - bk > tMin && sp - bk > nextEmit && src[offset - bk - 1] == src[sp - bk - 1] && l + bk < ZB_MAXML) bk++;
                    }
                    const int ofc = rep < 0 ? (int)high_bit((uint32_t)delta + 3u) : (int)high_bit((uint32_t)rep & 3u);
                    cv_ok = true;
                    c_l = l; c_bk = bk;
                    c_e0 = est_of(l, ofc);
                    c_e1 = bk ? est_of(l + bk, ofc) : c_e0;
                };
                ZbMatch best;
                // improve(&best, ...) for the candidates held by lanes [lo, hi), in lane order
                auto fold = [&](int lo, int hi, uint64_t enabled) {
                    int j = lo;
                    while (j < hi) {
                        bool ok = false;
                        ZbMatch cand;
                        cand.offset = 0; cand.s = 0; cand.length = 0; cand.rep = 0; cand.est = 0;
                        if (lane >= j && lane < hi && cv_ok && ((enabled >> lane) & 1ull)) {
                            bool rej = false;
                            if (best.length > 16) {  // quick reject against a long match (:215-230)
                                const int left = blkEnd - (best.s + best.length);
                                if (left <= 0) rej = true;
                                else {
                                    const int checkLen = best.length - (c_s - best.s) - 8;
                                    if (left > 2 && checkLen > 4 && ld32(src + c_off + checkLen) != ld32(src + c_s + checkLen)) rej = true;
                                }
                            }
                            if (!rej) {
                                const int bk = best.rep <= 0 ? c_bk : 0;
                                cand.offset = c_off - bk; cand.s = c_s - bk; cand.length = c_l + bk; cand.rep = c_rep;
                                cand.est = best.rep <= 0 ? c_e1 : c_e0;
                                if (cand.est > 0) { cand.length = 0; cand.est = ZB_HIGH; }
                                ok = best.est >= ZB_HIGH || cand.est - best.est + (((cand.s - best.s) * bpb) >> 10) < 0;
                            }
                        }
                        const uint64_t mask = ballot64(ok);
                        if (mask == 0) break;
                        const int k = ctz64(mask);
                        best.offset = (int)rdlane32((uint32_t)cand.offset, k); best.s = (int)rdlane32((uint32_t)cand.s, k);
                        best.length = (int)rdlane32((uint32_t)cand.length, k); best.rep = (int)rdlane32((uint32_t)cand.rep, k);
                        best.est = (int)rdlane32((uint32_t)cand.est, k);
                        j = k + 1;
                    }
                };

                uint32_t guard = 0;
                for (;;) {  // encodeLoop
                    if (++guard > 2u * (uint32_t)srcLen + 64u) break;  // every step advances s: cannot happen; never spin on the device
                    const bool canRepeat = nseq > 2;
                    const uint64_t cv = ld64(src + s);
                    const uint32_t hL = zb_hashL(cv), hS = zb_hashS(cv);
                    const uint64_t candL = lT[hL], candS = sT[hS];
                    {   // lanes 0-3: the table candidates at s; 4-6: repeats straight after a match; 7-9: repeats at s+1; 10-12: at s+3
                        int off = 0, sp = s, rep = -1; uint32_t first = (uint32_t)cv; bool act = lane < 4;
                        if (lane == 0) off = (int)((uint32_t)candL - (uint32_t)cur);
                        else if (lane == 1) off = (int)((uint32_t)(candL >> 32) - (uint32_t)cur);
                        else if (lane == 2) off = (int)((uint32_t)candS - (uint32_t)cur);
                        else if (lane == 3) off = (int)((uint32_t)(candS >> 32) - (uint32_t)cur);
                        else if (lane < 13 && canRepeat) {
                            act = true;
                            const int g = (lane - 4) / 3, r = (lane - 4) % 3;
                            if (g == 0) { off = r == 0 ? s - o2 : (r == 1 ? s - o3 : s - (o1 - 1)); rep = (r + 1) | 4; }
                            else {
                                sp = g == 1 ? s + 1 : s + 3;
                                first = g == 1 ? (uint32_t)(cv >> 8) : (uint32_t)(cv >> 24);
                                off = sp - (r == 0 ? o1 : (r == 1 ? o2 : o3));
                                rep = r + 1;
                            }
                        }
                        eval(act, off, sp, first, rep);
                    }
                    best.offset = 0; best.s = s; best.length = 0; best.rep = 0; best.est = ZB_HIGH;
                    fold(0, 4, ~0ull);
                    if (canRepeat && best.length < ZB_GOOD) {
                        if (s == nextEmit) fold(4, 7, o1 > 1 ? ~0ull : ~(1ull << 6));
                        if (best.rep <= 0) {
                            fold(7, 10, ~0ull);
                            if (best.rep < 0) fold(10, 13, ~0ull);
                        }
                    }
                    // the tables take s (:287-289), after every lane has read what this step looks up
                    KC_WAVE_SYNC();
                    if (lane == 0) {
                        lT[hL] = (uint64_t)(uint32_t)(s + cur) | (candL << 32);
                        sT[hS] = (uint64_t)(uint32_t)(s + cur) | (candS << 32);
                    }
                    KC_WAVE_SYNC();
                    int index0 = s + 1;
                    if (best.length < ZB_GOOD) {
                        if (best.length < 4) {  // no match: move forward (kSearchStrength 10)
                            s += 1 + ((s - nextEmit) >> 9);
                            if (s >= sLimit) break;
                            continue;
                        }
                        const uint64_t cS1 = sT[zb_hashS(cv >> 8)];
                        const uint64_t cv1 = ld64(src + s + 1), cv2 = ld64(src + s + 2);
                        const uint64_t cL1 = lT[zb_hashL(cv1)], cL2 = lT[zb_hashL(cv2)];
                        {   // short at s+1; long (both chain entries) at s+1 and s+2
                            int off = 0, sp = s + 1; uint32_t first = (uint32_t)cv1; const bool act = lane < 5;
                            if (lane == 0) off = (int)((uint32_t)cS1 - (uint32_t)cur);
                            else if (lane == 1) off = (int)((uint32_t)cL1 - (uint32_t)cur);
                            else if (lane == 2) off = (int)((uint32_t)(cL1 >> 32) - (uint32_t)cur);
                            else if (lane == 3) { off = (int)((uint32_t)cL2 - (uint32_t)cur); sp = s + 2; first = (uint32_t)cv2; }
                            else if (lane == 4) { off = (int)((uint32_t)(cL2 >> 32) - (uint32_t)cur); sp = s + 2; first = (uint32_t)cv2; }
                            eval(act, off, sp, first, -1);
                        }
                        fold(0, 5, ~0ull);
                        // where the current best ends: the offset that would continue it (:331-345; skipBeginning = 2)
                        if (best.s > s - 2) {
                            const int sAt = best.s + best.length;
                            if (sAt < sLimit) {
                                const uint64_t cE = lT[zb_hashL(ld64(src + sAt))];
                                const int off = (int)((uint32_t)cE - (uint32_t)cur) - best.length + 2;
                                if (off >= 0) {
                                    eval(lane == 0, off, best.s + 2, ld32(src + best.s + 2), -1);
                                    fold(0, 1, ~0ull);
                                    const int off2 = (int)((uint32_t)(cE >> 32) - (uint32_t)cur) - best.length + 2;
                                    if (off2 >= 0) {
                                        eval(lane == 0, off2, best.s + 2, ld32(src + best.s + 2), -1);
                                        fold(0, 1, ~0ull);
                                    }
                                }
                            }
                        }
                    }
                    // ---- a match (:364-455) ----
                    s = best.s;
                    int end;
                    if (best.rep > 0) {
                        emit(best.s - nextEmit, best.length - 3, (uint32_t)(best.rep & 3));
                        s = best.s + best.length;
                        nextEmit = s;
                        end = s < sLimit + 4 ? s : sLimit + 4;
                        {   // :397-404.  Written as selects: hipcc (ROCm 7.2) left the new offset3 undefined on the rep == 2|4 path of the
                            // equivalent if / else-if chain (found on the device: offset3 kept a stale register)
                            const int kind = best.rep == 1 ? 0 : ((best.rep == 2 || best.rep == 5) ? 1 : ((best.rep == 3 || best.rep == 6) ? 2 : 3));
                            const int a1 = o1, a2 = o2, a3 = o3;
                            o1 = kind == 0 ? a1 : (kind == 1 ? a2 : (kind == 2 ? a3 : a1 - 1));
                            o2 = kind == 0 ? a2 : a1;
                            o3 = kind <= 1 ? a3 : a2;
                        }
                    } else {
                        o3 = o2; o2 = o1; o1 = s - best.offset;
                        emit(s - nextEmit, best.length - 3, (uint32_t)(s - best.offset) + 3u);
                        s += best.length;
                        nextEmit = s;
                        end = s < sLimit - 4 ? s : sLimit - 4;
                    }
                    if (index0 < end) index_range(index0, end, end);
                    if (s >= sLimit) break;
                }
            }
            KC_WAVE_SYNC();
            if (lane < (nseq & 63)) sq[(nseq & ~63) + lane] = sbuf[lane];  // the buffered tail of the sequence list
            KC_WAVE_SYNC();
            const int extra = nextEmit < blkEnd ? blkEnd - nextEmit : 0;
            const int nlit = sumLL + extra;
            const bool rle = nseq == 1 && nlit <= 1 && (int)firstLL == nlit && firstOf - 3u == 1u;
            const int saved = srcLen - nlit - (srcLen >> 6);
            uint32_t flags = 0;
            if (nseq > 0 && !rle && saved < 16) flags |= KC_BF_POP_A;
            if (P.pop_blk != nullptr && P.pop_blk[blk0 + (uint32_t)b] != 0) flags |= KC_BF_FORCED;
            flags |= diag << 8;  // (diagnostics, like the other finders' round counts: the block's literal cost estimate)
            // No sequence of a block reads the repeat offsets before the block has three sequences of its own (canRepeat), and
            // three non-repeat sequences replace all three: what a block inherits never reaches its output, so a block re-emitted
            // raw (popOffsets) needs no re-run of its successors — o?_out is reported equal to o?_in.
            if (lane == 0) {
                KcBlkMeta m;
                m.nseq = (uint32_t)nseq;
                m.nlit = (uint32_t)nlit;
                m.extra_lits = (uint32_t)extra;
                m.flags = flags;
                m.o1_in = (uint32_t)o1_in; m.o2_in = (uint32_t)o2_in;
                m.o1_out = (uint32_t)o1_in; m.o2_out = (uint32_t)o2_in;
                P.meta[blk0 + (uint32_t)b] = m;
            }
        }
        cur64 += (int64_t)tlen;
    }
    if (lane == 0) slot_cur[slot] = (uint32_t)cur64;
}

void kc_launch_zbest_match(const KcMatchParams& P, uint64_t* tables, uint32_t* slot_cur, const int32_t* cost, uint32_t n_launch, uint32_t n_slots,
                           hipStream_t st) {
    if (n_launch == 0) return;
    const uint32_t g = n_launch < n_slots ? n_launch : n_slots;
    hipLaunchKernelGGL(kc_zbest_match_kernel, dim3(g), dim3(64), 0, st, P, tables, slot_cur, cost, n_launch, g);
}
