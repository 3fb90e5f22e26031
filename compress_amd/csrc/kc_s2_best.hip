// kc_s2_best.hip — s2.EncodeBest / s2.EncodeSnappyBest on the device: 16 lanes per block, four blocks per wave (round 6; one wave per
// block before: at most 16 lanes of it ever had a candidate to score, and the 28 waves a CU holds were 28 blocks in flight on a kernel
// that is three dependent round trips per input position — now 112).
//
// Replaces encodeBlockBest (s2/encode_best.go:22-455, dict == nil) and encodeBlockBestSnappy (:457-710) with their size
// estimates (:715-797), behind s2.EncodeBest / s2.EncodeSnappyBest (s2/encode.go:146-202, 278-305).  The best level keeps, per
// block, a long (8-byte hash, 2^19) and a short (4-byte hash, 2^16) table of {cur, prev} position pairs — 4.5 MiB, in an HBM
// arena — scores up to five candidates at a position and, once one matches, up to eleven more at s+1, s+2 and behind the end of
// the best match, and indexes every position of every match.
//
// Mapping: the control flow is uniform within a block's lane group (the sequential algorithm; the groups of a wave diverge freely);
// the lanes are used where
// the reference has independent work: every candidate of a phase is evaluated by its own lane (4-byte check, match extension,
// score), the winner is then folded in the reference's order — including its "same offset as the current best: not retested"
// rule, which is what makes the fold order-dependent —, literals are copied 8 bytes per lane, and the table updates after a
// match take one position per lane and pass (positions whose bucket an earlier position of the pass also hits are applied afterwards, in
// order: the {cur, prev} chain is order-dependent).  Tables are read and written with plain loads / stores: a wave's accesses to
// one address are performed in program order.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_s2_dev.h"
#include "kc_wave.h"

#ifndef SBG
#define SBG 16          // lanes per block (64: the round-3 form, one wave per block — measurement builds)
#endif
static_assert(SBG == 16 || SBG == 64, "phase B holds nine candidates: at least 16 lanes per block; the duplicate test below is written for 16 and 64");
#define SB_GMASK (SBG == 64 ? ~0ull : ((1ull << (SBG & 63)) - 1ull))
__device__ __forceinline__ uint64_t sb_ballot(bool p, int grp) { return (ballot64(p) >> (grp * SBG)) & SB_GMASK; }
__device__ __forceinline__ uint32_t sb_bcast32(uint32_t v, int grp, int k) { return (uint32_t)__shfl((int)v, grp * SBG + k, 64); }
// Source window (round 6, as in the zstd match finders): the block's bytes around the parse position in a 512-byte ring per block in
// LDS, refilled 16 bytes per lane a pass ahead of use: the 8 bytes at s, s+1, s+2, behind and at the end of the best match, the forward
// side of the match extensions and the positions indexed behind a match are LDS reads instead of the first of a step's three dependent
// round trips (rd64 / rd32 fall back to memory for what the window does not hold).
#ifndef SB_RING
#define SB_RING 1
#endif
#ifndef SB_PRE
#define SB_PRE 0        // 1: the buckets of s+1 looked up a pass ahead — measured 3 % slower (profiles/r06_ab_kernels.txt), kept for A/B
#endif
#define SB_RB 512
#define SB_MIRROR 32
#define SB_STRIDE (SB_RB + SB_MIRROR)
#define SB_AHEAD 224
#define SB_LBITS 19
#define SB_SBITS 16

__device__ __forceinline__ uint32_t sb_hash8(uint64_t u) { return (uint32_t)((u * KC_PRIME8) >> (64 - SB_LBITS)); }
__device__ __forceinline__ uint32_t sb_hash4(uint64_t u) { return ((uint32_t)u * KC_PRIME4) >> (32 - SB_SBITS); }

// size estimates of the best level (encode_best.go:715-797): NOT the sizes emitCopy / emitRepeat produce in every case
__device__ inline int sb_repeat_size(int offset, int length) {
    int total = 0;
    for (;;) {
        if (length <= 4 + 4 || (length < 8 + 4 && offset < 2048)) return total + 2;
        if (length < (1 << 8) + 4 + 4) return total + 3;
        if (length < (1 << 16) + (1 << 8) + 4) return total + 4;
        const int maxRepeat = (1 << 24) - 1;
        length -= (1 << 16) - 4;
        int left = 0;
        if (length > maxRepeat) left = length - maxRepeat + 4;
        total += 5;
        if (left <= 0) return total;
        length = left;
    }
}
__device__ inline int sb_copy_size(int offset, int length) {
    if (offset >= 65536) {
        int i = 0;
        if (length > 64) {
            length -= 64;
            if (length >= 4) return 5 + sb_repeat_size(offset, length);
            i = 5;
        }
        if (length == 0) return i;
        return i + 5;
    }
    if (length > 64) {
        if (offset < 2048) return 2 + sb_repeat_size(offset, length - 8);
        return 3 + sb_repeat_size(offset, length - 60);
    }
    if (length >= 12 || offset >= 2048) return 3;
    return 2;
}
__device__ inline int sb_copy_nr_size(int offset, int length) {  // emitCopyNoRepeatSize (:757-771)
    if (offset >= 65536) return 5 + 5 * (length / 64);
    if (length > 64) return 3 + 3 * (length / 60);
    if (length >= 12 || offset >= 2048) return 3;
    return 2;
}

struct SbMatch { int offset, s, length, score, rep; };

// Six waves per SIMD (80 VGPRs and 36-60 bytes of scratch instead of 109 and none): 24 576 blocks resident instead of 16 384.  A block
// takes ~170-200 ms whatever shares the chip with it, so the rate is the blocks in flight: at 1.5 GiB of 64 KiB blocks 7.9 GB/s against
// 5.5 at four waves (at 1 GiB, one residency either way, 6.3 against 6.5); seven waves lose to the spills and to the LDS ring's share of
// a CU (gpurun_out/r7b, r7c).
#ifndef SB_WPE
#define SB_WPE 6
#endif
#define SB_KATTR __attribute__((amdgpu_waves_per_eu(SB_WPE, SB_WPE)))
template <bool SNAPPY>
__global__ __launch_bounds__(64) SB_KATTR void kc_s2_best_kernel(KcS2Params P) {
    const int wl = (int)threadIdx.x;                 // lane of the wave
    const int lane = wl % SBG, grp = wl / SBG;       // lane of the block's group, group of the wave
    const uint32_t bi = blockIdx.x * (64 / SBG) + (uint32_t)grp;
    __shared__ uint32_t crcT[4][256];
    if (P.framed) {  // s2.Writer chunks carry the masked CRC32C of the uncompressed block (slice-by-4 tables, as kc_s2_encode_kernel)
        for (int i = wl; i < 256; i += 64) {
            uint32_t c = (uint32_t)i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            crcT[0][i] = c;
        }
        __syncthreads();  // (whole wave: before any group leaves)
        for (int i = wl; i < 256; i += 64) {
            uint32_t c = crcT[0][i];
            for (int t = 1; t < 4; t++) { c = crcT[0][c & 0xFF] ^ (c >> 8); crcT[t][i] = c; }
        }
        __syncthreads();
    }
    if (bi >= P.n_blocks) return;  // a group without a block (the launch's tail): the whole group leaves
    const uint8_t* __restrict__ src = P.src + P.blk_off[bi];
    const int len = (int)(P.blk_off[bi + 1] - P.blk_off[bi]);
    uint8_t* __restrict__ slot = P.stage + P.stage_off[bi];
    uint8_t* __restrict__ out = slot + (P.framed ? 8 : 0);  // chunk header (type, len24, crc) goes in front
    uint64_t* const lT = (uint64_t*)(P.tables + (size_t)bi * P.table_stride);  // zeroed by the host
    uint64_t* const sT = lT + ((size_t)1 << SB_LBITS);

    // uvarint(len) header (encode.go:161-176)
    int hdr = 0;
    {
        uint64_t x = (uint64_t)len;
        while (x >= 0x80) { if (lane == 0) out[hdr] = (uint8_t)x | 0x80; hdr++; x >>= 7; }
        if (lane == 0) out[hdr] = (uint8_t)x;
        hdr++;
    }
    uint8_t* __restrict__ dst = out + hdr;
    int d = 0;
    if (len == 0 && !P.framed) { if (lane == 0) P.out_size[bi] = (uint32_t)hdr; return; }
    bool stored = len < 32;  // minNonLiteralBlockSize

    __shared__ __attribute__((aligned(16))) uint8_t ring_all[SB_RING ? (64 / SBG) * SB_STRIDE : 16];
    uint8_t* const ring = ring_all + (SB_RING ? grp * SB_STRIDE : 0);
    const int boff = (int)((uintptr_t)src & 15);
    const uint8_t* __restrict__ abase = src - boff;
    int wlo = 0, whi = 0;   // the ring holds the bytes abase[wlo .. whi)
    bool pend = false;      // rf holds the 16 * SBG bytes abase[whi ..) loaded a pass ago
    uint4 rf = make_uint4(0, 0, 0, 0);
    auto window = [&](int sp) {  // top of every pass (group-uniform)
        if (!SB_RING) return;
        if (pend) {
            const int ro = (whi + 16 * lane) & (SB_RB - 1);
            if (lane < 16) {  // (SBG == 64: only the first 16 lanes carry a refill)
                *(uint4*)(ring + ro) = rf;
                if (ro < SB_MIRROR) *(uint4*)(ring + SB_RB + ro) = rf;
            }
            whi += 256;
            if (whi - wlo > SB_RB) wlo = whi - SB_RB;
            pend = false;
        }
        KC_WAVE_SYNC();
        const int sa = sp + boff;
        if (sa >= whi || sa < wlo) { const int w0 = sa & ~15; wlo = whi = w0; }
        if (whi - sa < SB_AHEAD) {
            const uint8_t* q = abase + whi + 16 * lane;
            rf = make_uint4(0, 0, 0, 0);
            if (lane < 16 && q < src + len) rf = *(const uint4*)q;  // aligned: never leaves the 16-byte granule of a readable byte
            pend = true;
        }
    };
    auto rd64 = [&](int pos) -> uint64_t {
        const int a = pos + boff, a4 = a & ~3;
        if (SB_RING && a4 >= wlo && a4 + 12 <= whi) {
            const uint32_t* r = (const uint32_t*)(ring + (a4 & (SB_RB - 1)));
            const uint32_t r0 = r[0], r1 = r[1], r2 = r[2];
            const uint32_t sh = (uint32_t)(a & 3);
            return (uint64_t)__builtin_amdgcn_alignbyte(r1, r0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(r2, r1, sh) << 32);
        }
        return ld64(src + pos);
    };
    auto rd32 = [&](int pos) -> uint32_t {
        const int a = pos + boff, a4 = a & ~3;
        if (SB_RING && a4 >= wlo && a4 + 8 <= whi) {
            const uint32_t* r = (const uint32_t*)(ring + (a4 & (SB_RB - 1)));
            return __builtin_amdgcn_alignbyte(r[1], r[0], (uint32_t)(a & 3));
        }
        return ld32(src + pos);
    };
    auto emit_lit = [&](int from, int n) -> int {  // emitLiteral (encode_go.go:80)
        if (n == 0) return 0;
        const uint32_t m = (uint32_t)(n - 1);
        uint8_t* __restrict__ o = dst + d;
        int i;
        if (m < 60) { i = 1; if (lane == 0) o[0] = (uint8_t)(m << 2); }
        else if (m < (1u << 8)) { i = 2; if (lane == 0) { o[0] = 60 << 2; o[1] = (uint8_t)m; } }
        else if (m < (1u << 16)) { i = 3; if (lane == 0) { o[0] = 61 << 2; o[1] = (uint8_t)m; o[2] = (uint8_t)(m >> 8); } }
        else if (m < (1u << 24)) { i = 4; if (lane == 0) { o[0] = 62 << 2; o[1] = (uint8_t)m; o[2] = (uint8_t)(m >> 8); o[3] = (uint8_t)(m >> 16); } }
        else { i = 5; if (lane == 0) { o[0] = 63 << 2; o[1] = (uint8_t)m; o[2] = (uint8_t)(m >> 8); o[3] = (uint8_t)(m >> 16); o[4] = (uint8_t)(m >> 24); } }
        const int body = n & ~7;
        for (int k = lane * 8; k < body; k += 8 * SBG) st64(o + i + k, ld64(src + from + k));
        for (int k = body + lane; k < n; k += SBG) o[i + k] = src[from + k];
        return i + n;
    };

    if (!stored) {
        const int sLimit = len - (8 + 2);  // inputMarginBest
        const int dstLimit = len - 5;
        int nextEmit = 0, s = 1, repeat = 1;
        bool fin = false;
        // one candidate per lane: the match matchAt would return WITHOUT its same-offset shortcut (applied by the fold below)
        auto eval = [&](bool act, int offset, int s_, uint32_t first, bool rep) -> SbMatch {
            SbMatch m;
            m.offset = offset; m.s = s_; m.length = 0; m.score = 0; m.rep = rep ? 1 : 0;
            if (!act) return m;
            if (rd32(offset) != first) return m;
            int la = 4 + offset;  // m.length while it is an absolute position
            int sp = s_ + 4;
            if (SNAPPY) {
                while (sp <= sLimit) {
                    const uint64_t diff = rd64(sp) ^ rd64(la);
                    if (diff != 0) { la += ctz64(diff) >> 3; break; }
                    sp += 8; la += 8;
                }
            } else {
                while (sp < len) {
                    if (len - sp < 8) {
                        if (src[sp] == src[la]) { la++; sp++; continue; }
                        break;
                    }
                    const uint64_t diff = rd64(sp) ^ rd64(la);
                    if (diff != 0) { la += ctz64(diff) >> 3; break; }
                    sp += 8; la += 8;
                }
            }
            m.length = la - offset;
            int sc = m.length - m.s;     // matches that start later are penalised: the bytes before go out as literals
            if (nextEmit == m.s) sc++;   // no literals to emit: one byte saved
            const int off = m.s - m.offset;
            if (SNAPPY) sc -= sb_copy_nr_size(off, m.length);
            else if (rep) sc -= sb_repeat_size(off, m.length);
            else sc -= sb_copy_size(off, m.length);
            m.score = sc;
            if (m.score <= -m.s) m.length = 0;  // no savings: maybe a better one turns up
            return m;
        };
        // best = bestOf(best, matchAt(candidate)) over the candidates held by lanes [lo, hi), in lane order, with matchAt's shortcut
        // (same offset as the current best: not retested; not applied to the lanes of `nosame`: bestOf(matchAt(cur L), matchAt(prev L))
        // evaluates both against the empty best of the step's start).  Every remaining lane tests its candidate against the current
        // best; the first lane that would replace it does, and the rest test again: one ballot per replacement instead of five
        // cross-lane reads per candidate.
        auto fold = [&](SbMatch& best, const SbMatch& mine, int lo, int hi, uint64_t enabled, uint64_t nosame) {
            int j = lo;
            while (j < hi) {
                bool ok = false;
                if (lane >= j && lane < hi && mine.length != 0 && ((enabled >> lane) & 1ull)) {
                    if (best.length == 0) ok = true;
                    else {
                        const bool same = best.s - best.offset == mine.s - mine.offset && !((nosame >> lane) & 1ull);
                        ok = !same && best.score + mine.s < mine.score + best.s;  // bestOf keeps a when a.score + b.s >= b.score + a.s
                    }
                }
                const uint64_t mask = sb_ballot(ok, grp);
                if (mask == 0) break;
                const int k = ctz64(mask);
                best.offset = (int)sb_bcast32((uint32_t)mine.offset, grp, k); best.s = (int)sb_bcast32((uint32_t)mine.s, grp, k);
                best.length = (int)sb_bcast32((uint32_t)mine.length, grp, k); best.score = (int)sb_bcast32((uint32_t)mine.score, grp, k);
                best.rep = (int)sb_bcast32((uint32_t)mine.rep, grp, k);
                j = k + 1;
            }
        };
        // The two buckets of s+1 are looked up a pass ahead (round 6): phase B of this pass needs them if this pass finds a match, the
        // next pass needs them if it does not (nextS == s + 1 while fewer than 256 literals are pending) — no lookup is wasted, and the
        // next pass starts its candidate loads without waiting for its own table round trip.  What this pass writes into the same bucket
        // is patched into the registers; anything else that writes the tables (the index pass behind a match) drops them.
        int prePos = -1;
        uint64_t preL = 0, preS = 0;
        uint32_t guard = 0;
        // ONE loop for the scan and for what follows a match (round 6): every pass, every group of the wave takes one probe step; the
        // groups that found a match then emit and index it while the others wait for that — not, as with the reference's nested loops
        // taken literally, every group waiting until ALL groups of the wave have found theirs.
        while (!fin && !stored) {
            SbMatch best;
            best.offset = 0; best.s = 0; best.length = 0; best.score = 0; best.rep = 0;
            {
                if (++guard > 2u * (uint32_t)len + 64u) { stored = true; break; }  // every step advances s: cannot happen; never spin on the device
                int nextS = ((s - nextEmit) >> 8) + 1;
                if (nextS > 64) nextS = s + 64; else nextS += s;  // maxSkip
                if (nextS > sLimit) { fin = true; break; }
                window(s);
                const uint64_t cv = rd64(s);
                const uint32_t hashL = sb_hash8(cv), hashS = sb_hash4(cv);
                uint64_t candidateL, candidateS;
                if (SB_PRE && prePos == s) { candidateL = preL; candidateS = preS; }
                else { candidateL = lT[hashL]; candidateS = sT[hashS]; }
                const int s1 = s + 1;
                const uint64_t cv1 = rd64(s1);
                const uint32_t hashL1 = sb_hash8(cv1), hashS1 = sb_hash4(cv1);  // (sb_hash4(cv >> 8) == hashS1: the same four bytes)
                uint64_t nextLong1 = 0, nextShort1 = 0;
                if (SB_PRE) { nextLong1 = lT[hashL1]; nextShort1 = sT[hashS1]; }
                // ---- phase A: the four table candidates at s, the repeat at s+1 (:232-250) ----
                {
                    int off = 0, sp = s; uint32_t first = (uint32_t)cv; bool rep = false, act = lane < 5;
                    if (lane == 0) off = (int)(uint32_t)candidateL;
                    else if (lane == 1) off = (int)(candidateL >> 32);
                    else if (lane == 2) off = (int)(uint32_t)candidateS;
                    else if (lane == 3) off = (int)(candidateS >> 32);
                    else if (lane == 4) { off = s - repeat + 1; sp = s + 1; first = (uint32_t)(cv >> 8); rep = !SNAPPY; act = SNAPPY || repeat > 0; }
                    const SbMatch mine = eval(act, off, sp, first, rep);
                    fold(best, mine, 0, 5, (SNAPPY || repeat > 0) ? ~0ull : ~(1ull << 4), 1ull << 1);
                }
                if (best.length > 0) {
                    // ---- phase B: s+1 and s+2 (:252-311) ----
                    if (!SB_PRE) { nextShort1 = sT[hashS1]; nextLong1 = lT[hashL1]; }
                    const uint64_t nextShort2 = sT[sb_hash4(cv1 >> 8)];
                    const int s2 = s + 2;
                    const uint64_t cv2 = rd64(s2);
                    const uint64_t nextLong2 = lT[sb_hash8(cv2)];
                    {
                        int off = 0, sp = s1; uint32_t first = (uint32_t)cv1; bool rep = false, act = lane < 10;
                        // lanes 0-3: s+1 table candidates; lane 4: the repeat of the s+1 / s+2 state; lanes 5-8: s+2 table candidates
                        if (lane == 0) off = (int)(uint32_t)nextShort1;
                        else if (lane == 1) off = (int)(nextShort1 >> 32);
                        else if (lane == 2) off = (int)(uint32_t)nextLong1;
                        else if (lane == 3) off = (int)(nextLong1 >> 32);
                        else if (lane == 4) {
                            if (SNAPPY) { off = s1 - repeat + 1; sp = s1 + 1; first = (uint32_t)(cv1 >> 8); }    // repeat at +2, from the s+1 state (:571)
                            else { off = s2 - repeat; sp = s2; first = (uint32_t)cv2; rep = true; act = repeat > 0; }  // repeat at +2 (:286-289)
                        }
                        else if (lane == 5) { off = (int)(uint32_t)nextShort2; sp = s2; first = (uint32_t)cv2; }
                        else if (lane == 6) { off = (int)(nextShort2 >> 32); sp = s2; first = (uint32_t)cv2; }
                        else if (lane == 7) { off = (int)(uint32_t)nextLong2; sp = s2; first = (uint32_t)cv2; }
                        else if (lane == 8) { off = (int)(nextLong2 >> 32); sp = s2; first = (uint32_t)cv2; }
                        else act = false;
                        const SbMatch mine = eval(act, off, sp, first, rep);
                        // (both variants take the repeat before the s+2 table candidates)
                        fold(best, mine, 0, 9, (SNAPPY || repeat > 0) ? ~0ull : ~(1ull << 4), 0ull);
                    }
                    // ---- phase C: a match at the end of the best match, shifted back over it (:313-345) ----
                    const int skipBeginning = SNAPPY ? 0 : 2, skipEnd = SNAPPY ? 0 : 1;
                    const int sAt = best.s + best.length - skipEnd;
                    if (sAt < sLimit) {
                        const int sBack = best.s + skipBeginning - skipEnd;
                        const int backL = best.length - skipBeginning;
                        const uint64_t cvb = rd64(sBack);
                        const uint64_t next = lT[sb_hash8(rd64(sAt))];
                        const int chk0 = (int)(uint32_t)next - backL, chk1 = (int)(next >> 32) - backL;
                        const bool act = (lane == 0 && chk0 > 0) || (lane == 1 && chk1 > 0);
                        const SbMatch mine = eval(act, lane == 0 ? chk0 : chk1, sBack, (uint32_t)cvb, false);
                        fold(best, mine, 0, 2, (chk0 > 0 ? 1ull : 0ull) | (chk1 > 0 ? 2ull : 0ull), 0ull);
                    }
                }
                // update the tables (:348-350), after every lane has read what this step looks up
                KC_WAVE_SYNC();
                if (lane == 0) {
                    lT[hashL] = (uint64_t)(uint32_t)s | (candidateL << 32);
                    sT[hashS] = (uint64_t)(uint32_t)s | (candidateS << 32);
                }
                KC_WAVE_SYNC();
                if (SB_PRE) {  // the entries of s+1 as the tables hold them now
                    if (hashL1 == hashL) nextLong1 = (uint64_t)(uint32_t)s | (candidateL << 32);
                    if (hashS1 == hashS) nextShort1 = (uint64_t)(uint32_t)s | (candidateS << 32);
                    prePos = s1; preL = nextLong1; preS = nextShort1;
                }
                if (best.length == 0) { s = nextS; continue; }
            }
            // ---- the match: extend backwards (not for repeats; always at the Snappy level), bail-outs, emit (:358-420) ----
            s = best.s;
            if (SNAPPY || !best.rep) {
                int kmax = best.offset < s - nextEmit ? best.offset : s - nextEmit;
                int cnt = 0;
                while (cnt < kmax) {
                    const int k = cnt + lane + 1;
                    bool ne = true;
                    if (k <= kmax) ne = src[best.offset - k] != src[s - k];
                    const uint64_t mm = sb_ballot(ne, grp);
                    const int c = mm ? ctz64(mm) : SBG;
                    cnt += c;
                    if (c < SBG) break;
                }
                if (cnt > kmax) cnt = kmax;
                best.offset -= cnt; best.length += cnt; s -= cnt;
            }
            if (d + (s - nextEmit) > dstLimit) { stored = true; break; }
            const int base = s;
            const int offset = s - best.offset;
            s += best.length;
            if (offset > 65535 && s - base <= 5 && (SNAPPY || !best.rep)) {  // equal or worse than the encoding
                s = best.s + 1;
                if (s >= sLimit) fin = true;
                continue;
            }
            d += emit_lit(nextEmit, base - nextEmit);
            if (SNAPPY) {
                if (lane == 0) s2_emit_copy_nr1(dst + d, offset, best.length);
                d += s2_copy_nr_size(offset, best.length);
            } else {
                uint64_t lo = 0, hi = 0;
                auto sink = [&](int k, uint8_t v) { if (k < 8) lo |= (uint64_t)v << (8 * k); else hi |= (uint64_t)v << (8 * (k - 8)); };
                const bool asRepeat = best.rep && nextEmit > 0;  // the first match cannot be a repeat
                const int n = asRepeat ? s2_put_repeat(sink, offset, best.length) : s2_put_copy(sink, offset, best.length);
                if (lane == 0) {
                    for (int k = 0; k < n && k < 8; k++) dst[d + k] = (uint8_t)(lo >> (8 * k));
                    for (int k = 8; k < n; k++) dst[d + k] = (uint8_t)(hi >> (8 * (k - 8)));
                }
                d += n;
            }
            repeat = offset;
            nextEmit = s;
            if (s >= sLimit) { fin = true; break; }
            if (d > dstLimit) { stored = true; break; }
            // ---- index every position of the match (:422-432): one position per lane and pass, chain order kept ----
            for (int i0 = best.s + 1; i0 < s; i0 += SBG) {
                const int i = i0 + lane;
                const bool act = i < s;
                uint32_t hl = 0xFFFFFFFFu, hs = 0xFFFFFFFFu;
                if (act) { const uint64_t cv0 = rd64(i); hl = sb_hash8(cv0); hs = sb_hash4(cv0); }
                // does an earlier position of this pass hit the same bucket?
                bool dupL = false, dupS = false;
#if SBG == 16
                // a group is one DPP row: row_shr:d hands every lane the value d lanes below it in its own group (0 where there is none)
#define SB_DUP_STEP(dd) do { const uint32_t kl = kc_dpp_or0<0x110 + (dd), 0xf>(hl), ks = kc_dpp_or0<0x110 + (dd), 0xf>(hs); \
                             if (lane >= (dd)) { dupL = dupL || kl == hl; dupS = dupS || ks == hs; } } while (0)
                SB_DUP_STEP(1); SB_DUP_STEP(2); SB_DUP_STEP(3); SB_DUP_STEP(4); SB_DUP_STEP(5); SB_DUP_STEP(6); SB_DUP_STEP(7); SB_DUP_STEP(8);
                SB_DUP_STEP(9); SB_DUP_STEP(10); SB_DUP_STEP(11); SB_DUP_STEP(12); SB_DUP_STEP(13); SB_DUP_STEP(14); SB_DUP_STEP(15);
#undef SB_DUP_STEP
#else
                const int npass = s - i0 < 64 ? s - i0 : 64;
                for (int k = 0; k + 1 < npass; k++) {
                    const uint32_t kl = rdlane32(hl, k), ks = rdlane32(hs, k);
                    if (k < lane) { dupL = dupL || kl == hl; dupS = dupS || ks == hs; }
                }
#endif
                if (act && !dupL) { const uint64_t old = lT[hl]; lT[hl] = (uint64_t)(uint32_t)i | (old << 32); }
                if (act && !dupS) { const uint64_t old = sT[hs]; sT[hs] = (uint64_t)(uint32_t)i | (old << 32); }
                uint64_t mL = sb_ballot(act && dupL, grp), mS = sb_ballot(act && dupS, grp);
                while (mL) {  // in order: each sees the entry its predecessors of the pass left
                    const int k = ctz64(mL);
                    mL &= mL - 1;
                    KC_WAVE_SYNC();
                    if (lane == k) { const uint64_t old = lT[hl]; lT[hl] = (uint64_t)(uint32_t)i | (old << 32); }
                }
                while (mS) {
                    const int k = ctz64(mS);
                    mS &= mS - 1;
                    KC_WAVE_SYNC();
                    if (lane == k) { const uint64_t old = sT[hs]; sT[hs] = (uint64_t)(uint32_t)i | (old << 32); }
                }
                KC_WAVE_SYNC();
            }
            prePos = -1;  // the index pass rewrote buckets
        }
        if (!stored) {  // emitRemainder (:435-450)
            if (nextEmit < len) {
                if (d + len - nextEmit > dstLimit) stored = true;
                else d += emit_lit(nextEmit, len - nextEmit);
            }
        }
    }
    if (stored) {
#ifndef KC_HIPEMU
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
        KC_WAVE_SYNC();
    }
    if (!P.framed) {
        if (stored) { d = 0; d = emit_lit(0, len); }  // encode.go:170-176: not compressible -> one literal
        if (lane == 0) P.out_size[bi] = (uint32_t)(hdr + d);
        return;
    }
    // ---- s2.Writer chunk (s2/writer.go:414-451): encodeBlock returned 0 -> uncompressed chunk, the raw bytes ----
    uint32_t chunkLen;
    uint8_t chunkType;
    if (stored) {
        for (int k = lane; k < len; k += SBG) out[k] = src[k];
        chunkType = 0x01;
        chunkLen = 4u + (uint32_t)len;
    } else {
        chunkType = 0x00;
        chunkLen = 4u + (uint32_t)hdr + (uint32_t)d;
    }
    if (lane == 0) {
        const uint32_t c = s2_crc32c(src, len, crcT);
        const uint32_t checksum = ((c >> 15) | (c << 17)) + 0xa282ead8u;
        slot[0] = chunkType;
        slot[1] = (uint8_t)chunkLen; slot[2] = (uint8_t)(chunkLen >> 8); slot[3] = (uint8_t)(chunkLen >> 16);
        slot[4] = (uint8_t)checksum; slot[5] = (uint8_t)(checksum >> 8); slot[6] = (uint8_t)(checksum >> 16); slot[7] = (uint8_t)(checksum >> 24);
        P.out_size[bi] = 4u + chunkLen;
    }
}

void kc_launch_s2_best(const KcS2Params& P, hipStream_t st) {
    if (P.n_blocks == 0) return;
    const uint32_t grid = (P.n_blocks + (64 / SBG) - 1) / (64 / SBG);
    if (P.level == 5) hipLaunchKernelGGL((kc_s2_best_kernel<true>), dim3(grid), dim3(64), 0, st, P);
    else hipLaunchKernelGGL((kc_s2_best_kernel<false>), dim3(grid), dim3(64), 0, st, P);
}
