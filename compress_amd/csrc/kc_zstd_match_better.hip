// kc_zstd_match_better.hip — SpeedBetterCompression match finder for gfx950.
//
// Replaces betterFastEncoder.Encode (zstd/enc_better.go:56-568; EncodeNoHist == ensureHist + Encode,
// :573-576) and betterFastEncoderDict.Encode (:579-1091).  16 lanes per unit, 4 units per wave; tables in
// an HBM arena per unit: long table 2^19 x {offset, prev} (8-byte hash, chain of length 2) followed by the
// short table 2^13 x u32 (5-byte hash).  Every stored position is (pos+1) | tag(4 source bytes) << PB, so
// candidates that the reference would reject on its 8-byte / 4-byte compare are mostly rejected on the tag
// without touching their (random) source line.
// The probe loop runs speculative rounds with ordered commit like the other match finders (round 1 probed one position per
// round: every literal position cost two dependent round trips, and with the 8192 units of a 1 GiB batch the kernel was
// latency-bound at 0.27 TB/s of HBM traffic); the group's lanes also cooperate on match extension (8 B per lane) and on the
// dense re-indexing of every second byte of a match, where lanes that hit the same long-table bucket are chained in order
// exactly as the sequential loop would chain them.
// Reproduced literally: RLE pre-check (non-dict), repeat at s+1, long / prev-long / short priority, the lazy
// long lookup at s+1 after a short match (with its table write), the end-of-match re-search with
// skipBeginning = 3 (non-dict) or 0 (dict), maxMatchLength caps, offset-2 loop, canRepeat snapshot.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_wave.h"

#define ZB_LONG_BITS 19
#define ZB_SHORT_BITS 13
#define ZB_MAX_MATCH_LENGTH 131074
#ifndef ZBG
#define ZBG 16  // lanes per unit: 16-wide speculation, 4 units per wave (ms per GiB of C5: 8 lanes 80.4, 16 lanes 61.1, 32 lanes 86.5)
#endif
static_assert(ZBG <= 32, "group ballots are 32-bit");

// Epoch stamps (round 3).  The tables are 4 MiB per unit: zeroing them — or, with a dictionary, copying the dictionary's tables
// into them — for every batch is 33 GB of writes per GiB of input (7.7 ms of a 68 ms step).  With P.epoch != 0 the top ZB_EPOCH_BITS
// bits of a long entry's `offset` word and of a short entry carry the stamp of the launch that wrote it: an entry with another
// stamp is one left by an earlier launch and reads as the dictionary's entry for that bucket (a shared, read-only table that
// stays in L2) or as empty (round 6 also tried a 64 KiB map of the buckets the dictionary fills in front of the shared table: slower
// still, profiles/r06_ab_kernels.txt).  The host advances the stamp per launch and clears the arena only when it wraps, when the position
// width changes or when the arena was used by something else.  `prev` words are stored resolved and unstamped: they are only read
// together with a valid `offset` word.  P.epoch == 0 is the old contract (tables fully initialised by the host): jobs, whose
// tables the host primes per unit, and units too long for the stamp to fit.
// Source window (round 6, the SpeedFastest / SpeedDefault kernels' ring): the unit's bytes around the parse position live in a per-unit
// ring in LDS, refilled 256 bytes at a time by the group's 16 lanes (one aligned 16-byte load each, one round ahead of use).  A probe
// round was three dependent round trips (8 source bytes at the probe position -> table entries -> candidate bytes) on a kernel
// that is latency-bound at the 8192 units of a 1 GiB batch (2 waves per SIMD); the first now is an LDS read, and so are the source
// reads of the re-indexing loop behind a match, the lazy lookup at s+1, the end-of-match re-search and the offset-2 loop whenever the
// window holds them (rd64 / rd32 fall back to memory where it does not: behind a match longer than the look-ahead).
#ifndef ZB_RB
#define ZB_RB 1024      // ring bytes per unit (power of two)
#endif
#define ZB_MIRROR 32    // the first 32 ring bytes again behind the ring: 12-byte reads never wrap
#define ZB_STRIDE (ZB_RB + ZB_MIRROR)
#ifndef ZB_AHEAD
#define ZB_AHEAD 384    // refill while fewer than this many bytes are buffered ahead of s
#endif
#ifndef ZB_RING
#define ZB_RING 1       // 0: measurement builds (KC_EXTRA_FLAGS=-DZB_RING=0): every source read from memory, as before round 6
#endif
#ifndef ZB_FUSE
#define ZB_FUSE 1       // 0: measurement builds: the candidate loads of a round / of the lazy lookup / of the re-search one dependent trip each, as before
#endif
#define ZB_EPOCH_BITS 4
#define ZB_EPOCH_SHIFT (32 - ZB_EPOCH_BITS)

struct ZbCtx {
    const uint8_t* base;   // hist: (dict ||) unit
    uint2* ltab;           // {offset, prev}
    uint32_t* stab;
    const uint2* pl;       // dictionary tables (epoch mode with a dictionary) or null
    const uint32_t* ps;
    uint32_t ep;           // this launch's stamp, 0 = none
    int PB, TB;
    uint32_t posMask;
    int mmo;
    __device__ __forceinline__ uint2 rdL(uint32_t h) const {
        uint2 e = ltab[h];
        if (ep == 0u) return e;
        const uint2 p = pl != nullptr ? pl[h] : make_uint2(0u, 0u);  // (issued beside the own load, not after it)
        if ((e.x >> ZB_EPOCH_SHIFT) == ep) { e.x &= (1u << ZB_EPOCH_SHIFT) - 1u; return e; }
        return p;
    }
    __device__ __forceinline__ uint32_t rdLx(uint32_t h) const {
        const uint32_t e = ltab[h].x;
        if (ep == 0u) return e;
        const uint32_t p = pl != nullptr ? pl[h].x : 0u;
        return (e >> ZB_EPOCH_SHIFT) == ep ? (e & ((1u << ZB_EPOCH_SHIFT) - 1u)) : p;
    }
    __device__ __forceinline__ uint32_t rdS(uint32_t h) const {
        const uint32_t e = stab[h];
        if (ep == 0u) return e;
        const uint32_t p = ps != nullptr ? ps[h] : 0u;
        return (e >> ZB_EPOCH_SHIFT) == ep ? (e & ((1u << ZB_EPOCH_SHIFT) - 1u)) : p;
    }
    __device__ __forceinline__ void wrL(uint32_t h, uint32_t x, uint32_t y) const { ltab[h] = make_uint2(x | (ep << ZB_EPOCH_SHIFT), y); }
    __device__ __forceinline__ void wrS(uint32_t h, uint32_t x) const { stab[h] = x | (ep << ZB_EPOCH_SHIFT); }
    __device__ __forceinline__ uint32_t tagOf(uint32_t v) const { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; }
    __device__ __forceinline__ uint32_t mk(int pos, uint32_t val) const { return ((uint32_t)pos + 1u) | (tagOf(val) << PB); }
    __device__ __forceinline__ int posOf(uint32_t e) const { return (int)(e & posMask) - 1; }  // -1 == empty
    // candidate acceptable for an 8-byte compare against cv at position s?
    __device__ __forceinline__ bool long_ok(uint32_t e, int s, uint64_t cv) const {
        const int t = posOf(e);
        if (t < 0 || (s - t) >= mmo) return false;
        if ((e >> PB) != tagOf((uint32_t)cv)) return false;
        return ld64(base + t) == cv;
    }
};

template <bool DICT>
__global__ __launch_bounds__(64) void kc_zbetter_match_grp_kernel(KcMatchParams P, uint8_t* __restrict__ tables, uint32_t n_launch) {
    constexpr int G = ZBG;
    constexpr int UPW = 64 / G;
    __shared__ uint64_t sbuf_all[UPW * G];  // per unit: the last (nseq mod G) sequences
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[UPW * ZB_STRIDE];
    const int lane = (int)threadIdx.x;
    const int lig = lane % G, grp = lane / G;
    uint64_t* const sbuf = sbuf_all + grp * G;
    uint8_t* const ring = ring_all + grp * ZB_STRIDE;
    const uint32_t ui = blockIdx.x * UPW + (uint32_t)grp;
    const bool gact = ui < n_launch;
    const uint32_t u = gact ? (P.unit_list ? P.unit_list[ui] : P.unit_base + ui) : 0u;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    // history in front of the unit: the dictionary content, or (jobs of a WithConcurrentBlocks stream) the unit's own overlap prefix
    const int hist0 = P.unit_hist != nullptr ? (int)P.unit_hist[u] : P.hist0;
    const int ulen = gact ? (int)(P.unit_off[u + 1] - P.unit_off[u]) - hist0 : 0;
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
    const int nblk = gact ? UB.nblk : 0;  // a group without a unit (the launch's tail) does nothing
    const size_t tabBytes = ((size_t)8 << ZB_LONG_BITS) + ((size_t)4 << ZB_SHORT_BITS);
    ZbCtx C;
    C.base = base;
    C.ltab = (uint2*)(tables + (size_t)ui * tabBytes);
    C.stab = (uint32_t*)(tables + (size_t)ui * tabBytes + ((size_t)8 << ZB_LONG_BITS));
    C.PB = P.pos_bits;  // per-launch constant so that dictionary-primed tables can be shared by all units
    C.ep = P.epoch;
    C.pl = (P.epoch != 0u && P.proto != nullptr) ? (const uint2*)P.proto : nullptr;
    C.ps = (P.epoch != 0u && P.proto != nullptr) ? (const uint32_t*)(P.proto + ((size_t)8 << ZB_LONG_BITS)) : nullptr;
    {
        const int avail = 32 - C.PB - (P.epoch != 0u ? ZB_EPOCH_BITS : 0);
        C.TB = avail > 16 ? 16 : avail;
    }
    C.posMask = (1u << C.PB) - 1u;
    C.mmo = P.max_match_off;
    const int mmo = C.mmo;
    auto hL = [&](uint64_t v) -> uint32_t { return hash8(v, ZB_LONG_BITS); };
    auto hS = [&](uint64_t v) -> uint32_t { return hash5(v, ZB_SHORT_BITS); };
    // ---- source window ----
    const int boff = (int)((uintptr_t)base & 15);  // window positions are relative to the 16-byte aligned abase
    const uint8_t* __restrict__ abase = base - boff;
    const uint8_t* const srcHi = P.src_end;
    int wlo = 0, whi = 0;   // the ring holds the bytes abase[wlo .. whi)
    bool pend = false;      // rf holds the 16*G bytes abase[whi ..) loaded during the previous round
    uint4 rf = make_uint4(0, 0, 0, 0);
    // top of every probe round (group-uniform): take in the refill issued a round ago, keep the window ahead of s
    auto window = [&](int s) {
        if (!ZB_RING) { KC_EMU_SYNC(); return; }
        if (pend) {
            const int ro = (whi + 16 * lig) & (ZB_RB - 1);
            *(uint4*)(ring + ro) = rf;
            if (ro < ZB_MIRROR) *(uint4*)(ring + ZB_RB + ro) = rf;
            whi += 16 * G;
            if (whi - wlo > ZB_RB) wlo = whi - ZB_RB;
            pend = false;
        }
        KC_EMU_SYNC();  // (the ring is written by all lanes of the group and read by all of them)
        const int sa = s + boff;
        if (sa >= whi || sa < wlo) {  // block start, or a match jumped past the window: restart it at s
            const int w0 = sa & ~15;
            wlo = whi = w0;
        }
        if (whi - sa < ZB_AHEAD) {
            const uint8_t* q = abase + whi + 16 * lig;
            rf = make_uint4(0, 0, 0, 0);
            if (q < srcHi) rf = *(const uint4*)q;  // aligned: never leaves the 16-byte granule of a readable byte
            pend = true;
        }
    };
    // 8 / 4 source bytes at unit position pos (any lane, any position): from the ring when it holds them, else from memory
    auto rd64 = [&](int pos) -> uint64_t {
        const int a = pos + boff;
        const int a4 = a & ~3;
        if (ZB_RING && a4 >= wlo && a4 + 12 <= whi) {
            const uint32_t* r = (const uint32_t*)(ring + (a4 & (ZB_RB - 1)));
            const uint32_t r0 = r[0], r1 = r[1], r2 = r[2];
            const uint32_t sh = (uint32_t)(a & 3);
            return (uint64_t)__builtin_amdgcn_alignbyte(r1, r0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(r2, r1, sh) << 32);
        }
        return ld64(base + pos);
    };
    auto rd32 = [&](int pos) -> uint32_t {
        const int a = pos + boff;
        const int a4 = a & ~3;
        if (ZB_RING && a4 >= wlo && a4 + 8 <= whi) {
            const uint32_t* r = (const uint32_t*)(ring + (a4 & (ZB_RB - 1)));
            return __builtin_amdgcn_alignbyte(r[1], r[0], (uint32_t)(a & 3));
        }
        return ld32(base + pos);
    };
    // candidate acceptable for an 8-byte compare against cv at position s? (ZbCtx::long_ok with the candidate's bytes from the window)
    auto long_ok = [&](uint32_t e, int sp, uint64_t cv) -> bool {
        const int t = C.posOf(e);
        if (t < 0 || (sp - t) >= mmo) return false;
        if ((e >> C.PB) != C.tagOf((uint32_t)cv)) return false;
        return rd64(t) == cv;
    };

    int o1 = P.rep1, o2 = P.rep2;  // {1,4} (blockenc.go:78) or the dictionary's offsets (enc_base.go:189-195)
    for (int b = 0; b < nblk; b++) {
        const int blkStart = hist0 + kc_blk_begin(P.blk_start, blk0, b, bs);
        const int blkEnd = hist0 + kc_blk_end(P.blk_start, blk0, b, nblk, bs, ulen);  // == len(e.hist)
        const int srcLen = blkEnd - blkStart;
        const int o1_in = o1, o2_in = o2;
        uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        int nseq = 0, sumLL = 0;
        uint32_t rounds = 0;
        int nextEmit = blkStart, s = blkStart;
        uint32_t firstLL = 0, firstOf = 0;
        bool rleBlock = false;
        auto emit = [&](int ll, int ml3, uint32_t of) {
            if (nseq == 0) { firstLL = (uint32_t)ll; firstOf = of; }
            if (lig == 0) sbuf[nseq & (G - 1)] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
            nseq++;
            sumLL += ll;
            if ((nseq & (G - 1)) == 0) {  // group-uniform: G sequences buffered in LDS -> one coalesced 64-byte store
                __builtin_amdgcn_wave_barrier();
                sq[nseq - G + lig] = sbuf[lig];
                __builtin_amdgcn_wave_barrier();
            }
        };
        // dense re-indexing of [from, s-1) every 2nd byte (:438-447 / :157-165), group-parallel with in-order chaining
        auto reindex = [&](int from, int upto /* exclusive: s-1 */) {
            for (int i0 = from; i0 < upto; i0 += 2 * G) {
                const int idx = i0 + 2 * lig;
                const bool act = idx < upto;
                uint64_t cv0 = 0;
                uint32_t h0 = 0xFFFFFFFFu - (uint32_t)lig, h1 = 0xFFFFFF00u - (uint32_t)lig;
                if (act) { cv0 = rd64(idx); h0 = hL(cv0); h1 = hS(cv0 >> 8); }
                uint32_t oldOff = 0;
                if (act) oldOff = C.rdLx(h0);
                // nearest lower lane with the same long bucket supplies `prev`; a higher lane with the same bucket owns the store
                uint32_t prevE = oldOff;
                bool laterL = false, laterS = false;
                const uint32_t myE = act ? C.mk(idx, (uint32_t)cv0) : 0u;
#pragma unroll
                for (int d = G - 1; d >= 1; d--) {  // far to near, so the nearest lower match wins
                    const uint32_t a0 = (uint32_t)__shfl_up((int)h0, d, G);
                    const uint32_t ae = (uint32_t)__shfl_up((int)myE, d, G);
                    if (lig >= d && a0 == h0) prevE = ae;
                }
#pragma unroll
                for (int d = 1; d < G; d++) {
                    const uint32_t b0 = (uint32_t)__shfl_down((int)h0, d, G);
                    const uint32_t b1 = (uint32_t)__shfl_down((int)h1, d, G);
                    if (lig + d < G && b0 == h0) laterL = true;
                    if (lig + d < G && b1 == h1) laterS = true;
                }
                if (act && !laterL) C.wrL(h0, myE, prevE);
                if (act && !laterS) C.wrS(h1, C.mk(idx + 1, (uint32_t)(cv0 >> 8)));
                KC_EMU_SYNC();
            }
        };

        if (!DICT && srcLen > 3) {
            // Check RLE first (:109-117): the whole block is one byte repeated
            const int ml = grp_matchlen<G>(base, blkStart + 1, blkStart, srcLen - 1, lig, grp);
            if (ml == srcLen - 1) {
                emit(1, (srcLen - 1) - 3, 1u + 3u);
                rleBlock = true;
            }
        }
        if (!rleBlock && srcLen >= 16) {
            const int sLimit = blkEnd - 10;
            bool fin = false;
            int W = G;  // speculation width of the probe rounds
            while (!fin) {  // encodeLoop
                int t = 0, matched = 0, index0 = 0;
                const bool canRep = nseq > 2;
                bool brk = false;
                for (;;) {  // search loop: speculative probe rounds with ordered commit (as in the other match finders)
                    rounds++;
                    window(s);
                    // While no match is found the probe positions are a pure function of (s, nextEmit): s += 1 + ((s-nextEmit)>>8).
                    // The lanes probe the next W of them against the pre-round tables; a lane whose long or short bucket was
                    // touched by a lower lane ends the round; lanes up to the first hit commit their table writes (the reference
                    // writes both tables before it checks, :216-224); the hit is then processed for that one position exactly as
                    // the sequential loop does, with the winner's loaded entries.
                    const int d0 = s - nextEmit;
                    const int k0 = d0 >> 8;  // kSearchStrength-1 == 8
                    const int step = 1 + k0;
                    const int pp = s + lig * step;
                    const bool valid = lig < W && (lig == 0 || ((d0 + (lig - 1) * step) >> 8) == k0) && pp < sLimit;
                    uint64_t cvl = 0;
                    uint32_t hl = 0xFFFFFFFFu - (uint32_t)lig, hs = 0xFFFFFF00u - (uint32_t)lig;
                    uint2 eL = make_uint2(0u, 0u);
                    uint32_t eS = 0;
                    if (valid) {
                        cvl = rd64(pp);
                        hl = hL(cvl);
                        hs = hS(cvl);
                        eL = C.rdL(hl);
                        eS = C.rdS(hs);
                    }
                    bool dep = false;
#pragma unroll
                    for (int dd = 1; dd < G; dd++) {
                        const uint32_t al = (uint32_t)__shfl_up((int)hl, dd, G), as = (uint32_t)__shfl_up((int)hs, dd, G);
                        if (lig >= dd && (al == hl || as == hs)) dep = true;
                    }
                    uint32_t hit = 0;  // 1 repeat at s+1, 2 long (offset), 4 long (prev), 8 short
                    if (valid) {
                        const int ri = pp - o1 + 1;
                        // (candidates a few hundred bytes back are in the window too: rd32 / rd64)
#if ZB_FUSE
                        // One round trip for the round's candidate bytes: the repeat, both long candidates and the short one are loaded
                        // together (each behind its own position / tag test) and judged afterwards in the reference's order — they used
                        // to be up to four dependent trips (repeat -> long -> prev -> short) on a kernel bound by exactly those.
                        const bool tryR = canRep && ri >= 0;
                        const int tx = C.posOf(eL.x), ty = C.posOf(eL.y), ts = C.posOf(eS);
                        const uint32_t tg = C.tagOf((uint32_t)cvl);
                        const bool okx = tx >= 0 && (pp - tx) < mmo && (eL.x >> C.PB) == tg;
                        const bool oky = ty >= 0 && (pp - ty) < mmo && (eL.y >> C.PB) == tg;
                        const bool oks = ts >= 0 && (pp - ts) < mmo && (eS >> C.PB) == tg;
                        uint32_t vr = 0, vs = 0;
                        uint64_t vx = 0, vy = 0;
                        if (tryR) vr = rd32(ri);
                        if (okx) vx = rd64(tx);
                        if (oky) vy = rd64(ty);
                        if (oks) vs = rd32(ts);
                        if (tryR && vr == (uint32_t)(cvl >> 8)) hit = 1;
                        else {
                            if (okx && vx == cvl) hit |= 2;
                            if (oky && vy == cvl) hit |= 4;
                            if (hit == 0 && oks && vs == (uint32_t)cvl) hit = 8;
                        }
#else
                        if (canRep && ri >= 0 && rd32(ri) == (uint32_t)(cvl >> 8)) hit = 1;
                        else {
                            if (long_ok(eL.x, pp, cvl)) hit |= 2;
                            if (long_ok(eL.y, pp, cvl)) hit |= 4;
                            if (hit == 0) {
                                const int ts = C.posOf(eS);
                                if (ts >= 0 && (pp - ts) < mmo && (eS >> C.PB) == C.tagOf((uint32_t)cvl) && rd32(ts) == (uint32_t)cvl) hit = 8;
                            }
                        }
#endif
                    }
                    const uint32_t vm = gballot<G>(valid, grp);
                    const uint32_t depm = gballot<G>(valid && dep, grp);
                    const uint32_t hm = gballot<G>(hit != 0, grp);
                    const int nvalid = __popc(vm);
                    const int cc = depm ? __builtin_ctz(depm) : G;
                    const uint32_t hmc = hm & (uint32_t)((1ull << cc) - 1ull);
                    const bool found = hmc != 0;
                    const int f = found ? __builtin_ctz(hmc) : 0;
                    const int commitUpTo = found ? f : ((cc < nvalid ? cc : nvalid) - 1);
                    if (valid && lig <= commitUpTo) {
                        const uint32_t e = C.mk(pp, (uint32_t)cvl);
                        C.wrL(hl, e, eL.x);
                        C.wrS(hs, e);
                    }
                    if (!found) {
                        W = P.spec_grow == 0 ? W : (P.spec_grow == 1 ? (W + 1 < G ? W + 1 : G) : ((2 * W < G) ? 2 * W : G));
                        if (cc < nvalid) {
                            s = s + cc * step;
                        } else {
                            const int pl = s + (nvalid - 1) * step;
                            s = pl + 1 + ((pl - nextEmit) >> 8);
                        }
                        if (s >= sLimit) { fin = true; brk = true; break; }
                        continue;
                    }
                    W = P.spec_w0;
                    s = s + f * step;
                    const uint64_t cv = gbcast64<G>(cvl, grp, f);
                    const uint2 cL = make_uint2(gbcast32<G>(eL.x, grp, f), gbcast32<G>(eL.y, grp, f));
                    const uint32_t cS = gbcast32<G>(eS, grp, f);
                    const uint32_t whit = gbcast32<G>(hit, grp, f);
                    const int repIndex = s - o1 + 1;
                    index0 = s + 1;
                    if (whit & 1u) {
                        int ri = repIndex;
                        const int length = 4 + grp_matchlen<G>(base, s + 5, ri + 4, blkEnd - (s + 5), lig, grp);
                        int start = s + 1;
                        const int startLimit = nextEmit + 1;
                        const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                        int kmax = ri - tMin;
                        if (start - startLimit < kmax) kmax = start - startLimit;
                        const int cap = (ZB_MAX_MATCH_LENGTH - 3 - 1) - (length - 3);
                        if (cap < kmax) kmax = cap;
                        if (kmax < 0) kmax = 0;
                        const int back = grp_backlen<G>(base, start, ri, kmax, lig, grp);
                        start -= back;
                        emit(start - nextEmit, length - 3 + back, 1u);
                        const int idx = s + 1;
                        s += length + 1;
                        nextEmit = s;
                        if (s >= sLimit) { fin = true; brk = true; break; }
                        reindex(idx, s - 1);
                        continue;
                    }
                    // long match on offset, possibly improved by prev (:264-296)
                    const bool okL = (whit & 2u) != 0;
                    const bool okP = (whit & 4u) != 0;
                    if (okL) {
                        const int tl = C.posOf(cL.x);
                        matched = grp_matchlen<G>(base, s + 8, tl + 8, blkEnd - (s + 8), lig, grp) + 8;
                        t = tl;
                        if (okP) {
                            const int tp = C.posOf(cL.y);
                            const int pm2 = grp_matchlen<G>(base, s + 8, tp + 8, blkEnd - (s + 8), lig, grp) + 8;
                            if (pm2 > matched) { matched = pm2; t = tp; }
                        }
                        break;
                    }
                    if (okP) {
                        const int tp = C.posOf(cL.y);
                        matched = grp_matchlen<G>(base, s + 8, tp + 8, blkEnd - (s + 8), lig, grp) + 8;
                        t = tp;
                        break;
                    }
                    {
                        const int ts = C.posOf(cS);
                        {  // whit == 8: the short candidate was accepted on its 4 bytes
#if ZB_FUSE
                            // long match at s+1? (:309-343) — its table entry is requested BEFORE the short candidate is extended (the
                            // two trips overlap), and its two candidates' bytes are loaded together
                            const uint64_t cv2 = rd64(s + 1);
                            const uint32_t nh2 = hL(cv2);
                            const uint2 c2 = C.rdL(nh2);
                            matched = grp_matchlen<G>(base, s + 4, ts + 4, blkEnd - (s + 4), lig, grp) + 4;
                            KC_EMU_SYNC();
                            if (lig == 0) C.wrL(nh2, C.mk(s + 1, (uint32_t)cv2), c2.x);
                            // s-coffsetL < maxMatchOff is evaluated with s (not s+1) in the reference
                            const int tl = C.posOf(c2.x), tp = C.posOf(c2.y);
                            const uint32_t tg2 = C.tagOf((uint32_t)cv2);
                            const bool okl = tl >= 0 && (s - tl) < mmo && (c2.x >> C.PB) == tg2;
                            const bool okp = tp >= 0 && (s - tp) < mmo && (c2.y >> C.PB) == tg2;
                            uint64_t vl = 0, vp = 0;
                            if (okl) vl = ld64(base + tl);
                            if (okp) vp = ld64(base + tp);
                            bool taken = false;
                            if (okl && vl == cv2) {
                                const int mn = grp_matchlen<G>(base, s + 9, tl + 8, blkEnd - (s + 9), lig, grp) + 8;
                                if (mn > matched) { t = tl; s += 1; matched = mn; taken = true; }
                            }
                            if (!taken && okp && vp == cv2) {
                                const int mn = grp_matchlen<G>(base, s + 9, tp + 8, blkEnd - (s + 9), lig, grp) + 8;
                                if (mn > matched) { t = tp; s += 1; matched = mn; taken = true; }
                            }
                            if (!taken) t = ts;
                            break;
#else
                            matched = grp_matchlen<G>(base, s + 4, ts + 4, blkEnd - (s + 4), lig, grp) + 4;
                            // long match at s+1? (:309-343)
                            const uint64_t cv2 = rd64(s + 1);
                            const uint32_t nh2 = hL(cv2);
                            const uint2 c2 = C.rdL(nh2);
                            KC_EMU_SYNC();
                            if (lig == 0) C.wrL(nh2, C.mk(s + 1, (uint32_t)cv2), c2.x);
                            // s-coffsetL < maxMatchOff is evaluated with s (not s+1) in the reference
                            bool taken = false;
                            {
                                const int tl = C.posOf(c2.x);
                                if (tl >= 0 && (s - tl) < mmo && (c2.x >> C.PB) == C.tagOf((uint32_t)cv2) && ld64(base + tl) == cv2) {
                                    const int mn = grp_matchlen<G>(base, s + 9, tl + 8, blkEnd - (s + 9), lig, grp) + 8;
                                    if (mn > matched) { t = tl; s += 1; matched = mn; taken = true; }
                                }
                            }
                            if (!taken) {
                                const int tp = C.posOf(c2.y);
                                if (tp >= 0 && (s - tp) < mmo && (c2.y >> C.PB) == C.tagOf((uint32_t)cv2) && ld64(base + tp) == cv2) {
                                    const int mn = grp_matchlen<G>(base, s + 9, tp + 8, blkEnd - (s + 9), lig, grp) + 8;
                                    if (mn > matched) { t = tp; s += 1; matched = mn; taken = true; }
                                }
                            }
                            if (!taken) t = ts;
                            break;
#endif
                        }
                    }
                    // not reached: a found round always ends in one of the branches above
                }
                if (brk) continue;  // leaves encodeLoop when fin, or restarts the search after nothing (not reached)
                // ---- end-of-match re-search (:419-460) ----
#if ZB_FUSE
                // The backward extension's first G bytes are requested here for the match as it stands, under the re-search's table trip:
                // the re-search rarely replaces the match, and when it does they are dropped and the extension starts over.
                auto back_kmax = [&](int sp, int tp, int len) -> int {
                    const int tMin = (sp - mmo) > 0 ? (sp - mmo) : 0;
                    int kmax = tp - tMin;
                    if (sp - nextEmit < kmax) kmax = sp - nextEmit;
                    if ((ZB_MAX_MATCH_LENGTH - len) < kmax) kmax = ZB_MAX_MATCH_LENGTH - len;
                    return kmax < 0 ? 0 : kmax;
                };
                const int s_pre = s, t_pre = t, m_pre = matched;
                const int kmax_pre = back_kmax(s, t, matched);
                uint32_t bpS = 0, bpT = 1;  // (unequal: a lane beyond kmax ends the extension)
                if (lig + 1 <= kmax_pre) { bpS = base[s - (lig + 1)]; bpT = base[t - (lig + 1)]; }
#endif
                if (s + matched < sLimit) {
                    const int skipBeginning = DICT ? 0 : 3;
                    const uint32_t nh = hL(rd64(s + matched));
                    const int s2 = s + skipBeginning;
                    const uint32_t cv4 = rd32(s2);
                    KC_EMU_SYNC();
                    const uint2 cE = C.rdL(nh);
#if ZB_FUSE
                    // both candidates' bytes in one trip; the prev candidate's position depends on the match length (:445), so when the
                    // first candidate lengthened the match it is looked at again
                    const int m0 = matched;
                    const int cox = C.posOf(cE.x) - m0 + skipBeginning;
                    int coy = C.posOf(cE.y) - m0 + skipBeginning;
                    const bool okx = C.posOf(cE.x) >= 0 && cox >= 0 && cox < s2 && (s2 - cox) < mmo;
                    bool oky = C.posOf(cE.y) >= 0 && coy >= 0 && coy < s2 && (s2 - coy) < mmo;
                    uint32_t vx = 0, vy = 0;
                    if (okx) vx = ld32(base + cox);
                    if (oky) vy = ld32(base + coy);
                    if (okx && cv4 == vx) {
                        const int mn = grp_matchlen<G>(base, s2 + 4, cox + 4, blkEnd - (s2 + 4), lig, grp) + 4;
                        if (mn > matched) { t = cox; s = s2; matched = mn; }
                    }
                    if (matched != m0) {
                        coy = C.posOf(cE.y) - matched + skipBeginning;
                        oky = C.posOf(cE.y) >= 0 && coy >= 0 && coy < s2 && (s2 - coy) < mmo;
                        vy = 0;
                        if (oky) vy = ld32(base + coy);
                    }
                    if (oky && cv4 == vy) {
                        const int mn = grp_matchlen<G>(base, s2 + 4, coy + 4, blkEnd - (s2 + 4), lig, grp) + 4;
                        if (mn > matched) { t = coy; s = s2; matched = mn; }
                    }
#else
                    {
                        const int co = C.posOf(cE.x) - matched + skipBeginning;
                        if (C.posOf(cE.x) >= 0 && co >= 0 && co < s2 && (s2 - co) < mmo && cv4 == ld32(base + co)) {
                            const int mn = grp_matchlen<G>(base, s2 + 4, co + 4, blkEnd - (s2 + 4), lig, grp) + 4;
                            if (mn > matched) { t = co; s = s2; matched = mn; }
                        }
                    }
                    {
                        const int co = C.posOf(cE.y) - matched + skipBeginning;
                        if (C.posOf(cE.y) >= 0 && co >= 0 && co < s2 && (s2 - co) < mmo && cv4 == ld32(base + co)) {
                            const int mn = grp_matchlen<G>(base, s2 + 4, co + 4, blkEnd - (s2 + 4), lig, grp) + 4;
                            if (mn > matched) { t = co; s = s2; matched = mn; }
                        }
                    }
#endif
                }
                o2 = o1;
                o1 = s - t;
                int l = matched;
#if ZB_FUSE
                {
                    const int kmax = back_kmax(s, t, l);
                    int back;
                    if (s == s_pre && t == t_pre && l == m_pre) {  // the match the bytes were requested for (kmax == kmax_pre)
                        const uint32_t m = gballot<G>(bpS != bpT, grp);
                        const int c = m ? __builtin_ctz(m) : G;
                        back = c;
                        if (c == G && G < kmax) back = G + grp_backlen<G>(base, s - G, t - G, kmax - G, lig, grp);
                        if (back > kmax) back = kmax;
                    } else {
                        back = grp_backlen<G>(base, s, t, kmax, lig, grp);
                    }
                    s -= back;
                    t -= back;
                    l += back;
                }
#else
                {
                    const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = t - tMin;
                    if (s - nextEmit < kmax) kmax = s - nextEmit;
                    if ((ZB_MAX_MATCH_LENGTH - l) < kmax) kmax = ZB_MAX_MATCH_LENGTH - l;
                    if (kmax < 0) kmax = 0;
                    const int back = grp_backlen<G>(base, s, t, kmax, lig, grp);
                    s -= back;
                    t -= back;
                    l += back;
                }
#endif
                emit(s - nextEmit, l - 3, (uint32_t)(s - t) + 3u);
                s += l;
                nextEmit = s;
                if (s >= sLimit) { fin = true; continue; }
#if ZB_FUSE
                uint32_t o2pre = 0;  // the offset-2 loop's first candidate bytes, requested in front of the re-indexing's table trips
                if (canRep) o2pre = ld32(base + (s - o2));
#endif
                reindex(index0, s - 1);
                if (!canRep) continue;
                for (bool first = true;; first = false) {  // offset-2 loop (:482-522)
                    const uint64_t cvs = rd64(s);
                    const int o2pos = s - o2;
#if ZB_FUSE
                    if ((first ? o2pre : ld32(base + o2pos)) != (uint32_t)cvs) break;
#else
                    if (ld32(base + o2pos) != (uint32_t)cvs) break;
#endif
                    const uint32_t nhL2 = hL(cvs), nhS2 = hS(cvs);
                    const int l2 = 4 + grp_matchlen<G>(base, s + 4, o2pos + 4, blkEnd - (s + 4), lig, grp);
                    const uint32_t oldx = C.rdLx(nhL2);
                    if (lig == 0) {
                        C.wrL(nhL2, C.mk(s, (uint32_t)cvs), oldx);
                        C.wrS(nhS2, C.mk(s, (uint32_t)cvs));
                    }
                    emit(0, l2 - 3, 1u);
                    s += l2;
                    nextEmit = s;
                    const int tmp = o1; o1 = o2; o2 = tmp;
                    if (s >= sLimit) { fin = true; break; }
                }
            }
        }
        pend = false;  // a refill still in flight at the end of a block is dropped; the window itself stays valid
        __builtin_amdgcn_wave_barrier();
        if (lig < (nseq & (G - 1))) sq[(nseq & ~(G - 1)) + lig] = sbuf[lig];  // the buffered tail of the sequence list
        int extra = nextEmit < blkEnd ? blkEnd - nextEmit : 0;
        int nlit = sumLL + extra;
        if (rleBlock) { extra = 0; nlit = 1; }  // literals = src[0]; recentOffsets untouched (early return, :114)
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

void kc_launch_zbetter_match_grp(const KcMatchParams& P, uint8_t* tables, uint32_t n_launch, bool dict, hipStream_t st) {
    constexpr uint32_t UPW = 64 / ZBG;
    if (dict) hipLaunchKernelGGL(kc_zbetter_match_grp_kernel<true>, dim3((n_launch + UPW - 1) / UPW), dim3(64), 0, st, P, tables, n_launch);
    else hipLaunchKernelGGL(kc_zbetter_match_grp_kernel<false>, dim3((n_launch + UPW - 1) / UPW), dim3(64), 0, st, P, tables, n_launch);
}
