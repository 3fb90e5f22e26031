// kc_zstd_match_lds.hip — SpeedFastest match finder with the hash table in LDS: ONE WAVE PER UNIT, the latency path.
//
// Same reference functions and the same decisions as kc_zstd_match.hip (fastEncoder.Encode / EncodeNoHist /
// fastEncoderDict.Encode, zstd/enc_fast.go:39-289 / 294-531 / 534-790), same output (packed sequences + KcBlkMeta per
// block, consumed by kc_zstd_entropy_kernel).  kc_zstd_match.hip keeps 8 units per wave in flight with their tables in
// HBM: it is bound by DRAM transactions and needs ~10^4 units to cover the latency of the dependent table -> candidate
// chain (~30 ms for one 128 KiB unit).  Here the unit's table (2^15 x u32 = 128 KiB of the CU's 160 KiB of LDS) is on
// chip, so a table lookup is an LDS round trip and only candidate bytes come through L2: a few ms per unit, whatever
// the number of units in flight — and a handful of rounds per block where the scan finds nothing (high-entropy input:
// 64 probe steps per round instead of 8, and no table traffic to HBM at all).  The dispatcher in kc_api.cpp picks the
// path by units in flight (measured crossover, profiles/r03_crossover_zfast.csv).
//
// Scheme: the 64 lanes evaluate the next W probe steps of the scan (positions follow s += 2 + (s-nextEmit)>>5 exactly,
// each lane iterating the recurrence up to its own step when the steps leave the first skip segment) against the
// pre-round table; ballot + ctz picks the first step that ends the scan in the reference's priority order (repeat at
// s+2, candidate at s, candidate at s+1); steps up to it commit their table writes.  Table entry =
// (position+1):26 | tag:6 — the tag (top bits of a multiplicative hash of the 4 source bytes) stands for
// tableEntry.val: a mismatch skips the candidate fetch exactly when the reference would reject.  Two steps of one round
// that share a bucket find each other through marker bytes (16 KiB of LDS, one byte per pair of buckets): every lane
// stores its lane id into the markers of its two buckets (ds_write_b8), reads them back, and a lane that finds another
// lane's id tells that lane through a 64-bit mask (ds_or_b64; only in rounds where some lane lost).  The round is cut at
// the lowest sharing lane other than lane 0, so every committed step saw the table the sequential encoder would have
// seen.  Units of any length up to 64 MiB: long streams and the jobs of a WithConcurrentBlocks stream (their overlap
// prefix is the unit's history, the table arrives primed from it) run here too.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_zfast_dev.h"

#define ZL_RB 4096      // source ring bytes (power of two)
#define ZL_RB_SMALL 65536          // ... of the instantiation for units up to ZL_SMALL_MAX_UNIT bytes without history: table 2^15 x 17 bits
#define ZL_SMALL_MAX_UNIT 131072   // (positions + 1 below 2^17)
#define ZL_MIRROR 32    // the first 32 ring bytes are mirrored behind the ring: 24-byte reads never wrap
#define ZL_BK 4         // bytes in front of a probe / candidate position kept for the backward extension
#define ZL_AHEAD 1536   // refill (1 KiB per round) while fewer than this many bytes are buffered ahead of s
#define ZL_PB 26        // position+1 bits: units (with their history) below 64 MiB
#define ZL_TAGB 6
#define ZL_MARK_BITS 14 // marker bytes: one per PAIR of buckets (two buckets sharing a marker are a false, harmless, conflict)
#define ZL_POS_MASK ((1u << ZL_PB) - 1u)

#ifdef KC_LDS_PROF  // diagnostics: shader clocks per phase of a round, summed over the launch (lane 0 of every wave)
#define LP_DECL unsigned long long lp_t = __builtin_amdgcn_s_memtime(), lp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define LP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); lp_acc[i] += t_ - lp_t; lp_t = t_; } while (0)
#define LP_FLUSH(ptr) do { if (lane == 0 && (ptr) != nullptr) for (int k_ = 0; k_ < 8; k_++) atomicAdd(&(ptr)[k_], lp_acc[k_]); } while (0)
#else
#define LP_DECL
#define LP(i)
#define LP_FLUSH(ptr)
#endif

__device__ __forceinline__ uint32_t zl_tag(uint32_t v) { return (v * 2654435761u) >> (32 - ZL_TAGB); }
__device__ __forceinline__ uint32_t zl_entry(int pos, uint32_t v) { return ((uint32_t)pos + 1u) | (zl_tag(v) << ZL_PB); }

// proto: null, or primed tables in the HBM kernels' format ((position+1) | tag << pos_bits, tag = top bits of the same hash),
// converted while loaded: ONE table for all units (dictionary), or one per launch slot (proto_stride = 2^15: the jobs of a
// WithConcurrentBlocks stream, each primed from its own overlap prefix).
template <bool SMALL>
__global__ __launch_bounds__(64) void kc_zfast_match_lds_kernel(KcMatchParams P, const uint32_t* __restrict__ proto, uint32_t proto_stride, uint32_t n_launch, bool skip_small) {
    constexpr int RB = SMALL ? ZL_RB_SMALL : ZL_RB;
    constexpr int AHEAD = SMALL ? 4096 : ZL_AHEAD;
    __shared__ uint32_t tab[SMALL ? (1 << ZF_TABLE_BITS) / 2 : (1 << ZF_TABLE_BITS)];  // SMALL: 2^15 x u16, (position + 1) & 0xFFFF
    __shared__ uint32_t hib[SMALL ? (1 << ZF_TABLE_BITS) / 32 : 1];                    // SMALL: bit 16 of position + 1, one bit per entry
    __shared__ uint8_t mark[1 << ZL_MARK_BITS];
    __shared__ __attribute__((aligned(16))) uint8_t ring[RB + ZL_MIRROR];
    uint16_t* const tab16 = (uint16_t*)tab;
    __shared__ uint64_t sbuf[64];  // the last (nseq mod 64) sequences, flushed 64 at a time as one 512-byte store
    __shared__ unsigned long long shareMask;
    const int lane = (int)threadIdx.x;
    const uint32_t ui = blockIdx.x;
    if (ui >= n_launch) return;
    const uint32_t u = P.unit_list ? P.unit_list[ui] : P.unit_base + ui;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int boff = (int)((uintptr_t)base & 15);  // window positions are relative to the 16-byte aligned abase
    const uint8_t* __restrict__ abase = base - boff;
    const int hist0 = P.unit_hist != nullptr ? (int)P.unit_hist[u] : P.hist0;  // dictionary content, or a job's overlap prefix
    const int ulen = (int)(P.unit_off[u + 1] - P.unit_off[u]) - hist0;
    if ((uint32_t)(ulen + hist0) > KC_ZFAST_LDS_MAX_UNIT) return;  // beyond the 26-bit position field (64 MiB): the HBM-table kernel's unit
    if (SMALL ? (ulen > ZL_SMALL_MAX_UNIT) : (skip_small && ulen <= ZL_SMALL_MAX_UNIT)) return;  // the other instantiation's unit
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const int mmo = P.max_match_off;
    const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
    const int nblk = (P.unit_done != nullptr && P.unit_done[u] != 0u) ? 0 : UB.nblk;  // (done: the pre-scan proved the unit free of matches, kc_zstd_prescan.hip)
    const bool HIST = ulen > bs || hist0 > 0 || UB.streamU || P.job_flags != nullptr;  // with a dictionary encodeAll always calls Encode (encoder.go:783-787), and so does compressJob
    const uint8_t* const srcLo = P.src;
    const uint8_t* const srcHi = P.src_end;
    if (proto != nullptr) proto += (size_t)ui * proto_stride;

    if (SMALL) {
        for (int i = lane * 4; i < (1 << ZF_TABLE_BITS) / 2; i += 256) *(uint4*)&tab[i] = make_uint4(0, 0, 0, 0);
        for (int i = lane; i < (1 << ZF_TABLE_BITS) / 32; i += 64) hib[i] = 0u;
    } else if (proto == nullptr) {
        for (int i = lane * 4; i < (1 << ZF_TABLE_BITS); i += 256) *(uint4*)&tab[i] = make_uint4(0, 0, 0, 0);
    } else {
        const int PBh = P.pos_bits;
        const int TBh = (32 - PBh) > 16 ? 16 : (32 - PBh);
        const uint32_t pmh = (1u << PBh) - 1u;
        for (int i = lane; i < (1 << ZF_TABLE_BITS); i += 64) {
            const uint32_t e = proto[i];
            // the HBM format's tag is the top TBh bits of the same hash; where the batch's position field leaves fewer than ZL_TAGB of
            // them (a unit of 64 MiB and more beside this one: pos_bits >= 27) the tag is taken from the entry's source bytes instead
            uint32_t tg = 0u;
            if (e != 0u) tg = TBh >= ZL_TAGB ? ((e >> PBh) >> (TBh - ZL_TAGB)) : zl_tag(ld32(base + ((e & pmh) - 1u)));
            tab[i] = e == 0u ? 0u : ((e & pmh) | (tg << ZL_PB));
        }
    }
    KC_WAVE_SYNC();
    LP_DECL;
    // SMALL: entries are position + 1 in 17 bits and carry no tag — with the source in LDS a candidate is verified on its bytes at the
    // price of an LDS read; bit 16 lives in `hib` and is only ever set (positions grow), so the first 64 KiB never touch it
    auto tab_get = [&](uint32_t h, bool hiLive) -> uint32_t {
        if (!SMALL) return tab[h];
        uint32_t e = tab16[h];
        if (hiLive) e |= ((hib[h >> 5] >> (h & 31u)) & 1u) << 16;
        return e;
    };
    auto tab_put = [&](uint32_t h, int pos, uint32_t v) {
        if (!SMALL) { tab[h] = zl_entry(pos, v); return; }
        tab16[h] = (uint16_t)(pos + 1);
        if (pos + 1 >= 65536) atomicOr(&hib[h >> 5], 1u << (h & 31u));
    };
    // 16 bytes at unit position x (x - 4 for the callers' [t-4, t+12) windows): from the ring where it holds them (SMALL: 64 KiB of it)
    auto in_ring = [&](int x, int n, int wlo_, int whi_) -> bool { return SMALL && x + boff >= wlo_ && x + boff + n <= whi_; };

    int o1 = P.rep1, o2 = P.rep2;  // {1,4} (blockenc.go:78) or the dictionary's offsets (enc_base.go:189-195)
    bool allDirty = false;  // fastEncoderDict.allDirty: small-input variant (kSearchStrength 7) only until a block > 32 KiB was seen
    int wlo = 0, whi = 0;   // the ring holds the bytes abase[wlo .. whi)
    bool pend = false;      // rf holds the 1024 bytes abase[whi ..) loaded during the previous round
    uint4 rf = make_uint4(0, 0, 0, 0);
    const int W0 = P.spec_w0 < 1 ? 1 : (P.spec_w0 > 64 ? 64 : P.spec_w0);
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
            if (lane == 0) sbuf[nseq & 63] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
            nseq++;
            sumLL += ll;
            if ((nseq & 63) == 0) {
                KC_WAVE_SYNC();
                sq[nseq - 64 + lane] = sbuf[lane];
                KC_WAVE_SYNC();
            }
        };
        int SK = 5;  // kSearchStrength - 1
        if (hist0 > 0 && P.job_flags == nullptr) {  // fastEncoderDict only (enc_fast.go:539-543,585); a job's prefix goes through fastEncoder
            if (allDirty || srcLen > (32 << 10)) allDirty = true; else SK = 6;
        }
        if (srcLen >= 10) {
            const int sLimit = blkEnd - 8;
            bool canRep = false, fin = false, pendO2 = false;
            int W = W0;
            while (!fin) {
                if (++rounds > (uint32_t)srcLen + 16u) break;  // every round advances s: cannot happen; never spin on the device
                LP(7);  // the tail of the previous round: winner broadcast, match extension, sequence emit
                // ---------------- source window (LDS ring) ----------------
                if (pend) {  // the refill issued one round ago has landed
                    const int ro = (whi + 16 * lane) & (RB - 1);
                    *(uint4*)(ring + ro) = rf;
                    if (ro < ZL_MIRROR) *(uint4*)(ring + RB + ro) = rf;
                    whi += 1024;
                    if (whi - wlo > RB) wlo = whi - RB;
                    pend = false;
                    KC_WAVE_SYNC();
                }
                const int sa = s + boff;
                if (sa >= whi || sa < wlo) {  // block start, or a match jumped past the window: restart it just behind s
                    int w0 = (sa - 16) & ~15;
                    if (w0 < 0) w0 = 0;
                    wlo = whi = w0;
                }
                if (whi - sa < AHEAD) {
                    const uint8_t* q = abase + whi + 16 * lane;
                    rf = make_uint4(0, 0, 0, 0);
                    if (q < srcHi) rf = *(const uint4*)q;  // aligned: never leaves the 16-byte granule of a readable byte
                    pend = true;
                }
                // ---------------- probe positions of this round: lane i = the i-th step from s ----------------
                const int d0 = s - nextEmit;
                const int step0 = 2 + (d0 >> SK);
                int p;
                if (((d0 + (W - 1) * step0) >> SK) == (d0 >> SK)) {
                    p = s + lane * step0;  // all W steps inside one skip segment
                } else {
                    p = s;
                    for (int k = 0; k + 1 < W; k++) if (k < lane) p += 2 + ((p - nextEmit) >> SK);
                }
                const bool valid = lane < W && p < sLimit;  // a prefix of the lanes
                LP(0);  // window upkeep
                // repeat and offset-2 candidates first: their addresses need p, o1, o2 only, their latency hides under the table lookup
                const int repIndex = p - o1 + 2;
                const bool repOk = valid && canRep && repIndex >= 0;
                uint4 cr = make_uint4(0, 0, 0, 0);
                bool repWide = false;
                if (repOk) {
                    const uint8_t* q = base + repIndex - ZL_BK;
                    repWide = q >= srcLo && q + 16 <= srcHi;
                    if (in_ring(repIndex - ZL_BK, 16, wlo, whi)) { cr = ld128u(ring + ((repIndex - ZL_BK + boff) & (RB - 1))); repWide = true; }
                    else if (repWide) cr = ld128u(q);
                    else cr.y = ld32(base + repIndex);
                }
                const bool doO2 = pendO2;
                const int o2pos = s - o2;
                uint4 co = make_uint4(0, 0, 0, 0);
                bool o2Wide = false;
                if (doO2 && lane == 0) {  // offset-2 check (enc_fast.go:250), speculatively in the same round trip as the probes
                    const uint8_t* q = base + o2pos;
                    o2Wide = q + 16 <= srcHi;
                    if (in_ring(o2pos, 16, wlo, whi)) { co = ld128u(ring + ((o2pos + boff) & (RB - 1))); o2Wide = true; }
                    else if (o2Wide) co = ld128u(q);
                    else co.x = ld32(q);
                }
                // R = the 20 source bytes [p-4, p+16): D1:D2 = cv, the rest feeds the fused candidate compares
                uint32_t D0 = 0, D1 = 0, D2 = 0, D3 = 0, D4 = 0;
                if (valid) {
                    const int a = p + boff - ZL_BK;
                    const int a4 = a & ~3;
                    if (a4 >= wlo && a4 + 24 <= whi) {
                        const uint8_t* r = ring + (a & (RB - 1));  // unaligned ds_read_b128 + b32; the mirror keeps the 20 bytes contiguous
                        const uint4 r4 = ld128u(r);
                        D0 = r4.x; D1 = r4.y; D2 = r4.z; D3 = r4.w;
                        D4 = ld32(r + 16);
                    } else {
                        const uint8_t* q = base + p - ZL_BK;
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
                LP(1);  // positions, repeat / offset-2 loads issued, probe bytes read
                // ---------------- table entries (LDS) ----------------
                uint32_t h0 = 0, h1 = 0, c0 = 0, c1 = 0;
                if (valid) {
                    h0 = hash6(cv, ZF_TABLE_BITS);
                    h1 = hash6(cv >> 8, ZF_TABLE_BITS);
                    mark[h0 >> (ZF_TABLE_BITS - ZL_MARK_BITS)] = (uint8_t)lane;
                    mark[h1 >> (ZF_TABLE_BITS - ZL_MARK_BITS)] = (uint8_t)lane;
                }
                KC_WAVE_SYNC();
                uint32_t m0 = (uint32_t)lane, m1 = (uint32_t)lane;
                if (valid) {
                    const bool hiLive = SMALL && s >= 65000;  // (some entry may carry bit 16 once a position + 1 reached 2^16)
                    c0 = tab_get(h0, hiLive); c1 = tab_get(h1, hiLive);
                    m0 = mark[h0 >> (ZF_TABLE_BITS - ZL_MARK_BITS)]; m1 = mark[h1 >> (ZF_TABLE_BITS - ZL_MARK_BITS)];
                }
                LP(2);  // hashes, markers, table entries
                // candidates: one 16-byte load of [t-4, t+12) each, only where the tag matches
                const uint32_t e0 = SMALL ? c0 : (c0 & ZL_POS_MASK), e1 = SMALL ? c1 : (c1 & ZL_POS_MASK);
                const int t0 = (int)e0 - 1, t1 = (int)e1 - 1;
                const bool ok0 = valid && e0 != 0 && (p - t0) < mmo && (SMALL || (c0 >> ZL_PB) == zl_tag((uint32_t)cv));
                const bool ok1 = valid && e1 != 0 && (p - t1 + 1) < mmo && (SMALL || (c1 >> ZL_PB) == zl_tag((uint32_t)(cv >> 8)));
                uint4 ca = make_uint4(0, 0, 0, 0), cb = make_uint4(0, 0, 0, 0);
                bool wide0 = false, wide1 = false;
                if (ok0) {
                    const uint8_t* q = base + t0 - ZL_BK;
                    wide0 = q >= srcLo && q + 16 <= srcHi;
                    if (in_ring(t0 - ZL_BK, 16, wlo, whi)) { ca = ld128u(ring + ((t0 - ZL_BK + boff) & (RB - 1))); wide0 = true; }
                    else if (wide0) ca = ld128u(q);
                    else ca.y = ld32(base + t0);
                }
                if (ok1) {
                    const uint8_t* q = base + t1 - ZL_BK;
                    wide1 = q >= srcLo && q + 16 <= srcHi;
                    if (in_ring(t1 - ZL_BK, 16, wlo, whi)) { cb = ld128u(ring + ((t1 - ZL_BK + boff) & (RB - 1))); wide1 = true; }
                    else if (wide1) cb = ld128u(q);
                    else cb.y = ld32(base + t1);
                }
                if (doO2) {
                    pendO2 = false;
                    uint32_t pk = 0;  // bit 0: hit, bit 1: length final, bits 8..: known length
                    if (lane == 0) {
                        int f;
                        int fa;
                        if (o2Wide) {
                            const uint32_t x0 = co.x ^ D1, x1 = co.y ^ D2, x2 = co.z ^ D3, x3 = co.w ^ D4;
                            f = x0 ? (__builtin_ctz(x0) >> 3) : (x1 ? 4 + (__builtin_ctz(x1) >> 3) : (x2 ? 8 + (__builtin_ctz(x2) >> 3) : (x3 ? 12 + (__builtin_ctz(x3) >> 3) : 16)));
                            fa = 16;
                        } else {
                            const uint32_t x0 = co.x ^ D1;
                            f = x0 ? (__builtin_ctz(x0) >> 3) : 4;
                            fa = 4;
                        }
                        const int limit = blkEnd - s;
                        const bool done = f < fa || f >= limit;
                        const int fk = f < limit ? f : limit;
                        pk = (f >= 4 ? 1u : 0u) | (done ? 2u : 0u) | ((uint32_t)fk << 8);
                    }
                    pk = rdlane32(pk, 0);
                    if (pk & 1u) {
                        int l2 = (int)(pk >> 8);
                        if (!(pk & 2u)) l2 += wave_matchlen(base + s + l2, base + o2pos + l2, blkEnd - (s + l2), lane);
                        if (lane == 0) tab_put(h0, s, (uint32_t)cv);
                        KC_WAVE_SYNC();
                        emit(0, l2 - 3, 1u);
                        W = W0;
                        s += l2;
                        nextEmit = s;
                        const int tmp = o1; o1 = o2; o2 = tmp;
                        canRep = nseq > 2;
                        if (s >= sLimit) fin = true;
                        continue;  // the speculative probes of this round are dropped (nothing was committed)
                    }
                }
                LP(3);  // candidate loads issued, offset-2 verdict
                // steps of this round that share a bucket
                const bool lost = valid && (m0 != (uint32_t)lane || m1 != (uint32_t)lane);
                bool dep = lost;
                if (ballot64(lost) != 0) {  // rare
                    if (lane == 0) shareMask = 0ull;
                    KC_WAVE_SYNC();
                    if (lost) {
                        if (m0 != (uint32_t)lane) atomicOr(&shareMask, 1ull << m0);
                        if (m1 != (uint32_t)lane) atomicOr(&shareMask, 1ull << m1);
                    }
                    KC_WAVE_SYNC();
                    dep = lost || ((shareMask >> lane) & 1ull) != 0;
                }
                if (lane == 0) dep = false;  // the first step depends on nothing: every round commits at least one step
                // per-lane verdict: kind 1 repeat (s+2), 2 candidate at s, 3 candidate2 at s+1 (enc_fast.go:133,176,188)
                int kind = 0, t = 0, fwd = 0, back = 0, ba = 0, fa = 0, kofs = 0;
                if (repOk) {
                    int f, bk;
                    zf_cmp16(cr, __builtin_amdgcn_alignbyte(D1, D0, 2), __builtin_amdgcn_alignbyte(D2, D1, 2),
                             __builtin_amdgcn_alignbyte(D3, D2, 2), __builtin_amdgcn_alignbyte(D4, D3, 2), f, bk);
                    if (f >= 4) { kind = 1; fwd = repWide ? f : 4; back = repWide ? bk : 0; fa = repWide ? 12 : 4; ba = repWide ? ZL_BK : 0; kofs = 2; }
                }
                if (kind == 0 && ok0) {
                    int f, bk;
                    zf_cmp16(ca, D0, D1, D2, D3, f, bk);
                    if (f >= 4) { kind = 2; t = t0; fwd = wide0 ? f : 4; back = wide0 ? bk : 0; fa = wide0 ? 12 : 4; ba = wide0 ? ZL_BK : 0; kofs = 0; }
                }
                if (kind == 0 && ok1) {
                    int f, bk;
                    zf_cmp16(cb, __builtin_amdgcn_alignbyte(D1, D0, 1), __builtin_amdgcn_alignbyte(D2, D1, 1),
                             __builtin_amdgcn_alignbyte(D3, D2, 1), __builtin_amdgcn_alignbyte(D4, D3, 1), f, bk);
                    if (f >= 4) { kind = 3; t = t1; fwd = wide1 ? f : 4; back = wide1 ? bk : 0; fa = wide1 ? 12 : 4; ba = wide1 ? ZL_BK : 0; kofs = 1; }
                }
                uint32_t vk = 0;  // kind:2 | length final:1 | known forward length:5 | equal bytes behind:3 | bytes behind examined:3
                if (kind != 0) {
                    const int limit = blkEnd - (p + kofs);
                    const bool done = fwd < fa || fwd >= limit;
                    const int fk = fwd < limit ? fwd : limit;
                    vk = (uint32_t)kind | (done ? 4u : 0u) | ((uint32_t)fk << 3) | ((uint32_t)back << 8) | ((uint32_t)ba << 11);
                }
                LP(4);  // bucket sharing, per-lane verdicts (waits for the candidate bytes)
                const uint64_t vm = ballot64(valid);
                const uint64_t depm = ballot64(valid && dep);
                const uint64_t hm = ballot64(kind != 0);
                const int nvalid = __popcll(vm);
                const int c = depm ? ctz64(depm) : 64;
                const uint64_t hmc = c >= 64 ? hm : (hm & ((1ull << c) - 1ull));
                const bool found = hmc != 0;
                const int f = found ? ctz64(hmc) : 0;
                const int commitUpTo = found ? f : ((c < nvalid ? c : nvalid) - 1);
                if (valid && lane <= commitUpTo) {
                    tab_put(h0, p, (uint32_t)cv);
                    tab_put(h1, p + 1, (uint32_t)(cv >> 8));  // program order: wins when h0 == h1
                }
                KC_WAVE_SYNC();
                LP(5);  // ballots, commit
                if (!found) {
                    W = 2 * W < 64 ? 2 * W : 64;
                    if (c < nvalid) {
                        s = (int)rdlane32((uint32_t)p, c);
                    } else {
                        const int pl = (int)rdlane32((uint32_t)p, nvalid - 1);  // nvalid >= 1: s < sLimit inside the loop
                        s = pl + 2 + ((pl - nextEmit) >> SK);
                    }
                    if (s >= sLimit) fin = true;
                    continue;
                }
                const uint32_t wk = rdlane32(vk, f);
                const int mk = (int)(wk & 3u);
                const bool fdone = (wk & 4u) != 0;
                const int fk = (int)((wk >> 3) & 31u);
                const int bke = (int)((wk >> 8) & 7u), bav = (int)((wk >> 11) & 7u);
                const int ps = (int)rdlane32((uint32_t)p, f);
                int mt = (int)rdlane32((uint32_t)t, f);
                // backward extension given the bke equal bytes found among the bav bytes examined (enc_fast.go:152-157, 230-234)
                auto backlen = [&](int sp, int tp, int kmax) -> int {
                    if (kmax <= 0) return 0;
                    if (bke < bav) return bke < kmax ? bke : kmax;
                    if (kmax <= bav) return kmax;
                    return bav + wave_backlen(base, sp - bav, tp - bav, kmax - bav, lane);
                };
                if (mk == 1) {
                    const int rI = ps - o1 + 2;
                    int length = fk;
                    if (!fdone) length += wave_matchlen(base + ps + 2 + fk, base + rI + fk, blkEnd - (ps + 2 + fk), lane);
                    int start = ps + 2;
                    const int startLimit = nextEmit + 1;
                    const int sMin = (ps - mmo) > 0 ? (ps - mmo) : 0;
                    int kmax = rI - sMin;
                    if (start - startLimit < kmax) kmax = start - startLimit;
                    if (HIST) {
                        const int cap = (ZF_MAX_MATCH_LENGTH - 3) - (length - 3);
                        if (cap < kmax) kmax = cap;
                    }
                    const int bk = backlen(start, rI, kmax);
                    start -= bk;
                    emit(start - nextEmit, length - 3 + bk, 1u);
                    W = W0;
                    s = ps + length + 2;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                s = ps + (mk == 3 ? 1 : 0);
                o2 = o1;
                o1 = s - mt;
                int l = fk;
                if (!fdone) l += wave_matchlen(base + s + fk, base + mt + fk, blkEnd - (s + fk), lane);
                {
                    const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = mt - tMin;
                    if (s - nextEmit < kmax) kmax = s - nextEmit;
                    if (HIST && (ZF_MAX_MATCH_LENGTH - l) < kmax) kmax = ZF_MAX_MATCH_LENGTH - l;
                    const int bk = backlen(s, mt, kmax);
                    s -= bk;
                    mt -= bk;
                    l += bk;
                }
                emit(s - nextEmit, l - 3, (uint32_t)(s - mt) + 3u);
                W = W0;
                s += l;
                nextEmit = s;
                const bool canRepO2 = HIST ? canRep : (nseq > 2);
                canRep = nseq > 2;
                if (s >= sLimit) { fin = true; continue; }
                pendO2 = canRepO2;
            }
        }
        pend = false;  // a refill still in flight at the end of a block is dropped; the window itself stays valid
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
        const int o1c = o1, o2c = o2;
        if (flags) { o1 = o1_in; o2 = o2_in; }
        flags |= rounds << 8;
        if (lane == 0) {
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
    LP_FLUSH(P.prof);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// kc_zfast_match_lds2_kernel — the "fused step" form of the kernel above for units of at most 128 KiB without history (one EncodeAll of
// up to two blocks: the latency case), same sequences.  What changes is what sits on the critical path of ONE wave (one instruction
// per ~4.7 clocks, an LDS round trip ~150 of them, a global one ~950):
//  * the SOURCE is in LDS too: the table shrinks to 2^15 x 16 bits + one bit per entry (position+1 below 2^17: the low half in a u16,
//    bit 16 in a 4 KiB bitmap that is only touched once positions pass 64 KiB) = 68 KiB, which leaves a 64 KiB ring (+ mirror) for the
//    source: every probe, candidate and extension byte of the first block, and of the second block unless the candidate lies more than
//    ~60 KiB back (then, and for matches longer than the lanes hold, the bytes come through L2 like before);
//  * without the tag (no room for it) every candidate is verified on its bytes — an LDS read now, and the same read gives both
//    extensions: a probe step is two dependent LDS trips.  Trip 1: the lanes of groups 0 / 1 hash the bytes at s / s+1 (their own
//    loads, issued a step ahead) and read their buckets; with an offset-2 test pending (enc_fast.go:250) all 64 lanes also compare
//    8 bytes each at s and s - offset2.  Then lanes 0 and 16 store s and s+1.  Trip 2: three 16-lane groups — candidate at s, candidate
//    at s+1, repeat at s+2 — load 8 bytes per lane on both sides: lane 0 of a group the 4 bytes before and at the candidate (backward
//    extension, verification), lanes 1..15 the 120 bytes behind (matchlen).  One ballot holds the verdicts and the lengths; the
//    reference's priority (repeat, candidate, candidate2; :133,176,188) picks the winner in scalar code.
#define ZF2_MAX_UNIT 131072
#define ZF2_RING 65536
#define ZF2_MIRROR 576   // the first bytes of the ring again behind it: reads of up to 63 x 8 + 8 bytes from any position never wrap
#define ZF2_FILL 1024    // bytes per refill (16 per lane)
__global__ __launch_bounds__(64) void kc_zfast_match_lds2_kernel(KcMatchParams P, uint32_t n_launch) {
    __shared__ uint16_t tab[1 << ZF_TABLE_BITS];          // (position + 1) & 0xFFFF, 0 with a clear bit below = empty
    __shared__ uint32_t hib[(1 << ZF_TABLE_BITS) / 32];   // bit 16 of position + 1
    __shared__ __attribute__((aligned(16))) uint8_t ring[ZF2_RING + ZF2_MIRROR];
    __shared__ uint64_t sbuf[64];
    __shared__ uint32_t sink[64];  // where the lanes without a table store of their own write (one store instruction, no exec masking)
    const int lane = (int)threadIdx.x;
    const uint32_t ui = blockIdx.x;
    if (ui >= n_launch) return;
    const uint32_t u = P.unit_list ? P.unit_list[ui] : P.unit_base + ui;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int boff = (int)((uintptr_t)base & 15);  // ring positions are relative to the 16-byte aligned abase
    const uint8_t* __restrict__ abase = base - boff;
    const int ulen = (int)(P.unit_off[u + 1] - P.unit_off[u]);
    if (ulen > ZF2_MAX_UNIT) return;  // the first form's unit
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const int mmo = P.max_match_off;
    const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
    const int nblk = (P.unit_done != nullptr && P.unit_done[u] != 0u) ? 0 : UB.nblk;
    const bool HIST = ulen > bs || UB.streamU;
    const uint8_t* const srcHi = P.src_end;
    for (int i = lane * 8; i < (1 << ZF_TABLE_BITS); i += 512) *(uint4*)&tab[i] = make_uint4(0, 0, 0, 0);
    for (int i = lane; i < (1 << ZF_TABLE_BITS) / 32; i += 64) hib[i] = 0u;
    KC_WAVE_SYNC();

    const int g4 = lane >> 4, j16 = lane & 15;
    const int off16 = j16 == 0 ? -4 : 8 * j16 - 4;  // a lane's 8 bytes inside its 16-lane group: [-4, +4) then [+4, +124) from the position
    const int pg = g4 < 2 ? g4 : 2;                  // groups 0 / 1: the candidates of s / s+1; group 2 (and 3, idle): the repeat candidate of s+2
    const int soff16 = pg + off16;
    const int off64 = lane == 0 ? -4 : 8 * lane - 4; // the offset-2 test: one 64-lane group, [+4, +508)
    const int q = g4 == 1 ? 1 : 0;                   // group 1 hashes the bytes at s+1, everybody else those at s
    const uint32_t hiOnly = j16 == 0 ? 0u : ~0u;     // a group's lane 0 verifies on the upper four bytes of its difference only
    const uint32_t hiOnly64 = lane == 0 ? 0u : ~0u;
    uint32_t* const sinkL = &sink[lane];
    uint16_t* const sink16 = (uint16_t*)sinkL;

    // ---- the source ring: abase[wlo, whi) at ring[(x) & (ZF2_RING - 1)] ----
    const int alen = boff + ulen;
    int wlo = 0, whi = 0;
    bool pend = false;
    uint4 rf = make_uint4(0, 0, 0, 0);
    auto ring_store = [&](int at, const uint4 v) {
        const int ro = (at + 16 * lane) & (ZF2_RING - 1);
        *(uint4*)(ring + ro) = v;
        if (ro < ZF2_MIRROR) *(uint4*)(ring + ZF2_RING + ro) = v;
    };
    auto gload16 = [&](int at) -> uint4 {
        const uint8_t* qq = abase + at + 16 * lane;
        return qq < srcHi ? *(const uint4*)qq : make_uint4(0, 0, 0, 0);  // aligned: never leaves the 16-byte granule of a readable byte
    };
    auto fill_to = [&](int upto) {  // blocking: until whi >= upto (or the unit's end), four refills per round trip
        if (pend) { ring_store(whi, rf); whi += ZF2_FILL; pend = false; }
        while (whi < upto && whi < alen) {
            const uint4 v0 = gload16(whi), v1 = gload16(whi + ZF2_FILL), v2 = gload16(whi + 2 * ZF2_FILL), v3 = gload16(whi + 3 * ZF2_FILL);
            ring_store(whi, v0); ring_store(whi + ZF2_FILL, v1); ring_store(whi + 2 * ZF2_FILL, v2); ring_store(whi + 3 * ZF2_FILL, v3);
            whi += 4 * ZF2_FILL;
        }
        if (whi - wlo > ZF2_RING) wlo = whi - ZF2_RING;
        KC_WAVE_SYNC();
    };
    auto rd64r = [&](int pos) -> uint64_t { return ld64(ring + ((pos + boff) & (ZF2_RING - 1))); };
    // 8 bytes of a candidate chunk at unit position c + off (off = -4: the four bytes before c and the four at c), from the ring, or
    // through L2 where the ring no longer holds them; positions before the unit's start read as zero (the callers never count them)
    auto cand8 = [&](int c, int off, bool inRing) -> uint64_t {
        if (inRing) return rd64r(c + off);
        if (off < 0) {
            const uint32_t hi = ld32(base + c);
            const uint32_t lo = c >= 4 ? ld32(base + c - 4) : (c > 0 ? ld32(base) << (8 * (4 - c)) : 0u);
            return (uint64_t)lo | ((uint64_t)hi << 32);
        }
        const uint8_t* qq = base + c + off;
        return qq + 8 <= srcHi ? ld64(qq) : 0ull;  // (past the end of the buffer: beyond the block's end too, the length is capped there)
    };

    int o1 = P.rep1, o2 = P.rep2;
    for (int b = 0; b < nblk; b++) {
        const int blkStart = kc_blk_begin(P.blk_start, blk0, b, bs);
        const int blkEnd = kc_blk_end(P.blk_start, blk0, b, nblk, bs, ulen);
        const int srcLen = blkEnd - blkStart;
        const int o1_in = o1, o2_in = o2;
        uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        int nseq = 0, sumLL = 0;
        uint32_t rounds = 0;
        int nextEmit = blkStart, s = blkStart;
        uint32_t firstLL = 0, firstOf = 0;
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
        const int SK = 5;  // kSearchStrength - 1
        if (srcLen >= 10) {
            const int sLimit = blkEnd - 8;
            bool canRep = false, fin = false, pendO2 = false;
            if (whi - (s + boff) < 4 * ZF2_FILL && whi < alen) fill_to(s + boff + 12 * ZF2_FILL);
            uint64_t cvL = rd64r(s + q);
            while (!fin) {
                if (++rounds > (uint32_t)srcLen + 16u) break;  // every step advances s: cannot happen; never spin on the device
                // ---------------- source window ----------------
                if (pend) {  // the refill issued one step ago has landed
                    ring_store(whi, rf);
                    whi += ZF2_FILL;
                    if (whi - wlo > ZF2_RING) wlo = whi - ZF2_RING;
                    pend = false;
                    KC_WAVE_SYNC();
                }
                if (whi < alen) {
                    const int ahead = whi - (s + boff);
                    if (ahead < 2 * ZF2_FILL) {  // a long match ran past what was buffered: restart the ring just behind s, or catch up
                        if (ahead < 0) { wlo = whi = (s + boff - 64) & ~15; if (wlo < 0) wlo = whi = 0; }
                        fill_to(s + boff + 12 * ZF2_FILL);
                        cvL = rd64r(s + q);
                    } else if (ahead < 6 * ZF2_FILL) {
                        rf = gload16(whi);
                        pend = true;
                    }
                }
                // ---------------- trip 1: the table (and the offset-2 test, enc_fast.go:250) ----------------
                const int nextS = s + 2 + ((s - nextEmit) >> SK);
                const bool hiMode = s + 2 >= 65536;  // positions + 1 with bit 16: the bitmap is live
                const uint32_t hL = hash6(cvL, ZF_TABLE_BITS);
                uint32_t e = tab[hL];
                if (hiMode) e |= ((hib[hL >> 5] >> (hL & 31u)) & 1u) << 16;
                const uint64_t cvN = rd64r(nextS + q);  // the next step's bytes travel with this step's table entries
                if (pendO2) {
                    pendO2 = false;
                    const int o2pos = s - o2;
                    const bool inR = wlo == 0 || o2pos + boff - 4 >= wlo;
                    const uint64_t dO = cand8(o2pos, off64, inR) ^ rd64r(s + off64);
                    const uint64_t BO = ballot64((((uint32_t)dO & hiOnly64) | (uint32_t)(dO >> 32)) != 0u);
                    if (!(BO & 1ull)) {  // four equal bytes at s and s - offset2
                        const uint64_t fw = BO >> 1;
                        int M;
                        if (fw != 0ull) M = s + 4 + 8 * ctz64(fw) + (ctz64(rdlane64(dO, 1 + ctz64(fw))) >> 3);
                        else if (s + 508 >= blkEnd) M = blkEnd;
                        else M = s + 508 + wave_matchlen(base + s + 508, base + o2pos + 508, blkEnd - (s + 508), lane);
                        const int l2 = (M < blkEnd ? M : blkEnd) - s;
                        KC_WAVE_SYNC();
                        *(lane == 0 ? &tab[hL] : sink16) = (uint16_t)(s + 1);  // table[hash(cv)] = s (:256)
                        if (s + 1 >= 65536) atomicOr(lane == 0 ? &hib[hL >> 5] : sinkL, 1u << (hL & 31u));
                        KC_WAVE_SYNC();
                        emit(0, l2 - 3, 1u);
                        s += l2;
                        nextEmit = s;
                        const int tmp = o1; o1 = o2; o2 = tmp;
                        canRep = nseq > 2;
                        if (s >= sLimit) fin = true;
                        else cvL = rd64r(s + q);
                        continue;
                    }
                }
                KC_WAVE_SYNC();
                *(lane == 0 ? &tab[hL] : sink16) = (uint16_t)(s + 1);    // table[nextHash] = s
                KC_WAVE_SYNC();
                *(lane == 16 ? &tab[hL] : sink16) = (uint16_t)(s + 2);   // table[nextHash2] = s + 1 (behind the first store: one bucket for both keeps s + 1)
                if (hiMode) {
                    atomicOr((lane == 0 && s + 1 >= 65536) || lane == 16 ? &hib[hL >> 5] : sinkL, 1u << (hL & 31u));
                }
                KC_WAVE_SYNC();
                // ---------------- trip 2: candidate at s, candidate2 at s+1, repeat at s+2 — verification and both extensions ----------------
                const int repIndex = s - o1 + 2;
                const int tL = (int)e - 1;
                const int cL = g4 < 2 ? tL : repIndex;
                const bool okL = g4 == 0 ? (e != 0u && (s - tL) < mmo) : (g4 == 1 ? (e != 0u && (s - tL + 1) < mmo) : (g4 == 2 && canRep && repIndex >= 0));
                const bool inRL = wlo == 0 || cL + boff - 4 >= wlo;
                uint64_t diff = ~0ull;
                if (okL) diff = cand8(cL, off16, inRL) ^ rd64r(s + soff16);
                const uint64_t B = ballot64((((uint32_t)diff & hiOnly) | (uint32_t)(diff >> 32)) != 0u);
                if ((~B & 0x0000000100010001ull) == 0ull) {  // no candidate verified
                    s = nextS;
                    cvL = cvN;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                int kind, gs;  // 1 repeat at s+2, 2 candidate at s, 3 candidate2 at s+1 — the reference's order (:133, 176, 188)
                if (!((B >> 32) & 1ull)) { kind = 1; gs = 2; }
                else if (!(B & 1ull)) { kind = 2; gs = 0; }
                else { kind = 3; gs = 1; }
                const int gb = 16 * gs;
                const int p = s + gs;              // (group g sits at s + g for all three)
                int mt = (int)rdlane32((uint32_t)cL, gb);
                // forward: the exact common prefix up to the block's end (matchlen, enc_base.go:117-131)
                int mEnd;
                {
                    const uint32_t fwd = ((uint32_t)(B >> gb) >> 1) & 0x7FFFu;
                    if (fwd != 0u) {
                        const int f = __builtin_ctz(fwd);
                        mEnd = p + 4 + 8 * f + (ctz64(rdlane64(diff, gb + 1 + f)) >> 3);
                    } else if (p + 124 >= blkEnd) mEnd = blkEnd;
                    else mEnd = p + 124 + wave_matchlen(base + p + 124, base + mt + 124, blkEnd - (p + 124), lane);
                    if (mEnd > blkEnd) mEnd = blkEnd;
                }
                const uint32_t dlo = rdlane32((uint32_t)diff, gb);
                const int nb = dlo == 0u ? 4 : (__builtin_clz(dlo) >> 3);  // equal bytes going down from p-1 / mt-1, of the 4 the lane holds
                auto backlen = [&](int kmax) -> int {
                    if (kmax <= 0) return 0;
                    if (nb < 4 || kmax <= 4) return nb < kmax ? nb : kmax;
                    return 4 + wave_backlen(base, p - 4, mt - 4, kmax - 4, lane);
                };
                if (kind == 1) {
                    // ---------------- repeat at s+2 (:133-173) ----------------
                    const int length = mEnd - p;
                    const int sMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = mt - sMin;
                    if (p - (nextEmit + 1) < kmax) kmax = p - (nextEmit + 1);
                    if (HIST) {
                        const int cap = (ZF_MAX_MATCH_LENGTH - 3) - (length - 3);
                        if (cap < kmax) kmax = cap;
                    }
                    const int bk = backlen(kmax);
                    emit(p - bk - nextEmit, length - 3 + bk, 1u);
                    s = p + length;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    else cvL = rd64r(s + q);
                    continue;
                }
                // ---------------- candidate / candidate2 (:176-247) ----------------
                o2 = o1;
                o1 = p - mt;
                int l = mEnd - p;
                int ms = p;
                {
                    const int tMin = (p - mmo) > 0 ? (p - mmo) : 0;
                    int kmax = mt - tMin;
                    if (p - nextEmit < kmax) kmax = p - nextEmit;
                    if (HIST && (ZF_MAX_MATCH_LENGTH - l) < kmax) kmax = ZF_MAX_MATCH_LENGTH - l;
                    const int bk = backlen(kmax);
                    ms -= bk;
                    mt -= bk;
                    l += bk;
                }
                emit(ms - nextEmit, l - 3, (uint32_t)(ms - mt) + 3u);
                s = ms + l;
                nextEmit = s;
                const bool canRepO2 = HIST ? canRep : (nseq > 2);
                canRep = nseq > 2;
                if (s >= sLimit) { fin = true; continue; }
                pendO2 = canRepO2;
                cvL = rd64r(s + q);
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
        const int o1c = o1, o2c = o2;
        if (flags) { o1 = o1_in; o2 = o2_in; }
        flags |= rounds << 8;
        if (lane == 0) {
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


// ---------------------------------------------------------------------------------------------------------------------------------
// kc_zfast_match_lds3_kernel — four probe steps per round in the fused layout.  The lone wave is issue-bound (~4.7 clocks per
// instruction, whatever it is), so the round is built for the fewest instructions per step, not for the fewest lanes:
//  * 16 lanes per step (step k = lanes 16k..16k+15), three groups inside: candidate at s_k (5 lanes), candidate2 at s_k+1 (5 lanes),
//    repeat at s_k+2 (6 lanes); every lane hashes its own group's position — one hash, one table read, two chunk loads and one compare
//    instruction stream serve all twelve candidates of the round;
//  * the table accesses of the four steps are ISSUED in the sequential encoder's order — read_k, store s_k, store s_k+1, k = 0..3 —
//    and LDS runs one wave's instructions in order, so step k sees what the steps before it wrote with no conflict detection at all;
//    the stores are speculative: every lane keeps what it read, and the steps behind the first hit are undone by storing those values
//    back, last step first (a handful of stores, only in rounds that end on a hit before the fourth step);
//  * one ballot carries the twelve verdicts and the forward lengths (36 / 44 bytes per candidate; longer matches take the generic
//    matchlen), the reference's order — first step, then repeat, candidate, candidate2 (enc_fast.go:133, 176, 188) — picks the winner.
// Units up to 128 KiB without history, window not smaller than the unit; table and source ring as in kc_zfast_match_lds2_kernel.
#define ZF3_K 4
__global__ __launch_bounds__(64) void kc_zfast_match_lds3_kernel(KcMatchParams P, uint32_t n_launch) {
    __shared__ uint16_t tab[1 << ZF_TABLE_BITS];          // (position + 1) & 0xFFFF
    __shared__ uint32_t hib[(1 << ZF_TABLE_BITS) / 32];   // bit 16 of position + 1
    __shared__ __attribute__((aligned(16))) uint8_t ring[ZF2_RING + ZF2_MIRROR];
    __shared__ uint64_t sbuf[64];
    __shared__ uint32_t sink[64];
    const int lane = (int)threadIdx.x;
    const uint32_t ui = blockIdx.x;
    if (ui >= n_launch) return;
    const uint32_t u = P.unit_list ? P.unit_list[ui] : P.unit_base + ui;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int boff = (int)((uintptr_t)base & 15);
    const uint8_t* __restrict__ abase = base - boff;
    const int ulen = (int)(P.unit_off[u + 1] - P.unit_off[u]);
    const int mmo = P.max_match_off;
    if (ulen > ZF2_MAX_UNIT) return;  // the first form's unit (the launcher sends windows below 128 KiB there too: every offset of these units is inside the window)
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
    const int nblk = (P.unit_done != nullptr && P.unit_done[u] != 0u) ? 0 : UB.nblk;
    const bool HIST = ulen > bs || UB.streamU;
    const uint8_t* const srcHi = P.src_end;
    for (int i = lane * 8; i < (1 << ZF_TABLE_BITS); i += 512) *(uint4*)&tab[i] = make_uint4(0, 0, 0, 0);
    for (int i = lane; i < (1 << ZF_TABLE_BITS) / 32; i += 64) hib[i] = 0u;
    KC_WAVE_SYNC();

    const int k4 = lane >> 4, j = lane & 15;
    const int gi = j >= 10 ? 2 : (j >= 5 ? 1 : 0);  // 0: candidate at s_k, 1: candidate2 at s_k+1, 2: repeat at s_k+2
    const int jj = j - 5 * gi;                       // 0: the 4 bytes before and at the position; 1..: the 8-byte chunks behind them
    const int off = jj == 0 ? -4 : 8 * jj - 4;
    const int soff = gi + off;
    const int hq = gi == 1 ? 1 : 0;                  // group 1 hashes the bytes at s_k + 1, the others those at s_k
    const uint32_t hiOnly = jj == 0 ? 0u : ~0u;
    const bool own0 = j == 0, own1 = j == 5;        // the lanes that store s_k / s_k + 1
    const int off64 = lane == 0 ? -4 : 8 * lane - 4;
    const uint32_t hiOnly64 = lane == 0 ? 0u : ~0u;
    uint32_t* const sinkL = &sink[lane];
    uint16_t* const sink16 = (uint16_t*)sinkL;
    const uint64_t VER = 0x0421042104210421ull;      // the verification lanes: bits 0 / 5 / 10 of every step
    const uint64_t REP = 0x0400040004000400ull;

    const int alen = boff + ulen;
    int wlo = 0, whi = 0;
    bool pend = false;
    uint4 rf = make_uint4(0, 0, 0, 0);
    auto ring_store = [&](int at, const uint4 v) {
        const int ro = (at + 16 * lane) & (ZF2_RING - 1);
        *(uint4*)(ring + ro) = v;
        if (ro < ZF2_MIRROR) *(uint4*)(ring + ZF2_RING + ro) = v;
    };
    auto gload16 = [&](int at) -> uint4 {
        const uint8_t* qq = abase + at + 16 * lane;
        return qq < srcHi ? *(const uint4*)qq : make_uint4(0, 0, 0, 0);
    };
    auto fill_to = [&](int upto) {
        if (pend) { ring_store(whi, rf); whi += ZF2_FILL; pend = false; }
        while (whi < upto && whi < alen) {
            const uint4 v0 = gload16(whi), v1 = gload16(whi + ZF2_FILL), v2 = gload16(whi + 2 * ZF2_FILL), v3 = gload16(whi + 3 * ZF2_FILL);
            ring_store(whi, v0); ring_store(whi + ZF2_FILL, v1); ring_store(whi + 2 * ZF2_FILL, v2); ring_store(whi + 3 * ZF2_FILL, v3);
            whi += 4 * ZF2_FILL;
        }
        if (whi - wlo > ZF2_RING) wlo = whi - ZF2_RING;
        KC_WAVE_SYNC();
    };
    auto rd64r = [&](int pos) -> uint64_t { return ld64(ring + ((pos + boff) & (ZF2_RING - 1))); };
    // a candidate chunk the ring no longer holds (second half of a unit, candidate more than ~60 KiB back): through L2
    auto cand8g = [&](int c, int o) -> uint64_t {
        if (o < 0) {
            const uint32_t hi = ld32(base + c);
            const uint32_t lo = c >= 4 ? ld32(base + c - 4) : (c > 0 ? ld32(base) << (8 * (4 - c)) : 0u);
            return (uint64_t)lo | ((uint64_t)hi << 32);
        }
        const uint8_t* qq = base + c + o;
        return qq + 8 <= srcHi ? ld64(qq) : 0ull;
    };

    int o1 = P.rep1, o2 = P.rep2;
    for (int b = 0; b < nblk; b++) {
        const int blkStart = kc_blk_begin(P.blk_start, blk0, b, bs);
        const int blkEnd = kc_blk_end(P.blk_start, blk0, b, nblk, bs, ulen);
        const int srcLen = blkEnd - blkStart;
        const int o1_in = o1, o2_in = o2;
        uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        int nseq = 0, sumLL = 0;
        uint32_t rounds = 0;
        int nextEmit = blkStart, s = blkStart;
        uint32_t firstLL = 0, firstOf = 0;
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
        if (srcLen >= 10) {
            const int sLimit = blkEnd - 8;
            bool canRep = false, fin = false, pendO2 = false, haveCv = false;
            if (whi - (s + boff) < 4 * ZF2_FILL && whi < alen) fill_to(s + boff + 12 * ZF2_FILL);
            uint64_t cvL = 0;
            while (!fin) {
                if (++rounds > (uint32_t)srcLen + 16u) break;  // every round advances s: cannot happen; never spin on the device
                // ---------------- source window ----------------
                if (pend) {
                    ring_store(whi, rf);
                    whi += ZF2_FILL;
                    if (whi - wlo > ZF2_RING) wlo = whi - ZF2_RING;
                    pend = false;
                    KC_WAVE_SYNC();
                }
                // ---------------- the round's positions: s_0 .. s_3 (and s_4 .. s_7 for the bytes of the next round) ----------------
                int sk[2 * ZF3_K];
                sk[0] = s;
                for (int k = 1; k < 2 * ZF3_K; k++) sk[k] = sk[k - 1] + 2 + ((sk[k - 1] - nextEmit) >> 5);
                int nv = 1;  // steps of this round inside the block (s_0 < sLimit holds)
                for (int k = 1; k < ZF3_K; k++) if (sk[k] < sLimit) nv = k + 1;
                if (whi < alen) {  // (long literal runs take long steps: what the round reads ahead is measured from its last step)
                    const int need = sk[nv - 1] + boff + 64;
                    const int ahead = whi - need;
                    if (ahead < ZF2_FILL) {
                        if (whi < s + boff) { wlo = whi = (s + boff - 64) & ~15; if (wlo < 0) wlo = whi = 0; }
                        fill_to(need + 8 * ZF2_FILL);
                        haveCv = false;
                    } else if (ahead < 5 * ZF2_FILL) {
                        rf = gload16(whi);
                        pend = true;
                    }
                }
                const bool prefOk = whi >= alen || sk[2 * ZF3_K - 1] + boff + 16 <= whi;  // the next round's bytes are in the ring already
                const int skL = k4 == 0 ? sk[0] : (k4 == 1 ? sk[1] : (k4 == 2 ? sk[2] : sk[3]));
                if (!haveCv) cvL = rd64r(skL + hq);
                const uint64_t cvN = rd64r((k4 == 0 ? sk[4] : (k4 == 1 ? sk[5] : (k4 == 2 ? sk[6] : sk[7]))) + hq);
                // ---------------- trip 1: the table, in the sequential encoder's order ----------------
                const bool hiMode = sk[ZF3_K - 1] + 2 >= 65536;
                const uint32_t hL = hash6(cvL, ZF_TABLE_BITS);
                uint16_t* const tp = &tab[hL];
                const uint32_t* const tp32 = (const uint32_t*)&tab[hL & ~1u];  // the entry is read as half of an aligned word and cut out behind the last store
                uint32_t* const hp = &hib[hL >> 5];
                const uint32_t hbit = 1u << (hL & 31u);
                const uint32_t myVal = (uint32_t)(skL + hq + 1);  // what this lane's group stores: position + 1
                uint64_t dO = 0;
                const bool doO2 = pendO2;
                const int o2pos = s - o2;
                if (doO2) {
                    const bool inR = wlo == 0 || o2pos + boff - 4 >= wlo;
                    dO = (inR ? rd64r(o2pos + off64) : cand8g(o2pos, off64)) ^ rd64r(s + off64);
                }
                uint32_t rr[ZF3_K] = {0, 0, 0, 0}, rh[ZF3_K] = {0, 0, 0, 0};  // what the steps read (used only behind the last store: one wait for all)
#pragma unroll
                for (int k = 0; k < ZF3_K; k++) {
                    if (k < nv) {
                        rr[k] = *tp32;
                        if (hiMode) rh[k] = *hp;
                        KC_WAVE_SYNC();
                        *((k4 == k && own0) ? tp : sink16) = (uint16_t)myVal;   // table[nextHash] = s_k
                        if (hiMode && sk[k] + 1 >= 65536) atomicOr((k4 == k && own0) ? hp : sinkL, hbit);
                        KC_WAVE_SYNC();
                        *((k4 == k && own1) ? tp : sink16) = (uint16_t)myVal;   // table[nextHash2] = s_k + 1
                        if (hiMode && sk[k] + 2 >= 65536) atomicOr((k4 == k && own1) ? hp : sinkL, hbit);
                        KC_WAVE_SYNC();
                    }
                }
                uint32_t eOwn = ((k4 == 0 ? rr[0] : (k4 == 1 ? rr[1] : (k4 == 2 ? rr[2] : rr[3]))) >> ((hL & 1u) << 4)) & 0xFFFFu;
                if (hiMode) eOwn |= (((k4 == 0 ? rh[0] : (k4 == 1 ? rh[1] : (k4 == 2 ? rh[2] : rh[3]))) & hbit) != 0u ? 1u : 0u) << 16;
                // undo the table stores of steps [from, nv), last first (every lane kept what it read in front of its step's stores)
                auto undo = [&](int from) {
                    for (int k = nv - 1; k >= from; k--) {
                        KC_WAVE_SYNC();
                        *((k4 == k && own1) ? tp : sink16) = (uint16_t)eOwn;
                        if (hiMode && !(eOwn >> 16)) atomicAnd((k4 == k && own1) ? hp : sinkL, ~hbit);
                        KC_WAVE_SYNC();
                        *((k4 == k && own0) ? tp : sink16) = (uint16_t)eOwn;
                        if (hiMode && !(eOwn >> 16)) atomicAnd((k4 == k && own0) ? hp : sinkL, ~hbit);
                        KC_WAVE_SYNC();
                    }
                };
                if (doO2) {
                    pendO2 = false;
                    const uint64_t BO = ballot64((((uint32_t)dO & hiOnly64) | (uint32_t)(dO >> 32)) != 0u);
                    if (!(BO & 1ull)) {  // four equal bytes at s and s - offset2 (enc_fast.go:250): only table[hash(cv)] = s stays
                        undo(1);
                        {   // ... and step 0's second store: back to what it read — or, where both stores hit one bucket, to the first store's s + 1
                            const bool same = rdlane32(hL, 0) == rdlane32(hL, 5);
                            const uint32_t rv = same ? (uint32_t)(s + 1) : eOwn;
                            KC_WAVE_SYNC();
                            *((lane == 5) ? tp : sink16) = (uint16_t)rv;
                            if (hiMode && !(rv >> 16)) atomicAnd((lane == 5) ? hp : sinkL, ~hbit);
                            KC_WAVE_SYNC();
                        }
                        const uint64_t fw = BO >> 1;
                        int M;
                        if (fw != 0ull) M = s + 4 + 8 * ctz64(fw) + (ctz64(rdlane64(dO, 1 + ctz64(fw))) >> 3);
                        else if (s + 508 >= blkEnd) M = blkEnd;
                        else M = s + 508 + wave_matchlen(base + s + 508, base + o2pos + 508, blkEnd - (s + 508), lane);
                        const int l2 = (M < blkEnd ? M : blkEnd) - s;
                        emit(0, l2 - 3, 1u);
                        s += l2;
                        nextEmit = s;
                        const int tmp = o1; o1 = o2; o2 = tmp;
                        canRep = nseq > 2;
                        haveCv = false;
                        if (s >= sLimit) fin = true;
                        continue;
                    }
                }
                // ---------------- trip 2: the twelve candidates — verification and both extensions ----------------
                const int cL = gi == 2 ? skL - o1 + 2 : (int)eOwn - 1;   // (an empty entry gives -1: not a candidate)
                const bool inRL = wlo == 0 || cL + boff - 4 >= wlo;
                uint64_t diff = (inRL ? rd64r(cL + off) : cand8g(cL < 0 ? 0 : cL, off)) ^ rd64r(skL + soff);
                const uint32_t bad = cL < 0 ? ~0u : 0u;
                const uint64_t B = ballot64(((((uint32_t)diff & hiOnly) | (uint32_t)(diff >> 32)) | bad) != 0u);
                uint64_t m = ~B & VER;
                if (nv < ZF3_K) m &= (1ull << (16 * nv)) - 1ull;
                if (!canRep) m &= ~REP;
                if (m == 0ull) {  // no candidate verified: all steps of the round stand
                    s = sk[nv];   // (nv < 4: the step behind the last one is at or past sLimit)
                    cvL = cvN;
                    haveCv = prefOk;
                    if (nv < ZF3_K || s >= sLimit) fin = true;
                    continue;
                }
                const int ks = ctz64(m) >> 4;
                undo(ks + 1);
                const uint32_t vb = (uint32_t)(m >> (16 * ks)) & 0x421u;
                int kind, g;  // 1 repeat at s+2, 2 candidate at s, 3 candidate2 at s+1 — the reference's order (:133, 176, 188)
                if (vb & 0x400u) { kind = 1; g = 2; }
                else if (vb & 1u) { kind = 2; g = 0; }
                else { kind = 3; g = 1; }
                const int sx = ks == 0 ? sk[0] : (ks == 1 ? sk[1] : (ks == 2 ? sk[2] : sk[3]));
                const int gb = 16 * ks + 5 * g;
                const int p = sx + g;
                int mt = (int)rdlane32((uint32_t)cL, gb);
                int mEnd;
                {
                    const int nch = g == 2 ? 5 : 4;  // forward chunks the group holds
                    const uint32_t fwd = ((uint32_t)(B >> (gb + 1))) & ((1u << nch) - 1u);
                    const int span = 4 + 8 * nch;
                    if (fwd != 0u) {
                        const int f = __builtin_ctz(fwd);
                        mEnd = p + 4 + 8 * f + (ctz64(rdlane64(diff, gb + 1 + f)) >> 3);
                    } else if (p + span >= blkEnd) mEnd = blkEnd;
                    else mEnd = p + span + wave_matchlen(base + p + span, base + mt + span, blkEnd - (p + span), lane);
                    if (mEnd > blkEnd) mEnd = blkEnd;
                }
                const uint32_t dlo = rdlane32((uint32_t)diff, gb);
                const int nb = dlo == 0u ? 4 : (__builtin_clz(dlo) >> 3);
                auto backlen = [&](int kmax) -> int {
                    if (kmax <= 0) return 0;
                    if (nb < 4 || kmax <= 4) return nb < kmax ? nb : kmax;
                    return 4 + wave_backlen(base, p - 4, mt - 4, kmax - 4, lane);
                };
                haveCv = false;
                if (kind == 1) {
                    // ---------------- repeat at s+2 (:133-173) ----------------
                    const int length = mEnd - p;
                    const int sMin = (sx - mmo) > 0 ? (sx - mmo) : 0;
                    int kmax = mt - sMin;
                    if (p - (nextEmit + 1) < kmax) kmax = p - (nextEmit + 1);
                    if (HIST) {
                        const int cap = (ZF_MAX_MATCH_LENGTH - 3) - (length - 3);
                        if (cap < kmax) kmax = cap;
                    }
                    const int bk = backlen(kmax);
                    emit(p - bk - nextEmit, length - 3 + bk, 1u);
                    s = p + length;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                // ---------------- candidate / candidate2 (:176-247) ----------------
                o2 = o1;
                o1 = p - mt;
                int l = mEnd - p;
                int ms = p;
                {
                    const int tMin = (p - mmo) > 0 ? (p - mmo) : 0;
                    int kmax = mt - tMin;
                    if (p - nextEmit < kmax) kmax = p - nextEmit;
                    if (HIST && (ZF_MAX_MATCH_LENGTH - l) < kmax) kmax = ZF_MAX_MATCH_LENGTH - l;
                    const int bk = backlen(kmax);
                    ms -= bk;
                    mt -= bk;
                    l += bk;
                }
                emit(ms - nextEmit, l - 3, (uint32_t)(ms - mt) + 3u);
                s = ms + l;
                nextEmit = s;
                const bool canRepO2 = HIST ? canRep : (nseq > 2);
                canRep = nseq > 2;
                if (s >= sLimit) { fin = true; continue; }
                pendO2 = canRepO2;
            }
        }
        KC_WAVE_SYNC();
        if (lane < (nseq & 63)) sq[(nseq & ~63) + lane] = sbuf[lane];
        KC_WAVE_SYNC();
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
        if (lane == 0) {
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

void kc_launch_zfast_match_lds(const KcMatchParams& P, const uint32_t* proto, uint32_t proto_stride, uint32_t n_launch, hipStream_t st) {
    if (n_launch == 0) return;
    // spec_w0 > 0: the first form for every unit at that width.  Units up to 128 KiB without history can take another kernel (the other
    // units of the launch stay with the first form at width 16): 0 the four-step fused kernel, -1 the single-step fused kernel, -2 the
    // first form's instantiation with the source ring
    const bool eligible = P.spec_w0 <= 0 && proto == nullptr && P.hist0 == 0 && P.unit_hist == nullptr && P.job_flags == nullptr && P.max_match_off >= 131072;
    KcMatchParams Q = P;
    if (Q.spec_w0 <= 0) Q.spec_w0 = 16;
    if (eligible && P.spec_w0 == 0) hipLaunchKernelGGL(kc_zfast_match_lds3_kernel, dim3(n_launch), dim3(64), 0, st, P, n_launch);
    if (eligible && P.spec_w0 == -1) hipLaunchKernelGGL(kc_zfast_match_lds2_kernel, dim3(n_launch), dim3(64), 0, st, P, n_launch);
    if (eligible && P.spec_w0 <= -2) hipLaunchKernelGGL(kc_zfast_match_lds_kernel<true>, dim3(n_launch), dim3(64), 0, st, Q, proto, proto_stride, n_launch, true);
    if (!eligible || P.lds_any_big != 0)
        hipLaunchKernelGGL(kc_zfast_match_lds_kernel<false>, dim3(n_launch), dim3(64), 0, st, Q, proto, proto_stride, n_launch, eligible);
}
