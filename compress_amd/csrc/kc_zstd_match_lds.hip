// kc_zstd_match_lds.hip — SpeedFastest match finder with the hash table in LDS: ONE WAVE PER UNIT, the latency path.
//
// Same reference functions and the same decisions as kc_zstd_match.hip (fastEncoder.Encode / EncodeNoHist /
// fastEncoderDict.Encode, zstd/enc_fast.go:39-289 / 294-531 / 534-790), same output (packed sequences + KcBlkMeta per
// block, consumed by kc_zstd_entropy_kernel).  kc_zstd_match.hip keeps 8 units per wave in flight with their tables in
// HBM: it is bound by DRAM transactions and needs ~10^4 units to cover the latency of the dependent table -> candidate
// chain (~30 ms for one 128 KiB unit).  Here the unit's table (2^15 x u32 = 128 KiB of the CU's 160 KiB of LDS) is on
// chip, so a table lookup is an LDS round trip and only candidate bytes come through L2: a few ms per unit, whatever
// the number of units in flight — and a handful of rounds per block where the scan finds nothing (high-entropy input:
// 64 probe steps per round instead of 8, and no table traffic to HBM at all).  The dispatcher in kc_batch.cpp picks the
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


void kc_launch_zfast_match_lds(const KcMatchParams& P, const uint32_t* proto, uint32_t proto_stride, uint32_t n_launch, hipStream_t st) {
    if (n_launch == 0) return;
    // spec_w0 0: units up to 128 KiB without history through the instantiation with the source ring (the other units of the launch stay
    // with the tagged form); measured no faster than the tagged form on text — the kernel is issue-bound, DESIGN.md 4.1c — so not the default
    const bool small = P.spec_w0 <= 0 && proto == nullptr && P.hist0 == 0 && P.unit_hist == nullptr && P.job_flags == nullptr;
    KcMatchParams Q = P;
    if (Q.spec_w0 <= 0) Q.spec_w0 = 16;
    if (small) hipLaunchKernelGGL(kc_zfast_match_lds_kernel<true>, dim3(n_launch), dim3(64), 0, st, Q, proto, proto_stride, n_launch, true);
    if (!small || P.lds_any_big != 0)
        hipLaunchKernelGGL(kc_zfast_match_lds_kernel<false>, dim3(n_launch), dim3(64), 0, st, Q, proto, proto_stride, n_launch, small);
}
