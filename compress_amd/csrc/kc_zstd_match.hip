// kc_zstd_match.hip — SpeedFastest match finder for gfx950: sub-wave groups, HBM tables, LDS source windows.
//
// Replaces fastEncoder.Encode / EncodeNoHist (zstd/enc_fast.go:39-289 / 294-531) and the small-input dictionary
// variant (enc_fast.go:534-790).  The reference parse is sequential; this kernel keeps its exact decisions and
// extracts parallelism by *speculative probing with ordered commit* inside G-lane groups (G = 8: 8 units per wave):
//   * while no match is found, probe positions are a pure function of (s, nextEmit)
//     (s += 2 + ((s-nextEmit)>>5), enc_fast.go:207), so the lanes of a group evaluate the next probes of the
//     current skip segment at once against the pre-round table;
//   * a lane whose bucket was touched by a lower lane of its group in the same round ends the round (exact
//     comparison of bucket indices via __shfl_up), so every committed lane saw the table state the sequential
//     encoder would have seen;
//   * group-ballot + ctz picks the first committed hit in the reference's priority order (repeat at s+2,
//     candidate at s, candidate at s+1); lanes up to the winner commit their table writes (the reference
//     writes before it checks).
// Tables (2^15 x u32 per unit: position+1 | tag of the 4 source bytes) live in HBM: 32768 units must be in
// flight to cover the latency of the dependent table -> candidate chain, and their tables (4 GiB) fit no
// on-chip memory.  tools/mem_probe.hip measures what the memory system gives this access pattern (scattered
// 4-byte read + write into per-unit 128 KiB tables over 4 GiB): 21 G read+write pairs/s, i.e. 65 k pairs per
// unit x 32768 units = 101 ms per 4 GiB before any other access.  Round 2 therefore removes the *other*
// accesses and dependent round trips of a probe round:
//   * the source bytes around the parse position come from a per-unit ring buffer in LDS (1 KiB per unit,
//     refilled 128 B at a time by the group's lanes with one aligned 16-byte load each, one round ahead of
//     use) instead of per-round 8-byte global loads that missed the thrashed L1/L2;
//   * a candidate is verified with ONE 16-byte load of [t-4, t+12): it yields the reference's 4-byte
//     acceptance test, the forward match length when it is < 12 (85 % of matches in text) and the backward
//     extension when it is < 4 (99 %); only longer matches enter the cooperative extension loops;
//   * the repeat-offset candidate (known before the table lookup) and the offset-2 check after a match
//     (enc_fast.go:250) are loaded in the same round trip as the table entries, the latter speculatively:
//     the probes of the round are discarded when the offset-2 check hits (13 times per 128 KiB of text).
// A probe round is thus: LDS window read -> {table, repeat, offset-2} loads -> candidate loads -> commit.
// Output: per block, packed sequences (no literal bytes are copied here — the entropy kernel gathers
// literals from the source using the sequence list) + a KcBlkMeta record.
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_zfast_dev.h"
#include "kc_wave.h"


#ifndef ZW_RB
#define ZW_RB 1024      // ring bytes per unit (power of two)
#endif
#define ZW_MIRROR 32    // the first 32 ring bytes are mirrored behind the ring: 24-byte reads never wrap
#define ZW_STRIDE (ZW_RB + ZW_MIRROR)
#define ZW_BK 4         // bytes in front of a probe / candidate position kept for the backward extension
#ifndef ZW_AHEAD
#define ZW_AHEAD 320    // refill while fewer than this many bytes are buffered ahead of s
#endif

// plain loads/stores: non-temporal table loads measured 174.7 vs 134.9 ms per 4 GiB (profiles/r02_match_finder_experiments.md)
#define KC_TAB_LD(p) (*(p))
#define KC_TAB_ST(v, p) (*(p) = (v))


template <int G, bool XSEG, bool FILT>
__global__ __launch_bounds__(64) void kc_zfast_match_grp_kernel(KcMatchParams P, uint32_t* __restrict__ tables, uint32_t n_launch) {
    constexpr int UPW = 64 / G;
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[UPW * ZW_STRIDE];
    __shared__ uint64_t sbuf_all[UPW * G];  // per unit: the last (nseq mod G) sequences, flushed G at a time as one 64-byte store
#ifdef KC_MATCH_PRIO
    __builtin_amdgcn_s_setprio(KC_MATCH_PRIO);  // (measurement builds: the wave's issue priority beside a co-resident entropy kernel)
#endif
    const int lane = (int)threadIdx.x;
    const int lig = lane % G, grp = lane / G;
    uint8_t* const ring = ring_all + grp * ZW_STRIDE;
    uint64_t* const sbuf = sbuf_all + grp * G;
    const uint32_t ui = blockIdx.x * UPW + (uint32_t)grp;
    const bool gact = ui < n_launch;
    const uint32_t u = gact ? (P.unit_list ? P.unit_list[ui] : P.unit_base + ui) : 0u;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int boff = (int)((uintptr_t)base & 15);       // window positions are relative to the 16-byte aligned abase
    const uint8_t* __restrict__ abase = base - boff;
    // history in front of the unit: the dictionary content, or (jobs of a WithConcurrentBlocks stream) the unit's own overlap prefix
    const int hist0 = P.unit_hist != nullptr ? (int)P.unit_hist[u] : P.hist0;
    const int ulen = gact ? (int)(P.unit_off[u + 1] - P.unit_off[u]) - hist0 : 0;
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const int mmo = P.max_match_off;
    const KcUnitBlocks UB = kc_unit_blocks(P.blk_start, P.unit_flags, P.unit_blk0, u, ulen, bs, P.stream_mode);
    const bool ldsUnit = P.lds_split != 0 && (uint32_t)(ulen + hist0) <= KC_ZFAST_LDS_MAX_UNIT;  // the LDS-table kernel's unit
    const bool doneU = gact && P.unit_done != nullptr && P.unit_done[u] != 0u;  // the pre-scan proved the unit free of matches (kc_zstd_prescan.hip)
    const int nblk = (gact && !ldsUnit && !doneU) ? UB.nblk : 0;  // a group without a unit (the launch's tail) does nothing
    const bool HIST = ulen > bs || hist0 > 0 || UB.streamU || P.job_flags != nullptr;  // compressJob always calls Encode (enc_jobs.go:114)  // with a dictionary encodeAll always calls Encode (encoder.go:783-787)
    uint32_t* __restrict__ tab = tables + (size_t)ui * (1u << ZF_TABLE_BITS);  // zeroed by the host before the launch
    // Table entry = (position+1) in the low PB bits | a TB-bit tag of the 4 source bytes at that position.
    // The reference accepts a candidate iff its 4 bytes equal the probe's (tableEntry.val == uint32(cv),
    // enc_fast.go:176,188); a tag mismatch proves they differ, so the (random, HBM-bound) candidate fetch is
    // skipped exactly when the reference would reject anyway; equal tags are still verified on the bytes.
    // Epoch stamps (round 3): with P.epoch != 0 the low KC_ZF_EPOCH_BITS of the field above the position hold this launch's
    // stamp and the tag loses those bits; an entry with another stamp — left by an earlier launch in a slot that is no longer
    // cleared per batch — fails the tag test like a mismatching tag and reads as "no candidate", exactly what a zeroed
    // bucket gives the reference (the host clears the arena when the stamp wraps).
    const int PB = P.pos_bits;  // per-launch constant (dictionary-primed tables are shared by all units)
    const int EB = P.epoch != 0u ? KC_ZF_EPOCH_BITS : 0;
    const int TB = (32 - PB - EB) > 16 ? 16 : (32 - PB - EB);
    const uint32_t posMask = (PB >= 32) ? 0xFFFFFFFFu : ((1u << PB) - 1u);
    const uint32_t stamp = P.epoch;
    auto tagOf = [&](uint32_t v) -> uint32_t { return (TB > 0 ? (((v * 2654435761u) >> (32 - TB)) << EB) : 0u) | stamp; };
    const uint8_t* const srcLo = P.src;
    const uint8_t* const srcHi = P.src_end;

    // "Nothing written there yet" filter (round 3): until the unit emits its first sequence the 64 bytes of its sequence buffer are
    // idle; they hold one bit per 64 consecutive buckets (4 table lines), set when the unit writes into them.  A lookup whose bit is
    // clear needs no load: the bucket is empty (zeroed, or stale-stamped) — on input without matches (high-entropy units: ~900
    // inserts into 2048 lines, no sequence ever) that is about half of the table reads, on a kernel bound by DRAM transactions.
    // Units whose tables arrive primed (dictionary, job prefix) do not use it; the first emit() ends it for good.
    uint32_t* const fw = (uint32_t*)sbuf;
    bool useF = FILT && gact && P.empty_filter != 0 && P.hist0 == 0 && P.unit_hist == nullptr;
    if (FILT) {
        sbuf[lig] = 0;  // (G == 8 lanes x 8 bytes: the whole buffer)
        KC_EMU_SYNC();
    }
    int o1 = P.rep1, o2 = P.rep2;  // {1,4} (blockenc.go:78) or the dictionary's offsets (enc_base.go:189-195)
    bool allDirty = false;  // fastEncoderDict.allDirty: small-input variant (kSearchStrength 7) only until a block > 32 KiB was seen
    int wlo = 0, whi = 0;   // the ring holds the bytes abase[wlo .. whi)
    bool pend = false;      // rf holds the 16*G bytes abase[whi ..) loaded during the previous round
    uint4 rf = make_uint4(0, 0, 0, 0);
    for (int b = 0; b < nblk; b++) {  // group-uniform trip count; groups diverge freely
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
            if (FILT) useF = false;  // the buffer is the sequence buffer from here on
            if (nseq == 0) { firstLL = (uint32_t)ll; firstOf = of; }
            if (lig == 0) sbuf[nseq & (G - 1)] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
            nseq++;
            sumLL += ll;
            if ((nseq & (G - 1)) == 0) {  // group-uniform: G sequences buffered -> one coalesced store (8-byte scattered stores cost a DRAM write each)
                __builtin_amdgcn_wave_barrier();
                sq[nseq - G + lig] = sbuf[lig];
                __builtin_amdgcn_wave_barrier();
            }
        };
        int SK = 5;  // kSearchStrength - 1
        if (hist0 > 0 && P.job_flags == nullptr) {  // fastEncoderDict only (enc_fast.go:539-543,585); a job's prefix goes through fastEncoder
            if (allDirty || srcLen > (32 << 10)) allDirty = true; else SK = 6;
        }
        if (srcLen >= 10) {
            const int sLimit = blkEnd - 8;
            bool canRep = false, fin = false, pendO2 = false;
            int W = G;  // speculation width: narrow right after a match (hits come early in text), growing on a miss
            while (!fin) {
                rounds++;
                // ---------------- source window (LDS ring) ----------------
                if (pend) {  // the refill issued one round ago has landed (its load precedes this round's loads in vmcnt order)
                    const int ro = (whi + 16 * lig) & (ZW_RB - 1);
                    *(uint4*)(ring + ro) = rf;
                    if (ro < ZW_MIRROR) *(uint4*)(ring + ZW_RB + ro) = rf;
                    whi += 16 * G;
                    if (whi - wlo > ZW_RB) wlo = whi - ZW_RB;
                    pend = false;
                }
                KC_EMU_SYNC();  // (the ring is written by all lanes of the group and read by all of them)
                const int sa = s + boff;
                if (sa >= whi || sa < wlo) {  // block start, or a match jumped past the window: restart it just behind s
                    int w0 = (sa - 16) & ~15;
                    if (w0 < 0) w0 = 0;
                    wlo = whi = w0;
                }
                if (whi - sa < ZW_AHEAD) {
                    const uint8_t* q = abase + whi + 16 * lig;
                    rf = make_uint4(0, 0, 0, 0);
                    if (q < srcHi) rf = *(const uint4*)q;  // aligned: never leaves the 16-byte granule of a readable byte
                    pend = true;
                }
                // ---------------- probe positions of this round ----------------
                // Lane i evaluates the i-th next probe of the scan.  The positions follow the reference's recurrence
                // s += 2 + ((s - nextEmit) >> SK) (enc_fast.go:207) step by step, so a round is not confined to one skip
                // segment (where the step is constant): on data without matches the step soon exceeds a segment's 32 bytes and
                // a segment-bound round is ONE probe (round 2: 250 rounds per 128 KiB of high-entropy input, now ~35).
                // P.xseg_k: a round crosses into the next segment only once the step has grown to 2 + xseg_k.
                const int d0 = s - nextEmit;
                const int k0 = d0 >> SK;
                const int step = 2 + k0;
                int p, pnext = 0;
                bool valid;
                if (XSEG) {
                    int pprev = s;
                    p = s;
#pragma unroll
                    for (int j = 1; j < G; j++) {
                        if (lig >= j) { pprev = p; p += 2 + ((p - nextEmit) >> SK); }
                    }
                    pnext = p + 2 + ((p - nextEmit) >> SK);  // where the scan continues when this lane is the round's last
                    valid = lig < W && (lig == 0 || k0 >= P.xseg_k || ((pprev - nextEmit) >> SK) == k0) && p < sLimit;
                } else {  // rounds confined to one skip segment (constant step): measured 2 % faster on text (no recurrence, no cross-lane read)
                    p = s + lig * step;
                    valid = lig < W && (lig == 0 || ((d0 + (lig - 1) * step) >> SK) == k0) && p < sLimit;
                }
                // R = the 20 source bytes [p-4, p+16): D1:D2 = cv, the rest feeds the fused candidate compares
                uint32_t D0 = 0, D1 = 0, D2 = 0, D3 = 0, D4 = 0;
                if (valid) {
                    const int a = p + boff - ZW_BK;
                    const int a4 = a & ~3;
                    if (a4 >= wlo && a4 + 24 <= whi) {
                        const uint32_t* r = (const uint32_t*)(ring + (a4 & (ZW_RB - 1)));
                        const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
                        const uint32_t sh = (uint32_t)(a & 3);
                        D0 = __builtin_amdgcn_alignbyte(r1, r0, sh);
                        D1 = __builtin_amdgcn_alignbyte(r2, r1, sh);
                        D2 = __builtin_amdgcn_alignbyte(r3, r2, sh);
                        D3 = __builtin_amdgcn_alignbyte(r4, r3, sh);
                        D4 = __builtin_amdgcn_alignbyte(r5, r4, sh);
                    } else {
                        const uint8_t* q = base + p - ZW_BK;
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
                // ---------------- round trip 1: table entries, repeat candidate, offset-2 candidate ----------------
                uint32_t h0 = 0xFFFFFFFFu, h1 = 0xFFFFFFFEu, c0 = 0, c1 = 0;
                if (valid) {
                    h0 = hash6(cv, ZF_TABLE_BITS);
                    h1 = hash6(cv >> 8, ZF_TABLE_BITS);
                    if (FILT && useF) {  // group-uniform
                        const uint32_t g0 = h0 >> 6, g1 = h1 >> 6;
                        if ((fw[g0 >> 5] >> (g0 & 31u)) & 1u) c0 = KC_TAB_LD(&tab[h0]);
                        if ((fw[g1 >> 5] >> (g1 & 31u)) & 1u) c1 = KC_TAB_LD(&tab[h1]);
                    } else {
                        c0 = KC_TAB_LD(&tab[h0]);
                        c1 = KC_TAB_LD(&tab[h1]);
                    }
                }
                const int repIndex = p - o1 + 2;
                const bool repOk = valid && canRep && repIndex >= 0;
                uint4 cr = make_uint4(0, 0, 0, 0);
                bool repWide = false;
                if (repOk) {
                    const uint8_t* q = base + repIndex - ZW_BK;
                    repWide = q >= srcLo && q + 16 <= srcHi;
                    if (repWide) cr = ld128u(q);
                    else cr.y = ld32(base + repIndex);
                }
                const bool doO2 = pendO2;
                const int o2pos = s - o2;
                uint4 co = make_uint4(0, 0, 0, 0);
                bool o2Wide = false;
                if (doO2 && lig == 0) {  // offset-2 check (enc_fast.go:250), speculatively in the same round trip as the probes
                    const uint8_t* q = base + o2pos;
                    o2Wide = q + 16 <= srcHi;
                    if (o2Wide) co = ld128u(q);
                    else co.x = ld32(q);
                }
                if (doO2) {
                    pendO2 = false;
                    uint32_t pk = 0;  // bit 0: hit, bit 1: length final, bits 8..: known length
                    if (lig == 0) {
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
                    pk = gbcast32<G>(pk, grp, 0);
                    if (pk & 1u) {
                        int l2 = (int)(pk >> 8);
                        if (!(pk & 2u)) l2 += grp_matchlen<G>(base, s + l2, o2pos + l2, blkEnd - (s + l2), lig, grp);
                        if (lig == 0) KC_TAB_ST(((uint32_t)s + 1u) | (PB < 32 ? tagOf((uint32_t)cv) << PB : 0u), &tab[h0]);
                        emit(0, l2 - 3, 1u);
                        W = P.spec_w0;
                        s += l2;
                        nextEmit = s;
                        const int tmp = o1; o1 = o2; o2 = tmp;
                        canRep = nseq > 2;
                        if (s >= sLimit) fin = true;
                        continue;  // the speculative probes of this round are dropped (nothing was committed)
                    }
                }
                // exact in-round conflict detection: a lower lane of the group touches one of my buckets
                // (the bucket indices of the d lanes below come over the DPP path — row_shr:d, one VALU operation each — instead of
                // ds_bpermute round trips through the LDS crossbar: a lane with lig >= d reads inside its own group, which is active as a
                // whole; what the other lanes receive is not looked at)
                bool dep = false;
#ifndef KC_MATCH_DEP_SHFL
                static_assert(G == 8, "row_shr distances below are written out for 8-lane groups");
#define KC_DEP_STEP(d) do { const uint32_t a0 = kc_dpp_or0<0x110 + (d), 0xf>(h0), a1 = kc_dpp_or0<0x110 + (d), 0xf>(h1); \
                            if (lig >= (d) && (a0 == h0 || a0 == h1 || a1 == h0 || a1 == h1)) dep = true; } while (0)
                KC_DEP_STEP(1); KC_DEP_STEP(2); KC_DEP_STEP(3); KC_DEP_STEP(4); KC_DEP_STEP(5); KC_DEP_STEP(6); KC_DEP_STEP(7);
#undef KC_DEP_STEP
#else
#pragma unroll
                for (int d = 1; d < G; d++) {
                    const uint32_t a0 = (uint32_t)__shfl_up((int)h0, d, G), a1 = (uint32_t)__shfl_up((int)h1, d, G);
                    if (lig >= d && (a0 == h0 || a0 == h1 || a1 == h0 || a1 == h1)) dep = true;
                }
#endif
                // ---------------- round trip 2: tagged candidates, one 16-byte load each ----------------
                const uint32_t e0 = c0 & posMask, e1 = c1 & posMask;
                const int t0 = (int)e0 - 1, t1 = (int)e1 - 1;
                const bool ok0 = valid && e0 != 0 && (p - t0) < mmo && (PB >= 32 || (c0 >> PB) == tagOf((uint32_t)cv));
                const bool ok1 = valid && e1 != 0 && (p - t1 + 1) < mmo && (PB >= 32 || (c1 >> PB) == tagOf((uint32_t)(cv >> 8)));
                uint4 ca = make_uint4(0, 0, 0, 0), cb = make_uint4(0, 0, 0, 0);
                bool wide0 = false, wide1 = false;
                if (ok0) {  // predicated: a masked-off lane costs no address slot in the texture path
                    const uint8_t* q = base + t0 - ZW_BK;
                    wide0 = q >= srcLo && q + 16 <= srcHi;
                    if (wide0) ca = ld128u(q);
                    else ca.y = ld32(base + t0);
                }
                if (ok1) {
                    const uint8_t* q = base + t1 - ZW_BK;
                    wide1 = q >= srcLo && q + 16 <= srcHi;
                    if (wide1) cb = ld128u(q);
                    else cb.y = ld32(base + t1);
                }
                // per-lane verdict: kind 1 repeat (s+2), 2 candidate at s, 3 candidate2 at s+1 (enc_fast.go:133,176,188)
                int kind = 0, t = 0, fwd = 0, back = 0, ba = 0, fa = 0, kofs = 0;
                if (repOk) {
                    int f, bk;
                    zf_cmp16(cr, __builtin_amdgcn_alignbyte(D1, D0, 2), __builtin_amdgcn_alignbyte(D2, D1, 2),
                             __builtin_amdgcn_alignbyte(D3, D2, 2), __builtin_amdgcn_alignbyte(D4, D3, 2), f, bk);
                    if (f >= 4) { kind = 1; fwd = repWide ? f : 4; back = repWide ? bk : 0; fa = repWide ? 12 : 4; ba = repWide ? ZW_BK : 0; kofs = 2; }
                }
                if (kind == 0 && ok0) {
                    int f, bk;
                    zf_cmp16(ca, D0, D1, D2, D3, f, bk);
                    if (f >= 4) { kind = 2; t = t0; fwd = wide0 ? f : 4; back = wide0 ? bk : 0; fa = wide0 ? 12 : 4; ba = wide0 ? ZW_BK : 0; kofs = 0; }
                }
                if (kind == 0 && ok1) {
                    int f, bk;
                    zf_cmp16(cb, __builtin_amdgcn_alignbyte(D1, D0, 1), __builtin_amdgcn_alignbyte(D2, D1, 1),
                             __builtin_amdgcn_alignbyte(D3, D2, 1), __builtin_amdgcn_alignbyte(D4, D3, 1), f, bk);
                    if (f >= 4) { kind = 3; t = t1; fwd = wide1 ? f : 4; back = wide1 ? bk : 0; fa = wide1 ? 12 : 4; ba = wide1 ? ZW_BK : 0; kofs = 1; }
                }
                uint32_t vk = 0;  // kind:2 | length final:1 | known forward length:5 | equal bytes behind:3 | bytes behind examined:3
                if (kind != 0) {
                    const int limit = blkEnd - (p + kofs);
                    const bool done = fwd < fa || fwd >= limit;
                    const int fk = fwd < limit ? fwd : limit;
                    vk = (uint32_t)kind | (done ? 4u : 0u) | ((uint32_t)fk << 3) | ((uint32_t)back << 8) | ((uint32_t)ba << 11);
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
                    KC_TAB_ST(((uint32_t)p + 1u) | (PB < 32 ? tagOf((uint32_t)cv) << PB : 0u), &tab[h0]);
                    KC_TAB_ST(((uint32_t)p + 2u) | (PB < 32 ? tagOf((uint32_t)(cv >> 8)) << PB : 0u), &tab[h1]);  // program order: wins when h0 == h1
                    if (FILT && useF) {
                        atomicOr(&fw[h0 >> 11], 1u << ((h0 >> 6) & 31u));
                        atomicOr(&fw[h1 >> 11], 1u << ((h1 >> 6) & 31u));
                    }
                }
                if (!found) {
                    W = P.spec_grow == 0 ? W : (P.spec_grow == 1 ? (W + 1 < G ? W + 1 : G) : ((2 * W < G) ? 2 * W : G));
                    if (XSEG) {
                        const int m = c < nvalid ? c : nvalid;  // the scan continues behind lane m - 1 (m >= 1: lane 0 depends on nobody; nvalid >= 1)
                        if (((d0 + (m - 1) * step) >> SK) == k0) s = s + m * step;  // lanes 0 .. m-1 in lane 0's skip segment: no cross-lane read
                        else s = (int)gbcast32<G>((uint32_t)pnext, grp, m - 1);
                    } else if (c < nvalid) {
                        s = s + c * step;
                    } else {
                        const int pl = s + (nvalid - 1) * step;
                        s = pl + 2 + ((pl - nextEmit) >> SK);
                    }
                    if (s >= sLimit) fin = true;
                    continue;
                }
                const uint32_t wk = gbcast32<G>(vk, grp, f);
                const int mk = (int)(wk & 3u);
                const bool fdone = (wk & 4u) != 0;
                const int fk = (int)((wk >> 3) & 31u);
                const int bke = (int)((wk >> 8) & 7u), bav = (int)((wk >> 11) & 7u);
                const int ps = (!XSEG || f == 0 || ((d0 + (f - 1) * step) >> SK) == k0) ? s + f * step : (int)gbcast32<G>((uint32_t)p, grp, f);
                int mt = (int)gbcast32<G>((uint32_t)t, grp, f);
                // backward extension given the bke equal bytes found among the bav bytes examined (enc_fast.go:152-157, 230-234)
                auto backlen = [&](int sp, int tp, int kmax) -> int {
                    if (kmax <= 0) return 0;
                    if (bke < bav) return bke < kmax ? bke : kmax;
                    if (kmax <= bav) return kmax;
                    return bav + grp_backlen<G>(base, sp - bav, tp - bav, kmax - bav, lig, grp);
                };
                if (mk == 1) {
                    const int rI = ps - o1 + 2;
                    int length = fk;
                    if (!fdone) length += grp_matchlen<G>(base, ps + 2 + fk, rI + fk, blkEnd - (ps + 2 + fk), lig, grp);
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
                    W = P.spec_w0;
                    s = ps + length + 2;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                s = ps + (mk == 3 ? 1 : 0);
                o2 = o1;
                o1 = s - mt;
                int l = fk;
                if (!fdone) l += grp_matchlen<G>(base, s + fk, mt + fk, blkEnd - (s + fk), lig, grp);
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
                W = P.spec_w0;
                s += l;
                nextEmit = s;
                const bool canRepO2 = HIST ? canRep : (nseq > 2);
                canRep = nseq > 2;
                if (s >= sLimit) { fin = true; continue; }
                pendO2 = canRepO2;
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
}

void kc_launch_zfast_match_grp(const KcMatchParams& P, uint32_t* tables, uint32_t n_launch, hipStream_t st) {
    // Two compiled forms, the same bytes: the plain one (rounds inside one skip segment, no filter) is ~1 % faster on compressible
    // input, where neither addition ever acts; the other one is for input that yields no matches (P.tuned: the host's choice).
    if (P.tuned) hipLaunchKernelGGL((kc_zfast_match_grp_kernel<8, true, true>), dim3((n_launch + 7) / 8), dim3(64), 0, st, P, tables, n_launch);
    else hipLaunchKernelGGL((kc_zfast_match_grp_kernel<8, false, false>), dim3((n_launch + 7) / 8), dim3(64), 0, st, P, tables, n_launch);
}
