// kc_zstd_match.hip — SpeedFastest match finder for gfx950 (one wave64 per unit).
//
// Replaces fastEncoder.Encode / EncodeNoHist (zstd/enc_fast.go:39-289 / 294-531).
// The reference parse is sequential; this kernel keeps its exact decisions and extracts
// wave parallelism by *speculative probing with ordered commit*:
//   * while no match is found, probe positions are a pure function of (s, nextEmit)
//     (s += 2 + ((s-nextEmit)>>5), enc_fast.go:207), so the 64 lanes evaluate the next
//     probes of the current skip segment at once against the pre-round table;
//   * a lane whose two table buckets were also touched by a lower lane in the same round
//     (detected exactly-or-conservatively through a small LDS mark array) ends the round:
//     only lanes below it are committed, so every committed lane saw exactly the table
//     state the sequential encoder would have seen;
//   * ballot + ctz picks the first committed lane with a hit in the reference's priority
//     order (repeat at s+2, candidate at s, candidate at s+1); lanes up to and including
//     the winner write their table entries (the reference writes before it checks);
//   * forward / backward match extension are wave-wide 8 B-per-lane compares + ballot.
// Hash table: 2^15 entries in LDS holding position+1 (0 = empty).  The reference's
// tableEntry.val is redundant with the source bytes (enc_fast.go:130-131), so comparing
// 4 source bytes at the candidate is exact; its `cur` epoch offset only invalidates stale
// entries, which a zeroed per-unit table reproduces.
// Output: per block, packed sequences (no literal bytes are copied here — the entropy
// kernel gathers literals from the source using the sequence list) + a KcBlkMeta record.
#include "kc_dev.h"
#include "kc_kernels.h"

#define ZF_TABLE_BITS 15
#define ZF_MARK_SLOTS 1024
#define ZF_MAX_MATCH_LENGTH 131074  // enc_fast.go:18

__global__ __launch_bounds__(64) void kc_zfast_match_kernel(KcMatchParams P) {
    __shared__ uint32_t tab[1 << ZF_TABLE_BITS];
    __shared__ uint32_t mark[ZF_MARK_SLOTS];
    const int lane = (int)threadIdx.x;
    const uint32_t u = P.unit_list ? P.unit_list[blockIdx.x] : P.unit_base + blockIdx.x;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int ulen = (int)(P.unit_off[u + 1] - P.unit_off[u]);
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const int mmo = P.max_match_off;
    const int nblk = (ulen + bs - 1) / bs;
    const bool HIST = ulen > bs;  // encodeAll: Encode for multi-block units, EncodeNoHist otherwise (encoder.go:775-823)
    const uint32_t pm = P.popmask ? P.popmask[u] : 0u;

    for (int i = lane; i < (1 << ZF_TABLE_BITS); i += 64) tab[i] = 0;
    for (int i = lane; i < ZF_MARK_SLOTS; i += 64) mark[i] = 0xFFFFFFFFu;
    __syncthreads();

    int o1 = 1, o2 = 4;  // blockEnc.initNewEncode: recentOffsets = {1,4,8} (blockenc.go:78)
    for (int b = 0; b < nblk; b++) {
        const int blkStart = b * bs;
        const int blkEnd = (blkStart + bs < ulen) ? blkStart + bs : ulen;  // == len(e.hist) after addBlock
        const int srcLen = blkEnd - blkStart;
        const int o1_in = o1, o2_in = o2;
        uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        int nseq = 0, sumLL = 0;
        uint32_t rounds = 0;  // probe rounds (diagnostics: KcBlkMeta.flags bits 8..31)
        int nextEmit = blkStart, s = blkStart;
        uint32_t firstLL = 0, firstOf = 0;

        auto emit = [&](int ll, int ml3, uint32_t of) {
            if (nseq == 0) { firstLL = (uint32_t)ll; firstOf = of; }
            if (lane == 0) sq[nseq] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
            nseq++;
            sumLL += ll;
        };

        if (srcLen >= 10) {  // minNonLiteralBlockSize = 1 + 1 + inputMargin
            const int sLimit = blkEnd - 8;
            bool canRep = false;  // len(blk.sequences) > 2 snapshot at outer-loop start (enc_fast.go:117)
            bool fin = false;
            while (!fin) {
                // ---------------- speculative probe round ----------------
                rounds++;
                const int d0 = s - nextEmit;
                const int k0 = d0 >> 5;  // kSearchStrength-1 == 5
                const int step = 2 + k0;
                const int p = s + lane * step;
                const bool valid = (lane == 0 || ((d0 + (lane - 1) * step) >> 5) == k0) && p < sLimit;
                uint64_t cv = 0;
                uint32_t h0 = 0, h1 = 0, c0 = 0, c1 = 0;
                if (valid) {
                    cv = ld64(base + p);
                    h0 = hash6(cv, ZF_TABLE_BITS);
                    h1 = hash6(cv >> 8, ZF_TABLE_BITS);
                    c0 = tab[h0];
                    c1 = tab[h1];
                    atomicMin(&mark[h0 & (ZF_MARK_SLOTS - 1)], (uint32_t)lane);
                    atomicMin(&mark[h1 & (ZF_MARK_SLOTS - 1)], (uint32_t)lane);
                }
                int kind = 0;  // 1 repeat (s+2), 2 candidate at s, 3 candidate2 at s+1
                int t = 0;
                if (valid) {
                    const int repIndex = p - o1 + 2;
                    if (canRep && repIndex >= 0 && ld32(base + repIndex) == (uint32_t)(cv >> 16)) {
                        kind = 1;
                    } else {
                        const int t0 = (int)c0 - 1, t1 = (int)c1 - 1;
                        if (c0 != 0 && (p - t0) < mmo && ld32(base + t0) == (uint32_t)cv) {
                            kind = 2;
                            t = t0;
                        } else if (c1 != 0 && (p - t1 + 1) < mmo && ld32(base + t1) == (uint32_t)(cv >> 8)) {
                            kind = 3;
                            t = t1;
                        }
                    }
                }
                bool dep = false;
                if (valid) {
                    const uint32_t m0 = mark[h0 & (ZF_MARK_SLOTS - 1)], m1 = mark[h1 & (ZF_MARK_SLOTS - 1)];
                    dep = m0 < (uint32_t)lane || m1 < (uint32_t)lane;
                    mark[h0 & (ZF_MARK_SLOTS - 1)] = 0xFFFFFFFFu;
                    mark[h1 & (ZF_MARK_SLOTS - 1)] = 0xFFFFFFFFu;
                }
                const uint64_t vm = ballot64(valid);
                const uint64_t depm = ballot64(dep);
                const uint64_t hm = ballot64(kind != 0);
                const int nvalid = __popcll(vm);  // valid lanes form a prefix
                const int c = depm ? ctz64(depm) : 64;
                const uint64_t lowmask = c >= 64 ? ~0ull : ((1ull << c) - 1ull);
                const uint64_t hmc = hm & lowmask;
                const bool found = hmc != 0;
                const int f = found ? ctz64(hmc) : 0;
                const int commitUpTo = found ? f : ((c < nvalid ? c : nvalid) - 1);
                if (valid && lane <= commitUpTo) {
                    tab[h0] = (uint32_t)p + 1u;  // table[nextHash]  = {s}
                    tab[h1] = (uint32_t)p + 2u;  // table[nextHash2] = {s+1}; later store wins when h0 == h1
                }
                if (!found) {
                    if (c < nvalid) {
                        s = s + c * step;  // first dependent lane restarts the next round as lane 0
                    } else {
                        const int pl = s + (nvalid - 1) * step;
                        s = pl + 2 + ((pl - nextEmit) >> 5);
                    }
                    if (s >= sLimit) fin = true;
                    continue;
                }
                const int mk = (int)bcast32((uint32_t)kind, f);
                const int ps = s + f * step;
                int mt = (int)bcast32((uint32_t)t, f);
                if (mk == 1) {
                    // ---------------- repeat match at s+2 (enc_fast.go:133-173) ----------------
                    int repIndex = ps - o1 + 2;
                    const int length = 4 + wave_matchlen(base + ps + 6, base + repIndex + 4, blkEnd - (ps + 6), lane);
                    int start = ps + 2;
                    const int startLimit = nextEmit + 1;
                    const int sMin = (ps - mmo) > 0 ? (ps - mmo) : 0;
                    int kmax = repIndex - sMin;
                    if (start - startLimit < kmax) kmax = start - startLimit;
                    if (HIST) {  // && seq.matchLen < maxMatchLength-zstdMinMatch (:147); EncodeNoHist has no cap (:385)
                        const int cap = (ZF_MAX_MATCH_LENGTH - 3) - (length - 3);
                        if (cap < kmax) kmax = cap;
                    }
                    if (kmax < 0) kmax = 0;
                    const int back = wave_backlen(base, start, repIndex, kmax, lane);
                    start -= back;
                    emit(start - nextEmit, length - 3 + back, 1u);
                    s = ps + length + 2;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;  // stays in the inner loop: canRepeat is not re-evaluated
                }
                // ---------------- regular match (enc_fast.go:211-247) ----------------
                s = ps + (mk == 3 ? 1 : 0);
                o2 = o1;
                o1 = s - mt;
                int l = wave_matchlen(base + s + 4, base + mt + 4, blkEnd - (s + 4), lane) + 4;
                {
                    const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = mt - tMin;
                    if (s - nextEmit < kmax) kmax = s - nextEmit;
                    if (HIST && (ZF_MAX_MATCH_LENGTH - l) < kmax) kmax = ZF_MAX_MATCH_LENGTH - l;  // && l < maxMatchLength (:230)
                    if (kmax < 0) kmax = 0;
                    const int back = wave_backlen(base, s, mt, kmax, lane);
                    s -= back;
                    mt -= back;
                    l += back;
                }
                emit(s - nextEmit, l - 3, (uint32_t)(s - mt) + 3u);
                s += l;
                nextEmit = s;
                // Encode uses the stale snapshot, EncodeNoHist re-evaluates (App. A-4; :251 vs :491)
                const bool canRepO2 = HIST ? canRep : (nseq > 2);
                canRep = nseq > 2;  // next outer iteration
                if (s >= sLimit) { fin = true; continue; }
                if (canRepO2) {
                    const uint64_t cv2 = ld64(base + s);
                    const int o2pos = s - o2;
                    if (ld32(base + o2pos) == (uint32_t)cv2) {
                        const int l2 = 4 + wave_matchlen(base + s + 4, base + o2pos + 4, blkEnd - (s + 4), lane);
                        if (lane == 0) tab[hash6(cv2, ZF_TABLE_BITS)] = (uint32_t)s + 1u;
                        emit(0, l2 - 3, 1u);
                        s += l2;
                        nextEmit = s;
                        const int tmp = o1; o1 = o2; o2 = tmp;
                        canRep = nseq > 2;
                        if (s >= sLimit) fin = true;
                    }
                }
            }
        }
        const int extra = nextEmit < blkEnd ? blkEnd - nextEmit : 0;
        const int nlit = sumLL + extra;
        // Verdicts of blockEnc.encode that the match finder can evaluate itself (blockenc.go:482-503):
        const bool rle = nseq == 1 && nlit <= 1 && (int)firstLL == nlit && firstOf - 3u == 1u;
        const int saved = srcLen - nlit - (srcLen >> 6);
        uint32_t flags = 0;
        if (nseq > 0 && !rle && saved < 16) flags |= KC_BF_POP_A;
        if ((pm >> b) & 1u) flags |= KC_BF_FORCED;
        const int o1c = o1, o2c = o2;
        if (flags) { o1 = o1_in; o2 = o2_in; }  // popOffsets
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


// =======================================================================================
// v2: LDS-resident current block + packed 17-bit table (units <= 128 KiB - 8, blocks <= 64 KiB)
// =======================================================================================
// LDS budget per workgroup (one wave64): 64 KiB table low halves (u16) + 4 KiB table high-bit
// plane + 64 KiB source block (+ pad) + 4 KiB conflict marks = 136 KiB of the CU's 160 KiB.
// Every read on the s side of the parse (probe bytes, forward/backward extension, offset-2
// check) and every candidate inside the current block is an LDS access (~64 clk) instead of a
// dependent global load (~500+ clk from L2/HBM); only candidates that point into the previous
// block of a multi-block unit go to global memory (the whole unit is always there).
#define ZL_SRC_WORDS ((65536 + 64) / 4)

struct ZlSrc {
    const uint32_t* w;      // LDS copy of [blkStart, blkEnd)
    const uint8_t* base;    // global: whole unit
    int blkStart;
    // 8 / 4 / 1 bytes at absolute position `pos` of the unit
    __device__ __forceinline__ uint64_t ld64l(int pos) const {  // pos inside the current block
        const int o = pos - blkStart;
        const int i = o >> 2, sh = o & 3;
        const uint32_t a = w[i], b = w[i + 1], c = w[i + 2];
        const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, sh);
        const uint32_t hi = __builtin_amdgcn_alignbyte(c, b, sh);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ uint32_t ld32l(int pos) const {
        const int o = pos - blkStart;
        const int i = o >> 2, sh = o & 3;
        return __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
    }
    __device__ __forceinline__ uint8_t ld8l(int pos) const { return ((const uint8_t*)w)[pos - blkStart]; }
    // anywhere in the unit: LDS when the whole access starts inside the current block
    __device__ __forceinline__ uint64_t ld64(int pos) const { return pos >= blkStart ? ld64l(pos) : ::ld64(base + pos); }
    __device__ __forceinline__ uint32_t ld32(int pos) const { return pos >= blkStart ? ld32l(pos) : ::ld32(base + pos); }
    __device__ __forceinline__ uint8_t ld8(int pos) const { return pos >= blkStart ? ld8l(pos) : base[pos]; }
};

// common prefix of [a, a+left) (current block) and b (earlier, anywhere)
__device__ __forceinline__ int zl_matchlen(const ZlSrc& S, int a, int b, int left, int lane) {
    int n = 0;
    int width = 8;
    for (;;) {
        const int words = (left - n) >> 3;
        const int active = words < width ? words : width;
        uint64_t diff = 0;
        if (lane < active) diff = S.ld64l(a + n + 8 * lane) ^ S.ld64(b + n + 8 * lane);
        const uint64_t m = ballot64(diff != 0);
        if (m) {
            const int fl = ctz64(m);
            const uint64_t d = bcast64(diff, fl);
            return n + 8 * fl + (ctz64(d) >> 3);
        }
        n += 8 * active;
        if (active < width) break;
        width = 64;
    }
    const int tail = left - n;
    const bool ne = lane < tail && S.ld8l(a + n + lane) != S.ld8(b + n + lane);
    const uint64_t m = ballot64(ne);
    return n + (m ? ctz64(m) : tail);
}
__device__ __forceinline__ int zl_backlen(const ZlSrc& S, int s, int t, int kmax, int lane) {
    int cnt = 0;
    while (cnt < kmax) {
        const int k = cnt + lane + 1;
        bool ne = true;
        if (k <= kmax) ne = S.ld8(t - k) != S.ld8l(s - k);
        const uint64_t m = ballot64(ne);
        const int c = m ? ctz64(m) : 64;
        cnt += c;
        if (c < 64) break;
    }
    return cnt < kmax ? cnt : kmax;
}

__global__ __launch_bounds__(64) void kc_zfast_match_lds_kernel(KcMatchParams P) {
    __shared__ uint16_t tabLo[1 << ZF_TABLE_BITS];
    __shared__ uint32_t tabHi[(1 << ZF_TABLE_BITS) / 32];
    __shared__ uint32_t srcw[ZL_SRC_WORDS];
    __shared__ uint32_t mark[ZF_MARK_SLOTS];
    const int lane = (int)threadIdx.x;
    const uint32_t u = P.unit_list ? P.unit_list[blockIdx.x] : P.unit_base + blockIdx.x;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int ulen = (int)(P.unit_off[u + 1] - P.unit_off[u]);
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const int mmo = P.max_match_off;
    const int nblk = (ulen + bs - 1) / bs;
    const bool HIST = ulen > bs;
    const uint32_t pm = P.popmask ? P.popmask[u] : 0u;

    for (int i = lane; i < (1 << ZF_TABLE_BITS) / 2; i += 64) ((uint32_t*)tabLo)[i] = 0;
    for (int i = lane; i < (1 << ZF_TABLE_BITS) / 32; i += 64) tabHi[i] = 0;
    for (int i = lane; i < ZF_MARK_SLOTS; i += 64) mark[i] = 0xFFFFFFFFu;
    __syncthreads();

    auto tab_get = [&](uint32_t h) -> uint32_t { return (uint32_t)tabLo[h] | (((tabHi[h >> 5] >> (h & 31)) & 1u) << 16); };
    auto tab_put = [&](uint32_t h, uint32_t v, uint32_t old) {
        tabLo[h] = (uint16_t)v;
        if ((v ^ old) & 0x10000u) {
            if (v & 0x10000u) atomicOr(&tabHi[h >> 5], 1u << (h & 31));
            else atomicAnd(&tabHi[h >> 5], ~(1u << (h & 31)));
        }
    };

    int o1 = 1, o2 = 4;
    for (int b = 0; b < nblk; b++) {
        const int blkStart = b * bs;
        const int blkEnd = (blkStart + bs < ulen) ? blkStart + bs : ulen;
        const int srcLen = blkEnd - blkStart;
        const int o1_in = o1, o2_in = o2;
        uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        int nseq = 0, sumLL = 0;
        uint32_t rounds = 0;  // probe rounds (diagnostics: KcBlkMeta.flags bits 8..31)
        int nextEmit = blkStart, s = blkStart;
        uint32_t firstLL = 0, firstOf = 0;

        // stage the block into LDS (coalesced dword loads; the global base may be unaligned)
        __syncthreads();
        {
            const int nw = (srcLen + 3) >> 2;
            const uint8_t* g = base + blkStart;
            for (int i = lane; i < nw; i += 64) {
                const int o = 4 * i;
                uint32_t v;
                if (o + 4 <= srcLen) v = ::ld32(g + o);
                else { v = 0; for (int q = 0; o + q < srcLen; q++) v |= (uint32_t)g[o + q] << (8 * q); }
                srcw[i] = v;
            }
            for (int i = nw + lane; i < nw + 8 && i < ZL_SRC_WORDS; i += 64) srcw[i] = 0;
        }
        __syncthreads();
        ZlSrc S{srcw, base, blkStart};

        auto emit = [&](int ll, int ml3, uint32_t of) {
            if (nseq == 0) { firstLL = (uint32_t)ll; firstOf = of; }
            if (lane == 0) sq[nseq] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
            nseq++;
            sumLL += ll;
        };

        if (srcLen >= 10) {
            const int sLimit = blkEnd - 8;
            bool canRep = false;
            bool fin = false;
            bool pendO2 = false;  // offset-2 check (enc_fast.go:250) folded into the next probe round's loads
            while (!fin) {
                rounds++;
                const int d0 = s - nextEmit;
                const int k0 = d0 >> 5;
                const int step = 2 + k0;
                const int p = s + lane * step;
                const bool valid = (lane == 0 || ((d0 + (lane - 1) * step) >> 5) == k0) && p < sLimit;
                // ---- phase 1: source bytes at the probe positions (+ the offset-2 bytes) ----
                const uint64_t cv = valid ? S.ld64l(p) : 0ull;
                if (pendO2) {
                    pendO2 = false;
                    const int o2pos = s - o2;
                    const uint32_t w2 = S.ld32(o2pos);
                    const uint64_t cv0 = bcast64(cv, 0);
                    if (w2 == (uint32_t)cv0) {
                        const int l2 = 4 + zl_matchlen(S, s + 4, o2pos + 4, blkEnd - (s + 4), lane);
                        if (lane == 0) { const uint32_t h = hash6(cv0, ZF_TABLE_BITS); tab_put(h, (uint32_t)s + 1u, tab_get(h)); }
                        emit(0, l2 - 3, 1u);
                        s += l2;
                        nextEmit = s;
                        const int tmp = o1; o1 = o2; o2 = tmp;
                        canRep = nseq > 2;
                        if (s >= sLimit) fin = true;
                        continue;
                    }
                }
                // ---- phase 2: table lookups + conflict marks ----
                uint32_t h0 = 0, h1 = 0, c0 = 0, c1 = 0;
                if (valid) {
                    h0 = hash6(cv, ZF_TABLE_BITS);
                    h1 = hash6(cv >> 8, ZF_TABLE_BITS);
                    c0 = tab_get(h0);
                    c1 = tab_get(h1);
                    atomicMin(&mark[h0 & (ZF_MARK_SLOTS - 1)], (uint32_t)lane);
                    atomicMin(&mark[h1 & (ZF_MARK_SLOTS - 1)], (uint32_t)lane);
                }
                // ---- phase 3: candidate bytes (all three loads issued unconditionally) ----
                int kind = 0, t = 0;
                bool dep = false;
                if (valid) {
                    const int repIndex = p - o1 + 2;
                    const bool repOk = canRep && repIndex >= 0;
                    const int t0 = (int)c0 - 1, t1 = (int)c1 - 1;
                    const bool ok0 = c0 != 0 && (p - t0) < mmo;
                    const bool ok1 = c1 != 0 && (p - t1 + 1) < mmo;
                    const uint32_t wr = S.ld32(repOk ? repIndex : p);
                    const uint32_t w0 = S.ld32(ok0 ? t0 : p);
                    const uint32_t w1 = S.ld32(ok1 ? t1 : p);
                    const uint32_t m0 = mark[h0 & (ZF_MARK_SLOTS - 1)], m1 = mark[h1 & (ZF_MARK_SLOTS - 1)];
                    dep = m0 < (uint32_t)lane || m1 < (uint32_t)lane;
                    mark[h0 & (ZF_MARK_SLOTS - 1)] = 0xFFFFFFFFu;
                    mark[h1 & (ZF_MARK_SLOTS - 1)] = 0xFFFFFFFFu;
                    if (repOk && wr == (uint32_t)(cv >> 16)) kind = 1;
                    else if (ok0 && w0 == (uint32_t)cv) { kind = 2; t = t0; }
                    else if (ok1 && w1 == (uint32_t)(cv >> 8)) { kind = 3; t = t1; }
                }
                const uint64_t vm = ballot64(valid);
                const uint64_t depm = ballot64(dep);
                const uint64_t hm = ballot64(kind != 0);
                const int nvalid = __popcll(vm);
                const int c = depm ? ctz64(depm) : 64;
                const uint64_t lowmask = c >= 64 ? ~0ull : ((1ull << c) - 1ull);
                const uint64_t hmc = hm & lowmask;
                const bool found = hmc != 0;
                const int f = found ? ctz64(hmc) : 0;
                const int commitUpTo = found ? f : ((c < nvalid ? c : nvalid) - 1);
                if (valid && lane <= commitUpTo) {
                    tab_put(h0, (uint32_t)p + 1u, c0);
                    // when h0 == h1 the second store must see the first one's high bit as "old"
                    tab_put(h1, (uint32_t)p + 2u, h1 == h0 ? (uint32_t)p + 1u : c1);
                }
                if (!found) {
                    if (c < nvalid) {
                        s = s + c * step;
                    } else {
                        const int pl = s + (nvalid - 1) * step;
                        s = pl + 2 + ((pl - nextEmit) >> 5);
                    }
                    if (s >= sLimit) fin = true;
                    continue;
                }
                const int mk = (int)bcast32((uint32_t)kind, f);
                const int ps = s + f * step;
                int mt = (int)bcast32((uint32_t)t, f);
                if (mk == 1) {
                    int repIndex = ps - o1 + 2;
                    const int length = 4 + zl_matchlen(S, ps + 6, repIndex + 4, blkEnd - (ps + 6), lane);
                    int start = ps + 2;
                    const int startLimit = nextEmit + 1;
                    const int sMin = (ps - mmo) > 0 ? (ps - mmo) : 0;
                    int kmax = repIndex - sMin;
                    if (start - startLimit < kmax) kmax = start - startLimit;
                    if (HIST) {
                        const int cap = (ZF_MAX_MATCH_LENGTH - 3) - (length - 3);
                        if (cap < kmax) kmax = cap;
                    }
                    if (kmax < 0) kmax = 0;
                    const int back = zl_backlen(S, start, repIndex, kmax, lane);
                    start -= back;
                    emit(start - nextEmit, length - 3 + back, 1u);
                    s = ps + length + 2;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                s = ps + (mk == 3 ? 1 : 0);
                o2 = o1;
                o1 = s - mt;
                int l = zl_matchlen(S, s + 4, mt + 4, blkEnd - (s + 4), lane) + 4;
                {
                    const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = mt - tMin;
                    if (s - nextEmit < kmax) kmax = s - nextEmit;
                    if (HIST && (ZF_MAX_MATCH_LENGTH - l) < kmax) kmax = ZF_MAX_MATCH_LENGTH - l;
                    if (kmax < 0) kmax = 0;
                    const int back = zl_backlen(S, s, mt, kmax, lane);
                    s -= back;
                    mt -= back;
                    l += back;
                }
                emit(s - nextEmit, l - 3, (uint32_t)(s - mt) + 3u);
                s += l;
                nextEmit = s;
                const bool canRepO2 = HIST ? canRep : (nseq > 2);
                canRep = nseq > 2;
                if (s >= sLimit) { fin = true; continue; }
                pendO2 = canRepO2;
            }
        }
        const int extra = nextEmit < blkEnd ? blkEnd - nextEmit : 0;
        const int nlit = sumLL + extra;
        const bool rle = nseq == 1 && nlit <= 1 && (int)firstLL == nlit && firstOf - 3u == 1u;
        const int saved = srcLen - nlit - (srcLen >> 6);
        uint32_t flags = 0;
        if (nseq > 0 && !rle && saved < 16) flags |= KC_BF_POP_A;
        if ((pm >> b) & 1u) flags |= KC_BF_FORCED;
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

// =======================================================================================
// v3: sub-wave groups — G lanes per unit, 64/G units per wave, hash tables in HBM
// =======================================================================================
// The wave-per-unit kernels above keep the table in LDS, which caps the chip at 256-512
// resident units and leaves each unit on ONE in-order wave: the parse then runs at single-wave
// issue latency (measured: ~3.5k cycles per sequence, LDS- or HBM-resident source alike).
// This variant trades on-chip tables for residency and SIMD efficiency: every group of G
// lanes runs the same speculative-probe / ordered-commit scheme for its own unit, so one
// instruction stream advances 64/G units, 4-8 waves per SIMD hide the memory latency, and all
// units of a 4 GiB batch are in flight at once.  Tables (2^15 x u32 position+1 per unit) live
// in a scratch arena in HBM; conflicts inside a probe round are detected exactly by comparing
// bucket indices across the group's lanes (no LDS at all).
#ifndef ZG_W0
#define ZG_W0 4  // initial speculation width after a match
#endif
#ifdef KC_TAB_NT
#define KC_TAB_LD(p) __builtin_nontemporal_load(p)
#define KC_TAB_ST(v, p) __builtin_nontemporal_store((uint32_t)(v), p)
#else
#define KC_TAB_LD(p) (*(p))
#define KC_TAB_ST(v, p) (*(p) = (v))
#endif
template <int G>
__global__ __launch_bounds__(64) void kc_zfast_match_grp_kernel(KcMatchParams P, uint32_t* __restrict__ tables, uint32_t n_launch) {
    constexpr int UPW = 64 / G;
    const int lane = (int)threadIdx.x;
    const int lig = lane % G, grp = lane / G;
    const uint32_t ui = blockIdx.x * UPW + (uint32_t)grp;
    const bool gact = ui < n_launch;
    const uint32_t u = gact ? (P.unit_list ? P.unit_list[ui] : P.unit_base + ui) : 0u;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int hist0 = P.hist0;  // dictionary content in front of the unit (history): fastEncoderDict (enc_fast.go:534-790)
    const int ulen = gact ? (int)(P.unit_off[u + 1] - P.unit_off[u]) - hist0 : 0;
    const uint32_t blk0 = P.unit_blk0[u];
    const int bs = P.block_size;
    const int mmo = P.max_match_off;
    const int nblk = (ulen + bs - 1) / bs;
    const bool HIST = ulen > bs || hist0 > 0 || (P.stream_mode && ulen >= bs);  // with a dictionary encodeAll always calls Encode (encoder.go:783-787)
    const uint32_t pm = (gact && P.popmask) ? P.popmask[u] : 0u;
    uint32_t* __restrict__ tab = tables + (size_t)ui * (1u << ZF_TABLE_BITS);  // zeroed by the host before the launch
    // Table entry = (position+1) in the low PB bits | a TB-bit tag of the 4 source bytes at that position.
    // The reference accepts a candidate iff its 4 bytes equal the probe's (tableEntry.val == uint32(cv),
    // enc_fast.go:176,188); a tag mismatch proves they differ, so the (random, HBM-bound) candidate fetch is
    // skipped exactly when the reference would reject anyway; equal tags are still verified on the bytes.
    const int PB = P.pos_bits;  // per-launch constant (dictionary-primed tables are shared by all units)
    const int TB = (32 - PB) > 16 ? 16 : (32 - PB);
    const uint32_t posMask = (PB >= 32) ? 0xFFFFFFFFu : ((1u << PB) - 1u);
    auto tagOf = [&](uint32_t v) -> uint32_t { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; };

    int o1 = P.rep1, o2 = P.rep2;  // {1,4} (blockenc.go:78) or the dictionary's offsets (enc_base.go:189-195)
    bool allDirty = false;  // fastEncoderDict.allDirty: small-input variant (kSearchStrength 7) only until a block > 32 KiB was seen
    for (int b = 0; b < nblk; b++) {  // group-uniform trip count; groups diverge freely
        const int blkStart = hist0 + b * bs;
        const int blkEnd = (blkStart + bs < hist0 + ulen) ? blkStart + bs : hist0 + ulen;
        const int srcLen = blkEnd - blkStart;
        const int o1_in = o1, o2_in = o2;
        uint64_t* __restrict__ sq = P.seqs + (size_t)(blk0 + (uint32_t)b) * P.seq_stride;
        int nseq = 0, sumLL = 0;
        uint32_t rounds = 0;
        int nextEmit = blkStart, s = blkStart;
        uint32_t firstLL = 0, firstOf = 0;
        auto emit = [&](int ll, int ml3, uint32_t of) {
            if (nseq == 0) { firstLL = (uint32_t)ll; firstOf = of; }
            if (lig == 0) sq[nseq] = seq_pack((uint32_t)ll, (uint32_t)ml3, of);
            nseq++;
            sumLL += ll;
        };
        int SK = 5;  // kSearchStrength - 1
        if (hist0 > 0) {  // enc_fast.go:539-543,585
            if (allDirty || srcLen > (32 << 10)) allDirty = true; else SK = 6;
        }
        if (srcLen >= 10) {
            const int sLimit = blkEnd - 8;
            bool canRep = false, fin = false, pendO2 = false;
            int W = G;  // speculation width: narrow right after a match (hits come early in text), doubling on a miss
            while (!fin) {
                rounds++;
                const int d0 = s - nextEmit;
                const int k0 = d0 >> SK;
                const int step = 2 + k0;
                const int p = s + lig * step;
                const bool valid = lig < W && (lig == 0 || ((d0 + (lig - 1) * step) >> SK) == k0) && p < sLimit;
                const uint64_t cv = valid ? ld64(base + p) : 0ull;
                if (pendO2) {  // offset-2 check (enc_fast.go:250) sharing this round's source load
                    pendO2 = false;
                    const int o2pos = s - o2;
                    const uint32_t w2 = ld32(base + o2pos);
                    const uint64_t cv0 = gbcast64<G>(cv, grp, 0);
                    if (w2 == (uint32_t)cv0) {
                        const int l2 = 4 + grp_matchlen<G>(base, s + 4, o2pos + 4, blkEnd - (s + 4), lig, grp);
                        if (lig == 0) tab[hash6(cv0, ZF_TABLE_BITS)] = ((uint32_t)s + 1u) | (PB < 32 ? tagOf((uint32_t)cv0) << PB : 0u);
                        emit(0, l2 - 3, 1u);
                        W = P.spec_w0;
                        s += l2;
                        nextEmit = s;
                        const int tmp = o1; o1 = o2; o2 = tmp;
                        canRep = nseq > 2;
                        if (s >= sLimit) fin = true;
                        continue;
                    }
                }
                uint32_t h0 = 0xFFFFFFFFu, h1 = 0xFFFFFFFEu, c0 = 0, c1 = 0;
                if (valid) {
                    h0 = hash6(cv, ZF_TABLE_BITS);
                    h1 = hash6(cv >> 8, ZF_TABLE_BITS);
                    c0 = KC_TAB_LD(&tab[h0]);
                    c1 = KC_TAB_LD(&tab[h1]);
                }
                // exact in-round conflict detection: a lower lane of the group touches one of my buckets
                bool dep = false;
#pragma unroll
                for (int d = 1; d < G; d++) {
                    const uint32_t a0 = (uint32_t)__shfl_up((int)h0, d, G), a1 = (uint32_t)__shfl_up((int)h1, d, G);
                    if (lig >= d && (a0 == h0 || a0 == h1 || a1 == h0 || a1 == h1)) dep = true;
                }
                int kind = 0, t = 0;
                if (valid) {
                    const int repIndex = p - o1 + 2;
                    const bool repOk = canRep && repIndex >= 0;
                    const uint32_t e0 = c0 & posMask, e1 = c1 & posMask;
                    const int t0 = (int)e0 - 1, t1 = (int)e1 - 1;
                    const bool ok0 = e0 != 0 && (p - t0) < mmo && (PB >= 32 || (c0 >> PB) == tagOf((uint32_t)cv));
                    const bool ok1 = e1 != 0 && (p - t1 + 1) < mmo && (PB >= 32 || (c1 >> PB) == tagOf((uint32_t)(cv >> 8)));
                    // predicated candidate fetches: a masked-off lane costs no address slot in the texture path
                    uint32_t wr = 0, w0 = 0, w1 = 0;
                    if (repOk) wr = ld32(base + repIndex);
                    if (ok0) w0 = ld32(base + t0);
                    if (ok1) w1 = ld32(base + t1);
                    if (repOk && wr == (uint32_t)(cv >> 16)) kind = 1;
                    else if (ok0 && w0 == (uint32_t)cv) { kind = 2; t = t0; }
                    else if (ok1 && w1 == (uint32_t)(cv >> 8)) { kind = 3; t = t1; }
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
                }
                if (!found) {
                    W = P.spec_grow == 0 ? W : (P.spec_grow == 1 ? (W + 1 < G ? W + 1 : G) : ((2 * W < G) ? 2 * W : G));
                    if (c < nvalid) {
                        s = s + c * step;
                    } else {
                        const int pl = s + (nvalid - 1) * step;
                        s = pl + 2 + ((pl - nextEmit) >> SK);
                    }
                    if (s >= sLimit) fin = true;
                    continue;
                }
                const int mk = (int)gbcast32<G>((uint32_t)kind, grp, f);
                const int ps = s + f * step;
                int mt = (int)gbcast32<G>((uint32_t)t, grp, f);
                if (mk == 1) {
                    int repIndex = ps - o1 + 2;
                    const int length = 4 + grp_matchlen<G>(base, ps + 6, repIndex + 4, blkEnd - (ps + 6), lig, grp);
                    int start = ps + 2;
                    const int startLimit = nextEmit + 1;
                    const int sMin = (ps - mmo) > 0 ? (ps - mmo) : 0;
                    int kmax = repIndex - sMin;
                    if (start - startLimit < kmax) kmax = start - startLimit;
                    if (HIST) {
                        const int cap = (ZF_MAX_MATCH_LENGTH - 3) - (length - 3);
                        if (cap < kmax) kmax = cap;
                    }
                    if (kmax < 0) kmax = 0;
                    const int back = grp_backlen<G>(base, start, repIndex, kmax, lig, grp);
                    start -= back;
                    emit(start - nextEmit, length - 3 + back, 1u);
                    W = P.spec_w0;
                    s = ps + length + 2;
                    nextEmit = s;
                    if (s >= sLimit) fin = true;
                    continue;
                }
                s = ps + (mk == 3 ? 1 : 0);
                o2 = o1;
                o1 = s - mt;
                int l = grp_matchlen<G>(base, s + 4, mt + 4, blkEnd - (s + 4), lig, grp) + 4;
                {
                    const int tMin = (s - mmo) > 0 ? (s - mmo) : 0;
                    int kmax = mt - tMin;
                    if (s - nextEmit < kmax) kmax = s - nextEmit;
                    if (HIST && (ZF_MAX_MATCH_LENGTH - l) < kmax) kmax = ZF_MAX_MATCH_LENGTH - l;
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
                pendO2 = canRepO2;
            }
        }
        const int extra = nextEmit < blkEnd ? blkEnd - nextEmit : 0;
        const int nlit = sumLL + extra;
        const bool rle = nseq == 1 && nlit <= 1 && (int)firstLL == nlit && firstOf - 3u == 1u;
        const int saved = srcLen - nlit - (srcLen >> 6);
        uint32_t flags = 0;
        if (nseq > 0 && !rle && saved < 16) flags |= KC_BF_POP_A;
        if ((pm >> b) & 1u) flags |= KC_BF_FORCED;
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

void kc_launch_zfast_match_grp(const KcMatchParams& P, uint32_t* tables, uint32_t n_launch, int G, hipStream_t st) {
    if (G == 16) hipLaunchKernelGGL(kc_zfast_match_grp_kernel<16>, dim3((n_launch + 3) / 4), dim3(64), 0, st, P, tables, n_launch);
    else if (G == 4) hipLaunchKernelGGL(kc_zfast_match_grp_kernel<4>, dim3((n_launch + 15) / 16), dim3(64), 0, st, P, tables, n_launch);
    else if (G == 2) hipLaunchKernelGGL(kc_zfast_match_grp_kernel<2>, dim3((n_launch + 31) / 32), dim3(64), 0, st, P, tables, n_launch);
    else hipLaunchKernelGGL(kc_zfast_match_grp_kernel<8>, dim3((n_launch + 7) / 8), dim3(64), 0, st, P, tables, n_launch);
}

void kc_launch_zfast_match(const KcMatchParams& P, uint32_t grid, hipStream_t st, bool lds_variant) {
    if (lds_variant) hipLaunchKernelGGL(kc_zfast_match_lds_kernel, dim3(grid), dim3(64), 0, st, P);
    else hipLaunchKernelGGL(kc_zfast_match_kernel, dim3(grid), dim3(64), 0, st, P);
}
