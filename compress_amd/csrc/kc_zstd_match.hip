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
    const uint32_t u = P.unit_list ? P.unit_list[blockIdx.x] : blockIdx.x;
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

void kc_launch_zfast_match(const KcMatchParams& P, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL(kc_zfast_match_kernel, dim3(grid), dim3(64), 0, st, P);
}
