// kc_zstd_prescan.hip — SpeedFastest: units that cannot contain a match are recognised, and their frames written, without a
// hash table in HBM (KcPrescanParams in kc_kernels.h has the argument).  One wave per unit; the unit's (bucket, value) pairs go
// into an open-addressed set of 64-bit keys in LDS (compare-and-swap inserts: the order of the inserts does not matter for
// "is any pair inserted twice"), every probe's 8 source bytes are loaded before the first one is hashed (the positions are known
// up front: one DRAM round trip per unit).
#include "kc_dev.h"
#include "kc_kernels.h"
#include "kc_zfast_dev.h"
#include "kc_frame_dev.h"

#define PS_SLOTS (2 * KC_PRESCAN_MAX_KEYS)
#define PS_PER_LANE (KC_PRESCAN_MAX_KEYS / 2 / 64)  // probes per lane at most

__global__ __launch_bounds__(64) void kc_zfast_prescan_kernel(KcPrescanParams P) {
    __shared__ unsigned long long set[PS_SLOTS];
    const int lane = (int)threadIdx.x;
    const uint32_t u = blockIdx.x;
    if (u >= P.n_units) return;
    const uint8_t* __restrict__ base = P.src + P.unit_off[u];
    const int ulen = (int)(P.unit_off[u + 1] - P.unit_off[u]);
    const int bs = P.block_size;
    const int nblk = (ulen + bs - 1) / bs;
    const uint32_t blk0 = P.unit_blk0[u];
    // probes of the unit: every full block has P.n_probe of them; the last (short) block those below its length - 8
    // (a block below minNonLiteralBlockSize = 10 bytes is not probed at all, enc_fast.go:52)
    const int lastLen = ulen - (nblk - 1) * bs;
    int nLast = 0;
    if (nblk > 0) {
        if (lastLen == bs) nLast = (int)P.n_probe;
        else if (lastLen >= 10) {  // count of probe_rel[k] < lastLen - 8 (ascending): binary search, wave-uniform
            int lo = 0, hi = (int)P.n_probe;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)P.probe_rel[mid] < lastLen - 8) lo = mid + 1; else hi = mid; }
            nLast = lo;
        }
    }
    const int nFull = nblk > 0 ? nblk - 1 : 0;
    const long long total = (long long)nFull * (long long)P.n_probe + nLast;
    if (ulen == 0 || 2 * total > KC_PRESCAN_MAX_KEYS) {  // empty units are the entropy kernel's; long units the match finder's
        if (lane == 0) P.unit_done[u] = 0u;
        return;
    }
    for (int i = lane; i < PS_SLOTS; i += 64) set[i] = 0ull;
    KC_WAVE_SYNC();
    uint64_t cv[PS_PER_LANE];
#pragma unroll
    for (int j = 0; j < PS_PER_LANE; j++) {
        const int idx = j * 64 + lane;
        cv[j] = 0;
        if (idx < (int)total) {
            const int b = P.n_probe ? idx / (int)P.n_probe : 0;
            const int k = idx - b * (int)P.n_probe;
            cv[j] = ld64(base + (size_t)b * (size_t)bs + P.probe_rel[k]);  // (the probe is below the block's end - 8)
        }
    }
    bool dup = false;
#pragma unroll
    for (int j = 0; j < PS_PER_LANE; j++) {
        const int idx = j * 64 + lane;
        if (idx < (int)total) {
#pragma unroll
            for (int w = 0; w < 2; w++) {  // the probe inserts (hash of 6 bytes at s, 4 bytes at s) and the same one byte on (enc_fast.go:127-132)
                const uint64_t c = cv[j] >> (8 * w);
                const uint32_t h = hash6(c, ZF_TABLE_BITS);
                const unsigned long long key = (1ull << 63) | ((unsigned long long)h << 32) | (unsigned long long)(uint32_t)c;
                uint32_t slot = (((uint32_t)c * 2654435761u) ^ (h * 0x9E3779B1u)) >> (32 - 11);
                for (;;) {
                    const unsigned long long old = atomicCAS(&set[slot], 0ull, key);
                    if (old == 0ull) break;
                    if (old == key) { dup = true; break; }
                    slot = (slot + 1u) & (PS_SLOTS - 1);
                }
            }
        }
    }
    static_assert(PS_SLOTS == 2048, "the slot hash above keeps 11 bits");
    const bool any = ballot64(dup) != 0ull;
    if (any) {
        if (lane == 0) P.unit_done[u] = 0u;
        return;
    }
    // ---- no sequence anywhere in the unit: block records, frame header, raw block headers, payload descriptors ----
    if (lane == 0) {
        uint8_t* outp = P.stage + P.stage_off[u];
        uint8_t hdr[16];
        int opos = kc_frame_header(hdr, ulen, P.window_size, P.single, P.crc, P.dict_id, false);
        for (int i = 0; i < opos; i++) outp[i] = hdr[i];
        for (int b = 0; b < nblk; b++) {
            const int blkStart = b * bs;
            const int size = (b == nblk - 1) ? lastLen : bs;
            KcBlkMeta m;
            m.nseq = 0; m.nlit = (uint32_t)size; m.extra_lits = (uint32_t)size; m.flags = 0;
            m.o1_in = (uint32_t)P.rep1; m.o2_in = (uint32_t)P.rep2; m.o1_out = (uint32_t)P.rep1; m.o2_out = (uint32_t)P.rep2;
            P.meta[blk0 + (uint32_t)b] = m;
            put_block_header(outp + opos, b == nblk - 1, 0u, (uint32_t)size);
            KcRawDef r;
            r.frame_pos = (uint32_t)(opos + 3); r.src_pos = (uint32_t)blkStart; r.size = (uint32_t)size; r.pad = 0;
            P.rawdef[blk0 + (uint32_t)b] = r;
            opos += 3 + size;
        }
        if (P.crc) opos += 4;  // (kc_xxh64_fin_kernel fills the field in)
        P.out_size[u] = (uint32_t)opos;
        P.unit_raw[u] = 1u;
        P.unit_done[u] = 1u;
    }
}

void kc_launch_zfast_prescan(const KcPrescanParams& P, hipStream_t st) {
    if (P.n_units == 0) return;
    hipLaunchKernelGGL(kc_zfast_prescan_kernel, dim3(P.n_units), dim3(64), 0, st, P);
}

uint32_t kc_zfast_probe_positions(int block_size, uint32_t* rel, uint32_t cap) {
    // fastEncoder.Encode from a block start with nothing emitted yet: s = nextEmit = start; probes while s < len - 8
    // (sLimit, enc_fast.go:96), s += stepSize + ((s - nextEmit) >> (kSearchStrength - 1)) = 2 + (s >> 5) (enc_fast.go:207)
    uint32_t n = 0;
    if (block_size < 10) return 0;
    for (int s = 0; s < block_size - 8; s += 2 + (s >> 5)) {
        if (n < cap) rel[n] = (uint32_t)s;
        n++;
    }
    return n;
}
