// kc_zstd_prime.hip — the tables of a WithConcurrentBlocks job as ResetPrefix leaves them, built on the device from the job's
// overlap prefix (KcPrimeParams in kc_kernels.h).  One wave per table slot; the slot was zeroed by the caller.
//
// The reference inserts the prefix's positions in ascending order, one at a time (enc_fast.go:800-811, enc_dfast.go:1040-1050,
// enc_better.go:1099-1112).  Here a round takes 64 of them, one per lane.  For the plain tables the result of a round is "the
// highest position of a bucket stays": lanes store, read back, and the ones that find a lower position of this round in their
// bucket store again until none does (rounds follow each other in program order, so later rounds overwrite earlier ones like the
// reference's loop does).  The SpeedBetterCompression long table keeps {offset, prev} with prev = the bucket's previous offset:
// lanes mark their bucket in an LDS scratch first; the lanes that have their scratch cell to themselves (hence their bucket) insert
// in parallel, the lanes that shared a cell insert one after the other in position order.
#include "kc_dev.h"
#include "kc_kernels.h"

// KC_WAVE_SYNC through GLOBAL memory (a lane reads what another lane of its wave has just stored): the stores are waited for
#ifdef KC_HIPEMU
#define KC_MEM_SYNC() hipemu::wave_sync()
#else
#define KC_MEM_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

#define PRIME_SCR 16384  // scratch cells of the collision check (u16 lane ids)

namespace {
struct PrimeFmt {
    int PB, TB;
    __device__ __forceinline__ uint32_t tagOf(uint32_t v) const { return TB > 0 ? ((v * 2654435761u) >> (32 - TB)) : 0u; }
    __device__ __forceinline__ uint32_t mk(uint32_t pos, uint32_t val) const { return (pos + 1u) | (PB < 32 ? tagOf(val) << PB : 0u); }
    __device__ __forceinline__ uint32_t posOf(uint32_t e) const { return PB < 32 ? (e & ((1u << PB) - 1u)) : e; }
};

// one round of last-position-wins inserts: lane `valid` puts entry e (position field ascending with the lane) into tab[h]
__device__ __forceinline__ void prime_put(uint32_t* tab, uint32_t h, uint32_t e, bool valid, const PrimeFmt& F) {
    if (valid) tab[h] = e;
    for (;;) {
        KC_MEM_SYNC();
        const bool again = valid && F.posOf(tab[h]) < F.posOf(e);
        if (ballot64(again) == 0ull) break;
        if (again) tab[h] = e;
    }
}
}  // namespace

__global__ __launch_bounds__(64) void kc_zstd_prime_kernel(KcPrimeParams P) {
    __shared__ uint16_t scr[PRIME_SCR];
    const int lane = (int)threadIdx.x;
    const uint32_t slot = blockIdx.x;
    if (slot >= P.n_launch) return;
    const uint32_t u = P.unit_list != nullptr ? P.unit_list[slot] : P.unit_base + slot;
    const uint32_t n = P.unit_hist[u];  // the prefix: the first n bytes of the unit in src
    if (n < 8u) return;
    const uint8_t* __restrict__ prefix = P.src + P.unit_off[u];
    const uint32_t end = n - 8u;
    uint8_t* out = P.tables + (size_t)slot * P.table_bytes;
    PrimeFmt F;
    F.PB = P.pos_bits;
    F.TB = (32 - P.pos_bits) > 16 ? 16 : (32 - P.pos_bits);
    if (P.level == 3) {  // i = 0, 2, ...: long table with its chain, short table one byte on
        uint32_t* ltab = (uint32_t*)out;  // pairs {offset, prev}
        uint32_t* stab = (uint32_t*)(out + ((size_t)8 << 19));
        for (uint32_t i0 = 0; i0 < end; i0 += 128u) {
            const uint32_t i = i0 + 2u * (uint32_t)lane;
            const bool valid = i < end;
            const uint64_t cv = valid ? ld64(prefix + i) : 0ull;
            const uint32_t h = (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 19));
            const uint32_t e = F.mk(i, (uint32_t)cv);
            const uint32_t k = h & (PRIME_SCR - 1u);
            if (valid) scr[k] = (uint16_t)lane;
            KC_WAVE_SYNC();
            const uint32_t w = valid ? (uint32_t)scr[k] : 0u;
            const uint64_t cm = ballot64(valid && w != (uint32_t)lane);  // lanes that lost their cell ...
            uint64_t inv = cm;
            for (uint64_t m = cm; m != 0ull; m &= m - 1ull) inv |= 1ull << (rdlane32(w, ctz64(m)) & 63u);  // ... and the lanes they lost it to
            auto insert = [&]() {
                const uint32_t old = ltab[2u * h];
                ltab[2u * h] = e;
                ltab[2u * h + 1u] = old;
            };
            if (valid && ((inv >> lane) & 1ull) == 0ull) insert();  // alone in its cell, hence in its bucket
            for (uint64_t m = inv; m != 0ull; m &= m - 1ull) {       // the others in the reference's order
                KC_MEM_SYNC();
                if (ctz64(m) == lane && valid) insert();
            }
            KC_MEM_SYNC();  // (the next round reads what this one stored, from other lanes)
            const uint64_t v = cv >> 8;
            prime_put(stab, (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 13)), F.mk(i + 1u, (uint32_t)v), valid, F);
        }
        return;
    }
    // fastEncoder.ResetPrefix: every 4th position from 1, 6-byte hash, 2^15 entries; doubleFastEncoder embeds it — the same entries
    // land in ITS short table — and adds every 2nd position from 1 to the long table
    uint32_t* ftab = P.level == 2 ? (uint32_t*)(out + ((size_t)4 << 17)) : (uint32_t*)out;
    for (uint32_t i0 = 1; i0 < end; i0 += 256u) {
        const uint32_t i = i0 + 4u * (uint32_t)lane;
        const bool valid = i < end;
        const uint64_t cv = valid ? ld64(prefix + i) : 0ull;
        prime_put(ftab, (uint32_t)(((cv << 16) * 227718039650203ULL) >> (64 - 15)), F.mk(i, (uint32_t)cv), valid, F);
    }
    if (P.level == 2) {
        uint32_t* ltab = (uint32_t*)out;
        for (uint32_t i0 = 1; i0 < end; i0 += 128u) {
            const uint32_t i = i0 + 2u * (uint32_t)lane;
            const bool valid = i < end;
            const uint64_t cv = valid ? ld64(prefix + i) : 0ull;
            prime_put(ltab, (uint32_t)((cv * 0xcf1bbcdcb7a56463ULL) >> (64 - 17)), F.mk(i, (uint32_t)cv), valid, F);
        }
    }
}

void kc_launch_zstd_prime(const KcPrimeParams& P, hipStream_t st) {
    if (P.n_launch == 0) return;
    hipLaunchKernelGGL(kc_zstd_prime_kernel, dim3(P.n_launch), dim3(64), 0, st, P);
}
