// kc_misc.hip — frame checksum (XXH64), size scan and output compaction kernels.
#include "kc_dev.h"
#include "kc_kernels.h"

// ---------------------------------------------------------------------------------------
// XXH64 (seed 0) per unit — replaces xxhash.Digest.Write/Sum64
// (zstd/internal/xxhash/xxhash.go:61-156).  The four accumulators of XXH64 are independent
// 8-byte lanes of a 32-byte stripe, each a serial multiply-rotate chain, so one unit maps to
// 4 lanes (one per accumulator) and a wave hashes 16 units at a time; the 32-byte stripes of a
// unit are read as 4 adjacent 8-byte loads.
// ---------------------------------------------------------------------------------------
#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
#define XP3 1609587929392839161ULL
#define XP4 9650029242287828579ULL
#define XP5 2870177450012600261ULL
__device__ __forceinline__ uint64_t xrol(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t input) { return xrol(acc + input * XP2, 31) * XP1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t acc, uint64_t val) { return (acc ^ xround(0, val)) * XP1 + XP4; }

__global__ __launch_bounds__(256) void kc_xxh64_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ unit_off,
                                                       uint32_t n_units, uint64_t* __restrict__ out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t u = gt >> 2;
    const int a = (int)(gt & 3);
    const bool active = u < n_units;
    const uint8_t* p = src;
    uint64_t len = 0;
    if (active) { p = src + unit_off[u]; len = unit_off[u + 1] - unit_off[u]; }
    uint64_t v = a == 0 ? XP1 + XP2 : (a == 1 ? XP2 : (a == 2 ? 0ULL : 0ULL - XP1));
    const uint64_t stripes = len >> 5;
    // Two stripes (64 bytes) per step: the four lanes of a unit load 16 bytes each — one 64-byte line per unit and load instead of
    // four 8-byte pieces 32 bytes apart — and every lane picks its accumulator's word of either stripe out of its quad with DPP
    // quad permutes (stripe 0 sits in lanes 0,1, stripe 1 in lanes 2,3; word a of a stripe is the (a & 1) half of lane a >> 1).
    // Four steps are in flight per lane (256 bytes per unit): with 8 waves per CU the kernel is bound by bytes in flight.
    const uint64_t pairs = stripes >> 1;
    auto quad = [&](uint32_t x, bool second) -> uint32_t {  // lane a reads lane (second ? 2 : 0) + (a >> 1) of its quad
        return second ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xFA, 0xF, 0xF, true)   // quad_perm:[2,2,3,3]
                      : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x50, 0xF, 0xF, true);  // quad_perm:[0,0,1,1]
    };
    auto word = [&](const uint4 w, bool second) -> uint64_t {
        const uint32_t x = quad(w.x, second), y = quad(w.y, second), z = quad(w.z, second), t = quad(w.w, second);
        return (a & 1) ? ((uint64_t)z | ((uint64_t)t << 32)) : ((uint64_t)x | ((uint64_t)y << 32));
    };
    const uint8_t* q16 = p + 16 * a;
    uint64_t i = 0;
    for (; i + 4 <= pairs; i += 4) {  // (the trip count is the same for the four lanes of a unit; other units' lanes run on under the mask)
        const uint4 w0 = ld128u(q16 + (i << 6)), w1 = ld128u(q16 + ((i + 1) << 6)), w2 = ld128u(q16 + ((i + 2) << 6)), w3 = ld128u(q16 + ((i + 3) << 6));
        v = xround(v, word(w0, false)); v = xround(v, word(w0, true));
        v = xround(v, word(w1, false)); v = xround(v, word(w1, true));
        v = xround(v, word(w2, false)); v = xround(v, word(w2, true));
        v = xround(v, word(w3, false)); v = xround(v, word(w3, true));
    }
    for (; i < pairs; i++) {
        const uint4 w0 = ld128u(q16 + (i << 6));
        v = xround(v, word(w0, false)); v = xround(v, word(w0, true));
    }
    const uint8_t* q = p + 8 * a;
    if (stripes & 1) v = xround(v, ld64(q + ((stripes - 1) << 5)));
    // combine the 4 accumulators of this unit (4 adjacent lanes)
    const int lane = (int)(threadIdx.x & 63);
    const int l0 = lane & ~3;
    const uint64_t v1 = bcast64(v, l0), v2 = bcast64(v, l0 + 1), v3 = bcast64(v, l0 + 2), v4 = bcast64(v, l0 + 3);
    if (!active || a != 0) return;
    uint64_t h;
    if (len >= 32) {
        h = xrol(v1, 1) + xrol(v2, 7) + xrol(v3, 12) + xrol(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = XP5;  // v3 + prime5 with v3 == 0
    }
    h += len;
    const uint8_t* t = p + (stripes << 5);
    int rem = (int)(len & 31);
    for (; rem >= 8; t += 8, rem -= 8) { h ^= xround(0, ld64(t)); h = xrol(h, 27) * XP1 + XP4; }
    if (rem >= 4) { h ^= (uint64_t)ld32(t) * XP1; h = xrol(h, 23) * XP2 + XP3; t += 4; rem -= 4; }
    for (; rem > 0; t++, rem--) { h ^= (uint64_t)t[0] * XP5; h = xrol(h, 11) * XP1; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    out[u] = h;
}
void kc_launch_xxh64(const uint8_t* src, const uint64_t* unit_off, uint32_t n_units, uint64_t* out, hipStream_t st) {
    if (n_units == 0) return;
    const uint32_t threads = n_units * 4;
    hipLaunchKernelGGL(kc_xxh64_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, src, unit_off, n_units, out);
}

// ---------------------------------------------------------------------------------------
// XXH64 at the END of the pipeline (round 3): the checksum is the last field of the frame, so nothing needs it before the frame
// is assembled.  Run behind the entropy stage and the size scan, this kernel knows which frames consist of raw blocks only
// (incompressible units) and copies their payload into the final output WHILE it hashes it — the bytes are read once instead of
// twice (checksum pass + compaction).  The four lanes of a unit hold one 64-byte line per load (see kc_xxh64_kernel); each lane
// stores its 16 bytes at the line's place in the frame.  The low 32 bits of the hash go into the frame's staging slot, from where
// the compaction takes them with the headers.
// ---------------------------------------------------------------------------------------
#ifdef KC_HIPEMU
#define KC_QUAD_SYNC() hipemu::wave_sync()  // (the emulator runs this kernel with the quads as rendezvous groups)
#else
#define KC_QUAD_SYNC() __builtin_amdgcn_wave_barrier()  // the hardware runs a wave's LDS instructions in order: only the compiler must not reorder
#endif
struct __attribute__((packed)) kc_u128s { uint32_t x, y, z, w; };
__device__ __forceinline__ void st128u(uint8_t* p, const uint4 v) {  // unaligned 16-byte store
    kc_u128s t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(kc_u128s*)p = t;
}

#define XF_RING 512  // MODE 2: bytes of a unit staged in LDS (two 256-byte steps; the last 16 bytes of the previous step are still needed)
// MODE 0: four loads, four stores, sixteen hash rounds per 256-byte step.  MODE 1: the same, software-pipelined.  MODE 2: the step's
// bytes go through an LDS ring per unit and leave as 16-byte ALIGNED stores (the payload's place in the frame is not aligned to the
// source: stored straight from the registers, every 64-byte store of a quad straddles two lines and every line is written twice).
template <int MODE>
__global__ __launch_bounds__(256) void kc_xxh64_fin_kernel(KcXxhFinParams P) {
    constexpr bool PIPE = MODE == 1;
    __shared__ __attribute__((aligned(16))) uint8_t stg_all[MODE == 2 ? 64 * (XF_RING + 16) : 16];
    uint8_t* const stg = stg_all + (MODE == 2 ? (threadIdx.x >> 2) * (XF_RING + 16) : 0);
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t u = gt >> 2;
    const int a = (int)(gt & 3);
    const bool active = u < P.n_units;
    const uint8_t* p = P.src;
    uint64_t len = 0;
    bool raw = false;
    if (active) {
        p = P.src + P.unit_off[u];
        len = P.unit_off[u + 1] - P.unit_off[u];
        raw = P.unit_raw != nullptr && P.unit_raw[u] != 0u && len > 0;
    }
    // raw units: source byte x of block b goes to dst + out_off[u] + rawdef[b].frame_pos + (x - rawdef[b].src_pos); the block size
    // is a multiple of 256 (host), so neither a 256-byte step nor a 64-byte line straddles two blocks
    uint8_t* dbase = raw ? P.dst + P.out_off[u] : nullptr;
    uint32_t rb = raw ? P.unit_blk0[u] : 0u;
    uint64_t bend = 0;       // end (unit offset) of the block the shift below belongs to
    int64_t shift = 0;       // frame position minus unit offset inside the current block
    auto block_of = [&](uint64_t x) {  // x is the first byte of a new block (x == bend)
        const KcRawDef r = P.rawdef[rb++];
        shift = (int64_t)r.frame_pos - (int64_t)r.src_pos;
        bend = (uint64_t)r.src_pos + r.size;
    };
    uint64_t v = a == 0 ? XP1 + XP2 : (a == 1 ? XP2 : (a == 2 ? 0ULL : 0ULL - XP1));
    const uint64_t stripes = len >> 5;
    const uint64_t pairs = stripes >> 1;
    auto quad = [&](uint32_t x, bool second) -> uint32_t {
        return second ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xFA, 0xF, 0xF, true)
                      : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x50, 0xF, 0xF, true);
    };
    auto word = [&](const uint4 w, bool second) -> uint64_t {
        const uint32_t x = quad(w.x, second), y = quad(w.y, second), z = quad(w.z, second), t = quad(w.w, second);
        return (a & 1) ? ((uint64_t)z | ((uint64_t)t << 32)) : ((uint64_t)x | ((uint64_t)y << 32));
    };
    const uint8_t* q16 = p + 16 * a;
    uint64_t i = 0;
    if (MODE == 2) {
        uint8_t* dcur = nullptr;  // where unit byte 0 would sit in the frame under the current block's shift
        int64_t hi = 0;           // aligned stores have covered the frame up to dcur + hi (a 16-byte aligned address)
        uint32_t al = 0;          // dcur modulo 16
        bool staged = false;
        auto bytes_out = [&](int64_t from, int64_t to) {  // unit bytes [from, to): fewer than 16, lane 0 of the quad
            if (a == 0) for (int64_t q = from; q < to; q++) dcur[q] = p[q];
        };
        for (; i + 4 <= pairs; i += 4) {
            const uint4 c0 = ld128u(q16 + (i << 6)), c1 = ld128u(q16 + ((i + 1) << 6)), c2 = ld128u(q16 + ((i + 2) << 6)), c3 = ld128u(q16 + ((i + 3) << 6));
            if (raw) {
                const int64_t x = (int64_t)(i << 6);
                if ((uint64_t)x >= bend) {  // a new block (its first byte is the first byte of this step)
                    if (staged) bytes_out(hi, x);  // what the previous block's last aligned word did not reach
                    block_of((uint64_t)x);
                    dcur = dbase + shift;
                    al = (uint32_t)((uintptr_t)dcur & 15);
                    hi = ((x + al + 15) & ~(int64_t)15) - al;
                    bytes_out(x, hi);
                    staged = true;
                }
                const uint32_t o = ((uint32_t)x + 16u * (uint32_t)a) & (XF_RING - 1);
                *(uint4*)(stg + o) = c0;
                *(uint4*)(stg + ((o + 64) & (XF_RING - 1))) = c1;
                *(uint4*)(stg + ((o + 128) & (XF_RING - 1))) = c2;
                *(uint4*)(stg + ((o + 192) & (XF_RING - 1))) = c3;
                if (o == 0) *(uint4*)(stg + XF_RING) = c0;  // the ring's first 16 bytes again behind its end: a 16-byte read never wraps
                KC_QUAD_SYNC();
                const int64_t nhi = ((x + 256 + al) & ~(int64_t)15) - al;
                const bool last = a < 3 || (nhi - hi) == 256;  // 15 or 16 aligned words: only the quad's very last one may be missing
                const int64_t s0 = hi + 16 * a;                // this lane's words: s0, s0 + 64, s0 + 128, s0 + 192 (unit offsets)
                const uint4 w0 = ld128u(stg + ((uint32_t)s0 & (XF_RING - 1)));
                const uint4 w1 = ld128u(stg + ((uint32_t)(s0 + 64) & (XF_RING - 1)));
                const uint4 w2 = ld128u(stg + ((uint32_t)(s0 + 128) & (XF_RING - 1)));
                const uint4 w3 = ld128u(stg + ((uint32_t)(s0 + 192) & (XF_RING - 1)));
                uint4* dq = (uint4*)(dcur + s0);  // 16-byte aligned
                dq[0] = w0; dq[4] = w1; dq[8] = w2;
                if (last) dq[12] = w3;
                hi = nhi;
                KC_QUAD_SYNC();
            }
            v = xround(v, word(c0, false)); v = xround(v, word(c0, true));
            v = xround(v, word(c1, false)); v = xround(v, word(c1, true));
            v = xround(v, word(c2, false)); v = xround(v, word(c2, true));
            v = xround(v, word(c3, false)); v = xround(v, word(c3, true));
        }
        if (raw && staged) bytes_out(hi, (int64_t)(i << 6));  // (the rest of the unit, below, is stored straight from the registers)
    } else if (MODE == 3) {
        // MODE 1 with twice the bytes in flight: the kernel's parallelism is fixed by XXH64 (four accumulators = four lanes per unit:
        // 8 waves per CU whatever the launch shape), so what it keeps in flight is lanes x loads per lane — 4 x 16 bytes per lane are
        // 8 MiB on the chip, which at a ~2 us round trip is the 4 TB/s MODE 1 measures; eight loads of the next step are issued
        // before the current step's eight stores and 32 hash rounds.
        uint4 c[8], n[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = make_uint4(0, 0, 0, 0);
        if (pairs >= 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = ld128u(q16 + 64 * k);
        }
        for (; i + 8 <= pairs; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) n[k] = c[k];
            if (i + 16 <= pairs) {
                const uint8_t* qq = q16 + ((i + 8) << 6);
#pragma unroll
                for (int k = 0; k < 8; k++) n[k] = ld128u(qq + 64 * k);
            }
            if (raw) {  // (the block size is a multiple of 256: a 256-byte half step never straddles two blocks)
#pragma unroll
                for (int hlf = 0; hlf < 2; hlf++) {
                    const uint64_t x = (i + 4 * hlf) << 6;
                    if (x >= bend) block_of(x);
                    uint8_t* d = dbase + (int64_t)x + shift + 16 * a;
                    st128u(d, c[4 * hlf]); st128u(d + 64, c[4 * hlf + 1]); st128u(d + 128, c[4 * hlf + 2]); st128u(d + 192, c[4 * hlf + 3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) { v = xround(v, word(c[k], false)); v = xround(v, word(c[k], true)); }
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = n[k];
        }
    } else if (PIPE) {
        // software-pipelined: the loads of step i+1 are issued BEFORE the stores of step i.  gfx9 counts loads and stores in one
        // in-order counter (vmcnt), so with the stores first every wait for a load also waits for the (slower, misaligned) stores in
        // front of it; this way the stores of a step drain under its sixteen hash rounds and under the next step's loads.
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, c2 = c0, c3 = c0;
        if (pairs >= 4) { c0 = ld128u(q16); c1 = ld128u(q16 + 64); c2 = ld128u(q16 + 128); c3 = ld128u(q16 + 192); }
        for (; i + 4 <= pairs; i += 4) {
            uint4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
            if (i + 8 <= pairs) {
                const uint8_t* qq = q16 + ((i + 4) << 6);
                n0 = ld128u(qq); n1 = ld128u(qq + 64); n2 = ld128u(qq + 128); n3 = ld128u(qq + 192);
            }
            if (raw) {
                const uint64_t x = i << 6;
                if (x >= bend) block_of(x);
                uint8_t* d = dbase + (int64_t)x + shift + 16 * a;
                st128u(d, c0); st128u(d + 64, c1); st128u(d + 128, c2); st128u(d + 192, c3);
            }
            v = xround(v, word(c0, false)); v = xround(v, word(c0, true));
            v = xround(v, word(c1, false)); v = xround(v, word(c1, true));
            v = xround(v, word(c2, false)); v = xround(v, word(c2, true));
            v = xround(v, word(c3, false)); v = xround(v, word(c3, true));
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
    } else
    for (; i + 4 <= pairs; i += 4) {
        const uint4 w0 = ld128u(q16 + (i << 6)), w1 = ld128u(q16 + ((i + 1) << 6)), w2 = ld128u(q16 + ((i + 2) << 6)), w3 = ld128u(q16 + ((i + 3) << 6));
        if (raw) {
            const uint64_t x = i << 6;
            if (x >= bend) block_of(x);
            uint8_t* d = dbase + (int64_t)x + shift + 16 * a;
            st128u(d, w0); st128u(d + 64, w1); st128u(d + 128, w2); st128u(d + 192, w3);
        }
        v = xround(v, word(w0, false)); v = xround(v, word(w0, true));
        v = xround(v, word(w1, false)); v = xround(v, word(w1, true));
        v = xround(v, word(w2, false)); v = xround(v, word(w2, true));
        v = xround(v, word(w3, false)); v = xround(v, word(w3, true));
    }
    for (; i < pairs; i++) {
        const uint4 w0 = ld128u(q16 + (i << 6));
        if (raw) {
            const uint64_t x = i << 6;
            if (x >= bend) block_of(x);
            st128u(dbase + (int64_t)x + shift + 16 * a, w0);
        }
        v = xround(v, word(w0, false)); v = xround(v, word(w0, true));
    }
    const uint8_t* q = p + 8 * a;
    if (stripes & 1) v = xround(v, ld64(q + ((stripes - 1) << 5)));
    const int lane = (int)(threadIdx.x & 63);
    const int l0 = lane & ~3;
    const uint64_t v1 = bcast64(v, l0), v2 = bcast64(v, l0 + 1), v3 = bcast64(v, l0 + 2), v4 = bcast64(v, l0 + 3);
    if (!active || a != 0) return;
    if (raw) {  // the last (len mod 64) bytes of the unit
        for (uint64_t x = pairs << 6; x < len; x++) {
            if (x >= bend) block_of(x);
            dbase[(int64_t)x + shift] = p[x];
        }
    }
    uint64_t h;
    if (len >= 32) {
        h = xrol(v1, 1) + xrol(v2, 7) + xrol(v3, 12) + xrol(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = XP5;
    }
    h += len;
    const uint8_t* t = p + (stripes << 5);
    int rem = (int)(len & 31);
    for (; rem >= 8; t += 8, rem -= 8) { h ^= xround(0, ld64(t)); h = xrol(h, 27) * XP1 + XP4; }
    if (rem >= 4) { h ^= (uint64_t)ld32(t) * XP1; h = xrol(h, 23) * XP2 + XP3; t += 4; rem -= 4; }
    for (; rem > 0; t++, rem--) { h ^= (uint64_t)t[0] * XP5; h = xrol(h, 11) * XP1; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    if (P.xxh_out != nullptr) P.xxh_out[u] = h;
    const uint32_t fsz = P.out_size[u];
    if (len > 0 && fsz >= 4) {  // enc_base.go:34-38: the frame ends in the low four bytes of the digest
        uint8_t* c = P.stage + P.stage_off[u] + fsz - 4;
        c[0] = (uint8_t)h; c[1] = (uint8_t)(h >> 8); c[2] = (uint8_t)(h >> 16); c[3] = (uint8_t)(h >> 24);
    }
}
void kc_launch_xxh64_fin(const KcXxhFinParams& P, hipStream_t st) {
    if (P.n_units == 0) return;
    const uint32_t threads = P.n_units * 4;
    if (P.mode == 3) hipLaunchKernelGGL(kc_xxh64_fin_kernel<3>, dim3((threads + 255) / 256), dim3(256), 0, st, P);
    else if (P.mode == 2) hipLaunchKernelGGL(kc_xxh64_fin_kernel<2>, dim3((threads + 255) / 256), dim3(256), 0, st, P);
    else if (P.mode == 1) hipLaunchKernelGGL(kc_xxh64_fin_kernel<1>, dim3((threads + 255) / 256), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(kc_xxh64_fin_kernel<0>, dim3((threads + 255) / 256), dim3(256), 0, st, P);
}

// ---------------------------------------------------------------------------------------
// the small per-batch flag arrays (re-run masks, error flag, raw-block descriptors, ...) zeroed by ONE launch: five
// hipMemsetAsync calls cost ~30 us each on the stream, which is 4 % of a 4 ms high-entropy batch
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kc_clear_kernel(KcClearList L) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < L.count; k++) {
        uint4* p = (uint4*)L.p[k];
        const uint64_t n16 = L.n16[k];
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) p[i] = z;
    }
}
void kc_launch_clear(const KcClearList& L, hipStream_t st) {
    uint64_t mx = 0;
    for (int k = 0; k < L.count; k++) mx = L.n16[k] > mx ? L.n16[k] : mx;
    if (mx == 0) return;
    uint64_t grid = (mx + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(kc_clear_kernel, dim3((unsigned)grid), dim3(256), 0, st, L);
}

// ---------------------------------------------------------------------------------------
// exclusive scan of per-unit sizes -> output offsets.  ONE wave (round 6; before: one workgroup of 16 waves): n is at most a few
// 1e5, and a 16-wave workgroup needs 16 free wave slots on ONE CU — beside other streams' long-running one-wave workgroups (the
// rolling host pipeline: three other lanes' match finders / S2 encoders fill every slot that frees up) it waited 5-25 ms for a
// 50 us job (gpurun_out/r6d/timeline_C4.txt).  A single wave fits any slot.  Eight sizes per lane and step.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void kc_scan_sizes_kernel(const uint32_t* __restrict__ sizes, uint32_t n, uint64_t* __restrict__ out_off) {
    const int lane = threadIdx.x & 63;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += 512) {
        const uint32_t i0 = base + (uint32_t)lane * 8u;
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = i0 + k < n ? sizes[i0 + k] : 0u;
        uint64_t mine = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) mine += v[k];
        uint64_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d, 64);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d, 64);
            if (lane >= d) inc += ((uint64_t)hi << 32) | lo;
        }
        uint64_t run = carry + inc - mine;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k < n) out_off[i0 + k] = run;
            run += v[k];
        }
        const uint32_t tlo = (uint32_t)__shfl((int)(uint32_t)inc, 63, 64), thi = (uint32_t)__shfl((int)(uint32_t)(inc >> 32), 63, 64);
        carry += ((uint64_t)thi << 32) | tlo;
    }
    if (lane == 0) out_off[n] = carry;
}
void kc_launch_scan_sizes(const uint32_t* sizes, uint32_t n, uint64_t* out_off, hipStream_t st) {
    hipLaunchKernelGGL(kc_scan_sizes_kernel, dim3(1), dim3(64), 0, st, sizes, n, out_off);
}

// ---------------------------------------------------------------------------------------
// compaction: variable-size frames from the per-unit staging slots to the contiguous output
// ---------------------------------------------------------------------------------------
// s[0, len) -> d[0, len): 16-byte stores once the destination is aligned, unaligned source loads
__device__ __forceinline__ void kc_copy_bytes(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, uint32_t len, int tid) {
    if (len == 0) return;
    uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > len) head = len;
    for (uint32_t i = tid; i < head; i += 256) d[i] = s[i];
    const uint32_t body = (len - head) >> 4;
    const uint8_t* sb = s + head;
    uint4* db = (uint4*)(d + head);
    for (uint32_t i = tid; i < body; i += 256) db[i] = ld128u(sb + ((size_t)i << 4));
    for (uint32_t i = head + (body << 4) + tid; i < len; i += 256) d[i] = s[i];
}

__global__ __launch_bounds__(256) void kc_compact_kernel(const uint8_t* __restrict__ stage, const uint64_t* __restrict__ stage_off,
                                                         const uint32_t* __restrict__ sizes, const uint64_t* __restrict__ out_off,
                                                         uint8_t* __restrict__ dst, uint32_t n, const uint8_t* __restrict__ src,
                                                         const uint64_t* __restrict__ unit_off, const uint32_t* __restrict__ unit_blk0,
                                                         const KcRawDef* __restrict__ rawdef, const uint32_t* __restrict__ unit_raw) {
    const uint32_t u = blockIdx.x;
    if (u >= n) return;
    const uint8_t* s = stage + stage_off[u];  // 16-byte aligned
    uint8_t* d = dst + out_off[u];
    const uint32_t len = sizes[u];
    const int tid = threadIdx.x;
    uint32_t cursor = 0;
    if (rawdef != nullptr) {  // raw blocks: the staging slot holds their 3-byte header, the payload comes from the source, once
        const uint8_t* us = src + unit_off[u];
        const uint32_t b1 = unit_blk0[u + 1];
        const bool placed = unit_raw != nullptr && unit_raw[u] != 0u;  // kc_xxh64_fin_kernel has put the payloads where they belong
        for (uint32_t b = unit_blk0[u]; b < b1; b++) {
            const KcRawDef r = rawdef[b];
            if (r.size == 0 || r.frame_pos < cursor || r.frame_pos + r.size > len) continue;
            kc_copy_bytes(d + cursor, s + cursor, r.frame_pos - cursor, tid);
            if (!placed) kc_copy_bytes(d + r.frame_pos, us + r.src_pos, r.size, tid);
            cursor = r.frame_pos + r.size;
        }
    }
    kc_copy_bytes(d + cursor, s + cursor, len - cursor, tid);
}
void kc_launch_compact(const uint8_t* stage, const uint64_t* stage_off, const uint32_t* sizes, const uint64_t* out_off,
                       uint8_t* dst, uint32_t n, hipStream_t st, const uint8_t* src, const uint64_t* unit_off, const uint32_t* unit_blk0,
                       const KcRawDef* rawdef, const uint32_t* unit_raw) {
    if (n == 0) return;
    hipLaunchKernelGGL(kc_compact_kernel, dim3(n), dim3(256), 0, st, stage, stage_off, sizes, out_off, dst, n, src, unit_off, unit_blk0, rawdef, unit_raw);
}

// ---------------------------------------------------------------------------------------
// dictionary support: prefix every unit with the dictionary content (history), broadcast primed tables
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kc_prefix_units_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ unit_off,
                                                              const uint64_t* __restrict__ work_off, const uint8_t* __restrict__ dict,
                                                              uint32_t dict_len, uint8_t* __restrict__ work, uint32_t n) {
    const uint32_t u = blockIdx.x;
    if (u >= n) return;
    uint8_t* w = work + work_off[u];
    const uint8_t* s = src + unit_off[u];
    const uint32_t len = (uint32_t)(unit_off[u + 1] - unit_off[u]);
    kc_copy_bytes(w, dict, dict_len, (int)threadIdx.x);  // (16-byte stores; a byte per thread and step made this kernel 1.7 ms per GiB of C5)
    kc_copy_bytes(w + dict_len, s, len, (int)threadIdx.x);
}
void kc_launch_prefix_units(const uint8_t* src, const uint64_t* unit_off, const uint64_t* work_off, const uint8_t* dict, uint32_t dict_len,
                            uint8_t* work, uint32_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(kc_prefix_units_kernel, dim3(n), dim3(256), 0, st, src, unit_off, work_off, dict, dict_len, work, n);
}
#ifndef KC_BCAST_V
#define KC_BCAST_V 1   // 0: measurement builds: one 16-byte word per thread, plain stores (4 KiB per workgroup and table)
#endif
// The dictionary-primed table image (4 MiB for SpeedBetter) written into every unit's table slot: a pure HBM write stream.  Each
// workgroup holds 16 KiB of the image in registers (four 16-byte words per thread, read once) and writes it to its share of the
// slots as four contiguous 4 KiB pieces per slot with non-temporal stores (nothing reads the lines before the match finder does).
__global__ __launch_bounds__(256) void kc_bcast_kernel(const uint4* __restrict__ proto, uint4* __restrict__ dst, size_t n16, uint32_t n) {
#if KC_BCAST_V
    const size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = i0 + 256u * q < n16 ? proto[i0 + 256u * q] : make_uint4(0, 0, 0, 0);
    for (uint32_t k = blockIdx.y; k < n; k += gridDim.y) {
        uint4* d = dst + (size_t)k * n16 + i0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (i0 + 256u * q < n16) {
#ifdef KC_HIPEMU
                d[256u * q] = v[q];
#else
                typedef uint32_t kc_v4u __attribute__((ext_vector_type(4)));
                kc_v4u w; w.x = v[q].x; w.y = v[q].y; w.z = v[q].z; w.w = v[q].w;
                __builtin_nontemporal_store(w, (kc_v4u*)(d + 256u * q));
#endif
            }
    }
#else
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 v = proto[i];
    for (uint32_t k = blockIdx.y; k < n; k += gridDim.y) dst[(size_t)k * n16 + i] = v;
#endif
}
void kc_launch_bcast(const uint8_t* proto, uint8_t* dst, size_t bytes, uint32_t n, hipStream_t st) {
    if (n == 0 || bytes == 0) return;
    const size_t n16 = bytes / 16;
    const uint32_t gy = n < 64 ? n : 64;
    hipLaunchKernelGGL(kc_bcast_kernel, dim3((unsigned)(KC_BCAST_V ? (n16 + 1023) / 1024 : (n16 + 255) / 256), gy), dim3(256), 0, st, (const uint4*)proto, (uint4*)dst, n16, n);
}
