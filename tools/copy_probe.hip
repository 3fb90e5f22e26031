// tools/copy_probe.hip — what does the MI355X give a "hash the unit and copy it into its raw frame" pass?
// 32768 units of 128 KiB are copied into frames (12-byte head, two 64 KiB payloads 3 bytes apart, 4-byte tail) by
// several access layouts, with and without the XXH64 rounds, so that the layout's share and the hash's share of
// kc_xxh64_fin_kernel's 2.04 ms per 4 GiB can be told apart.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
__device__ __forceinline__ uint64_t xrol(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t input) { return xrol(acc + input * XP2, 31) * XP1; }

struct __attribute__((packed)) u128s { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 ld128u(const uint8_t* p) { const u128s t = *(const u128s*)p; return make_uint4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void st128u(uint8_t* p, const uint4 v) { u128s t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(u128s*)p = t; }

constexpr uint32_t USZ = 128u << 10, BSZ = 64u << 10, FSZ = 9 + 2 * (3 + BSZ) + 4;

// P0/P1: plain grid-stride 16-byte copy of the whole arena, dst shifted by `mis` bytes
__global__ __launch_bounds__(256) void k_flat(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n16, int mis) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = *(const uint4*)(src + (i << 4));
        if (mis) st128u(dst + (i << 4) + mis, v); else *(uint4*)(dst + (i << 4)) = v;
    }
}
// P0b: the same, four loads in flight per lane
__global__ __launch_bounds__(256) void k_flat4(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n16, int mis) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = *(const uint4*)(src + (i << 4)), b = *(const uint4*)(src + ((i + stride) << 4));
        const uint4 c = *(const uint4*)(src + ((i + 2 * stride) << 4)), d = *(const uint4*)(src + ((i + 3 * stride) << 4));
        st128u(dst + (i << 4) + mis, a); st128u(dst + ((i + stride) << 4) + mis, b);
        st128u(dst + ((i + 2 * stride) << 4) + mis, c); st128u(dst + ((i + 3 * stride) << 4) + mis, d);
    }
    for (; i < n16; i += stride) st128u(dst + (i << 4) + mis, *(const uint4*)(src + (i << 4)));
}

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// N: flat copies with non-temporal loads / stores (MODE bit 0: nt load, bit 1: nt store), grid-stride, U loads in flight per lane
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_flat_nt(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        v4u v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = (MODE & 1) ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) { if (MODE & 2) __builtin_nontemporal_store(v[k], dst + i + k * stride); else dst[i + k * stride] = v[k]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// C: every workgroup copies ONE contiguous chunk of the arena (chunk = n16 / gridDim.x words), U loads in flight per lane
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_chunk(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t n16) {
    const uint64_t per = n16 / gridDim.x;
    const v4u* s = src + (uint64_t)blockIdx.x * per;
    v4u* d = dst + (uint64_t)blockIdx.x * per;
    for (uint64_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        v4u v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = (MODE & 1) ? __builtin_nontemporal_load(s + i + k * 256) : s[i + k * 256];
#pragma unroll
        for (int k = 0; k < U; k++) { if (MODE & 2) __builtin_nontemporal_store(v[k], d + i + k * 256); else d[i + k * 256] = v[k]; }
    }
}

// Q: the quad layout of kc_xxh64_fin_kernel: 4 lanes per unit, 16 units per wave, K loads of 16 bytes in flight per lane.
// HASH: the XXH64 rounds (each lane hashes its own words: the same ALU work, no quad permutes).  FRAME: the frame layout (else a
// plain unit -> unit copy, aligned).
template <int K, bool HASH, bool FRAME, bool PIPE>
__global__ __launch_bounds__(256) void k_quad(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t n_units, uint64_t* sink) {
    const uint32_t gt = blockIdx.x * 256 + threadIdx.x;
    const uint32_t u = gt >> 2;
    const int a = (int)(gt & 3);
    if (u >= n_units) return;
    const uint8_t* q16 = src + (size_t)u * USZ + 16 * a;
    uint8_t* d16 = dst + (FRAME ? (size_t)u * FSZ : (size_t)u * USZ) + 16 * a;
    uint64_t v = a;
    const uint32_t lines = USZ >> 6;
    uint4 c[K], n[K];
    if (PIPE) {
#pragma unroll
        for (int k = 0; k < K; k++) c[k] = ld128u(q16 + ((size_t)k << 6));
    }
    for (uint32_t i = 0; i < lines; i += K) {
        if (PIPE) {
            if (i + K < lines) {
#pragma unroll
                for (int k = 0; k < K; k++) n[k] = ld128u(q16 + ((size_t)(i + K + k) << 6));
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) c[k] = ld128u(q16 + ((size_t)(i + k) << 6));
        }
        const uint32_t x = i << 6;
        const int shift = FRAME ? (x < BSZ ? 12 : 15) : 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (FRAME) st128u(d16 + x + shift + (k << 6), c[k]); else *(uint4*)(d16 + x + (k << 6)) = c[k];
        }
        if (HASH) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                v = xround(v, (uint64_t)c[k].x | ((uint64_t)c[k].y << 32));
                v = xround(v, (uint64_t)c[k].z | ((uint64_t)c[k].w << 32));
            }
        }
        if (PIPE) {
#pragma unroll
            for (int k = 0; k < K; k++) c[k] = n[k];
        }
    }
    if (HASH && v == 0x1234567ull) sink[u] = v;
}

// W: one wave instruction = 1 KiB of ONE unit (64 lanes x 16 bytes), UPW units per wave, K such loads in flight per unit-slot.
// With HASH the chunk goes through LDS to the unit's four hash lanes (lane 4k+a of the wave hashes accumulator a of unit-slot k).
template <int UPW, bool HASH, bool FRAME>
__global__ __launch_bounds__(64) void k_wide(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t n_units, uint64_t* sink) {
    constexpr int STR = 1024 + 16;
    __shared__ __attribute__((aligned(16))) uint8_t lds[HASH ? UPW * STR : 16];
    const int lane = threadIdx.x;
    const uint32_t u0 = blockIdx.x * UPW;
    if (u0 >= n_units) return;
    const int hk = lane >> 2, ha = lane & 3;  // hash role: unit-slot hk (< UPW), accumulator ha
    uint64_t v = ha;
    for (uint32_t x = 0; x < USZ; x += 1024) {
        uint4 c[UPW];
#pragma unroll
        for (int k = 0; k < UPW; k++) c[k] = *(const uint4*)(src + (size_t)(u0 + k) * USZ + x + 16 * lane);
        const int shift = FRAME ? (x < BSZ ? 12 : 15) : 0;
#pragma unroll
        for (int k = 0; k < UPW; k++) {
            uint8_t* d = dst + (FRAME ? (size_t)(u0 + k) * FSZ : (size_t)(u0 + k) * USZ) + x + shift + 16 * lane;
            if (FRAME) st128u(d, c[k]); else *(uint4*)d = c[k];
        }
        if (HASH) {
#pragma unroll
            for (int k = 0; k < UPW; k++) *(uint4*)(lds + k * STR + 16 * lane) = c[k];
            __builtin_amdgcn_wave_barrier();
            if (hk < UPW) {
                const uint8_t* p = lds + hk * STR + 8 * ha;
#pragma unroll 8
                for (int j = 0; j < 32; j++) v = xround(v, *(const uint64_t*)(p + 32 * j));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (HASH && v == 0x1234567ull) sink[u0] = v;
}

// W2: like W, but the wave's hash lanes are all busy: the wave owns 16 unit-slots, loads them in G groups of UPW = 16 / G ... (kept simple: UPW = 16)

int main(int argc, char** argv) {
    const uint32_t n_units = argc > 1 ? (uint32_t)atoi(argv[1]) : 32768u;
    const size_t in_bytes = (size_t)n_units * USZ, out_bytes = (size_t)n_units * FSZ + 64;
    uint8_t *src, *dst; uint64_t* sink;
    CK(hipMalloc(&src, in_bytes)); CK(hipMalloc(&dst, out_bytes + 4096)); CK(hipMalloc(&sink, (size_t)n_units * 8));
    CK(hipMemset(src, 0x5a, in_bytes)); CK(hipMemset(dst, 0, out_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        std::vector<float> ms;
        for (int it = 0; it < 7; it++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (it >= 2) ms.push_back(t);
        }
        CK(hipGetLastError());
        std::sort(ms.begin(), ms.end());
        const double med = ms[ms.size() / 2];
        printf("%-34s %8.3f ms  (min %.3f)  %6.2f TB/s read+write\n", name, med, ms[0], 2.0 * in_bytes / med / 1e9);
        fflush(stdout);
    };
    const uint64_t n16 = in_bytes >> 4;
    for (int grid : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, sizeof nm, "flat aligned grid=%d", grid);
        run(nm, [&] { hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, src, dst, n16, 0); });
        snprintf(nm, sizeof nm, "flat dst+12 grid=%d", grid);
        run(nm, [&] { hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, src, dst, n16, 12); });
        snprintf(nm, sizeof nm, "flat4 dst+12 grid=%d", grid);
        run(nm, [&] { hipLaunchKernelGGL(k_flat4, dim3(grid), dim3(256), 0, 0, src, dst, n16, 12); });
    }
#define FNT(M, U, G) run("flat nt mode=" #M " U=" #U " grid=" #G, [&] { hipLaunchKernelGGL((k_flat_nt<M, U>), dim3(G), dim3(256), 0, 0, (const v4u*)src, (v4u*)dst, n16); })
#define CHK(M, U, G) run("chunk mode=" #M " U=" #U " grid=" #G, [&] { hipLaunchKernelGGL((k_chunk<M, U>), dim3(G), dim3(256), 0, 0, (const v4u*)src, (v4u*)dst, n16); })
    FNT(0, 4, 1024); FNT(0, 4, 2048); FNT(0, 4, 4096); FNT(0, 8, 2048);
    FNT(1, 4, 2048); FNT(2, 4, 2048); FNT(3, 4, 2048); FNT(3, 4, 1024); FNT(3, 8, 2048); FNT(3, 4, 8192);
    CHK(0, 4, 1024); CHK(0, 4, 2048); CHK(0, 4, 4096); CHK(0, 4, 16384); CHK(3, 4, 2048); CHK(3, 4, 16384); CHK(3, 8, 4096); CHK(0, 8, 65536); CHK(3, 4, 65536);
    const dim3 gq((n_units * 4 + 255) / 256);
#define QUAD(K, H, F, P) run("quad K=" #K " hash=" #H " frame=" #F " pipe=" #P, [&] { hipLaunchKernelGGL((k_quad<K, H, F, P>), gq, dim3(256), 0, 0, src, dst, n_units, sink); })
    QUAD(4, false, false, false);
    QUAD(4, false, true, false);
    QUAD(4, true, true, false);
    QUAD(4, true, true, true);
    QUAD(8, false, true, false);
    QUAD(8, true, true, false);
    QUAD(8, true, true, true);
    QUAD(16, false, true, false);
    QUAD(16, true, true, false);
    QUAD(4, true, false, false);  // hash + aligned unit copy
    {   // read-only hash (no stores): the kernel's ALU/latency floor — emulate by K=4 with dst == a tiny region? (skipped: kc_xxh64_kernel is 0.79 ms)
    }
#define WIDE(U, H, F) run("wide UPW=" #U " hash=" #H " frame=" #F, [&] { hipLaunchKernelGGL((k_wide<U, H, F>), dim3((n_units + U - 1) / U), dim3(64), 0, 0, src, dst, n_units, sink); })
    WIDE(4, false, false);
    WIDE(4, false, true);
    WIDE(8, false, true);
    WIDE(16, false, true);
    WIDE(4, true, true);
    WIDE(8, true, true);
    WIDE(16, true, true);
    return 0;
}
