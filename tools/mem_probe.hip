// tools/mem_probe.hip — microbenchmark of the match finders' table access pattern on MI355X:
// scattered 4-byte accesses into per-unit 128 KiB tables (8 lanes per unit, 8 units per wave),
// in the instruction flavours a table probe could use.  Prints requests/s per flavour and arena size.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mem_probe tools/mem_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { RD4 = 0, RD4_SC1, RD4_NT, RD4_ST4, EXCH, RD8, RD16, RD4_ST4_SC1, RD4_SYS, ST4_ONLY, RD4_ST64, ST4_BOTH_ONLY, NVAR };
static const char* NAMES[NVAR] = {"rd4", "rd4_sc1", "rd4_nt", "rd4+st4", "atomic_exch", "rd8", "rd16", "rd4sc1+st4sc0sc1", "rd4_sys", "st4_only", "rd4+st64B_line", "st4+sibling_only"};

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int V, int K>
__global__ __launch_bounds__(64) void probe(uint32_t* __restrict__ arena, uint32_t n_tables, uint32_t iters, uint32_t* sink) {
    const uint32_t gl = blockIdx.x * 64 + threadIdx.x;
    const uint32_t unit = (gl >> 3) % n_tables;
    uint32_t* tab = arena + (size_t)unit * 32768u;
    uint32_t rs = gl * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t v[K];
        uint32_t idx[K];
#pragma unroll
        for (int k = 0; k < K; k++) idx[k] = lcg(rs) & 32767u;
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint32_t* p = tab + idx[k];
            if (V == RD4 || V == RD4_ST4 || V == RD4_ST64) v[k] = *(volatile uint32_t*)p;
            else if (V == RD4_SC1 || V == RD4_ST4_SC1) v[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (V == RD4_SYS) v[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else if (V == RD4_NT) v[k] = __builtin_nontemporal_load(p);
            else if (V == EXCH) v[k] = __hip_atomic_exchange(p, idx[k] + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (V == RD8) { uint64_t w = *(volatile uint64_t*)(tab + (idx[k] & ~1u)); v[k] = (uint32_t)w ^ (uint32_t)(w >> 32); }
            else if (V == RD16) { const uint4 w = *(const uint4*)(tab + (idx[k] & ~3u)); v[k] = w.x ^ w.y ^ w.z ^ w.w; }
            else v[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            acc += v[k];
            if (V == RD4_ST4) tab[idx[k]] = v[k] + 1u;
            if (V == RD4_ST4_SC1) __hip_atomic_store(tab + idx[k], v[k] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (V == ST4_ONLY) tab[idx[k]] = idx[k] + it;
            if (V == ST4_BOTH_ONLY) { tab[idx[k]] = idx[k] + it; tab[idx[k] ^ 8u] = idx[k] + it + 1u; }
            if (V == RD4_ST64) {  // rewrite the whole 64-byte line
                uint4* q = (uint4*)(tab + (idx[k] & ~15u));
                const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
                q[0] = make_uint4(a.x + 1u, a.y, a.z, a.w); q[1] = b; q[2] = c; q[3] = d;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int V, int K>
static double run(uint32_t* arena, uint32_t n_tables, uint32_t waves, uint32_t iters, uint32_t* sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<V, K>), dim3(waves), dim3(64), 0, 0, arena, n_tables, iters / 8, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<V, K>), dim3(waves), dim3(64), 0, 0, arena, n_tables, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char** argv) {
    const uint32_t waves = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
    const uint32_t iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 512;
    const uint32_t vmask = argc > 3 ? (uint32_t)strtoul(argv[3], nullptr, 0) : 0xFFFFFFFFu;
    const uint32_t max_tables = 32768;
    uint32_t* arena; uint32_t* sink;
    CK(hipMalloc(&arena, (size_t)max_tables * 131072));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(arena, 0, (size_t)max_tables * 131072));
    const uint32_t sizes[] = {32768, 8192, 2048, 512, 128};
    printf("waves %u, iters %u, K=2 independent accesses per lane per iteration\n", waves, iters);
    printf("%-18s", "flavour");
    for (uint32_t nt : sizes) printf(" %9uMiB", nt / 8);
    printf("   (G requests/s)\n");
    for (int v = 0; v < NVAR; v++) {
        if (!((vmask >> v) & 1u)) continue;
        printf("%-18s", NAMES[v]);
        for (uint32_t nt : sizes) {
            double ms = 0;
            switch (v) {
            case RD4: ms = run<RD4, 2>(arena, nt, waves, iters, sink); break;
            case RD4_SC1: ms = run<RD4_SC1, 2>(arena, nt, waves, iters, sink); break;
            case RD4_NT: ms = run<RD4_NT, 2>(arena, nt, waves, iters, sink); break;
            case RD4_ST4: ms = run<RD4_ST4, 2>(arena, nt, waves, iters, sink); break;
            case EXCH: ms = run<EXCH, 2>(arena, nt, waves, iters, sink); break;
            case RD8: ms = run<RD8, 2>(arena, nt, waves, iters, sink); break;
            case RD16: ms = run<RD16, 2>(arena, nt, waves, iters, sink); break;
            case RD4_ST4_SC1: ms = run<RD4_ST4_SC1, 2>(arena, nt, waves, iters, sink); break;
            case RD4_SYS: ms = run<RD4_SYS, 2>(arena, nt, waves, iters, sink); break;
            case ST4_ONLY: ms = run<ST4_ONLY, 2>(arena, nt, waves, iters, sink); break;
            case RD4_ST64: ms = run<RD4_ST64, 2>(arena, nt, waves, iters, sink); break;
            case ST4_BOTH_ONLY: ms = run<ST4_BOTH_ONLY, 2>(arena, nt, waves, iters, sink); break;
            }
            const double req = (double)waves * 64.0 * iters * 2.0;
            printf(" %12.1f", req / (ms * 1e-3) / 1e9);
            fflush(stdout);
        }
        printf("\n");
    }
    // K sweep for the plain load: memory-level parallelism per lane
    printf("rd4 @4GiB, waves sweep (K=2):");
    for (uint32_t w : {1024u, 2048u, 4096u, 8192u}) {
        double ms = run<RD4, 2>(arena, 32768, w, iters, sink);
        printf("  %u waves: %.1f", w, (double)w * 64.0 * iters * 2.0 / (ms * 1e-3) / 1e9);
    }
    printf("\nrd4 @4GiB, K=8:");
    {
        double ms = run<RD4, 8>(arena, 32768, waves, iters / 4, sink);
        printf(" %.1f G/s\n", (double)waves * 64.0 * (iters / 4) * 8.0 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
