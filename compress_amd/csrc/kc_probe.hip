// kc_probe.hip — measurement probes behind the C ABI (include/kcgpu.h "probes"): what THIS box's memory system and PCIe link give
// the access patterns of the encode path, measured in the same process as the bench line that quotes them.
//   kc_probe_table_pattern  the match finders' hash-table traffic: scattered 4-byte read + 4-byte write-back pairs (and plain reads /
//                           plain stores) into per-unit tables of an HBM arena, 8 lanes per unit, 8 units per wave — the pattern of
//                           zstd/enc_fast.go:147-207 (two buckets looked up and overwritten per step) as kc_zstd_match.hip issues it.
//                           Its rates are the ceiling the bench's roofline.floor prices the match finder's transactions at.
//   kc_probe_pcie           pinned H2D / D2H / both at once, and the pageable <-> pinned host copies of the host-buffer entry points
//                           (kc_hostpipe.h) with the context's copy threads: the ceilings of end_to_end.
// Nothing here is on the encode path.
#include "kc_hostpipe.h"

namespace {

__device__ __forceinline__ uint32_t kc_lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0: read + write-back of the same entry (a table probe), 1: read only, 2: store only.  K independent accesses per lane per
// iteration (the match finder looks up two buckets per step).
template <int MODE>
__global__ __launch_bounds__(64) void kc_probe_table_kernel(uint32_t* __restrict__ arena, uint32_t n_tables,
                                                            uint32_t table_words, uint32_t iters, uint32_t* sink) {
    const uint32_t gl = blockIdx.x * 64 + threadIdx.x;
    const uint32_t unit = (gl >> 3) % n_tables;
    uint32_t* tab = arena + (size_t)unit * table_words;
    uint32_t rs = gl * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        // (24 random bits scaled to the table: any table size, not only powers of two — SpeedDefault's tables are 640 KiB per unit)
        const uint32_t i0 = (uint32_t)(((uint64_t)kc_lcg(rs) * table_words) >> 24), i1 = (uint32_t)(((uint64_t)kc_lcg(rs) * table_words) >> 24);
        uint32_t v0 = 0, v1 = 0;
        if (MODE != 2) {
            v0 = *(volatile uint32_t*)(tab + i0);
            v1 = *(volatile uint32_t*)(tab + i1);
        }
        acc += v0 + v1;
        if (MODE != 1) {
            tab[i0] = v0 + it + 1u;
            tab[i1] = v1 + it + 2u;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
hipError_t run_table(uint32_t* arena, uint32_t n_tables, uint32_t table_words, uint32_t waves, uint32_t iters, uint32_t* sink,
                     hipStream_t st, hipEvent_t a, hipEvent_t b, double* req_per_s) {
    hipLaunchKernelGGL((kc_probe_table_kernel<MODE>), dim3(waves), dim3(64), 0, st, arena, n_tables, table_words, iters / 8 + 1, sink);
    hipError_t e = hipEventRecord(a, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((kc_probe_table_kernel<MODE>), dim3(waves), dim3(64), 0, st, arena, n_tables, table_words, iters, sink);
    if ((e = hipEventRecord(b, st)) != hipSuccess) return e;
    if ((e = hipEventSynchronize(b)) != hipSuccess) return e;
    float ms = 0;
    if ((e = hipEventElapsedTime(&ms, a, b)) != hipSuccess) return e;
    *req_per_s = (double)waves * 64.0 * (double)iters * 2.0 / ((double)ms * 1e-3);
    return hipGetLastError();
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" {

kc_status kc_probe_table_pattern(kc_ctx* c, uint32_t n_tables, uint32_t table_bytes, uint32_t waves, uint32_t iters, double* out3) {
    if (!c || !out3 || n_tables == 0 || waves == 0 || iters == 0) return KC_ERR_BAD_ARG;
    if (table_bytes < 1024 || (table_bytes & 3)) { c->err = "table_bytes must be a multiple of 4, at least 1024"; return KC_ERR_BAD_ARG; }
    c->err.clear();
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t* arena = nullptr;
    uint32_t* sink = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    const size_t bytes = (size_t)n_tables * table_bytes;
    hipError_t e = hipMalloc((void**)&arena, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&sink, 64);
    if (e == hipSuccess) e = hipMemsetAsync(arena, 0, bytes, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    const uint32_t words = table_bytes / 4;
    if (e == hipSuccess) e = run_table<0>(arena, n_tables, words, waves, iters, sink, c->stream, a, b, &out3[0]);
    if (e == hipSuccess) e = run_table<1>(arena, n_tables, words, waves, iters, sink, c->stream, a, b, &out3[1]);
    if (e == hipSuccess) e = run_table<2>(arena, n_tables, words, waves, iters, sink, c->stream, a, b, &out3[2]);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (arena) (void)hipFree(arena);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) { c->err = std::string("kc_probe_table_pattern: ") + hipGetErrorString(e); return KC_ERR_HIP; }
    return KC_OK;
}

// out[0] H2D GB/s (pinned, one copy of `bytes`), out[1] D2H GB/s, out[2] / out[3] H2D / D2H GB/s with both running at once,
// out[4] pageable -> pinned host copy GB/s with the context's copy threads, out[5] pinned -> pageable GB/s, out[6] copy threads.
kc_status kc_probe_pcie(kc_ctx* c, uint64_t bytes, double* out7) {
    if (!c || !out7 || bytes < (1u << 20)) return KC_ERR_BAD_ARG;
    c->err.clear();
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t *pin_a = nullptr, *pin_b = nullptr, *dev_a = nullptr, *dev_b = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<uint8_t> pageable;
    hipError_t e = hipHostMalloc((void**)&pin_a, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&pin_b, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&dev_a, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&dev_b, bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&ev[i]);
    if (e == hipSuccess) {
        memset(pin_a, 0x5A, bytes);
        memset(pin_b, 0xA5, bytes);
        try { pageable.assign(bytes, 0x33); } catch (...) { e = hipErrorOutOfMemory; }
    }
    auto timed = [&](bool h2d, bool d2h, double* r_h2d, double* r_d2h) -> hipError_t {
        hipError_t q = hipSuccess;
        for (int rep = 0; rep < 2 && q == hipSuccess; rep++) {  // the first pass warms the queues; the second is reported
            if (h2d) { q = hipEventRecord(ev[0], s1); if (q == hipSuccess) q = hipMemcpyAsync(dev_a, pin_a, bytes, hipMemcpyHostToDevice, s1); if (q == hipSuccess) q = hipEventRecord(ev[1], s1); }
            if (d2h && q == hipSuccess) { q = hipEventRecord(ev[2], s2); if (q == hipSuccess) q = hipMemcpyAsync(pin_b, dev_b, bytes, hipMemcpyDeviceToHost, s2); if (q == hipSuccess) q = hipEventRecord(ev[3], s2); }
            if (q == hipSuccess) q = hipStreamSynchronize(s1);
            if (q == hipSuccess) q = hipStreamSynchronize(s2);
        }
        float ms = 0;
        if (h2d && q == hipSuccess) { q = hipEventElapsedTime(&ms, ev[0], ev[1]); *r_h2d = (double)bytes / ((double)ms * 1e-3) / 1e9; }
        if (d2h && q == hipSuccess) { q = hipEventElapsedTime(&ms, ev[2], ev[3]); *r_d2h = (double)bytes / ((double)ms * 1e-3) / 1e9; }
        return q;
    };
    for (int i = 0; i < 7; i++) out7[i] = 0;
    double dummy = 0;
    if (e == hipSuccess) e = timed(true, false, &out7[0], &dummy);
    if (e == hipSuccess) e = timed(false, true, &dummy, &out7[1]);
    if (e == hipSuccess) e = timed(true, true, &out7[2], &out7[3]);
    if (e == hipSuccess) {
        const int T = host_copy_threads(c);
        out7[6] = T;
        parallel_memcpy(pin_a, pageable.data(), (size_t)bytes, T);
        double t0 = now_s();
        parallel_memcpy(pin_a, pageable.data(), (size_t)bytes, T);
        out7[4] = (double)bytes / (now_s() - t0) / 1e9;
        parallel_memcpy(pageable.data(), pin_b, (size_t)bytes, T);
        t0 = now_s();
        parallel_memcpy(pageable.data(), pin_b, (size_t)bytes, T);
        out7[5] = (double)bytes / (now_s() - t0) / 1e9;
    }
    for (int i = 0; i < 4; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (s1) (void)hipStreamDestroy(s1);
    if (s2) (void)hipStreamDestroy(s2);
    if (pin_a) (void)hipHostFree(pin_a);
    if (pin_b) (void)hipHostFree(pin_b);
    if (dev_a) (void)hipFree(dev_a);
    if (dev_b) (void)hipFree(dev_b);
    if (e != hipSuccess) { c->err = std::string("kc_probe_pcie: ") + hipGetErrorString(e); return KC_ERR_HIP; }
    return KC_OK;
}

}  // extern "C"
