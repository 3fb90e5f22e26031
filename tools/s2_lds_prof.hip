// s2_lds_prof.hip — per-phase shader clocks of the S2 LDS-table kernel's fused step (diagnostics, GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKC_S2_PROF -I compress_amd/csrc tools/s2_lds_prof.hip -o /tmp/s2prof && /tmp/s2prof block.bin
// One block (<= 64 KiB, read from the file) through kc_s2_encode_lds_kernel<0, true, 2>; prints the kernel time and, per phase, clocks and events.
#include "../compress_amd/csrc/kc_s2_lds.hip"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: s2prof block.bin [spec_w0]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    std::vector<uint8_t> blk(65536);
    const size_t n = fread(blk.data(), 1, blk.size(), f);
    fclose(f);
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    uint8_t *d_src, *d_stage;
    uint64_t *d_off, *d_soff;
    uint32_t* d_size;
    const uint64_t off[2] = {0, n}, soff[2] = {0, 80000};
    hipMalloc(&d_src, n + 64); hipMalloc(&d_stage, 160000); hipMalloc(&d_off, 16); hipMalloc(&d_soff, 16); hipMalloc(&d_size, 8);
    hipMemcpy(d_src, blk.data(), n, hipMemcpyHostToDevice);
    hipMemcpy(d_off, off, 16, hipMemcpyHostToDevice);
    hipMemcpy(d_soff, soff, 16, hipMemcpyHostToDevice);
    KcS2Params P;
    memset(&P, 0, sizeof(P));
    P.src = d_src; P.blk_off = d_off; P.stage_off = d_soff; P.stage = d_stage; P.out_size = d_size; P.n_blocks = 1; P.level = 0; P.spec_w0 = mode;
    for (int rep = 0; rep < 3; rep++) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        kc_launch_s2_encode_lds(P, true, false, nullptr);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        uint32_t sz = 0;
        hipMemcpy(&sz, d_size, 4, hipMemcpyDeviceToHost);
        printf("mode %d: %zu -> %u bytes, %.3f ms\n", mode, n, sz, ms);
    }
#ifdef KC_S2_PROF
    unsigned long long pr[16];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(kc_s2_prof), sizeof(pr));
    const char* nm[8] = {"loop overhead", "probe step", "match ends", "emit (repeat)", "emit (lit+copy)", "immediate test", "immediate end", "-"};
    unsigned long long tot = 0;
    for (int k = 0; k < 8; k++) tot += pr[k];
    for (int k = 0; k < 7; k++) printf("%-18s %10llu clocks %8llu events %8.1f clocks/event %5.1f %%\n", nm[k], pr[k], pr[8 + k], pr[8 + k] ? (double)pr[k] / pr[8 + k] : 0.0, 100.0 * pr[k] / (tot ? tot : 1));
    printf("total %llu clocks\n", tot);
#endif
    return 0;
}
