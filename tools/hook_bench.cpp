// hook_bench.cpp — the s2.WriterCustomEncoder hook (kc_s2_encode_block, s2/writer.go:1053-1064) under concurrent callers, through the C ABI
// as a cgo binding would call it: N native threads (one per goroutine of s2.Writer's block pool, writer.go:455-460), each encoding
// 64 KiB JSON blocks one call at a time.  Prints MB/s per caller count as one JSON object.
//   g++ -O2 -std=c++17 -I include tools/hook_bench.cpp -o /tmp/hook_bench -L compress_amd -lkcgpu -Wl,-rpath,$PWD/compress_amd -lpthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "kcgpu.h"

int main(int argc, char** argv) {
    const uint32_t nblk = 2048, bsz = 65536;
    const int lanes = argc > 1 ? atoi(argv[1]) : 4;
    std::vector<uint8_t> data((size_t)nblk * bsz);
    if (kc_corpus_fill('J', 0x5EED0001, 0, nblk, bsz, data.data(), 8) != KC_OK) { fprintf(stderr, "corpus\n"); return 1; }
    printf("{\"lanes\": %d", lanes);
    for (int nthr : {1, 4, 16, 64}) {
        kc_ctx* c = nullptr;
        if (kc_ctx_create(&c, 0, nullptr) != KC_OK) { fprintf(stderr, "ctx\n"); return 1; }
        kc_ctx_set_option(c, KC_OPT_S2_HOOK_LANES, lanes);
        std::vector<uint8_t> warm(bsz + 64);
        kc_s2_encode_block(c, warm.data(), warm.size(), data.data(), bsz);
        const uint32_t todo = nthr == 1 ? 256 : nblk;
        std::atomic<uint32_t> next{0};
        std::atomic<uint64_t> outb{0};
        std::atomic<int> bad{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; t++)
            th.emplace_back([&] {
                std::vector<uint8_t> dst(bsz + 64);
                for (;;) {
                    const uint32_t i = next.fetch_add(1);
                    if (i >= todo) return;
                    const int64_t r = kc_s2_encode_block(c, dst.data(), dst.size(), data.data() + (size_t)i * bsz, bsz);
                    if (r <= 0) bad++;
                    else outb += (uint64_t)r;
                }
            });
        for (auto& t : th) t.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t calls = 0, batches = 0;
        kc_s2_hook_stats(c, &calls, &batches);
        printf(", \"hook_%dcallers_MBps\": %.1f, \"hook_%dcallers_blocks_per_batch\": %.2f", nthr, todo * (double)bsz / dt / 1e6, nthr, batches ? (double)calls / batches : 0.0);
        if (bad) printf(", \"hook_%dcallers_failed\": %d", nthr, bad.load());
        kc_ctx_destroy(c);
    }
    printf("}\n");
    return 0;
}
