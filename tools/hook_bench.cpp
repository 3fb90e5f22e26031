// hook_bench.cpp — the s2.WriterCustomEncoder hook (kc_s2_encode_block, s2/writer.go:1053-1064) under concurrent callers, through the C ABI
// as a cgo binding would call it: N native threads (one per goroutine of s2.Writer's block pool, writer.go:455-460), each encoding
// 64 KiB JSON blocks one call at a time.  A caller that gets -1 encodes its block with the built-in encoder like s2.Writer does
// (writer.go:455-460) — here the reference's own amd64 assembly encoder out of oracle/_ref/libs2ref.so (dlopen; a checker library,
// used by this measurement tool only).  Three arrangements per caller count: the built-in encoder alone (no hook wired), the hook
// with the library's default host-first rule, and the hook forced to the device (KC_OPT_S2_HOOK_HOST_FIRST 0: round 4's behaviour).
// Prints one JSON object.
//   g++ -O2 -std=c++17 -I include tools/hook_bench.cpp -o /tmp/hook_bench -L compress_amd -lkcgpu -Wl,-rpath,$PWD/compress_amd -lpthread -ldl
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "kcgpu.h"

typedef int64_t (*builtin_fn)(int level, uint8_t* dst, uint64_t dst_len, const uint8_t* src, uint64_t n);

int main(int argc, char** argv) {
    const uint32_t nblk = 2048, bsz = 65536;
    const char* refpath = argc > 1 ? argv[1] : "oracle/_ref/libs2ref.so";
    void* so = dlopen(refpath, RTLD_NOW);
    builtin_fn builtin = so ? (builtin_fn)dlsym(so, "s2ref_encode_block") : nullptr;
    if (!builtin) { fprintf(stderr, "no built-in encoder (%s): %s\n", refpath, dlerror()); return 1; }
    std::vector<uint8_t> data((size_t)nblk * bsz);
    if (kc_corpus_fill('J', 0x5EED0001, 0, nblk, bsz, data.data(), 8) != KC_OK) { fprintf(stderr, "corpus\n"); return 1; }
    printf("{\"block_bytes\": %u, \"blocks\": %u", bsz, nblk);
    for (int nthr : {1, 4, 16, 64}) {
        for (int mode = 0; mode < 3; mode++) {  // 0: built-in only, 1: hook (host first, default), 2: hook, every caller to the device
            kc_ctx* c = nullptr;
            if (kc_ctx_create(&c, 0, nullptr) != KC_OK) { fprintf(stderr, "ctx\n"); return 1; }
            if (mode == 2) kc_ctx_set_option(c, KC_OPT_S2_HOOK_HOST_FIRST, 0);
            std::vector<uint8_t> warm(bsz + 64);
            if (mode) {
                kc_ctx_set_option(c, KC_OPT_S2_HOOK_HOST_FIRST, 0);
                kc_s2_encode_block(c, warm.data(), warm.size(), data.data(), bsz);  // staging and lanes exist before the clock starts
                kc_ctx_set_option(c, KC_OPT_S2_HOOK_HOST_FIRST, mode == 2 ? 0 : -1);
            }
            // the 2048 blocks eight times over (1 GiB) so that a device batch in flight at the end is not a visible share of the run;
            // the device-only arrangement at 23 MB/s per caller gets fewer
            const uint32_t todo = mode == 2 ? (nthr == 1 ? 256 : nblk) : 8 * nblk;
            std::atomic<uint32_t> next{0};
            std::atomic<uint64_t> outb{0};
            std::atomic<int> bad{0}, host{0};
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < nthr; t++)
                th.emplace_back([&] {
                    std::vector<uint8_t> dst(bsz + 64);
                    for (;;) {
                        const uint32_t i = next.fetch_add(1);
                        if (i >= todo) return;
                        const uint8_t* src = data.data() + (size_t)(i % nblk) * bsz;
                        int64_t r = mode ? kc_s2_encode_block(c, dst.data(), dst.size(), src, bsz) : -1;
                        if (r < 0) { r = builtin(0, dst.data(), dst.size(), src, bsz); host++; }
                        if (r <= 0) bad++;
                        else outb += (uint64_t)r;
                    }
                });
            for (auto& t : th) t.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const char* nm = mode == 0 ? "builtin" : (mode == 1 ? "hook_host_first" : "hook_device_only");
            printf(", \"%s_%dcallers_MBps\": %.1f", nm, nthr, todo * (double)bsz / dt / 1e6);
            if (mode == 1) printf(", \"%s_%dcallers_on_host\": %d", nm, nthr, host.load());
            if (bad) printf(", \"%s_%dcallers_failed\": %d", nm, nthr, bad.load());
            kc_ctx_destroy(c);
        }
    }
    printf("}\n");
    return 0;
}
