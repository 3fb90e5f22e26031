// hipemu host API — the HIP runtime calls the product's host code (compress_amd/csrc/kc_*.cpp) makes, as synchronous CPU
// stand-ins, so that the WHOLE library (C ABI + batching + scratch sizing + kernels) can be built for the wave emulator and run
// under AddressSanitizer.  TEST INFRASTRUCTURE ONLY (tools/emu_host_check.py); nothing under compress_amd/ includes this.
//
// What it models and what it does not:
//   * hipMalloc / hipHostMalloc are malloc blocks of their own, so an out-of-bounds access of ANY device buffer the host code
//     sized is an AddressSanitizer report with the kernel's source line — on the hardware the same access lands in a neighbouring
//     allocation or faults, depending on the address layout of the day;
//   * fresh device memory is POISONED (HIPEMU_POISON=<byte>, default 0xA7; "rand" for pseudo-random bytes): a kernel that reads a
//     buffer nothing wrote does not see the zeros a freshly booted GPU often has;
//   * streams and events are synchronous (every call completes before it returns): ordering bugs between streams are NOT seen.
#pragma once
#include <mutex>
#include <chrono>

enum { hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
#define hipEventDisableTiming 2

struct hipemuEvent { double t_ms; };
typedef hipemuEvent* hipEvent_t;
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    size_t maxSharedMemoryPerMultiProcessor;
    int clockRate;
};

namespace hipemu {
inline std::recursive_mutex& launch_mutex() { static std::recursive_mutex m; return m; }
inline size_t& device_bytes_in_use() { static size_t n = 0; return n; }
inline std::mutex& alloc_mutex() { static std::mutex m; return m; }
inline size_t device_capacity() {
    static size_t cap = 0;
    if (!cap) { const char* e = getenv("HIPEMU_DEVICE_MIB"); cap = (size_t)(e ? atol(e) : 4096) << 20; }
    return cap;
}
inline void poison(void* p, size_t n) {
    static int mode = -2;
    if (mode == -2) { const char* e = getenv("HIPEMU_POISON"); mode = !e ? 0xA7 : (strcmp(e, "rand") == 0 ? -1 : (int)strtol(e, nullptr, 0) & 0xFF); }
    if (mode >= 0) { memset(p, mode, n); return; }
    static uint64_t s = 0x9E3779B97F4A7C15ull;
    uint8_t* b = (uint8_t*)p;
    for (size_t i = 0; i < n; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; b[i] = (uint8_t)(s >> 56); }
}
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// every allocation carries its size in a 64-byte header in front of the block the caller sees (the capacity accounting of hipFree)
inline hipError_t dev_alloc(void** p, size_t n) {
    std::lock_guard<std::mutex> g(alloc_mutex());
    if (device_bytes_in_use() + n > device_capacity()) { *p = nullptr; return hipErrorOutOfMemory; }
    char* raw = nullptr;
    if (posix_memalign((void**)&raw, 256, n + 256) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
    *(size_t*)raw = n;
    device_bytes_in_use() += n;
    poison(raw + 256, n);
    *p = raw + 256;
    return hipSuccess;
}
inline hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> g(alloc_mutex());
    char* raw = (char*)p - 256;
    device_bytes_in_use() -= *(size_t*)raw;
    free(raw);
    return hipSuccess;
}
}  // namespace hipemu

// Kernels that give each G-lane group of a wave its own work item let the groups diverge (hipemu::set_group): the group width by
// kernel name, as tools/hipemu/kcemu.cpp sets it around its launches.
namespace hipemu {
inline unsigned group_for(const char* k) {
    if (strstr(k, "kc_zbetter_match_grp") || strstr(k, "kc_s2_best_kernel")) return 16;
    if (strstr(k, "kc_zfast_match_grp") || strstr(k, "kc_zdfast_match_grp") || strstr(k, "kc_s2_encode_kernel") || strstr(k, "kc_s2_decode_kernel")) return 8;
    if (strstr(k, "kc_xxh64")) return 4;
    return 64;
}
}  // namespace hipemu
// kernel launches from several host threads (hook lanes, job threads): the emulator has one scheduler
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                         \
    do {                                                                                   \
        std::lock_guard<std::recursive_mutex> g__(hipemu::launch_mutex());                 \
        hipemu::set_group(hipemu::group_for(#kernel));                                     \
        hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); });           \
        hipemu::set_group(64);                                                             \
    } while (0)

static inline const char* hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : e == hipErrorNotReady ? "not ready" : "hipemu error";
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
namespace hipemu {
inline int device_count() { static int n = 0; if (!n) { const char* e = getenv("HIPEMU_DEVICES"); n = e ? atoi(e) : 1; if (n < 1) n = 1; } return n; }
}  // namespace hipemu
// (HIPEMU_DEVICES=N: N "devices" that are all this process's memory — one rank per device in tools/emu_bench_rank.py)
static inline hipError_t hipGetDeviceCount(int* n) { *n = hipemu::device_count(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d >= 0 && d < hipemu::device_count() ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hipemu");
    strcpy(p->gcnArchName, "hipemu-wave64");
    p->totalGlobalMem = hipemu::device_capacity();
    p->multiProcessorCount = 256;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->clockRate = 2400000;
    return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) {
    std::lock_guard<std::mutex> g(hipemu::alloc_mutex());
    *tot = hipemu::device_capacity();
    *fr = *tot - hipemu::device_bytes_in_use();
    return hipSuccess;
}
static inline hipError_t hipMalloc(void** p, size_t n) { return hipemu::dev_alloc(p, n); }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipemu::dev_alloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { return hipemu::dev_free(p); }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) {
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    hipemu::poison(*p, n);
    return hipSuccess;
}
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
// (kc_roll.cpp asks whether a caller's buffer is page-locked; in the emulator nothing is: ordinary memory yields an error, as on the device)
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t*, const void*) { return hipErrorInvalidValue; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent{0.0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemuEvent{0.0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t_ms = hipemu::now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
